R=$GRAFT_REPO_ROOT; A=$R/gpurun_out/r5s8; mkdir -p $A; cd $R
V=$R/tools/scratch/variants
python -m pytest tests/test_sepconv_frames8_gpu.py tests/test_hip_ops_gpu.py tests/test_ws_timeout_gpu.py -x -q -k "sepconv or frames8 or pair or wait" 2>&1 | tail -3
python tools/frames8_time.py 8 256 448 fwd_six,fwd_frames8,fwd_frames8_unit16 2>&1 | grep op | cut -c1-220
SAVFI_HIP_LIB=$V/libsavfi_trace.so python tools/ws_trace_fwd.py 8 1 > $A/fwd_trace.txt 2>&1
tail -14 $A/fwd_trace.txt | cut -c1-400
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), {k:(round(v["avg_us"],1), round(v["min_us"],1)) for k,v in d["kernels"].items()}, round(d["roofline"]["frac"],4))'
$B 2>/dev/null | python -c "$P" default
$B 2>/dev/null | python -c "$P" default
