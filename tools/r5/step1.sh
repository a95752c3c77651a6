R=$GRAFT_REPO_ROOT; A=$R/gpurun_out/r5s1; mkdir -p $A; cd $R
V=$R/tools/scratch/variants
python -m pytest tests/test_sepconv_frames8_gpu.py tests/test_ws_timeout_gpu.py tests/test_hip_ops_gpu.py -x -q -k "sepconv or frames8 or wait" 2>&1 | tail -5 > $A/tests.txt
python tools/frames8_time.py 8 256 448 bwd_frames8,bwd_frames8_unit16,fwd_frames8_unit16 > $A/time_b8.txt 2>&1
python tools/frames8_time.py 4 256 448 bwd_frames8_unit16,fwd_frames8_unit16 > $A/time_b4.txt 2>&1
SAVFI_HIP_LIB=$V/libsavfi_trace.so python tools/ws_trace.py 8 f8 3 > $A/trace_u16.txt 2>&1
SAVFI_HIP_LIB=$V/libsavfi_trace.so python tools/ws_trace.py 4 f8 3 > $A/trace_u16_b4.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4 > $A/bench.json 2>$A/bench.err
cat $A/tests.txt; grep op $A/time_b8.txt $A/time_b4.txt | cut -c1-230; tail -3 $A/trace_u16.txt | cut -c1-300; tail -3 $A/trace_u16_b4.txt | cut -c1-300; python -c "
import json; d=json.loads(open('$A/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['kernels']['sepconv_bwd'], d['kernels']['sepconv_fwd'], d['roofline']['frac'], d.get('parity_check'))"
