R=$GRAFT_REPO_ROOT; A=$R/gpurun_out/r5s3; mkdir -p $A; cd $R
V=$R/tools/scratch/variants
for rep in 1 2; do
python tools/frames8_time.py 8 256 448 bwd_frames8_unit16 2>&1 | grep op | cut -c1-200
SAVFI_HIP_LIB=$V/libsavfi_aligned.so python tools/frames8_time.py 8 256 448 bwd_frames8_unit16 2>&1 | grep op | sed "s/^/aligned /" | cut -c1-200
done
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), {k:(round(v["avg_us"],1), round(v["min_us"],1)) for k,v in d["kernels"].items()}, round(d["roofline"]["frac"],4))'
$B 2>/dev/null | python -c "$P" default
SAVFI_HIP_LIB=$V/libsavfi_aligned.so $B 2>/dev/null | python -c "$P" aligned
$B 2>/dev/null | python -c "$P" default
