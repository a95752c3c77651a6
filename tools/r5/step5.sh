R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_hip_ops_gpu.py -x -q -k "upsample or wgrad or conv3x3 or winograd or weight" 2>&1 | tail -4
python tools/upsample_bench.py 2>&1 | grep lib | cut -c1-250
SAVFI_UPSAMPLE_BWD_FORM=1 python tools/upsample_bench.py 2>&1 | grep lib | cut -c1-250
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), {k:(round(v["avg_us"],1), round(v["min_us"],1)) for k,v in d["kernels"].items()}, round(d["roofline"]["frac"],4), d.get("parity_check"))'
$B 2>/dev/null | python -c "$P" default
SAVFI_UPSAMPLE_BWD_FORM=1 $B 2>/dev/null | python -c "$P" tiled_upsample_bwd
$B 2>/dev/null | python -c "$P" default
