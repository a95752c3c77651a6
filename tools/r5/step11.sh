R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_hip_ops_gpu.py -x -q -k "hands_out or wgrad or conv" 2>&1 | tail -4
python -m pytest tests/test_system_gpu.py -x -q -k "sepconv" 2>&1 | tail -3
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],2), round(d["roofline"]["frac"],4), d.get("parity_check"))'
$B 2>/dev/null | python -c "$P" default
SAVFI_WGRAD_NO_BIAS=1 $B 2>/dev/null | python -c "$P" no_fused_bias
$B 2>/dev/null | python -c "$P" default
SAVFI_WGRAD_NO_BIAS=1 $B 2>/dev/null | python -c "$P" no_fused_bias
