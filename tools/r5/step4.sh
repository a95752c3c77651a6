R=$GRAFT_REPO_ROOT; cd $R
V=$R/tools/scratch/variants
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), {k:(round(v["avg_us"],1), round(v["min_us"],1)) for k,v in d["kernels"].items()}, round(d["roofline"]["frac"],4))'
for rep in 1 2; do
$B 2>/dev/null | python -c "$P" default
for n in $VARIANTS; do SAVFI_HIP_LIB=$V/libsavfi_$n.so $B 2>/dev/null | python -c "$P" $n; done
done
$B 2>/dev/null | python -c "$P" default
