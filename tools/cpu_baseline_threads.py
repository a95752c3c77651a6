"""The CPU baseline of bench.py (the oracle on task 0 of the C2 workload: 5 inner steps + target pass + outer backward at 256 x 448) at
several thread counts on THIS host -- the evidence behind bench.py's CPU_BASELINE_THREADS ("the same box's host cores,
core count stated").  One child process per count (OMP / torch thread pools are sized at start-up)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import bench
line, _ = bench.cpu_baseline('sepconv', 256, 448, 5, dict(optimizer='SGD', loss='1*L1', inner_lr=1e-5))
print(json.dumps(line))
''' % ROOT
counts = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, 256]
print("host: %d logical CPUs" % os.cpu_count())
best = None
for n in counts:
    if n > (os.cpu_count() or 1):
        continue
    env = dict(os.environ, SAVFI_CPU_BASELINE_THREADS=str(n), OMP_NUM_THREADS=str(n))
    out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=1800)
    try:
        line = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print("threads %4d: failed: %s" % (n, out.stderr[-300:]))
        continue
    print("threads %4d: %.3f inner-loop steps/s   (%s; %s)" % (n, line['value'], line['cpu_model'], line['sample']), flush=True)
    if best is None or line['value'] > best[1]:
        best = (n, line['value'])
print("best: %d threads, %.3f inner-loop steps/s" % best)
