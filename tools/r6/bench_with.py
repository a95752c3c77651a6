"""A/B runs of bench.py with routing attributes of hip_ops changed:  python tools/r6/bench_with.py NAME=VALUE [...] -- <bench.py arguments>
(the attributes are module attributes by design -- no environment switch in the product; this tool sets them before bench.py's main runs)."""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
cut = sys.argv.index('--') if '--' in sys.argv else len(sys.argv)
sets, rest = sys.argv[1:cut], sys.argv[cut + 1:]
sys.argv = [os.path.join(root, 'bench.py')] + rest
import bench  # noqa: E402,F401  (installs the package alias)
from meta_interpolation_amd import hip_ops, model_utils  # noqa: E402
for item in sets:
    name, value = item.split('=')
    mod = model_utils if hasattr(model_utils, name) and not hasattr(hip_ops, name) else hip_ops
    assert hasattr(mod, name), name
    setattr(mod, name, type(getattr(mod, name))(eval(value)))
    print("[bench_with] %s.%s = %r" % (mod.__name__, name, getattr(mod, name)), file=sys.stderr)
runpy.run_path(sys.argv[0], run_name='__main__')
