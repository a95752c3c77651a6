"""Import shim: the package directory is `meta-interpolation_amd/` (not a valid Python identifier),
so `import meta_interpolation_amd` resolves here and re-points the package path at it."""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "meta-interpolation_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
