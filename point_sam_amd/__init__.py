"""Import shim: the package sources live in ``point-sam_amd/`` (a directory name Python cannot
import directly because of the hyphen); this shim makes them importable as ``point_sam_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "point-sam_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
