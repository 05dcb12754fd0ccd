"""Import shim: the product package lives in `open-simulator_amd/` (hyphenated, per the repo
layout contract); this makes it importable as `open_simulator_amd`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "open-simulator_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
