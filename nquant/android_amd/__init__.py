"""Import shim: makes `import nquant.android_amd` load the real package from the sibling directory
`<repo>/nquant.android_amd/` (a dotted directory name cannot be imported directly)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), "nquant.android_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f, _real
