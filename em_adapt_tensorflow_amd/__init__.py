"""Import alias for the package directory ``em-adapt-tensorflow_amd/`` (a hyphen is not a valid
Python identifier, so ``import em_adapt_tensorflow_amd`` resolves here and re-exports that tree)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "em-adapt-tensorflow_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
