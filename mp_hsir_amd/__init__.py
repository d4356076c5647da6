"""Import alias: ``import mp_hsir_amd`` resolves to the package directory ``mp-hsir_amd/``
(whose name, fixed by the project layout, is not a valid Python identifier)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mp-hsir_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
