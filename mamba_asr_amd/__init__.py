# Import alias: the product package lives in the directory "mamba-asr_amd/", whose name is not a
# valid Python identifier.  This stub makes `import mamba_asr_amd` resolve to it.
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mamba-asr_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
