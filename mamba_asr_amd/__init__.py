"""mamba_asr_amd — MI355X-native ConMamba ASR encoder hot path.

Python host code (this package) calls through a C ABI (include/conmamba_hip.h,
mamba_asr_amd/lib/libconmamba_hip.so) into hand-written HIP kernels for gfx950.
Importing the package does not load the library; the first op does, and raises if it is missing.

"""
__version__ = "0.1.0"

from . import _native  # noqa: F401
from . import ops  # noqa: F401
