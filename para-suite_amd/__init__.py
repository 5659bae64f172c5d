"""parasuite-hip: MI355X-native replacement for the aligner child processes of PARA-suite's `map`.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI -> libparasuite_hip.so),
capi (ctypes binding), mapping (host-side mirror of the reference's Mapping classes) and
simulate (synthetic PAR-CLIP data).  Importing capi fails loudly if the library is not built.
"""
from . import capi, simulate  # noqa: F401
from . import mapping  # noqa: F401
