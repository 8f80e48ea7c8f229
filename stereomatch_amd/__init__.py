"""stereomatch_amd -- MI355X-native cost-volume front end of StereoMatch.

Only what the hot path needs: `csrc/` (HIP kernels + the C-ABI of include/smhip.h, built
into lib/libsmhip.so), `capi` (ctypes binding), `pnm` (PGM/PPM I/O with the reference's
band layout) and `synth` (deterministic synthetic stereo pairs for the benchmarks).
"""
from . import capi, pnm, sharding, synth  # noqa: F401
