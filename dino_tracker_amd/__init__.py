"""MI355X-native (gfx950) implementation of DINO-Tracker's per-video inference hot path.

Host side mirrors the reference's Python API (Tracker / ModelInference / TrackerHead / DeltaDINO /
RangeNormalizer / VitExtractor); all arithmetic on the hot path runs in hand-written HIP kernels reached through
the C-ABI library `dino_tracker_amd/csrc/libdtk.so` (include/dtk.h).  There is no CPU fallback: every hot-path
call raises if the library or a GPU is missing.
"""
__version__ = "0.1.0"
