"""psgradientsdf_amd — MI355X-native Gradient-SDF photometric-stereo engine (hot path only).

csrc/   hand-written HIP kernels (gfx950) + the C ABI of include/psgsdf.h -> libpsgsdf.so
host/   C++ mirror of the reference's PsOptimizer / LedOptimizer interface over that ABI
capi.py ctypes binding used by tests and bench.py
synth.py deterministic synthetic RGB-D scenes (test / bench inputs)
"""
__all__ = ["capi", "synth"]
