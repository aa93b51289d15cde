"""nrtsearch_amd -- MI355X-native (gfx950) query-execution path for nrtsearch.

Only what the hot path needs: csrc/ (HIP kernels + C-ABI runtime, built into libnrtgpu.so),
_lib (ctypes binding of include/nrtgpu.h), api (host-side mirror of the reference's search
interface), synth (deterministic Zipf corpora / queries).  No CPU implementation of the path.
"""
__all__ = ["api", "synth", "build"]
