"""MI355X-native LFA analysis hot path (strip detection, row-mean profiles, top-2 peaks).

`detection` mirrors the reference's LFA-Image_Analysis/detection.py function for function; `_capi` is the ctypes binding
of include/lfa.h.  Importing either needs liblfa_hip.so (no CPU fallback)."""
__all__ = ["detection", "_capi"]
