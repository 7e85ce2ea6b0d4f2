"""monorec_amd - MI355X (gfx950) native implementation of the MonoRec cost-volume inference path.

Drop-in for `model.monorec.monorec_model.MonoRecModel` of Brummi/MonoRec;
the hot path runs in hand-written HIP kernels behind the C ABI of include/monorec_hip.h.
"""
from .model import MonoRecModel  # noqa: F401

__all__ = ["MonoRecModel"]
