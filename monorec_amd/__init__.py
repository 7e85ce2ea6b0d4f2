"""monorec_amd - MI355X (gfx950) native implementation of the MonoRec cost-volume inference path.

Drop-in for `model.monorec.monorec_model.MonoRecModel` of Brummi/MonoRec;
the hot path runs in hand-written HIP kernels behind the C ABI of include/monorec_hip.h.
"""
from .model import MonoRecModel  # noqa: F401

__all__ = ["MonoRecModel"]

# ROCm maps hipStreams round-robin onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one serialise.  A keyframe in
# flight uses three streams, two keyframes six: with 4 queues they alias (measured on MI355X, round 3: 533 -> 567 keyframes/s with 16).
# The variable is read when the HIP runtime initialises (torch initialises it lazily, at the first device call), so importing this
# package first is enough; an exported value wins.
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
