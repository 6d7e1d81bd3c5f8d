"""openset-rcnn_amd: MI355X (gfx950) implementation of Openset R-CNN's per-image detection hot path.

csrc/   hand-written HIP kernels + the C ABI (include/osr.h)  -> libosr_hip.so
host/   ctypes binding, torch-tensor op wrappers, and the mirror of the reference's registry surface
        (ClsFreeRPNHead / ClsFreeRPN / OpensetROIHeads / GeneralizedRCNN, yaml config keys).

The directory name is not a Python identifier; import it through ``__graft_entry__.load_package()``
(registers it as ``openset_rcnn_amd``).
"""
from .host import _lib, ops  # noqa: F401
from .host._lib import OsrError  # noqa: F401
