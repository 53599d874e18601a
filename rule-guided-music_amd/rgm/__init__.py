"""rgm -- ctypes binding of librgm_hip.so (include/rgm.h) for torch-ROCm tensors.

PyTorch is plumbing here (device memory, streams, torch.distributed); every kernel on the
hot path is hand-written HIP for gfx950 inside librgm_hip.so.  There is NO CPU fallback: if the
library is missing or a tensor is not on a HIP device, calls raise.
"""
from .native import lib, RgmError, check, ptr, current_stream, require_cuda, DitCfg  # noqa: F401
