"""flairhip: Python binding of libflairhip (hand-written HIP kernels for gfx950) and the layers built on it."""
import os as _os

# RCCL / device-tensor sharing between the ranks of one node needs dmabuf IPC on this driver; ROCr reads the flag once,
# when the HSA runtime initialises (first torch.cuda call), so it is set when the package is imported, not when the
# process group is created (flairhip.distributed.ensure_process_group keeps its own setdefault for bare scripts)
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
