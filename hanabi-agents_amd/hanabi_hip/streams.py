"""HIP streams restricted to a subset of the MI355X's 256 CUs (hipExtStreamCreateWithCUMask), wrapped for torch.

Used to partition the chip between the actor (big GEMMs) and the learner (many small kernels) so the learner's
launches never queue behind a GEMM that occupies every CU. Mask bit i enables logical CU i; consecutive logical
CUs are dealt round-robin over the 8 XCDs, so a contiguous bit range is balanced across XCDs.
"""
import ctypes as C

import torch

_HIP = None


def masked_stream(first_cu, last_cu, device=None, total_cus=256):
    """torch.cuda.ExternalStream running only on logical CUs [first_cu, last_cu)."""
    global _HIP
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")
    device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    words = (C.c_uint32 * (total_cus // 32))()
    for i in range(first_cu, last_cu):
        words[i // 32] |= 1 << (i % 32)
    handle = C.c_void_p()
    with torch.cuda.device(device):
        rc = _HIP.hipExtStreamCreateWithCUMask(C.byref(handle), total_cus // 32, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(handle.value, device=device)
