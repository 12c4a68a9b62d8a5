"""Helpers for the -m gpu parity tests: call liblft_hip.so's per-stage C-ABI entry points on torch
device tensors and bring results back in the oracle's [B,C,V,h,w] layout."""
import ctypes

import numpy as np
import torch

from lft_amd import _lib
from lft_amd.params import deterministic_state, param_table

DEV = "cuda:0"
PRECS = {"fp32": _lib.PREC_F32, "bf16": _lib.PREC_BF16, "fp16": _lib.PREC_F16}
ACT_DTYPE = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def stream():
    return torch.cuda.current_stream().cuda_stream


class Packed:
    """Packed weights + workspace for one (state dict, A, h, w, s, prec, B)."""

    def __init__(self, sd_np, A, h, w, s, prec, B):
        self.A, self.h, self.w, self.s, self.B = A, h, w, s, B
        self.prec_name, self.prec = prec, PRECS[prec]
        names = [n for n, _, _ in param_table(64, s)]
        self.params = [torch.from_numpy(sd_np[n]).to(DEV).contiguous() for n in names]
        self.buf = torch.empty(_lib.packed_bytes(A, h, w, s, self.prec), dtype=torch.uint8, device=DEV)
        arr = (ctypes.c_void_p * len(self.params))(*[p.data_ptr() for p in self.params])
        _lib.check(_lib.lib().lft_pack_weights(arr, len(self.params), self.buf.data_ptr(), A, h, w, s, self.prec, stream()),
                   "lft_pack_weights")
        self.work = torch.empty(_lib.workspace_bytes(B, A, h, w, s, self.prec), dtype=torch.uint8, device=DEV)
        torch.cuda.synchronize()

    def dims(self):
        return (self.B, self.A, self.h, self.w, self.s, self.prec)

    def new_act(self):
        return torch.empty((self.B, self.A * self.A, self.h, self.w, 64), dtype=ACT_DTYPE[self.prec_name], device=DEV)


def to_act(x_bcvhw: torch.Tensor, prec: str) -> torch.Tensor:
    """oracle layout [B,C,V,h,w] (cpu fp32) -> device channels-last [B,V,h,w,C] in the activation dtype."""
    return x_bcvhw.permute(0, 2, 3, 4, 1).contiguous().to(DEV).to(ACT_DTYPE[prec])


def from_act(a: torch.Tensor) -> torch.Tensor:
    return a.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def err_report(got: torch.Tensor, ref: torch.Tensor) -> str:
    e = (got - ref).abs()
    idx = np.unravel_index(int(e.argmax()), tuple(e.shape))
    return (f"max|err|={float(e.max()):.3e} at {idx} (got {float(got[idx]):.6f} ref {float(ref[idx]):.6f}) "
            f"rms err={float(e.pow(2).mean().sqrt()):.3e} ref rms={float(ref.pow(2).mean().sqrt()):.3e} "
            f"nan={int(torch.isnan(got).sum())}")


def rel_rms(got, ref):
    return float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


def rel_max(got, ref):
    return float((got - ref).abs().max() / ref.abs().max())
