"""Host-side mirror of the reference plugin surface (reference model/LFT.py:8, :269, :280):
``get_model(args)`` / ``get_loss(args)`` / ``weights_init(m)``, with the reference's 78 state-dict
keys, whose ``forward`` runs entirely in liblft_hip.so on the input's HIP device.

PyTorch is used here only for device memory, streams and the nn.Module/state_dict plumbing the
reference's train.py / test.py expect (load_state_dict, .to(device), .eval(), net(x)).
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .params import param_table

_PREC = {"fp32": _lib.PREC_F32, "f32": _lib.PREC_F32, "float32": _lib.PREC_F32,
         "bf16": _lib.PREC_BF16, "bfloat16": _lib.PREC_BF16,
         "fp16": _lib.PREC_F16, "f16": _lib.PREC_F16, "float16": _lib.PREC_F16}


class _Node(nn.Module):
    """Parameter container; gives state-dict paths like ``altblock.0.spa_trans.MLP.weight``."""


def _attach(root: nn.Module, dotted: str, p: nn.Parameter) -> None:
    parts = dotted.split(".")
    node = root
    for name in parts[:-1]:
        if name not in node._modules:
            node.add_module(name, _Node())
        node = node._modules[name]
    node.register_parameter(parts[-1], p)


class get_model(nn.Module):
    """LFT network (reference model/LFT.py:8-83).  ``args`` needs ``channels`` (64), ``angRes``,
    ``scale_factor`` exactly as the reference reads them (LFT.py:11-14).

    ``precision``: 'fp32' (exact-fp32 MFMA; 3e-7 of the reference, default), 'fp16' (IEEE-half MFMA operands and
    inter-kernel tensors, fp32 accumulation; ~2e-4 of the reference at the speed of 'bf16') or 'bf16' (bf16 operands
    and tensors; ~1.7e-3).  May also be given as ``args.lft_precision``.
    """

    def __init__(self, args, precision: Optional[str] = None, streams: Optional[int] = None):
        super().__init__()
        # Patches never interact, so a batch can be split over several HIP streams: the kernels of one sub-batch
        # (e.g. the VALU-bound windowed attention) then overlap the MFMA-bound kernels of another.
        self.streams = int(streams if streams is not None else getattr(args, "lft_streams", 2))
        self.channels = int(args.channels)
        self.angRes = int(args.angRes)
        self.factor = int(args.scale_factor)
        if self.channels != 64:
            raise ValueError("this build supports channels=64 only (reference default)")
        self.precision = precision or getattr(args, "lft_precision", "fp32")
        if self.precision not in _PREC:
            raise ValueError(f"unknown precision {self.precision!r}")
        self._names = []
        for name, shape, kind in param_table(self.channels, self.factor):
            t = torch.empty(shape, dtype=torch.float32)
            if kind.startswith("w"):                       # PyTorch default Conv/Linear init and the explicit
                b = 1.0 / math.sqrt(int(kind[1:]))         # kaiming_uniform_(a=sqrt(5)) of LFT.py:132,204
                t.uniform_(-b, b)
            elif kind == "ln_w":
                t.fill_(1.0)
            else:
                t.zero_()
            _attach(self, name, nn.Parameter(t))
            self._names.append(name)
        # GEMM arithmetic of the training kernels: 'fp32' (exact fp32 MFMA), 'bf16x6' (three exact bf16 parts per operand, six MFMAs
        # per product: fp32-class, ~20 % faster) or 'bf16x3' (split-bf16, ~1e-5 relative, ~45 % faster)
        self.train_math = getattr(args, "lft_train_math", "fp32")
        # After every eager forward, read the workspaces' status word and raise on non-finite activations (an fp16 range
        # overflow).  Costs a device synchronisation per call, so it is on by default only where the risk is (fp16);
        # captured graphs / pipelines never check by themselves: call check_status() (GraphedForward.check / PipelinedForward.sync).
        self.check_finite = bool(getattr(args, "lft_check_finite", self.precision in ("fp16", "f16", "float16")))
        self._packed = None        # (key, tensor)
        self._work = {}            # slot -> (key, tensor)
        self._side_streams = {}    # device -> [torch.cuda.Stream]

    # ------------------------------------------------------------------ packing / buffers
    def _params_in_order(self):
        d = dict(self.named_parameters())
        return [d[n] for n in self._names]

    def _pack_key(self, ps, h, w, prec):
        return (h, w, prec, tuple((p.data_ptr(), p._version) for p in ps))

    def _ensure_packed(self, dev, h, w, prec, stream):
        ps = self._params_in_order()
        for p in ps:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.LftError("all parameters must be contiguous float32 tensors on the input's device "
                                    f"({dev}); call net.to(device) first")
        key = self._pack_key(ps, h, w, prec)
        if self._packed is None or self._packed[0] != key:
            nbytes = _lib.packed_bytes(self.angRes, h, w, self.factor, prec)
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            arr = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
            _lib.check(_lib.lib().lft_pack_weights(arr, len(ps), buf.data_ptr(), self.angRes, h, w, self.factor,
                                                   prec, stream), "lft_pack_weights")
            self._packed = (key, buf)
        return self._packed[1]

    def _ensure_work(self, dev, B, h, w, prec, slot=0):
        key = (str(dev), B, h, w, prec)
        cur = self._work.get(slot)
        if cur is None or cur[0] != key:
            if torch.cuda.is_current_stream_capturing():
                # A workspace created inside a capture would put its status reset into the graph: every replay would then
                # clear the sticky word, and an overflow in replay n would be erased by replay n+1 (one read must cover every
                # replay since the last reset, include/lft_hip.h).  Allocation during capture is not wanted either.
                raise _lib.LftError("a workspace would have to be created while a HIP graph is being captured: run one eager "
                                    "forward of this shape (same _slot_base) first (GraphedForward needs warmup >= 1)")
            nbytes = _lib.workspace_bytes(B, self.angRes, h, w, self.factor, prec)
            cur = (key, torch.empty(nbytes, dtype=torch.uint8, device=dev))
            self._work[slot] = cur
            # the workspace's sticky status word starts clear (ordered on the current stream, like every later use)
            _lib.check(_lib.lib().lft_status_reset(cur[1].data_ptr(), B, self.angRes, h, w, self.factor, prec,
                                                   torch.cuda.current_stream(dev).cuda_stream), "lft_status_reset")
        return cur[1]

    def check_status(self, reset: bool = True) -> None:
        """Raise LftError if any forward since the last check saw a non-finite activation or output -- on the fp16 path: an
        activation left the half-precision range (|x| > 65504), which must be a loud error, never a silently wrong image.
        Reads the sticky status word of every workspace this module owns (eager forwards, captured graphs, pipelines);
        synchronises the device."""
        bad = []
        for slot, (key, buf) in self._work.items():
            _, B, h, w, prec = key
            with torch.cuda.device(buf.device):
                stream = torch.cuda.current_stream(buf.device).cuda_stream
                torch.cuda.synchronize(buf.device)              # forwards on side / pipeline streams included
                flags = ctypes.c_uint(0)
                rc = _lib.lib().lft_status_read(buf.data_ptr(), B, self.angRes, h, w, self.factor, prec, stream, ctypes.byref(flags))
                if rc == _lib.STATUS_NONFINITE:
                    bad.append((slot, _lib.lib().lft_last_error().decode()))
                    if reset:
                        _lib.check(_lib.lib().lft_status_reset(buf.data_ptr(), B, self.angRes, h, w, self.factor, prec, stream), "lft_status_reset")
                else:
                    _lib.check(rc, "lft_status_read")
        if bad:
            raise _lib.LftError(f"{self.precision} forward: {bad[0][1]} (workspace slots {[b[0] for b in bad]})")

    # ------------------------------------------------------------------ forward
    def forward(self, lr: torch.Tensor, _slot_base: int = 0) -> torch.Tensor:
        """lr: float32 [B,1,A*h,A*w] on a HIP device -> float32 [B,1,A*h*s,A*w*s] (reference LFT.py:52-83).
        ``_slot_base`` selects a private set of workspaces (PipelinedForward keeps several forwards in flight)."""
        if lr.dim() != 4 or lr.size(1) != 1:
            raise ValueError(f"expected [B,1,A*h,A*w], got {tuple(lr.shape)}")
        if not lr.is_cuda:
            raise _lib.LftError("lft_amd runs on a HIP device only (no CPU fallback); move the input and the model to 'cuda'")
        A, s = self.angRes, self.factor
        B, _, H, W = lr.shape
        if H % A or W % A:
            raise ValueError(f"mosaic {H}x{W} is not divisible by angRes {A}")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training (reference train.py:89-107; nn.Module starts in training mode, as the reference's net does): the
            # fp32 forward-with-tape / backward kernels, whatever self.precision says; gradients reach the 78 parameters,
            # none flows to the input (the reference's data has none either).  After net.eval() (reference test.py:53)
            # forward is the inference path and builds no autograd graph, with or without torch.no_grad().
            from .train import LFTFunction
            return LFTFunction.apply(lr.contiguous().float(), A, s, self.train_math, *self._params_in_order())
        h, w = H // A, W // A
        x = lr.contiguous().float()
        prec = _PREC[self.precision]
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            main = torch.cuda.current_stream(x.device)
            packed = self._ensure_packed(x.device, h, w, prec, stream)
            out = torch.empty((B, 1, H * s, W * s), dtype=torch.float32, device=x.device)
            nsplit = max(1, min(self.streams, B))
            if nsplit == 1:
                work = self._ensure_work(x.device, B, h, w, prec, slot=(_slot_base, 0))
                _lib.check(_lib.lib().lft_forward(packed.data_ptr(), x.data_ptr(), out.data_ptr(), work.data_ptr(),
                                                  B, A, h, w, s, prec, stream), "lft_forward")
            else:
                side = self._side_streams.setdefault(x.device, [])
                while len(side) < nsplit:
                    side.append(torch.cuda.Stream(device=x.device))
                from .dp import shard_range
                shards = [shard_range(B, i, nsplit) for i in range(nsplit)]
                works = [self._ensure_work(x.device, b1 - b0, h, w, prec, slot=(_slot_base, i)) for i, (b0, b1) in enumerate(shards)]
                ready = torch.cuda.Event()
                ready.record(main)                      # inputs, packed weights, `out` and fresh workspaces are ordered on the caller's stream
                for i, (b0, b1) in enumerate(shards):
                    st = side[i]
                    st.wait_event(ready)
                    _lib.check(_lib.lib().lft_forward(packed.data_ptr(), x[b0:b1].data_ptr(), out[b0:b1].data_ptr(), works[i].data_ptr(),
                                                      b1 - b0, A, h, w, s, prec, st.cuda_stream), "lft_forward")
                    main.wait_stream(st)                # the caller's stream sees the finished sub-batch
            if self.check_finite and not torch.cuda.is_current_stream_capturing():
                self.check_status()                     # fp16 by default: an overflow raises here (synchronises, like the .cpu() that follows in test.py)
        return out


class GraphedForward:
    """Replay of one captured forward (HIP graph) for a fixed input shape: the ~22 launches per sub-batch and the
    stream fork/join become a single graph launch.  Inference only; weights must not change between replays
    (re-create after load_state_dict / optimizer steps).  Usage: g = GraphedForward(net, example_lr); out = g(lr)."""

    def __init__(self, net: "get_model", example: torch.Tensor, warmup: int = 3, slot_base: int = 0):
        if warmup < 1:
            raise ValueError("GraphedForward needs at least one eager warm-up forward: it creates the workspaces and clears their "
                             "status words OUTSIDE the captured graph")
        self.net = net
        self.static_in = example.detach().clone().contiguous()
        with torch.no_grad():
            s = torch.cuda.Stream(device=example.device)
            s.wait_stream(torch.cuda.current_stream(example.device))
            with torch.cuda.stream(s):
                for _ in range(warmup):                # packs weights, sizes buffers, sets kernel attributes
                    net(self.static_in, _slot_base=slot_base)
            torch.cuda.current_stream(example.device).wait_stream(s)
            torch.cuda.synchronize(example.device)
            self.graph = torch.cuda.CUDAGraph()
            # thread-local capture errors: another thread's HIP calls (torch's NCCL watchdog polling its events under bench.py --gpus N)
            # must not invalidate this capture (see lft_amd/train.py: CAPTURE_MODE)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.static_out = net(self.static_in, _slot_base=slot_base)
            # The graph has the packed-weight buffer's address baked in: keep the buffer alive, and refuse to replay once
            # the module has dropped or replaced it, or once any parameter has changed since it was packed (optimizer step,
            # load_state_dict, in-place edits: they bump the tensors' version counters without touching net._packed) -- a
            # replay would read freed memory or silently use the old weights.
            self._packed = net._packed
            self._params = net._params_in_order()     # the Parameter objects whose (data_ptr, version) the pack key records

    def check_fresh(self) -> None:
        """Raise unless the captured packed-weight buffer still is the module's and still matches its parameters
        (78 data_ptr / version reads on the host, ~15 us)."""
        net = self.net
        if net._packed is not self._packed:
            raise RuntimeError("the model's weights were re-packed or dropped since this graph was captured "
                               "(optimizer step / load_state_dict): create a new GraphedForward")
        key = self._packed[0]
        if net._pack_key(self._params, key[0], key[1], key[2]) != key:
            raise RuntimeError("the model's parameters changed since this graph was captured (load_state_dict / optimizer "
                               "step / in-place update): create a new GraphedForward")

    def check(self) -> None:
        """Overflow / non-finite check of every replay so far (module.check_status: synchronises, raises LftError)."""
        self.net.check_status()

    def __call__(self, lr: torch.Tensor) -> torch.Tensor:
        if lr.shape != self.static_in.shape:
            raise ValueError(f"graph was captured for {tuple(self.static_in.shape)}, got {tuple(lr.shape)}")
        self.check_fresh()
        if lr.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(lr)
        self.graph.replay()
        return self.static_out


class PipelinedForward:
    """``depth`` captured forwards (own input / output / workspace buffers, shared packed weights) replayed round-robin on
    ``depth`` HIP streams, so that consecutive steps overlap: the tail of step i (few workgroups left, CUs draining) runs
    under the head of step i+1.  Throughput device for serving loops -- each call still processes its whole batch through
    the whole network; only completion is asynchronous: the returned tensor is valid after ``sync()`` (or once the
    caller's stream has waited on ``event``), and is overwritten ``depth`` calls later."""

    def __init__(self, net: "get_model", example: torch.Tensor, depth: int = 2):
        self.depth = int(depth)
        self.graphs = [GraphedForward(net, example, slot_base=1 + k) for k in range(self.depth)]
        self.streams = [torch.cuda.Stream(device=example.device) for _ in range(self.depth)]
        self.events = [torch.cuda.Event() for _ in range(self.depth)]
        self.i = 0
        self.event = None

    def __call__(self, lr: Optional[torch.Tensor] = None) -> torch.Tensor:
        k = self.i % self.depth
        self.i += 1
        g, st = self.graphs[k], self.streams[k]
        g.check_fresh()                                                      # weights re-packed, dropped or changed in place: refuse
        st.wait_stream(torch.cuda.current_stream(g.static_in.device))       # the caller's input is ready
        with torch.cuda.stream(st):
            if lr is not None and lr.data_ptr() != g.static_in.data_ptr():
                g.static_in.copy_(lr)
            g.graph.replay()
            self.events[k].record(st)
        self.event = self.events[k]
        return g.static_out

    def sync(self, check: bool = False) -> None:
        """Order the caller's stream after every step in flight.  check=True additionally reads the status words (device
        synchronisation) and raises LftError if any replay saw a non-finite activation (fp16 overflow)."""
        cur = torch.cuda.current_stream(self.graphs[0].static_in.device)
        for st in self.streams:
            cur.wait_stream(st)
        if check:
            self.graphs[0].net.check_status()


class get_loss(nn.Module):
    """Mean absolute error between SR and HR mosaics (reference model/LFT.py:269-277)."""

    def __init__(self, args=None):
        super().__init__()

    def forward(self, SR, HR):
        return (SR - HR).abs().mean()


def weights_init(m):
    """No-op, as in the reference (model/LFT.py:280-282); drivers call ``net.apply(weights_init)``."""
    return None
