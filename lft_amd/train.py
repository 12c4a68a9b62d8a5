"""Training side of the plugin surface: what the reference gets from PyTorch autograd when ``train.py:89-107`` runs
``net(data) -> criterion -> loss.backward() -> optimizer.step()`` -- here the forward-with-tape and the backward of
the whole network are liblft_hip.so calls (fp32 kernels, lft_amd/csrc/lft_train*.cuh).

Two ways in:
  * ``get_model.forward`` under autograd returns a tensor whose ``grad_fn`` is :class:`LFTFunction`; ``loss.backward()``
    fills ``p.grad`` of the 78 parameters, so the reference's ``torch.optim.Adam`` loop works unchanged.
  * :class:`TrainStep` is the MI355X-first loop: parameters, gradients and Adam moments live in three flat fp32
    buffers (the module's parameters become views), the loss gradient, backward, ONE all-reduce over RCCL and the
    fused Adam update are five enqueues per step.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _lib
from .params import param_table


def tape_bytes(B, A, h, w, s) -> int:
    n = ctypes.c_size_t(0)
    _lib.check(_lib.lib().lft_train_tape_bytes(B, A, h, w, s, ctypes.byref(n)), "lft_train_tape_bytes")
    return n.value


def grad_floats(s) -> int:
    n = ctypes.c_size_t(0)
    _lib.check(_lib.lib().lft_train_grad_floats(s, ctypes.byref(n)), "lft_train_grad_floats")
    return n.value


def tape_view(tape: torch.Tensor, name: str, B, A, h, w, s, shape) -> torch.Tensor:
    """A saved activation of the last lft_train_forward as a tensor view (tests / debugging)."""
    off = ctypes.c_size_t(0)
    _lib.check(_lib.lib().lft_train_tape_offset(name.encode(), B, A, h, w, s, ctypes.byref(off)), "lft_train_tape_offset")
    n = 1
    for d in shape:
        n *= d
    return tape.view(torch.float32)[off.value:off.value + n].view(*shape)


def _check_params(ps: List[torch.Tensor], dev) -> None:
    if len(ps) != _lib.NUM_PARAMS:
        raise _lib.LftError(f"expected {_lib.NUM_PARAMS} parameters, got {len(ps)}")
    for p in ps:
        if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
            raise _lib.LftError(f"all parameters must be contiguous float32 tensors on {dev}; call net.to(device) first")


def _ptr_array(ps):
    return (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])


MATH = {"fp32": _lib.MATH_F32, "bf16x3": _lib.MATH_BF16X3, "bf16x6": _lib.MATH_BF16X6}
# HIP-graph captures use the thread-local error mode: with a process group alive, torch's NCCL watchdog THREAD polls its work
# events (hipEventQuery) at any time; under the default global mode such a call from another thread while this thread captures
# invalidates the capture ("operation not permitted when stream is capturing" -- seen once in round 4 on the RCCL test, a race
# that had been there since the graphs were introduced).  Only calls of the capturing thread itself matter to these captures.
CAPTURE_MODE = "thread_local"


def train_forward(ps, lr, A, s, tape=None, math="fp32"):
    """lft_train_forward: returns (out, tape).  math: 'fp32' (exact fp32 MFMA), 'bf16x6' (fp32-class six-product split) or 'bf16x3' (split-bf16 products)."""
    B, _, H, W = lr.shape
    h, w = H // A, W // A
    dev = lr.device
    _check_params(ps, dev)
    if tape is None:
        tape = torch.empty(tape_bytes(B, A, h, w, s), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 1, H * s, W * s), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().lft_train_forward(_ptr_array(ps), len(ps), lr.data_ptr(), out.data_ptr(), tape.data_ptr(),
                                            B, A, h, w, s, MATH[math], stream), "lft_train_forward")
    return out, tape


def train_backward(ps, lr, tape, dout, A, s, grads=None, math="fp32"):
    """lft_train_backward: returns the flat gradient buffer (78 gradients back to back, state_dict order)."""
    B, _, H, W = lr.shape
    h, w = H // A, W // A
    dev = lr.device
    if grads is None:
        grads = torch.empty(grad_floats(s), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().lft_train_backward(_ptr_array(ps), len(ps), lr.data_ptr(), tape.data_ptr(), dout.data_ptr(), grads.data_ptr(),
                                             B, A, h, w, s, MATH[math], stream),
               "lft_train_backward")
    return grads


def block_backward(ps, lr, tape, block, layer, d_out, A, s, grads, math="fp32"):
    """lft_train_block_backward: the backward pass of ONE block (include/lft_hip.h: LFT_BLOCK_*) against the tape of a full forward;
    returns the block's outgoing gradient [B, A*A, h, w, 64] (None for the feature extractor) and writes the block's parameter
    gradients into the flat buffer `grads`."""
    B, _, H, W = lr.shape
    h, w = H // A, W // A
    dev = lr.device
    d_in = None if block == _lib.BLOCK_INIT else torch.empty((B, A * A, h, w, 64), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().lft_train_block_backward(_ptr_array(ps), len(ps), lr.data_ptr(), tape.data_ptr(), block, layer, d_out.data_ptr(),
                                                   None if d_in is None else d_in.data_ptr(), grads.data_ptr(), B, A, h, w, s, MATH[math], stream),
               "lft_train_block_backward")
    return d_in


def grad_bucket(s: int, bucket: int):
    """(first_float, n_floats) of gradient bucket `bucket` in the flat buffer (include/lft_hip.h: lft_train_grad_bucket)."""
    first, count = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.check(_lib.lib().lft_train_grad_bucket(s, bucket, ctypes.byref(first), ctypes.byref(count)), "lft_train_grad_bucket")
    return first.value, count.value


def train_backward_buckets(ps, lr, tape, dout, A, s, grads, on_bucket, math="fp32"):
    """lft_train_backward_buckets: the backward pass, calling on_bucket(bucket, first_float, n_floats) on the host each
    time a contiguous range of the flat gradient buffer is final (its last kernel enqueued on the current stream).
    An exception raised by on_bucket stops the pass at that boundary (the C call enqueues nothing further and returns
    LFT_ERR_CALLBACK) and is re-raised here."""
    B, _, H, W = lr.shape
    h, w = H // A, W // A
    dev = lr.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    err = []

    def trampoline(_user, bucket, first, count):
        try:
            on_bucket(int(bucket), int(first), int(count))
            return 0
        except BaseException as e:          # noqa: BLE001 -- must not propagate through the C frame
            err.append(e)
            return 1                        # tells lft_train_backward_buckets to stop: nothing further is enqueued

    cb = _lib.BUCKET_FN(trampoline)
    rc = _lib.lib().lft_train_backward_buckets(_ptr_array(ps), len(ps), lr.data_ptr(), tape.data_ptr(), dout.data_ptr(), grads.data_ptr(),
                                               B, A, h, w, s, MATH[math], stream,
                                               ctypes.cast(cb, ctypes.c_void_p), None)
    if err:
        raise err[0]
    _lib.check(rc, "lft_train_backward_buckets")
    return grads


class LFTFunction(torch.autograd.Function):
    """autograd node of the whole network: forward saves the tape, backward returns the 78 parameter gradients."""

    @staticmethod
    def forward(ctx, lr, A, s, math, *params):
        ps = [p.detach() for p in params]
        with torch.cuda.device(lr.device):
            out, tape = train_forward(ps, lr, A, s, math=math)
        ctx.A, ctx.s, ctx.tape, ctx.lr, ctx.math = A, s, tape, lr, math
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, dout):
        ps = [p.detach() for p in ctx.saved_tensors]
        with torch.cuda.device(dout.device):
            flat = train_backward(ps, ctx.lr, ctx.tape, dout.contiguous().float(), ctx.A, ctx.s, math=ctx.math)
        ctx.tape = None
        grads, off = [], 0
        for p in ps:
            grads.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return (None, None, None, None, *grads)


class TrainStep:
    """One data-parallel training step of the reference's loop (train.py:89-107) with flat buffers.

    net: lft_amd.module.get_model on a HIP device.  After construction the module's parameters are views into
    ``self.flat_params`` (state_dict / checkpoints keep working).  ``step(lr, hr)`` returns the loss tensor (device
    scalar, local shard).  With torch.distributed initialised, the flat gradient buffer is summed over the ranks in the three
    contiguous buckets in which the backward pass finishes it -- each bucket's all-reduce starts as soon as its last
    kernel is enqueued and runs beside the rest of the backward pass -- and averaged inside the Adam kernel (L1Loss is a
    mean over the local shard, shards are equal: SURVEY 8e).
    """

    def __init__(self, net, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8, process_group=None, math: Optional[str] = None,
                 graph: bool = True, weight_decay: float = 0.0):
        self.net, self.lr, self.betas, self.eps = net, float(lr), betas, float(eps)
        self.weight_decay = float(weight_decay)                        # reference train.py:82 weight_decay=args.decay_rate (option.py default 0)
        self.math = math or getattr(net, "train_math", "fp32")
        # forward + loss + backward are ~450 kernel launches; for a fixed batch shape they are captured once into a HIP
        # graph and replayed (the kernels read the weights through the same flat buffer every step, and re-pack them
        # inside the graph).  The all-reduce and the Adam kernel (whose bias corrections change every step) stay eager.
        self.use_graph = bool(graph)
        self._graphs = {}
        self.group = process_group
        self.exchange = True               # False: skip the gradient exchange although a process group exists (bench.py times the step both ways)
        ps = net._params_in_order()
        dev = ps[0].device
        _check_params(ps, dev)
        self.s, self.A = net.factor, net.angRes
        n = grad_floats(self.s)
        assert n == sum(p.numel() for p in ps)
        self.flat_params = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat_params[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_params[off:off + k].view(p.shape)
                p.grad = self.flat_grads[off:off + k].view(p.shape)
                off += k
        self.params = ps
        # Replicas must start from the same weights (the reference has no DP; torch's DDP broadcasts rank 0's
        # parameters at construction): a network built from scratch draws its weights from this process's own RNG.
        from .dp import broadcast_
        broadcast_(self.flat_params, src=0, group=self.group)
        self.t = 0
        self._tape = None
        self.last_out = None               # [B,1,A*h*s,A*w*s]: what the network produced in the last step (before the update), for per-batch metrics
        self._scratch = torch.empty(1024 + 1, dtype=torch.float32, device=dev)

    def _fwd_loss_bwd(self, lr_in, hr, tape, dout, loss, on_bucket=None):
        L = _lib.lib()
        dev = lr_in.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        out, _ = train_forward(self.params, lr_in, self.A, self.s, tape=tape, math=self.math)
        n = out.numel()
        _lib.check(L.lft_l1_loss(out.data_ptr(), hr.data_ptr(), n, dout.data_ptr(), 1.0 / n, loss.data_ptr(),
                                 self._scratch.data_ptr(), stream), "lft_l1_loss")
        if on_bucket is None:
            train_backward(self.params, lr_in, tape, dout, self.A, self.s, grads=self.flat_grads, math=self.math)
        else:
            train_backward_buckets(self.params, lr_in, tape, dout, self.A, self.s, self.flat_grads, on_bucket, math=self.math)
        return out

    def _graph_for(self, lr_in, hr, bucketed=False):
        """Captured forward + loss + backward for this batch shape.  bucketed (data-parallel): one graph per gradient
        bucket -- the capture is ended and the next one begun at every bucket boundary the backward pass reports, so that
        step() can start a bucket's all-reduce between two replays."""
        key = (tuple(lr_in.shape), str(lr_in.device), bool(bucketed))
        g = self._graphs.get(key)
        if g is None:
            dev = lr_in.device
            B, _, H, W = lr_in.shape
            h, w = H // self.A, W // self.A
            g = {"lr": lr_in.clone(), "hr": hr.clone(), "dout": torch.empty_like(hr),
                 "tape": torch.empty(tape_bytes(B, self.A, h, w, self.s), dtype=torch.uint8, device=dev)}
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                      # eager warm-up: sets kernel attributes, sizes nothing lazily later
                self._fwd_loss_bwd(g["lr"], g["hr"], g["tape"], g["dout"], self._scratch[1024:1025])
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            if not bucketed:
                g["graphs"] = [torch.cuda.CUDAGraph()]
                with torch.cuda.graph(g["graphs"][0], capture_error_mode=CAPTURE_MODE):
                    g["out"] = self._fwd_loss_bwd(g["lr"], g["hr"], g["tape"], g["dout"], self._scratch[1024:1025])
            else:
                graphs = [torch.cuda.CUDAGraph() for _ in range(_lib.GRAD_BUCKETS)]
                state = {"open": 0}

                def boundary(bucket, first, count):            # host callback between two kernels of the backward pass
                    assert bucket == state["open"], (bucket, state["open"])
                    graphs[bucket].capture_end()
                    state["open"] = bucket + 1
                    if bucket + 1 < len(graphs):
                        graphs[bucket + 1].capture_begin(pool=graphs[0].pool(), capture_error_mode=CAPTURE_MODE)

                with torch.cuda.stream(side):
                    graphs[0].capture_begin(capture_error_mode=CAPTURE_MODE)
                    try:
                        g["out"] = self._fwd_loss_bwd(g["lr"], g["hr"], g["tape"], g["dout"], self._scratch[1024:1025], on_bucket=boundary)
                    except BaseException as e:
                        # A failed capture must not leave a stream in capture mode: the backward pass has stopped at the failing
                        # boundary (nothing was enqueued after it), so end whatever capture is still open, drop every graph of this
                        # attempt and wait for the device before handing the error on (an error of the clean-up itself is chained).
                        try:
                            if state["open"] < len(graphs):
                                graphs[state["open"]].capture_end()
                        except BaseException as e2:            # noqa: BLE001
                            e.__context__ = e2
                        finally:
                            graphs.clear()
                            torch.cuda.synchronize(dev)
                        raise
                torch.cuda.current_stream(dev).wait_stream(side)
                g["graphs"] = graphs
            g["buckets"] = [grad_bucket(self.s, b) for b in range(_lib.GRAD_BUCKETS)]
            self._graphs[key] = g
        return g

    def step(self, lr_in: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
        from . import dp
        dev = lr_in.device
        B, _, H, W = lr_in.shape
        h, w = H // self.A, W // self.A
        L = _lib.lib()
        exchange = self.exchange and dp.dp_active(self.group)
        handles = []

        def start_bucket(bucket, first, count):                # the all-reduce of a finished bucket, beside the rest of the backward
            handles.append(dp.sum_gradients_start_(self.flat_grads[first:first + count], self.group))

        with torch.cuda.device(dev):
            loss = self._scratch[1024:1025]
            if self.use_graph:
                g = self._graph_for(lr_in.contiguous().float(), hr.contiguous().float(), bucketed=exchange)
                g["lr"].copy_(lr_in)
                g["hr"].copy_(hr)
                for b, graph in enumerate(g["graphs"]):
                    graph.replay()
                    if exchange:
                        start_bucket(b, *g["buckets"][b])
                self.last_out = g["out"]                       # SR output of this step's forward (overwritten by the next step of this shape)
            else:
                nb = tape_bytes(B, self.A, h, w, self.s)
                if self._tape is None or self._tape.numel() != nb:
                    self._tape = torch.empty(nb, dtype=torch.uint8, device=dev)
                self.last_out = self._fwd_loss_bwd(lr_in.contiguous().float(), hr.contiguous().float(), self._tape, torch.empty_like(hr), loss,
                                                   on_bucket=start_bucket if exchange else None)
            dp.sum_gradients_finish(handles)                   # the Adam kernel is ordered after the collectives
            gscale = dp.grad_scale(self.group) if exchange else 1.0
            stream = torch.cuda.current_stream(dev).cuda_stream
            self.t += 1
            _lib.check(L.lft_adam_step(self.flat_params.data_ptr(), self.flat_grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                       self.flat_params.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.t,
                                       gscale, self.weight_decay, stream), "lft_adam_step")
        self.net._packed = None            # the inference path must re-pack the new weights
        return loss.clone()


def names(channels: int = 64, scale: int = 2):
    return [n for n, _, _ in param_table(channels, scale)]
