"""CPU ORACLE for the LFT forward hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (lft_amd/ + liblft_hip.so) never calls it and has no CPU fallback.

What it is: a functional restatement, on stock PyTorch CPU operators, of the algorithm of the
reference's model/LFT.py forward.  Every function cites the reference lines it follows.  It keeps
the reference's operator sequence (dense additive window mask fed to scaled-dot-product attention,
unfold + linear for the spatial token embedding, per-view bicubic) so that timing it on host cores
is a fair stand-in for "the reference CPU path" (cpu_baseline kind = "port").

Pinning: tests/test_oracle_golden.py checks this file against fixtures under tests/golden/ that
tools/gen_golden.py produced by importing the real reference module in the build container
(the reference has no tests or golden vectors of its own -- SURVEY.md section 8c).

Third-party arithmetic: the reference's arithmetic lives in PyTorch (README pins 1.3.0; the
container and the GPU box run 2.10.0+rocm7.0 on CPU).  Semantics relied on: bicubic A=-0.75 with
align_corners=False and clamped taps, LayerNorm eps 1e-5, MultiheadAttention = packed in_proj
(q,k from the query/key input, v from the value input), scale 1/sqrt(head_dim), softmax over keys,
out_proj without bias.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

HEADS = 8          # reference LFT.py:19
LAYERS = 4         # reference LFT.py:15
WINDOW = 5         # reference LFT.py:123
TEMPERATURE = 10000.0  # reference LFT.py:17


# --------------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------------
def mosaic_to_views(x: torch.Tensor, A: int) -> torch.Tensor:
    """[B,c,A*h,A*w] -> [B,c,A*A,h,w]; view index v = a1*A + a2 (reference LFT.py:58)."""
    B, c, H, W = x.shape
    h, w = H // A, W // A
    return x.reshape(B, c, A, h, A, w).permute(0, 1, 2, 4, 3, 5).reshape(B, c, A * A, h, w)


def views_to_mosaic(x: torch.Tensor, A: int) -> torch.Tensor:
    """[B,c,A*A,h,w] -> [B,c,A*h,A*w] (reference LFT.py:79)."""
    B, c, V, h, w = x.shape
    return x.reshape(B, c, A, A, h, w).permute(0, 1, 2, 4, 3, 5).reshape(B, c, A * h, A * w)


def bicubic_skip(lr: torch.Tensor, A: int, s: int) -> torch.Tensor:
    """Per-view bicubic x s of the LR mosaic, re-assembled as a mosaic (reference LFT.py:255-266)."""
    B, _, H, W = lr.shape
    v = mosaic_to_views(lr, A)                               # [B,1,V,h,w]
    v = v.reshape(B * A * A, 1, H // A, W // A)
    v = F.interpolate(v, scale_factor=s, mode="bicubic", align_corners=False)
    v = v.reshape(B, 1, A * A, (H // A) * s, (W // A) * s)
    return views_to_mosaic(v, A)


def sinusoid_table(length: int, dim: int) -> torch.Tensor:
    """[length, dim] rows concat(sin(p[:,0::2]), cos(p[:,1::2])), p[l,i] = l / T^(2*(i//2)/dim)
    (reference LFT.py:94-104: sin block first, then cos block -- not interleaved)."""
    i = torch.arange(dim, dtype=torch.float32)
    g = TEMPERATURE ** (2 * torch.div(i, 2, rounding_mode="floor") / dim)
    p = torch.arange(length, dtype=torch.float32).view(-1, 1) / g
    return torch.cat([p[:, 0::2].sin(), p[:, 1::2].cos()], dim=1)


def spatial_pe(h: int, w: int, dim: int) -> torch.Tensor:
    """[dim,h,w] = (PE_h[y] + PE_w[x]) / 2 (reference LFT.py:69,107-115 with dim=[3,4])."""
    return ((sinusoid_table(h, dim).view(h, 1, dim) + sinusoid_table(w, dim).view(1, w, dim)) / 2).permute(2, 0, 1)


def angular_pe(V: int, dim: int) -> torch.Tensor:
    """[V,dim] (reference LFT.py:70 with dim=[2]; divided by len(dim)=1)."""
    return sinusoid_table(V, dim)


def window_mask(h: int, w: int, k: int = WINDOW) -> torch.Tensor:
    """Additive [h*w,h*w] mask: 0 inside the clamped k x k window of the query, -inf outside
    (reference LFT.py:147-162).  The reference clamps the column range with ``h`` (not ``w``) at
    :155; reproduced faithfully, which only matters for h != w."""
    kl, kr = k // 2, k - k // 2
    yy = torch.arange(h).view(h, 1, 1, 1)
    xx = torch.arange(w).view(1, w, 1, 1)
    ky = torch.arange(h).view(1, 1, h, 1)
    kx = torch.arange(w).view(1, 1, 1, w)
    rows = (ky >= (yy - kl).clamp(min=0)) & (ky < (yy + kr).clamp(max=h))
    cols = (kx >= (xx - kl).clamp(min=0)) & (kx < (xx + kr).clamp(max=h))   # sic: h
    inside = (rows & cols).reshape(h * w, h * w)
    m = torch.full((h * w, h * w), float("-inf"))
    m[inside] = 0.0
    return m


def mha(q_in: torch.Tensor, v_in: torch.Tensor, w_in: torch.Tensor, w_out: torch.Tensor,
        mask: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.MultiheadAttention(E, 8, bias=False) with query=key=q_in, value=v_in, [L,N,E] layout
    (reference LFT.py:183-187, 230-233).  q,k projected from q_in, v from v_in; head c//(E/8)."""
    L, N, E = q_in.shape
    d = E // HEADS
    wq, wk, wv = w_in[:E], w_in[E:2 * E], w_in[2 * E:]
    q = (q_in @ wq.t()).reshape(L, N, HEADS, d).permute(1, 2, 0, 3)     # [N,H,L,d]
    k = (q_in @ wk.t()).reshape(L, N, HEADS, d).permute(1, 2, 0, 3)
    v = (v_in @ wv.t()).reshape(L, N, HEADS, d).permute(1, 2, 0, 3)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)          # scale 1/sqrt(d)
    if mask is not None:
        # A query whose mask row is all -inf (only h < w, see window_mask) gets a ZERO attention output in the reference as
        # run under torch >= 2.5 (tests/golden/wide_a2_s2_b1_6x12.npz); torch < 2.5 produced NaN.  Made explicit so the
        # oracle does not depend on the installed torch.
        empty = torch.isinf(mask).all(dim=-1)
        if bool(empty.any()):
            o = torch.where(empty.view(1, 1, L, 1), torch.zeros((), dtype=o.dtype), o)
    o = o.permute(2, 0, 1, 3).reshape(L, N, E)
    return o @ w_out.t()


# Branch override for gradient parity tests.  The network is piecewise linear at every ReLU / LeakyReLU: where a
# pre-activation is within fp32 rounding of 0, two correct fp32 implementations take different branches and their
# gradients differ at the 1e-3 level (worse at small token counts).  With `branch_masks` set to {tag: bool tensor in the
# layout of the pre-activation}, the activation uses the GIVEN branch (True = positive side) instead of sign(z), so a
# backward pass can be checked exactly against an implementation whose forward made the decisions.
branch_masks: Optional[dict] = None
# When a dict: every activation records its pre-activation tensor under its tag (tests compare the oracle's branch decisions
# with the reference's, tests/golden/train_kink_*.npz).
branch_record: Optional[dict] = None


def _act(z: torch.Tensor, tag: str, slope: float) -> torch.Tensor:
    if branch_record is not None:
        branch_record[tag] = z.detach().clone()
    if branch_masks is not None and tag in branch_masks:
        m = branch_masks[tag].to(z.dtype)
        return z * (m + (1.0 - m) * slope)
    return F.relu(z) if slope == 0.0 else F.leaky_relu(z, slope)


def ffn(t: torch.Tensor, ln_w, ln_b, w1, w2, tag: str = "") -> torch.Tensor:
    """LayerNorm -> Linear -> ReLU -> Linear, no biases, dropout 0 (reference LFT.py:135-142, 207-214)."""
    n = F.layer_norm(t, (t.shape[-1],), ln_w, ln_b, 1e-5)
    return _act(n @ w1.t(), tag, 0.0) @ w2.t()


# --------------------------------------------------------------------------------------------
# blocks; activations in the reference's [B,C,V,h,w] layout
# --------------------------------------------------------------------------------------------
def conv_views(x: torch.Tensor, wgt: torch.Tensor) -> torch.Tensor:
    """Conv3d kernel (1,3,3) pad (0,1,1) no bias == per-view 3x3 conv (reference LFT.py:24,27-31)."""
    return F.conv3d(x, wgt, padding=(0, 1, 1))


def init_features(sd: Dict[str, torch.Tensor], lr_views: torch.Tensor) -> torch.Tensor:
    """conv_init0 then 3 x (conv + LeakyReLU 0.2), plus residual (reference LFT.py:65-66)."""
    f0 = conv_views(lr_views, sd["conv_init0.0.weight"])
    f = f0
    for i in (0, 2, 4):
        f = _act(conv_views(f, sd[f"conv_init.{i}.weight"]), f"conv{i}", 0.2)
    return f + f0


def ang_block(sd, l: int, x: torch.Tensor) -> torch.Tensor:
    """AngTrans.forward (reference LFT.py:225-238): tokens 'b c a h w -> a (b h w) c'."""
    p = f"altblock.{l}.ang_trans."
    B, C, V, h, w = x.shape
    t = x.permute(2, 0, 3, 4, 1).reshape(V, B * h * w, C)
    pe = angular_pe(V, C).view(V, 1, C).to(x.dtype)
    n = F.layer_norm(t + pe, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    t = mha(n, t, sd[p + "attention.in_proj_weight"], sd[p + "attention.out_proj.weight"], None) + t
    t = ffn(t, sd[p + "feed_forward.0.weight"], sd[p + "feed_forward.0.bias"],
            sd[p + "feed_forward.1.weight"], sd[p + "feed_forward.4.weight"], f"ang{l}") + t
    return t.reshape(V, B, h, w, C).permute(1, 4, 0, 2, 3)


def spa_tokens(x: torch.Tensor, mlp_w: torch.Tensor) -> torch.Tensor:
    """SpaTrans.SAI2Token (reference LFT.py:164-169): unfold 3x3 (zero pad per view, feature index
    c*9+ky*3+kx) then Linear 576->128.  x [B,C,V,h,w] -> [h*w, B*V, 2C]."""
    B, C, V, h, w = x.shape
    img = x.permute(0, 2, 1, 3, 4).reshape(B * V, C, h, w)
    u = F.unfold(img, kernel_size=3, padding=1).permute(2, 0, 1)
    return u @ mlp_w.t()


def spa_block(sd, l: int, x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SpaTrans.forward (reference LFT.py:176-191) followed by Token2SAI's 1x1x1 conv (:171-174)."""
    p = f"altblock.{l}.spa_trans."
    B, C, V, h, w = x.shape
    if mask is None:
        mask = window_mask(h, w)
    t = spa_tokens(x, sd[p + "MLP.weight"])
    pe = spa_tokens(spatial_pe(h, w, C).view(1, C, 1, h, w).to(x.dtype), sd[p + "MLP.weight"])
    n = F.layer_norm(t + pe, (2 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    t = mha(n, t, sd[p + "attention.in_proj_weight"], sd[p + "attention.out_proj.weight"], mask.to(x.dtype)) + t
    t = ffn(t, sd[p + "feed_forward.0.weight"], sd[p + "feed_forward.0.bias"],
            sd[p + "feed_forward.1.weight"], sd[p + "feed_forward.4.weight"], f"spa{l}") + t
    t = t.reshape(h, w, B, V, 2 * C).permute(2, 4, 3, 0, 1)               # [B,2C,V,h,w]
    return F.conv3d(t, sd[p + "linear.0.weight"])


def upsample(sd, x_mosaic: torch.Tensor, s: int) -> torch.Tensor:
    """1x1 conv -> PixelShuffle(s) -> LeakyReLU 0.2 -> 3x3 conv over the whole mosaic
    (reference LFT.py:39-44, 80)."""
    u = F.conv2d(x_mosaic, sd["upsampling.0.weight"])
    u = F.pixel_shuffle(_act(u, "up", 0.2), s)                            # elementwise, so it commutes with the shuffle
    return F.conv2d(u, sd["upsampling.3.weight"], padding=1)


def forward(sd: Dict[str, torch.Tensor], lr: torch.Tensor, A: int, s: int,
            taps: Optional[dict] = None) -> torch.Tensor:
    """get_model.forward (reference LFT.py:52-83).  lr [B,1,A*h,A*w] float32 -> [B,1,A*h*s,A*w*s].
    ``taps`` (optional dict) receives intermediate activations in [B,C,V,h,w] layout.
    Runs without autograd unless a weight requires grad (loss_and_grads below)."""
    if not any(v.requires_grad for v in sd.values()):
        with torch.no_grad():
            return _forward(sd, lr, A, s, taps)
    return _forward(sd, lr, A, s, taps)


def _forward(sd, lr, A, s, taps):
    skip = bicubic_skip(lr, A, s)
    x = init_features(sd, mosaic_to_views(lr, A))
    h, w = x.shape[-2:]
    mask = window_mask(h, w)
    if taps is not None:
        taps["skip"] = skip
        taps["conv0"] = conv_views(mosaic_to_views(lr, A), sd["conv_init0.0.weight"])
        taps["feat"] = x
    y = x
    for l in range(LAYERS):
        y = ang_block(sd, l, y)                     # angular first (reference LFT.py:249-250)
        if taps is not None:
            taps[f"ang{l}"] = y
        y = spa_block(sd, l, y, mask)
        if taps is not None:
            taps[f"spa{l}"] = y
    y = y + x                                        # reference LFT.py:76
    r = upsample(sd, views_to_mosaic(y, A), s)
    if taps is not None:
        taps["body"] = y
        taps["res"] = r
    return r + skip                                  # reference LFT.py:81


def l1_loss(sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
    """get_loss.forward: mean absolute error (reference LFT.py:269-277)."""
    return (sr - hr).abs().mean()


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """10*log10(1/MSE) on [0,1] data (SURVEY.md 8c parity PSNR)."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)


def state_from_numpy(sd_np) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(v) for k, v in sd_np.items()}


# --------------------------------------------------------------------------------------------
# Scene tiling around the hot path (reference utils/utils.py:91-157, used by test.py:83-101)
# --------------------------------------------------------------------------------------------
def _reflect(i: torch.Tensor, n: int) -> torch.Tensor:
    """Symmetric extension index (edge pixel repeated), as ImageExtend's flipped copies give
    (reference utils.py:126-138): -1 -> 0, -2 -> 1, n -> n-1, n+1 -> n-2."""
    i = torch.where(i < 0, -i - 1, i)
    return torch.where(i >= n, 2 * n - 1 - i, i)


def lf_divide_counts(h0: int, w0: int, patch: int, stride: int):
    """numU, numV of LFdivide (reference utils.py:95-105)."""
    bdr = (patch - stride) // 2
    h, w = h0 + 2 * bdr, w0 + 2 * bdr
    nu = (h - patch) // stride + (2 if (h - patch) % stride else 1)
    nv = (w - patch) // stride + (2 if (w - patch) % stride else 1)
    return nu, nv


def lf_divide(data: torch.Tensor, A: int, patch: int, stride: int) -> torch.Tensor:
    """LFdivide (reference utils.py:91-123): scene mosaic [A*h0, A*w0] -> [numU, numV, A*patch, A*patch].
    Every view is mirror-extended by bdr = (patch-stride)//2, zero-filled up to the last patch, and cut into
    patch x patch crops with the given stride."""
    h0, w0 = data.shape[0] // A, data.shape[1] // A
    bdr = (patch - stride) // 2
    nu, nv = lf_divide_counts(h0, w0, patch, stride)
    views = data.reshape(A, h0, A, w0).permute(0, 2, 1, 3)                    # [u, v, h0, w0]
    ey = (torch.arange(nu).view(-1, 1) * stride + torch.arange(patch).view(1, -1))   # [nu, patch] extended-image rows
    ex = (torch.arange(nv).view(-1, 1) * stride + torch.arange(patch).view(1, -1))
    oky, okx = ey < h0 + 2 * bdr, ex < w0 + 2 * bdr
    sy, sx = _reflect(ey - bdr, h0).clamp(0, h0 - 1), _reflect(ex - bdr, w0).clamp(0, w0 - 1)
    g = views[:, :, sy][:, :, :, :, sx]                                        # [u, v, nu, patch, nv, patch]
    g = g * (oky.view(1, 1, nu, patch, 1, 1) & okx.view(1, 1, 1, 1, nv, patch)).to(g.dtype)
    return g.permute(2, 4, 0, 3, 1, 5).reshape(nu, nv, A * patch, A * patch)


def lf_integrate(sub: torch.Tensor, A: int, pz: int, stride: int, h0: int, w0: int) -> torch.Tensor:
    """LFintegrate (reference utils.py:141-157): [numU, numV, A*pz, A*pz] -> [A, A, h0, w0], keeping the central
    stride x stride region of every patch (pz, stride, h0, w0 already multiplied by the scale factor)."""
    nu, nv = sub.shape[:2]
    bdr = (pz - stride) // 2
    s = sub.reshape(nu, nv, A, pz, A, pz)[:, :, :, bdr:bdr + stride, :, bdr:bdr + stride]     # [ku, kv, u, i, v, j]
    t = s.permute(2, 4, 0, 3, 1, 5).reshape(A, A, nu * stride, nv * stride)
    return t[:, :, :h0, :w0].contiguous()


def views_to_scene_mosaic(x: torch.Tensor) -> torch.Tensor:
    """[A, A, H, W] -> [A*H, A*W] (reference test.py:100-101)."""
    A, _, H, W = x.shape
    return x.permute(0, 2, 1, 3).reshape(A * H, A * W)


# ------------------------------------------------------------------------------------------------
# Training step (reference train.py:74-107): autograd over the restatement above, and torch.optim.Adam's update rule
# (betas (0.9, 0.999), eps 1e-8, weight_decay 0, train.py:77-83) written out.  Pinned by tests/golden/train_*.npz,
# which were produced by the real reference network + torch.optim.Adam (tools/gen_golden.py:train_case).
# ------------------------------------------------------------------------------------------------
def loss_and_grads(sd: Dict[str, torch.Tensor], lr: torch.Tensor, hr: torch.Tensor, A: int, s: int):
    """(loss, out, {name: d loss / d param}) with loss = L1Loss(forward(lr), hr)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    out = forward(leaves, lr, A, s)
    loss = l1_loss(out, hr)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), out.detach(), {k: g for k, g in zip(leaves.keys(), grads)}


def param_grads(sd: Dict[str, torch.Tensor], lr: torch.Tensor, A: int, s: int, dout: torch.Tensor):
    """{name: d<out, dout>/d param}: the vector-Jacobian product of the whole network for a given output gradient."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    out = forward(leaves, lr, A, s)
    grads = torch.autograd.grad(out, list(leaves.values()), grad_outputs=dout)
    return {k: g for k, g in zip(leaves.keys(), grads)}


def adam_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, t: int, lr: float = 2e-4,
                b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """One torch.optim.Adam step (t counts from 1); returns (p, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = v.sqrt() / math.sqrt(1 - b2 ** t) + eps
    return p - (lr / (1 - b1 ** t)) * m / denom, m, v


def train_steps(sd: Dict[str, torch.Tensor], lr: torch.Tensor, hr: torch.Tensor, A: int, s: int, steps: int, rate: float = 2e-4):
    """`steps` Adam steps on one batch; returns (losses, first-step grads, final state)."""
    cur = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    losses, first = [], None
    for t in range(1, steps + 1):
        loss, _, grads = loss_and_grads(cur, lr, hr, A, s)
        losses.append(float(loss))
        if first is None:
            first = grads
        for k in cur:
            cur[k], m[k], v2[k] = adam_update(cur[k], grads[k], m[k], v2[k], t, rate)
    return losses, first, cur
