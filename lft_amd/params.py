"""Parameter table of the LFT light-field SR network and a portable deterministic initialiser.

The table reproduces the 78 state-dict entries (names, shapes, registration order) of the
reference model (reference model/LFT.py:9-44 builds conv_init0 / conv_init / altblock /
upsampling; :119-145 SpaTrans; :195-214 AngTrans; :245-246 registers spa_trans before
ang_trans) so a reference ``.pth`` checkpoint loads unchanged.

The initialiser draws from numpy's PCG64 keyed by (seed, entry index) so the GPU box can
regenerate exactly the weights the golden fixtures were made with, without any weight blob.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import List, Tuple

import numpy as np

NUM_LAYERS = 4          # reference LFT.py:15
NUM_HEADS = 8           # reference LFT.py:19
LN_EPS = 1e-5           # nn.LayerNorm default, reference LFT.py:127,136,199,208
LRELU_SLOPE = 0.2       # reference LFT.py:28,42
PE_TEMPERATURE = 10000  # reference LFT.py:17
SPA_WINDOW = 5          # kernel_search, reference LFT.py:123
SPA_FIELD = 3           # kernel_field, reference LFT.py:122


def param_table(channels: int = 64, scale: int = 2) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(name, shape, kind)] in reference registration order.

    kind: 'w<fan_in>' uniform(-1/sqrt(fan_in), +1/sqrt(fan_in)) -- what kaiming_uniform_(a=sqrt(5))
    gives for Conv/Linear defaults and for the explicit in_proj init (reference LFT.py:132,204);
    'ln_w' ones; 'ln_b' zeros.
    """
    C = channels
    E = 2 * C
    t: List[Tuple[str, Tuple[int, ...], str]] = []
    t.append(("conv_init0.0.weight", (C, 1, 1, 3, 3), "w9"))
    for i in (0, 2, 4):
        t.append((f"conv_init.{i}.weight", (C, C, 1, 3, 3), f"w{9 * C}"))
    for l in range(NUM_LAYERS):
        p = f"altblock.{l}.spa_trans."
        t.append((p + "MLP.weight", (E, 9 * C), f"w{9 * C}"))
        t.append((p + "norm.weight", (E,), "ln_w"))
        t.append((p + "norm.bias", (E,), "ln_b"))
        t.append((p + "attention.in_proj_weight", (3 * E, E), f"w{E}"))
        t.append((p + "attention.out_proj.weight", (E, E), f"w{E}"))
        t.append((p + "feed_forward.0.weight", (E,), "ln_w"))
        t.append((p + "feed_forward.0.bias", (E,), "ln_b"))
        t.append((p + "feed_forward.1.weight", (2 * E, E), f"w{E}"))
        t.append((p + "feed_forward.4.weight", (E, 2 * E), f"w{2 * E}"))
        t.append((p + "linear.0.weight", (C, E, 1, 1, 1), f"w{E}"))
        p = f"altblock.{l}.ang_trans."
        t.append((p + "norm.weight", (C,), "ln_w"))
        t.append((p + "norm.bias", (C,), "ln_b"))
        t.append((p + "attention.in_proj_weight", (3 * C, C), f"w{C}"))
        t.append((p + "attention.out_proj.weight", (C, C), f"w{C}"))
        t.append((p + "feed_forward.0.weight", (C,), "ln_w"))
        t.append((p + "feed_forward.0.bias", (C,), "ln_b"))
        t.append((p + "feed_forward.1.weight", (2 * C, C), f"w{C}"))
        t.append((p + "feed_forward.4.weight", (C, 2 * C), f"w{2 * C}"))
    t.append(("upsampling.0.weight", (C * scale * scale, C, 1, 1), f"w{C}"))
    t.append(("upsampling.3.weight", (1, C, 3, 3), f"w{9 * C}"))
    return t


def deterministic_state(channels: int = 64, scale: int = 2, seed: int = 1,
                        flavor: str = "default", gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Seeded float32 weights in the reference's default-init ranges.

    flavor 'default': LayerNorm weight 1 / bias 0 (PyTorch default).
    flavor 'stress' : LayerNorm weight U(0.5,1.5), bias U(-0.2,0.2), so affine terms are exercised.
    gain multiplies every non-LayerNorm weight (used to make the residual branch non-negligible).
    """
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for idx, (name, shape, kind) in enumerate(param_table(channels, scale)):
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        if kind.startswith("w"):
            bound = gain / math.sqrt(int(kind[1:]))
            a = rng.uniform(-bound, bound, size=shape)
        elif kind == "ln_w":
            a = np.ones(shape) if flavor == "default" else rng.uniform(0.5, 1.5, size=shape)
        elif kind == "ln_b":
            a = np.zeros(shape) if flavor == "default" else rng.uniform(-0.2, 0.2, size=shape)
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def synthetic_lr(batch: int, ang: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """U[0,1) float32 LR mosaic [B,1,A*h,A*w] (SURVEY 8d synthetic input)."""
    rng = np.random.Generator(np.random.PCG64([seed, batch, ang, h, w]))
    return rng.random(size=(batch, 1, ang * h, ang * w), dtype=np.float32)


def num_params(channels: int = 64, scale: int = 2) -> int:
    return sum(int(np.prod(s)) for _, s, _ in param_table(channels, scale))
