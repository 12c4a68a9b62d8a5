#!/usr/bin/env python3
"""Diagnostic: build liblft_hip with -DLFT_EXPERIMENT, run one spatial block, print mean cycles between the
LFT_STAMP() points of k_spa1 / k_spa2 (wave 0 of each workgroup).  GPU box only; never quote its run time."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
# LFT_STAMPS_SO: a library built beforehand with the product flags + -DLFT_EXPERIMENT (hipcc cross-compiles in the build container:
#   python tools/ab_build.py --build stamps:-DLFT_EXPERIMENT   ->  ab_so/liblft_stamps.so); otherwise it is built here, on GPU time.
so = os.environ.get("LFT_STAMPS_SO")
if not so:
    so = os.path.join(ROOT, "gpurun_out", "liblft_hip_stamps.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLFT_EXPERIMENT",
                           os.path.join(ROOT, "lft_amd", "csrc", "lft_api.hip"), "-o", so])
from lft_amd import _lib
_lib.LIB_PATH = so
import numpy as np, torch
from lft_amd.params import deterministic_state
import gpu_util as G
L = _lib.lib()
L.lft_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.lft_debug_clear_stamps.argtypes = []
A, s, B, h, w = 5, 4, 4, 32, 32
pk = G.Packed(deterministic_state(64, s, seed=1), A, h, w, s, "bf16", B)
xin = (torch.randn(B, A * A, h, w, 64, device=G.DEV) * 0.3).to(torch.bfloat16)
act = pk.new_act()
nwg = 800
def run_and_read(which):
    for _ in range(3):
        _lib.check(L.lft_spa_block_fwd(pk.buf.data_ptr(), 1, xin.data_ptr(), None, act.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "spa")
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 32, dtype=np.uint64)
    _lib.check(L.lft_debug_read_stamps(buf.ctypes.data, buf.size), "read")
    return buf.reshape(4096, 32)[:nwg].astype(np.int64)
allst = run_and_read(0)
def report(title, st, names):
    d = np.diff(st, axis=1)
    print(title + ": mean cycles per phase, wave 0 of each workgroup (s_memtime ticks):")
    for i, nm in enumerate(names):
        print(f"  {nm:40s} mean {d[:, i].mean():9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
    print(f"  total {np.mean(st[:, -1] - st[:, 0]):.0f}")
report("k_spa1", allst[:, [0, 1, 2, 3, 4, 6, 7, 9, 11]], ["petok loads, ring init, stage input", "conv 64->128 (144 MFMA, 9 chunks)", "__syncthreads",
                                   "store TOK tile", "V (32 MFMA) + 2 half-tile stores", "+PE, LN, frags", "Q (32 MFMA) + 2 half-tile stores",
                                   "K (32 MFMA) + 2 half-tile stores"])

st = allst
first, second = st[:512], st[512:800]
for nm, g in (("first-round WGs (block < 512)", first), ("second-round WGs", second)):
    print(f"{nm}: issue loads + DMA {np.mean(g[:,12]-g[:,0]):.0f}, wait vmcnt(0) {np.mean(g[:,13]-g[:,12]):.0f}, barrier {np.mean(g[:,1]-g[:,13]):.0f}, "
          f"conv {np.mean(g[:,2]-g[:,1]):.0f}, rest {np.mean(g[:,11]-g[:,2]):.0f}, total {np.mean(g[:,11]-g[:,0]):.0f}")

report("k_spa_b", allst[:, 16:31], ["ring init, params, touch, DMA plan, bias, offsets", "head pair 0: K/V DMA + wait + barrier", "head pair 0: 2 heads",
                                    "head pair 1: DMA + wait", "head pair 1: 2 heads", "head pair 2: DMA + wait", "head pair 2: 2 heads",
                                    "head pair 3: DMA + wait", "head pair 3: 2 heads", "barrier + TOK tile load", "out_proj (32 MFMA, 4 chunks)",
                                    "LN + FFN (128 MFMA, 16 chunks)", "1x1x1 conv (16 MFMA)", "store tile"])

# k_ang: its own launch (stamps 0..11 of the first 512 workgroups; two pixel tiles per wave at this size)
for _ in range(3):
    _lib.check(L.lft_ang_block_fwd(pk.buf.data_ptr(), 1, xin.data_ptr(), act.data_ptr(), *pk.dims(), G.stream()), "ang")
torch.cuda.synchronize()
buf = np.zeros(4096 * 32, dtype=np.uint64)
_lib.check(L.lft_debug_read_stamps(buf.ctypes.data, buf.size), "read")
ang = buf.reshape(4096, 32)[:512].astype(np.int64)
report("k_ang", ang[:, 0:12], ["weights DMA + wait + barrier", "tile 1: load", "tile 1: LN, Q/K/V (24 MFMA)", "tile 1: 8 heads attention (24 MFMA)",
                               "tile 1: out_proj + FFN (40 MFMA)", "tile 1: store", "tile 2: load", "tile 2: LN, Q/K/V", "tile 2: attention",
                               "tile 2: out_proj + FFN", "tile 2: store"])
