"""ctypes binding of liblft_hip.so (C ABI declared in include/lft_hip.h) and its in-tree build.

The product path has NO CPU fallback: if the library is missing or fails to load, every entry
point raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or ``build()`` here.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("LFT_LIB_PATH") or os.path.join(HERE, "liblft_hip.so")   # LFT_LIB_PATH: experiment builds (tools/ab_build.py)
SOURCES = ["lft_api.hip", "lft_common.cuh", "lft_kernels_a.cuh", "lft_kernels_b.cuh", "lft_train.cuh", "lft_train_host.cuh", "lft_metrics.cuh"]
ABI_VERSION = 5                      # LFT_ABI_VERSION of include/lft_hip.h: lib() refuses a library that reports another one
STATUS_NONFINITE = 1001              # LFT_STATUS_NONFINITE

PREC_F32, PREC_BF16, PREC_F16 = 0, 1, 2
MATH_F32, MATH_BF16X3, MATH_BF16X6 = 0, 1, 2
NUM_PARAMS = 78

_lib = None


class LftError(RuntimeError):
    pass


def _stamp_path(lib_path: str) -> str:
    d, _ = _objects(lib_path)
    return os.path.join(d, os.path.basename(lib_path) + ".flags")


def _flags_stamp() -> str:
    """What the library was built with: the compile / link commands themselves (the per-unit flags live in this file, not
    in a source the mtime check sees)."""
    return "\n".join(" ".join(c) for c in build_commands("hipcc", "liblft_hip.so"))


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "lft_hip.h")]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    # a change of compiler flags must rebuild too; a shipped library without a stamp (the GPU box gets the .so, not the
    # build directory) is taken as current -- its ABI version is still checked at load time
    stamp = _stamp_path(LIB_PATH)
    return os.path.exists(stamp) and open(stamp).read() != _flags_stamp()


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles for gfx950 (works without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for cmd in build_commands(hipcc, LIB_PATH):
        if verbose:
            print(" ".join(cmd))
    compile_and_link(hipcc, LIB_PATH)
    with open(_stamp_path(LIB_PATH), "w") as f:
        f.write(_flags_stamp())
    return LIB_PATH


# Flags of both translation units (lft_api.hip compiles as LFT_TU=1, the inference side, and LFT_TU=2, the training step: two
# units only so that they build in parallel).
# -amdgpu-schedule-relaxed-occupancy: the kernels' occupancy is set by their LDS footprint and launch bounds, so the
#   scheduler may spend registers on a better instruction order (+1.2 %, k_spa1 43 -> 41 us event-timed); no arithmetic changes.
# -fno-slp-vectorize: hipcc's SLP pass packs neighbouring scalar f32 operations into v_pk_*_f32, which issue slower beside
#   MFMAs than the two scalar instructions (+1.4 % on the inference bench).  Until round 3 the training unit kept the default
#   because the pass also decides which fp32 multiply-adds get contracted, enough to flip a ReLU unit in the screened-seed
#   gradient test; the kink-aligned gradient tests (tests/test_gpu_train.py) pass with and without the flag, so the flags
#   are no longer pinned by a test and both units use the same set.
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-schedule-relaxed-occupancy=true", "-fPIC"]
UNIT_FLAGS = {1: ["-fno-slp-vectorize"], 2: ["-fno-slp-vectorize"]}


def _objects(lib_path):
    d = os.path.join(os.path.dirname(lib_path), "_build")
    return d, {tu: os.path.join(d, os.path.basename(lib_path) + f".tu{tu}.o") for tu in UNIT_FLAGS}


def build_commands(hipcc, lib_path, extra=()):
    _, objs = _objects(lib_path)
    src = os.path.join(CSRC, "lft_api.hip")
    cmds = [[hipcc, *COMMON_FLAGS, *UNIT_FLAGS[tu], *extra, f"-DLFT_TU={tu}", "-c", src, "-o", objs[tu]] for tu in sorted(objs)]
    cmds.append([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *[objs[tu] for tu in sorted(objs)], "-o", lib_path])
    return cmds


def compile_and_link(hipcc, lib_path, extra=()):
    """Both units in parallel, then the link; raises LftError with the compiler output on failure."""
    d, _ = _objects(lib_path)
    os.makedirs(d, exist_ok=True)
    cmds = build_commands(hipcc, lib_path, extra)
    procs = [subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for c in cmds[:-1]]
    outs = [p.communicate()[0] for p in procs]
    for p, o, c in zip(procs, outs, cmds):
        if p.returncode != 0:
            raise LftError("hipcc failed: " + " ".join(c) + "\n" + o)
    r = subprocess.run(cmds[-1], capture_output=True, text=True)
    if r.returncode != 0:
        raise LftError("link failed:\n" + r.stdout + r.stderr)

_SIGS = {
    "lft_version": (c_int, []),
    "lft_last_error": (c_char_p, []),
    "lft_packed_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "lft_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "lft_pack_weights": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_status_reset": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_status_read": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, POINTER(ctypes.c_uint)]),
    "lft_forward_profiled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_int, POINTER(c_float), POINTER(c_char_p), POINTER(c_int)]),
    "lft_kernel_time": (c_int, [c_char_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, POINTER(c_float)]),
    "lft_bicubic_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_init_features_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_ang_block_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_spa_block_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_upsample_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_scene_counts": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "lft_scene_divide": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_scene_integrate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_train_tape_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "lft_train_grad_floats": (c_int, [c_int, POINTER(c_size_t)]),
    "lft_train_tape_offset": (c_int, [c_char_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "lft_train_forward": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_train_backward": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_train_backward_buckets": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p]),
    "lft_train_block_backward": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_train_grad_bucket": (c_int, [c_int, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "lft_train_step_profiled": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_int, POINTER(c_float), POINTER(c_char_p), POINTER(c_int)]),
    "lft_l1_loss": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "lft_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_float, c_float, c_float, c_int, c_float, c_float, c_void_p]),
    "lft_view_metrics_scratch_bytes": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "lft_view_metrics": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    # test-only entry points (include/lft_hip_test.h)
    "lft_debug_conv64": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lft_mfma_selftest": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
}
EXPORTS = tuple(_SIGS)
TEST_EXPORTS = ("lft_debug_conv64", "lft_mfma_selftest")         # declared in include/lft_hip_test.h, not in the product header
BLOCK_UPSAMPLE, BLOCK_SPA, BLOCK_ANG, BLOCK_INIT = 0, 1, 2, 3    # LFT_BLOCK_* of include/lft_hip.h
GRAD_BUCKETS = 3
BUCKET_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_size_t, c_size_t)     # lft_bucket_fn of include/lft_hip.h: 0 = go on, else stop


def lib() -> ctypes.CDLL:
    """Load (once) the in-tree shared library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LftError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                           "(run __graft_entry__.build()); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        L.lft_version.restype, L.lft_version.argtypes = c_int, []
        got = L.lft_version()
        # (tools/ab_build.py times inference kernels of older experiment builds: LFT_AB_ANY_ABI=1 lets it load them -- never set it elsewhere)
        if got != ABI_VERSION and not (os.environ.get("LFT_AB_ANY_ABI") == "1" and os.environ.get("LFT_LIB_PATH")):   # a stale or foreign LFT_LIB_PATH build: its entry points may take other arguments
            raise LftError(f"{LIB_PATH} reports ABI version {got}, this binding needs {ABI_VERSION}: rebuild it (__graft_entry__.build())")
        for name, (res, args) in _SIGS.items():
            if not hasattr(L, name) and got != ABI_VERSION:
                continue                        # an older experiment build under LFT_AB_ANY_ABI
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().lft_last_error()
        raise LftError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def packed_bytes(A, h, w, s, prec) -> int:
    n = c_size_t(0)
    check(lib().lft_packed_bytes(A, h, w, s, prec, ctypes.byref(n)), "lft_packed_bytes")
    return n.value


def workspace_bytes(B, A, h, w, s, prec) -> int:
    n = c_size_t(0)
    check(lib().lft_workspace_bytes(B, A, h, w, s, prec, ctypes.byref(n)), "lft_workspace_bytes")
    return n.value
