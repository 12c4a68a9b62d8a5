#!/usr/bin/env python3
"""A/B harness: liblft_hip variants built with extra -D flags, timed by bench.py's per-kernel breakdown, each run
in a fresh subprocess, several interleaved rounds; prints one line per variant and round.  Variants that change results
(LFT_EXP_*) are for timing experiments only.

  tools/ab_build.py --build name1:-DFLAG1,-DFLAG2 name2: ...   build only (hipcc cross-compiles without a GPU) into ab_so/
  tools/ab_build.py name1:-DFLAG1 name2: ...                   build what is missing, then time (GPU box)
  AB_ROUNDS=3 AB_TEST=1 ...                                    rounds; AB_TEST=1 also runs the GPU parity tests per variant
("name:" = no extra flags).  ab_so/ is git-ignored but travels with gpurun, so variants are built here, not on GPU time."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outdir = os.path.join(ROOT, "ab_so"); os.makedirs(outdir, exist_ok=True)
args = sys.argv[1:]
build_only = bool(args) and args[0] == "--build"
if build_only:
    args = args[1:]
variants = [a.split(":", 1) for a in args] or [["base", ""]]
sys.path.insert(0, ROOT)
from lft_amd import _lib
for name, flags in variants:
    so = os.path.join(outdir, f"liblft_{name}.so")
    if os.path.exists(so) and not build_only:
        continue
    try:
        _lib.compile_and_link("/opt/rocm/bin/hipcc", so, extra=[f for f in flags.split(",") if f])     # the library's own two-unit build + the variant's flags
    except _lib.LftError as e:
        raise SystemExit(f"build of {name} failed\n{str(e)[-2000:]}")
if build_only:
    raise SystemExit(0)
extra = os.environ.get("AB_BENCH_ARGS", "").split()
if os.environ.get("AB_TEST"):
    for name, flags in variants:
        env = dict(os.environ, LFT_LIB_PATH=os.path.join(outdir, f"liblft_{name}.so"))
        targets = os.environ.get("AB_TEST_ARGS", "tests/test_gpu_parity.py tests/test_gpu_module.py tests/test_gpu_determinism.py").split()
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu"] + targets, env=env, capture_output=True, text=True, cwd=ROOT)
        print(f"[test] {name:14s} rc={r.returncode} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
        if r.returncode != 0:
            print(r.stdout[-3000:], flush=True)
rounds = int(os.environ.get("AB_ROUNDS", "2"))
if os.environ.get("AB_TEST") == "only":
    rounds = 0
for rnd in range(rounds):
    for name, flags in variants:
        env = dict(os.environ, LFT_LIB_PATH=os.path.join(outdir, f"liblft_{name}.so"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "15", "--no-cpu-baseline", "--no-extras"] + extra,
                           env=env, capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            ks = j["roofline"]["kernels"]
            print(f"[{rnd}] {name:14s} {j['value']:8.0f} patches/s  " + " ".join(f"{k[2:]}={v['ms']*1e3:.0f}" for k, v in ks.items()), flush=True)
        except Exception as e:
            print(name, "FAILED", e, r.stderr[-400:], flush=True)
