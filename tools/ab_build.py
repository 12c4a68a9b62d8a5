#!/usr/bin/env python3
"""A/B harness (GPU box): build liblft_hip variants with extra -D flags, run the bench's per-kernel breakdown
for each variant in a fresh subprocess, print one line per variant.  Variants that change results
(LFT_EXP_*) are for timing experiments only.
usage: tools/ab_build.py name1:-DFLAG1,-DFLAG2 name2: ...      ("name:" = no extra flags)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outdir = os.path.join(ROOT, "gpurun_out", "ab"); os.makedirs(outdir, exist_ok=True)
variants = [a.split(":", 1) for a in sys.argv[1:]] or [["base", ""]]
for name, flags in variants:
    so = os.path.join(outdir, f"liblft_{name}.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + [f for f in flags.split(",") if f] + \
          [os.path.join(ROOT, "lft_amd/csrc/lft_api.hip"), "-o", so]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
rounds = int(os.environ.get("AB_ROUNDS", "2"))
for rnd in range(rounds):
    for name, flags in variants:
        env = dict(os.environ, LFT_LIB_PATH=os.path.join(outdir, f"liblft_{name}.so"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "15", "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            ks = j["roofline"]["kernels"]
            print(f"[{rnd}] {name:14s} {j['value']:8.0f} patches/s  " + " ".join(f"{k[2:]}={v['ms']*1e3:.0f}" for k, v in ks.items()), flush=True)
        except Exception as e:
            print(name, "FAILED", e, r.stderr[-400:], flush=True)
