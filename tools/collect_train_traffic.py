#!/usr/bin/env python3
"""Two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of tools/train_bench.py --no-graph --warmup 0 --steps N ->
profiles/<name>.json: HBM-side bytes PER TRAINING STEP and kernel family (all launches of a step summed).  As in
tools/collect_traffic.py, FETCH_SIZE is doubled for the families that read wide (16 B per lane) coalesced streams -- the gfx950
correction of MI355X_MICROARCH.md -- and left as counted for the ones that read dwords (k_wgrad, reductions).

  tools/collect_train_traffic.py <fetch dir> <write dir> <steps profiled> <math> <out.json>"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

fetch_dir, write_dir, steps, math, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def source_hash():
    from lft_amd import _lib
    h = hashlib.sha256()
    for name in sorted(_lib.SOURCES):
        h.update(open(os.path.join(_lib.CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def load(d, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            if m:
                fam = {"k_linr": "k_lin"}.get(m.group(1), m.group(1))          # the ring-fed form belongs to the k_lin family (bench.train_gemm_work)
                acc[fam] += float(r["Counter_Value"]) * 1024.0
                n[fam] += 1
    return acc, n


F, nf = load(fetch_dir, "FETCH_SIZE")
W, _ = load(write_dir, "WRITE_SIZE")
narrow = {"k_wgrad", "k_reduce_all", "k_sum_images", "k_conv0", "k_conv0_wgrad", "k_pack", "k_pack_split", "k_pe_plain", "k_l1_partial", "k_l1_final",
          "k_up_gather_bwd", "k_upm_fold", "k_assemble_t"}
res = {}
for k in sorted(set(F) | set(W)):
    corr = 1.0 if k in narrow else 2.0
    res[k] = {"launches_per_step": nf.get(k, 0) / steps, "fetch_raw": F.get(k, 0.0) / steps, "fetch_correction": corr, "fetch": F.get(k, 0.0) * corr / steps,
              "write": W.get(k, 0.0) / steps, "total": (F.get(k, 0.0) * corr + W.get(k, 0.0)) / steps}
json.dump({"source_hash": source_hash(), "unit": "bytes per training step (all launches of the family summed)", "math": math,
           "workload": "tools/train_bench.py: A5, 2x, B=8, 32x32 LR (BASELINE configs[2] shape on one GPU), eager launches",
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; counter x 1024; FETCH x2 for wide coalesced reads (gfx950)",
           "steps_profiled": steps, "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["total"]):
    print(f"{k:16s} {v['launches_per_step']:6.1f} launches  fetch {v['fetch'] / 1e9:7.2f} GB (raw {v['fetch_raw'] / 1e9:6.2f} x{v['fetch_correction']:.0f})  "
          f"write {v['write'] / 1e9:6.2f} GB  total {v['total'] / 1e9:7.2f} GB")
