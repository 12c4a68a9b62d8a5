#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/<name>.json: mean HBM-side bytes per launch and kernel.  FETCH_SIZE is doubled for the kernels whose
reads are wide (16 B/lane) coalesced streams -- the gfx950 correction of the guide; kernels reading dwords
(k_assemble, k_conv0) are left uncorrected and flagged."""
import csv, glob, hashlib, json, os, re, sys, collections
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def source_hash():                      # the same hash bench.py computes: ties this file to the sources it was measured on
    from lft_amd import _lib
    h = hashlib.sha256()
    for name in sorted(_lib.SOURCES):
        h.update(open(os.path.join(_lib.CSRC, name), "rb").read())
    return h.hexdigest()[:16]

def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            if m: acc[m.group(1).replace("k_spa_attn_lds", "k_spa_attn").replace("k_spa_attn_mfma", "k_spa_attn")].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}
F, W = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
narrow = {"k_assemble", "k_conv0", "k_pack", "k_pe_tables", "k_copy_f32"}
res = {}
for k in sorted(set(F) | set(W)):
    corr = 1.0 if k in narrow else 2.0
    res[k] = {"fetch_raw": F.get(k, 0.0), "fetch_correction": corr, "fetch": F.get(k, 0.0) * corr, "write": W.get(k, 0.0),
              "total": F.get(k, 0.0) * corr + W.get(k, 0.0)}
json.dump({"source_hash": source_hash(), "unit": "bytes per launch (mean)", "workload": "bench.py default: A5, 4x, B=4, 32x32 LR, bf16, single stream",
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; counter x 1024; FETCH x2 for wide coalesced reads (gfx950)",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in res.items():
    print(f"{k:12s} fetch {v['fetch']/1e6:8.1f} MB (raw {v['fetch_raw']/1e6:6.1f} x{v['fetch_correction']:.0f})  write {v['write']/1e6:7.1f} MB  total {v['total']/1e6:7.1f} MB")
