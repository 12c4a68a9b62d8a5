#!/usr/bin/env python3
"""Instruction mix per kernel from a gfx950 assembly listing (hipcc -S): counts of MFMA, transcendental, packed, convert,
other VALU, SALU, LDS, vector-memory, waits and barriers between a kernel's label and its s_endpgm -- the static stream,
which for these straight-line (fully unrolled) kernels is close to what one wave executes.

  tools/isa_mix.py file.s [substring ...]        kernels whose demangled name contains every substring
  tools/isa_mix.py --build [substring ...]       compile the inference unit with the product flags first (to /tmp/lft_isa)
  tools/isa_mix.py --phases [substring ...]      the same build with -DLFT_EXPERIMENT -DLFT_ISA_MARKS: every LFT_STAMP(n) of a
                                                 kernel becomes a mark in the listing and the counts are given per phase
                                                 (the instructions between mark n and the next one; "pre" = before the first)
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(extra=()):
    sys.path.insert(0, ROOT)
    from lft_amd import _lib
    os.makedirs("/tmp/lft_isa", exist_ok=True)
    out = "/tmp/lft_isa/tu1.s"
    cmd = ["/opt/rocm/bin/hipcc", *_lib.COMMON_FLAGS, *_lib.UNIT_FLAGS[1], "-DLFT_TU=1", "--cuda-device-only", "-S",
           os.path.join(_lib.CSRC, "lft_api.hip"), "-o", out] + [a for a in sys.argv[1:] if a.startswith("-D")] + list(extra)
    subprocess.run(cmd, check=True)
    return out


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
        return "trans"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_cndmask"):
        return "cnd"
    if op.startswith("v_permlane") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "lane"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("v_mov") or op.startswith("v_nop"):
        return "mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and False):
        return "dma"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "vload"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"):
        return "vstore"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    phases = bool(args) and args[0] == "--phases"
    if phases:
        path, subs = build(["-DLFT_EXPERIMENT", "-DLFT_ISA_MARKS", "-Wno-pass-failed"]), args[1:]
    elif args and args[0] == "--build":
        path, subs = build(), args[1:]
    else:
        path, subs = args[0], args[1:]
    names = {}
    cur, counts, order = None, {}, []
    label = re.compile(r"^(_Z\w+):")
    mark = re.compile(r"^\s*; LFT_MARK (\d+)")
    sym = None
    for line in open(path, errors="replace"):
        m = label.match(line)
        if m and cur is None:
            sym = m.group(1)
            cur = sym
            counts[cur] = collections.Counter()
            order.append(cur)
            names[cur] = sym
            continue
        if cur is None:
            continue
        m = mark.match(line)
        if m:
            if cur == sym:                       # what came before the first mark
                counts[sym + "@pre"] = counts.pop(sym)
                order[order.index(sym)] = sym + "@pre"
                names[sym + "@pre"] = sym
            cur = f"{sym}@{int(m.group(1)):02d}"
            if cur not in counts:
                counts[cur] = collections.Counter()
                order.append(cur)
                names[cur] = sym
            continue
        s = line.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        if op == "s_endpgm":
            cur = None
            continue
        counts[cur][classify(op)] += 1
        if op.startswith("v_"):
            counts[cur]["VALU_total"] += 0 if op.startswith("v_mfma") else 1
    dem = subprocess.run(["c++filt"] + [names.get(o, o).replace("DF16b", "u6__bf16").replace("DF16_", "u8_Float16") for o in order], capture_output=True, text=True).stdout.splitlines()
    dem = [d + ("  [" + o.split("@")[1] + "]" if "@" in o else "") for d, o in zip(dem, order)]
    cols = ["mfma", "VALU_total", "trans", "pk", "cvt", "cnd", "lane", "mov", "valu", "salu", "lds", "dma", "vload", "vstore", "scratch", "wait", "barrier", "s_nop", "branch"]
    print(f"{'kernel':64s} " + " ".join(f"{c[:7]:>7s}" for c in cols))
    for sym, d in zip(order, dem):
        if "k_" not in d or not all(x in d for x in subs):
            continue
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        ph = re.search(r"  \[\w+\]$", d)
        d = re.sub(r"\(.*$", "", d).replace("void ", "") + (ph.group(0) if ph else "")
        c = counts[sym]
        print(f"{d[:64]:64s} " + " ".join(f"{c.get(k, 0):7d}" for k in cols))


if __name__ == "__main__":
    main()
