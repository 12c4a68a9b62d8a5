#!/usr/bin/env python3
"""Static check of every LDS-DMA pipeline (weight rings, the K / V halo buffers of k_spa_b, the conv input tile, k_ang's resident
weights) in a gfx950 assembly listing (hipcc -S) of the PRODUCT build.

The kernels name their protocol in the listing through comment-only asm statements (LFT_DMA_NOTE in lft_common.cuh):
    ; LFT_NOTE DMA  ring=R slot=S      the global_load_lds pieces that follow fill slot S of ring R
    ; LFT_NOTE USE  ring=R slot=S      from here on the slot is read (ds_read)
    ; LFT_NOTE DONE ring=R slot=S      the reads of the slot have all been issued
    ; LFT_NOTE ALIAS_LT ring=R slot=N  slots < N of ring R lie in K/V buffer 0, the others in buffer 1 (k_spa_b hands the buffers over)
The walk is in file order (the kernels concerned are fully unrolled; notes are only emitted where the slot is a compile-time
constant).  Model: vector-memory operations retire in issue order, `s_waitcnt vmcnt(N)` retires all but the N youngest; an
s_barrier PUBLISHES a slot whose pieces (of this wave; all waves run the same code) have all retired.
Checked, for every slot:
  (1) USE only of a published slot: every piece of its last fill was covered by a counted wait and a barrier after that wait;
  (2) DMA only into a free slot: its previous USE has a DONE, the wave's LDS reads were waited for (lgkmcnt(0)) and a barrier came
      after that -- and the same for a slot it aliases;
  (3) no fill is left unpublished at the end of the kernel while the slot is read.
Reported per kernel as well: global_load_lds instructions outside any noted fill (run-time ring positions in loops: not modelled).

  tools/lds_dma_hazards.py file.s [kernel-name-substring ...]
  tools/lds_dma_hazards.py --build [substring ...]      both units of the library, product flags"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VM = re.compile(r"^\s*(global_load|global_store|buffer_load|buffer_store|scratch_load|scratch_store|flat_load|flat_store|global_atomic)")
NOTE = re.compile(r";\s*LFT_NOTE (\w+) ring=(\d+) slot=(\d+)")
RING = {0: "WRing", 1: "WRingPipe", 2: "K/V buffer", 3: "conv input", 4: "k_ang weights"}
STATS = {}          # kernel -> {"use": n, "dma": n, "unnoted": n}


class Slot:
    def __init__(self):
        self.state = "free"          # free | filling | ready | in_use | done
        self.pieces = []             # indices into the VM queue log of the last fill
        self.fill_line = 0


def check(path, filters):
    bad = []
    kernel = None
    for ln, line in enumerate(open(path, errors="replace"), 1):
        if line.startswith("_Z") and ":" in line and line.rstrip().split(":")[0].startswith("_Z"):
            kernel = line.split(":")[0]
            slots, retired, issued = {}, 0, 0          # VM operations are numbered in issue order; `retired` = how many have completed
            dma_of = {}                                 # VM op number -> (ring, slot)
            open_fill, lgkm_clean, alias_lt = None, True, None
            st = STATS.setdefault(kernel, {"use": 0, "dma": 0, "unnoted": 0})
            continue
        if kernel is None:
            continue
        if filters and not all(f in kernel for f in filters):
            continue
        text = line.split(";")[0] if "LFT_NOTE" not in line else line
        m = NOTE.search(line)
        if m:
            kind, ring, slot = m.group(1), int(m.group(2)), int(m.group(3))
            if kind == "ALIAS_LT":
                alias_lt = (ring, slot)
                continue
            key = (ring, slot)
            s = slots.setdefault(key, Slot())
            if kind == "DMA":
                others = []
                if alias_lt and ring == alias_lt[0]:
                    others.append((2, 0 if slot < alias_lt[1] else 1))
                for k2 in [key] + others:
                    s2 = slots.get(k2)
                    if s2 is not None and s2.state not in ("free",):
                        bad.append((kernel, ln, f"DMA into {RING.get(ring, ring)} slot {slot} while {RING.get(k2[0], k2[0])} slot {k2[1]} is '{s2.state}' "
                                                f"(its reads were not retired by lgkmcnt(0) + barrier)"))
                s.state, s.pieces, s.fill_line = "filling", [], ln
                open_fill = key
            elif kind == "USE":
                st["use"] += 1
                if s.state != "ready":
                    pend = [p for p in s.pieces if p >= retired]
                    bad.append((kernel, ln, f"USE of {RING.get(ring, ring)} slot {slot} in state '{s.state}' (fill at line {s.fill_line}: {len(s.pieces)} pieces, "
                                            f"{len(pend)} not retired; a retired fill still needs a barrier after its wait)"))
                s.state = "in_use"
            elif kind == "DONE":
                if s.state != "in_use":
                    bad.append((kernel, ln, f"DONE of {RING.get(ring, ring)} slot {slot} in state '{s.state}'"))
                s.state = "done"
            continue
        body = text.strip()
        if not body or body.endswith(":") or body.startswith("."):
            continue
        op = body.split()[0]
        if op == "s_endpgm":
            for key, s in slots.items():
                if s.state == "filling" and s.pieces:
                    pass                                   # a fill nobody reads any more (the stream's tail) is harmless
            kernel = None
            continue
        if op == "s_waitcnt":
            mv = re.search(r"vmcnt\((\d+)\)", body)
            if mv:
                retired = max(retired, issued - int(mv.group(1)))
            if "lgkmcnt(0)" in body:
                lgkm_clean = True
            continue
        if op == "s_barrier":
            open_fill = None
            for key, s in slots.items():
                if s.state == "filling" and s.pieces and all(p < retired for p in s.pieces):
                    s.state = "ready"
                elif s.state == "done" and lgkm_clean:
                    s.state = "free"
            continue
        if op.startswith("ds_read") or op.startswith("ds_load"):
            lgkm_clean = False
            continue
        if VM.match(body):
            if op.startswith("global_load_lds"):
                if open_fill is not None:
                    slots[open_fill].pieces.append(issued)
                    dma_of[issued] = open_fill
                    st["dma"] += 1
                else:
                    st["unnoted"] += 1
            issued += 1
    return bad


EXPECT_USES = {"7k_spa_b": 20, "6k_spa1": 10, "8k_conv64": 5, "5k_angI": 1}      # at least this many USE notes must have been checked per instantiation


def main():
    args = sys.argv[1:]
    paths = []
    if args and args[0] == "--build":
        sys.path.insert(0, ROOT)
        from lft_amd import _lib
        os.makedirs("/tmp/lft_isa", exist_ok=True)
        for unit in (1, 2):
            out = f"/tmp/lft_isa/hz{unit}.s"
            if not os.environ.get("LFT_HAZARD_REUSE") or not os.path.exists(out):
                subprocess.run(["/opt/rocm/bin/hipcc", *_lib.COMMON_FLAGS, *_lib.UNIT_FLAGS[unit], f"-DLFT_TU={unit}", "--cuda-device-only", "-S",
                                os.path.join(_lib.CSRC, "lft_api.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
            paths.append(out)
        filters = args[1:]
    else:
        paths, filters = [args[0]], args[1:]
    bad = []
    for p in paths:
        bad += check(p, filters)
    for kernel, ln, msg in bad:
        print(f"{kernel}:{ln}: {msg}")
    checked = {k: v for k, v in STATS.items() if v["use"] or v["dma"] or v["unnoted"]}
    print(f"{len(bad)} LDS-DMA protocol violation(s); {sum(v['use'] for v in checked.values())} slot uses and {sum(v['dma'] for v in checked.values())} DMA pieces "
          f"checked in {sum(1 for v in checked.values() if v['use'])} kernels")
    unn = {k: v["unnoted"] for k, v in checked.items() if v["unnoted"]}
    if unn:
        print("not modelled (DMA at run-time ring positions, i.e. inside loops): " + ", ".join(f"{k[:40]}..: {n}" for k, n in sorted(unn.items())[:12]) + (" ..." if len(unn) > 12 else ""))
    vacuous = []
    if not filters:
        for frag, least in EXPECT_USES.items():
            ks = [k for k in STATS if frag in k]
            if not ks or any(STATS[k]["use"] < least for k in ks if "Lb1ELi" not in k or True):
                low = [f"{k[:48]}: {STATS[k]['use']}" for k in ks if STATS[k]["use"] < least]
                if not ks or low:
                    vacuous.append(f"*{frag}*: expected >= {least} checked uses per instantiation; " + ("no such kernel" if not ks else "; ".join(low[:4])))
    for v in vacuous:
        print("VACUOUS: " + v)
    return 1 if bad or vacuous else 0


if __name__ == "__main__":
    sys.exit(main())
