#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing (hipcc -S) for the hazard that bit k_linr's first version: a register filled by an
inline-asm global load (which hipcc believes complete the moment the asm statement ends) is READ -- typically copied, to set up
a tied asm operand -- before the counted s_waitcnt that guards it.

The listing is walked in file order (the kernels concerned are straight-line between their barriers; loops make the walk an
approximation, reported as such).  Every vector-memory operation enters an in-order queue; `s_waitcnt vmcnt(N)` retires all but
the N youngest (vmcnt retires in issue order).  A register written by an ASM load (inside ;;#ASMSTART / ;;#ASMEND) that is still
in the queue and appears as a source operand of any later instruction is reported.

  tools/asm_load_hazards.py file.s [kernel-name-substring ...]
  tools/asm_load_hazards.py --build [substring ...]      both units of the library, product flags"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VM = re.compile(r"^\s*(global_load|global_store|buffer_load|buffer_store|scratch_load|scratch_store|flat_load|flat_store|global_atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(line):
    """(destination registers, source registers) of one instruction line."""
    body = line.split(";")[0].strip()
    if not body or body.endswith(":") or body.startswith("."):
        return set(), set()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return set(), set()
    ops = parts[1].split(",")
    op = parts[0]
    if op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "ds_write", "ds_store", "s_", "v_cmp", "global_load_lds")):
        return set(), regs(parts[1])
    return regs(ops[0]), regs(",".join(ops[1:]))


ASM_LOADS = {}          # kernel symbol -> inline-asm global loads seen (a parsing change must not make the check pass vacuously)


def check(path, filters):
    kernel, in_asm, queue, bad = None, False, [], []        # queue: (is_asm_load, destination registers, line number)
    for ln, line in enumerate(open(path), 1):
        if line.startswith("_Z") and line.rstrip().split(":")[0].startswith("_Z") and ":" in line:
            kernel = line.split(":")[0]
            queue = []
            continue
        if kernel is None or (filters and not all(f in kernel for f in filters)):
            continue
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        text = line.split(";")[0]
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", text)
        if m:
            n = int(m.group(1))
            if n == 0:
                queue = []
            elif n < len(queue):                 # n >= len(queue): nothing retires (a negative slice start would DROP pending loads)
                queue = queue[len(queue) - n:]
            continue
        if "s_endpgm" in text:
            kernel, queue = None, []
            continue
        dst, src = split_operands(line)
        pending = set().union(*[d for a, d, _ in queue if a]) if queue else set()
        hit = (src | (dst if not VM.match(text) else set())) & pending      # a read, or an overwrite by something that is not a load
        if hit and not in_asm:
            bad.append((kernel, ln, sorted(hit), text.strip()))
        if VM.match(text):
            is_load = "load" in text and "lds" not in text
            queue.append((in_asm and is_load, dst if is_load else set(), ln))
            if in_asm and is_load:
                ASM_LOADS[kernel] = ASM_LOADS.get(kernel, 0) + 1
    return bad


# kernels that are KNOWN to use inline-asm loads under counted waits (mangled-name fragments): the check must have seen them
EXPECT_ASM_LOADS = ("7k_spa_b",)            # (k_linr's rows became ordinary loads in round 3)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--build":
        sys.path.insert(0, ROOT)
        from lft_amd import _lib
        os.makedirs("/tmp/lft_isa", exist_ok=True)
        bad = []
        for unit in (1, 2):
            out = f"/tmp/lft_isa/hz{unit}.s"
            subprocess.run(["/opt/rocm/bin/hipcc", *_lib.COMMON_FLAGS, *_lib.UNIT_FLAGS[unit], f"-DLFT_TU={unit}", "--cuda-device-only", "-S",
                            os.path.join(_lib.CSRC, "lft_api.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
            bad += check(out, args[1:])
    else:
        bad = check(args[0], args[1:])
    for kernel, ln, hit, text in bad:
        print(f"{kernel}:{ln}: v{hit} read before its asm load is waited for: {text}")
    print(f"{len(bad)} suspicious read(s)")
    print(f"inline-asm loads modelled: {sum(ASM_LOADS.values())} in {len(ASM_LOADS)} kernels")
    vacuous = []
    if args and args[0] == "--build" and not args[1:]:
        vacuous = [k for k in EXPECT_ASM_LOADS if not any(k in name for name in ASM_LOADS)]
        for k in vacuous:
            print(f"NO inline-asm load was matched in any kernel named *{k}*: the listing format or the kernels changed -- the check is vacuous")
    return 1 if bad or vacuous else 0


if __name__ == "__main__":
    sys.exit(main())
