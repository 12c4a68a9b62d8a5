#!/usr/bin/env python3
"""One-off (round 4): re-order DESIGN.md into current state first, history in appendices.  Kept for the record of what moved where."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "DESIGN.md")).read()


def section(title_re, text=src):
    m = re.search(r"^(##+ " + title_re + r".*?)(?=^## |\Z)", text, flags=re.M | re.S)
    assert m, title_re
    return m.group(1).rstrip() + "\n"


intro = src[:src.index("## 0. Coverage")]
s0 = section(r"0\. Coverage")
r3_review = s0[s0.index("### Round 3 against"):]
s0 = s0[:s0.index("### Round 3 against")]
s1, s2, s3, s4 = section(r"1\. The path"), section(r"2\. Precision"), section(r"3\. Data layout"), section(r"4\. Kernel design")
s5, s5b, s5a = section(r"5\. Measurement \(round 3\)"), section(r"5b\. "), section(r"5a\. ")
s6, s7, s8, s9, s10 = section(r"6\. Multi-GPU"), section(r"7\. Correctness"), section(r"8\. Next"), section(r"9\. Training"), section(r"10\. Rows")

new = open(os.path.join(ROOT, "tools", "_design_round4.md")).read()


def part(name):
    m = re.search(r"<!-- PART " + name + r" -->\n(.*?)(?=<!-- PART |\Z)", new, flags=re.S)
    assert m, name
    return m.group(1).rstrip() + "\n"


# the per-kernel table of section 4 is replaced by the generated block
t0 = s4.index("Per-kernel table (bf16, cfg 2")
t1 = s4.index('"Kernel-contract bytes" =')
s4 = s4[:t0] + part("S4_TABLE") + "\n" + s4[t1:]
s4 = s4.replace("Details that matter:\n", "Details that matter (the `k_spa_b` entries describe the round-2/3 kernel where round 4 changed it; §4.1 has the current attention phase):\n")

out = [intro, part("S0"), part("R4_REVIEW"), s1.replace("ABI 4", "ABI 5"), s2, part("S2_ADD"), s3, part("S3_ADD"), s4, part("S4_ADD"), part("S5"),
       s6, part("S6_ADD"), s7.replace("`tests/test_gpu_determinism.py` (400 bit-identical forwards", part("S7_ADD") + "`tests/test_gpu_determinism.py` (400 bit-identical forwards"),
       part("S8"), s9, part("S9_ADD"), s10,
       "## Appendix A. Earlier reviews\n\n" + r3_review.replace("### Round 3 against", "### A.1 Round 3 against"),
       "## Appendix B. Measurement history\n\n" + s5.replace("## 5. Measurement (round 3)", "### B.3 Round 3") + "\n" +
       s5b.replace("## 5b. Measurement history (round 2)", "### B.2 Round 2") + "\n" + s5a.replace("## 5a. Measurement history (round 1)", "### B.1 Round 1") + "\n" +
       "### B.0 The round-3 \"Next (ranked)\" list\n\n" + s8.split("\n", 1)[1]]
open(os.path.join(ROOT, "DESIGN.md"), "w").write("\n".join(x.rstrip() + "\n" for x in out))
print("DESIGN.md rewritten:", sum(x.count("\n") for x in out), "lines")
