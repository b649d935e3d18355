"""Instruction histogram of one kernel in a device assembly listing (hipcc --cuda-device-only -S):
python scripts/asm_histogram.py file.s <substring of the mangled kernel name> [loops]
With 'loops' it also prints, per basic-block label, the VALU / lane-access counts (to tell hot blocks apart)."""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if key in l and re.match(r"^\S+:\s", l + " "))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
c = Counter()
per_block, block = {}, "entry"
for l in body:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        block = m.group(1)
    m = re.match(r"\s+([a-z_0-9]+)\s", l)
    if m:
        op = m.group(1)
        c[op] += 1
        b = per_block.setdefault(block, Counter())
        b["all"] += 1
        if op.startswith("v_"):
            b["valu"] += 1
        if op in ("v_readlane_b32", "v_writelane_b32"):
            b["lane"] += 1
        if op.startswith("v_accvgpr"):
            b["acc"] += 1
        if re.match(r"v_(fma|mul|add)_f64", op):
            b["f64"] += 1
valu = sum(v for k, v in c.items() if k.startswith("v_"))
print("instructions %d, VALU %d" % (sum(c.values()), valu))
for k, v in c.most_common(30):
    print("%6d %s" % (v, k))
if len(sys.argv) > 3:
    for b, k in per_block.items():
        if k["all"] >= 60:
            print("%-14s all %5d valu %5d f64 %5d lane %4d acc %4d" % (b, k["all"], k["valu"], k["f64"], k["lane"], k["acc"]))
