#!/usr/bin/env python3
"""Copy the rocprofv3 summaries collected by scripts/collect_profiles.sh into profiles/ and derive
profiles/traffic.json (HBM bytes per k_solve launch) from the FETCH_SIZE / WRITE_SIZE passes.

  python scripts/summarise_profiles.py gpurun_out/prof_<tag> <name-prefix>

FETCH_SIZE on gfx950 reports half the bytes actually fetched.  The pack kernel (k_pack_jobs), whose traffic is known exactly (it
reads every element once and writes it once), calibrates the factor in the same run (k_unpack does not
qualify: with per-problem source selection it touches cache lines of both copies).  WRITE_SIZE needs no
correction.
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")


def counters(path):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        key = next((k for k in ("k_solve", "k_pack_rows", "k_pack", "k_unpack") if k in name), None)
        if key:
            tot[(key, r["Counter_Name"])] += float(r["Counter_Value"])
            n[(key, r["Counter_Name"])] += 1
    return tot, n


shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, prefix + "_kernel_stats.csv"))
for kind in ("fetch", "write", "sq"):
    p = os.path.join(src, kind, kind + "_counter_collection.csv")
    if os.path.exists(p):
        # keep only the library's kernels (the torch fill kernels are noise) to keep the file small
        rows = [r for r in csv.reader(open(p))]
        keep = [rows[0]] + [r for r in rows[1:] if "k_" in ",".join(r)]
        with open(os.path.join(dst, "%s_pmc_%s.csv" % (prefix, kind)), "w", newline="") as fh:
            csv.writer(fh, quoting=csv.QUOTE_MINIMAL).writerows(keep)

f, fn = counters(os.path.join(src, "fetch", "fetch_counter_collection.csv"))
w, wn = counters(os.path.join(src, "write", "write_counter_collection.csv"))
fetch_kib = f[("k_solve", "FETCH_SIZE")] / fn[("k_solve", "FETCH_SIZE")]
write_kib = w[("k_solve", "WRITE_SIZE")] / wn[("k_solve", "WRITE_SIZE")]
# calibration: k_pack reads exactly what it writes
cal_w = w[("k_pack", "WRITE_SIZE")]
cal_f = f[("k_pack", "FETCH_SIZE")]
factor = cal_w / cal_f
out = {
    "kernel": "k_solve",
    "batch": 16384,
    "workload": "bench.py default (cart-pole swing-up, T=100, <=200 iterations)",
    "fetch_size_kib_raw": fetch_kib,
    "write_size_kib_raw": write_kib,
    "fetch_correction_measured": factor,
    "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
    "correction": "FETCH_SIZE x2 on gfx950 (the transpose k_pack reads what it writes; in the same run its "
                  "WRITE_SIZE / FETCH_SIZE is %.3f), WRITE_SIZE as reported; separate --pmc passes" % factor,
    "source": ["profiles/%s_pmc_fetch.csv" % prefix, "profiles/%s_pmc_write.csv" % prefix],
}
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
