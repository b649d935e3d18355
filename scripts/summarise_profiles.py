#!/usr/bin/env python3
"""Copy the rocprofv3 summaries collected by scripts/collect_profiles.sh (or collect_profiles_large.sh) into
profiles/ and add this workload's entry to profiles/solve_counters.json, the file bench.py reads its
PMC-derived roofline numbers from.

  python scripts/summarise_profiles.py gpurun_out/prof_<tag> <name-prefix> <workload> <batch>
  python scripts/summarise_profiles.py --bench-line <output of a later bench.py run> profiles/<name>.json

workload: "cartpole" (kernel k_solve) or "large" (kernel k_large_solve).  Per launch of that kernel:
  * average duration                       (--kernel-trace --stats pass)
  * HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  On gfx950 FETCH_SIZE reports half the bytes of
    a streaming read; the factor is checked in the same session on k_solve launches of known byte count
    (scripts/fetch_calibration.py: 8-byte-per-lane accesses into the wave-tiled workspace, 32 and 64 lanes
    per wave), and the measured factors are recorded beside the numbers they correct.
  * SQ counters and the fractions derived from them.
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict


def keep_bench_line(log, out):
    """--bench-line <bench.py output> <profiles/name.json>: keep the JSON line of a bench run as a profile
    record, REFUSING a line whose live kernel time and whose embedded PMC profile belong to different builds
    (roofline.kernel_ms vs roofline.counters.kernel_avg_ms more than 3 % apart)."""
    lines = [ln for ln in open(log) if ln.startswith("{")]
    if len(lines) != 1:
        raise SystemExit("%s holds %d JSON lines" % (log, len(lines)))
    line = json.loads(lines[0])
    roof = line.get("roofline", {})
    cnt = roof.get("counters")
    if cnt is None or not cnt.get("kernel_avg_ms"):
        raise SystemExit("refused: the line embeds no PMC profile (profiles/solve_counters.json has no entry for it)")
    live, prof = roof["kernel_ms"], cnt["kernel_avg_ms"]
    if abs(live - prof) > 0.03 * prof:
        raise SystemExit("refused: the run's kernel time %.3f ms and the embedded profile's %.3f ms differ by more "
                         "than 3 %%: collect the profile again from this build" % (live, prof))
    with open(out, "w") as fh:
        fh.write(json.dumps(line) + "\n")
    print("%s: kernel %.3f ms (profile %.3f ms), value %.1f %s" % (out, live, prof, line["value"], line["unit"]))


if len(sys.argv) > 1 and sys.argv[1] == "--bench-line":
    keep_bench_line(sys.argv[2], sys.argv[3])
    raise SystemExit(0)

src, prefix, workload, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
kernel_key = {"cartpole": "k_solve", "large": "k_large_solve"}[workload]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
N_SIMD = 1024


def per_launch(path, key):
    """counter -> mean over the launches of the kernel whose name contains `key`; also (grid, workgroup)."""
    tot, n, shape = defaultdict(float), defaultdict(int), None
    for r in csv.DictReader(open(path)):
        if key in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
            shape = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]),
                     int(r["Scratch_Size"]), int(r["LDS_Block_Size"]))
    return {c: tot[c] / n[c] for c in tot}, shape


def keep_library_rows(path_in, path_out):
    rows = [r for r in csv.reader(open(path_in))]
    keep = [rows[0]] + [r for r in rows[1:] if "k_" in ",".join(r)]
    with open(path_out, "w", newline="") as fh:
        csv.writer(fh, quoting=csv.QUOTE_MINIMAL).writerows(keep)


import subprocess  # noqa: E402

try:
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                          check=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "ilqr_amd", "include"],
                                capture_output=True, text=True, check=True).stdout.strip())
except Exception:
    head, dirty = None, None
entry = {"workload": workload, "batch": batch, "kernel": kernel_key, "git_head": head, "sources_modified": dirty,
         "source": []}
stats = os.path.join(src, "stats", "stats_kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, prefix + "_kernel_stats.csv"))
entry["source"].append("profiles/%s_kernel_stats.csv" % prefix)
for r in csv.DictReader(open(stats)):
    if kernel_key in r["Name"]:
        entry["kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
        entry["kernel_calls"] = int(r["Calls"])
        entry["kernel_share_of_gpu_time_pct"] = float(r["Percentage"])
        break
for kind in ("fetch", "write", "sq"):
    p = os.path.join(src, kind, kind + "_counter_collection.csv")
    if os.path.exists(p):
        keep_library_rows(p, os.path.join(dst, "%s_pmc_%s.csv" % (prefix, kind)))
        entry["source"].append("profiles/%s_pmc_%s.csv" % (prefix, kind))
p_f = os.path.join(src, "fetch", "fetch_counter_collection.csv")
p_w = os.path.join(src, "write", "write_counter_collection.csv")
if os.path.exists(p_f) and os.path.exists(p_w):
    f, _ = per_launch(p_f, kernel_key)
    w, _ = per_launch(p_w, kernel_key)
    entry["fetch_size_kib_raw"] = f["FETCH_SIZE"]
    entry["write_size_kib_raw"] = w["WRITE_SIZE"]
    entry["hbm_read_bytes_per_launch"] = 2.0 * f["FETCH_SIZE"] * 1024.0
    entry["hbm_write_bytes_per_launch"] = w["WRITE_SIZE"] * 1024.0
    entry["hbm_bytes_per_launch"] = entry["hbm_read_bytes_per_launch"] + entry["hbm_write_bytes_per_launch"]
    entry["correction"] = "FETCH_SIZE x 2 (gfx950), WRITE_SIZE as reported; separate --pmc passes"
    cal = {}
    for cb in (16384, 131072):
        jf = os.path.join(src, "cal_fetch_%d.json" % cb)
        cf = os.path.join(src, "cal_fetch_%d" % cb, "cal_counter_collection.csv")
        cw = os.path.join(src, "cal_write_%d" % cb, "cal_counter_collection.csv")
        if not (os.path.exists(jf) and os.path.exists(cf) and os.path.exists(cw)):
            continue
        known = json.loads(open(jf).read().strip().split("\n")[-1])
        ff, _ = per_launch(cf, "k_solve")
        ww, _ = per_launch(cw, "k_solve")
        cal[str(cb)] = {"known_read_bytes": known["read_bytes"], "known_write_bytes": known["write_bytes"],
                        "first_iteration_accepted_fraction": known["first_iteration_accepted_fraction"],
                        "fetch_size_bytes_raw": ff["FETCH_SIZE"] * 1024.0,
                        "write_size_bytes_raw": ww["WRITE_SIZE"] * 1024.0,
                        "read_factor": known["read_bytes"] / (ff["FETCH_SIZE"] * 1024.0),
                        "write_factor": known["write_bytes"] / (ww["WRITE_SIZE"] * 1024.0)}
        keep_library_rows(cf, os.path.join(dst, "%s_cal%d_pmc_fetch.csv" % (prefix, cb)))
        keep_library_rows(cw, os.path.join(dst, "%s_cal%d_pmc_write.csv" % (prefix, cb)))
    if cal:
        entry["calibration_on_k_solve_launches_of_known_bytes"] = cal
p_s = os.path.join(src, "sq", "sq_counter_collection.csv")
if os.path.exists(p_s):
    s, shape = per_launch(p_s, kernel_key)
    entry["sq_counters_per_launch"] = s
    entry["launch"] = dict(zip(("grid_threads", "workgroup", "vgpr", "agpr", "scratch_bytes_per_lane", "lds_bytes"), shape))
    waves = shape[0] // 64
    entry["waves"] = waves
    if s.get("SQ_WAVE_CYCLES"):
        busy = s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"]
        entry["valu_busy_frac_per_wave"] = busy
        if workload == "cartpole":  # every wave lives for the whole launch
            occ = min(1.0, waves / float(N_SIMD))
            entry["simd_occupancy"] = occ
            entry["valu_busy_frac_of_chip"] = busy * occ
            if s.get("SQ_THREAD_CYCLES_VALU"):
                lanes = s["SQ_THREAD_CYCLES_VALU"] / (64.0 * s["SQ_ACTIVE_INST_VALU"])
                entry["active_lane_frac_of_valu_cycles"] = lanes
                entry["valu_lane_weighted_frac_of_chip"] = busy * occ * lanes
        entry["wait_frac_per_wave"] = s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
        if s.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            entry["mfma_busy_frac_of_wave_cycles"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * s["SQ_WAVE_CYCLES"])
        if s.get("SQ_INSTS_VALU_MFMA_F64"):
            entry["mfma_share_of_valu_instructions"] = s["SQ_INSTS_VALU_MFMA_F64"] / s["SQ_INSTS_VALU"]
path = os.path.join(dst, "solve_counters.json")
doc = json.load(open(path)) if os.path.exists(path) else {"entries": []}
doc["entries"] = [e for e in doc["entries"] if not (e["workload"] == workload and e["batch"] == batch)] + [entry]
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
