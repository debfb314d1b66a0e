#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof_*) into the small tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv   -- `rocprofv3 --kernel-trace --stats` summary (kernel names shortened)
  profiles/<tag>_pmc.json           -- per-kernel FETCH_SIZE / WRITE_SIZE averages, raw and corrected
  profiles/pmc_summary.json         -- what bench.py reports as roofline.traffic
Correction (MI355X_MICROARCH.md, HBM section; re-calibrated here with 8- and 16-byte-per-lane copy kernels of a
known 512 MiB): FETCH_SIZE reads exactly 1/2 of the streamed bytes on gfx950 -> x2; WRITE_SIZE is exact; unit KiB.
usage: tools/summarize_profiles.py <tag> [gpurun_out]"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
os.makedirs("profiles", exist_ok=True)


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")[:80]


for f in glob.glob(os.path.join(src, "prof_stats*", "**", "*kernel_stats.csv"), recursive=True):
    sub = f.split(os.sep)[1].replace("prof_stats", "")
    rows = list(csv.DictReader(open(f)))
    out = os.path.join("profiles", "%s%s_kernel_stats.csv" % (tag, sub))
    with open(out, "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])
    print("wrote", out)

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(src, "*")):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[os.path.basename(d) + ":" + short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in sorted(pmc.items()):
    e = {}
    for c, v in cs.items():
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            e[c + "_raw_avg_KiB"] = sum(v) / len(v)
            e[c + "_launches"] = len(v)
            e[c + "_corrected_bytes"] = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)
        else:  # SQ / GRBM counters of the stall analysis (tools/scripts/pmc_lab.sh, pmc_fhew.sh): per-launch average
            e[c + "_avg"] = sum(v) / len(v)
            e[c + "_launches"] = len(v)
    summary[k] = e
json.dump(summary, open(os.path.join("profiles", tag + "_pmc.json"), "w"), indent=1)
print("wrote profiles/%s_pmc.json" % tag)


def find(kern, counter):
    for k, e in summary.items():
        if all(part in k for part in kern) and (counter + "_corrected_bytes") in e:
            return e[counter + "_corrected_bytes"]
    return None


def find_avg(kern, counter):
    for k, e in summary.items():
        if all(part in k for part in kern) and (counter + "_avg") in e:
            return e[counter + "_avg"]
    return None


fw_r, fw_w = find(("ntt14_fwd_kernel<", "ArithPM<60>"), "FETCH_SIZE"), find(("ntt14_fwd_kernel<", "ArithPM<60>"), "WRITE_SIZE")
if fw_r and fw_w:
    insts, waves = find_avg(("ntt14_fwd_kernel<", "ArithPM<60>, false"), "SQ_INSTS_VALU"), find_avg(("ntt14_fwd_kernel<", "ArithPM<60>, false"), "SQ_WAVES")
    json.dump({"source": "profiles/%s_pmc.json" % tag, "kernel": "ntt14_fwd_kernel<ArithPM<60>> (forward, batch 4096)",
               "ntt_fwd_read_bytes_per_launch": fw_r, "ntt_fwd_write_bytes_per_launch": fw_w,
               "ntt_fwd_bytes_per_launch": fw_r + fw_w,
               "ntt_fwd_valu_insts_per_wave": (insts / waves) if insts and waves else None,
               "correction": "FETCH_SIZE x2 (gfx950 reads 1/2, calibrated on copy8/copy16 of 512 MiB), WRITE_SIZE x1, KiB"},
              open(os.path.join("profiles", "pmc_summary.json"), "w"), indent=1)
    print("wrote profiles/pmc_summary.json", fw_r + fw_w)
