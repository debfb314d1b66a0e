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


FWD = ("prof_pmc", "ntt14w_fwd_kernel<", "ArithDS<60>, false")
fw_r, fw_w = find(FWD, "FETCH_SIZE"), find(FWD, "WRITE_SIZE")
if fw_r and fw_w:
    insts, waves = find_avg(FWD, "SQ_INSTS_VALU"), find_avg(FWD, "SQ_WAVES")
    cyc, wait, stall, act = (find_avg(FWD, c) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
    calib = {k: e for k, e in summary.items() if "prof_calib" in k}
    json.dump({"source": "profiles/%s_pmc.json" % tag, "kernel": "ntt14w_fwd_kernel<ArithDS<60>, false> (forward, batch 4096)",
               "ntt_fwd_read_bytes_per_launch": fw_r, "ntt_fwd_write_bytes_per_launch": fw_w,
               "ntt_fwd_bytes_per_launch": fw_r + fw_w,
               "ntt_fwd_valu_insts_per_wave": (insts / waves) if insts and waves else None,
               "ntt_fwd_wave_cycle_split": {"parked (s_waitcnt / barrier)": wait / cyc, "issue stalled": stall / cyc, "issuing": act / cyc} if cyc and wait and stall and act else None,
               "calibration": {k.split(":")[1]: {c: v for c, v in e.items() if c.endswith("_corrected_bytes")} for k, e in calib.items()},
               "correction": "FETCH_SIZE x2 (gfx950 reads 1/2, calibrated on copy8/copy16 of 512 MiB), WRITE_SIZE x1, KiB"},
              open(os.path.join("profiles", "pmc_summary.json"), "w"), indent=1)
    print("wrote profiles/pmc_summary.json", fw_r + fw_w)

# per-kernel counters of the secondary workloads (bench.py blocks ntt_mul / fhew / ckks / tfhe): HBM bytes per launch and the
# wave-cycle split, for the kernels that dominate them
SECONDARY = ("blind_rotate_kernel", "torus30_blind_rotate_kernel", "external_product_kernel", "ntt14w_fwd_kernel", "ntt14w_inv_kernel", "ntt_big_fwd_pass",
             "ntt_big_inv_pass", "rns_rescale_kernel", "rns_extend_kernel", "rns_rescale_edge_kernel", "rns_extend_edge_kernel", "tlwe_key_switch", "lwe_key_switch")
sec = {}
for k, e in summary.items():
    if not k.startswith("prof_sec"):
        continue
    name = k.split(":", 1)[1]
    if not any(n in name for n in SECONDARY):
        continue
    d = sec.setdefault(name, {})
    for c, v in e.items():
        if c.endswith("_corrected_bytes") or c.endswith("_avg"):
            d[c] = v
for name, d in sec.items():
    if "SQ_WAVE_CYCLES_avg" in d and d["SQ_WAVE_CYCLES_avg"]:
        d["wave_cycle_split"] = {"parked": d.get("SQ_WAIT_ANY_avg", 0) / d["SQ_WAVE_CYCLES_avg"], "issue_stalled": d.get("SQ_WAIT_INST_ANY_avg", 0) / d["SQ_WAVE_CYCLES_avg"],
                                 "issuing": d.get("SQ_ACTIVE_INST_ANY_avg", 0) / d["SQ_WAVE_CYCLES_avg"]}
    if "SQ_INSTS_VALU_avg" in d and d.get("SQ_WAVES_avg"):
        d["valu_insts_per_wave"] = d["SQ_INSTS_VALU_avg"] / d["SQ_WAVES_avg"]
    if "FETCH_SIZE_corrected_bytes" in d and "WRITE_SIZE_corrected_bytes" in d:
        d["hbm_bytes_per_launch"] = d["FETCH_SIZE_corrected_bytes"] + d["WRITE_SIZE_corrected_bytes"]
if sec:
    json.dump(sec, open(os.path.join("profiles", tag + "_secondary_pmc.json"), "w"), indent=1)
    print("wrote profiles/%s_secondary_pmc.json (%d kernels)" % (tag, len(sec)))
