#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof_*) into the small tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv      -- per (kernel, launch shape) durations from the kernel trace of the default bench command
  profiles/<tag>_pmc.json              -- per (kernel, launch shape) counter averages, raw and corrected
  profiles/pmc_summary.json            -- what bench.py reports as roofline.traffic / roofline.issue (the batch-4096 launches alone)
  profiles/<tag>_secondary_pmc.json    -- the kernels of the secondary workloads
Every row is keyed by (kernel name, workgroups of the launch): one kernel runs at several launch sizes inside bench.py (4096
polynomials in the timed steps, 2048 in the ring-product block, 64 in the verification), and an average over unequal launches is
the average of nothing (round 2's summaries did that: a 'write bytes' figure of (12 x 512 MiB + 8 MiB) / 13).
Correction (MI355X_MICROARCH.md, HBM section; re-calibrated here with 8- and 16-byte-per-lane copy kernels of a known 512 MiB):
FETCH_SIZE reads exactly 1/2 of the streamed bytes on gfx950 -> x2; WRITE_SIZE is exact; unit KiB.
usage: tools/summarize_profiles.py <tag> [gpurun_out]"""
import collections
import csv
import glob
import json
import math
import os
import re
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
os.makedirs("profiles", exist_ok=True)


def newest(pattern_dir, suffix):
    """gpurun merges new files into gpurun_out/ without removing older runs': per directory only the most recent file counts"""
    fs = glob.glob(os.path.join(pattern_dir, "**", "*" + suffix), recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("fhe::", "")[:90]


# ---- durations: our own statistics over the kernel TRACE, one row per (kernel, workgroups per launch) ----
for f in [x for d in glob.glob(os.path.join(src, "prof_stats*")) if os.path.isdir(d) for x in newest(d, "kernel_trace.csv")]:
    sub = os.path.relpath(f, src).split(os.sep)[0].replace("prof_stats", "")
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        groups[(short(r["Kernel_Name"]), grid // max(wg, 1), wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = float(sum(sum(v) for v in groups.values())) or 1.0
    out = os.path.join("profiles", "%s%s_kernel_stats.csv" % (tag, sub))
    with open(out, "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Workgroups", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for (name, wgs, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(v) / len(v)
            sd = math.sqrt(sum((x - avg) ** 2 for x in v) / len(v))
            w.writerow([name, wgs, wg, len(v), sum(v), "%.1f" % avg, "%.3f" % (100.0 * sum(v) / total), min(v), max(v), "%.1f" % sd])
    print("wrote", out)

# ---- counters: per (directory, kernel, workgroups per launch) ----
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(src, "*")):
    for f in (newest(d, "counter_collection.csv") if os.path.isdir(d) else []):
        for r in csv.DictReader(open(f)):
            wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            key = "%s:%s:wg%d" % (os.path.basename(d), short(r["Kernel_Name"]), wgs)
            pmc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":  # the dispatch's duration under the profiler, for the effective clock
                pmc[key]["DURATION_NS"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
summary = {}
for k, cs in sorted(pmc.items()):
    e = {}
    for c, v in cs.items():
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            e[c + "_raw_avg_KiB"] = sum(v) / len(v)
            e[c + "_launches"] = len(v)
            e[c + "_corrected_bytes"] = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)
        else:  # SQ / GRBM counters of the stall analysis: per-launch average
            e[c + "_avg"] = sum(v) / len(v)
            e[c + "_launches"] = len(v)
    summary[k] = e
json.dump(summary, open(os.path.join("profiles", tag + "_pmc.json"), "w"), indent=1)
print("wrote profiles/%s_pmc.json" % tag)


def find(parts, field):
    for k, e in summary.items():
        if all(part in k for part in parts) and field in e:
            return e[field]
    return None


def headline(kind):
    """the batch-4096 launches (4096 workgroups) of the 2^14 forward / inverse kernel in the headline counter passes"""
    parts = ("prof_pmc", "ntt14w_%s_kernel<" % kind, "ArithDS<60>, false", ":wg4096")
    rd, wr = find(parts, "FETCH_SIZE_corrected_bytes"), find(parts, "WRITE_SIZE_corrected_bytes")
    if not (rd and wr):
        return None
    insts, waves = find(parts, "SQ_INSTS_VALU_avg"), find(parts, "SQ_WAVES_avg")
    cyc, wait, stall, act = (find(parts, c + "_avg") for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
    gui, dur = find(parts, "GRBM_GUI_ACTIVE_avg"), find(parts, "DURATION_NS_avg")
    return {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
            # MI355X_MICROARCH.md, DVFS give-back: effective clock = GRBM_GUI_ACTIVE / 8 XCDs / wall time (reads a little high below 0.3 ms)
            "effective_clock_mhz": (gui / 8.0 / dur * 1e3) if gui and dur else None, "profiled_launch_ms": dur / 1e6 if dur else None,
            "launches_averaged": find(parts, "WRITE_SIZE_launches"),
            "valu_insts_per_wave": (insts / waves) if insts and waves else None,
            "wave_cycle_split": {"parked (s_waitcnt / barrier)": wait / cyc, "issue stalled": stall / cyc, "issuing": act / cyc} if cyc and wait and stall and act else None}


fwd, inv = headline("fwd"), headline("inv")
if fwd:
    calib = {k: e for k, e in summary.items() if "prof_calib" in k}
    out = {"source": "profiles/%s_pmc.json" % tag, "launch": "4096 polynomials = 4096 workgroups (the timed steps' launches only)",
           "kernel": "ntt14w_fwd_kernel<ArithDS<60>, false, 3> (forward)",
           "ntt_fwd_read_bytes_per_launch": fwd["read_bytes_per_launch"], "ntt_fwd_write_bytes_per_launch": fwd["write_bytes_per_launch"],
           "ntt_fwd_bytes_per_launch": fwd["bytes_per_launch"], "ntt_fwd_launches_averaged": fwd["launches_averaged"],
           "ntt_fwd_valu_insts_per_wave": fwd["valu_insts_per_wave"], "ntt_fwd_wave_cycle_split": fwd["wave_cycle_split"],
           "ntt_fwd_effective_clock_mhz": fwd["effective_clock_mhz"], "ntt_fwd_profiled_launch_ms": fwd["profiled_launch_ms"],
           "calibration": {k.split(":")[1]: {c: v for c, v in e.items() if c.endswith("_corrected_bytes")} for k, e in calib.items()},
           "correction": "FETCH_SIZE x2 (gfx950 reads 1/2, calibrated on copy8/copy16 of 512 MiB), WRITE_SIZE x1, KiB"}
    if inv:
        out.update({"inv_kernel": "ntt14w_inv_kernel<ArithDS<60>, false, false, 3> (inverse)",
                    "ntt_inv_read_bytes_per_launch": inv["read_bytes_per_launch"], "ntt_inv_write_bytes_per_launch": inv["write_bytes_per_launch"],
                    "ntt_inv_bytes_per_launch": inv["bytes_per_launch"], "ntt_inv_valu_insts_per_wave": inv["valu_insts_per_wave"],
                    "ntt_inv_wave_cycle_split": inv["wave_cycle_split"], "ntt_inv_effective_clock_mhz": inv["effective_clock_mhz"],
                    "ntt_inv_profiled_launch_ms": inv["profiled_launch_ms"]})
    json.dump(out, open(os.path.join("profiles", "pmc_summary.json"), "w"), indent=1)
    print("wrote profiles/pmc_summary.json", fwd["bytes_per_launch"], inv["bytes_per_launch"] if inv else None)

# ---- the secondary workloads' kernels (bench.py blocks ntt_mul / fhew / ckks / tfhe): HBM bytes per launch, wave-cycle split ----
SECONDARY = ("blind_rotate_kernel", "torus30_blind_rotate_kernel", "torusf_blind_rotate_kernel", "torusx3_blind_rotate_kernel", "external_product_kernel", "gadget_product_kernel", "ntt14w_fwd_kernel", "ntt14w_inv_kernel",
             "ntt14w_mul_kernel", "ntt_big_fwd_pass", "ntt_big_inv_pass", "rns_rescale", "rns_extend", "tlwe_key_switch", "lwe_key_switch")
sec = {}
for k, e in summary.items():
    if not k.startswith("prof_sec"):
        continue
    name = k.split(":", 1)[1]
    if not any(n in name for n in SECONDARY):
        continue
    d = sec.setdefault(name, {})
    for c, v in e.items():
        if c.endswith("_corrected_bytes") or c.endswith("_avg") or c.endswith("_launches"):
            d[c] = v
for name, d in sec.items():
    if d.get("SQ_WAVE_CYCLES_avg"):
        d["wave_cycle_split"] = {"parked": d.get("SQ_WAIT_ANY_avg", 0) / d["SQ_WAVE_CYCLES_avg"], "issue_stalled": d.get("SQ_WAIT_INST_ANY_avg", 0) / d["SQ_WAVE_CYCLES_avg"],
                                 "issuing": d.get("SQ_ACTIVE_INST_ANY_avg", 0) / d["SQ_WAVE_CYCLES_avg"]}
    if "SQ_INSTS_VALU_avg" in d and d.get("SQ_WAVES_avg"):
        d["valu_insts_per_wave"] = d["SQ_INSTS_VALU_avg"] / d["SQ_WAVES_avg"]
    if "FETCH_SIZE_corrected_bytes" in d and "WRITE_SIZE_corrected_bytes" in d:
        d["hbm_bytes_per_launch"] = d["FETCH_SIZE_corrected_bytes"] + d["WRITE_SIZE_corrected_bytes"]
if sec:
    json.dump(sec, open(os.path.join("profiles", tag + "_secondary_pmc.json"), "w"), indent=1)
    print("wrote profiles/%s_secondary_pmc.json (%d kernel x launch-shape rows)" % (tag, len(sec)))
