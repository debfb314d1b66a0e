#!/usr/bin/env python3
"""Instruction-class histogram of one kernel in a hipcc -S listing (static counts of the straight-line code).
usage: isa_hist.py listing.s [--json out.json] 'kernel-name-substring' [more substrings...]
--json also writes, per kernel, the VALU count and the stream PRICED with the issue costs of DESIGN.md section 4 (SIMD cycles per
wave-instruction with every CU busy, tools/microbench_issue.hip): what bench.py reports as roofline.issue."""
import collections
import re
import sys


def kernels(path):
    cur, body, out = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\S+):\s", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(line)
            if "s_endpgm" in line:
                out[cur] = body
                cur = None
    meta = {}
    txt = open(path).read()
    for k in out:
        m = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(k), txt)
        s = re.search(r"\.set %s\.private_seg_size, (\d+)" % re.escape(k), txt)
        meta[k] = (int(m.group(1)) if m else -1, int(s.group(1)) if s else -1)
    return out, meta


def classify(op):
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"): return "mad64"
    if op.startswith("v_mul_lo") or op.startswith("v_mul_hi"): return "mul32"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "mov"
    if op.startswith("v_lshl_add_u64"): return "add64"
    if op.startswith(("v_add_co", "v_addc", "v_sub_co", "v_subb", "v_subrev_co", "v_subbrev")): return "add/sub carry"
    if op.startswith(("v_and", "v_or", "v_xor", "v_bfe", "v_bfi", "v_lshr", "v_lshl", "v_ashr", "v_alignbit", "v_bitop", "v_perm", "v_not")): return "bit/shift"
    if op.startswith(("v_cndmask", "v_cmp", "v_min", "v_max")): return "select/compare"
    if op.startswith("v_"): return "other valu"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_"): return "salu/smem"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch (spill)"
    return "other"


# SIMD cycles per wave-instruction, chip full (measured with tools/microbench_issue.hip)
COST = {"mad64": 4.4, "add64": 4.1, "mul32": 4.1, "mov": 2.4, "add/sub carry": 2.4, "bit/shift": 2.4, "select/compare": 2.4, "other valu": 2.4, "s_nop": 0.7}
WIDE_SHIFT = ("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64")  # 64-bit shifts cost what v_lshl_add_u64 costs


def main():
    import json
    args = sys.argv[1:]
    jout = None
    if "--json" in args:
        i = args.index("--json")
        jout = args[i + 1]
        del args[i:i + 2]
    ks, meta = kernels(args[0])
    report = {}
    for pat in args[1:]:
        for name, body in ks.items():
            if pat not in name:
                continue
            ops = [l.split()[0] for l in body if re.match(r"^\s+[a-z]+_", l)]
            cls = collections.Counter(classify(o) for o in ops)
            valu = sum(v for k, v in cls.items() if k in ("mad64", "mul32", "mov", "add64", "add/sub carry", "bit/shift", "select/compare", "other valu"))
            print("== %s" % name)
            print("   vgprs %d  scratch bytes %d  VALU %d  total %d" % (meta[name][0], meta[name][1], valu, len(ops)))
            for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
                print("   %-18s %5d" % (k, v))
            top = collections.Counter(ops).most_common(14)
            print("   top opcodes: " + ", ".join("%s %d" % t for t in top))
            wide = sum(1 for o in ops if o.startswith(WIDE_SHIFT))
            priced = sum(COST.get(k, 0.0) * v for k, v in cls.items()) + wide * (COST["add64"] - COST["bit/shift"])
            print("   priced VALU stream: %.0f SIMD cycles per wave (costs of DESIGN.md section 4)" % priced)
            report[name] = {"vgprs": meta[name][0], "scratch_bytes": meta[name][1], "valu_insts": valu, "classes": dict(cls),
                            "priced_simd_cycles_per_wave": priced, "cost_table": COST}
    if jout:
        json.dump(report, open(jout, "w"), indent=1)


if __name__ == "__main__":
    main()
