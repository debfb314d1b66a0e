#!/usr/bin/env python3
"""Instruction-class histogram of one kernel in a hipcc -S listing (static counts of the straight-line code).
usage: isa_hist.py listing.s 'kernel-name-substring' [more substrings...]"""
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


def main():
    ks, meta = kernels(sys.argv[1])
    for pat in sys.argv[2:]:
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


if __name__ == "__main__":
    main()
