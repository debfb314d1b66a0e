#!/usr/bin/env python3
"""Bank-conflict check for the LDS exchange patterns of ntt14.hpp (MI355X rules, /opt/skills/guides/MI355X_MICROARCH.md):
ds_write_b64: 4 groups of 16 lanes, bank = (addr/4) mod 32;  ds_read_b64: 2 groups of 32 lanes, bank = (addr/4) mod 64;
ds_read_b128 / ds_write_b128: see the guide's lane groups.  Prints the worst multiplicity per pattern (1 = conflict free)."""


def conflicts(slots, kind):
    """slots: list of 64 8-byte-slot indices (one per lane) for one wave instruction"""
    if kind == "write64":
        groups, banks = [range(16 * g, 16 * g + 16) for g in range(4)], 32
    elif kind == "read64":
        groups, banks = [range(32 * g, 32 * g + 32) for g in range(2)], 64
    else:
        raise ValueError(kind)
    worst = 1
    for g in groups:
        per_bank = {}
        for lane in g:
            for dw in range(2):
                addr = slots[lane] * 8 + dw * 4
                per_bank.setdefault((addr // 4) % banks, set()).add(addr)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


def bits(x, *idx):
    """pack the given bit positions of x (MSB first)"""
    r = 0
    for i in idx:
        r = (r << 1) | ((x >> i) & 1)
    return r


def check(name, kind, fn, regs):
    worst = max(conflicts([fn(lane, r) for lane in range(64)], kind) for r in regs)
    print("%-52s %-8s worst multiplicity %d" % (name, kind, worst))
    return worst


if __name__ == "__main__":
    # X01 (cross-wave): natural order, lane = low 6 bits -> trivially conflict free
    check("X01 write / read (lane = slot low bits)", "write64", lambda lane, r: r * 64 + lane, range(16))
    check("X01 write / read (lane = slot low bits)", "read64", lambda lane, r: r * 64 + lane, range(16))

    # X12 (wave-local): half image index j = (b10 b9 b8 b7) << 6 | (b5..b0); phys = j + 4 * (j >> 6)
    p12 = lambda j: j + 4 * (j >> 6)
    check("X12 write: reg = b10..b7, lane = b5..b0", "write64", lambda lane, r: p12((r << 6) | lane), range(16))
    # reader: lane = (b10 b9 b8 b7 b1 b0), reg r4 = (b5 b4 b3 b2)
    check("X12 read: reg = b5..b2, lane = (b10..b7, b1 b0)", "read64", lambda lane, r: p12(((lane >> 2) << 6) | (r << 2) | (lane & 3)), range(16))

    # X23 (wave-local): half image (bit 2 removed): j = (b10 b9) << 8 | (b1 b0) << 6 | (b8 b7 b6 b5 b4 b3); phys = j + (j >> 4)
    p23 = lambda j: j + (j >> 4)
    # writer: lane = (b10 b9 b8 b7 b1 b0), reg = (b6 b5 b4 b3)
    def w23(lane, r):
        b10_9, b8_7, b1_0 = lane >> 4, (lane >> 2) & 3, lane & 3
        return p23((b10_9 << 8) | (b1_0 << 6) | (b8_7 << 4) | r)
    check("X23 write: reg = b6..b3, lane = (b10..b7, b1 b0)", "write64", w23, range(16))
    # reader: lane = (b8..b3), reg = (b10 b9 b1 b0)
    check("X23 read: reg = (b10 b9 b1 b0), lane = b8..b3", "read64", lambda lane, r: p23((r << 6) | lane), range(16))
    for name, pad in (("pad 2 per 32", lambda j: j + 2 * (j >> 5)), ("pad 1 per 16 + 1 per 256", lambda j: j + (j >> 4) + (j >> 8)),
                      ("xor swizzle", lambda j: j ^ ((j >> 4) & 15))):
        def w(lane, r, pad=pad):
            b10_9, b8_7, b1_0 = lane >> 4, (lane >> 2) & 3, lane & 3
            return pad((b10_9 << 8) | (b1_0 << 6) | (b8_7 << 4) | r)
        check("X23 write, " + name, "write64", w, range(16))
        check("X23 read, " + name, "read64", lambda lane, r, pad=pad: pad((r << 6) | lane), range(16))
    # the layouts ntt14w.hpp ships
    print("-- ntt14w.hpp --")
    check("w14 X12 write (68 n4 + lane)", "write64", lambda lane, n4: 68 * n4 + lane, range(16))
    check("w14 X12 read (68 (lane>>2) + (lane&3) + 4 r4)", "read64", lambda lane, r4: 68 * (lane >> 2) + (lane & 3) + 4 * r4, range(16))
    def w14_wr23(lane, c):
        hi5 = ((lane >> 4) << 3) | ((lane & 3) << 1) | ((lane >> 3) & 1)
        return 34 * hi5 + ((lane >> 2) & 1) + 2 * c
    check("w14 X23 write", "write64", w14_wr23, range(16))
    check("w14 X23 read", "read64", lambda lane, R: 34 * (lane >> 5) + 2 * (lane & 15) + ((lane >> 4) & 1) + 68 * R, range(16))
    # the same four patterns in the other direction (inverse transform): reads <-> writes
    check("w14 X21 write (pass-2 side)", "write64", lambda lane, r4: 68 * (lane >> 2) + (lane & 3) + 4 * r4, range(16))
    check("w14 X21 read (pass-1 side)", "read64", lambda lane, n4: 68 * n4 + lane, range(16))
    check("w14 X32 write (pass-3 side)", "write64", lambda lane, R: 34 * (lane >> 5) + 2 * (lane & 15) + ((lane >> 4) & 1) + 68 * R, range(16))
    check("w14 X32 read (pass-2 side)", "read64", w14_wr23, range(16))
    # injectivity of the two layouts
    s12 = {68 * n4 + l for n4 in range(16) for l in range(64)}
    s23 = {w14_wr23(l, c) for l in range(64) for c in range(16)}
    print("slots used: X12 %d (max %d), X23 %d (max %d)" % (len(s12), max(s12), len(s23), max(s23)))
    check("w14 X32 write, inverse layout 17 L + c", "write64", lambda lane, R: 68 * (lane >> 4) + (lane & 15) + 17 * (((R >> 2) << 4) | (R & 3)), range(16))
    check("w14 X32 read, inverse layout 17 L + c", "read64", lambda lane, c: 17 * lane + c, range(16))
    s32 = {68 * (l >> 4) + (l & 15) + 17 * (((R >> 2) << 4) | (R & 3)) for l in range(64) for R in range(16)}
    print("slots used: X32 %d (max %d), same set as the reader's: %s" % (len(s32), max(s32), s32 == {17 * l + c for l in range(64) for c in range(16)}))


def conflicts128(slot16, kind):
    """slot16: per-lane 8-byte slot index of a 16-byte access"""
    if kind == "write128":
        groups, banks = [range(8 * g, 8 * g + 8) for g in range(8)], 32
    else:  # read128: 4 groups of 16 lanes, 64 banks
        groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
                  list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
        banks = 64
    worst = 1
    for g in groups:
        per_bank = {}
        for lane in g:
            for dw in range(4):
                addr = slot16[lane] * 8 + dw * 4
                per_bank.setdefault((addr // 4) % banks, set()).add(addr)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


def check128(name, kind, fn, regs):
    worst = max(conflicts128([fn(lane, r) for lane in range(64)], kind) for r in regs)
    print("%-52s %-8s worst multiplicity %d" % (name, kind, worst))


if __name__ == "__main__":
    print("-- store staging (forward): one (i10 i9) value = 512 contiguous coefficients per round --")
    for name, pad in (("no pad", lambda e: e), ("2 per 8", lambda e: e + 2 * (e >> 3)), ("2 per 16", lambda e: e + 2 * (e >> 4)),
                      ("2 per 32", lambda e: e + 2 * (e >> 5)), ("4 per 32", lambda e: e + 4 * (e >> 5)), ("2 per 8 + 2 per 128", lambda e: e + 2 * (e >> 3) + 2 * (e >> 7))):
        # writer: lane l holds elements 8 l + 2 j .. (16-byte piece j = 0..3)
        check128("stage write b128, " + name, "write128", lambda lane, j, pad=pad: pad(8 * lane + 2 * j), range(4))
        # reader: lane l reads the pair at 128 k + 2 l
        check128("stage read b128, " + name, "read128", lambda lane, k, pad=pad: pad(128 * k + 2 * lane), range(4))
        check("stage write b64, " + name, "write64", lambda lane, j, pad=pad: pad(8 * lane + j), range(8))
