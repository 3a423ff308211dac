#!/usr/bin/env python3
"""LDS bank-conflict model for the MFMA A-operand reads (ds_read_b128) of the band kernels.

A wave64 ds_read_b128 is served in four 16-lane groups (lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}
and the same +32); a 16-byte aligned read covers one of the 16 four-bank slots of the 64-bank row,
slot = (float offset / 4) mod 16; a group costs as many LDS cycles as its busiest slot holds
distinct addresses. Lane (q = lane >> 4, p = lane & 15) of an operand read addresses
pixel_offset(p) * CPP + 4 * q floats, where CPP is the padded per-pixel stride of the tile.
Prints cycles per read (4 = conflict free) for each kernel geometry and candidate padding.
"""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def cycles(cpp, tw, lw, s=1):
    total = 0
    for g in GROUPS:
        slots = {}
        for lane in g:
            q, p = lane >> 4, lane & 15
            off = (((p // tw) * s) * lw + (p % tw) * s) * cpp + 4 * q
            slots.setdefault((off // 4) % 16, set()).add(off)
        total += max(len(v) for v in slots.values())
    return total


if __name__ == "__main__":
    geoms = {"16ch 32x32 s1": (16, 32, 34, 1), "32ch 16x16 s1": (32, 16, 18, 1), "64ch 8x8 s1": (64, 8, 10, 1),
             "16->32 s2": (16, 16, 33, 2), "32->64 s2": (32, 8, 17, 2)}
    for name, (cp, tw, lw, s) in geoms.items():
        print(name, {pad: cycles(cp + pad, tw, lw, s) for pad in (0, 4, 8, 12, 16, 20, 24)})
