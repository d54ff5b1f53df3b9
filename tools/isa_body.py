#!/usr/bin/env python3
"""tools/isa_body.py FILE.s KERNEL_SUBSTR -- classify the VALU instructions of a kernel's hottest basic block.

The LDS body of k_yuv_tile2 is one straight-line block (the one with the most tap blends).  Classes follow DESIGN.md 5.1:
slow = the quarter-rate pipe (conversions, min/max/med3, trunc, cndmask, shifts-left, packed, v_fma_mix, SDWA, compares,
any fp op with an SGPR operand), fast = everything else.  Prints counts per pixel (px per lane per block given as arg 3).
"""
import collections
import re
import sys

SLOW_PREFIX = ("v_cvt", "v_min", "v_max", "v_med3", "v_trunc", "v_floor", "v_fract", "v_rndne", "v_cndmask", "v_lshlrev", "v_bfe",
               "v_lshl_add", "v_lshl_or", "v_add3", "v_mad_u32", "v_mul_u32", "v_perm", "v_pk_", "v_fma_mix", "v_dot2", "v_cmp",
               "v_readlane", "v_writelane", "v_readfirstlane", "v_mul_lo", "v_mul_hi", "v_ashrrev", "v_and_or", "v_or3", "v_bfi", "v_alignbit")
MEDIUM_PREFIX = ("v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_sub_u32", "v_add_u32", "v_subrev_u32", "v_add_co", "v_sub_co")


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return None
    if "sdwa" in op or op.startswith(SLOW_PREFIX):
        return "slow"
    ops = line[len(op):]
    has_s = re.search(r"(?<![a-z_])s\d+|s\[\d+:\d+\]|vcc|exec", ops) is not None
    if op.startswith(MEDIUM_PREFIX):
        return "medium_s" if has_s else "medium"
    return "slow_sgpr" if has_s else "fast"


def main():
    path, key = sys.argv[1], sys.argv[2]
    px = float(sys.argv[3]) if len(sys.argv) > 3 else 16.0
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    blocks, cur = [], []
    for l in text[start + 1:end]:
        t = l.strip()
        t = t.split(";")[0].strip()
        if not t or t.startswith("//") or (t.startswith(".") and not t.endswith(":")):
            continue
        if t.endswith(":") or t.split()[0].startswith(("s_cbranch", "s_branch")):
            if cur:
                blocks.append(cur)
            cur = []
            continue
        cur.append(t.split(";")[0].strip())
    if cur:
        blocks.append(cur)
    hot = max(blocks, key=lambda b: sum(1 for i in b if i.startswith("v_fma_mix") or i.startswith("v_fma_f32") or i.startswith("v_mul_f32")))
    cls = collections.Counter()
    byop = collections.defaultdict(collections.Counter)
    for i in hot:
        c = classify(i)
        if c:
            cls[c] += 1
            byop[c][i.split()[0]] += 1
    lds = sum(1 for i in hot if i.startswith("ds_"))
    waits = sum(1 for i in hot if i.startswith("s_waitcnt"))
    n = sum(cls.values())
    print(f"hot block: {len(hot)} instructions, {n} VALU ({n / px:.1f}/px), {lds} LDS ({lds / px:.2f}/px), {waits} s_waitcnt")
    for c in ("fast", "medium", "medium_s", "slow_sgpr", "slow"):
        print(f"  {c:10s} {cls[c]:5d}  {cls[c] / px:6.2f}/px   " + " ".join(f"{k}:{v}" for k, v in byop[c].most_common(12)))


if __name__ == "__main__":
    main()
