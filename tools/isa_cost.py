#!/usr/bin/env python3
"""Prices the VALU instructions of one kernel's ISA with the issue costs measured by tools/micro/valu_issue.hip on gfx950
(cycles per wave64 instruction per SIMD at >= 4 waves per SIMD): 2.3 for the plain f32 / integer ALU ops, 8.2 for the
transcendental unit, 4.2 for everything else (min/max, compares, selects, shifts, VOP3-only forms, conversions).
Prints per basic block: VALU count, priced cycles, the most frequent mnemonics.

    hipcc ... -S --cuda-device-only -o k.s file.hip ;  python tools/isa_cost.py k.s <kernel-name-substring> [min_cycles]"""
import collections
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_mac_f32", "v_mad_f32"}
SLOW = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def cost(m):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
    return 2.3 if base in FAST else (8.2 if base in SLOW else 4.2)


def main(path, kernel, min_cycles=40.0):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kernel, l))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        if "s_endpgm" in l:
            break
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        t = l.strip().split()
        if t and (t[0].startswith("v_") or t[0].startswith("s_") or t[0].startswith("ds_") or t[0].startswith("global_") or t[0].startswith("scratch_")):
            blocks[cur].append(t[0])
    tot_v = tot_c = 0
    for name, ins in blocks.items():
        v = [i for i in ins if i.startswith("v_")]
        c = sum(cost(i) for i in v)
        tot_v += len(v)
        tot_c += c
        if c >= min_cycles:
            top = collections.Counter(re.sub(r"_e(32|64)$", "", i) for i in v).most_common(6)
            print("%-12s VALU %4d  cycles %7.1f  SALU %3d  mem %2d   %s" % (name, len(v), c, sum(i.startswith("s_") for i in ins),
                  sum(i.startswith(("ds_", "global_", "scratch_")) for i in ins), " ".join("%s:%d" % kv for kv in top)))
    print("total VALU %d, priced cycles %.0f, average %.2f cycles per instruction" % (tot_v, tot_c, tot_c / max(tot_v, 1)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 40.0)
