#!/usr/bin/env python3
"""Prices the VALU instructions of one kernel's ISA with the issue costs measured by tools/micro/valu_issue.hip on gfx950
(cycles per wave64 instruction per SIMD at >= 4 waves per SIMD): 2.3 for the plain f32 / integer ALU ops, 8.2 for the
transcendental unit, 4.2 for everything else (min/max, compares, selects, shifts, VOP3-only forms, conversions).
Prints per basic block: VALU count, priced cycles, the most frequent mnemonics.

    hipcc ... -S --cuda-device-only -o k.s file.hip ;  python tools/isa_cost.py k.s <kernel-name-substring> [min_cycles]
    python tools/isa_cost.py --json out.json        compiles csrc/*.hip to ISA here and writes, per kernel, the VALU count and
                                                    the priced cycles per VALU instruction of its STATIC instruction mix
                                                    (profiles/isa_cost.json: what tools/summarize_sq.py prices SQ_INSTS_VALU with)"""
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


def kernel_table(path):
    """{kernel symbol: (valu, priced cycles)} for every kernel of one .s file"""
    out, cur = {}, None
    for l in open(path).read().splitlines():
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            out[cur] = [0, 0.0]
            continue
        if cur is None:
            continue
        if "s_endpgm" in l:
            cur = None
            continue
        t = l.strip().split()
        if t and t[0].startswith("v_"):
            out[cur][0] += 1
            out[cur][1] += cost(t[0])
    return out


def write_json(dst):
    import hashlib, json, os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "gdpathtracing_amd", "csrc")
    sys.path.insert(0, root)
    from tools.csrc_sha import csrc_sha
    res = {"_meta": {"csrc_sha": csrc_sha(), "prices": "2.3 plain f32/int ALU, 8.2 transcendental, 4.2 everything else (tools/micro/valu_issue.hip)",
                     "mix": "static (every instruction of the kernel's ISA counted once)"}}
    for f in ("jpt_kernels_wf2.hip", "jpt_kernels_ref.hip", "jpt_kernels_post.hip"):
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                                   "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", tmp.name, os.path.join(csrc, f)],
                                  stderr=subprocess.DEVNULL)
            for sym, (n, c) in kernel_table(tmp.name).items():
                m = re.search(r"(wf2_\w+?|ref_frame_kernel|temporal_kernel|assemble_ldr_kernel|assemble_kernel|instance_refit_kernel|tlas4_refit_kernel|quantize_tail_kernel)(ILb([01])|E|P|I)", sym)
                if not m or n == 0:
                    continue
                name = m.group(1)
                counted = "ILb1" in sym[sym.find(name):sym.find(name) + len(name) + 5]
                if counted:
                    continue                       # the counting builds are not what the profiles run
                w4 = "ELb1" in sym or "ILb0ELb1" in sym
                key = name
                if name in ("wf2_primary", "wf2_trace", "wf2_finish") and not w4:
                    key = name + "_w2"           # two-child records (reference-exact trees)
                res[key] = {"valu": n, "cycles_per_valu": round(c / n, 3), "symbol": sym}
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v.get("cycles_per_valu") for k, v in res.items() if k != "_meta"}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--json":
        write_json(sys.argv[2])
    else:
        main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 40.0)
