#!/usr/bin/env python3
"""Static VALU mix of a gfx950 kernel by issue class (classes measured with tools/valu_rate.hip:
fast ~2.5 cycles per wave64 instruction, slow ~4.3, transcendental ~8.4; an SGPR source operand or two
VGPR sources in one bank make a fast opcode slow).
usage: isa_classes.py file.s kernel_name [first_line last_line]"""
import re, sys, collections

FAST = {"v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
        "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32",
        "v_not_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}

def classify(line):
    t = line.split(None, 1)
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
    ops = t[1] if len(t) > 1 else ""
    if not op.startswith("v_"):
        return None, op
    if op in TRANS:
        return "trans", op
    if op in FAST:
        args = [a.strip() for a in ops.split(",")]
        srcs = args[1:]
        if any(re.fullmatch(r"-?\|?s\d+\|?|s\[\d+:\d+\]|vcc|exec|m0|vcc_lo|vcc_hi|exec_lo|exec_hi", a) for a in srcs):
            return "slow(sgpr)", op
        # measured: v_fmac with src0 and src1 in one bank, v_fma reading its own destination with src1 and src2 in
        # one bank are slow; other same-bank pairs are not
        regs = [int(m.group(1)) if m else None for a in srcs for m in [re.fullmatch(r"-?\|?v(\d+)\|?", a.replace(" clamp", ""))]]
        if op == "v_fmac_f32" and len(regs) >= 2 and None not in regs[:2] and regs[0] % 4 == regs[1] % 4 and regs[0] != regs[1]:
            return "slow(bank)", op
        return "fast", op
    return "slow", op

def main():
    path, kernel = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
    cls = collections.Counter()
    ops = collections.defaultdict(collections.Counter)
    other = collections.Counter()
    for l in body[lo:hi]:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        c, op = classify(l)
        if c is None:
            other["salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem"] += 1
        else:
            cls[c] += 1
            ops[c][op] += 1
    total = sum(cls.values())
    cyc = {"fast": 2.5, "slow": 4.3, "slow(sgpr)": 4.3, "slow(bank)": 4.3, "trans": 8.4}
    tc = sum(cyc[c] * n for c, n in cls.items())
    print("lines %d..%d: %d VALU, est. %.0f cycles (%.2f per instruction); other: %s" % (lo, hi, total, tc, tc / max(total, 1), dict(other)))
    for c, n in cls.most_common():
        print("  %-11s %5d (%4.1f %% of instructions, %4.1f %% of cycles)  %s" % (
            c, n, 100.0 * n / total, 100.0 * cyc[c] * n / tc, ", ".join("%s %d" % kv for kv in ops[c].most_common(8))))

main()
