#!/usr/bin/env python3
"""Static opcode histogram of one kernel from the gfx950 assembly of the library
(hipcc ... --cuda-device-only -S context.hip -o context.s with the Makefile's flags).

    python tools/isa_histogram.py context.s 'k_trace_f64ILb1ELb1ELi1E' [--top 40] [--json]

Instructions are grouped by the class the issue-cycle table of tools/microbench/valu_issue.hip
measures (f32 / f64 arithmetic, transcendentals, 32-bit multiplies, 64-bit shifts, moves and
selects, compares, conversions, cross-lane), then SALU, LDS, vector and scalar memory, branches."""
import collections
import json
import re
import sys

CLASSES = [
    ("valu_f64_fma", r"v_fma_f64|v_fmac_f64"),
    ("valu_f64_mul", r"v_mul_f64"),
    ("valu_f64_add", r"v_add_f64|v_min_f64|v_max_f64"),
    ("valu_f64_trans", r"v_rcp_f64|v_rsq_f64|v_sqrt_f64"),
    ("valu_f64_other", r"v_div_scale_f64|v_div_fmas_f64|v_div_fixup_f64|v_ldexp_f64|v_frexp_\w+_f64|v_trig_preop_f64|v_fract_f64|v_floor_f64|v_ceil_f64|v_rndne_f64|v_trunc_f64|v_cmp_class_f64"),
    ("valu_cmp_f64", r"v_cmpx?_\w+_f64"),
    ("valu_cvt", r"v_cvt_\w+"),
    ("valu_f32_trans", r"v_rcp_f32|v_rsq_f32|v_sqrt_f32|v_sin_f32|v_cos_f32|v_exp_f32|v_log_f32|v_rcp_iflag_f32"),
    ("valu_f32", r"v_(add|sub|subrev|mul|fma|fmac|mad|mac|min|max|med3|min3|max3|ldexp|fract|floor|rndne|trunc|div_\w+|frexp_\w+|mul_legacy)_f32|v_pk_\w+_f32"),
    ("valu_cmp_f32", r"v_cmpx?_\w+_f32"),
    ("valu_mul_int32", r"v_mul_lo_u32|v_mul_hi_u32|v_mul_hi_i32|v_mul_lo_i32|v_mad_u64_u32|v_mad_i64_i32|v_mul_u32_u24|v_mul_i32_i24|v_mad_u32_u24|v_mad_i32_i24|v_mul_hi_u32_u24"),
    ("valu_shift64", r"v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64|v_lshl_add_u64"),
    ("valu_cmp_int", r"v_cmpx?_\w+_[iu](16|32|64)|v_cmp_class_f32"),
    ("valu_crosslane", r"v_readlane_b32|v_readfirstlane_b32|v_writelane_b32|v_permlane\w+|v_mov_b32_dpp|\w+_dpp|v_mbcnt_\w+|ds_bpermute_b32|ds_permute_b32|ds_swizzle_b32"),
    ("valu_mov_select", r"v_mov_b32|v_mov_b64|v_cndmask_b32|v_accvgpr_\w+|v_swap_b32"),
    ("valu_int32", r"v_\w+"),
    ("lds", r"ds_\w+"),
    ("vmem", r"global_\w+|buffer_\w+|flat_\w+|scratch_\w+"),
    ("smem", r"s_load_\w+|s_buffer_load_\w+|s_memtime|s_memrealtime"),
    ("branch", r"s_cbranch_\w+|s_branch|s_setpc_b64|s_swappc_b64|s_endpgm"),
    ("wait", r"s_waitcnt\w*|s_nop|s_barrier|s_sleep"),
    ("salu", r"s_\w+"),
]
CLASSES = [(n, re.compile("^(" + p + ")$")) for n, p in CLASSES]


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        if base.startswith("v_"):
            pass
    for name, rx in CLASSES:
        if rx.match(base):
            return name
    return "other"


def kernel_body(path, pattern):
    out, on = [], False
    for line in open(path):
        if not on:
            if line.startswith("_Z") and pattern in line.split(":")[0] and ":" in line:
                on = True
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.lstrip().startswith(".Lfunc_end"):
            break
        out.append(line)
    return out


def main():
    path, pattern = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    ops = collections.Counter()
    for line in kernel_body(path, pattern):
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if re.match(r"^[a-z][a-z0-9_]+$", op):
            ops[op] += 1
    cls = collections.Counter()
    for op, n in ops.items():
        cls[classify(op)] += n
    total = sum(ops.values())
    if "--json" in sys.argv:
        print(json.dumps({"kernel": pattern, "instructions": total, "classes": dict(cls), "opcodes": dict(ops.most_common())}, indent=1))
        return
    print("%s: %d instructions" % (pattern, total))
    for name, n in cls.most_common():
        print("  %-18s %6d  %5.1f %%" % (name, n, 100.0 * n / total))
    print("top opcodes:")
    for op, n in ops.most_common(top):
        print("  %-28s %6d  %-16s" % (op, n, classify(op)))


if __name__ == "__main__":
    main()
