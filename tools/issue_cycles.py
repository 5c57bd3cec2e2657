#!/usr/bin/env python3
"""Issue-cycle accounting of the kernels of one bench run: how many of the cycles a SIMD had did its
vector ALU spend ISSUING -- counted, not estimated.

    issue cycles = sum over instruction classes of (wave64 instructions the hardware counted in the
                   class, SQ_INSTS_VALU_*) x (cycles one instruction of the class occupies the SIMD's
                   issue port, measured by tools/microbench/valu_issue.hip on the same GPU)
    available    = duration of the kernel (kernel trace) x shader clock x 1024 SIMDs; the clock is
                   GRBM_GUI_ACTIVE / duration of the same launches

The class counters cover arithmetic (add / mul / fma / transcendental in f32 and f64, int32, int64,
conversions).  What they leave over -- SQ_INSTS_VALU minus their sum: moves, selects, compares,
shifts and logic, cross-lane reads -- is priced by the kernel's STATIC mix of exactly those opcodes
(tools/isa_histogram.py over the assembly, option --isa) at each opcode's measured cost; without the
assembly at the mean of move, select and compare.  The hardware's own busy measure is printed beside
it: SQ_ACTIVE_INST_VALU (quad-cycles a wave spent in VALU instructions) x 4 / available.

Beside that weighted sum stand two ISSUE-SLOT figures.  In the micro-benchmark's mixed streams an f32
fma alternating with a compare, a conversion, a max or a scalar add of the SAME wave takes 4.28 cycles
per pair whichever the partner (although compare, conversion and max cost 4.1 back to back): a wave's
instruction takes a slot of about 2.14 cycles, and a 4-cycle class keeps only its own unit busy
beyond the slot.
    valu_slot_frac  = VALU instructions x slot / available: what the vector issue port is busy with
                      if every vector instruction overlaps the tail of the one before (lower bound)
    all_slot_frac   = (VALU + SALU + LDS + VMEM + SMEM) x slot / available: the same if scalar and
                      memory instructions of OTHER waves took slots of the vector port too.  They do
                      not always: the f32 traversal kernel (8 waves per SIMD) comes out above 1, so
                      this is an upper bound, and the truth lies between the two
    f64_unit_frac   = f64 arithmetic x its back-to-back cost / available (the double-precision unit)
issue_cycle_frac above prices every VALU instruction at its back-to-back cost, as if no two classes
overlapped.

    issue_cycles.py dump <kt.db> <a.db> <b.db> <c.db>                 (GPU box) -> raw counters, JSON
    issue_cycles.py combine <tag> <valu_issue.json> <bench.json> <raw.json> [--isa context.s]
                                                                      -> profiles/<tag>_issue_cycles.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
N_SIMD = 1024


def dump(kt_db, dbs):
    from make_profiles import kernel_stats, pmc_sums, short
    out = {}
    for n, calls, tot, avg, mn, mx, pct in kernel_stats(kt_db):
        if "ptx::" in short(n):
            out[short(n)] = {"trace_calls": calls, "trace_total_ns": tot}
    for db in dbs:
        acc, launches = pmc_sums(db)
        for k, cs in acc.items():
            if short(k) in out:
                e = out[short(k)]
                e["launches"] = launches[k]
                e.update(cs)
    print(json.dumps(out, indent=1))


def opcode_cost(c, op):
    """measured cycles of one static opcode (tools/microbench/valu_issue.hip names)"""
    op = re.sub(r"_(e32|e64)$", "", op)
    table = [(r"v_mov_b32|v_accvgpr", "mov_b32"), (r"v_mov_b64", "mov_b64"), (r"v_cndmask_b32", None),
             (r"v_cmpx?_\w+_f64|v_cmp_class_f64", "cmp_lt_f64"), (r"v_cmpx?_\w+_f32|v_cmp_class_f32", "cmp_lt_f32"),
             (r"v_cmpx?_\w+", "cmp_lt_u32"), (r"v_readlane", "readlane"), (r"v_readfirstlane", "readfirstlane"),
             (r"v_writelane", "writelane"), (r"v_mbcnt", "mbcnt"), (r"v_(and|or|xor|not)_b32", "and_b32"),
             (r"v_(lshlrev|lshrrev|ashrrev)_b32|v_(lshlrev|lshrrev|ashrrev)_i32|v_bfe|v_bfi|v_ffb|v_bcnt|v_alignbit", "lshlrev_b32"),
             (r"v_(and_or|or3|lshl_or|xad|perm)_\w+", "and_or_b32")]
    for rx, name in table:
        if re.match("^(" + rx + ")", op):
            if name is None:
                return 0.5 * (c["add_cndmask_pair"] - c["add_u32"]) + 0.5 * c["cndmask_e64_sgpr"]   # vcc form and SGPR-pair form
            return c[name]
    return None


def remainder_price(c, isa, pattern):
    from isa_histogram import kernel_body
    import collections
    ops = collections.Counter()
    for line in kernel_body(isa, pattern):
        s = line.strip()
        if s and not s.startswith((";", ".")) and not s.endswith(":"):
            op = s.split()[0]
            if op.startswith("v_"):
                ops[op] += 1
    tot = n = 0.0
    used = {}
    for op, k in ops.items():
        cost = opcode_cost(c, op)
        if cost is not None:
            tot += cost * k
            n += k
            used[op] = k
    return tot / n, int(n), sum(ops.values())


def mangled_pattern(kernel):
    """'ptx::k_trace_f64<true, true, 1>' -> substring of its mangled name"""
    m = re.match(r"(?:ptx::)?(\w+)<(.*)>", kernel)
    if not m:
        return kernel
    name, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    enc = ""
    for a in args:
        if a in ("true", "false"):
            enc += "Lb%dE" % (a == "true")
        elif re.match(r"^-?\d+$", a):
            enc += "Li%sE" % a
        elif a == "double":
            enc += "d"
        elif a == "float":
            enc += "f"
        else:
            return name
    return "%d%sI%sE" % (len(name), name, enc)


def combine(tag, cost_json, bench_json, raw_json, isa):
    cost = json.load(open(cost_json))["cycles_per_wave_instruction_per_simd"]
    c = {k: min(v["2"], v["4"]) for k, v in cost.items()}      # issue cost: two or more waves per SIMD
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    raw = json.load(open(raw_json))
    price = {
        "SQ_INSTS_VALU_ADD_F32": c["add_f32"], "SQ_INSTS_VALU_MUL_F32": c["mul_f32"], "SQ_INSTS_VALU_FMA_F32": c["fma_f32"],
        "SQ_INSTS_VALU_TRANS_F32": (c["rcp_f32"] + c["rsq_f32"] + c["sqrt_f32"] + c["sin_f32"] + c["exp_f32"]) / 5,
        "SQ_INSTS_VALU_INT32": (c["add_u32"] + c["add3_u32"] + c["mul_lo_u32"]) / 3,
        "SQ_INSTS_VALU_INT64": (c["lshlrev_b64"] + c["lshl_add_u64"] + c["mad_u64_u32"]) / 3,
        "SQ_INSTS_VALU_CVT": (c["cvt_f64_f32"] + c["cvt_f32_f64"]) / 2,
        "SQ_INSTS_VALU_ADD_F64": c["add_f64"], "SQ_INSTS_VALU_MUL_F64": c["mul_f64"], "SQ_INSTS_VALU_FMA_F64": c["fma_f64"],
        "SQ_INSTS_VALU_TRANS_F64": (c["rcp_f64"] + c["rsq_f64"] + c["sqrt_f64"]) / 3,
    }
    cnd = c["add_cndmask_pair"] - c["add_u32"]
    default_rest = (c["mov_b32"] + c["mov_b64"] + cnd + c["cndmask_e64_sgpr"] + c["cmp_lt_f64"] + c["cmp_lt_u32"]) / 6
    out = {"tag": tag, "bench": {k: bench[k] for k in ("metric", "value", "unit", "dtype", "ms_per_step")},
           "workload": bench["config"]["workload"],
           "opcode_cost_cycles": {k: round(v, 3) for k, v in c.items()},
           "opcode_cost_note": "min over 2 and 4 waves per SIMD; cndmask_b32 back to back is an artefact of the loop (20 cycles), "
                               "its cost between other instructions is add_cndmask_pair - add_u32",
           "class_price_cycles": {k: round(v, 3) for k, v in price.items()}, "kernels": {}}
    for k, e in sorted(raw.items(), key=lambda kv: -kv[1].get("trace_total_ns", 0)):
        if "SQ_INSTS_VALU" not in e or "GRBM_GUI_ACTIVE" not in e:
            continue
        per = lambda name: e.get(name, 0.0) / e["launches"]
        ns = e["trace_total_ns"] / e["trace_calls"]
        gui = per("GRBM_GUI_ACTIVE")      # summed over the instances rocprofv3 reports (one per XCD): peak clock 2.4 GHz
        inst = 1
        while gui / inst / ns > 2.5:
            inst += 1
        clock_ghz = gui / inst / ns
        avail = ns * clock_ghz * N_SIMD
        rest_price, rest_static, valu_static = default_rest, None, None
        if isa:
            try:
                rest_price, rest_static, valu_static = remainder_price(c, isa, mangled_pattern(k))
            except Exception as ex:  # kernel not in this assembly file
                print("issue_cycles: no static mix for %s (%s)" % (k, ex), file=sys.stderr)
        classes = {n: per(n) for n in price}
        valu = per("SQ_INSTS_VALU")
        rest = max(0.0, valu - sum(classes.values()))
        cycles = {n: classes[n] * price[n] for n in price}
        cycles["unclassified (moves, selects, compares, shifts and logic, cross-lane)"] = rest * rest_price
        issue = sum(cycles.values())
        slot = 0.5 * c.get("mix_fma_max", 2.0 * c["fma_f32"])
        others = {n: per("SQ_INSTS_" + n) for n in ("SALU", "LDS", "VMEM", "SMEM")}
        all_insts = valu + sum(others.values())
        f64_unit = sum(cycles[n] for n in price if n.endswith("_F64"))
        act = per("SQ_ACTIVE_INST_VALU")
        lanes = per("SQ_THREAD_CYCLES_VALU") / act / 64.0 if act else None
        out["kernels"][k] = {
            "launches_profiled": e["trace_calls"], "avg_launch_ms": round(ns / 1e6, 4), "shader_clock_ghz": round(clock_ghz, 3),
            "valu_wave_instructions_per_launch": round(valu), "per_class": {n: round(v) for n, v in classes.items()},
            "unclassified_valu": round(rest), "unclassified_price_cycles": round(rest_price, 3),
            "unclassified_static_opcodes": rest_static, "static_valu_opcodes": valu_static,
            "issue_cycles_per_launch": round(issue), "available_simd_cycles_per_launch": round(avail),
            "issue_cycle_frac": round(issue / avail, 4),
            "issue_slot_cycles": round(slot, 3), "instructions_per_launch": {"VALU": round(valu), **{n: round(v) for n, v in others.items()}},
            "valu_slot_frac": round(valu * slot / avail, 4), "all_slot_frac": round(all_insts * slot / avail, 4),
            "f64_unit_frac": round(f64_unit / avail, 4),
            "hw_valu_busy_frac": round(4.0 * act / avail, 4) if act else None,
            "issue_cycle_share_by_class": {n: round(v / issue, 4) for n, v in sorted(cycles.items(), key=lambda kv: -kv[1])},
            "mean_cycles_per_valu_instruction": round(issue / valu, 3) if valu else None,
            "active_lanes_per_valu_instruction": round(lanes, 4) if lanes else None,
            "salu_per_valu": round(per("SQ_INSTS_SALU") / valu, 3) if valu else None,
            "lds_per_valu": round(per("SQ_INSTS_LDS") / valu, 4) if valu else None,
            "vmem_per_valu": round(per("SQ_INSTS_VMEM") / valu, 4) if valu else None,
            "wait_any_share_of_wave_cycles": round(per("SQ_WAIT_ANY") / per("SQ_WAVE_CYCLES"), 4) if per("SQ_WAVE_CYCLES") else None,
        }
    path = os.path.join(ROOT, "profiles", tag + "_issue_cycles.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    return out


def main():
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3:6])
    else:
        tag, cost_json, bench_json, raw_json = sys.argv[2:6]
        isa = sys.argv[sys.argv.index("--isa") + 1] if "--isa" in sys.argv else None
        out = combine(tag, cost_json, bench_json, raw_json, isa)
        for k, v in out["kernels"].items():
            print("%-46s %8.3f ms  slots VALU %.3f .. all %.3f  f64 unit %.3f  weighted VALU %.3f  hw busy %s  lanes %s  wait %s  cycles/inst %.2f" % (
                k.replace("ptx::", ""), v["avg_launch_ms"], v["valu_slot_frac"], v["all_slot_frac"], v["f64_unit_frac"], v["issue_cycle_frac"], v["hw_valu_busy_frac"],
                v["active_lanes_per_valu_instruction"], v["wait_any_share_of_wave_cycles"], v["mean_cycles_per_valu_instruction"]))


if __name__ == "__main__":
    main()
