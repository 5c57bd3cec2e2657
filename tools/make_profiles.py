#!/usr/bin/env python3
"""Turns the rocprofv3 databases of one profiling session into the summaries kept under profiles/.

usage: make_profiles.py <tag> <bench.json> <kernel-trace.db> <sq-pmc.db> <fetch-pmc.db> <write-pmc.db> [scene]

Writes profiles/<tag>_kernel_stats.csv (per-kernel calls / total / average, as rocprofv3 --stats
reports them), profiles/<tag>_pmc.json (per-kernel counter sums and per-launch HBM traffic) and
profiles/calibration.json (calibration_<scene>.json for a scene other than cbox: VALU instructions per 64-lane bounce iteration of the tracer and its
measured HBM bytes per launch, which bench.py quotes next to its live timings).
"""
import collections
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tables(cur, prefix):
    return [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith(prefix)]


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    kd = tables(cur, "rocpd_kernel_dispatch")[0]
    ks = tables(cur, "rocpd_info_kernel_symbol")[0]
    names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
    acc = collections.defaultdict(lambda: [0, 0, 1 << 62, 0])
    for kid, start, end in cur.execute(f"select kernel_id, start, end from {kd}"):
        a = acc[names[kid]]
        d = end - start
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in acc.values())
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
    return [(n, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / total) for n, a in rows]


def pmc_sums(path):
    cur = sqlite3.connect(path).cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    ki, ci, vi, di = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in cur.execute("select * from counters_collection"):
        acc[r[ki]][r[ci]] += r[vi]
        launches[r[ki]].add(r[di])
    return acc, {k: len(v) for k, v in launches.items()}


def short(name):
    if name.startswith("_Z"):  # some rocprofv3 runs record mangled names (with a ".kd" suffix)
        import subprocess
        try:
            name = subprocess.run(["c++filt", name[:-3] if name.endswith(".kd") else name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    elif name.endswith(".kd"):
        name = name[:-3]
    return name.split("(")[0].replace("void ", "")


def main():
    tag, bench_json, kt_db, sq_db, fetch_db, write_db = sys.argv[1:7]
    scene = sys.argv[7] if len(sys.argv) > 7 else "cbox"
    out_dir = os.path.join(ROOT, "profiles")
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
        for n, calls, tot, avg, mn, mx, pct in kernel_stats(kt_db):
            f.write('"%s",%d,%d,%.0f,%d,%d,%.2f\n' % (short(n), calls, tot, avg, mn, mx, pct))
    pmc = {}
    for label, db in (("sq", sq_db), ("fetch", fetch_db), ("write", write_db)):
        acc, launches = pmc_sums(db)
        for k, c in acc.items():
            if "ptx::" not in k:
                continue
            e = pmc.setdefault(short(k), {"launches": launches[k]})
            for name, v in c.items():
                e[name] = v
    # per-launch HBM traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
    # 128-byte read requests at 64 bytes, so it is doubled
    for k, e in pmc.items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_read_bytes_per_launch"] = 2.0 * e["FETCH_SIZE"] * 1024 / e["launches"]
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024 / e["launches"]
    cal = {}
    modes = {"f64": ("_f64", bench["step_stats"]["wave_iterations"] / bench["step_stats"]["passes"]),
             "f32": ("_f32", bench["other_mode"]["wave_iterations"] / bench["other_mode"]["passes"])}
    if bench["dtype"] == "f32":
        modes = {"f32": ("_f32", modes["f64"][1]), "f64": ("_f64", modes["f32"][1])}
    for mode, (kname, iters_per_launch) in modes.items():
        for k, e in pmc.items():
            # the reference estimator's tracer (bench.py also runs the physically based one once, for the record)
            if "k_trace" in k and "fast" not in k and kname in k and "SQ_INSTS_VALU" in e:
                cal[mode] = {"valu_insts_per_wave_iteration": round(e["SQ_INSTS_VALU"] / e["launches"] / iters_per_launch, 1),
                             "active_lanes_per_valu_inst": round(e["SQ_THREAD_CYCLES_VALU"] / e["SQ_ACTIVE_INST_VALU"] / 64.0, 4)
                             if e.get("SQ_ACTIVE_INST_VALU") else None,
                             "hbm_bytes_per_launch": round(e.get("hbm_read_bytes_per_launch", 0) + e.get("hbm_write_bytes_per_launch", 0)),
                             "paths_per_launch": bench["config"]["paths_per_step"] // bench["step_stats"]["passes"],
                             "kernel": k.replace("ptx::", ""),
                             "source": "profiles/%s_pmc.json (rocprofv3 --pmc on bench.py, %s)" % (tag, k)}
        # the wavefront tracer is several kernels per batch (camera rays, traversal and shading per
        # bounce, resolve): their counters are summed and quoted per batch, under the name of the
        # traversal kernel (what the library reports as the tracer that ran)
        suffix = "<double" if mode == "f64" else "<float"
        group = {k: e for k, e in pmc.items() if "k_wf_" in k and (kname in k or suffix in k) and "SQ_INSTS_VALU" in e}
        trav = [k for k in group if "k_wf_traverse" in k]
        if trav and mode not in cal:
            batches = max(1, min(e["launches"] for k, e in group.items() if "k_wf_resolve" in k) if any("k_wf_resolve" in k for k in group) else 1)
            tot = lambda name: sum(e.get(name, 0.0) for e in group.values())
            cal[mode] = {"valu_insts_per_wave_iteration": round(tot("SQ_INSTS_VALU") / batches / iters_per_launch, 1),
                         "active_lanes_per_valu_inst": round(tot("SQ_THREAD_CYCLES_VALU") / tot("SQ_ACTIVE_INST_VALU") / 64.0, 4)
                         if tot("SQ_ACTIVE_INST_VALU") else None,
                         "hbm_bytes_per_launch": round((2.0 * tot("FETCH_SIZE") + tot("WRITE_SIZE")) * 1024 / batches),
                         "paths_per_launch": bench["config"]["paths_per_step"] // bench["step_stats"]["passes"],
                         "kernel": trav[0].replace("ptx::", ""),
                         "kernels_summed": sorted(group),
                         "source": "profiles/%s_pmc.json (rocprofv3 --pmc on bench.py, the %d k_wf_* kernels of a batch summed)" % (tag, len(group))}
    json.dump(pmc, open(os.path.join(out_dir, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    cal_name = "calibration.json" if scene == "cbox" else "calibration_%s.json" % scene
    json.dump(cal, open(os.path.join(out_dir, cal_name), "w"), indent=1, sort_keys=True)
    print(json.dumps(cal, indent=1))


if __name__ == "__main__":
    main()
