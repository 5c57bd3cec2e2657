#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s on the Cornell box at 1024 x 1024, 256 spp per GPU.

One "step" = one full pass of the hot path over the workload: camera rays, up to 15 bounces of
closest hit / shading / NEE with visibility per path, resolve, accumulate -- i.e.
cPathTracer::Process of the reference for
config/cbox.conf at 1024 x 1024 x 256 spp (the configuration the headline metric is quoted on).  With N GPUs every rank
renders its own range of 256 samples of the same image (weak scaling: 256 * N spp in total) and the
per-pixel sums are reduced onto rank 0 over RCCL inside the timed region (--scheme bands: the ranks
share the rows of one frame and rank 0 gathers each rank's packed rows).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode f64|f32] [--spp S] [--size W]

N > 1: either under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`, one rank per GPU) or plainly as `python bench.py --gpus N`, in which case this process
starts that launcher itself -- before it has touched a GPU -- relays rank 0's line and exits with
the ranks' status.  Rank 0 prints ONE JSON line.  Inputs are resident in HBM when the timed region
starts; the output is a device buffer (no PCIe traffic inside the timed region).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, spec
N_SIMD = 1024                # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4              # peak engine clock
VALU_ISSUE_PEAK = N_SIMD * CLOCK_GHZ / 2.0   # G wave64 instructions/s: a SIMD-32 issues one every 2 cycles
VALU_ISSUE_PEAK_F64 = N_SIMD * CLOCK_GHZ / 4.0  # f64 arithmetic runs at half that rate (78.6 TFLOP/s FMA)


def algorithmic_bytes(mode, st, n_pix_touched, wavefront=False):
    """HBM bytes each kernel must move for the work of one step (scene tables excluded: they are a
    few KB read through the scalar cache / LDS).  R = size of the scalar type.
      k_trace ...... writes one 6-scalar (direct, coefficient) record per shaded segment and 3 result scalars
                     per path; rays, hits and NEE samples never leave the registers, and the records are read
                     back for the resolve pass by the wave that wrote them (from L2)
      k_accumulate . reads 3 scalars per path, reads + writes the f64 rgb sum of every pixel it touches"""
    R = 8 if mode == "f64" else 4
    rec, paths = st["records"], st["paths"]
    if wavefront:
        # the wavefront tracer of BVH scenes moves its segments through memory on purpose: per segment
        # the ray written by the shading kernel (6R + 4) and read by the traversal kernel (6R), the hit
        # written (R + 4) and the whole entry read by the shading kernel (7R + 8); per record 6R written
        # and read back by the resolve kernel; per path the end-state word (written, read) and 3R of result
        return {"trace": st["segments"] * (20 * R + 16) + rec * 12 * R + paths * (3 * R + 8),
                "accumulate": paths * 3 * R + n_pix_touched * 48}
    return {"trace": rec * 6 * R + paths * 3 * R,
            "accumulate": paths * 3 * R + n_pix_touched * 48}


def reference_stream_bytes(st, n_square):
    """Bytes per launch under the accounting that prices the reference algorithm's data streams
    (f32 SoA, per path segment: ray write+read 48, hit record write+read 32, 72 of triangle
    attributes, 24 pushed on the stack, 64 per shadow ray; per path: 24 read back per segment at
    the resolve + 12 accumulated; scenes of the all-triangles intersector are LDS-resident, BVH
    scenes add 32 per node and 48 per triangle the traversals visit, counted on the device):
    B_seg = 176 + 64 * N_shadow (+ BVH term), B_path = segments/path * (B_seg + 24) + 12.  The
    tracer here is designed NOT to move most of these bytes; the figure is the yardstick shared
    with a streaming implementation."""
    seg_per_path = st["segments"] / max(st["paths"], 1)
    b_seg = 176 + 64 * n_square
    bvh = 32 * st.get("bvh_nodes", 0) + 48 * st.get("bvh_tris", 0)
    return st["paths"] * (seg_per_path * (b_seg + 24) + 12) + bvh


def pmc_calibration(mode, scene="cbox"):
    """What only hardware counters can tell, from the committed PMC profile of this same command
    (tools/make_profiles.py): VALU instructions per 64-lane bounce iteration of the tracer
    (SQ_INSTS_VALU / wave iterations) and its HBM bytes per launch (FETCH_SIZE, WRITE_SIZE).  None
    when the profile is absent."""
    path = os.path.join(ROOT, "profiles", "calibration.json" if scene == "cbox" else "calibration_%s.json" % scene)
    try:
        with open(path) as f:
            return json.load(f).get(mode)
    except (OSError, ValueError):
        return None


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks of this same
    command under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1), pass
    everything they print on stderr through, print rank 0's JSON line, return their exit status.
    Nothing in this process has initialised a GPU."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes here
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    elif proc.returncode == 0:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


def host_cpu():
    """(model string of the host CPU, logical CPUs of the machine, CPUs this process may run on)"""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default=os.environ.get("PTX_BENCH_MODE", "f64"), choices=["f64", "f32"])
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--width", type=int, default=0, help="image width (default: --size)")
    ap.add_argument("--height", type=int, default=0, help="image height (default: --size)")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--conf", default="config/cbox.conf")
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-estimator", action="store_true",
                    help="skip the one step of the physically based estimator that is run for the record (profiling sessions: "
                         "the wavefront tracer's traversal kernel serves both estimators, and counters are summed per kernel)")
    ap.add_argument("--estimator", default="reference", choices=["reference", "fast"],
                    help="the estimator of the timed steps: the reference's own (the metric) or the physically based one "
                         "(profiling sessions of its kernels; the line then says so in config.estimator)")
    ap.add_argument("--no-scenes", action="store_true", help="skip the one timed step each of config/room.conf and config/mis.conf")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--scheme", default="samples", choices=["samples", "bands"],
                    help="multi-GPU sharding: 'samples' = every rank renders --spp samples of every pixel (weak "
                         "scaling, the default), 'bands' = the ranks share the rows of one --spp frame in interleaved "
                         "bands of 8 (strong scaling)")
    ap.add_argument("--rehearse", default="",
                    help="(tests) MODULE whose setup(args, rank, world, W, H) returns (render_fn, out tensor): the launch, "
                         "the sharding, the reduce, the timing and the result line run over gloo on the CPU with that "
                         "render function in place of the GPU core; the line is marked as a rehearsal and is no measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d inside a launcher of %d ranks: the two must agree" % (args.gpus, world))
    rehearsal = bool(args.rehearse)
    if not rehearsal:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the path-tracing core has no CPU fallback")
        torch.cuda.set_device(local_rank)
    # launched by torch.distributed.run (RANK and MASTER_PORT set): the collectives run whatever the
    # world size, so that a one-rank launch on a one-GPU box goes through the same RCCL calls as N > 1
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    W, H = args.width or args.size, args.height or args.size
    scene_name = os.path.splitext(os.path.basename(args.conf))[0]
    headline = scene_name == "cbox" and W == 1024 and H == 1024 and args.spp == 256
    spp_total = args.spp * world if args.scheme == "samples" else args.spp

    from pathtracer_amd.distributed import render_frame

    if rehearsal:
        import importlib
        render_fn, out = importlib.import_module(args.rehearse).setup(args, rank, world, W, H)
        ctx = info = None
    else:
        import pathtracer_amd as pt
        scene = pt.Scene.load_conf(os.path.join(ROOT, args.conf), root_dir=ROOT, width=W, height=H, spp=args.spp)
        info = scene.info
        stream = torch.cuda.current_stream()
        ctx = pt.Context(local_rank, stream=stream.cuda_stream, batch_paths=args.batch_paths or None)
        ctx.build_accel(scene)
        out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")

        def render_fn(o, **kw):
            return ctx.render_device(o.data_ptr(), **kw)

    def step(profile=False, mode=None, reduce=True, scheme=None, reduce_ms=None):
        # samples: rank r renders samples [r*spp, (r+1)*spp) of every pixel; bands: its rows of the
        # frame (the rest of `out` stays zero); then RCCL reduce(sum) -> rank 0
        scheme = scheme or args.scheme
        return render_frame(render_fn, out, args.spp * world if scheme == "samples" else args.spp, rank, world, scheme=scheme,
                            reduce=reduce and use_dist, collective=use_dist, reduce_ms=reduce_ms,
                            mode=mode or args.mode, seed=13, profile=profile,
                            **({"estimator": "fast"} if args.estimator == "fast" else {}))

    def fence():
        if use_dist:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    def timed(scheme, steps, warmup):
        """warmup untimed steps, then `steps` steps between two fences; the MAX over the ranks"""
        for _ in range(warmup):
            step(scheme=scheme)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            st_ = step(scheme=scheme)
        fence()
        dt_ = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt_], dtype=torch.float64, device=out.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, st_

    dt, st = timed(args.scheme, args.steps, args.warmup)
    paths_per_step = W * H * spp_total
    value = paths_per_step * args.steps / dt / 1e6
    frame_checksum = float(out.double().sum().item()) if rehearsal and rank == 0 else None   # of the frame the timed steps made

    # N > 1: the other way of cutting the frame too, in the same line, so that a scaling record cannot
    # be misread: "weak" = every rank renders --spp samples of every pixel (N x the work), "strong" =
    # the ranks share the rows of ONE --spp frame; each with the duration of its exchange
    both = None
    if world > 1:
        other = "bands" if args.scheme == "samples" else "samples"
        dt_o, _ = timed(other, args.steps, 1)
        rms = {}
        for sch in ("samples", "bands"):
            acc = []
            step(scheme=sch, reduce_ms=acc)
            fence()
            rms[sch] = acc[0] if acc else None
        if use_dist:
            for sch in rms:
                t = torch.tensor([rms[sch] or 0.0], dtype=torch.float64, device=out.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                rms[sch] = float(t.item())
        dts = {args.scheme: dt, other: dt_o}
        both = {
            "weak": {"scheme": "samples", "value": round(W * H * args.spp * world * args.steps / dts["samples"] / 1e6, 6), "unit": "Mpaths/s",
                     "ms_per_step": round(dts["samples"] / args.steps * 1e3, 3), "global_spp": args.spp * world,
                     "reduce_ms": None if rms["samples"] is None else round(rms["samples"], 3)},
            "strong": {"scheme": "bands", "value": round(W * H * args.spp * args.steps / dts["bands"] / 1e6, 6), "unit": "Mpaths/s",
                       "ms_per_step": round(dts["bands"] / args.steps * 1e3, 3), "global_spp": args.spp,
                       "reduce_ms": None if rms["bands"] is None else round(rms["bands"], 3)},
            "exchange": "weak: reduce(sum) of the whole f64 frame onto rank 0 (%d bytes per rank); strong: gather of every rank's "
                        "own rows, packed (%d bytes per rank)" % (W * H * 24, W * H * 24 // world)}
        both["weak"]["exchange_bytes_per_rank"] = W * H * 24
        both["strong"]["exchange_bytes_per_rank"] = W * H * 24 // world

    result = None
    if rank == 0 and rehearsal:
        result = {"metric": "Mpaths/s at %s %dx%d (whole job, all ranks)" % (scene_name, W, H), "value": round(value, 6),
                  "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
                  "scaling": "weak" if args.scheme == "samples" else "strong", "vs_baseline": None, "dtype": args.mode,
                  "data": "rehearsal on the CPU over gloo (%s): no measurement" % args.rehearse,
                  "config": {"workload": "%s %dx%d, %d spp per rank" % (scene_name, W, H, args.spp), "global_spp": spp_total,
                             "paths_per_step": paths_per_step},
                  "weak": both and both["weak"], "strong": both and both["strong"],
                  "frame_checksum": frame_checksum, "roofline": None, "cpu_baseline": None}
    elif rank == 0:
        # ---- roofline of the dominant kernel: one more step with HIP events around every launch
        # on the render stream (same workload, same stream; not part of the timed region above)
        ps = step(profile=True, reduce=False)   # rank 0 only: no collective in here
        torch.cuda.synchronize()
        ms = {"trace": ps["ms_trace"], "accumulate": ps["ms_accumulate"]}
        passes = int(ps["passes"])
        wavefront = ctx.last_kernel().startswith("k_wf_")
        ab = algorithmic_bytes(args.mode, ps, int(ps["paths"]) // args.spp * passes, wavefront)  # pixels of this rank x launches
        dom = max(ms, key=ms.get)
        kernels = {}
        for k in ms:
            gbs = ab[k] / (ms[k] * 1e-3) / 1e9 if ms[k] > 0 else 0.0
            kernels["k_" + k] = {"ms_per_step": round(ms[k], 3), "launches_per_step": passes,
                                 "algorithmic_bytes_per_launch": round(ab[k] / passes),
                                 "achieved_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
        avg_ms = ms[dom] / passes
        stream_bytes = reference_stream_bytes(ps, info.n_square_lights) / passes
        achieved = stream_bytes / (avg_ms * 1e-3) / 1e9
        design_bytes = ab[dom] / passes
        design_gbs = design_bytes / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "bytes_per_launch": round(stream_bytes), "avg_launch_ms": round(avg_ms, 5),
                    "launches_per_step": passes,
                    "bytes_per_path": round(stream_bytes * passes / max(ps["paths"], 1), 1),
                    "design_bytes_per_launch": round(design_bytes), "design_achieved": round(design_gbs, 2),
                    "design_frac": round(design_gbs / HBM_PEAK_GBS, 5),
                    "note": ("achieved prices the data streams of the algorithm (f32 SoA ray / hit / shadow / stack "
                             "streams, %.0f B per path on this scene); the tracer keeps rays, hits and light samples "
                             "in registers and LDS, so the bytes it must move (design_*) and the bytes it does move "
                             "(traffic, PMC) are far below that, and it is bound by vector-ALU issue, see valu"
                             if ctx.accel_info()["brute_force"] else
                             "achieved prices the data streams of the algorithm (f32 SoA ray / hit / shadow / stack "
                             "streams plus 32 B per BVH node and 48 B per triangle visited, %.0f B per path on this "
                             "scene); " + ("the wavefront tracer moves rays, hits and records through device-side queues "
                                           "between its traversal, shading and resolve kernels (design_*: what those streams "
                                           "amount to in this precision; traffic, PMC: all kernels of a batch, BVH records "
                                           "included); the traversal kernel is bound by the latency of dependent record "
                                           "fetches at 6 (f32: 8) waves per SIMD and by divergence, see valu" if wavefront else
                                           "rays and hits stay in registers, LDS and wave-private rings, the BVH records and "
                                           "the path records do go through memory (traffic, PMC); the kernel is bound by memory "
                                           "latency and traversal divergence, see valu"))
                            % (stream_bytes * passes / max(ps["paths"], 1))}
        # counter calibrations are per scene (and only quoted at the launch size they were collected at)
        cal = pmc_calibration(args.mode, scene_name)
        # a calibration speaks for ONE kernel: the tracer instantiation it was collected on must be
        # the one this run launched (a changed kernel or another plan makes the committed counters stale)
        ran = ctx.last_kernel()
        cal_kernel = (cal or {}).get("kernel") or ((cal or {}).get("source", "").rsplit("ptx::", 1)[-1].rstrip(")") if cal else None)
        if cal and cal_kernel != ran:
            print("bench.py: counter calibration of %s is for %s, this run launched %s: not quoted"
                  % (scene_name, cal_kernel, ran), file=sys.stderr)
            cal = None
        if cal and dom == "trace" and cal.get("paths_per_launch") == paths_per_step // world // passes:
            roofline["traffic"] = cal.get("hbm_bytes_per_launch")
            roofline["traffic_source"] = cal.get("source")
        valu = None
        if cal and ms["trace"] > 0:
            rate = cal["valu_insts_per_wave_iteration"] * ps["wave_iterations"] / (ms["trace"] * 1e-3) / 1e9
            valu = {"kernel": "k_trace", "achieved": round(rate, 1), "peak": VALU_ISSUE_PEAK,
                    "unit": "G wave64 VALU instructions/s", "frac": round(rate / VALU_ISSUE_PEAK, 4),
                    "valu_insts_per_wave_iteration": cal["valu_insts_per_wave_iteration"],
                    "peak_if_all_f64": VALU_ISSUE_PEAK_F64,
                    "path_lane_fraction": round(ps["segments"] / (64.0 * max(ps["wave_iterations"], 1)), 4),
                    "active_lanes_per_valu_inst": cal.get("active_lanes_per_valu_inst"),
                    "source": cal.get("source")}
        # the committed issue account of the kernel this run launched (tools/issue_cycles.py: hardware
        # instruction-class counters priced with the opcode costs measured on this GPU), quoted, not re-measured
        if valu is not None:
            acct_path = os.path.join(ROOT, "profiles", "r03_%s_issue_cycles.json" % scene_name)
            if os.path.exists(acct_path):
                acct = json.load(open(acct_path)).get("kernels", {})
                first = ran.split(" + ")[0]
                entry = next((v for k, v in acct.items() if k.replace("ptx::", "") == first), None)
                if entry:
                    valu["issue_account"] = {k: entry.get(k) for k in ("avg_launch_ms", "valu_slot_frac", "all_slot_frac", "f64_unit_frac",
                                                                      "issue_cycle_frac", "hw_valu_busy_frac", "salu_per_valu",
                                                                      "wait_any_share_of_wave_cycles")}
                    valu["issue_account"]["kernel"] = first
                    valu["issue_account"]["source"] = "profiles/r03_%s_issue_cycles.json" % scene_name
        kernel_ms = {k: round(v, 3) for k, v in ms.items()}
        kernel_ms["other"] = round(ps["ms_other"], 3)

        # the other precision, one step, for the record
        other_mode = "f32" if args.mode == "f64" else "f64"
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ost = step(mode=other_mode, reduce=False)
        torch.cuda.synchronize()
        other_value = W * H * args.spp / (time.perf_counter() - t1) / 1e6 if world == 1 else None

        # the physically based estimator on the same frame, one step, for the record (another
        # estimator: its paths are shorter and every light costs a visibility test; not the metric)
        fast = None
        if world == 1 and not args.no_fast_estimator and args.estimator != "fast":
            render_fn(out, spp=1, sample_offset=0, spp_total=1, mode=args.mode, seed=13, estimator="fast")   # its kernel and state, untimed
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fst = render_fn(out, spp=args.spp, sample_offset=0, spp_total=args.spp, mode=args.mode, seed=13, estimator="fast")
            torch.cuda.synchronize()
            fast = {"value": round(W * H * args.spp / (time.perf_counter() - t1) / 1e6, 3), "unit": "Mpaths/s", "dtype": args.mode,
                    "kernel": ctx.last_kernel(), "segments_per_path": round(fst["segments"] / fst["paths"], 3),
                    "shadow_rays_per_path": round(fst["shadow_rays"] / fst["paths"], 3)}

        cpu = None
        checks = None
        if not args.no_cpu_baseline and world == 1 and args.estimator == "reference":
            cpu, checks = cpu_baseline_and_checks(ctx, args, W, H)

        # the two BVH scenes of the configuration list, one timed step each at their own sizes, after the
        # timed region of the metric and outside it (headline run on one GPU only)
        scenes = None
        if headline and world == 1 and not args.no_scenes:
            scenes = {}
            for name, sw, sh, sspp in (("room", 1024, 1024, 64), ("mis", 768, 512, 256)):
                try:
                    scenes[name] = time_scene(name, sw, sh, sspp, args, local_rank, stream, check=not args.no_cpu_baseline)
                except Exception as e:  # a missing scene file must not cost the metric its line
                    scenes[name] = {"error": str(e)[:200]}

        result = {
            "metric": "Mpaths/s at %s %dx%d (whole job, all GPUs)" % (scene_name, W, H), "value": round(value, 3), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak" if args.scheme == "samples" else "strong",
            "vs_baseline": None, "dtype": args.mode, "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d spp per GPU, depth %d, %d triangles, %d lights (%d square) (%s%s)"
                                   % (scene_name, W, H, args.spp, info.depth, info.n_tris, info.n_lights, info.n_square_lights,
                                      args.conf, ", the headline configuration" if headline else ""),
                       "intersector": "all-triangles (plane records)" if ctx.accel_info()["brute_force"] else "BVH",
                       "tracer_kernel": ran, "state_bytes": int(ps["state_bytes"]), "estimator": args.estimator,
                       "global_spp": spp_total, "paths_per_step": paths_per_step,
                       "sharding": ("sample ranges per rank + RCCL reduce(sum) of the radiance sums to rank 0" if args.scheme == "samples"
                                    else "interleaved row bands of 8 per rank + RCCL gather of each rank's packed rows to rank 0")},
            "mrays_per_s": round((st["segments"] + st["shadow_rays"]) * world * args.steps / dt / 1e6, 1),
            "segments_per_path": round(st["segments"] / st["paths"], 3),
            "roofline": roofline, "valu": valu, "kernels": kernels, "kernel_ms_per_step": kernel_ms,
            "step_stats": {k: int(ps[k]) for k in ("paths", "segments", "shadow_rays", "records", "wave_iterations", "passes",
                                                   "bvh_nodes", "bvh_tris")},
            "other_mode": {"dtype": other_mode, "value": None if other_value is None else round(other_value, 3),
                           "wave_iterations": int(ost["wave_iterations"]), "passes": int(ost["passes"])},
            "fast_estimator": fast, "cpu_baseline": cpu, "checks": checks,
            "weak": both and both["weak"], "strong": both and both["strong"], "scenes": scenes,
        }
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def time_scene(name, W, H, spp, args, device, stream, check=True):
    """One timed step of another scene of the configuration list (f64 and f32, the reference's
    estimator): Mpaths/s over the whole step with the frame resident on the device, and -- with the CPU
    oracle at hand -- the first-hit face ids of all its camera rays against the oracle's."""
    import torch

    import pathtracer_amd as pt
    conf = "config/%s.conf" % name
    scene = pt.Scene.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=W, height=H, spp=spp)
    ctx = pt.Context(device, stream=stream.cuda_stream)
    ctx.build_accel(scene)
    out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    rec = {"workload": "%s %dx%d, %d spp, depth %d, %d triangles, %d lights (%s)" % (
        name, W, H, spp, scene.info.depth, scene.info.n_tris, scene.info.n_lights, conf)}
    for mode in (args.mode, "f32" if args.mode == "f64" else "f64"):
        ctx.render_device(out.data_ptr(), mode=mode, spp=1, seed=13)          # kernels and state, untimed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ctx.render_device(out.data_ptr(), mode=mode, spp=spp, seed=13)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        key = "value" if mode == args.mode else "value_" + mode
        rec[key] = round(W * H * spp / dt / 1e6, 3)
        if mode == args.mode:
            rec.update({"unit": "Mpaths/s", "dtype": mode, "ms_per_step": round(dt * 1e3, 3), "tracer_kernel": ctx.last_kernel(),
                        "segments_per_path": round(st["segments"] / st["paths"], 3),
                        "mrays_per_s": round((st["segments"] + st["shadow_rays"]) / dt / 1e6, 1),
                        "bvh_nodes_per_ray": round(st["bvh_nodes"] / max(st["segments"] + st["shadow_rays"], 1), 2),
                        "bvh_tris_per_ray": round(st["bvh_tris"] / max(st["segments"] + st["shadow_rays"], 1), 2)})
    if check:
        from oracle import oracle, scene_io
        ref = scene_io.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=W, height=H, spp=1)
        oids = oracle.Oracle(ref, seed=13).first_hit()
        ids = ctx.first_hit()
        rec.update({"first_hit_mismatches": int((ids != oids).sum()), "first_hit_pixels": int(ids.size)})
    return rec


def cpu_baseline_and_checks(ctx, args, W, H):
    """The CPU oracle (a port of the reference's OpenMP row loop) timed on a bounded sample of the
    same workload, and the two correctness gates next to the throughput number."""
    from oracle import oracle, scene_io

    # every CPU this process may run on (its affinity mask / the harness's OMP_NUM_THREADS share of a
    # larger host), not a fixed cap
    model, host_cpus, usable = host_cpu()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or usable
    ref = scene_io.load_conf(os.path.join(ROOT, args.conf), root_dir=ROOT, width=W, height=H, spp=1)
    O = oracle.Oracle(ref, seed=13)
    # calibrate on a few rows, then pick the spp that fills the time budget
    ref_small = scene_io.load_conf(os.path.join(ROOT, args.conf), root_dir=ROOT, width=256, height=256, spp=2)
    t0 = time.perf_counter()
    oracle.Oracle(ref_small, seed=13).render(threads=threads)
    rate = 256 * 256 * 2 / (time.perf_counter() - t0)
    spp = int(max(1, min(args.spp, round(rate * args.cpu_seconds / (W * H)))))
    t0 = time.perf_counter()
    oimg, ost = O.render(spp=spp, threads=threads, stats=True)
    cpu_dt = time.perf_counter() - t0
    cpu = {"value": round(W * H * spp / cpu_dt / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "port",
           "cpu_model": model, "host_logical_cpus": host_cpus, "cpus_usable_by_this_process": usable,
           "threads_from": "OMP_NUM_THREADS" if os.environ.get("OMP_NUM_THREADS") else "affinity mask",
           "sample": "%s %dx%d at %d spp (%.1f s, oracle/pt_oracle.c, OpenMP dynamic over rows, f64, %s)"
                     % (os.path.splitext(os.path.basename(args.conf))[0], W, H, spp, cpu_dt,
                        "BVH" if ref.verts.shape[0] > 256 else "all-triangles loop")}
    # gates: first-hit ids and the image at equal spp
    ids = ctx.first_hit()
    oids = O.first_hit()
    gimg, _ = ctx.render(mode=args.mode, spp=spp, seed=13)
    rmse = oracle.rmse_rel(gimg, oimg)
    checks = {"first_hit_mismatches": int((ids != oids).sum()), "first_hit_pixels": int(ids.size),
              "rmse_vs_cpu_equal_spp": rmse, "spp": spp}
    if args.mode == "f64":
        # f64 runs the oracle's arithmetic on the oracle's random numbers: the bar is 1e-9 per pixel, and a
        # pixel beyond it must be matched, sample by sample, to a named tie alternative of its path
        # (oracle/parity.py); a bounded number of them is put to that proof here, all of them in the tests
        from oracle import parity
        r = parity.check_f64_image(gimg, oimg, ctx, O, range(spp), seed=13, max_pixels=24)
        checks.update({"f64_pixels_beyond_1e-9": r["beyond"], "put_to_the_proof": r["put_to_the_proof"],
                       "proven_ties": r["proven_ties"], "amplified_roundings": r["amplified"],
                       "largest_amplification": max(r["amplifications"]) if r["amplifications"] else None,
                       "unmatched_samples": len(r["unmatched"]),
                       "largest_difference": r["largest_difference"], "rmse_bar": 1e-4,
                       "pass": bool((ids == oids).all() and not r["unmatched"] and r["no_sample_differs"] == 0
                                    and r["explained"] == r["put_to_the_proof"] and rmse < 1e-4)})
    else:
        # f32: statistical bar  max(2 x the oracle's own two-seed noise floor, 1e-3); the floor costs a
        # second CPU render and is only measured at low spp -- without it the bar is the 1e-3 it can never go below
        o2 = oracle.Oracle(ref, seed=14).render(spp=spp, threads=threads) if spp <= 4 else None
        floor = oracle.rmse_rel(o2, oimg) if o2 is not None else None
        tol = 1e-3 if floor is None else max(2 * floor, 1e-3)
        checks.update({"rmse_tolerance": tol, "oracle_two_seed_rmse": floor,
                       "pass": bool((ids == oids).all() and rmse <= tol)})
    return cpu, checks


if __name__ == "__main__":
    main()
