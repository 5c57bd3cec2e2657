"""world_size-2 test of the multi-GPU sharding logic on CPU (gloo).  The render function is the
CPU oracle here (the GPU core has no CPU fallback); what is under test is the sharding arithmetic
and the final exchange of pathtracer_amd/distributed.py, which bench.py uses unchanged with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sample_shards_tile_the_range():
    for spp, world in [(256, 8), (10, 4), (3, 8), (16, 1)]:
        parts = [D.shard_samples(spp, r, world) for r in range(world)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == spp
        for (o0, c0), (o1, _) in zip(parts, parts[1:]):
            assert o0 + c0 == o1
    rows = np.arange(64)
    cover = np.zeros(64, int)
    for r in range(3):
        b = D.shard_bands(r, 3, band=4)
        cover += ((rows // b["band"]) % b["row_stride"] == b["row_phase"])
    assert (cover == 1).all()


def _worker(rank, world, port, scheme, ret):
    sys.path.insert(0, ROOT)
    from oracle import oracle, scene_io
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scene_io.load_conf(os.path.join(ROOT, "config/cbox.conf"), root_dir=ROOT, width=24, height=24, spp=6)
    O = oracle.Oracle(sc, seed=13)

    def render_fn(out, spp, sample_offset, spp_total, row_stride=1, row_phase=0, band=1):
        img = O.render(spp=spp, sample_offset=sample_offset, spp_total=spp_total, threads=1)
        if row_stride > 1:
            rows = np.arange(img.shape[0])
            img[(rows // band) % row_stride != row_phase] = 0
        out.copy_(torch.from_numpy(img))

    out = torch.zeros((24, 24, 3), dtype=torch.float64)
    D.render_frame(render_fn, out, 6, rank, world, scheme=scheme, band=4)
    if rank == 0:
        full = O.render(spp=6, sample_offset=0, spp_total=6, threads=1)
        ret["max_diff"] = float(np.abs(out.numpy() - full).max())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scheme", ["samples", "bands"])
def test_two_ranks_reproduce_the_single_process_frame(scheme):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, scheme, ret), nprocs=2, join=True)
    # sample ranges: sums re-associated across ranks -> 1 ulp; bands: exact union
    assert ret["max_diff"] <= (1e-14 if scheme == "samples" else 0.0)


@pytest.mark.parametrize("scheme", ["samples", "bands"])
def test_bench_starts_its_own_ranks(scheme):
    """`python bench.py --gpus 2` as the driver invokes it -- no launcher, WORLD_SIZE unset -- starts
    two ranks itself, and rank 0's line comes back through the parent with the world size the
    process group saw.  Rehearsed on the CPU: gloo, the oracle as the render function."""
    import json
    import subprocess

    from oracle import oracle, scene_io

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "24",
           "--spp", "3", "--scheme", scheme, "--rehearse", "bench_rehearsal"]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                  # ONE line on stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1 and res["value"] > 0
    assert res["scaling"] == ("weak" if scheme == "samples" else "strong")
    total = 6 if scheme == "samples" else 3                 # weak scaling: 3 spp per rank
    assert res["config"]["global_spp"] == total
    sc = scene_io.load_conf(os.path.join(ROOT, "config/cbox.conf"), root_dir=ROOT, width=24, height=24, spp=total)
    full = oracle.Oracle(sc, seed=13).render(threads=1)
    assert abs(res["frame_checksum"] - float(full.sum())) <= 1e-10
    # both cuts of the frame are in the one line, whichever the headline value is: weak = 3 spp per rank,
    # strong = the two ranks share one 3-spp frame; each with the duration of its exchange
    assert res["weak"]["global_spp"] == 6 and res["strong"]["global_spp"] == 3
    assert res["weak"]["value"] > 0 and res["strong"]["value"] > 0
    assert res["weak"]["reduce_ms"] >= 0 and res["strong"]["reduce_ms"] >= 0
    assert abs(res[("weak" if scheme == "samples" else "strong")]["value"] / res["value"] - 1) < 1e-3


def test_bench_reports_failing_ranks():
    """A rank that dies takes the parent's exit status with it (no result line)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "8",
           "--spp", "1", "--rehearse", "no_such_module"]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert proc.returncode != 0 and proc.stdout.strip() == ""
