"""world_size-2 gloo test (CPU) of the multi-GPU data path: shard planning, the padded
all-gather of per-rank segmentation scores / embeddings, and the rank-0 assembly must reproduce the
single-process arrays.  The per-rank "inference" is a deterministic function of the chunk index
(the HIP kernels need a GPU; tests/test_gpu_parity.py::test_sharded_equals_unsharded covers them)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import sdhip


def _fake_infer(lo, hi):
    idx = torch.arange(lo, hi, dtype=torch.float32)
    seg = (idx[:, None, None] * 0.001 + torch.arange(293)[None, :, None] * 1e-6 + torch.arange(3)[None, None, :] * 1e-9).float()
    emb = (torch.arange(lo * 3, hi * 3, dtype=torch.float32)[:, None] + torch.arange(192)[None, :] / 1000.0)
    return seg.contiguous(), emb.contiguous()


def _worker(rank, world, port, n_total, q, dedicated=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C, _ = sdhip.num_chunks(n_total)
    per, ranges, off = sdhip.plan_ranks(n_total, world, dedicated)
    lo, hi = ranges[rank]
    seg = torch.zeros((per, 293, 3))
    emb = torch.zeros((per * 3, 192))
    s, e = _fake_infer(lo, hi)
    seg[:hi - lo] = s
    emb[:(hi - lo) * 3] = e
    gs = [torch.zeros_like(seg) for _ in range(world)]
    ge = [torch.zeros_like(emb) for _ in range(world)]
    dist.all_gather(gs, seg)
    dist.all_gather(ge, emb)
    if rank == 0:
        S, E = torch.cat(gs)[off:off + C], torch.cat(ge)[off * 3:(off + C) * 3]
        s0, e0 = _fake_infer(0, C)
        q.put((bool(torch.equal(S, s0)), bool(torch.equal(E, e0)), per, ranges))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_single_process_arrays():
    n_total = 16000 * 1300                      # 2591 chunks -> shards of 1312 and 1279 chunks
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_s, ok_e, per, ranges = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_s and ok_e
    assert per % 32 == 0 and ranges[0] == (0, per) and ranges[1][0] == per


def test_three_rank_gather_with_dedicated_finalizer():
    """bench.py's plan from 4 GPUs up, at world 3 here: rank 0 holds no chunks and finalizes, ranks 1-2 infer; the
    gathered buffer holds chunk 0 at slot 1"""
    n_total = 16000 * 900
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, n_total, q, True)) for r in range(3)]
    for p in procs:
        p.start()
    ok_s, ok_e, per, ranges = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_s and ok_e
    assert ranges[0] == (0, 0) and ranges[1] == (0, per) and ranges[2][0] == per


def test_shard_plan_properties():
    for n_total, world in [(57600000, 1), (57600000 * 2, 2), (57600000 * 8, 8), (944000, 4), (100000, 8)]:
        C, _ = sdhip.num_chunks(n_total)
        per, ranges = sdhip.plan_shards(n_total, world)
        assert per % 32 == 0 and len(ranges) == world
        assert ranges[0][0] == 0 and ranges[-1][1] == C
        for (a, b), (c, d) in zip(ranges, ranges[1:]):
            assert b == c and a <= b
        for lo, hi in ranges:
            assert hi == lo or (lo * 3) % 32 == 0          # every non-empty shard starts on a reference batch boundary
            s0, s1 = sdhip.shard_sample_range(lo, hi, n_total)
            assert 0 <= s0 <= s1 <= n_total
