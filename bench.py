#!/usr/bin/env python3
"""bench.py -- real-time factor of the MI355X diarization hot path (BASELINE.json metric).

One step = one pass of the whole path (int16 PCM resident in HBM -> speaker turns on the host)
over N hours of synthetic 16 kHz mono audio on N GPUs: every rank runs segmentation + embeddings
for its contiguous chunk range (multiple of 32 chunks), segmentation scores and
embeddings are all-gathered with RCCL, rank 0 runs counting / clustering / reconstruction.
N = 1 is BASELINE.json configs[2] (1 h, single GPU, full pipeline).

Prints ONE JSON line on rank 0 (contract in the task brief) including
  roofline     : the dominant kernel (f32-MFMA conv_gemm) measured live with HIP events on the
                 library's own stream over the timed region
  cpu_baseline : the oracle ("port") timed on this box's host cores on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import tempfile
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HOUR = 3600
SR = 16000
F32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0         # same table: "Peak BF16/FP16 MFMA ~2.5 PF dense"


def cpu_baseline(ws, we, seconds):
    """oracle pipeline (torch CPU + C restatement) on `seconds` of the same synthetic audio"""
    import synth
    from oracle import pipeline_oracle
    pcm = synth.make_pcm(seconds, seed=1234)
    # threads actually used: the cores this process may run on, capped at 32 (torch's intra-op pool
    # collapses on small batches when handed hundreds of threads)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(32, avail)))
    t0 = time.time()
    turns = pipeline_oracle.diarize_ref(pcm, ws, we)
    dt = time.time() - t0
    return {"value": round(seconds / dt, 4), "unit": "x real-time (audio-s / wall-s)", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "first %d s of the synthetic hour (seed 1234): %d chunks, %d embedding items, oracle "
            "pipeline (torch-CPU PyanNet/ECAPA fp32 + C restatement), %.1f s wall" % (seconds, (seconds * SR - 80000) // 8000 + 1,
                                                                                      3 * ((seconds * SR - 80000) // 8000 + 1), dt),
            "turns": len(turns)}


def rank0_share_estimate(world, hours_per_gpu):
    """share of the chunks for rank 0 such that its finalize + inference takes as long as the other ranks' inference.
    Stage rates measured on MI355X (profiles/r01_bench_1h_v5_summary.txt, r01_bench_8h_on_1gpu.json): inference 2.23 s per hour
    of audio, finalize 0.13 s at 1 h and about 1.9 s at 8 h (~ h^1.28).  A wrong estimate only unbalances the ranks."""
    if world == 1:
        return 1.0
    total_h = world * hours_per_gpu
    t_inf = 2.23 * total_h
    t_fin = 0.132 * total_h ** 1.28 + 0.02
    s0 = (t_inf - (world - 1) * t_fin) / world              # s0 + t_fin == (t_inf - s0) / (world - 1)
    return max(0.0, min(1.0 / world, s0 / t_inf))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hours-per-gpu", type=float, default=1.0)
    ap.add_argument("--cpu-seconds", type=int, default=60, help="audio seconds for the cpu_baseline sample (0 = skip)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"], help="f32 = the measured configuration (f32 MFMA); f16 = "
                    "BASELINE configs[4]: ECAPA conv layers on the fp16 MFMA with f32 accumulation (secondary, tolerance-checked mode)")
    ap.add_argument("--rank0-share", type=float, default=-1.0, help="fraction of the chunks rank 0 infers itself (it also finalizes: count / "
                    "clustering / reconstruction).  The other ranks start step k+1 right after the all-gather of step k, so rank 0's finalize(k) "
                    "overlaps their inference; a smaller rank-0 share balances finalize + inference on rank 0 against inference on the others. "
                    "-1 = from the measured stage rates (see rank0_share_estimate), 1/N = equal shares, 0 = rank 0 only finalizes")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path (process group, all-gather, assembly) even with "
                    "one rank: exercises RCCL and the collectives on a 1-GPU box")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real path) | gloo (plumbing test of the multi-rank code "
                    "on a box with fewer GPUs than ranks: gathers go through host memory, ranks may share a GPU)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0 and world > 1:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libsdhip has no CPU fallback")
    if a.backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        if a.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import sdhip
    import synth
    import weightpack as nn                     # seeded synthetic weights + .sdw container (package data, not the oracle)

    tmp = tempfile.mkdtemp(prefix="sdw_r%d_" % rank)
    ws, we = nn.synth_segmentation_weights(4321), nn.synth_embedding_weights(4322)
    nn.save_pack(os.path.join(tmp, "segment.sdw"), ws)
    nn.save_pack(os.path.join(tmp, "embedding.sdw"), we)
    d = sdhip.Diarizer(os.path.join(tmp, "segment.sdw"), os.path.join(tmp, "embedding.sdw"), local)
    if a.precision == "f16":
        d.set_option("ecapa_precision", 1)

    # ---- the job: `world` x hours_per_gpu of audio; rank r owns a contiguous, 32-aligned chunk range
    per_samples = int(round(a.hours_per_gpu * HOUR * SR))
    n_total = per_samples * world
    C, _ = sdhip.num_chunks(n_total)
    share0 = a.rank0_share if a.rank0_share >= 0 else rank0_share_estimate(world, a.hours_per_gpu)
    per, ranges = sdhip.plan_ranks(n_total, world, share0 if world > 1 else None)
    pieces_g = sdhip.gather_pieces(per, ranges)
    contiguous = all(off == sum(n for _, n in pieces_g[:i]) for i, (off, _) in enumerate(pieces_g))
    lo, hi = ranges[rank]
    first, need_hi = sdhip.shard_sample_range(lo, hi, n_total)
    # synthesise only what this rank reads: its own hour(s) + the 72 000-sample halo of the next one
    pieces, pos = [], first
    while pos < need_hi:
        h = pos // per_samples
        off = pos - h * per_samples
        take = min(per_samples - off, need_hi - pos)
        seg_pcm = synth.make_pcm(a.hours_per_gpu * HOUR, seed=1234 + h, limit=off + take)
        pieces.append(seg_pcm[off:off + take])
        pos += take
    pcm_host = np.concatenate(pieces) if pieces else np.zeros(1, np.int16)
    d_pcm = torch.from_numpy(pcm_host).to(dev)
    nloc = hi - lo
    d_seg = torch.zeros((per, sdhip.FRAMES, 3), dtype=torch.float32, device=dev)
    d_emb = torch.zeros((per * 3, sdhip.EMB_DIM), dtype=torch.float32, device=dev)
    if use_dist:
        g_seg = torch.empty((world * per, sdhip.FRAMES, 3), dtype=torch.float32, device=dev)
        g_emb = torch.empty((world * per * 3, sdhip.EMB_DIM), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    turns_box = [None]

    def step():
        if nloc > 0:
            d.shard_infer_dev(d_pcm.data_ptr(), first, int(d_pcm.numel()), n_total, lo, hi, d_seg.data_ptr(), d_emb.data_ptr())
        if use_dist:
            if a.backend == "gloo":                      # test-only path: same assembly through host memory
                hs, he = [torch.empty(d_seg.shape) for _ in range(world)], [torch.empty(d_emb.shape) for _ in range(world)]
                dist.all_gather(hs, d_seg.cpu())
                dist.all_gather(he, d_emb.cpu())
                g_seg.copy_(torch.cat(hs))
                g_emb.copy_(torch.cat(he))
            else:
                dist.all_gather_into_tensor(g_seg, d_seg)
                dist.all_gather_into_tensor(g_emb, d_emb)
            torch.cuda.synchronize()
            if rank == 0:
                # gathered layout: rank r's shard in slot [r * per, (r + 1) * per).  Rank 0 returns to the next all-gather only
                # after this call: the other ranks' next inference runs meanwhile (software pipeline over steps)
                if contiguous:
                    fs, fe = g_seg, g_emb
                else:
                    fs = torch.cat([g_seg[o:o + m] for o, m in pieces_g])
                    fe = torch.cat([g_emb[3 * o:3 * (o + m)] for o, m in pieces_g])
                    torch.cuda.synchronize()
                turns_box[0] = d.finalize_dev(fs.data_ptr(), fe.data_ptr(), C, n_total)
        else:
            turns_box[0] = d.finalize_dev(d_seg.data_ptr(), d_emb.data_ptr(), C, n_total)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    d.set_option("profile", 1)
    d.reset_stats()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / max(a.steps, 1) * 1e3
    audio_s = n_total / SR

    if rank == 0:
        cg = d.kernel_stats("conv_gemm")
        stages = d.stage_ms()
        ach = cg["flops"] / max(cg["ms"], 1e-9) / 1e9      # TFLOP/s
        extra = {}
        for k in ("stft_mel", "lstm_rec", "pdist", "linkage", "row_nn", "se_apply", "asp_pool"):
            s = d.kernel_stats(k)
            extra[k] = {"ms_per_step": round(s["ms"] / max(a.steps, 1), 3), "launches_per_step": s["launches"] // max(a.steps, 1)}
            if k == "stft_mel" and s["ms"] > 0:       # front end (north star: HBM GB/s for the STFT): algorithmic bytes of SURVEY 8(d), 481 492 B per live item
                extra[k].update({"hbm_GBps_algorithmic": round(s["bytes"] / s["ms"] / 1e6, 1), "hbm_frac_of_8TBps": round(s["bytes"] / s["ms"] / 1e6 / 8000.0, 4),
                                 "fp64_mfma_TFLOPs": round(s["flops"] / s["ms"] / 1e9, 1), "fp64_mfma_frac_of_78.6": round(s["flops"] / s["ms"] / 1e9 / 78.6, 3),
                                 "bound": "fp64 MFMA (400-point DFT as GEMM), not HBM, at this size"})
        traffic, traffic_src, mfma_util = None, None, None
        peak = F32_MFMA_PEAK_TFLOPS if a.precision == "f32" else F16_MFMA_PEAK_TFLOPS
        pmc_path = os.path.join(ROOT, "profiles", "pmc_conv_gemm_bench.json")
        if world == 1 and a.precision == "f32" and os.path.exists(pmc_path):
            try:
                pj = json.load(open(pmc_path))
                traffic, traffic_src = pj["bytes_per_launch"], pj["source"]
                mfma_util = pj.get("mfma")
            except Exception:
                pass
        out = {
            "metric": "real-time factor (audio-sec/wall-sec), %g h 16 kHz mono per GPU" % a.hours_per_gpu,
            "value": round(audio_s / (ms_per_step / 1e3), 2),
            "unit": "x real-time",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "%g h synthetic 16 kHz mono per GPU (%g h total), full pipeline: PyanNet segmentation + "
                                   "post-seg + STFT/fbank + ECAPA-TDNN + centroid AHC + reconstruction" % (a.hours_per_gpu, audio_s / HOUR),
                       "audio_seconds": audio_s, "chunks": C, "embedding_items": 3 * C,
                       "weights": "seeded synthetic (seg 4321, emb 4322): the reference's ONNX blobs are not in the checkout",
                       "sharding": "contiguous 32-aligned chunk ranges %s, RCCL all-gather of scores+embeddings (slots of %d chunks), clustering on rank 0; "
                                   "rank 0 infers %.1f %% of the chunks so that its finalize(step k) + inference balances the other ranks' "
                                   "inference(step k+1), which starts right after the all-gather" % (ranges if world <= 8 else ranges[:8], per, 100.0 * (ranges[0][1] - ranges[0][0]) / max(C, 1)),
                       "turns": len(turns_box[0] or []),
                       "turns_crc32": zlib.crc32("\n".join(sdhip.format_turn(t) for t in (turns_box[0] or [])).encode()),
                       "stage_ms_last_step": {"segmentation": round(stages[0], 1), "embedding": round(stages[1], 1), "clustering": round(stages[2], 1)}},
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_conv_gemm (v_mfma_f32_32x32x2_f32)" if a.precision == "f32" else "k_conv_gemm (v_mfma_f32_32x32x16_f16; segmentation and skinny layers stay f32)", "launches_per_step": cg["launches"] // max(a.steps, 1),
                         "kernel_ms_per_step": round(cg["ms"] / max(a.steps, 1), 2),
                         "algorithmic_gflop_per_step": round(cg["flops"] / max(a.steps, 1) / 1e9, 1),
                         "algorithmic_bytes_per_launch": round(cg["bytes"] / max(cg["launches"], 1)),
                         "mfma_utilisation_pmc": mfma_util},
            "other_kernels": extra,
        }
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(ws, we, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    d.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
