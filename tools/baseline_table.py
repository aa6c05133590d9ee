#!/usr/bin/env python3
"""Fills BASELINE.md section 4 on the GPU box: tools/baseline_table.py [--skip-cpu]
  * CPU restatement (oracle pipeline: torch-CPU networks + C glue) on the 1-min wav with 1 thread (the reference's ORT
    intra-op setting, onnx_model.cc:26-27,43) and with all usable cores, and on the first 600 s of the synthetic hour (all cores);
    per-stage wall times labelled like the reference's timers (sd.cpp:3028, 3110, 3231, 3434)
  * GPU: the 1-min wav through the speakerDiarizer CLI, bench.py at 10 min and 1 h (f32) and 1 h fp16
Writes gpurun_out/baseline_table.json and prints the markdown rows."""
import json, os, subprocess, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import sdhip, synth, weightpack as nn
from oracle import orc, pipeline_oracle, nn_oracle

out = {"host": {"cpu_count": os.cpu_count(), "usable": len(os.sched_getaffinity(0)),
                "model": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")}}
ws, we = nn.synth_segmentation_weights(4321), nn.synth_embedding_weights(4322)
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", ws); nn.save_pack(tmp + "/e.sdw", we)
wav = os.path.join(ROOT, "tests", "golden", "multi-speaker_1min.wav")


def cpu_run(pcm, threads, planted=None):
    """oracle pipeline with the reference's four timers"""
    torch.set_num_threads(threads)
    t0 = time.time()
    w = (pcm.astype(np.float32) * np.float32(1.0)) / np.float32(32768.0)
    nc, last_len = orc.num_chunks(len(w))
    net = nn_oracle.PyanNetOracle(ws)
    seg = np.zeros((nc, 293, 3), np.float32)
    full = nc - 1 if (0 < last_len < 80000) else nc
    for b0 in range(0, full, 32):
        b1 = min(full, b0 + 32)
        seg[b0:b1] = net(np.stack([w[i * 8000:i * 8000 + 80000] for i in range(b0, b1)])).numpy()
    if full < nc:
        y = net(w[None, full * 8000:]).numpy()[0]
        seg[full, :y.shape[0]] = y[:293]
    t1 = time.time()
    emb = pipeline_oracle.diarize_ref(pcm, ws, we, seg_override=seg if planted is None else planted[0], return_all=True)[1]["emb"]
    t2 = time.time()
    turns = pipeline_oracle.diarize_ref(pcm, ws, we, seg_override=seg if planted is None else planted[0], emb_override=emb, planted=planted)
    t3 = time.time()
    return {"threads": threads, "segmentation_s": round(t1 - t0, 2), "embedding_s": round(t2 - t1, 2), "clustering_s": round(t3 - t2, 2),
            "total_s": round(t3 - t0, 2), "rtf": round(len(pcm) / 16000.0 / (t3 - t0), 4), "turns": len(turns)}


if "--skip-cpu" not in sys.argv:
    pcm, sr, ch = sdhip.read_wav(wav)
    allc = max(1, min(32, out["host"]["usable"]))
    out["cpu_1min_wav_all_cores"] = cpu_run(pcm, allc)
    print(json.dumps(out["cpu_1min_wav_all_cores"]), flush=True)
    out["cpu_1min_wav_1_thread"] = cpu_run(pcm, 1)
    print(json.dumps(out["cpu_1min_wav_1_thread"]), flush=True)
    p600 = synth.make_pcm(3600.0, 1234, limit=600 * 16000)
    nc = synth.num_chunks(len(p600))
    sc, asg = synth.planted_scores(synth.with_duets(synth.schedule(3600.0, 1234, limit=len(p600))), len(p600), 0, nc)
    out["cpu_first_600s_all_cores_planted"] = cpu_run(p600, allc, planted=(sc, synth.planted_embeddings(asg)))
    print(json.dumps(out["cpu_first_600s_all_cores_planted"]), flush=True)

# ---- GPU
exe = os.path.join(PKG, "speakerDiarizer")
t0 = time.time()
r = subprocess.run([exe, tmp + "/s.sdw", tmp + "/e.sdw", wav], capture_output=True, text=True)
wall = time.time() - t0
cost = [l for l in r.stdout.splitlines() if l.startswith("Time cost")]
out["gpu_1min_wav_cli"] = {"process_wall_s": round(wall, 2), "time_cost_line": cost[0] if cost else None, "rtf_process": round(59.0 / wall, 1),
                           "turns": sum(1 for l in r.stdout.splitlines() if "--> Speaker_" in l)}
print(json.dumps(out["gpu_1min_wav_cli"]), flush=True)
for name, args in [("gpu_10min", ["--hours-per-gpu", str(600 / 3600.0)]), ("gpu_1h", []), ("gpu_1h_fp16", ["--precision", "f16"])]:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--cpu-seconds", "0"] + args, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    j = json.loads(line[-1]) if line else {"error": r.stderr[-500:]}
    out[name] = j
    if "value" in j:
        st = j["other_kernels"].get("stft_mel", {})
        print(name, j["value"], "x RT,", j["ms_per_step"], "ms; stft", st.get("hbm_GBps_algorithmic"), "GB/s; conv_gemm", j["roofline"]["achieved"], "TF", j["roofline"]["frac"], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "baseline_table.json"), "w"), indent=1)
