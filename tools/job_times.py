"""Wall time and stage times of eight consecutive jobs of one context (the first is the cold one): tools/job_times.py [hours] (OPTS=key=value,...)"""
import os, sys, time, tempfile, numpy as np, torch
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip, synth, weightpack as nn
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights(4321)); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights(4322))
t_c = time.perf_counter()
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
print("sd_create %.1f ms" % ((time.perf_counter() - t_c) * 1e3))
for kv in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = kv.split("="); d.set_option(k, int(v)); print("option", k, v)
sec = 3600 * hours
pcm = synth.make_pcm(sec, seed=1234); n = len(pcm)
dev = torch.device("cuda", 0)
d_pcm = torch.from_numpy(pcm).to(dev)
nc = synth.num_chunks(n)
sc, asg = synth.planted_scores(synth.with_duets(synth.schedule(sec, 1234)), n, 0, nc)
d_sc, d_pe = torch.from_numpy(sc).to(dev), torch.from_numpy(synth.planted_embeddings(asg)).to(dev)
d.set_planted(d_sc.data_ptr(), d_pe.data_ptr(), 0, nc)
for prof in (0, 0, 0, 1, 1, 1, 0, 0):
    d.set_option("profile", prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = d.diarize_dev(d_pcm.data_ptr(), n)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("profile %d: %.1f ms, stages %s, %d turns" % (prof, (t1 - t0) * 1e3, [round(x, 1) for x in d.stage_ms()], len(t)), flush=True)
