"""scratch GPU check #2: segmentation, post-seg, clustering, reconstruction, whole path"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
from oracle import nn_oracle as nn, orc

wdir = "/tmp/sdw"; os.makedirs(wdir, exist_ok=True)
ws = nn.synth_segmentation_weights(); nn.save_pack(wdir + "/segment.sdw", ws)
we = nn.synth_embedding_weights(); nn.save_pack(wdir + "/embedding.sdw", we)
d = sdhip.Diarizer(wdir + "/segment.sdw", wdir + "/embedding.sdw")
rng = np.random.default_rng(0)

def report(name, a, b, rtol=1e-3, atol=1e-4):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b); tol = atol + rtol * np.abs(b)
    print("%-28s max_abs %.3e  max_err/tol %.3f  ref_absmax %.3e" % (name, np.nanmax(err), np.nanmax(err / tol), np.nanmax(np.abs(b))), flush=True)

# ---- segmentation: 3 full chunks + short tail
n = 80000 + 8000 * 2 + 3000
wav = (0.1 * rng.standard_normal(n)).astype(np.float32) * (1 + np.sin(np.arange(n) / 2000.0)).astype(np.float32)
seg = d.segment(wav)
nc, ll = orc.num_chunks(n)
print("chunks", seg.shape, nc, ll)
m = nn.PyanNetOracle(ws)
ref = np.zeros((nc, 293, 3), np.float32)
for i in range(nc):
    ch = wav[i * 8000: i * 8000 + 80000]
    y = m(ch[None, :]).numpy()[0]
    ref[i, :y.shape[0]] = y
report("segmentation full chunks", seg[:nc - 1], ref[:nc - 1])
report("segmentation tail chunk", seg[nc - 1], ref[nc - 1])

# ---- postseg on crafted scores (include exact-onset ties impossible in f32, so just random)
c = 40
sc = rng.random((c, 293, 3)).astype(np.float32)
sc[:, :, 2] *= 0.5
sc[5] = 0.1            # inactive chunk
sc[6, :, 0] = 0.9; sc[6, :, 1] = 0.9   # all overlap -> clean mask empty
b_gpu, m_gpu, cnt_gpu = d.postseg(sc)
b_ref = orc.binarize(sc); m_ref = orc.select_masks(b_ref); cnt_ref, win, _ = orc.speaker_count(b_ref)
print("binarize equal", np.array_equal(b_gpu.astype(np.float64), b_ref), "masks equal", np.array_equal(m_gpu, m_ref),
      "count equal", np.array_equal(cnt_gpu, cnt_ref), len(cnt_gpu), len(cnt_ref))

# ---- clustering
for N in (12, 300, 2000):
    cen = rng.standard_normal((4, 192))
    X = cen[rng.integers(0, 4, N)] + 0.6 * rng.standard_normal((N, 192))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    t = time.time(); Z = d.linkage(X); t1 = time.time() - t
    T_ref, Z_ref = orc.ahc(X, 0.7153814381597874)
    T = d.cluster(X, 0.7153814381597874)
    print("N=%d linkage Z bit-equal %s  maxdiff %.3e  labels equal %s  (%.3f s)" % (N, np.array_equal(Z, Z_ref), np.abs(Z - Z_ref).max(), np.array_equal(T, T_ref), t1), flush=True)
P = np.array([[0, 0], [0, 1], [1, 0], [0, 4], [0, 3], [1, 4], [4, 0], [3, 0], [4, 1], [4, 4], [3, 4], [4, 3]], float)
print("toy", d.cluster(P, 1.1), orc.ahc(P, 1.1)[0])
# full Cluster::clustering with NaN rows and small clusters
cN = 200
emb = np.zeros((cN, 3, 192))
cen = rng.standard_normal((5, 192)) * 2
lab = rng.integers(0, 5, (cN, 3))
lab[rng.random((cN, 3)) < 0.03] = 4
emb = cen[lab] + 0.5 * rng.standard_normal((cN, 3, 192))
emb[lab == 4] = cen[4] + 0.05 * rng.standard_normal(((lab == 4).sum(), 192))
emb = emb.astype(np.float32).astype(np.float64)
emb[rng.random((cN, 3)) < 0.2] = np.nan
h_gpu, K = d.clustering(emb)
h_ref, K_ref, _ = orc.clustering(emb)
print("clustering hard equal", np.array_equal(h_gpu, h_ref), "K", K, K_ref)

# ---- reconstruct
c = 30
n_s = 80000 + 8000 * (c - 1)
sc = rng.random((c, 293, 3)).astype(np.float32)
b_gpu, m_gpu, cnt_gpu = d.postseg(sc)
hard = rng.integers(0, 3, (c, 3)).astype(np.int32)
turns = d.reconstruct(sc, b_gpu, hard, cnt_gpu, n_s)
b_ref = orc.binarize(sc); cnt_ref, win, ft = orc.speaker_count(b_ref)
h2 = orc.mark_inactive(b_ref, hard)
binr, st = orc.reconstruct(sc, h2, cnt_ref, win, ft, n_s)
t_ref = orc.to_annotation(binr, st)
key = lambda t: (round(t[0], 9), t[2])
print("reconstruct turns", len(turns), len(t_ref), "equal", sorted(turns, key=key) == sorted(t_ref, key=key))
if sorted(turns, key=key) != sorted(t_ref, key=key):
    print(turns[:5]); print(t_ref[:5])

# ---- whole path on ~40 s of synthetic audio vs oracle pipeline
n = 16000 * 40 + 1234
t_ = np.arange(n) / 16000.0
wav = (0.3 * np.sin(2 * np.pi * 150 * t_) * (np.sin(2 * np.pi * 0.3 * t_) > 0) + 0.05 * rng.standard_normal(n)).astype(np.float32)
pcm = np.clip(np.round(wav * 32768), -32768, 32767).astype(np.int16)
t = time.time(); turns = d.diarize(pcm); t1 = time.time() - t
print("diarize: %d turns in %.2f s; stage ms %s" % (len(turns), t1, d.stage_ms()))
for tt in turns[:10]: print("  ", sdhip.format_turn(tt))
# oracle pipeline
wf = pcm.astype(np.float32) / 32768.0
nc, ll = orc.num_chunks(n)
seg_g = d.segment(wf)
b_ref = orc.binarize(seg_g); masks = orc.select_masks(b_ref); cnt_ref, win, ft = orc.speaker_count(b_ref)
emb_g = d.embed(wf, masks)
hard_ref, K_ref, _ = orc.clustering(emb_g.astype(np.float64).reshape(nc, 3, 192))
hard_ref = orc.mark_inactive(b_ref, hard_ref)
binr, st = orc.reconstruct(seg_g, hard_ref, cnt_ref, win, ft, n)
t_ref = orc.to_annotation(binr, st)
print("whole path vs (gpu NN + oracle rest): equal", sorted(turns, key=key) == sorted(t_ref, key=key), len(t_ref), "K", K_ref)
