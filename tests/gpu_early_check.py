"""scratch GPU check used while bringing kernels up (superseded by tests/test_gpu_*.py)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
from oracle import nn_oracle as nn, orc

out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
wdir = "/tmp/sdw"; os.makedirs(wdir, exist_ok=True)
we = nn.synth_embedding_weights(); nn.save_pack(wdir + "/embedding.sdw", we)
d = sdhip.Diarizer(None, wdir + "/embedding.sdw")
rng = np.random.default_rng(0)

def report(name, a, b, rtol=1e-3, atol=1e-4):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b); tol = atol + rtol * np.abs(b)
    print("%-28s max_abs %.3e  max_err/tol %.3f  ref_absmax %.3e  nan(a,b)=%d,%d" % (name, np.nanmax(err), np.nanmax(err / tol), np.nanmax(np.abs(b)), np.isnan(a).sum(), np.isnan(b).sum()), flush=True)

# --- ecapa body
items = 5
feats = (3.0 * rng.standard_normal((items, 501, 80))).astype(np.float32)
lens = np.array([1.0, 0.7311, 0.25, 0.5, 0.9991], np.float32)
t = time.time(); e_gpu = d.ecapa(feats, lens); t1 = time.time() - t
e_ref, inter = nn.EcapaOracle(we)(feats, lens, return_intermediate=True)
report("ecapa emb", e_gpu, e_ref.numpy())
cos = torch.nn.functional.cosine_similarity(torch.from_numpy(e_gpu), e_ref).numpy()
print("ecapa cosine distance max", (1 - cos).max(), "gpu time", t1)

# --- frontend
n = 8000 * 20 + 80000
wav = (0.2 * rng.standard_normal(n)).astype(np.float32)
wav *= (1 + np.sin(np.arange(n) / 3000.0)).astype(np.float32)
items = 64
masks = (rng.random((items, 293)) > 0.4).astype(np.float32)
masks[3] = 0; masks[3, :2] = 1          # too short (546 samples)
masks[7] = 1                             # full
masks[9] = 0                             # empty
masks[40:64] = 0; masks[40:64, :1] = 1   # whole second batch... only items 40-63 short
f_gpu, l_gpu = d.frontend(wav, masks)
sigs = np.zeros((items, 80000), np.float32); cnts = np.zeros(items, np.int64)
for i in range(items):
    ch = orc.crop(wav, (i // 3) * 8000)
    sigs[i], cnts[i] = orc.mask_compact(ch, masks[i])
l_ref = np.zeros(items, np.float32); ts = np.zeros(items, bool)
for b0 in range(0, items, 32):
    l, t_, an = orc.wav_lens(cnts[b0:b0 + 32]); l_ref[b0:b0 + 32] = l; ts[b0:b0 + 32] = t_ | an
print("wav_lens equal:", np.array_equal(l_gpu, l_ref), "too_short", ts.sum())
st = nn.stft_ref(sigs, we["stft.window"])
f_ref = nn.fbank_norm_ref(st, l_ref, we["fbank.matrix"]).numpy()
ok = ~ts
report("frontend feats (valid)", f_gpu[ok], f_ref[ok])
f64 = nn.fbank_norm_ref(st, l_ref, we["fbank.matrix"], torch.float64).numpy()
report("oracle f32 vs f64 feats", f_ref[ok], f64[ok])
report("gpu vs f64 feats", f_gpu[ok], f64[ok])

# --- full embed
t = time.time(); e_gpu = d.embed(wav, masks); t1 = time.time() - t
e_ref = nn.EcapaOracle(we)(f_ref, l_ref).numpy()
e_ref[ts] = np.nan
print("nan rows equal:", np.array_equal(np.isnan(e_gpu[:, 0]), ts))
report("embed (valid rows)", e_gpu[ok], e_ref[ok])
cos = torch.nn.functional.cosine_similarity(torch.from_numpy(e_gpu[ok]), torch.from_numpy(e_ref[ok])).numpy()
print("embed cosine distance max", (1 - cos).max(), "gpu time", t1)
# timing of a bigger batch
d.set_option("profile", 1)
items = 768
masks = (rng.random((items, 293)) > 0.3).astype(np.float32)
n = 8000 * (items // 3) + 80000
wav = (0.2 * rng.standard_normal(n)).astype(np.float32)
for rep in range(2):
    d.reset_stats(); t = time.time(); e = d.embed(wav, masks); t1 = time.time() - t
    cg = d.kernel_stats("conv_gemm"); sm = d.kernel_stats("stft_mel")
    print("768 items: wall %.3f s; conv_gemm %.1f ms %.1f TFLOP/s (%d launches); stft_mel %.2f ms %.1f GB/s" % (t1, cg["ms"], cg["flops"] / cg["ms"] / 1e9, cg["launches"], sm["ms"], sm["bytes"] / sm["ms"] / 1e6), flush=True)
    for k in ("se_mean", "se_apply", "asp_stats", "asp_pool", "fbank_norm"):
        s = d.kernel_stats(k); print("   %s %.2f ms %.0f GB/s" % (k, s["ms"], s["bytes"] / max(s["ms"], 1e-9) / 1e6))
