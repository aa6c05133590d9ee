#!/usr/bin/env python3
"""Where the fp16 mode's error comes from: relative L2 error of the stored activations (block0 output, the three block outputs, MFA,
pooled statistics, embedding) of fp16 mode against f32 mode, on the 32-item batches of the planted 10-min set that hold the worst items.
tools/diag_fp16_layers.py [option=value ...]   (options applied to the fp16 run, e.g. ecapa_f16_hp=1)"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'pyannote-audio_speaker-diarization_cpp_amd'); sys.path.insert(0, 'tests')
import sdhip, weightpack as nn, tempfile
from test_planted import planted_case, nan_rule
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights())
d = sdhip.Diarizer(None, tmp + "/e.sdw")
pcm, scores, assign, embp = planted_case(600.0, 1234)
b, masks, counts, bad = nan_rule(scores)
wav = pcm.astype(np.float32) / np.float32(32768.0)
feats, lens = d.frontend(wav, masks)
sel = []
for w in (2509, 3240, 3315, 3036, 460, 2167):
    b0 = (w // 32) * 32
    sel += [i for i in range(b0, b0 + 32) if not bad[i] and i not in sel]
sel = np.array(sel)
f, l = np.ascontiguousarray(feats[sel]), np.ascontiguousarray(lens[sel])
n = len(sel)
d.set_option("ecapa_keep_cat", 1)
d.set_option("skip_dead_rows", 0)           # every item 501 rows in every space: [item][501][ld]
nv = np.minimum(501, np.maximum(1, np.ceil(l * np.float32(501)).astype(int)))
valid = (np.arange(501)[None, :] < nv[:, None])
def grab(half):
    dt = np.float16 if half else np.float32
    out = {}
    out["x0"] = d.read_ws("ec_x0", dt, n * 501 * 1024).reshape(n, 501, 1024).astype(np.float32)
    cat = d.read_ws("ec_cat", dt, n * 501 * 3072).reshape(n, 501, 3072).astype(np.float32)
    hp = half and any(a.startswith("ecapa_f16_hp=") and int(a.split("=")[1]) & 1 for a in sys.argv[1:])
    out["mfa"] = (d.read_ws("ec_mfa32", np.float32, n * 501 * 3072) if hp else d.read_ws("ec_mfa", dt, n * 501 * 3072)).reshape(n, 501, 3072).astype(np.float32)
    if not half:
        pass                                  # f32 mode: the logits overwrite cat -> the block outputs are gone; second run below
    out["cat"] = cat
    tr = d.read_ws("ec_tr", dt, n * 501 * 1920).reshape(n, 501, 1920).astype(np.float32)      # [t1 sub-bands 1..7 | t1 sub-band 0 | r1..r7]
    out["t1"] = np.concatenate([tr[:, :, 896:1024], tr[:, :, :896]], axis=2)
    out["r"] = tr[:, :, 896:]
    out["t2"] = d.read_ws("ec_t2", dt, n * 501 * 1024).reshape(n, 501, 1024).astype(np.float32)
    out["pooled"] = d.read_ws("ec_pooled", np.float32, n * 6144).reshape(n, 6144)
    out["se_g"] = d.read_ws("ec_se_g", np.float32, n * 1024).reshape(n, 1024)
    out["se_s"] = d.read_ws("ec_se_s", np.float32, n * 1024).reshape(n, 1024)
    return out
e32 = d.ecapa(f, l); a32 = grab(False)
d.set_option("ecapa_precision", 1)
for kv in sys.argv[1:]:
    k, v = kv.split("="); d.set_option(k, int(v))
e16 = d.ecapa(f, l); a16 = grab(True)
def rel(a, b, m=None):
    if m is not None: a, b = a[m], b[m]
    return np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())
print("items", n)
print("x0 (block0 out)      rel %.2e" % rel(a16["x0"], a32["x0"], valid))
for bk in range(3):
    print("x%d (block %d out)      rel %.2e" % (bk + 1, bk + 1, rel(a16["cat"][:, :, bk * 1024:(bk + 1) * 1024], a32["cat"][:, :, bk * 1024:(bk + 1) * 1024], valid)))
print("block 3: tdnn1 out   rel %.2e" % rel(a16["t1"], a32["t1"], valid))
for sb in range(8):
    print("block 3: res2net sub-band %d rel %.2e" % (sb, rel(a16["r"][:, :, sb * 128:(sb + 1) * 128], a32["r"][:, :, sb * 128:(sb + 1) * 128], valid)))
print("block 3: tdnn2 out   rel %.2e" % rel(a16["t2"], a32["t2"], valid))
print("block 3: SE mean rel %.2e  gate rel %.2e  gate min %.3f max %.3f" % (rel(a16["se_s"], a32["se_s"]), rel(a16["se_g"], a32["se_g"]), a32["se_g"].min(), a32["se_g"].max()))
gt = a32["se_g"][:, None, :] * a32["t2"]
x2_, x3_ = a32["cat"][:, :, 1024:2048], a32["cat"][:, :, 2048:3072]
print("block 3 norms over valid frames: |gate*t2| %.3e  |x2| %.3e  |x3| %.3e   |t2| %.3e" % (np.linalg.norm(gt[valid]), np.linalg.norm(x2_[valid]), np.linalg.norm(x3_[valid]), np.linalg.norm(a32["t2"][valid])))
x3_from16 = a16["se_g"][:, None, :] * a16["t2"] + a16["cat"][:, :, 1024:2048]
print("x3 recomputed in f32 from the fp16 run's gate, t2, x2: rel %.2e;  with the f32 gate instead: rel %.2e" % (rel(x3_from16, x3_, valid), rel(a32["se_g"][:, None, :] * a16["t2"] + a16["cat"][:, :, 1024:2048], x3_, valid)))
print("mfa out              rel %.2e" % rel(a16["mfa"], a32["mfa"], valid))
print("pooled mean          rel %.2e   pooled std rel %.2e" % (rel(a16["pooled"][:, :3072], a32["pooled"][:, :3072]), rel(a16["pooled"][:, 3072:], a32["pooled"][:, 3072:])))
cd = 1 - (e16.astype(np.float64) * e32).sum(1) / np.linalg.norm(e16.astype(np.float64), axis=1) / np.linalg.norm(e32.astype(np.float64), axis=1)
per_mfa = np.array([rel(a16["mfa"][i][valid[i]], a32["mfa"][i][valid[i]]) for i in range(n)])
per_pool = np.array([rel(a16["pooled"][i], a32["pooled"][i]) for i in range(n)])
print("embedding cos-dist   max %.2e median %.2e" % (cd.max(), np.median(cd)))
for i in np.argsort(cd)[-6:]:
    print("  item %d nvalid %d: mfa rel %.2e  pooled rel %.2e  cos-dist %.2e" % (sel[i], nv[i], per_mfa[i], per_pool[i], cd[i]))
print("  median item: mfa rel %.2e pooled rel %.2e" % (np.median(per_mfa), np.median(per_pool)))
