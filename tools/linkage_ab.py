#!/usr/bin/env python3
"""A/B of the linkage kernels on ONE box in ONE process (VERDICT r05 #2a): the library as built (HEAD) and an older build loaded side by side
(tools/bin/libsdhip_<rev>.so, see tools/linkage_ab.sh), k_linkage_rg (linkage_kernel = 1) and k_linkage_mw (= 0), on the clustering input of the
bench's own planted hour (N = 12 989 rows: the embeddings of a real job, NaN rows dropped, rows normalised as the pipeline does), `reps`
interleaved rounds.  Prints every kernel time and the medians; Z must be the same array everywhere.
    python tools/linkage_ab.py [reps] [old library path ...]"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
olds = sys.argv[2:]


def load(tag, libpath):
    """a private copy of the ctypes binding bound to `libpath`"""
    if libpath:
        os.environ["SDHIP_LIB"] = libpath
    else:
        os.environ.pop("SDHIP_LIB", None)
    spec = importlib.util.spec_from_file_location("sdhip_" + tag, os.path.join(PKG, "sdhip.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


import torch, synth, weightpack as nn, tempfile
mods = [("HEAD", load("head", None))] + [(os.path.basename(p).replace("libsdhip_", "").replace(".so", ""), load("old%d" % i, os.path.abspath(p))) for i, p in enumerate(olds)]
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights(4321)); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights(4322))
# the bench's job, once, with HEAD: its embedding rows are the clustering input
sd = mods[0][1]
d0 = sd.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
pcm = synth.make_pcm(3600, seed=1234)
n = len(pcm)
dev = torch.device("cuda", 0)
d_pcm = torch.from_numpy(pcm).to(dev)
nc = synth.num_chunks(n)
sc, asg = synth.planted_scores(synth.with_duets(synth.schedule(3600, 1234)), n, 0, nc)
d_sc, d_pe = torch.from_numpy(sc).to(dev), torch.from_numpy(synth.planted_embeddings(asg)).to(dev)
d0.set_planted(d_sc.data_ptr(), d_pe.data_ptr(), 0, nc)
torch.cuda.synchronize()
turns = d0.diarize_dev(d_pcm.data_ptr(), n)
e = d0.read_ws("dz_emb", np.float32, nc * 3 * 192).reshape(-1, 192)
d0.close()
X = e[~np.isnan(e[:, 0])].astype(np.float64)
X /= np.linalg.norm(X, axis=1, keepdims=True)
N = len(X)
print("clustering input of the planted hour: N = %d rows (%d turns)" % (N, len(turns)), flush=True)
ds = []
for tag, m in mods:
    d = m.Diarizer(None, None)
    d.set_option("profile", 1)
    ds.append((tag, d))
res, Z0 = {}, None
for r in range(reps + 1):           # round 0 warms every context up
    for tag, d in ds:
        for kern, kname in ((1, "k_linkage_rg"), (0, "k_linkage_mw")):
            d.set_option("linkage_kernel", kern)
            d.reset_stats()
            Z = d.linkage(X)
            ms = d.kernel_stats("linkage")["ms"]
            if Z0 is None:
                Z0 = Z
            same = bool(np.array_equal(Z, Z0))
            if r > 0:
                res.setdefault((tag, kname), []).append(ms)
                print("round %d  %-10s %-13s %7.2f ms  same Z %s" % (r, tag, kname, ms, same), flush=True)
            assert same, (tag, kname)
print("medians over %d interleaved rounds, N = %d:" % (reps, N))
for (tag, kname), v in sorted(res.items()):
    v = sorted(v)
    print("  %-10s %-13s median %7.2f ms   min %7.2f   max %7.2f   (%.2f us per merge)" % (tag, kname, v[len(v) // 2], v[0], v[-1], v[len(v) // 2] * 1e3 / (N - 1)))
