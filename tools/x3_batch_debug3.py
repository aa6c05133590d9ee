"""debug: where the last block's tdnn1 / tdnn2 outputs of the wide x3 kernel differ from the previous kernel's: tools/x3_batch_debug3.py [conv_rot]"""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pyannote-audio_speaker-diarization_cpp_amd")
import sdhip, weightpack as nn, tempfile
tmp = tempfile.mkdtemp()
nn.save_pack(tmp + "/s.sdw", nn.synth_segmentation_weights(4321)); nn.save_pack(tmp + "/e.sdw", nn.synth_embedding_weights(4322))
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
rng = np.random.default_rng(41)
n = 2100
lens = np.full(n, 0.06, np.float32); lens[::7] = 0.2
feats = (3.0 * rng.standard_normal((n, 501, 80))).astype(np.float32)
rot = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d.set_option("ecapa_precision", 3); d.set_option("conv_rot", rot)
d.set_option("emb_batch_items", 3072)
C, LDT, LD = 1024, 1920, 1024
def grab():
    e = d.ecapa(feats, lens)
    ro = None
    M, M1 = 60000, 55000
    tr = d.read_ws("ec_tr", np.float32, M * LDT).reshape(M, LDT)
    t2 = d.read_ws("ec_t2", np.float32, M1 * LD).reshape(M1, LD)
    x0 = d.read_ws("ec_x0", np.float32, M * LD).reshape(M, LD)
    cat = d.read_ws("ec_cat", np.float32, M1 * 3 * LD).reshape(M1, 3 * LD)
    return e, tr, t2, (x0, cat)
d.set_option("conv_pp", 0); e0, tr0, t20, ro = grab()
d.set_option("conv_pp", 1)
def show(name, a, b):
    dm = np.abs(a - b) > 1e-4 * (1 + np.abs(b))
    rows = np.flatnonzero(dm.any(1))
    print(" ", name, "rows differing", len(rows))
    if len(rows) == 0: return
    # group rows into 256-row tiles
    for tile in np.unique(rows // 256)[:12]:
        rr = rows[rows // 256 == tile]
        cols = np.flatnonzero(dm[rr].any(0))
        print("    tile", int(tile), "xcd", int(tile) & 7, "j", int(tile) >> 3, "rows in tile", (rr % 256).tolist()[:24], "n", len(rr), "cols", len(cols), cols[:8].tolist(), "..", cols[-4:].tolist(),
              "max", float(np.abs(a - b)[rr].max()))
for k in range(6):
    e, tr, t2, (x0, cat) = grab()
    print("run", k, "bad items", np.flatnonzero(np.abs(e - e0).max(1) > 1e-4)[:20].tolist())
    show("block0", x0, ro[0])
    for b in range(3): show("block%d out" % (b + 1), cat[:, b * C:(b + 1) * C], ro[1][:, b * C:(b + 1) * C])
