import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"), os.path.join(ROOT, "tests")]
import numpy as np, tempfile
import sdhip, synth, weightpack as wp
from test_planted import planted_case, nan_rule
tmp = tempfile.mkdtemp()
wp.save_pack(tmp + "/s.sdw", wp.synth_segmentation_weights(4321)); wp.save_pack(tmp + "/e.sdw", wp.synth_embedding_weights(4322))
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
pcm, scores, assign, emb_planted = planted_case(600.0, 1234)
b, masks, counts, bad = nan_rule(scores)
wav = pcm.astype(np.float32) / np.float32(32768.0)
for mode in (0, 3, 1):
    d.set_option("ecapa_precision", mode)
    res = {}
    for nb in (96, 768, 1536, 3072):
        d.set_option("emb_batch_items", nb)
        res[nb] = d.embed(wav, masks)
    for nb in (768, 1536, 3072):
        diff = ~np.all((res[nb] == res[96]) | (np.isnan(res[nb]) & np.isnan(res[96])), axis=1)
        idx = np.flatnonzero(diff)
        print("mode", mode, "batch", nb, "vs 96: rows differing", len(idx), idx[:10], "max abs diff", np.nanmax(np.abs(res[nb] - res[96])) if len(idx) else 0)
