"""GPU box: the step dumps of the 1-min wav (sd_set_dump_dir) + the exact scores / embeddings they were made from, for
tools/run_reference_verifier.py (which runs the REFERENCE's verifyEveryStepResult.py, unchanged, in the build container)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")]
import numpy as np, tempfile
import sdhip, synth, weightpack as wp
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
tmp = tempfile.mkdtemp()
wp.save_pack(tmp + "/s.sdw", wp.synth_segmentation_weights(4321)); wp.save_pack(tmp + "/e.sdw", wp.synth_embedding_weights(4322))
d = sdhip.Diarizer(tmp + "/s.sdw", tmp + "/e.sdw", 0)
if len(sys.argv) > 2 and sys.argv[2] == "planted":
    import torch
    sec = 180.0
    pcm = synth.make_pcm(sec, seed=77); n = len(pcm); nc = synth.num_chunks(n)
    sc, asg = synth.planted_scores(synth.with_duets(synth.schedule(sec, 77)), n, 0, nc)
    pe = synth.planted_embeddings(asg, outlier_every=41)
    dev = torch.device("cuda", 0)
    d_pcm, d_sc, d_pe = torch.from_numpy(pcm).to(dev), torch.from_numpy(sc).to(dev), torch.from_numpy(pe).to(dev)
    torch.cuda.synchronize()
    d.set_planted(d_sc.data_ptr(), d_pe.data_ptr(), 0, nc)
    d.set_dump_dir(out, 1)
    turns = d.diarize_dev(d_pcm.data_ptr(), n)
else:
    pcm, sr, ch = sdhip.read_wav(os.path.join(ROOT, "tests", "golden", "multi-speaker_1min.wav"))
    n = len(pcm); nc = synth.num_chunks(n)
    d.set_dump_dir(out, 1)
    turns = d.diarize(pcm)
d.set_dump_dir(None)
np.save(os.path.join(out, "seg.npy"), d.read_ws("dz_seg", np.float32, nc * 293 * 3).reshape(nc, 293, 3))
np.save(os.path.join(out, "emb.npy"), d.read_ws("dz_emb", np.float32, nc * 3 * 192).reshape(nc * 3, 192))
np.save(os.path.join(out, "meta.npy"), np.array([n, nc, len(turns)], np.int64))
print("dumped %d files for %d chunks, %d turns" % (len([f for f in os.listdir(out) if f.startswith("cpp_")]), nc, len(turns)))
