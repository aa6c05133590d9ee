"""Build container (where /root/reference exists): runs the REFERENCE's pipeline/script/verifyEveryStepResult.py, UNCHANGED, against the step
dumps this build wrote on the GPU box (tools/dump_for_verifier.py -> gpurun_out/<dir>).  The script compares /tmp/cpp_<item>.txt with
/tmp/py_<item>.txt; its "py" side normally comes from an instrumented pyannote install, which does not exist here -- it is stood in by the
reference's OWN C++ writing its WRITE_DATA dumps (oracle/_ref/libref_glue_dump.so) from the very scores and embeddings the GPU produced.
So: /tmp/cpp_* = this build (GPU), /tmp/py_* = the reference's code; the verdicts are the script's own.
  python tools/run_reference_verifier.py gpurun_out/r04_dumps > profiles/r04_reference_verifier.txt"""
import ctypes as C, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")]
import numpy as np
from oracle import orc
src = sys.argv[1]
script = "/root/reference/pipeline/script/verifyEveryStepResult.py"
seg, emb = np.load(os.path.join(src, "seg.npy")), np.load(os.path.join(src, "emb.npy"))
n, nc, nturns = [int(x) for x in np.load(os.path.join(src, "meta.npy"))]
for f in glob.glob("/tmp/cpp_*.txt") + glob.glob("/tmp/py_*.txt"):
    os.remove(f)
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_glue_dump.so"))
R.ref_finalize.restype = C.c_long
R.ref_finalize.argtypes = [orc.c_fp, C.c_long, C.c_int, C.c_int, orc.c_dp, C.c_int, C.c_long, C.POINTER(orc.Turn), C.c_long, C.POINTER(C.c_int)]
buf = (orc.Turn * (nc * 8 + 64))(); K = C.c_int(0)
nt = R.ref_finalize(np.ascontiguousarray(seg), nc, 293, 3, np.ascontiguousarray(emb.astype(np.float64)), 192, n, buf, len(buf), C.byref(K))
# getEmbedding's per-batch files (masks<n>, imasks<n>, wav_lens<n>): the reference's Helper::interpolate / padSequence / wav_lens rule on the batches
# speakerDiarization() forms (sd.cpp:3047-3107) from the same scores; the audio is regenerated here (seeded synthetic, or the golden 1-min wav)
R.ref_embedding_inputs.restype = C.c_int
R.ref_embedding_inputs.argtypes = [orc.c_fp, orc.c_fp, C.c_int, C.c_int, C.c_long, orc.c_fp, orc.c_fp, orc.c_bp]
R.ref_set_batch_number.argtypes = [C.c_int]
import synth, sdhip
if "planted" in src:
    pcm = synth.make_pcm(180.0, seed=77)
else:
    pcm = sdhip.read_wav(os.path.join(ROOT, "tests", "golden", "multi-speaker_1min.wav"))[0]
assert len(pcm) == n
wav = pcm.astype(np.float32) / np.float32(32768.0)
masks = orc.select_masks(orc.binarize(seg))
number = 0
for b0 in range(0, nc * 3, 32):
    items = list(range(b0, min(nc * 3, b0 + 32)))
    wavs = np.stack([orc.crop(wav, (i // 3) * 8000) for i in items])
    R.ref_set_batch_number(number)
    sig = np.zeros_like(wavs); lens = np.zeros(len(items), np.float32); ts = np.zeros(len(items), np.uint8)
    if not R.ref_embedding_inputs(np.ascontiguousarray(wavs), np.ascontiguousarray(masks[items]), len(items), 293, 80000, sig, lens, ts):
        number += 1                       # `number++` sits behind the early return of an all-too-short batch (sd.cpp:2479-2519)
for f in glob.glob("/tmp/cpp_imasks*.txt"):                # 15 MB each, written by this build only at dump level 2
    os.remove(f)
for f in glob.glob("/tmp/cpp_*.txt"):                      # the reference's own dumps play the script's "py" side
    shutil.move(f, f.replace("/tmp/cpp_", "/tmp/py_"))
mine = sorted(glob.glob(os.path.join(src, "cpp_*.txt")))
for f in mine:
    shutil.copy(f, "/tmp/" + os.path.basename(f))
print("# %s: %d chunks, %d samples; this build: %d turns, %d dump files; the reference's C++ on the same scores / embeddings: %d turns, K = %d, %d dump files"
      % (src, nc, n, nturns, len(mine), nt, K.value, len(glob.glob("/tmp/py_*.txt"))))
print("# running %s (unchanged)\n" % script)
out = subprocess.run([sys.executable, script], capture_output=True, text=True)
print(out.stdout)
bad = out.stdout.count("Difference is detected")
ok = out.stdout.count("Checking passed")
print("# summary: %d items passed, %d differ" % (ok, bad))
for f in glob.glob("/tmp/cpp_*.txt") + glob.glob("/tmp/py_*.txt"):
    os.remove(f)
sys.exit(1 if bad else 0)
