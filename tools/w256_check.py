import sys, numpy as np, tempfile
sys.path.insert(0,'.'); sys.path.insert(0,'pyannote-audio_speaker-diarization_cpp_amd')
import sdhip, weightpack as nn
tmp=tempfile.mkdtemp(); nn.save_pack(tmp+"/e.sdw", nn.synth_embedding_weights())
d=sdhip.Diarizer(None, tmp+"/e.sdw")
rng=np.random.default_rng(5)
lens=np.array([1.0,0.5,0.25,0.9,0.7,0.33,1.0,0.6,0.8,0.45],np.float32)
feats=(3.0*rng.standard_normal((len(lens),501,80))).astype(np.float32)
e0=d.ecapa(feats,lens); d.set_option("conv_w256_f32",1); e1=d.ecapa(feats,lens)
print("f32 wide tile bit-identical:", np.array_equal(e0,e1), np.isfinite(e1).all())
