import sys, numpy as np, tempfile, time
sys.path.insert(0,'.'); sys.path.insert(0,'pyannote-audio_speaker-diarization_cpp_amd')
import sdhip, synth, weightpack as nn
tmp=tempfile.mkdtemp(); nn.save_pack(tmp+"/s.sdw", nn.synth_segmentation_weights(4321))
d=sdhip.Diarizer(tmp+"/s.sdw", None)
pcm=synth.make_pcm(600.0, seed=3); wav=pcm.astype(np.float32)/np.float32(32768)
d.set_option("seg_wide_ih",0); a=d.segment(wav); d.set_option("seg_wide_ih",1); b=d.segment(wav)
print("bit-identical scores:", np.array_equal(a,b), a.shape)
d.set_option("profile",2)
for v in (0,1,0,1):
    d.set_option("seg_wide_ih",v); d.reset_stats(); d.segment(wav); d.segment(wav)
    s=d.kernel_stats("conv_gemm:lstm_ih"); print("seg_wide_ih",v,"lstm_ih %.3f ms per pass, %.1f TF"%(s["ms"]/2, s["flops"]/s["ms"]/1e9))
