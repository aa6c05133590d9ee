import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'pyannote-audio_speaker-diarization_cpp_amd'); sys.path.insert(0,'tests')
import sdhip, tempfile
from oracle import nn_oracle as nn
from test_planted import planted_case, nan_rule
tmp=tempfile.mkdtemp(); we=nn.synth_embedding_weights(); nn.save_pack(tmp+"/e.sdw", we)
d=sdhip.Diarizer(None, tmp+"/e.sdw")
pcm, scores, assign, embp = planted_case(600.0, 1234)
b, masks, counts, bad = nan_rule(scores)
wav = pcm.astype(np.float32)/np.float32(32768.0)
feats, lens = d.frontend(wav, masks)
def rounded(w):
    out=dict(w)
    for k,v in w.items():
        if k.endswith("conv.weight") and ".se." not in k and not k.startswith("fc"): out[k]=np.asarray(v,np.float32).astype(np.float16).astype(np.float32)
    return out
wq=rounded(we)
for w0 in (2509, 3240, 3315):
    b0=(w0//32)*32; idx=np.array([i for i in range(b0,b0+32) if not bad[i]])
    f,l=np.ascontiguousarray(feats[idx]),np.ascontiguousarray(lens[idx])
    e_exact=nn.EcapaOracle(we)(f,l).numpy().astype(np.float64); e_q=nn.EcapaOracle(wq)(f,l).numpy().astype(np.float64)
    d.set_option("ecapa_precision",0); e32=d.ecapa(f,l).astype(np.float64)
    d.set_option("ecapa_precision",1); e16=d.ecapa(f,l).astype(np.float64)
    d.set_option("ecapa_precision",2); e162=d.ecapa(f,l).astype(np.float64)
    cd=lambda a,b: 1-(a*b).sum(1)/np.linalg.norm(a,axis=1)/np.linalg.norm(b,axis=1)
    print("batch",b0,"n",len(idx))
    print("  hip f32  vs oracle exact      max %.2e"%cd(e32,e_exact).max())
    print("  oracle fp16-weights vs exact  max %.2e"%cd(e_q,e_exact).max())
    print("  hip fp16 vs oracle exact      max %.2e"%cd(e16,e_exact).max())
    print("  hip fp16 vs oracle fp16-w     max %.2e"%cd(e16,e_q).max())
    print("  hip mode2 vs oracle exact     max %.2e"%cd(e162,e_exact).max())
