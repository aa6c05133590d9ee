import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'pyannote-audio_speaker-diarization_cpp_amd'); sys.path.insert(0,'tests')
import sdhip, synth, weightpack as nn, tempfile
from test_planted import planted_case, nan_rule
tmp=tempfile.mkdtemp()
nn.save_pack(tmp+"/s.sdw", nn.synth_segmentation_weights()); nn.save_pack(tmp+"/e.sdw", nn.synth_embedding_weights())
d=sdhip.Diarizer(tmp+"/s.sdw", tmp+"/e.sdw")
pcm, scores, assign, embp = planted_case(600.0, 1234)
b, masks, counts, bad = nan_rule(scores)
wav = pcm.astype(np.float32)/np.float32(32768.0)
e32 = d.embed(wav, masks)
def cosd(e16):
    live=~bad
    a,c=e16[live].astype(np.float64), e32[live].astype(np.float64)
    return 1-(a*c).sum(1)/np.linalg.norm(a,axis=1)/np.linalg.norm(c,axis=1)
d.set_option("ecapa_precision",1)
e16=d.embed(wav,masks); cd=cosd(e16)
print("default fp16: max %.2e  q99 %.2e q90 %.2e median %.2e" % (cd.max(), np.quantile(cd,.99), np.quantile(cd,.9), np.median(cd)))
idx=np.where(~bad)[0]
worst=np.argsort(cd)[-8:]
for w in worst: print("item",idx[w],"count",counts[idx[w]],"cosd %.2e"%cd[w], "norm32 %.1f"%np.linalg.norm(e32[idx[w]]))
# correlation with count
for lo,hi in [(0,2000),(2000,10000),(10000,40000),(40000,80001)]:
    m=(counts[idx]>=lo)&(counts[idx]<hi)
    if m.any(): print("count [%d,%d): n=%d max %.2e median %.2e"%(lo,hi,m.sum(),cd[m].max(),np.median(cd[m])))
for hp in (1, 3):
    d.set_option("ecapa_f16_hp", hp); cdh = cosd(d.embed(wav, masks))
    print("ecapa_f16_hp=%d: max %.2e  q99 %.2e q90 %.2e median %.2e  items > 1e-3: %d" % (hp, cdh.max(), np.quantile(cdh, .99), np.quantile(cdh, .9), np.median(cdh), (cdh > 1e-3).sum()))
d.set_option("ecapa_f16_hp", 0)
d.set_option("ecapa_precision", 2); cd22 = cosd(d.embed(wav, masks)); d.set_option("ecapa_precision", 1)
print("ecapa_precision=2 (hi + lo weight planes): max %.2e  q99 %.2e q90 %.2e median %.2e  items > 1e-3: %d" % (cd22.max(), np.quantile(cd22, .99), np.quantile(cd22, .9), np.median(cd22), (cd22 > 1e-3).sum()))
print("default: items > 1e-3: %d of %d" % ((cd > 1e-3).sum(), len(cd)))
d.set_option("conv_h256",0); cd2=cosd(d.embed(wav,masks)); print("no h256: max %.2e"%cd2.max()); d.set_option("conv_h256",1)
d.set_option("skip_dead_rows",0); cd3=cosd(d.embed(wav,masks)); print("no skip: max %.2e"%cd3.max()); d.set_option("skip_dead_rows",1)
import torch
dev=torch.device("cuda",0)
d_pcm=torch.from_numpy(pcm).to(dev); d_sc=torch.from_numpy(scores).to(dev); torch.cuda.synchronize()
d.set_planted(d_sc.data_ptr(),0,0,scores.shape[0])
t16=d.diarize_dev(d_pcm.data_ptr(),len(pcm))
d.set_option("ecapa_precision",0)
t32=d.diarize_dev(d_pcm.data_ptr(),len(pcm))
print("real embeddings, planted scores: turns16 == turns32:", t16==t32, len(t16), len(t32), sorted({x[2] for x in t32}))
rel=np.linalg.norm(e16[~bad].astype(np.float64)-e32[~bad].astype(np.float64),axis=1)/np.linalg.norm(e32[~bad].astype(np.float64),axis=1)
print("rel L2: max %.2e q99 %.2e median %.2e"%(rel.max(),np.quantile(rel,.99),np.median(rel)))
