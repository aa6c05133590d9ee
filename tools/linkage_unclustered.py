import sys, numpy as np, time
sys.path.insert(0,'.'); sys.path.insert(0,'pyannote-audio_speaker-diarization_cpp_amd')
import sdhip
from oracle import orc
N=int(sys.argv[1]) if len(sys.argv)>1 else 32000
X=np.random.default_rng(3).random((N,3))
t=time.time(); _,Zr=orc.ahc(X, orc.THRESH_F32); print("oracle %.1f s"%(time.time()-t), flush=True)
d=sdhip.Diarizer(None,None); d.set_option("profile",1)
for sq in (1,0):
    d.set_option("linkage_square",sq); d.reset_stats(); Z=d.linkage(X)
    print("square",sq,"N",N,"equal",np.array_equal(Z,Zr),"linkage %.1f ms"%d.kernel_stats("linkage")["ms"],"retry rounds",d.kernel_stats("linkage_retry_rounds")["flops"],"fallbacks",d.kernel_stats("linkage_fallbacks")["launches"], flush=True)
