#!/usr/bin/env python3
"""sd_finalize_dev (count, clustering, reconstruction, turns) at the 2 / 4 / 8 GPU gather sizes on ONE GPU, from synthetic
scores + planted-speaker embeddings: tools/finalize_scale.py [hours ...].  What rank 0 does after the all-gather."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
hours = [float(h) for h in sys.argv[1:]] or [8.0]
d = sdhip.Diarizer(None, None)
dev = torch.device("cuda", 0)
for h in hours:
    n = int(h * 3600 * 16000)
    C, _ = sdhip.num_chunks(n)
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    K = 6
    # per chunk: which planted speaker each of the 3 local slots carries (slot 2 mostly silent), scores high where active
    spk = torch.randint(0, K, (C, 3), generator=g)
    active = torch.rand((C, 3), generator=g) < torch.tensor([0.9, 0.5, 0.05])
    seg = torch.rand((C, 293, 3), generator=g) * 0.3
    on = (torch.rand((C, 293, 3), generator=g) < 0.7) & active[:, None, :]
    seg = torch.where(on, 0.6 + 0.4 * torch.rand((C, 293, 3), generator=g), seg).float()
    cen = torch.randn((K, 192), generator=g)
    emb = (cen[spk] + 0.6 * torch.randn((C, 3, 192), generator=g)).float()
    emb[~active] = float("nan")
    d_seg = seg.to(dev); d_emb = emb.reshape(C * 3, 192).to(dev)
    torch.cuda.synchronize()
    t0 = time.time()
    turns = d.finalize_dev(d_seg.data_ptr(), d_emb.data_ptr(), C, n)
    dt = time.time() - t0
    labs = sorted({t[2] for t in turns})
    ok = all(0.0 <= t[0] < t[1] <= h * 3600 + 1 for t in turns)
    print("%.0f h: chunks %d, live embeddings %d, finalize %.2f s, %d turns, labels %s, turns sane %s" %
          (h, C, int(active.sum()), dt, len(turns), labs, ok), flush=True)
    del d_seg, d_emb
