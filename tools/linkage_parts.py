#!/usr/bin/env python3
"""tools/linkage_parts.py [N] [G]: time per synthetic merge round of k_linkage_rg's geometry with its parts switched on one by one
(sd_bench_linkage_parts: 1 row loads, 2 arithmetic, 4 stores, 8 workgroup reductions, 16 slot exchange + digest)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12602
Gs = [int(v) for v in sys.argv[2:]] or [32]
d = sdhip.Diarizer(None, None)
L = sdhip.lib()
names = {1: "loads", 2: "math", 4: "stores", 8: "wave-reduce", 32: "lds-fold", 16: "exchange", 64: "no-digest", 128: "per-wave-slots", 256: "mirror-stores"}
for onex in ((1,) if os.environ.get('ONEX', '1') == '1' else (0,)):
    for G in Gs:
        for parts in (0, 2, 8, 32, 8 | 32, 1, 1 | 2 | 4, 1 | 2 | 4 | 8, 1 | 2 | 4 | 8 | 32, 1 | 2 | 4 | 8 | 32 | 16 | 64, 1 | 2 | 4 | 8 | 32 | 16, 1 | 2 | 4 | 8 | 16 | 128, 1 | 2 | 4 | 8 | 16 | 128 | 64, 1 | 2 | 4 | 8 | 32 | 16, 1 | 2 | 4 | 8 | 32 | 16 | 256, 1 | 2 | 4 | 256):
            us = C.c_double(0)
            rc = L.sd_bench_linkage_parts(d._h, N, G, 4000, parts, onex, C.byref(us))
            lab = "+".join(names[b] for b in (1, 2, 4, 8, 32, 16, 64, 128, 256) if parts & b)
            print("N=%d G=%d one_xcd=%d %-34s %s" % (N, G, onex, lab, "%.2f us/round" % us.value if rc == 0 else "rc=%d %s" % (rc, sdhip.lib().sd_last_error(d._h).decode())), flush=True)
