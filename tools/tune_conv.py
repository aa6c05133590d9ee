#!/usr/bin/env python3
"""Time conv_gemm on the ECAPA layer shapes on the GPU: tools/tune_conv.py [items] [dbg,...]
(ablations dbg 1-3 need the library built with `make EXTRA=-DSD_CONV_ABLATIONS`)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
import sdhip
items = int(sys.argv[1]) if len(sys.argv) > 1 else 768
d = sdhip.Diarizer(None, None)
shapes = [("block0 96->1024 k5", 96, 1024, 5, 1, 0), ("tdnn 1024->1024", 1024, 1024, 1, 1, 0),
          ("res2net 128->128 k3 d2", 128, 128, 3, 2, 0), ("res2net +x2", 128, 128, 3, 2, 1),
          ("mfa 3072->3072", 3072, 3072, 1, 1, 0), ("asp_tdnn 3072->128", 3072, 128, 1, 1, 0), ("asp_conv 128->3072", 128, 3072, 1, 1, 0)]
for name, cin, cout, kt, dil, x2 in shapes:
    row = []
    for dbg in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0".split(","))]:
        ms = d.bench_conv(items, 501, 501, cin, cout, kt, dil, x2, dbg, 5)
        fl = 2.0 * items * 501 * cin * cout * kt
        row.append("d%-3d %6.2f ms %5.1f TF" % (dbg, ms, fl / ms / 1e9))
    print("%-26s %s" % (name, " | ".join(row)), flush=True)
