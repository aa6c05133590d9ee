#!/usr/bin/env python3
"""One launch set per ECAPA conv shape, for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/pmc_conv.py`.
Each shape is launched 3 times (2 warm + 1 timed, sd_bench_conv), in the order printed; tools/pmc_conv_parse.py groups the
counter rows of k_conv_gemm by that order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd"))
SHAPES = [("block0 96->1024 k5", 96, 1024, 5, 1, 0), ("tdnn 1024->1024", 1024, 1024, 1, 1, 0),
          ("res2net 128->128 k3 d2", 128, 128, 3, 2, 0), ("res2net +x2", 128, 128, 3, 2, 1),
          ("mfa 3072->3072", 3072, 3072, 1, 1, 0), ("asp_tdnn 3072->128", 3072, 128, 1, 1, 0), ("asp_conv 128->3072", 128, 3072, 1, 1, 0)]
if __name__ == "__main__":
    import sdhip
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    d = sdhip.Diarizer(None, None)
    for name, cin, cout, kt, dil, x2 in SHAPES:
        ms = d.bench_conv(items, 501, 501, cin, cout, kt, dil, x2, int(sys.argv[3]) if len(sys.argv) > 3 else 0, 1)
        print("%-26s %.2f ms" % (name, ms), flush=True)
