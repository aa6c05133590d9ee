#!/usr/bin/env python3
"""tools/isa_store_hazard_scan.py [file.hip ...]: compile the given HIP sources (default: every csrc/*.hip) to gfx950 assembly and report every 12- / 16-byte
`buffer_store` whose data registers are written by one of the next two instructions.  With an immediate soffset hipcc pads that pair itself; with an SGPR
soffset it does not (its hazard rule is gfx9's), and on gfx950 the store was seen to send the overwritten value (profiles/r06_px_attempt.txt).  A hit with
an SGPR soffset and distance +1 is a bug; put the column offset into the address VGPR / the immediate field, or fence `s_nop 1` behind the store
(PP_STORE_PAD in conv_gemm_p.hip).  Runs in the build container (no GPU): ~10 - 60 s per file."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
tmp = tempfile.mkdtemp(prefix="isa_")
bad = 0
for src in srcs:
    out = os.path.join(tmp, os.path.basename(src) + ".s")
    exact = ["-ffp-contract=off"] if os.path.basename(src) in ("postseg.hip", "cluster.hip", "reconstruct.hip", "linkage_rg.hip", "linkage_hx.hip") else []
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"),
                        "-S", "--cuda-device-only", "-o", out, src] + exact, capture_output=True, text=True)
    if r.returncode != 0:
        print(src, "did not compile:", r.stderr[-400:]); bad += 1; continue
    ins = [(i + 1, l.strip()) for i, l in enumerate(open(out)) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    hits = 0
    for k, (ln, l) in enumerate(ins):
        m = re.match(r"buffer_store_dwordx([34]) v\[(\d+):(\d+)\], (\S+), s\[\d+:\d+\], (\S+)", l)
        if not m:
            continue
        lo, hi, so = int(m.group(2)), int(m.group(3)), m.group(5)
        sgpr = not re.match(r"^(0x[0-9a-f]+|\d+)$", so) and so != "off"
        for d in (1, 2):
            if k + d >= len(ins):
                break
            n = ins[k + d][1]
            w = re.match(r"v_\S+ v\[(\d+):(\d+)\]", n) or re.match(r"v_\S+ v(\d+)\b", n)
            if not w or n.startswith("v_cmp"):
                continue
            w0 = int(w.group(1)); w1 = int(w.group(2)) if w.lastindex > 1 else w0
            if w0 <= hi and w1 >= lo and (sgpr or d == 1):
                print("%s:%d  %s | +%d: %s | soffset %s%s" % (os.path.basename(out), ln, l, d, n, so, "  <-- SGPR soffset: not padded by hipcc" if sgpr else ""))
                hits += 1
                if sgpr and d == 1:
                    bad += 1
    print(os.path.basename(src), "stores followed by a write of their data:", hits)
print("UNPADDED SGPR-soffset hazards:", bad)
sys.exit(1 if bad else 0)
