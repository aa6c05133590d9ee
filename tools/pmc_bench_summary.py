#!/usr/bin/env python3
"""Fold the counter CSVs of separate `rocprofv3 --pmc ... -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0` passes into
profiles/pmc_conv_gemm_bench.json (what bench.py quotes as roofline.traffic / mfma_utilisation_pmc -- only while the recording's
conv_gemm.hip blob and workload match the running build):
  tools/pmc_bench_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv [MFMA_counter_collection.csv] [workload]
Counter passes are separate because gfx950 cannot schedule these counters together (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, collections, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CONV_SOURCES = ("conv_gemm.hip", "conv_gemm_h.hip", "conv_narrow.hip")   # the MFMA convolution kernels the counters are folded over


def blob(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def agg(path, counter=None):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        kn = r['Kernel_Name']
        if kn.startswith('void k_conv_gemm<') or kn.startswith('void k_conv_gemm_w256<') or kn.startswith('void k_conv_narrow<') or kn.startswith('k_stft_fbank'):
            k = kn.split('(')[0]
            k = k[5:] if k.startswith('void ') else k
            d[k][r['Counter_Name']] += float(r['Counter_Value'])
            n[k].add(r['Dispatch_Id'])
    return d, {k: len(v) for k, v in n.items()}


fpath, wpath = sys.argv[1], sys.argv[2]
mpath = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3].endswith(".csv") else None
workload = sys.argv[-1] if not sys.argv[-1].endswith(".csv") else "planted"
f, fn = agg(fpath)
w, wn = agg(wpath)
# the front end is reported beside the dominant kernel, not folded into it
stft = None
if 'k_stft_fbank' in f and 'k_stft_fbank' in w:
    sf, sw = f.pop('k_stft_fbank'), w.pop('k_stft_fbank')
    sn = fn.pop('k_stft_fbank'); wn.pop('k_stft_fbank')
    stft = {"launches": sn, "fetch_x2_bytes_per_launch": round(2 * sf['FETCH_SIZE'] * 1024 / sn), "write_bytes_per_launch": round(sw['WRITE_SIZE'] * 1024 / sn),
            "note": "k_stft_fbank: algorithmic bytes are 481 492 B per live item (321 172 read + 160 320 written); the counters also see the per-workgroup dB scratch "
                    "(written once, read twice, L2 / Infinity-Cache resident) and count Infinity-Cache hits as fetches"}
n = sum(fn.values())
assert n == sum(wn.values()), (fn, wn)
fb = sum(v['FETCH_SIZE'] for v in f.values()) * 1024
wb = sum(v['WRITE_SIZE'] for v in w.values()) * 1024
out = {"conv_gemm_blob": "+".join(blob(os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd", "csrc", f)) for f in CONV_SOURCES), "workload": workload, "hours_per_gpu": 1.0,
       "bytes_per_launch": round((2 * fb + wb) / n), "launches": n,
       "fetch_raw_bytes_per_launch": round(fb / n), "fetch_x2_bytes_per_launch": round(2 * fb / n), "write_bytes_per_launch": round(wb / n),
       "per_kernel": {k: {"launches": fn[k], "fetch_x2_bytes_per_launch": round(2 * f[k]['FETCH_SIZE'] * 1024 / fn[k]),
                          "write_bytes_per_launch": round(w[k]['WRITE_SIZE'] * 1024 / wn[k])} for k in f},
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0` (%s workload) on MI355X; "
                 "Counter_Value is KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; Infinity-Cache hits are counted as "
                 "fetches); mean over all k_conv_gemm launches of the run (warm-up + 1 step)" % workload}
if stft:
    out["stft_fbank"] = stft
if mpath:
    m, mn = agg(mpath)
    mf = {}
    for k, v in m.items():
        act = v.get('GRBM_GUI_ACTIVE', 0.0)
        if act > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
            mf[k] = {"mfma_busy_frac_of_simd_cycles": round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (act / 8 * 256 * 4), 4), "launches": mn[k]}
    mf["source"] = "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of the same command; busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)"
    out["mfma"] = mf
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_conv_gemm_bench.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
