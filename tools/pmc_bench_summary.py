#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE counter CSVs of two `rocprofv3 --pmc ... -- python3 bench.py --steps 1 --warmup 0` passes into
profiles/pmc_conv_gemm_bench.json (what bench.py reports as roofline.traffic):
  tools/pmc_bench_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv"""
import csv, collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Kernel_Name'].startswith('void k_conv_gemm<'):
            k = r['Kernel_Name'].split('(')[0][5:]; d[k][0] += 1; d[k][1] += float(r['Counter_Value']) * 1024
    return d
f, w = agg(sys.argv[1]), agg(sys.argv[2])
n = sum(v[0] for v in f.values()); assert n == sum(v[0] for v in w.values())
fb = sum(v[1] for v in f.values()); wb = sum(v[1] for v in w.values())
out = {"bytes_per_launch": round((2 * fb + wb) / n), "launches": n,
       "fetch_raw_bytes_per_launch": round(fb / n), "fetch_x2_bytes_per_launch": round(2 * fb / n), "write_bytes_per_launch": round(wb / n),
       "per_kernel": {k: {"launches": f[k][0], "fetch_x2_bytes_per_launch": round(2 * f[k][1] / f[k][0]), "write_bytes_per_launch": round(w[k][1] / w[k][0])} for k in f},
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0` on MI355X; "
                 "Counter_Value is KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; Infinity-Cache hits are counted as "
                 "fetches); mean over all k_conv_gemm launches of one 1 h diarization"}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_conv_gemm_bench.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
