#!/usr/bin/env python3
"""tools/pmc_conv_parse.py FETCH_counter_collection.csv WRITE_counter_collection.csv [items] -> table of HBM-side traffic per launch"""
import csv, sys
from pmc_conv import SHAPES
items = int(sys.argv[3]) if len(sys.argv) > 3 else 768
def rows(path):
    r = [x for x in csv.DictReader(open(path)) if x["Kernel_Name"].startswith("void k_conv_gemm<")]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return [float(x["Counter_Value"]) * 1024 for x in r]
f, w = rows(sys.argv[1]), rows(sys.argv[2])
assert len(f) == len(w) == 3 * len(SHAPES), (len(f), len(w))
M = items * 501
print("%-24s %12s %12s %12s %12s %12s" % ("shape", "alg read MB", "FETCH MB", "FETCHx2 MB", "alg write MB", "WRITE MB"))
for i, (name, cin, cout, kt, dil, x2) in enumerate(SHAPES):
    ar = 4.0 * (M * cin * (2 if x2 else 1) + cout * cin * kt); aw = 4.0 * M * cout
    print("%-24s %12.1f %12.1f %12.1f %12.1f %12.1f" % (name, ar / 1e6, f[3 * i + 2] / 1e6, 2 * f[3 * i + 2] / 1e6, aw / 1e6, w[3 * i + 2] / 1e6))
