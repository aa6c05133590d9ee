#!/usr/bin/env python3
"""tools/pmc_generic_parse.py counter_collection.csv -> per conv shape (3rd launch of each), every counter in the file"""
import csv, sys, collections
from pmc_conv import SHAPES
by = collections.defaultdict(dict)
for x in csv.DictReader(open(sys.argv[1])):
    if x["Kernel_Name"].startswith("void k_conv_gemm<"):
        by[int(x["Dispatch_Id"])][x["Counter_Name"]] = float(x["Counter_Value"])
ids = sorted(by)
assert len(ids) == 3 * len(SHAPES), len(ids)
names = sorted(by[ids[0]])
print("%-24s " % "shape" + " ".join("%16s" % n for n in names))
for i, sh in enumerate(SHAPES):
    print("%-24s " % sh[0] + " ".join("%16.4g" % by[ids[3 * i + 2]][n] for n in names))
