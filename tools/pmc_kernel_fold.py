#!/usr/bin/env python3
"""tools/pmc_kernel_fold.py counter_collection.csv [name-prefix ...] -> per kernel name: launches and the SUM of every counter in the file
(one rocprofv3 --pmc pass of any command), plus the derived LDS / MFMA utilisation when their counters are present"""
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
pref = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    k = k[5:] if k.startswith("void ") else k
    if pref and not any(k.startswith(p) for p in pref):
        continue
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
for k in sorted(d, key=lambda k: -d[k].get("GRBM_GUI_ACTIVE", 0)):
    c = d[k]
    line = "%-44s launches %5d " % (k[:44], len(n[k])) + " ".join("%s %.4g" % (a, b) for a, b in sorted(c.items()))
    g = c.get("GRBM_GUI_ACTIVE")
    if g and "SQ_LDS_IDX_ACTIVE" in c:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs by the tool: /8 = cycles; SQ counters are summed over all CUs
        line += " | LDS busy %.1f %% of CU cycles, bank-conflict cycles %.1f %%" % (100 * c["SQ_LDS_IDX_ACTIVE"] / (g / 8 * 256), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / (g / 8 * 256))
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        line += " | MFMA busy %.1f %% of SIMD cycles" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8 * 256 * 4))
    print(line)
