#!/bin/bash
# which kernels the vendor library picks for the shapes of tools/gemm_library_compare.py (names encode tile / staging choices), with their
# launch geometry, LDS and register counts from rocprofv3's kernel trace.  -> gpurun_out/gemm_library_kernels.txt
cd "$(dirname "$0")/.." && R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/libk; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/tools/gemm_library_compare.py > $O/run.txt 2>&1
python3 - "$O" <<'P' > $R/gpurun_out/gemm_library_kernels.txt
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "Cijk" not in n and "gemm" not in n.lower(): continue
    k = (n, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("LDS_Block_Size", r.get("Group_Segment_Size")), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size", r.get("Private_Segment_Size")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
for k, (c, t) in agg.items():
    print("%4d launches, %8.3f ms each  grid %s wg %s lds %s vgpr %s agpr %s sgpr %s scratch %s\n      %s" % (c, t / c, k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[0]))
P
cat $O/run.txt >> $R/gpurun_out/gemm_library_kernels.txt
