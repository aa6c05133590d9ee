cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  rm -rf /tmp/pmc_o; rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_o -o p --output-format csv -- python3 $R/tools/layer_profile.py planted 1 f16 > /dev/null 2>&1
  f=$(find /tmp/pmc_o -name '*counter_collection.csv' | head -1)
  echo "== $C ($f)"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "conv_gemm" in k: print(k, {c: "%.3g" % v for c, v in agg[k].items()}, {c: cnt[(k, c)] for c in agg[k]})
PY
done
