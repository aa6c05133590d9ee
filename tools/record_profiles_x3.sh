#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/record_profiles_x3.sh <tag>): kernel table and MFMA / LDS counters of the x3 mode (bench.py --precision x3).
# Every rocprofv3 invocation profiles `python3 bench.py ...` directly; counter passes are separate from the trace pass and from each other.
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_${tag}_x3
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 bench.py --precision x3 --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/${tag}_bench_1h_x3_under_trace.json 2> $out/trace.err
cp "$(find $out/trace -name "*kernel_stats.csv" | head -1)" gpurun_out/${tag}_bench_1h_x3_kernel_stats.csv
{
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$out/pmc_$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $d -o p -- python3 bench.py --precision x3 --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $d.err
  echo "== rocprofv3 --pmc $set -- python3 bench.py --precision x3 --steps 1 --warmup 1 --cpu-seconds 0"
  python3 tools/pmc_kernel_fold.py "$(find $d -name "*counter_collection.csv" | head -1)" k_conv_gemm
done
} > gpurun_out/${tag}_pmc_x3.txt
head -8 gpurun_out/${tag}_bench_1h_x3_kernel_stats.csv | cut -c1-150
cat gpurun_out/${tag}_pmc_x3.txt
rm -rf $out
