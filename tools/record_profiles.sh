#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/record_profiles.sh <tag> [bench args]): the kernel table and the PMC passes the bench line's
# roofline fields refer to.  Every rocprofv3 invocation profiles `python3 bench.py ...` directly (no env / bash -c hop), the
# counter passes are separate from the trace pass and from each other.
tag=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > $out/bench_under_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2> $out/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o mfma -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2> $out/mfma.err
find $out -name "*.csv" | head -20
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1)
cp "$ks" gpurun_out/${tag}_kernel_stats.csv
python3 tools/pmc_bench_summary.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) $(find $out/mfma -name "*counter_collection.csv" | head -1) planted > gpurun_out/${tag}_pmc_summary.json
cp profiles/pmc_conv_gemm_bench.json gpurun_out/${tag}_pmc_conv_gemm_bench.json
head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
tail -c 1200 gpurun_out/${tag}_pmc_summary.json
