cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r04f16; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --precision f16 --fp16-steps 0 --x3-steps 0 > $out/bench_under_trace.json 2> $out/trace.err
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp "$ks" gpurun_out/r04f16_kernel_stats.csv
head -14 gpurun_out/r04f16_kernel_stats.csv | cut -c1-170
python3 -c "
import json; d=json.load(open('gpurun_out/prof_r04f16/bench_under_trace.json')); print(d['ms_per_step'], d['value'], d['dtype'], d['roofline']['achieved'], d['roofline']['frac'])"
