# GPU box: HBM-side traffic of the fp16 wide kernel over one planted hour in fp16 mode (two jobs): FETCH_SIZE and WRITE_SIZE in separate passes
# (Counter_Value is KB; FETCH_SIZE is doubled in the summary as MI355X_MICROARCH.md prescribes for gfx950)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_g256t; rm -rf $out; mkdir -p $out
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --output-format csv -d $out/$set -o p -- python3 tools/layer_profile.py planted 1 f16 > $out/$set.log 2> $out/$set.err
  f=$(find $out/$set -name "*counter_collection.csv" | head -1)
  echo "== $set"
  if [ -n "$f" ]; then python3 tools/pmc_kernel_fold.py $f k_conv_gemm_g256; else tail -3 $out/$set.err; fi
done
grep -E "conv_gemm:(tdnn1|tdnn2|mfa|block0)" $out/WRITE_SIZE.log
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
