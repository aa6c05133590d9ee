# GPU box: counters of the fp16 mode's ping-pong wide kernel (k_conv_gemm_pp, conv_gemm_p.hip) over one planted hour in fp16 mode (two jobs per pass), one
# rocprofv3 --pmc pass per set, FETCH_SIZE and WRITE_SIZE in passes of their own (Counter_Value is KB; FETCH_SIZE is doubled in the summary as
# MI355X_MICROARCH.md prescribes for gfx950).  -> profiles/r06_pmc_pp.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_pp; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 tools/layer_profile.py planted 1 f16 > $out/p$i.log 2> $out/p$i.err
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  echo "== $set"
  if [ -n "$f" ]; then python3 tools/pmc_kernel_fold.py $f k_conv_gemm_pp; else tail -3 $out/p$i.err; fi
done
grep -E "conv_gemm:(tdnn1|tdnn2|mfa|block0)" $out/p$i.log
rm -rf $out/p*/
