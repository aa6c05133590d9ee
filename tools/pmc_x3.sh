cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export OPTS=ecapa_precision=3
out=gpurun_out/pmc_x3; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 tools/layer_profile.py planted 1 f32 > $out/p$i.log 2> $out/p$i.err
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  echo "== $set"
  if [ -n "$f" ]; then python3 tools/pmc_kernel_fold.py $f k_conv_gemm_w256; else tail -3 $out/p$i.err; fi
done
rm -rf $out/p*/ 
