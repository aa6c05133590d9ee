# GPU box: counters of the fp16 mode's LDS-DMA kernel (k_conv_gemm_g256<1>) over one planted hour, one rocprofv3 --pmc pass per set
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_g256; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 tools/layer_profile.py planted 1 f16 > $out/p$i.log 2> $out/p$i.err
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  echo "== $set"
  if [ -n "$f" ]; then python3 tools/pmc_kernel_fold.py $f k_conv_gemm_g256; else tail -3 $out/p$i.err; fi
done
rm -rf $out/p*/
