#!/bin/bash
# GPU box: stall breakdown of the front end kernel (k_stft_fbank, frontend.hip) over one planted hour (two jobs per pass, one launch per job), as built and
# without its FFT arithmetic (tools/bin/libsdhip_feabl2.so: `make -C pyannote-audio_speaker-diarization_cpp_amd libsdhip_feabl2.so` first) -- VERDICT r05 #5:
# where the 5.5 ms that remain without the FFT go.  One rocprofv3 --pmc pass per counter set.  -> profiles/r06_pmc_frontend.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_fe; rm -rf $out; mkdir -p $out
for lib in pyannote-audio_speaker-diarization_cpp_amd/libsdhip.so tools/bin/libsdhip_feabl2.so; do
  echo "#### $lib"
  export SDHIP_LIB=$PWD/$lib
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 tools/layer_profile.py planted 1 f16 > $out/p$i.log 2> $out/p$i.err
    f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
    echo "== $set"
    if [ -n "$f" ]; then python3 tools/pmc_kernel_fold.py $f k_stft_fbank; else tail -3 $out/p$i.err; fi
  done
  grep -E "stft_mel" $out/p$i.log
  rm -rf $out/p*/
done
