#!/bin/bash
# Run ON THE GPU BOX after `make -C pyannote-audio_speaker-diarization_cpp_amd libsdhip_pxabl1.so ... libsdhip_pxabl7.so`:
# time of the round-6 x3 kernel (k_conv_gemm_px, all its 32 launches of the 1 h job: stat conv_w256_x3) with parts of its K-step removed.  Results are
# garbage, so the mode's overflow check reruns the batches on the f32 kernels: the per-layer stats then hold both, conv_w256_x3 only the x3 launches
cd "$GRAFT_REPO_ROOT"
for v in "" ${PX_VARIANTS:-1 2 4 6 7}; do
  lib=pyannote-audio_speaker-diarization_cpp_amd/libsdhip.so; [ -n "$v" ] && lib=tools/bin/libsdhip_pxabl$v.so
  echo "== ${v:-product build} ($lib)"
  SDHIP_LIB=$PWD/$lib OPTS=ecapa_precision=3 python3 tools/layer_profile.py planted 1 f32 2>&1 | grep "conv_w256_x3\|x3_overflow"
done
echo "== previous x3 kernel (conv_pp=0)"
OPTS=ecapa_precision=3,conv_pp=0 python3 tools/layer_profile.py planted 1 f32 2>&1 | grep "conv_w256_x3\|x3_overflow"
