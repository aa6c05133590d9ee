#!/bin/bash
# Run ON THE GPU BOX after `make -C pyannote-audio_speaker-diarization_cpp_amd libsdhip_x3abl1.so libsdhip_x3abl2.so libsdhip_x3abl3.so`:
# layer times of the x3 wide kernel with parts of its K-step removed (results are garbage, times are not) -- where the kernel's time goes
cd "$GRAFT_REPO_ROOT"
for v in "" 1 2 3; do
  lib=pyannote-audio_speaker-diarization_cpp_amd/libsdhip.so; [ -n "$v" ] && lib=tools/bin/libsdhip_x3abl$v.so
  echo "== ${v:-product build} ($lib)"
  SDHIP_LIB=$PWD/$lib OPTS=ecapa_precision=3 python3 tools/layer_profile.py planted 1 f32 2>&1 | grep "block0\|tdnn\|mfa \|asp_conv"
done
