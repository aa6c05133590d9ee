#!/bin/bash
# tools/frontend_ablate.sh (GPU box): the front end kernel (k_stft_fbank) of the planted hour as built, without its dB scratch round trip,
# without its FFT arithmetic (make libsdhip_feabl1.so libsdhip_feabl2.so first).  Prints other_kernels.stft_mel of three short bench runs.
cd "$(dirname "$0")/.."
P=$PWD/pyannote-audio_speaker-diarization_cpp_amd
for lib in $P/libsdhip.so tools/bin/libsdhip_feabl1.so tools/bin/libsdhip_feabl2.so; do
  echo "== $lib"
  SDHIP_LIB=$(realpath $lib) python bench.py --steps 2 --warmup 1 --fp16-steps 0 --x3-steps 0 --cpu-seconds 0 --ref-finalize 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(json.dumps(j['other_kernels']['stft_mel'])); print('ms_per_step', j['ms_per_step'])"
done
