cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, struct
sys.path[:0]=['.','pyannote-audio_speaker-diarization_cpp_amd']
import synth, weightpack as wp, numpy as np
os.makedirs('/tmp/cc', exist_ok=True)
wp.save_pack('/tmp/cc/s.sdw', wp.synth_segmentation_weights(4321)); wp.save_pack('/tmp/cc/e.sdw', wp.synth_embedding_weights(4322))
for sec,name in ((600,'m10'),(3600,'h1')):
    pcm=synth.make_pcm(float(sec),seed=1234); data=pcm.tobytes()
    fmt=struct.pack("<HHIIHH",1,1,16000,32000,2,16)
    open('/tmp/cc/%s.wav'%name,'wb').write(b"RIFF"+struct.pack("<I",36+len(data))+b"WAVE"+b"fmt "+struct.pack("<I",16)+fmt+b"data"+struct.pack("<I",len(data))+data)
PY
E=pyannote-audio_speaker-diarization_cpp_amd/speakerDiarizer
echo "--- bare process (usage line)"; ( time $E ) 2>&1 | tail -4
echo "--- loader statistics"; LD_DEBUG=statistics $E 2>&1 | grep -i 'total startup\|relocation' | head -4
for f in m10 h1; do echo "--- $f"; ( time SD_TRACE_CREATE=1 $E /tmp/cc/s.sdw /tmp/cc/e.sdw /tmp/cc/$f.wav ) 2>&1 | grep -v 'Speaker_\|^---\|^$\|amdgpu.ids' | tail -14; done
