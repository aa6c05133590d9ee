#!/bin/sh
# make_glue_tu.sh REF OUT -- writes ONE translation unit to OUT (a scratch path outside the repo; oracle/Makefile
# uses a mktemp directory and deletes it after the compile) that consists of
#   * line ranges of the REFERENCE's pipeline/src/speakerDiarizer.cpp, UNEDITED, read where the file lies under REF:
#     every part of that file that does not touch onnxruntime / libtorch (the model classes do; nothing else does),
#   * one-line wrappers of ours (printed by the `echo`s below) that give a range a scope to stand in when the
#     reference has it inside a class that derives from OnnxModel or inside speakerDiarization(),
#   * #include of our C shim (ref_glue_shim.inc), which only CALLS what the ranges define.
# Nothing of the reference is written into the repo and no header / library / class of the reference is replaced by
# a stand-in: the ranges are compiled as they are, against the standard headers their own file includes (15-27).
# TEST INFRASTRUCTURE ONLY.
set -e
REF="$1"; OUT="$2"
SD="$REF/pipeline/src/speakerDiarizer.cpp"
HERE="$(cd "$(dirname "$0")" && pwd)"
# the ranges below are line numbers of THIS file; refuse anything else
want="3442"
have="$(wc -l < "$SD" | tr -d ' ')"
[ "$have" = "$want" ] || { echo "make_glue_tu.sh: $SD has $have lines, expected $want" >&2; exit 1; }
r() { sed -n "$1,$2p" "$SD"; }
{
  r 15 27                      # the file's own standard includes
  echo '#include <cassert>'
  echo '#include "clustering.h"'   # sd.cpp:29 (the reference header, where it lies: -I$REF/pipeline/src/clustering)
  r 39 1331                    # constants, debugWrite*, Helper, Segment, Annotation, SlidingWindow, PipelineHelper::aggregate
  # SegmentModel's ORT-free members (binarize_swf, binarize_ndarray, crop, speaker_count, trim) + its data members;
  # the class head (": public OnnxModel", ctor, infer, slide) is what needs onnxruntime and is left out
  echo 'class RefSegmentGlue {'
  r 1334 1342
  echo 'public:'
  r 1506 1783
  echo '};'
  r 2044 2432                  # class Cluster, frame constants
  # getEmbedding's wav_lens rule (sd.cpp:2466-2510) between the two Helper calls and em.infer()
  echo 'static std::vector<std::vector<double>> ref_wav_lens_block( const std::vector<std::vector<bool>>& imasks, size_t batch_size, std::vector<float>& out_lens, std::vector<bool>& out_short, int number = 0 ) {'
  r 2466 2510
  echo '  out_lens = wav_lens; out_short = too_short; return std::vector<std::vector<double>>(); }'
  r 2563 2935                  # crop_segment, to_diarization, max_segmentation_cluster, reconstruct, to_annotation
  # speakerDiarization(): min_num_frames + clean masks (3012-3022), the mask choice loop (3050-3082)
  echo 'static void ref_mask_choice_block( std::vector<std::vector<std::vector<double>>>& binarized, std::vector<std::vector<float>>& batchMasks ) {'
  echo '  std::vector<std::vector<float>> batchData; std::vector<float> chunkData;'
  r 3012 3022
  echo '  for( size_t i = 0; i < binarized.size(); ++i ) {'
  r 3050 3082
  echo '  } } }'
  # speakerDiarization(): inactive speakers (3166-3191)
  echo 'static void ref_inactive_block( const std::vector<std::vector<std::vector<double>>>& binarized, std::vector<std::vector<int>>& hard_clusters ) {'
  r 3166 3191
  echo '}'
  echo "#include \"$HERE/ref_glue_shim.inc\""
} > "$OUT"
