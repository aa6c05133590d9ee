#!/bin/bash
# CPU-only sanitizer build of the host code that parses untrusted files (ONNX reader + weight-pack reader): AddressSanitizer + UBSan.
# (GPU sanitizers are not available on the pool; this is the CPU build the advisor's findings are checked with.)
#   tools/sanitize/build.sh <out-binary>;   <out-binary> file.onnx ...   -> one line per (file, kind); sanitizer reports go to stderr
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
src=$root/pyannote-audio_speaker-diarization_cpp_amd/csrc
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ \
    -I/opt/rocm/include -I$root/include -x c++ $src/onnx_reader.cpp $src/weights.cpp $here/onnx_convert_main.cpp -o "$1" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
