// ref_wav_shim.cpp -- C-ABI driver around the REFERENCE's own frontend/wav.h (header-only WavReader) so that
// orc_read_wav in ../sd_oracle.c and sd_read_wav_f32 in the library can be checked against the real thing.
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes the reference header where it lies
// (-I/root/reference/pipeline/src/frontend).
#include <cstring>
#include "wav.h"

extern "C" {

// wav::WavReader::Open (wav.h:62-126).  Returns the number of values read (num_samples * channels) or -1; copies at
// most cap raw sample values (not yet divided by 32768, exactly what WavReader::data() holds) into out.
long ref_wav_read(const char* path, float* out, long cap, int* sample_rate, int* channels, int* bits)
{
    wav::WavReader r;
    if (!r.Open(path)) return -1;
    const long n = (long)r.num_samples() * r.num_channels();
    if (sample_rate) *sample_rate = r.sample_rate();
    if (channels) *channels = r.num_channels();
    if (bits) *bits = r.bits_per_sample();
    if (out) std::memcpy(out, r.data(), sizeof(float) * (size_t)(n < cap ? n : cap));
    return n;
}

}
