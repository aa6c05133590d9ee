// frontend.hip -- embedding front end on the GPU:
//   a7  mask nearest-interp + stream compaction + wav_lens   (sd.cpp:746-797, 2436-2510)
//   a8  STFT n_fft=400 hop=160 periodic-Hamming, center, zero pad, fp64 (sd.cpp:1977-2036)
//   a9h power -> 80-mel -> 10 log10 -> top-dB 80 -> masked mean-norm (threeModel.py:212-220, 333-369)
//
// The reference materialises imask[80000], the compacted signal[80000] and the
// [501][201][2] STFT tensor per item on the host; here an item never leaves the chip
// between the waveform read and the log-mel feature rows:
//   k_mask_prefix  : per item, exclusive scan of the 293 mask frames' sample counts
//   k_wav_lens     : per reference batch of 32 items: max_len, wav_len, too-short flags
//   k_compact_active : items that are NaN by the reference's rule are dropped before any arithmetic
//   k_stft_fbank   : one persistent workgroup per item: gather the compacted samples into LDS
//                    (inverse of the compaction via the prefix table), 400-point real DFT in fp64
//                    as a 16 x 25 mixed-radix FFT, power in fp32, sparse mel, dB, item maximum,
//                    then top-dB clamp, mean over the first round(len*501) frames, and the
//                    channels-last [96] feature rows of the frames the network needs (ecapa.hip)
#include "common.h"

#define PW_LD 209
#define MEL_MAX_NNZ 1536

__device__ __forceinline__ int frame_start(int f) { return (int)(((int64_t)SD_CHUNK * f + (SD_FRAMES - 1)) / SD_FRAMES); }

// ---------------------------------------------------------------- k_mask_prefix
// prefix[item][f] = number of selected samples before mask frame f (f = 0..293)
__global__ void k_mask_prefix(const float* __restrict__ masks, int* __restrict__ prefix, int* __restrict__ counts, int items)
{
    __shared__ int buf[2][512];
    const int item = blockIdx.x, tid = threadIdx.x;
    int v = 0;
    if (tid < SD_FRAMES) {
        const bool on = masks[(size_t)item * SD_FRAMES + tid] > 0.5f;       // sd.cpp:761 (threshold 0.5)
        v = on ? (frame_start(tid + 1) - frame_start(tid)) : 0;              // samples j with j*293/80000 == tid
    }
    buf[0][tid] = v;
    __syncthreads();
    int cur = 0;
    for (int off = 1; off < 512; off <<= 1) {
        int x = buf[cur][tid];
        if (tid >= off) x += buf[cur][tid - off];
        buf[cur ^ 1][tid] = x;
        cur ^= 1;
        __syncthreads();
    }
    // inclusive scan in buf[cur]; exclusive prefix
    if (tid <= SD_FRAMES) prefix[(size_t)item * 296 + tid] = (tid == 0) ? 0 : buf[cur][tid - 1];
    if (tid == 0) counts[item] = buf[cur][SD_FRAMES - 1];
}

// ---------------------------------------------------------------- k_wav_lens
// one wave per reference batch (32 consecutive items).  sd.cpp:2467-2510.
// flags: 1 = output row is NaN (too short, or whole batch below min_num_samples)
__global__ void k_wav_lens(const int* __restrict__ counts, int items, float* __restrict__ wav_lens,
                           int* __restrict__ nnorm, int* __restrict__ nvalid, int* __restrict__ flags)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const int item = g * SD_EMB_BATCH + lane;
    const bool in = lane < SD_EMB_BATCH && item < items;
    float cnt = in ? (float)counts[item] : 0.0f;
    float mx = cnt;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (!in) return;
    const float min_samples = 640.0f;                                       // sd.cpp:44
    const bool all_nan = mx < min_samples;                                   // sd.cpp:2479
    const bool too_short = cnt < min_samples;                                // sd.cpp:2501
    float len = too_short ? 1.0f : cnt / mx;                                 // sd.cpp:2503, 2508
    if (all_nan) len = 1.0f;
    wav_lens[item] = len;
    const float lt = len * (float)SD_T;                                      // float32 product as in torch
    nnorm[item] = (int)rintf(lt);                                            // torch.round (half to even), threeModel.py:358
    int nv = (int)ceilf(lt);                                                 // arange(L) < len*L
    if (nv > SD_T) nv = SD_T;
    if (nv < 1) nv = 1;
    nvalid[item] = nv;
    flags[item] = (all_nan || too_short) ? 1 : 0;
}

// ---------------------------------------------------------------- k_stft_fbank : the whole front end of one item in one workgroup
// gather (inverse of the reference's stream compaction) -> fp64 STFT as a mixed-radix FFT -> power -> mel -> dB, then -- same
// workgroup, second phase -- top-dB clamp with the item's maximum, mean over the first nnorm frames, compact feature rows.
//
// 400-point real DFT of a windowed frame, 400 = 16 x 25 (decimation in time over the residue r = n mod 16):
//   Y_r[k1]      = sum_{m<25} xw[16 m + r] W25^{m k1}              k1 = 0..12   (real input: Y_r[25-k1] = conj Y_r[k1])
//   X[k1+25 k2]  = sum_{r<16} (W400^{r k1} Y_r[k1]) W16^{r k2}      k2 = 0..15
// For k1 = 1..12 the 16 outputs are the bins k1 + 25 k2 (k2 <= 7) and, conjugated, 400 - (k1 + 25 k2) (k2 >= 8); k1 = 0 gives the
// bins 25 k2, k2 <= 8: all 201 bins, each once.  ~15 kFLOP per frame instead of 170 kFLOP for the DFT-as-GEMM it replaces.
//   stage A: lane = (frame f of 4, residue r): 25 samples from LDS (consecutive lanes = consecutive samples), window in fp64,
//            straight-line 25-point real DFT (300 FMAs), twiddle, 13 complex values to LDS
//   stage B: lane = (frame f of 4, k1 of 13): 16 complex values from LDS, radix-4 x 4 FFT in registers, power (f32, as the
//            reference casts the STFT to f32 before the ONNX graph squares it) to LDS
// 16 frames per tile (4 per wave).  One persistent workgroup walks the tiles of an item, then the items blockIdx.x, + gridDim.x, ...
// The dB values of the item in flight go to a per-workgroup scratch (160 KB, re-used item after item: it stays in L2 / the
// Infinity Cache and is never re-read by another kernel).
#include "dft25_gen.h"
// A/B builds (tools/frontend_ablate.sh; wrong results, timing only): SD_FE_ABLATE = 1: no dB scratch -- phase 1 does not store the tile's dB values and
// phase 2 reads none (what removing the per-workgroup scratch round trip could gain AT MOST); 2: no FFT arithmetic (stages A and B skipped: the
// memory traffic alone)
#ifndef SD_FE_ABLATE
#define SD_FE_ABLATE 0
#endif
#define SIGP 3072                          // padded LDS signal: sample s of the tile sits at s + 16 (s / 160)
#define MEL_LDS_NNZ 512
#define YK 17                              // stage A -> B exchange: Y[frame][k1][r], k1 stride padded to 17 complex (bank spread)
__global__ __launch_bounds__(256, 2) void k_stft_fbank(
    const float* __restrict__ wav /* wav[0] = sample `origin` of the recording */, int64_t origin, int64_t n, const int* __restrict__ prefix, const int* __restrict__ counts, int64_t first_item,
    const float* __restrict__ window, const double* __restrict__ twc, const double* __restrict__ twns,
    const float* __restrict__ mel_w, const int* __restrict__ mel_lo, const int* __restrict__ mel_cnt, const int* __restrict__ mel_off,
    int mel_nnz, const int* __restrict__ alist, const int* __restrict__ rowoff, const int* __restrict__ nnorm, int run_items,
    float* __restrict__ scratch /*[gridDim.x][501][80]*/, float* __restrict__ feats, int sig_mode)
{
    __shared__ __attribute__((aligned(16))) double2 Y[16 * 13 * YK];
    __shared__ __attribute__((aligned(16))) float sigpw[16 * PW_LD > SIGP ? 16 * PW_LD : SIGP];    // signal tile, then the power spectra of the tile
    __shared__ int pre[296];
    __shared__ int mlo[SD_NMELS], mcnt[SD_NMELS], moff[SD_NMELS];
    __shared__ float part[3][SD_NMELS], mean[SD_NMELS];
    __shared__ float s_wmax[4];
    __shared__ float win[400];
    __shared__ float mw[MEL_LDS_NNZ];                          // mel weights (402 non-zeros for the 80 x 201 triangular bank)
    float* const sig = sigpw;
    float* const pw = sigpw;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < SD_NMELS) { mlo[tid] = mel_lo[tid]; mcnt[tid] = mel_cnt[tid]; moff[tid] = mel_off[tid]; }
    for (int k = tid; k < mel_nnz && k < MEL_LDS_NNZ; k += 256) mw[k] = mel_w[k];
    const bool mel_lds = mel_nnz <= MEL_LDS_NNZ;
    // per-lane constants of stage A: the window at this lane's 25 samples and the twiddles W400^(r k1)
    const int fA = lane >> 4, r = lane & 15;
    for (int k = tid; k < 400; k += 256) win[k] = window[k];
    double tcr[13], tsr[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) { const int j = (r * k) % 400; tcr[k] = twc[j]; tsr[k] = twns[j]; }
    const int fB = lane / 13, k1 = lane - 13 * fB;                 // stage B role (lanes 52..63 idle)
    float* const db = scratch + (size_t)blockIdx.x * SD_T * SD_NMELS;

    for (int slot = blockIdx.x; slot < run_items; slot += gridDim.x) {
        const int item = alist ? alist[slot] : slot;
        const int64_t gitem = first_item + item;
        // sig_mode (sd_embed_signals = EmbeddingModel1::infer as declared, sd.cpp:1977): item i is row i of a [B][80000] signal matrix, already
        // compacted by the caller; the dB maximum runs over all 501 frames as the reference's does (nothing is known about the samples behind wav_lens)
        const int64_t chunk_start = sig_mode ? gitem * (int64_t)SD_CHUNK : (gitem / SD_SPEAKERS) * (int64_t)SD_HOP;     // crop(), sd.cpp:1643
        const int cnt = counts[item];
        const int row0 = rowoff[slot], need = rowoff[slot + 1] - row0;
        __syncthreads();                                       // previous item's phase 2 is done with pre / mean / db
        for (int k = tid; k < 296; k += 256) pre[k] = prefix[(size_t)item * 296 + k];
        float vmax = -INFINITY;
        const int t_end = sig_mode ? SD_T : need;
        const int64_t room = n - chunk_start;
        // the compacted samples [160 t0 - 200, +2800) of a tile: each thread a run of 11 consecutive ones.  The requests of tile t0 + 16 are
        // issued BEFORE tile t0 is computed and land in registers meanwhile (r05: a tile's gather -- a binary search in the prefix table, then
        // loads that miss the L2 for every chunk's newest tenth -- used to be waited for at the head of every tile; the kernel without any FFT
        // arithmetic still took 5.7 of its 7.4 ms, profiles/r05_frontend_ablation_a.txt)
        float vn[11];
        auto gather_issue = [&](int t0) {
            const int mstart = 160 * t0 - 200;
            const int s0 = tid * 11;
            int f = -1, fend = 0, fbeg = 0, fsrc = 0;     // current mask frame: compacted range [fbeg, fend), first source sample
            int src[11];                                  // source sample relative to the chunk start, -1 = zero
#pragma unroll
            for (int q = 0; q < 11; ++q) {                // addresses first ...
                const int m = mstart + s0 + q;
                src[q] = -1;
                if (s0 + q < 2800 && m >= 0 && m < cnt) {
                    if (f < 0 || m >= fend) {             // smallest f with pre[f+1] > m (binary search once per run, then walk)
                        int lo = 0, hi = SD_FRAMES - 1;
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (pre[mid + 1] > m) hi = mid; else lo = mid + 1; }
                        f = lo; fbeg = pre[f]; fend = pre[f + 1]; fsrc = frame_start(f);
                    }
                    const int sp = fsrc + (m - fbeg);
                    if ((int64_t)sp < room) src[q] = sp;
                }
            }
#pragma unroll
            for (int q = 0; q < 11; ++q) vn[q] = src[q] >= 0 ? wav[chunk_start - origin + src[q]] : 0.0f;      // ... then all loads in flight together
        };
        __syncthreads();                                   // pre is loaded
        gather_issue(0);
        for (int t0 = 0; t0 < t_end; t0 += 16) {
            __syncthreads();                               // the previous tile's mel pass is done with pw
            {
                const int s0 = tid * 11;
#pragma unroll
                for (int q = 0; q < 11; ++q) { const int sidx = s0 + q; if (sidx < 2800) sig[sidx + 16 * (sidx / 160)] = vn[q]; }
            }
            if (t0 + 16 < t_end) gather_issue(t0 + 16);
            __syncthreads();
            // ---- stage A
            const int flA = 4 * w + fA;
            if (SD_FE_ABLATE != 2 && t0 + 4 * w < SD_T) {
                double x[25], yr[13], yi[13];
                const float* sp = sig + 176 * flA + r;
#pragma unroll
                for (int m = 0; m < 25; ++m) x[m] = (double)sp[16 * m + 16 * (m / 10)] * (double)win[16 * m + r];
                dft25_real(x, yr, yi);
                double2* yo = Y + (size_t)flA * 13 * YK + r;
#pragma unroll
                for (int k = 0; k < 13; ++k) yo[k * YK] = make_double2(yr[k] * tcr[k] - yi[k] * tsr[k], yr[k] * tsr[k] + yi[k] * tcr[k]);
            }
            __syncthreads();
            // ---- stage B (pw aliases sig: every wave is past its stage A reads)
            if (SD_FE_ABLATE != 2 && lane < 52 && t0 + 4 * w < SD_T) {
                const int flB = 4 * w + fB;
                double vr[16], vi[16];
                const double2* yi_ = Y + ((size_t)flB * 13 + k1) * YK;
#pragma unroll
                for (int q = 0; q < 16; ++q) { const double2 v = yi_[q]; vr[q] = v.x; vi[q] = v.y; }
                fft16(vr, vi);
                float* po = pw + flB * PW_LD;
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {
                    if (k1 == 0 && k2 > 8) continue;
                    const int k = k1 + 25 * k2;
                    const float fr = (float)vr[k2], fi = (float)vi[k2];            // STFT cast to f32 (sd.cpp:2031)
                    po[k <= 200 ? k : 400 - k] = __fadd_rn(__fmul_rn(fr, fr), __fmul_rn(fi, fi));
                }
            }
            __syncthreads();
            // ---- mel filterbank + dB for the tile's frames
            for (int o = tid; o < 16 * SD_NMELS; o += 256) {
                const int fr = o / SD_NMELS, m = o - fr * SD_NMELS;
                const int t = t0 + fr;
                if (t >= SD_T) continue;
                const float* pp = &pw[fr * PW_LD + mlo[m]];
                float acc = 0.0f;
                const int c = mcnt[m];
                if (mel_lds) { const float* qq = &mw[moff[m]]; for (int b = 0; b < c; ++b) acc = fmaf(pp[b], qq[b], acc); }
                else { const float* qq = &mel_w[moff[m]]; for (int b = 0; b < c; ++b) acc = fmaf(pp[b], qq[b], acc); }
                const float v = 10.0f * log10f(fmaxf(acc, 1e-10f));
#if SD_FE_ABLATE != 1
                db[(size_t)t * SD_NMELS + m] = v;
#endif
                vmax = fmaxf(vmax, v);
            }
        }
        // ---- phase 2: the item's maximum (frames beyond `need` are all-zero signal: -100 dB, never the maximum), clamp, mean, write
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0) s_wmax[w] = vmax;
        __syncthreads();
        const float floor_db = fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3])) - 80.0f;       // top_db = 80
        const int nn = nnorm[slot];
        if (tid < 240) {
            const int c = tid % SD_NMELS, g = tid / SD_NMELS;
            float sum = 0.0f;
            int t = g;
#if SD_FE_ABLATE == 1
            t = nn; sum = floor_db * (float)nn;
#endif
            for (; t + 9 < nn; t += 12) {                       // 4 loads in flight; the sum keeps its order
                const float v0 = db[(size_t)t * SD_NMELS + c], v1 = db[(size_t)(t + 3) * SD_NMELS + c], v2 = db[(size_t)(t + 6) * SD_NMELS + c], v3 = db[(size_t)(t + 9) * SD_NMELS + c];
                sum += fmaxf(v0, floor_db); sum += fmaxf(v1, floor_db); sum += fmaxf(v2, floor_db); sum += fmaxf(v3, floor_db);
            }
            for (; t < nn; t += 3) sum += fmaxf(db[(size_t)t * SD_NMELS + c], floor_db);
            part[g][c] = sum;
        }
        __syncthreads();
        if (tid < SD_NMELS) mean[tid] = (part[0][tid] + part[1][tid] + part[2][tid]) / (float)nn;
        __syncthreads();
        float* dst = feats + (size_t)row0 * SD_FEAT_LD;
        for (int idx = tid; idx < need * SD_FEAT_LD; idx += 256) {
            const int t = idx / SD_FEAT_LD, c = idx - t * SD_FEAT_LD;
            float v = 0.0f;
#if SD_FE_ABLATE == 1
            if (c < SD_NMELS) v = floor_db - mean[c];
#else
            if (c < SD_NMELS) v = fmaxf(db[(size_t)t * SD_NMELS + c], floor_db) - mean[c];
#endif
            dst[idx] = v;
        }
    }
}

// items whose output row is NaN anyway (too short / whole batch too short, sd.cpp:2479-2549) are dropped
// before the STFT and the network: alist[a] = item, cidx[item] = a or -1, compacted nnorm / nvalid
__global__ __launch_bounds__(1024) void k_compact_active(const int* __restrict__ flags, int items, int* __restrict__ alist, int* __restrict__ cidx,
                                                         const int* __restrict__ nnorm, const int* __restrict__ nvalid,
                                                         int* __restrict__ nnorm_c, int* __restrict__ nvalid_c, int* __restrict__ n_active)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < items; i0 += 1024) {
        const int i = i0 + tid;
        const int act = (i < items && flags[i] == 0) ? 1 : 0;
        int incl = act;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int off = base;
        for (int k = 0; k < w; ++k) off += wsum[k];
        const int pos = off + incl - act;
        if (i < items) {
            cidx[i] = act ? pos : -1;
            if (act) { alist[pos] = i; nnorm_c[pos] = nnorm[i]; nvalid_c[pos] = nvalid[i]; }
        }
        __syncthreads();
        if (tid == 1023) base = off + incl;
        __syncthreads();
    }
    if (tid == 0) *n_active = base;
}

// Phase A: mask prefix tables, wav_lens / nnorm / nvalid / too-short flags.  compact = true: items whose embedding is NaN by
// rule are dropped -- *h_n_active receives the number of live items, d_nnorm / d_nvalid are indexed by the compact position,
// d_cidx[item] = compact position or -1.  The prefix tables, sample counts and the active list stay in workspaces for phase B.
int frontend_prepare(sd_ctx* c, const float* d_masks, int64_t items, int64_t first_item, float* d_wav_lens, int* d_nnorm, int* d_nvalid,
                     int* d_flags, bool compact, int* h_n_active, int* d_cidx, std::vector<int>* h_nvalid)
{
    if (!c->ew.loaded) SD_FAIL(c, SD_ERR_MODEL, "embedding model not loaded");
    if (h_n_active) *h_n_active = (int)items;
    if (items <= 0) return SD_OK;
    if (first_item % SD_EMB_BATCH != 0) SD_FAIL(c, SD_ERR_ARG, "first_item must be a multiple of 32 (reference batches)");
    if (c->ew.mel_nnz > MEL_MAX_NNZ) SD_FAIL(c, SD_ERR_MODEL, "mel filterbank too dense (%d non-zeros)", c->ew.mel_nnz);
    WS(c, int, d_prefix, "fe_prefix", items * 296);
    WS(c, int, d_counts, "fe_counts", items);
    hipLaunchKernelGGL(k_mask_prefix, dim3((unsigned)items), dim3(512), 0, c->stream, d_masks, d_prefix, d_counts, (int)items);
    KCHECK(c);
    if (compact) {
        WS(c, int, t_nnorm, "fe_nnorm_all", items);
        WS(c, int, t_nvalid, "fe_nvalid_all", items);
        WS(c, int, t_alist, "fe_alist", items);
        WS(c, int, t_nact, "fe_nact", 4);
        hipLaunchKernelGGL(k_wav_lens, dim3((unsigned)((items + 31) / 32)), dim3(64), 0, c->stream, d_counts, (int)items, d_wav_lens, t_nnorm, t_nvalid, d_flags);
        KCHECK(c);
        hipLaunchKernelGGL(k_compact_active, dim3(1), dim3(1024), 0, c->stream, d_flags, (int)items, t_alist, d_cidx, t_nnorm, t_nvalid, d_nnorm, d_nvalid, t_nact);
        KCHECK(c);
        int na = 0;
        HIPCHK(c, hipMemcpyAsync(&na, t_nact, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        // the compacted nvalid array travels with the count (one synchronisation for both: the host plans the row spaces from it)
        if (h_nvalid) { h_nvalid->assign((size_t)items, 0); HIPCHK(c, hipMemcpyAsync(h_nvalid->data(), d_nvalid, (size_t)items * sizeof(int), hipMemcpyDeviceToHost, c->stream)); }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (h_n_active) *h_n_active = na;
        if (h_nvalid) h_nvalid->resize((size_t)na);
    } else {
        hipLaunchKernelGGL(k_wav_lens, dim3((unsigned)((items + 31) / 32)), dim3(64), 0, c->stream, d_counts, (int)items, d_wav_lens, d_nnorm, d_nvalid, d_flags);
        KCHECK(c);
    }
    return SD_OK;
}

// Phase B: compaction gather + STFT + mel + dB, then top-dB clamp and mean normalisation into the compact feature rows
// d_feats [rowoff[run_items]][96] (item's frame t at row d_rowoff[item] + t; frames beyond an item's rows are not computed).
int frontend_features(sd_ctx* c, const float* d_wav, int64_t n, int64_t first_item, int64_t run_items, bool compact, const int* d_nnorm,
                      const int* d_rowoff, float* d_feats, bool sig_mode)
{
    if (run_items <= 0) return SD_OK;
    const EcapaWeights& E = c->ew;
    const int* d_prefix = c->ws["fe_prefix"].as<int>();
    const int* d_counts = c->ws["fe_counts"].as<int>();
    const int* alist = compact ? c->ws["fe_alist"].as<int>() : nullptr;
    if (!d_prefix || !d_counts || (compact && !alist)) SD_FAIL(c, SD_ERR_ARG, "frontend_features before frontend_prepare");
    int grid = 2 * c->num_cu;
    if (grid > run_items) grid = (int)run_items;
    WS(c, float, d_scratch, "fe_db", (size_t)grid * SD_T * SD_NMELS);
    {
        // algorithmic bytes (SURVEY 8d's per-item rule on what this launch really touches): the selected samples of every live item
        // + its 293 mask values read, its stored frames x 96 floats written; ~15 kFLOP fp64 per frame (FFT) + 32 kFLOP mel.  Callers
        // that know the launch's sample / frame totals leave them in the context (run_embed, profiling only); otherwise the
        // full-length figure of 8d is billed (321 172 B read + 160 320 B written per item).
        const double by = c->fe_bill_samples >= 0 ? (double)c->fe_bill_samples * 4.0 + (double)run_items * 293.0 * 4.0 + (double)c->fe_bill_frames * SD_FEAT_LD * 4.0
                                                  : (double)run_items * (321172.0 + 160320.0);
        const double fr = c->fe_bill_samples >= 0 ? (double)c->fe_bill_frames : (double)run_items * SD_T;
        ProfScope ps(c, "stft_mel", fr * (15000.0 + 201.0 * 80 * 2), by);
        hipLaunchKernelGGL(k_stft_fbank, dim3(grid), dim3(256), 0, c->stream, d_wav, c->wav_origin, n, d_prefix, d_counts, first_item, E.window, E.tw_cos, E.tw_nsin,
                           E.mel_w, E.mel_lo, E.mel_cnt, E.mel_off, E.mel_nnz, alist, d_rowoff, d_nnorm, (int)run_items, d_scratch, d_feats, sig_mode ? 1 : 0);
        KCHECK(c);
    }
    return SD_OK;
}

// sd_embed_signals: the rows are already compacted (Helper::padSequence, sd.cpp:2463), so the gather of k_stft_fbank is the identity:
// every mask frame selected, prefix[f] = first sample of frame f, 80000 samples per item
__global__ void k_identity_prefix(int* __restrict__ prefix, int* __restrict__ counts, int items)
{
    const int item = blockIdx.x, tid = threadIdx.x;
    if (tid <= SD_FRAMES) prefix[(size_t)item * 296 + tid] = frame_start(tid);
    if (tid == 0) counts[item] = SD_CHUNK;
}
int frontend_prepare_signals(sd_ctx* c, int64_t items)
{
    if (!c->ew.loaded) SD_FAIL(c, SD_ERR_MODEL, "embedding model not loaded");
    if (c->ew.mel_nnz > MEL_MAX_NNZ) SD_FAIL(c, SD_ERR_MODEL, "mel filterbank too dense (%d non-zeros)", c->ew.mel_nnz);
    WS(c, int, d_prefix, "fe_prefix", items * 296);
    WS(c, int, d_counts, "fe_counts", items);
    hipLaunchKernelGGL(k_identity_prefix, dim3((unsigned)items), dim3(512), 0, c->stream, d_prefix, d_counts, (int)items);
    KCHECK(c);
    return SD_OK;
}
