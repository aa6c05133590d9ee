// frontend.hip -- embedding front end on the GPU:
//   a7  mask nearest-interp + stream compaction + wav_lens   (sd.cpp:746-797, 2436-2510)
//   a8  STFT n_fft=400 hop=160 periodic-Hamming, center, zero pad, fp64 (sd.cpp:1977-2036)
//   a9h power -> 80-mel -> 10 log10 -> top-dB 80 -> masked mean-norm (threeModel.py:212-220, 333-369)
//
// The reference materialises imask[80000], the compacted signal[80000] and the
// [501][201][2] STFT tensor per item on the host; here an item never leaves the chip
// between the waveform read and the [501][80] log-mel write:
//   k_mask_prefix : per item, exclusive scan of the 293 mask frames' sample counts
//   k_wav_lens    : per reference batch of 32 items: max_len, wav_len, too-short flags
//   k_stft_mel    : per (item, 64-frame tile): gather the compacted samples into LDS
//                   (inverse of the compaction via the prefix table), 400-point real
//                   DFT as fp64 MFMA (v_mfma_f64_16x16x4_f64, even/odd folded to K = 204,
//                   twiddles from a 400-entry LDS table), power in fp32, sparse mel, dB,
//                   per-item max (atomic)
//   k_fbank_norm  : top-dB clamp, mean over the first round(len*501) frames, subtract,
//                   write channels-last [501][96] rows for the MFMA convs
#include "common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define FT 16                         // STFT frames per workgroup: one MFMA row tile; the 4 waves split the 13 bin tiles
#define SIG_LEN ((FT - 1) * 160 + 400)  // 2800 samples feed FT frames
#define SIG_LDS (SIG_LEN + 2 * (SIG_LEN / 160) + 8)
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

__device__ __forceinline__ void atomic_max_float(float* addr, float v)
{
    if (v >= 0.0f) atomicMax((int*)addr, __float_as_int(v));
    else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

template <int NT>
__device__ __forceinline__ void dft_tiles(const float* sig, const float* win, const double* tc, const double* ts,
                                          float* pw, int w, int lane, int tile0)
{
    const int i = lane & 15, kq = lane >> 4;
    const int fl = i;                      // every wave works on the workgroup's 16 frames, on its own bin tiles
    const int sbase = 162 * fl;            // padded LDS position of the frame's first sample
    f64x4 re[NT], im[NT];
    int idx[NT], inc[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        re[b] = (f64x4){0, 0, 0, 0};
        im[b] = (f64x4){0, 0, 0, 0};
        const int j = (tile0 + b) * 16 + i;
        idx[b] = (kq * j) % 400;
        inc[b] = (4 * j) % 400;
    }
    // Real-input symmetry halves the contraction: with xw[n] = x[n] w[n],
    //   Re X[k] =  sum_{n=0..200} e[n] cos(2 pi k n / 400),  e[n] = xw[n] + xw[400-n]  (e[0] = xw[0], e[200] = xw[200])
    //   Im X[k] = -sum_{n=1..199} o[n] sin(2 pi k n / 400),  o[n] = xw[n] - xw[400-n]
    // 51 K-steps of 4 instead of 100 (indices 201..203 contribute zeros).
    for (int s = 0; s < 51; ++s) {
        const int nn = 4 * s + kq;
        const int nc = nn <= 200 ? nn : 200;
        const int nm = 400 - nc;                               // mirror index (400 for nc == 0: not used)
        const double a0 = (double)sig[sbase + nc + 2 * (nc / 160)] * (double)win[nc];
        const int nmc = nm < 400 ? nm : 399;
        const double a1 = (double)sig[sbase + nmc + 2 * (nmc / 160)] * (double)win[nmc];
        const bool mid = (nn >= 1) && (nn <= 199);
        const double e = (nn > 200) ? 0.0 : (mid ? a0 + a1 : a0);
        const double o = mid ? a0 - a1 : 0.0;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const double cv = tc[idx[b]], sv = ts[idx[b]];
            re[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(e, cv, re[b], 0, 0, 0);
            im[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(o, sv, im[b], 0, 0, 0);
            idx[b] += inc[b];
            if (idx[b] >= 400) idx[b] -= 400;
        }
    }
    // C layout (f64 16x16x4): col = lane&15 (bin), row = (lane>>4) + 4*r (frame)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int bin = (tile0 + b) * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float fr = (float)re[b][r], fi = (float)im[b][r];        // STFT cast to f32 (sd.cpp:2031)
            pw[(kq + 4 * r) * PW_LD + bin] = __fadd_rn(__fmul_rn(fr, fr), __fmul_rn(fi, fi));
        }
    }
}

// ---------------------------------------------------------------- k_stft_mel
__global__ __launch_bounds__(256) void k_stft_mel(
    const float* __restrict__ wav, int64_t n, const int* __restrict__ prefix, const int* __restrict__ counts,
    int64_t first_item, const float* __restrict__ window, const double* __restrict__ twc, const double* __restrict__ twns,
    const float* __restrict__ mel_w, const int* __restrict__ mel_lo, const int* __restrict__ mel_cnt,
    const int* __restrict__ mel_off, int mel_nnz, float* __restrict__ db, float* __restrict__ item_max,
    const int* __restrict__ alist /* active-item list or null: outputs are indexed by the compact position */,
    const int* __restrict__ rowoff /* [slots + 1]: frames at or beyond rowoff[slot + 1] - rowoff[slot] are not needed (ecapa.hip) */)
{
    __shared__ double tc[400], ts[400];
    __shared__ float sig[SIG_LDS];
    __shared__ float win[400];
    __shared__ float pw[16 * PW_LD];
    __shared__ float mw[MEL_MAX_NNZ];
    __shared__ int pre[296];
    __shared__ int mlo[SD_NMELS], mcnt[SD_NMELS], moff[SD_NMELS];

    const int slot = blockIdx.y, t0 = blockIdx.x * FT;
    // frames the network never reads are all-zero signal (every frame with content lies below nvalid + 2): they can neither
    // raise the item's maximum nor enter the mean, so whole tiles of them are skipped
    if (t0 >= rowoff[slot + 1] - rowoff[slot]) return;
    const int item = alist ? alist[slot] : slot;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t gitem = first_item + item;
    const int64_t chunk_start = (gitem / SD_SPEAKERS) * (int64_t)SD_HOP;     // crop(), sd.cpp:1643
    const int cnt = counts[item];

    for (int k = tid; k < 400; k += 256) { tc[k] = twc[k]; ts[k] = twns[k]; win[k] = window[k]; }
    for (int k = tid; k < 296; k += 256) pre[k] = prefix[(size_t)item * 296 + k];
    for (int k = tid; k < mel_nnz; k += 256) mw[k] = mel_w[k];
    if (tid < SD_NMELS) { mlo[tid] = mel_lo[tid]; mcnt[tid] = mel_cnt[tid]; moff[tid] = mel_off[tid]; }
    __syncthreads();

    // gather the compacted signal samples m in [160*t0-200, +SIG_LEN) (zeros outside [0,cnt))
    const int mstart = 160 * t0 - 200;
    for (int mm = tid; mm < SIG_LEN; mm += 256) {
        const int m = mstart + mm;
        float v = 0.0f;
        if (m >= 0 && m < cnt) {
            // smallest f with pre[f+1] > m  (f is an active mask frame containing compacted sample m)
            int lo = 0, hi = SD_FRAMES - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (pre[mid + 1] > m) hi = mid; else lo = mid + 1; }
            const int64_t s = chunk_start + frame_start(lo) + (m - pre[lo]);
            if (s < n) v = wav[s];
        }
        sig[mm + 2 * (mm / 160)] = v;
    }
    __syncthreads();

    // 13 bin tiles of 16: wave 0 takes tiles 0-3, waves 1-3 three each (4 waves per 16 frames keep 3 workgroups = 12 waves
    // resident per CU; with 64 frames per workgroup the LDS footprint allowed one wave per SIMD)
    if (w == 0) dft_tiles<4>(sig, win, tc, ts, pw, w, lane, 0);
    else dft_tiles<3>(sig, win, tc, ts, pw, w, lane, 1 + 3 * w);
    __syncthreads();

    float vmax = -INFINITY;
    for (int o = tid; o < 16 * SD_NMELS; o += 256) {
        const int fr = o / SD_NMELS, m = o - fr * SD_NMELS;
        const float* p = &pw[fr * PW_LD + mlo[m]];
        const float* q = &mw[moff[m]];
        float acc = 0.0f;
        const int c = mcnt[m];
        for (int b = 0; b < c; ++b) acc = fmaf(p[b], q[b], acc);
        const float v = 10.0f * log10f(fmaxf(acc, 1e-10f));
        const int t = t0 + fr;
        if (t < SD_T) {
            db[((size_t)slot * SD_T + t) * SD_NMELS + m] = v;
            vmax = fmaxf(vmax, v);
        }
    }
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0 && vmax > -INFINITY) atomic_max_float(&item_max[slot], vmax);
}

// ---------------------------------------------------------------- k_fbank_norm
// feats: compact rows (see ecapa.hip): item's frame t goes to row rowoff[item] + t, t < rowoff[item + 1] - rowoff[item]
__global__ __launch_bounds__(256) void k_fbank_norm(const float* __restrict__ db, const float* __restrict__ item_max,
                                                    const int* __restrict__ nnorm, const int* __restrict__ rowoff, float* __restrict__ feats)
{
    __shared__ float part[3][SD_NMELS];
    __shared__ float mean[SD_NMELS];
    const int item = blockIdx.x, tid = threadIdx.x;
    const float floor_db = item_max[item] - 80.0f;                           // top_db = 80
    const int nn = nnorm[item];
    const float* src = db + (size_t)item * SD_T * SD_NMELS;
    if (tid < 240) {
        const int c = tid % SD_NMELS, g = tid / SD_NMELS;
        float s = 0.0f;
        for (int t = g; t < nn; t += 3) s += fmaxf(src[t * SD_NMELS + c], floor_db);
        part[g][c] = s;
    }
    __syncthreads();
    if (tid < SD_NMELS) mean[tid] = (part[0][tid] + part[1][tid] + part[2][tid]) / (float)nn;
    __syncthreads();
    const int r0 = rowoff[item], need = rowoff[item + 1] - r0;
    float* dst = feats + (size_t)r0 * SD_FEAT_LD;
    for (int idx = tid; idx < need * SD_FEAT_LD; idx += 256) {
        const int t = idx / SD_FEAT_LD, c = idx - t * SD_FEAT_LD;
        float v = 0.0f;
        if (c < SD_NMELS) v = fmaxf(src[t * SD_NMELS + c], floor_db) - mean[c];
        dst[idx] = v;
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

__global__ void k_fill_f32(float* p, float v, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Phase A: mask prefix tables, wav_lens / nnorm / nvalid / too-short flags.  compact = true: items whose embedding is NaN by
// rule are dropped -- *h_n_active receives the number of live items, d_nnorm / d_nvalid are indexed by the compact position,
// d_cidx[item] = compact position or -1.  The prefix tables, sample counts and the active list stay in workspaces for phase B.
int frontend_prepare(sd_ctx* c, const float* d_masks, int64_t items, int64_t first_item, float* d_wav_lens, int* d_nnorm, int* d_nvalid,
                     int* d_flags, bool compact, int* h_n_active, int* d_cidx)
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
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (h_n_active) *h_n_active = na;
    } else {
        hipLaunchKernelGGL(k_wav_lens, dim3((unsigned)((items + 31) / 32)), dim3(64), 0, c->stream, d_counts, (int)items, d_wav_lens, d_nnorm, d_nvalid, d_flags);
        KCHECK(c);
    }
    return SD_OK;
}

// Phase B: compaction gather + STFT + mel + dB, then top-dB clamp and mean normalisation into the compact feature rows
// d_feats [rowoff[run_items]][96] (item's frame t at row d_rowoff[item] + t; frames beyond an item's rows are not computed).
int frontend_features(sd_ctx* c, const float* d_wav, int64_t n, int64_t first_item, int64_t run_items, bool compact, const int* d_nnorm,
                      const int* d_rowoff, float* d_feats)
{
    if (run_items <= 0) return SD_OK;
    const EcapaWeights& E = c->ew;
    const int* d_prefix = c->ws["fe_prefix"].as<int>();
    const int* d_counts = c->ws["fe_counts"].as<int>();
    const int* alist = compact ? c->ws["fe_alist"].as<int>() : nullptr;
    if (!d_prefix || !d_counts || (compact && !alist)) SD_FAIL(c, SD_ERR_ARG, "frontend_features before frontend_prepare");
    WS(c, float, d_db, "fe_db", run_items * SD_T * SD_NMELS);
    WS(c, float, d_max, "fe_max", run_items);
    hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((run_items + 255) / 256)), dim3(256), 0, c->stream, d_max, -INFINITY, run_items);
    KCHECK(c);
    {
        // algorithmic bytes per item: 80000*4 + 293*4 read, 501*80*4 written
        ProfScope ps(c, "stft_mel", (double)run_items * (SD_TP * 208.0 * 204 * 2 * 2 + SD_T * 201.0 * 80 * 2), (double)run_items * (321172.0 + 160320.0));
        hipLaunchKernelGGL(k_stft_mel, dim3((SD_T + FT - 1) / FT, (unsigned)run_items), dim3(256), 0, c->stream, d_wav, n, d_prefix, d_counts,
                           first_item, E.window, E.tw_cos, E.tw_nsin, E.mel_w, E.mel_lo, E.mel_cnt, E.mel_off, E.mel_nnz, d_db, d_max, alist, d_rowoff);
        KCHECK(c);
    }
    {
        ProfScope ps(c, "fbank_norm", 0, (double)run_items * (160320.0 + SD_TP * SD_FEAT_LD * 4.0));
        hipLaunchKernelGGL(k_fbank_norm, dim3((unsigned)run_items), dim3(256), 0, c->stream, d_db, d_max, d_nnorm, d_rowoff, d_feats);
        KCHECK(c);
    }
    return SD_OK;
}
