/*
 * sd_oracle.c -- CPU restatement (plain C99) of every NON-neural stage of the
 * reference speaker-diarization hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the shipped HIP path never links or calls it.
 *
 * Reference = /root/reference (leohuang2013/pyannote-audio_speaker-diarization_cpp).
 * "sd.cpp" below means pipeline/src/speakerDiarizer.cpp, "cl.cpp" means
 * pipeline/src/clustering/clustering.cpp.  Every function cites the lines it
 * restates.  Data are flat row-major arrays instead of nested std::vector.
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   - orc_closest_frame / orc_np_rint  : pinned by the reference fixture
 *     pipeline/src/test/closest_frame.txt (copied as data to tests/golden/).
 *   - orc_pdist / orc_linkage_centroid / orc_fcluster_distance : pinned against
 *     oracle/_ref/libref_clustering.so (the reference's own clustering.cpp built
 *     in place) and scipy.cluster.hierarchy.
 *   - orc_read_wav : pinned against oracle/_ref/libref_wav.so (the reference's own
 *     header-only WavReader, frontend/wav.h, built in place) on 8 / 16 / 32-bit files,
 *     extra sub-chunks, long fmt chunks and interleaved stereo.
 *   - orc_closest_frame, orc_window_start, orc_range_to_segment, orc_support (gap / merge), the frame count of
 *     orc_speaker_count : pinned on the reference's own segment/utils.py (Segment, SlidingWindow), imported in the
 *     build container -> tests/golden/ref_glue.npz (tools/mint_reference_fixtures.py, tests/test_reference_glue.py).
 *   - orc_binarize, orc_select_masks, orc_crop, orc_mask_compact, orc_wav_lens : pinned on binarize_ndarray / the mask choice of
 *     forward / crop / embedding_mask of the
 *     reference's segment/mysegment.py, executed from the file's own definitions -> tests/golden/ref_nn_glue.npz
 *     (tools/mint_reference_fixtures_nn.py, tests/test_reference_nn_glue.py).
 *   - orc_clustering_ex / orc_clustering_full / orc_cluster_embeddings_ex / orc_constrained_argmax (filter, AHC + small ->
 *     large re-assignment, recuts, assignment, constrained assignment) : pinned on the reference's clustering/Clustering.py
 *     executed on the real scipy -> tests/golden/ref_clustering.npz (tools/mint_reference_fixtures_clustering.py,
 *     tests/test_reference_clustering.py).
 *   - what is left -- the overlap-add of orc_aggregate beyond its frame indexing, the state machine of
 *     to_annotation beyond the Segment operations, reconstruct / to_diarization -- is a line-by-line restatement whose only
 *     golden in the reference is the README sample output (needs the missing ONNX blobs) => "parity unpinned" beyond the
 *     restatement itself.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -o libsd_oracle.so sd_oracle.c -lm
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS DBL_EPSILON

/* ------------------------------------------------------------------ */
/* a5/a16 helper: numpy-style rint, sd.cpp:260-272                      */
/* ------------------------------------------------------------------ */
int orc_np_rint(double val)
{
    double sgn = (val > 0) ? 1.0 : -1.0;
    /* sd.cpp:263 -- tie test on the fractional part (int() truncates) */
    if (fabs(val - (double)(int)val - 0.5 * sgn) < ORC_EPS) {
        int tmp = (int)round(val);          /* half away from zero */
        if (tmp % 2 == 0) return tmp;
        return tmp - (int)sgn;
    }
    return (int)round(val);
}

/* SlidingWindow::closest_frame, sd.cpp:1084-1090 (negative clamps to 0) */
long orc_closest_frame(double w_start, double w_step, double w_dur, double t)
{
    double closest = (t - w_start - .5 * w_dur) / w_step;
    if (closest < 0.0) closest = 0.0;
    return (long)orc_np_rint(closest);
}

/* ------------------------------------------------------------------ */
/* a2 framing rule, sd.cpp:1407-1480                                    */
/* ------------------------------------------------------------------ */
/* number of chunks SegmentModel::slide emits for n samples; *last_len gets
 * the length of the final ("last chunk" branch) chunk or 0 if none. */
long orc_num_chunks(long n, long window, long step, long *last_len)
{
    long i = 0, cnt = 0;
    while (i + window < n) { cnt++; i += step; }       /* sd.cpp:1419 strict < */
    long ll = 0;
    if (i + 1 < n) { ll = n - i; cnt++; }              /* sd.cpp:1457 */
    if (last_len) *last_len = ll;
    return cnt;
}

/* SegmentModel::crop, sd.cpp:1641-1662: [start, start+L) zero padded */
void orc_crop(const float *wav, long n, long start, long L, float *out)
{
    for (long j = 0; j < L; ++j) {
        long s = start + j;
        out[j] = (s >= 0 && s < n) ? wav[s] : 0.0f;
    }
}

/* ------------------------------------------------------------------ */
/* a4 hysteresis binarisation, sd.cpp:1506-1639 (+ Helper 623-708)      */
/* seg  : float [c][F][K]   out : double [c][F][K] in {0,1}             */
/* onset==offset: a frame whose score equals onset (within DBL_EPSILON) */
/* copies the last well-defined frame before it (initial_state if none) */
/* ------------------------------------------------------------------ */
void orc_binarize(const float *seg, long c, int F, int K, double onset,
                  int initial_state, double *out)
{
    for (long i = 0; i < c; ++i)
        for (int k = 0; k < K; ++k) {
            int have = 0, last = 0;
            for (int f = 0; f < F; ++f) {
                double s = (double)seg[(i * F + f) * K + k];   /* sd.cpp:1526 */
                int v;
                if (fabs(s - onset) < ORC_EPS) {               /* sd.cpp:1595 */
                    v = have ? last : (initial_state != 0);    /* sd.cpp:693-702 */
                } else {
                    v = (s > onset);                           /* sd.cpp:1582 */
                    have = 1; last = v;
                }
                out[(i * F + f) * K + k] = v ? 1.0 : 0.0;
            }
        }
}

/* ------------------------------------------------------------------ */
/* PipelineHelper::aggregate, sd.cpp:1167-1311                           */
/* scores [c][F][K] (NaN = missing).  frames window = (sf_start,         */
/* fr_step, fr_dur).  Returns num_frames (rows written to out [..][K]);  */
/* if out==NULL only the count is returned.                              */
/* ------------------------------------------------------------------ */
long orc_aggregate(const double *scores, long c, int F, int K,
                   double sf_start, double sf_step, double sf_dur,
                   double fr_step, double fr_dur,
                   double missing, int skip_average, double *out, long cap)
{
    double frame_target = sf_start + sf_dur + (double)(c - 1) * sf_step;   /* :1232 */
    long nf = orc_closest_frame(sf_start, fr_step, fr_dur, frame_target) + 1;
    if (!out) return nf;
    if (nf > cap) return -nf;
    double *cnt = (double *)calloc((size_t)nf * K, sizeof(double));
    double *msk = (double *)calloc((size_t)nf * K, sizeof(double));
    for (long i = 0; i < nf * K; ++i) out[i] = 0.0;
    double start = sf_start;
    for (long i = 0; i < c; ++i) {
        long sfr = orc_closest_frame(sf_start, fr_step, fr_dur, start);    /* :1251 */
        start += sf_step;                                                   /* :1253 */
        for (int j = 0; j < F; ++j) {
            long r = j + sfr;
            if (r >= nf) continue;            /* reference would write OOB */
            for (int k = 0; k < K; ++k) {
                double s = scores[(i * F + j) * K + k];
                double m = 1.0;
                if (isnan(s)) { m = 0.0; s = 0.0; }                         /* :1197-1201 */
                out[r * K + k] += s * m;                                    /* :1260 */
                cnt[r * K + k] += m;
                if (m > msk[r * K + k]) msk[r * K + k] = m;
            }
        }
    }
    for (long i = 0; i < nf * K; ++i) {
        if (!skip_average) out[i] /= fmax(cnt[i], ORC_EPS);                 /* :1288 */
        if (fabs(msk[i]) < ORC_EPS) out[i] = missing;                       /* :1302 */
    }
    free(cnt); free(msk);
    return nf;
}

/* ------------------------------------------------------------------ */
/* a5 speaker_count, sd.cpp:1665-1738 (trim 1742-1782)                   */
/* bin [c][F][K] -> count int[nf]; returns nf.  win[3] receives the      */
/* resulting count_frames (start, step, duration) -- note start=0.5.     */
/* ------------------------------------------------------------------ */
long orc_speaker_count(const double *bin, long c, int F, int K,
                       int *count, long cap, double *win)
{
    const double left = 0.1, right = 0.1, step = 0.5, dur = 5.0;
    long nl = (long)floor((double)F * left);            /* sd.cpp:1755 */
    long nr = (long)floor((double)F * right);
    int Ft = (int)(F - nr - nl);
    double *sum = (double *)malloc(sizeof(double) * (size_t)c * Ft);
    for (long i = 0; i < c; ++i)
        for (int j = 0; j < Ft; ++j) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += bin[(i * F + (j + nl)) * K + k];
            sum[i * Ft + j] = s;
        }
    double t_start = 0.0 + left * dur;                  /* sd.cpp:1776 */
    double t_dur = (1 - left - right) * dur;            /* sd.cpp:1778 */
    const double fstep = 0.016875, fdur = 0.016875;     /* sd.cpp:2430-2431 */
    long nf = orc_aggregate(sum, c, Ft, 1, t_start, step, t_dur, fstep, fdur,
                            0.0, 0, NULL, 0);
    if (nf > cap) { free(sum); return -nf; }
    double *agg = (double *)malloc(sizeof(double) * (size_t)nf);
    orc_aggregate(sum, c, Ft, 1, t_start, step, t_dur, fstep, fdur, 0.0, 0, agg, nf);
    for (long i = 0; i < nf; ++i) count[i] = orc_np_rint(agg[i]);   /* :1734 */
    if (win) { win[0] = t_start; win[1] = fstep; win[2] = fdur; }
    free(sum); free(agg);
    return nf;
}

/* ------------------------------------------------------------------ */
/* a6 clean-overlap + mask choice, sd.cpp:710-743, 3047-3078             */
/* bin [c][F][K] -> masks float [c*K][F] (item order chunk-major)        */
/* ------------------------------------------------------------------ */
void orc_select_masks(const double *bin, long c, int F, int K, float *masks)
{
    /* sd.cpp:3017 : ceil(F * 640 / (5.0*16000)) */
    size_t min_num_frames = (size_t)ceil((double)F * 640.0 / (5.0 * 16000));
    for (long i = 0; i < c; ++i)
        for (int k = 0; k < K; ++k) {
            float sum = 0.0f;
            for (int f = 0; f < F; ++f) {
                double tot = 0.0;
                for (int q = 0; q < K; ++q) tot += bin[(i * F + f) * K + q];
                double cv = (tot < 2.0) ? bin[(i * F + f) * K + k] : 0.0;  /* :730 */
                sum += (float)cv;                                           /* :3066 */
            }
            int use_clean = (sum > (float)min_num_frames);                  /* :3071 */
            for (int f = 0; f < F; ++f) {
                double v = bin[(i * F + f) * K + k];
                if (use_clean) {
                    double tot = 0.0;
                    for (int q = 0; q < K; ++q) tot += bin[(i * F + f) * K + q];
                    if (!(tot < 2.0)) v = 0.0;
                }
                masks[(i * K + k) * F + f] = (float)v;
            }
        }
}

/* ------------------------------------------------------------------ */
/* a7 mask interpolation + stream compaction, sd.cpp:746-797, 2451-2476  */
/* returns number of selected samples; signal zero-filled to L.         */
/* ------------------------------------------------------------------ */
long orc_mask_compact(const float *chunk, const float *mask, int F, long L,
                      float thr, float *signal)
{
    long cnt = 0;
    for (long j = 0; j < L; ++j) signal[j] = 0.0f;
    for (long j = 0; j < L; ++j) {
        int src = (int)(j * F / L);                     /* sd.cpp:760 int math */
        if (mask[src] > thr) signal[cnt++] = chunk[j];  /* sd.cpp:761, 791 */
    }
    return cnt;
}

/* wav_lens / too_short / whole-batch-NaN rule, sd.cpp:2467-2510 */
void orc_wav_lens(const long *counts, int B, long min_samples, float *lens,
                  unsigned char *too_short, int *all_nan)
{
    float max_len = 0;
    for (int i = 0; i < B; ++i) { float t = (float)counts[i]; if (t > max_len) max_len = t; }
    *all_nan = (max_len < (float)min_samples);          /* sd.cpp:2479 */
    for (int i = 0; i < B; ++i) {
        float w = (float)counts[i];
        if (w < (float)min_samples) { lens[i] = 1.0f; too_short[i] = 1; }
        else { lens[i] = w / max_len; too_short[i] = 0; }
    }
}

/* ------------------------------------------------------------------ */
/* a12 euclidean pdist (condensed), cl.cpp:408-431                       */
/* ------------------------------------------------------------------ */
void orc_pdist(const double *X, long N, int d, double *D)
{
    long p = 0;
    for (long i = 0; i < N; ++i)
        for (long j = i + 1; j < N; ++j) {
            double sum = 0.0;
            for (int q = 0; q < d; ++q) {
                double diff = X[i * d + q] - X[j * d + q];
                sum += diff * diff;
            }
            D[p++] = sqrt(sum);
        }
}

static inline long cidx(long n, long i, long j)        /* cl.cpp:236-242 */
{
    if (i < j) return n * i - (i * (i + 1) / 2) + (j - i - 1);
    return n * j - (j * (j + 1) / 2) + (i - j - 1);
}

/* indexed binary min-heap with the reference's exact tie behaviour, cl.cpp:28-119 */
typedef struct { long *pos_of_key, *key_at; double *val; long size; } orc_heap;

static void hp_swap(orc_heap *h, long a, long b)
{
    double tv = h->val[a]; h->val[a] = h->val[b]; h->val[b] = tv;
    long ka = h->key_at[a], kb = h->key_at[b];
    h->key_at[a] = kb; h->key_at[b] = ka;
    h->pos_of_key[ka] = b; h->pos_of_key[kb] = a;
}
static void hp_down(orc_heap *h, long idx)
{
    long ch = 2 * idx + 1;
    while (ch < h->size) {
        if (ch + 1 < h->size && h->val[ch + 1] < h->val[ch]) ch += 1;
        if (h->val[idx] > h->val[ch]) { hp_swap(h, idx, ch); idx = ch; ch = 2 * idx + 1; }
        else break;
    }
}
static void hp_up(orc_heap *h, long idx)
{
    long par = (idx - 1) >> 1;
    while (idx > 0 && h->val[par] > h->val[idx]) { hp_swap(h, idx, par); idx = par; par = (idx - 1) >> 1; }
}
static void hp_change(orc_heap *h, long key, double v)
{
    long idx = h->pos_of_key[key];
    double old = h->val[idx];
    h->val[idx] = v;
    if (v < old) hp_up(h, idx); else hp_down(h, idx);
}

/* nearest active neighbour with higher index, cl.cpp:259-276 */
static void nn_above(long n, const double *D, const int *size, long x, long *y, double *dmin)
{
    double cur = INFINITY; long best = -1;
    for (long i = x + 1; i < n; ++i) {
        if (size[i] == 0) continue;
        double dist = D[cidx(n, x, i)];
        if (dist < cur) { cur = dist; best = i; }
    }
    *y = best; *dmin = cur;
}

/* Lance-Williams centroid update, cl.cpp:250-256 (operation order kept) */
static inline double lw_centroid(double d_xi, double d_yi, double d_xy, int sx, int sy)
{
    return sqrt((((sx * d_xi * d_xi) + (sy * d_yi * d_yi)) -
                 (sx * sy * d_xy * d_xy) / (sx + sy)) / (sx + sy));
}

/* scipy-generic centroid linkage on a condensed matrix, cl.cpp:289-406.
 * Din is not modified.  Z is [n-1][4].                                   */
void orc_linkage_centroid(const double *Din, long n, double *Z)
{
    if (n < 2) return;
    long m = n * (n - 1) / 2;
    double *D = (double *)malloc(sizeof(double) * (size_t)m);
    memcpy(D, Din, sizeof(double) * (size_t)m);
    int *size = (int *)malloc(sizeof(int) * n);
    long *cid = (long *)malloc(sizeof(long) * n);
    long *nb = (long *)malloc(sizeof(long) * (n - 1));
    double *md = (double *)malloc(sizeof(double) * (n - 1));
    for (long i = 0; i < n; ++i) { size[i] = 1; cid[i] = i; }
    for (long x = 0; x < n - 1; ++x) nn_above(n, D, size, x, &nb[x], &md[x]);

    orc_heap h;
    h.size = n - 1;
    h.pos_of_key = (long *)malloc(sizeof(long) * (n - 1));
    h.key_at = (long *)malloc(sizeof(long) * (n - 1));
    h.val = (double *)malloc(sizeof(double) * (n - 1));
    for (long i = 0; i < n - 1; ++i) { h.pos_of_key[i] = i; h.key_at[i] = i; h.val[i] = md[i]; }
    for (long i = h.size / 2; i >= 0; --i) hp_down(&h, i);            /* cl.cpp:94 */

    long x = 0, y = 0; double dist = 0;
    for (long k = 0; k < n - 1; ++k) {
        for (long i = 0; i < n - k; ++i) {                             /* cl.cpp:323 */
            x = h.key_at[0]; dist = h.val[0]; y = nb[x];
            if (dist == D[cidx(n, x, y)]) break;
            nn_above(n, D, size, x, &y, &dist);
            nb[x] = y; md[x] = dist; hp_change(&h, x, dist);
        }
        hp_swap(&h, 0, h.size - 1); h.size -= 1; hp_down(&h, 0);       /* remove_min */

        long idx = cid[x], idy = cid[y];
        int nx = size[x], ny = size[y];
        if (idx > idy) { long t = idx; idx = idy; idy = t; }
        Z[k * 4 + 0] = (double)idx; Z[k * 4 + 1] = (double)idy;
        Z[k * 4 + 2] = dist;        Z[k * 4 + 3] = (double)(nx + ny);
        size[x] = 0; size[y] = nx + ny; cid[y] = n + k;

        for (long z = 0; z < n; ++z) {                                 /* cl.cpp:361 */
            if (size[z] == 0 || z == y) continue;
            D[cidx(n, z, y)] = lw_centroid(D[cidx(n, z, x)], D[cidx(n, z, y)], dist, nx, ny);
        }
        for (long z = 0; z < x; ++z)                                   /* cl.cpp:374 */
            if (size[z] > 0 && nb[z] == x) nb[z] = y;
        for (long z = 0; z < y; ++z) {                                 /* cl.cpp:381 */
            if (size[z] == 0) continue;
            double dz = D[cidx(n, z, y)];
            if (dz < md[z]) { nb[z] = y; md[z] = dz; hp_change(&h, z, dz); }
        }
        if (y < n - 1) {                                               /* cl.cpp:395 */
            long z; double dz;
            nn_above(n, D, size, y, &z, &dz);
            if (z != -1) { nb[y] = z; md[y] = dz; hp_change(&h, y, dz); }
        }
    }
    free(D); free(size); free(cid); free(nb); free(md);
    free(h.pos_of_key); free(h.key_at); free(h.val);
}

/* fcluster(criterion="distance"), cl.cpp:121-232, 442-457; T is 1-based */
void orc_fcluster_distance(const double *Z, long n, double cutoff, int *T)
{
    if (n < 2) { if (n == 1) T[0] = 1; return; }
    double *MD = (double *)calloc((size_t)n, sizeof(double));
    long *stack = (long *)malloc(sizeof(long) * n);
    unsigned char *vis = (unsigned char *)calloc((size_t)(2 * n), 1);
    long k = 0; stack[0] = 2 * n - 2;
    while (k >= 0) {                                   /* max merge height per node */
        long root = stack[k] - n;
        long lc = (long)Z[root * 4 + 0], rc = (long)Z[root * 4 + 1];
        if (lc >= n && !vis[lc]) { vis[lc] = 1; stack[++k] = lc; continue; }
        if (rc >= n && !vis[rc]) { vis[rc] = 1; stack[++k] = rc; continue; }
        double mx = Z[root * 4 + 2];
        if (lc >= n && MD[lc - n] > mx) mx = MD[lc - n];
        if (rc >= n && MD[rc - n] > mx) mx = MD[rc - n];
        MD[root] = mx;
        k--;
    }
    memset(vis, 0, (size_t)(2 * n));
    long ncl = 0, leader = -1;
    k = 0; stack[0] = 2 * n - 2;
    while (k >= 0) {                                   /* left-first DFS labelling */
        long root = stack[k] - n;
        long lc = (long)Z[root * 4 + 0], rc = (long)Z[root * 4 + 1];
        if (leader == -1 && MD[root] <= cutoff) { leader = root; ncl++; }
        if (lc >= n && !vis[lc]) { vis[lc] = 1; stack[++k] = lc; continue; }
        if (rc >= n && !vis[rc]) { vis[rc] = 1; stack[++k] = rc; continue; }
        if (lc < n) { if (leader == -1) ncl++; T[lc] = (int)ncl; }
        if (rc < n) { if (leader == -1) ncl++; T[rc] = (int)ncl; }
        if (leader == root) leader = -1;
        k--;
    }
    free(MD); free(stack); free(vis);
}

/* Clustering::cluster, cl.cpp:459-468 : pdist + linkage + fcluster */
void orc_ahc_labels(const double *Xn, long N, int d, double cutoff, int *T, double *Zout)
{
    if (N < 2) { if (N == 1) T[0] = 1; return; }
    double *D = (double *)malloc(sizeof(double) * (size_t)(N * (N - 1) / 2));
    double *Z = Zout ? Zout : (double *)malloc(sizeof(double) * (size_t)(N - 1) * 4);
    orc_pdist(Xn, N, d, D);
    orc_linkage_centroid(D, N, Z);
    orc_fcluster_distance(Z, N, cutoff, T);
    free(D); if (!Zout) free(Z);
}

/* cosine distance, sd.cpp:476-498.  returns NaN for NaN input; zero norm
 * (reference throws) is reported through *err. */
static double cos_dist(const double *a, const double *b, int d, int *err)
{
    double dot = 0.0, m1 = 0.0, m2 = 0.0;
    for (int i = 0; i < d; ++i) { dot += a[i] * b[i]; m1 += a[i] * a[i]; m2 += b[i] * b[i]; }
    if (m1 == 0.0 || m2 == 0.0) { if (err) *err = 1; return NAN; }
    return 1.0 - (dot / (sqrt(m1) * sqrt(m2)));
}

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

/* ------------------------------------------------------------------ */
/* a11 Cluster::cluster, sd.cpp:2300-2422                                */
/* X [N][d] un-normalised train embeddings -> labels 0..K-1; returns K   */
/* (or -1 on zero-norm centroid = reference throws).  threshold is the   */
/* reference's float-typed member (sd.cpp:2049) promoted to double.      */
/* ------------------------------------------------------------------ */
/* number of fcluster labels (1-based) with at least mcs members */
static int count_large_1based(const int *T, long N, long mcs)
{
    int mx = 0;
    for (long i = 0; i < N; ++i) if (T[i] > mx) mx = T[i];
    long *cnt = (long *)calloc((size_t)mx + 1, sizeof(long));
    for (long i = 0; i < N; ++i) cnt[T[i]]++;
    int nl = 0;
    for (int k = 1; k <= mx; ++k) if (cnt[k] >= mcs) nl++;
    free(cnt);
    return nl;
}

typedef struct { double key; long idx; } orc_key;
static int cmp_key(const void *a, const void *b)
{
    const orc_key *x = (const orc_key *)a, *y = (const orc_key *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);          /* stable */
}

/* Constrained number of clusters: unimplemented in the C++ reference (assert(false), sd.cpp:2368-2369);
 * restated from the Python it was ported from, clustering/Clustering.py:352-399.  labels: 0-based out. */
static void orc_constrained_recut(const double *Z, long N, double threshold, long mcs, int num_clusters, int *labels)
{
    double *Zi = (double *)malloc(sizeof(double) * (size_t)(N - 1) * 4);
    memcpy(Zi, Z, sizeof(double) * (size_t)(N - 1) * 4);
    orc_key *ord = (orc_key *)malloc(sizeof(orc_key) * (size_t)(N - 1));
    for (long k = 0; k < N - 1; ++k) { Zi[k * 4 + 2] = (double)k; ord[k].key = fabs(Z[k * 4 + 2] - threshold); ord[k].idx = k; }
    qsort(ord, (size_t)(N - 1), sizeof(orc_key), cmp_key);
    int *T = (int *)malloc(sizeof(int) * (size_t)N);
    long best_iteration = N - 1; int best_large = 1, exact = 0;
    for (long q = 0; q < N - 1; ++q) {
        const long it = ord[q].idx;
        if (Zi[it * 4 + 3] < (double)mcs) continue;
        orc_fcluster_distance(Zi, N, (double)it, T);
        const int nl = count_large_1based(T, N, mcs);
        if (abs(nl - num_clusters) < abs(best_large - num_clusters)) { best_iteration = it; best_large = nl; }
        if (nl == num_clusters) { exact = 1; break; }
    }
    if (!exact) orc_fcluster_distance(Zi, N, (double)best_iteration, T);
    for (long i = 0; i < N; ++i) labels[i] = T[i] - 1;
    free(Zi); free(ord); free(T);
}

int orc_cluster_embeddings_ex(const double *X, long N, int d, float threshold,
                              long min_cluster_size_cfg, int num_clusters, int min_clusters, int max_clusters, int *labels);
int orc_cluster_embeddings(const double *X, long N, int d, float threshold,
                           long min_cluster_size_cfg, int *labels)
{
    return orc_cluster_embeddings_ex(X, N, d, threshold, min_cluster_size_cfg, -1, -1, -1, labels);
}

int orc_cluster_embeddings_ex(const double *X, long N, int d, float threshold,
                              long min_cluster_size_cfg, int num_clusters, int min_clusters, int max_clusters, int *labels)
{
    /* sd.cpp:2308 */
    long r = (long)round(0.1 * (double)N);
    long mcs = r > 1 ? r : 1;
    if (min_cluster_size_cfg < mcs) mcs = min_cluster_size_cfg;

    double *Xn = (double *)malloc(sizeof(double) * (size_t)N * d);
    for (long i = 0; i < N; ++i) {                         /* sd.cpp:332-357 */
        double s = 0.0;
        for (int q = 0; q < d; ++q) s += X[i * d + q] * X[i * d + q];
        /* Helper::L2Norm returns float (sd.cpp:332): sqrt result narrowed */
        double nrm = (double)(float)sqrt(s);
        for (int q = 0; q < d; ++q) Xn[i * d + q] = (nrm != 0.0) ? X[i * d + q] / nrm : X[i * d + q];
    }
    double *Zc = (double *)malloc(sizeof(double) * (size_t)(N > 1 ? N - 1 : 1) * 4);
    orc_ahc_labels(Xn, N, d, (double)threshold, labels, Zc);
    free(Xn);
    for (long i = 0; i < N; ++i) labels[i] -= 1;
    if (num_clusters != -1 || min_clusters != -1 || max_clusters != -1) {
        /* set_num_clusters with the Python semantics (Clustering.py:21-43) */
        if (num_clusters != -1) { min_clusters = num_clusters; max_clusters = num_clusters; }
        if (min_clusters == -1) min_clusters = 1;
        if (max_clusters == -1) max_clusters = (int)N;
        if (min_clusters > N) min_clusters = (int)N; if (min_clusters < 1) min_clusters = 1;
        if (max_clusters > N) max_clusters = (int)N; if (max_clusters < 1) max_clusters = 1;
        if (min_clusters > max_clusters) min_clusters = max_clusters;
        int target = (min_clusters == max_clusters) ? min_clusters : -1;
        int *T1 = (int *)malloc(sizeof(int) * (size_t)N);
        for (long i = 0; i < N; ++i) T1[i] = labels[i] + 1;
        const int nlarge0 = count_large_1based(T1, N, mcs);
        free(T1);
        if (nlarge0 < min_clusters) target = min_clusters;
        if (nlarge0 > max_clusters) target = max_clusters;
        if (target != -1) orc_constrained_recut(Zc, N, (double)threshold, mcs, target, labels);
    }
    free(Zc);
    int maxl = 0;
    for (long i = 0; i < N; ++i) if (labels[i] > maxl) maxl = labels[i];
    int nl = maxl + 1;
    long *cnt = (long *)calloc((size_t)nl, sizeof(long));
    for (long i = 0; i < N; ++i) cnt[labels[i]]++;
    int *large = (int *)malloc(sizeof(int) * nl), *small = (int *)malloc(sizeof(int) * nl);
    int nlarge = 0, nsmall = 0;
    for (int k = 0; k < nl; ++k) {
        if (cnt[k] == 0) continue;
        if (cnt[k] >= mcs) large[nlarge++] = k; else small[nsmall++] = k;
    }
    int K;
    if (nlarge == 0) {                                     /* sd.cpp:2371 (assert build aborts at 2368) */
        for (long i = 0; i < N; ++i) labels[i] = 0;
        K = 1; goto done;
    }
    if (nsmall == 0) { K = nl; goto done; }                /* sd.cpp:2377: labels returned as is */
    qsort(large, nlarge, sizeof(int), cmp_int);
    qsort(small, nsmall, sizeof(int), cmp_int);
    {
        double *lc = (double *)calloc((size_t)nlarge * d, sizeof(double));
        double *sc = (double *)calloc((size_t)nsmall * d, sizeof(double));
        for (int a = 0; a < nlarge; ++a) {                 /* sd.cpp:442-473 means of UN-normalised rows */
            long c = 0;
            for (long i = 0; i < N; ++i) if (labels[i] == large[a]) { for (int q = 0; q < d; ++q) lc[a * d + q] += X[i * d + q]; c++; }
            if (c > 0) for (int q = 0; q < d; ++q) lc[a * d + q] /= (double)c;
        }
        for (int a = 0; a < nsmall; ++a) {
            long c = 0;
            for (long i = 0; i < N; ++i) if (labels[i] == small[a]) { for (int q = 0; q < d; ++q) sc[a * d + q] += X[i * d + q]; c++; }
            if (c > 0) for (int q = 0; q < d; ++q) sc[a * d + q] /= (double)c;
        }
        int err = 0;
        int *target = (int *)malloc(sizeof(int) * nsmall);
        for (int s = 0; s < nsmall; ++s) {
            float minVal = FLT_MAX; int best = -1;          /* sd.cpp:2396 float accumulator */
            for (int a = 0; a < nlarge; ++a) {
                double dd = cos_dist(&lc[a * d], &sc[s * d], d, &err);
                if (dd < minVal) { minVal = (float)dd; best = a; }
            }
            target[s] = best;
        }
        /* sd.cpp:2394-2412 : sequential relabel, one small cluster at a time */
        for (int s = 0; s < nsmall; ++s)
            for (long i = 0; i < N; ++i)
                if (labels[i] == small[s] && target[s] >= 0) labels[i] = large[target[s]];
        free(lc); free(sc); free(target);
        if (err) { K = -1; goto done; }
    }
    {   /* findUniqueClusters, sd.cpp:519-548 : renumber in sorted-id order */
        int *map = (int *)malloc(sizeof(int) * nl);
        for (int k = 0; k < nl; ++k) map[k] = -1;
        for (long i = 0; i < N; ++i) map[labels[i]] = 0;
        int nk = 0;
        for (int k = 0; k < nl; ++k) if (map[k] == 0) map[k] = nk++;
        for (long i = 0; i < N; ++i) labels[i] = map[labels[i]];
        free(map);
        K = nk;
    }
done:
    free(cnt); free(large); free(small);
    return K;
}

/* ------------------------------------------------------------------ */
/* a10 + a11 + a14 Cluster::clustering, sd.cpp:2063-2259                 */
/* emb [c][S][d] (NaN rows = no embedding) -> hard [c][S]; returns K or  */
/* 0 when max_clusters<2 path (all zeros) was taken, -1 on error.        */
/* ------------------------------------------------------------------ */
int orc_clustering_ex(const double *emb, long c, int S, int d, float threshold, long min_cluster_size_cfg,
                      int num_clusters, int min_clusters, int max_clusters, int *hard, int *train_labels_out, long *ntrain_out);
int orc_clustering_full(const double *emb, long c, int S, int d, float threshold, long min_cluster_size_cfg,
                        int num_clusters, int min_clusters, int max_clusters, int constrained, int *hard, int *train_labels_out,
                        long *ntrain_out, double *soft_out, long soft_cap);
int orc_clustering(const double *emb, long c, int S, int d, float threshold,
                   long min_cluster_size_cfg, int *hard, int *train_labels_out, long *ntrain_out)
{
    return orc_clustering_ex(emb, c, S, d, threshold, min_cluster_size_cfg, -1, -1, -1, hard, train_labels_out, ntrain_out);
}
int orc_clustering_ex(const double *emb, long c, int S, int d, float threshold, long min_cluster_size_cfg,
                      int num_clusters, int min_clusters, int max_clusters, int *hard, int *train_labels_out, long *ntrain_out)
{
    return orc_clustering_full(emb, c, S, d, threshold, min_cluster_size_cfg, num_clusters, min_clusters, max_clusters, 0,
                               hard, train_labels_out, ntrain_out, NULL, 0);
}

/* scipy.optimize.linear_sum_assignment (minimisation, nr <= nc), the dependency clustering/Clustering.py:90 calls
 * (third-party, version not pinned by the reference).  Restated from scipy's rectangular_lsap.cpp (scipy >= 1.6: Crouse's
 * shortest augmenting path; `remaining` filled in reverse; ties go to an unassigned column); pinned against the scipy
 * installed in this image by tests/test_oracle.py. */
static void orc_lsap(int nr, int nc, const double *cost, int *col4row)
{
    double *u = (double *)calloc((size_t)nr, sizeof(double)), *v = (double *)calloc((size_t)nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * (size_t)nc);
    int *path = (int *)malloc(sizeof(int) * (size_t)nc), *row4col = (int *)malloc(sizeof(int) * (size_t)nc);
    int *remaining = (int *)malloc(sizeof(int) * (size_t)nc);
    char *SR = (char *)malloc((size_t)nr), *SC = (char *)malloc((size_t)nc);
    for (int j = 0; j < nc; ++j) { path[j] = -1; row4col[j] = -1; }
    for (int i = 0; i < nr; ++i) col4row[i] = -1;
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0.0;
        int num_remaining = nc;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        memset(SR, 0, (size_t)nr); memset(SC, 0, (size_t)nc);
        for (int j = 0; j < nc; ++j) spc[j] = INFINITY;
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1; double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                int j = remaining[it];
                double r = minVal + cost[i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (index < 0) goto out;
            int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r) if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j) if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        while (1) {
            int r = path[j];
            row4col[j] = r;
            int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
out:
    free(u); free(v); free(spc); free(path); free(row4col); free(remaining); free(SR); free(SC);
}
/* exported for the pin test: maximise or minimise an nr x nc cost matrix like scipy (transposing when nr > nc);
 * row_ind/col_ind get min(nr, nc) pairs sorted by row */
int orc_linear_sum_assignment(const double *cost, int nr, int nc, int maximize, int *row_ind, int *col_ind)
{
    int tr = nc < nr, R = tr ? nc : nr, Cc = tr ? nr : nc;
    double *t = (double *)malloc(sizeof(double) * (size_t)nr * nc);
    for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) {
        double x = maximize ? -cost[i * nc + j] : cost[i * nc + j];
        if (tr) t[j * nr + i] = x; else t[i * nc + j] = x;
    }
    int *c4r = (int *)malloc(sizeof(int) * (size_t)R);
    orc_lsap(R, Cc, t, c4r);
    int n = 0;
    if (!tr) { for (int i = 0; i < R; ++i) { row_ind[n] = i; col_ind[n] = c4r[i]; n++; } }
    else {                                               /* rows of the transposed problem are columns: report sorted by row */
        for (int i = 0; i < nr; ++i) for (int k = 0; k < R; ++k) if (c4r[k] == i) { row_ind[n] = i; col_ind[n] = k; n++; }
    }
    free(t); free(c4r);
    return n;
}

/* constrained_argmax, clustering/Clustering.py:81-94: soft [c][S][K] (NaN allowed) -> hard [c][S] (-2 = unassigned) */
void orc_constrained_argmax(const double *soft, long c, int S, int K, int *hard)
{
    double fill = INFINITY;                                        /* np.nanmin over the whole tensor, :83 */
    for (long i = 0; i < c * S * K; ++i) if (!isnan(soft[i]) && soft[i] < fill) fill = soft[i];
    if (fill == INFINITY) fill = 0.0;
    double *cost = (double *)malloc(sizeof(double) * (size_t)S * K);
    int *ri = (int *)malloc(sizeof(int) * (size_t)(S > K ? S : K)), *ci = (int *)malloc(sizeof(int) * (size_t)(S > K ? S : K));
    for (long i = 0; i < c; ++i) {
        for (int q = 0; q < S * K; ++q) { double x = soft[i * S * K + q]; cost[q] = isnan(x) ? fill : x; }
        for (int s = 0; s < S; ++s) hard[i * S + s] = -2;          /* :88 */
        int n = orc_linear_sum_assignment(cost, S, K, 1, ri, ci);  /* :90-92 */
        for (int q = 0; q < n; ++q) hard[i * S + ri[q]] = ci[q];
    }
    free(cost); free(ri); free(ci);
}

int orc_clustering_full(const double *emb, long c, int S, int d, float threshold, long min_cluster_size_cfg,
                        int num_clusters, int min_clusters, int max_clusters, int constrained, int *hard, int *train_labels_out,
                        long *ntrain_out, double *soft_out, long soft_cap)
{
    long M = c * S, N = 0;
    long *tidx = (long *)malloc(sizeof(long) * (size_t)M);
    for (long i = 0; i < M; ++i) if (!isnan(emb[i * d])) tidx[N++] = i;     /* sd.cpp:2224 */
    if (ntrain_out) *ntrain_out = N;
    {
        int mxc = (num_clusters != -1) ? num_clusters : ((max_clusters != -1) ? max_clusters : (int)N);
        if (mxc > N) mxc = (int)N;
        if (N < 2 || mxc < 2) {                                             /* sd.cpp:2081 */
            for (long i = 0; i < M; ++i) hard[i] = 0;
            free(tidx); return 0;
        }
    }
    double *X = (double *)malloc(sizeof(double) * (size_t)N * d);
    for (long i = 0; i < N; ++i) memcpy(&X[i * d], &emb[tidx[i] * d], sizeof(double) * d);
    int *lab = (int *)malloc(sizeof(int) * (size_t)N);
    int K = orc_cluster_embeddings_ex(X, N, d, threshold, min_cluster_size_cfg, num_clusters, min_clusters, max_clusters, lab);
    if (K < 0) { free(tidx); free(X); free(lab); return -1; }
    if (train_labels_out) memcpy(train_labels_out, lab, sizeof(int) * (size_t)N);
    /* assign_embeddings, sd.cpp:2119-2212 */
    int nk = 0;
    for (long i = 0; i < N; ++i) if (lab[i] + 1 > nk) nk = lab[i] + 1;
    double *cen = (double *)calloc((size_t)nk * d, sizeof(double));
    for (int k = 0; k < nk; ++k) {
        size_t mc = 0;
        for (long j = 0; j < N; ++j) if (lab[j] == k) { mc++; for (int q = 0; q < d; ++q) cen[k * d + q] += X[j * d + q]; }
        for (int q = 0; q < d; ++q) cen[k * d + q] /= (double)mc;           /* sd.cpp:2165 */
    }
    int err = 0;
    double *softm = (double *)malloc(sizeof(double) * (size_t)M * nk);
    for (long i = 0; i < M; ++i) {
        int best = 0; double mv = -DBL_MAX;                                 /* sd.cpp:293-316 */
        for (int k = 0; k < nk; ++k) {
            double soft = 2.0 - cos_dist(&emb[i * d], &cen[k * d], d, &err);   /* sd.cpp:2191-2207 */
            softm[i * nk + k] = soft;
            if (soft > mv) { mv = soft; best = k; }
        }
        hard[i] = best;
    }
    if (constrained) orc_constrained_argmax(softm, c, S, nk, hard);         /* Clustering.py:156-157 */
    if (soft_out && (long)M * nk <= soft_cap) memcpy(soft_out, softm, sizeof(double) * (size_t)M * nk);
    free(softm);
    free(tidx); free(X); free(lab); free(cen);
    return err ? -1 : nk;
}

/* inactive local speakers -> -2, sd.cpp:3172-3191 */
void orc_mark_inactive(const double *bin, long c, int F, int S, int *hard)
{
    for (long i = 0; i < c; ++i)
        for (int k = 0; k < S; ++k) {
            float s = 0.0f;
            for (int f = 0; f < F; ++f) s += (float)bin[(i * F + f) * S + k];
            if (fabs((double)s) < ORC_EPS) hard[i * S + k] = -2;
        }
}

/* SlidingWindow::operator[], sd.cpp:1092-1115 (walks from 0.0, may bail) */
double orc_window_start(double step, double dur, long num_samples, int pos);
static double sw_index_start(double step, double dur, long num_samples, int pos) { return orc_window_start(step, dur, num_samples, pos); }
double orc_window_start(double step, double dur, long num_samples, int pos)
{
    int window_size = (int)round(dur * 16000.0), step_size = (int)round(step * 16000.0);
    double start = 0.0; size_t cur = 0; int index = 0;
    while (1) {
        if (index == pos) return start;
        if (cur + (size_t)window_size >= (size_t)num_samples) break;
        start += step; cur += (size_t)step_size; index++;
    }
    return 0.0;
}

/* SlidingWindow::range_to_segment as to_diarization uses it for the extents (sd.cpp:2691-2706, 1029-1083):
 * start + (i0 - .5) * step + .5 * dur, end = that + n * step, and i0 == 0 extends the start to the window start
 * (segment/utils.py:497-540).  out[0] = start, out[1] = end */
void orc_range_to_segment(double w_start, double w_step, double w_dur, long i0, long n, double *out)
{
    double start = w_start + ((double)i0 - .5) * w_step + .5 * w_dur;
    double end = start + (double)n * w_step;
    if (i0 == 0) start = w_start;
    out[0] = start; out[1] = end;
}

/* crop_segment index part, sd.cpp:2567-2618; returns rows [*r0,*r1) */
static void crop_range(double w_start, double w_step, double w_dur, long w_ns,
                       double f_start, double f_end, long n_rows,
                       long *r0, long *r1, float *new_start)
{
    float i_ = (float)((f_start - w_dur - w_start) / w_step);     /* :2577 */
    int rs = (int)ceilf(i_);
    if (rs < 0) rs = 0;
    float j_ = (float)((f_end - w_start) / w_step);               /* :2586 */
    int re = (int)floorf(j_) + 1;
    *new_start = (float)sw_index_start(w_step, w_dur, w_ns, rs);  /* :2590 */
    size_t s = (size_t)(double)rs, e = (size_t)(double)re;
    if (s >= (size_t)n_rows) { *r0 = 0; *r1 = 0; return; }
    *r0 = (long)s;
    *r1 = (long)(e < (size_t)n_rows ? e : (size_t)n_rows);
    if (*r1 < *r0) *r1 = *r0;
}

/* ------------------------------------------------------------------ */
/* a15 + a16 reconstruct + to_diarization, sd.cpp:2638-2848              */
/* seg float [c][F][S], hard int [c][S], count int [ncount] with         */
/* count window cwin = (start, step, dur) and cwin_ns (its num_samples). */
/* Output: binary double [rows][K]; returns rows (or -needed if cap too  */
/* small); *K_out, *start_out (float-rounded frame start).               */
/* ------------------------------------------------------------------ */
long orc_reconstruct(const float *seg, long c, int F, int S, const int *hard,
                     const int *count, long ncount, const double *cwin, long cwin_ns,
                     long n_samples, double *binary, long cap, int *K_out, double *start_out)
{
    int K = 0;
    for (long i = 0; i < c * S; ++i) if (hard[i] > K) K = hard[i];
    K += 1;                                                        /* sd.cpp:2803-2812 */
    double *cl = (double *)malloc(sizeof(double) * (size_t)c * F * K);
    for (long i = 0; i < c * F * K; ++i) cl[i] = NAN;
    for (long i = 0; i < c; ++i)
        for (int a = 0; a < S; ++a) {
            int k = hard[i * S + a];
            if (k == -2) continue;                                 /* :2824 */
            for (int f = 0; f < F; ++f) {
                float mx = -INFINITY;                              /* :2767-2786 */
                for (int j = 0; j < S; ++j)
                    if (hard[i * S + j] == k) { float v = seg[(i * F + f) * S + j]; mx = (mx < v) ? v : mx; }
                cl[(i * F + f) * K + k] = (double)mx;
            }
        }
    const double fstep = cwin[1], fdur = cwin[2];
    long nact = orc_aggregate(cl, c, F, K, 0.0, 0.5, 5.0, fstep, fdur, 0.0, 1, NULL, 0);
    double *act = (double *)malloc(sizeof(double) * (size_t)nact * K);
    orc_aggregate(cl, c, F, K, 0.0, 0.5, 5.0, fstep, fdur, 0.0, 1, act, nact);
    free(cl);
    /* extents, sd.cpp:2691-2706 ; activations window = (0.0, fstep, fdur) */
    double ext[2];
    orc_range_to_segment(0.0, fstep, fdur, 0, nact, ext);
    double a_start = ext[0], a_end = ext[1];
    orc_range_to_segment(cwin[0], cwin[1], cwin[2], 0, ncount, ext);
    double c_start = ext[0], c_end = ext[1];
    double f0 = a_start > c_start ? a_start : c_start;
    double f1 = a_end < c_end ? a_end : c_end;
    long ar0, ar1, cr0, cr1; float astart, cstart_unused;
    crop_range(0.0, fstep, fdur, n_samples, f0, f1, nact, &ar0, &ar1, &astart);
    crop_range(cwin[0], cwin[1], cwin[2], cwin_ns, f0, f1, ncount, &cr0, &cr1, &cstart_unused);
    long rows = ar1 - ar0, crow = cr1 - cr0;
    if (K_out) *K_out = K;
    if (start_out) *start_out = (double)astart;
    if (rows > cap) { free(act); return -rows; }
    for (long i = 0; i < rows * K; ++i) binary[i] = 0.0;
    int *order = (int *)malloc(sizeof(int) * K);
    for (long i = 0; i < crow && i < rows; ++i) {
        const double *a = &act[(ar0 + i) * K];
        /* stable argsort of -a, sd.cpp:2724-2730 (insertion sort is stable) */
        for (int k = 0; k < K; ++k) order[k] = k;
        for (int p = 1; p < K; ++p) {
            int o = order[p]; int q = p - 1;
            while (q >= 0 && (-1.0 * a[o]) < (-1.0 * a[order[q]])) { order[q + 1] = order[q]; q--; }
            order[q + 1] = o;
        }
        int kk = count[cr0 + i]; if (kk > K) kk = K;               /* :2681 */
        for (int j = 0; j < kk; ++j) binary[i * K + order[j]] = 1.0;
    }
    free(order); free(act);
    return rows;
}

typedef struct { double start, end; int label; } orc_turn;
void orc_final_sort(orc_turn *t, long n);            /* final_sort.cpp */

/* Track::support, sd.cpp:911-941 with Segment::gap / ::merge (831-860): start-sorted segments of one label, merged in place
 * while the gap to the running segment is < collar.  Returns the new count. */
long orc_support(orc_turn *segs, long ns, double collar)
{
    if (ns <= 0) return 0;
    long w = 0; orc_turn cur = segs[0];
    for (long i = 1; i < ns; ++i) {
        double gap;
        if (cur.start < segs[i].start) gap = (cur.end >= segs[i].start) ? 0.0 : segs[i].start - cur.end;
        else gap = (cur.start <= segs[i].end) ? 0.0 : cur.start - segs[i].end;
        if (gap < collar) {
            if (segs[i].start < cur.start) cur.start = segs[i].start;
            if (segs[i].end > cur.end) cur.end = segs[i].end;
        } else { segs[w++] = cur; cur = segs[i]; }
    }
    segs[w++] = cur;
    return w;
}

static int cmp_seg(const void *a, const void *b)
{
    double x = ((const orc_turn *)a)->start, y = ((const orc_turn *)b)->start;
    return (x > y) - (x < y);
}

/* ------------------------------------------------------------------ */
/* a17 to_annotation + support + finalResult, sd.cpp:2852-2935,          */
/* 911-941, 962-978.  scores [rows][K].  min_off is the reference's      */
/* float-typed constant promoted to double (sd.cpp:3210).                */
/* finalResult() orders the turns with std::sort (unstable): turns with   */
/* EQUAL start times come out in whatever order libstdc++'s introsort     */
/* leaves them, so that call is made for real in final_sort.cpp (g++, the */
/* same libstdc++ the reference links) on the same input order (tracks in */
/* ascending label, segments in time order).                              */
/* ------------------------------------------------------------------ */
long orc_to_annotation(const double *scores, long rows, int K, double w_start,
                       double w_step, double w_dur, double onset, double offset,
                       double min_on, double min_off, orc_turn *out, long cap)
{
    if (rows <= 0) return 0;
    double *ts = (double *)malloc(sizeof(double) * (size_t)rows);
    for (long i = 0; i < rows; ++i) {
        double s = w_start + (double)i * w_step, e = s + w_dur;     /* :2865-2867 */
        ts[i] = (s + e) / 2;
    }
    long nout = 0;
    orc_turn *segs = (orc_turn *)malloc(sizeof(orc_turn) * (size_t)(rows + 1));
    for (int k = 0; k < K; ++k) {
        long ns = 0;
        double start = ts[0];
        int active = scores[0 * K + k] > onset;
        for (long j = 1; j < rows; ++j) {
            double v = scores[j * K + k];
            if (active) {
                if (v < offset) { segs[ns].start = start; segs[ns].end = ts[j]; segs[ns].label = k; ns++; start = ts[j]; active = 0; }
            } else if (v > onset) { start = ts[j]; active = 1; }
        }
        if (active) { segs[ns].start = start; segs[ns].end = ts[rows - 1]; segs[ns].label = k; ns++; }
        if (ns == 0) continue;
        if (min_off > 0.0) ns = orc_support(segs, ns, min_off);    /* Track::support :911-941; segments are start-sorted by construction */
        if (min_on > 0) {                                           /* removeShort skips index 0, :943-953 */
            long w = 1;
            for (long i = 1; i < ns; ++i) if (!((segs[i].end - segs[i].start) < min_on)) segs[w++] = segs[i];
            ns = w;
        }
        for (long i = 0; i < ns; ++i) { if (nout < cap) out[nout] = segs[i]; nout++; }
    }
    free(ts); free(segs);
    if (nout <= cap) orc_final_sort(out, nout);       /* finalResult: std::sort by start, sd.cpp:973 (final_sort.cpp) */
    return nout;
}

/* a18 print format, sd.cpp:3439 (iostream default = %g, 6 significant) */
int orc_format_turn(const orc_turn *t, char *buf, int cap)
{
    return snprintf(buf, (size_t)cap, "[%g -- %g] --> Speaker_%d", t->start, t->end, t->label);
}

/* ------------------------------------------------------------------ */
/* a1 RIFF/WAVE reader, wav.h:62-126 + scale sd.cpp:2945-2951            */
/* returns malloc'd float[n] (caller frees with orc_free) or NULL.       */
/* ------------------------------------------------------------------ */
float *orc_read_wav(const char *path, long *n_out, int *channels, int *rate, int *bits)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    unsigned char h[44];
    if (fread(h, 1, 44, fp) != 44) { fclose(fp); return NULL; }
    uint32_t fmt_size; memcpy(&fmt_size, h + 16, 4);
    uint16_t ch, bit; uint32_t sr, dsz; char tag[4];
    memcpy(&ch, h + 22, 2); memcpy(&sr, h + 24, 4); memcpy(&bit, h + 34, 2);
    memcpy(tag, h + 36, 4); memcpy(&dsz, h + 40, 4);
    if (fmt_size < 16) { fclose(fp); return NULL; }
    if (fmt_size > 16) {                                            /* wav.h:75-79 */
        fseek(fp, 44 - 8 + (long)fmt_size - 16, SEEK_SET);
        unsigned char t8[8];
        if (fread(t8, 1, 8, fp) != 8) { fclose(fp); return NULL; }
        memcpy(tag, t8, 4); memcpy(&dsz, t8 + 4, 4);
    }
    while (strncmp(tag, "data", 4) != 0) {                          /* wav.h:85-90 */
        fseek(fp, (long)dsz, SEEK_CUR);
        unsigned char t8[8];
        if (fread(t8, 1, 8, fp) != 8) { fclose(fp); return NULL; }
        memcpy(tag, t8, 4); memcpy(&dsz, t8 + 4, 4);
    }
    if (bit != 8 && bit != 16 && bit != 32) { fclose(fp); return NULL; }
    long num = (long)(dsz / (bit / 8));
    float *data = (float *)malloc(sizeof(float) * (size_t)(num > 0 ? num : 1));
    for (long i = 0; i < num; ++i) {
        float v = 0.0f;
        if (bit == 8) { signed char s = 0; if (fread(&s, 1, 1, fp) != 1) s = 0; v = (float)s; }
        else if (bit == 16) { int16_t s = 0; if (fread(&s, 1, 2, fp) != 2) s = 0; v = (float)s; }
        else { int32_t s = 0; if (fread(&s, 1, 4, fp) != 4) s = 0; v = (float)s; }
        data[i] = (float)((double)(v * 1.0f) / 32768.0);             /* sd.cpp:2950 */
    }
    fclose(fp);
    *n_out = num / (ch ? ch : 1);                                   /* wav.h:97 */
    if (channels) *channels = ch; if (rate) *rate = (int)sr; if (bits) *bits = bit;
    return data;
}
void orc_free(void *p) { free(p); }
