"""ctypes binding of oracle/libsd_oracle.so (C restatement of the reference's
non-neural stages), oracle/_ref/libref_clustering.so (the reference's own
clustering.cpp) and oracle/_ref/libref_glue.so (the reference's own speakerDiarizer.cpp glue).  TEST INFRASTRUCTURE ONLY -- imported by tests/, smoke() and
bench.py's cpu_baseline leg; never by the product path."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
c_fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
c_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
c_lp = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
c_bp = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


# ---- constants of the reference ----
SR = 16000
WINDOW = 80000
STEP = 8000
FRAMES = 293
SPEAKERS = 3
ONSET = 0.4442333667381752
MIN_OFF_F32 = float(np.float32(0.5817029604921046))       # sd.cpp:3210 (float)
THRESH_F32 = float(np.float32(0.7153814381597874))         # sd.cpp:2049 (float)
MIN_CLUSTER_SIZE = 15
MIN_SAMPLES = 640
FRAME_STEP = 0.016875
EMB_BATCH = 32


class Turn(C.Structure):
    _fields_ = [("start", C.c_double), ("end", C.c_double), ("label", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libsd_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.orc_np_rint.restype = C.c_int
    L.orc_np_rint.argtypes = [C.c_double]
    L.orc_closest_frame.restype = C.c_long
    L.orc_closest_frame.argtypes = [C.c_double] * 4
    L.orc_num_chunks.restype = C.c_long
    L.orc_num_chunks.argtypes = [C.c_long, C.c_long, C.c_long, C.POINTER(C.c_long)]
    L.orc_crop.argtypes = [c_fp, C.c_long, C.c_long, C.c_long, c_fp]
    L.orc_binarize.argtypes = [c_fp, C.c_long, C.c_int, C.c_int, C.c_double, C.c_int, c_dp]
    L.orc_aggregate.restype = C.c_long
    L.orc_aggregate.argtypes = [c_dp, C.c_long, C.c_int, C.c_int] + [C.c_double] * 6 + [C.c_int, C.c_void_p, C.c_long]
    L.orc_speaker_count.restype = C.c_long
    L.orc_speaker_count.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_ip, C.c_long, c_dp]
    L.orc_select_masks.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_fp]
    L.orc_mask_compact.restype = C.c_long
    L.orc_mask_compact.argtypes = [c_fp, c_fp, C.c_int, C.c_long, C.c_float, c_fp]
    L.orc_wav_lens.argtypes = [c_lp, C.c_int, C.c_long, c_fp, c_bp, C.POINTER(C.c_int)]
    L.orc_pdist.argtypes = [c_dp, C.c_long, C.c_int, c_dp]
    L.orc_linkage_centroid.argtypes = [c_dp, C.c_long, c_dp]
    L.orc_fcluster_distance.argtypes = [c_dp, C.c_long, C.c_double, c_ip]
    L.orc_ahc_labels.argtypes = [c_dp, C.c_long, C.c_int, C.c_double, c_ip, C.c_void_p]
    L.orc_cluster_embeddings.restype = C.c_int
    L.orc_cluster_embeddings.argtypes = [c_dp, C.c_long, C.c_int, C.c_float, C.c_long, c_ip]
    L.orc_clustering.restype = C.c_int
    L.orc_clustering.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, C.c_float, C.c_long, c_ip, C.c_void_p, C.POINTER(C.c_long)]
    L.orc_clustering_ex.restype = C.c_int
    L.orc_clustering_ex.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, C.c_float, C.c_long, C.c_int, C.c_int, C.c_int, c_ip, C.c_void_p, C.POINTER(C.c_long)]
    L.orc_clustering_full.restype = C.c_int
    L.orc_clustering_full.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, C.c_float, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, c_ip, C.c_void_p,
                                      C.POINTER(C.c_long), C.c_void_p, C.c_long]
    L.orc_linear_sum_assignment.restype = C.c_int
    L.orc_linear_sum_assignment.argtypes = [c_dp, C.c_int, C.c_int, C.c_int, c_ip, c_ip]
    L.orc_constrained_argmax.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_ip]
    L.orc_cluster_embeddings_ex.restype = C.c_int
    L.orc_cluster_embeddings_ex.argtypes = [c_dp, C.c_long, C.c_int, C.c_float, C.c_long, C.c_int, C.c_int, C.c_int, c_ip]
    L.orc_mark_inactive.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_ip]
    L.orc_reconstruct.restype = C.c_long
    L.orc_reconstruct.argtypes = [c_fp, C.c_long, C.c_int, C.c_int, c_ip, c_ip, C.c_long, c_dp, C.c_long,
                                  C.c_long, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.orc_to_annotation.restype = C.c_long
    L.orc_to_annotation.argtypes = [c_dp, C.c_long, C.c_int] + [C.c_double] * 7 + [C.POINTER(Turn), C.c_long]
    L.orc_window_start.restype = C.c_double
    L.orc_window_start.argtypes = [C.c_double, C.c_double, C.c_long, C.c_int]
    L.orc_range_to_segment.argtypes = [C.c_double, C.c_double, C.c_double, C.c_long, C.c_long, c_dp]
    L.orc_support.restype = C.c_long
    L.orc_support.argtypes = [C.POINTER(Turn), C.c_long, C.c_double]
    L.orc_format_turn.restype = C.c_int
    L.orc_format_turn.argtypes = [C.POINTER(Turn), C.c_char_p, C.c_int]
    L.orc_read_wav.restype = C.c_void_p
    L.orc_read_wav.argtypes = [C.c_char_p, C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_free.argtypes = [C.c_void_p]
    _LIB = L
    return L


def ref():
    """the reference's own clustering.cpp (None when oracle/_ref is absent)"""
    global _REF
    if _REF is not None:
        return _REF
    path = os.path.join(_HERE, "_ref", "libref_clustering.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.ref_linkage.argtypes = [c_dp, C.c_long, C.c_int, c_dp]
    R.ref_fcluster.argtypes = [c_dp, C.c_long, C.c_double, c_ip]
    R.ref_cluster.argtypes = [c_dp, C.c_long, C.c_int, C.c_double, c_ip]
    _REF = R
    return R


_REFWAV = None


def ref_wav_read(path):
    """the reference's own WavReader (frontend/wav.h, oracle/_ref/libref_wav.so): raw sample values as WavReader::data()
    holds them (not yet divided by 32768), sample rate, channels, bits.  None when oracle/_ref is absent."""
    global _REFWAV
    if _REFWAV is None:
        p = os.path.join(_HERE, "_ref", "libref_wav.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_wav_read.restype = C.c_long
        R.ref_wav_read.argtypes = [C.c_char_p, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _REFWAV = R
    sr, ch, bits = C.c_int(0), C.c_int(0), C.c_int(0)
    n = _REFWAV.ref_wav_read(path.encode(), None, 0, C.byref(sr), C.byref(ch), C.byref(bits))
    if n < 0:
        raise IOError("reference WavReader cannot open " + path)
    out = np.zeros(max(n, 1), np.float32)
    _REFWAV.ref_wav_read(path.encode(), out.ctypes.data_as(C.c_void_p), n, None, None, None)
    return out[:n], sr.value, ch.value, bits.value


_REFGLUE = None


def refglue():
    """the reference's own speakerDiarizer.cpp glue (oracle/_ref/libref_glue.so: every ORT-free line range of that file,
    compiled unedited -- see ref_build/make_glue_tu.sh).  None when oracle/_ref is absent."""
    global _REFGLUE
    if _REFGLUE is not None:
        return _REFGLUE
    p = os.path.join(_HERE, "_ref", "libref_glue.so")
    if not os.path.exists(p):
        return None
    R = C.CDLL(p)
    R.ref_np_rint.restype = C.c_int
    R.ref_np_rint.argtypes = [C.c_double]
    R.ref_closest_frame.restype = C.c_long
    R.ref_closest_frame.argtypes = [C.c_double] * 4
    R.ref_argsort.argtypes = [c_dp, C.c_long, c_ip]
    R.ref_argmax.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_ip]
    R.ref_cosine_cdist.argtypes = [c_dp, C.c_long, c_dp, C.c_long, C.c_int, c_dp]
    R.ref_binarize.argtypes = [c_fp, C.c_long, C.c_int, C.c_int, c_dp]
    R.ref_crop.restype = C.c_long
    R.ref_crop.argtypes = [c_fp, C.c_long, C.c_double, c_fp, C.c_long]
    R.ref_clean_segmentations.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_dp]
    R.ref_select_masks.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_fp]
    R.ref_speaker_count.restype = C.c_long
    R.ref_speaker_count.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, C.c_long, c_ip, C.c_long, c_dp]
    R.ref_aggregate.restype = C.c_long
    R.ref_aggregate.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_long,
                                C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_long]
    R.ref_embedding_inputs.restype = C.c_int
    R.ref_embedding_inputs.argtypes = [c_fp, c_fp, C.c_int, C.c_int, C.c_long, c_fp, c_fp, c_bp]
    R.ref_cluster_embeddings.restype = C.c_int
    R.ref_cluster_embeddings.argtypes = [c_dp, C.c_long, C.c_int, c_ip]
    R.ref_clustering.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_ip]
    R.ref_mark_inactive.argtypes = [c_dp, C.c_long, C.c_int, C.c_int, c_ip]
    R.ref_reconstruct.restype = C.c_long
    R.ref_reconstruct.argtypes = [c_fp, C.c_long, C.c_int, C.c_int, c_ip, c_ip, C.c_long, c_dp, C.c_long, C.c_void_p, C.c_long,
                                  C.POINTER(C.c_int), c_dp]
    R.ref_to_annotation.restype = C.c_long
    R.ref_to_annotation.argtypes = [c_dp, C.c_long, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(Turn), C.c_long]
    R.ref_support.restype = C.c_long
    R.ref_support.argtypes = [C.POINTER(Turn), C.c_long, C.c_double]
    R.ref_finalize.restype = C.c_long
    R.ref_finalize.argtypes = [c_fp, C.c_long, C.c_int, C.c_int, c_dp, C.c_int, C.c_long, C.POINTER(Turn), C.c_long, C.POINTER(C.c_int)]
    _REFGLUE = R
    return R


class RefGlue:
    """numpy-level view of refglue(): the same call shapes as the oracle functions of this module, so that a test reads
    `orc.f(x) == orc.RefGlue().f(x)`.  Every method runs the REFERENCE's compiled C++."""

    def __init__(self):
        self.R = refglue()
        if self.R is None:
            raise RuntimeError("oracle/_ref/libref_glue.so is absent (built only where /root/reference exists)")

    def np_rint(self, v):
        return self.R.ref_np_rint(float(v))

    def closest_frame(self, t, start=0.0, step=FRAME_STEP, dur=FRAME_STEP):
        return self.R.ref_closest_frame(start, step, dur, float(t))

    def argsort(self, v):
        v = np.ascontiguousarray(v, np.float64)
        idx = np.zeros(len(v), np.int32)
        self.R.ref_argsort(v, len(v), idx)
        return idx

    def argmax(self, soft):
        soft = np.ascontiguousarray(soft, np.float64)
        c, S, K = soft.shape
        out = np.zeros((c, S), np.int32)
        self.R.ref_argmax(soft, c, S, K, out)
        return out

    def cosine_cdist(self, A, B):
        A, B = np.ascontiguousarray(A, np.float64), np.ascontiguousarray(B, np.float64)
        out = np.zeros((len(A), len(B)), np.float64)
        self.R.ref_cosine_cdist(A, len(A), B, len(B), A.shape[1], out)
        return out

    def binarize(self, seg):
        seg = np.ascontiguousarray(seg, np.float32)
        c, F, K = seg.shape
        out = np.empty((c, F, K), np.float64)
        self.R.ref_binarize(seg, c, F, K, out)
        return out

    def crop(self, wav, start_sample):
        wav = np.ascontiguousarray(wav, np.float32)
        out = np.zeros(WINDOW + 16, np.float32)
        n = self.R.ref_crop(wav, len(wav), start_sample / float(SR), out, len(out))
        return out[:n].copy()

    def clean_segmentations(self, b):
        b = np.ascontiguousarray(b, np.float64)
        c, F, K = b.shape
        out = np.empty((c, F, K), np.float64)
        self.R.ref_clean_segmentations(b, c, F, K, out)
        return out

    def select_masks(self, b):
        b = np.ascontiguousarray(b, np.float64)
        c, F, K = b.shape
        m = np.empty((c * K, F), np.float32)
        self.R.ref_select_masks(b, c, F, K, m)
        return m

    def speaker_count(self, b, num_samples):
        b = np.ascontiguousarray(b, np.float64)
        c, F, K = b.shape
        cap = c * 40 + 400
        cnt = np.zeros(cap, np.int32)
        win = np.zeros(4, np.float64)
        nf = self.R.ref_speaker_count(b, c, F, K, num_samples, cnt, cap, win)
        assert 0 <= nf <= cap
        return cnt[:nf].copy(), win

    def aggregate(self, scores, sf_start, sf_step, sf_dur, sf_ns, fr_step=FRAME_STEP, fr_dur=FRAME_STEP, missing=np.nan,
                  skip_average=False):
        scores = np.ascontiguousarray(scores, np.float64)
        c, F, K = scores.shape
        nf = self.R.ref_aggregate(scores, c, F, K, sf_start, sf_step, sf_dur, sf_ns, fr_step, fr_dur, missing, int(skip_average), None, 0)
        out = np.empty((nf, K), np.float64)
        self.R.ref_aggregate(scores, c, F, K, sf_start, sf_step, sf_dur, sf_ns, fr_step, fr_dur, missing, int(skip_average),
                             out.ctypes.data, nf)
        return out

    def embedding_inputs(self, wavs, masks):
        """-> (signals [B][L], wav_lens [B], too_short [B], all_nan)"""
        wavs, masks = np.ascontiguousarray(wavs, np.float32), np.ascontiguousarray(masks, np.float32)
        B, L = wavs.shape
        sig = np.zeros((B, L), np.float32)
        lens = np.zeros(B, np.float32)
        ts = np.zeros(B, np.uint8)
        an = self.R.ref_embedding_inputs(wavs, masks, B, masks.shape[1], L, sig, lens, ts)
        return sig, lens, ts.astype(bool), bool(an)

    def cluster_embeddings(self, X):
        X = np.ascontiguousarray(X, np.float64)
        lab = np.zeros(len(X), np.int32)
        K = self.R.ref_cluster_embeddings(X, len(X), X.shape[1], lab)
        return lab, K

    def clustering(self, emb):
        emb = np.ascontiguousarray(emb, np.float64)
        c, S, d = emb.shape
        hard = np.zeros((c, S), np.int32)
        self.R.ref_clustering(emb, c, S, d, hard)
        return hard

    def mark_inactive(self, b, hard):
        b = np.ascontiguousarray(b, np.float64)
        c, F, S = b.shape
        hard = np.ascontiguousarray(hard, np.int32).copy()
        self.R.ref_mark_inactive(b, c, F, S, hard)
        return hard

    def reconstruct(self, seg, hard, count, cwin, n_samples):
        seg = np.ascontiguousarray(seg, np.float32)
        c, F, S = seg.shape
        hard = np.ascontiguousarray(hard, np.int32)
        count = np.ascontiguousarray(count, np.int32)
        cwin = np.ascontiguousarray(cwin, np.float64)
        K = C.c_int(0)
        fr = np.zeros(3, np.float64)
        rows = self.R.ref_reconstruct(seg, c, F, S, hard, count, len(count), cwin, n_samples, None, 0, C.byref(K), fr)
        out = np.zeros((rows, max(K.value, 1)), np.float64)
        self.R.ref_reconstruct(seg, c, F, S, hard, count, len(count), cwin, n_samples, out.ctypes.data, rows, C.byref(K), fr)
        return out, fr

    def to_annotation(self, binary, start, step=FRAME_STEP, dur=FRAME_STEP):
        b = np.ascontiguousarray(binary, np.float64)
        rows, K = b.shape
        cap = rows * K + 8
        buf = (Turn * cap)()
        n = self.R.ref_to_annotation(b, rows, K, start, step, dur, buf, cap)
        return [(buf[i].start, buf[i].end, buf[i].label) for i in range(n)]

    def support(self, segs, collar=None):
        arr = (Turn * max(len(segs), 1))()
        for i, (a, b) in enumerate(segs):
            arr[i] = Turn(a, b, 0)
        n = self.R.ref_support(arr, len(segs), MIN_OFF_F32 if collar is None else collar)
        return [(arr[i].start, arr[i].end) for i in range(n)]

    def finalize(self, seg, emb, num_samples):
        """everything of speakerDiarization() behind the two networks -> (turns sorted as finalResult sorts them, K)"""
        seg = np.ascontiguousarray(seg, np.float32)
        c, F, S = seg.shape
        emb = np.ascontiguousarray(emb, np.float64).reshape(c, S, -1)
        cap = c * 8 + 64
        buf = (Turn * cap)()
        K = C.c_int(0)
        n = self.R.ref_finalize(seg, c, F, S, emb, emb.shape[2], num_samples, buf, cap, C.byref(K))
        assert n <= cap
        return [(buf[i].start, buf[i].end, buf[i].label) for i in range(n)], K.value


def np_rint(v):
    return lib().orc_np_rint(float(v))


def closest_frame(t, start=0.0, step=FRAME_STEP, dur=FRAME_STEP):
    return lib().orc_closest_frame(start, step, dur, float(t))


def num_chunks(n, window=WINDOW, step=STEP):
    ll = C.c_long(0)
    c = lib().orc_num_chunks(n, window, step, C.byref(ll))
    return c, ll.value


def crop(wav, start, L=WINDOW):
    wav = np.ascontiguousarray(wav, np.float32)
    out = np.empty(L, np.float32)
    lib().orc_crop(wav, len(wav), start, L, out)
    return out


def binarize(seg, onset=ONSET, initial_state=False):
    seg = np.ascontiguousarray(seg, np.float32)
    c, F, K = seg.shape
    out = np.empty((c, F, K), np.float64)
    lib().orc_binarize(seg, c, F, K, onset, int(initial_state), out)
    return out


def aggregate(scores, sf_start, sf_step, sf_dur, fr_step=FRAME_STEP, fr_dur=FRAME_STEP, missing=np.nan,
              skip_average=False):
    scores = np.ascontiguousarray(scores, np.float64)
    c, F, K = scores.shape
    nf = lib().orc_aggregate(scores, c, F, K, sf_start, sf_step, sf_dur, fr_step, fr_dur, missing,
                             int(skip_average), None, 0)
    out = np.empty((nf, K), np.float64)
    lib().orc_aggregate(scores, c, F, K, sf_start, sf_step, sf_dur, fr_step, fr_dur, missing,
                        int(skip_average), out.ctypes.data, nf)
    return out


def speaker_count(binarized):
    b = np.ascontiguousarray(binarized, np.float64)
    c, F, K = b.shape
    win = np.zeros(3, np.float64)
    cap = c * 40 + 400
    cnt = np.zeros(cap, np.int32)
    nf = lib().orc_speaker_count(b, c, F, K, cnt, cap, win)
    assert nf >= 0
    F_trim = F - 2 * int(np.floor(F * 0.1))
    return cnt[:nf].copy(), win, F_trim


def select_masks(binarized):
    b = np.ascontiguousarray(binarized, np.float64)
    c, F, K = b.shape
    m = np.empty((c * K, F), np.float32)
    lib().orc_select_masks(b, c, F, K, m)
    return m


def mask_compact(chunk, mask, thr=0.5):
    chunk = np.ascontiguousarray(chunk, np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    sig = np.empty(len(chunk), np.float32)
    n = lib().orc_mask_compact(chunk, mask, len(mask), len(chunk), thr, sig)
    return sig, n


def wav_lens(counts, min_samples=MIN_SAMPLES):
    counts = np.ascontiguousarray(counts, np.int64)
    B = len(counts)
    lens = np.empty(B, np.float32)
    ts = np.empty(B, np.uint8)
    an = C.c_int(0)
    lib().orc_wav_lens(counts, B, min_samples, lens, ts, C.byref(an))
    return lens, ts.astype(bool), bool(an.value)


def pdist(X):
    X = np.ascontiguousarray(X, np.float64)
    N, d = X.shape
    D = np.empty(N * (N - 1) // 2, np.float64)
    lib().orc_pdist(X, N, d, D)
    return D


def linkage_centroid(D, N):
    D = np.ascontiguousarray(D, np.float64)
    Z = np.zeros((N - 1, 4), np.float64)
    lib().orc_linkage_centroid(D, N, Z)
    return Z


def fcluster_distance(Z, cutoff):
    Z = np.ascontiguousarray(Z, np.float64)
    N = Z.shape[0] + 1
    T = np.zeros(N, np.int32)
    lib().orc_fcluster_distance(Z, N, cutoff, T)
    return T


def ahc(X, cutoff):
    """Clustering::cluster equivalent: returns (labels1, Z)"""
    X = np.ascontiguousarray(X, np.float64)
    N, d = X.shape
    T = np.zeros(N, np.int32)
    Z = np.zeros((max(N - 1, 0), 4), np.float64)
    lib().orc_ahc_labels(X, N, d, cutoff, T, Z.ctypes.data)
    return T, Z


def cluster_embeddings(X, threshold=THRESH_F32, min_cluster_size=MIN_CLUSTER_SIZE):
    X = np.ascontiguousarray(X, np.float64)
    N, d = X.shape
    lab = np.zeros(N, np.int32)
    K = lib().orc_cluster_embeddings(X, N, d, threshold, min_cluster_size, lab)
    return lab, K


def clustering(emb, threshold=THRESH_F32, min_cluster_size=MIN_CLUSTER_SIZE, num_clusters=-1, min_clusters=-1, max_clusters=-1):
    """emb [c][S][d] float64 with NaN rows -> hard [c][S], K, train_labels"""
    emb = np.ascontiguousarray(emb, np.float64)
    c, S, d = emb.shape
    hard = np.zeros((c, S), np.int32)
    tl = np.zeros(c * S, np.int32)
    nt = C.c_long(0)
    K = lib().orc_clustering_ex(emb, c, S, d, threshold, min_cluster_size, num_clusters, min_clusters, max_clusters,
                                hard, tl.ctypes.data, C.byref(nt))
    return hard, K, tl[:nt.value].copy()


def clustering_full(emb, constrained=False, num_clusters=-1, min_clusters=-1, max_clusters=-1, threshold=THRESH_F32,
                    min_cluster_size=MIN_CLUSTER_SIZE):
    """like clustering(), optionally with constrained_argmax (Clustering.py:81-94); also returns soft [c][S][K]"""
    emb = np.ascontiguousarray(emb, np.float64)
    c, S, d = emb.shape
    hard = np.zeros((c, S), np.int32)
    tl = np.zeros(c * S, np.int32)
    nt = C.c_long(0)
    cap = c * S * 256
    soft = np.zeros(cap, np.float64)
    K = lib().orc_clustering_full(emb, c, S, d, threshold, min_cluster_size, num_clusters, min_clusters, max_clusters, int(constrained),
                                  hard, tl.ctypes.data, C.byref(nt), soft.ctypes.data, cap)
    return hard, K, soft[:c * S * max(K, 0)].reshape(c, S, max(K, 0)).copy()


def linear_sum_assignment(cost, maximize=False):
    cost = np.ascontiguousarray(cost, np.float64)
    nr, nc = cost.shape
    ri, ci = np.zeros(max(nr, nc), np.int32), np.zeros(max(nr, nc), np.int32)
    n = lib().orc_linear_sum_assignment(cost, nr, nc, int(maximize), ri, ci)
    return ri[:n].copy(), ci[:n].copy()


def constrained_argmax(soft):
    soft = np.ascontiguousarray(soft, np.float64)
    c, S, K = soft.shape
    hard = np.zeros((c, S), np.int32)
    lib().orc_constrained_argmax(soft, c, S, K, hard)
    return hard


def cluster_embeddings_ex(X, num_clusters=-1, min_clusters=-1, max_clusters=-1, threshold=THRESH_F32, min_cluster_size=MIN_CLUSTER_SIZE):
    X = np.ascontiguousarray(X, np.float64)
    N, d = X.shape
    lab = np.zeros(N, np.int32)
    K = lib().orc_cluster_embeddings_ex(X, N, d, threshold, min_cluster_size, num_clusters, min_clusters, max_clusters, lab)
    return lab, K


def mark_inactive(binarized, hard):
    b = np.ascontiguousarray(binarized, np.float64)
    c, F, S = b.shape
    hard = np.ascontiguousarray(hard, np.int32).copy()
    lib().orc_mark_inactive(b, c, F, S, hard)
    return hard


def reconstruct(seg, hard, count, cwin, cwin_ns, n_samples):
    seg = np.ascontiguousarray(seg, np.float32)
    c, F, S = seg.shape
    hard = np.ascontiguousarray(hard, np.int32)
    count = np.ascontiguousarray(count, np.int32)
    cwin = np.ascontiguousarray(cwin, np.float64)
    K = C.c_int(0)
    st = C.c_double(0)
    rows = lib().orc_reconstruct(seg, c, F, S, hard, count, len(count), cwin, cwin_ns, n_samples, None, 0,
                                 C.byref(K), C.byref(st))
    rows = -rows if rows < 0 else rows
    out = np.zeros((rows, max(K.value, 1)), np.float64)
    r2 = lib().orc_reconstruct(seg, c, F, S, hard, count, len(count), cwin, cwin_ns, n_samples,
                               out.ctypes.data, rows, C.byref(K), C.byref(st))
    assert r2 == rows
    return out, st.value


def to_annotation(binary, start, step=FRAME_STEP, dur=FRAME_STEP, onset=0.5, offset=0.5, min_on=0.0,
                  min_off=MIN_OFF_F32):
    b = np.ascontiguousarray(binary, np.float64)
    rows, K = b.shape
    cap = rows * K + 8
    buf = (Turn * cap)()
    n = lib().orc_to_annotation(b, rows, K, start, step, dur, onset, offset, min_on, min_off, buf, cap)
    return [(buf[i].start, buf[i].end, buf[i].label) for i in range(n)]


def window_start(pos, step=FRAME_STEP, dur=FRAME_STEP, num_samples=1 << 40):
    """SlidingWindow::operator[](pos).start, sd.cpp:1092-1115 (accumulates `start += step` from 0.0)"""
    return float(lib().orc_window_start(step, dur, num_samples, pos))


def range_to_segment(i0, n, start=0.0, step=FRAME_STEP, dur=FRAME_STEP):
    out = np.zeros(2)
    lib().orc_range_to_segment(start, step, dur, i0, n, out)
    return float(out[0]), float(out[1])


def support(segs, collar=None):
    """Track::support on start-sorted (start, end) pairs of one label"""
    arr = (Turn * max(len(segs), 1))()
    for i, (a, b) in enumerate(segs):
        arr[i] = Turn(a, b, 0)
    n = lib().orc_support(arr, len(segs), MIN_OFF_F32 if collar is None else collar)
    return [(arr[i].start, arr[i].end) for i in range(n)]


def format_turn(t):
    tt = Turn(t[0], t[1], t[2])
    buf = C.create_string_buffer(128)
    lib().orc_format_turn(C.byref(tt), buf, 128)
    return buf.value.decode()


def read_wav(path):
    n = C.c_long(0)
    ch = C.c_int(0)
    sr = C.c_int(0)
    bits = C.c_int(0)
    p = lib().orc_read_wav(path.encode(), C.byref(n), C.byref(ch), C.byref(sr), C.byref(bits))
    if not p:
        raise IOError("cannot read wav: " + path)
    total = n.value * max(ch.value, 1)
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(total,)).copy()
    lib().orc_free(p)
    return arr[:n.value], sr.value, ch.value, bits.value
