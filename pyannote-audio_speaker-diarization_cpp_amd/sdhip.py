"""ctypes mirror of include/sdhip.h -- the host-side binding the tests, bench.py and
__graft_entry__ use.  It only loads libsdhip.so (built in-tree next to this file) and
fails loudly when the library or the GPU is missing: there is no CPU fallback.

Names follow the reference's operator seams: segment ~ SegmentModel::slide,
embed ~ getEmbedding + EmbeddingModel1::infer, cluster ~ Clustering::cluster,
diarize ~ speakerDiarization()."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDHIP_LIB") or os.path.join(_HERE, "libsdhip.so")      # (SDHIP_LIB: instrumented builds of the tuning tools)

CHUNK, HOP, FRAMES, SPEAKERS, EMB_DIM, EMB_BATCH = 80000, 8000, 293, 3, 192, 32
T_FRAMES, N_MELS = 501, 80


class Turn(C.Structure):
    _fields_ = [("start", C.c_double), ("end", C.c_double), ("label", C.c_int32), ("_pad", C.c_int32)]


class SdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsdhip error %d: %s" % (code, msg))
        self.code = code


_lib = None

# every symbol include/sdhip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "sd_create", "sd_destroy", "sd_last_error", "sd_create_error", "sd_num_chunks", "sd_segment", "sd_segment_dev",
    "sd_postseg", "sd_count_frames", "sd_embed", "sd_embed_dev", "sd_frontend", "sd_ecapa", "sd_linkage", "sd_cluster",
    "sd_clustering", "sd_clustering_ex", "sd_reconstruct", "sd_diarize", "sd_diarize_dev", "sd_free_turns", "sd_shard_infer_dev",
    "sd_finalize_dev", "sd_read_wav", "sd_free_pcm", "sd_format_turn", "sd_stage_ms", "sd_kernel_stats",
    "sd_reset_stats", "sd_set_option", "sd_bench_conv", "sd_bench_barrier", "sd_convert_onnx", "sd_convert_error", "sd_read_wav_f32", "sd_free_wav", "sd_diarize_f32",
    "sd_write_rttm", "sd_set_planted", "sd_comm_unique_id", "sd_comm_init", "sd_comm_destroy", "sd_comm_info", "sd_shard_plan",
    "sd_diarize_sharded", "sd_diarize_sharded_dev", "sd_write_rttm_ex", "sd_relabel_turns", "sd_relabel_turns_ex", "sd_last_confidence",
    "sd_debug_read_ws", "sd_test_pack_split_weights", "sd_resample", "sd_resample_len", "sd_diarize_wav", "sd_set_dump_dir",
    "sd_fcluster", "sd_segment_chunks", "sd_embed_signals", "sd_bench_linkage_parts",
]
COMM_ID_BYTES = 128


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsdhip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or make -C '%s')" % _HERE)
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    L.sd_create.restype = vp
    L.sd_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.sd_destroy.argtypes = [vp]
    L.sd_last_error.restype = C.c_char_p
    L.sd_last_error.argtypes = [vp]
    L.sd_create_error.restype = C.c_char_p
    L.sd_num_chunks.restype = i64
    L.sd_num_chunks.argtypes = [i64, C.POINTER(i64)]
    L.sd_count_frames.restype = i64
    L.sd_count_frames.argtypes = [i64]
    L.sd_set_dump_dir.argtypes = [vp, C.c_char_p, C.c_int]
    L.sd_resample_len.restype = i64
    L.sd_resample_len.argtypes = [i64, i32, i32]
    L.sd_resample.argtypes = [vp, vp, i64, i32, i32, vp, i64, C.POINTER(i64)]
    L.sd_diarize_wav.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.POINTER(Turn)), C.POINTER(i64)]
    L.sd_segment.argtypes = [vp, vp, i64, vp, C.POINTER(i64)]
    L.sd_segment_dev.argtypes = [vp, vp, i64, vp, i64]
    L.sd_postseg.argtypes = [vp, vp, i64, vp, vp, vp, i64, C.POINTER(i64)]
    L.sd_embed.argtypes = [vp, vp, i64, vp, i64, vp]
    L.sd_embed_dev.argtypes = [vp, vp, i64, vp, i64, i64, vp]
    L.sd_frontend.argtypes = [vp, vp, i64, vp, i64, vp, vp]
    L.sd_ecapa.argtypes = [vp, vp, vp, i64, vp]
    L.sd_linkage.argtypes = [vp, vp, i64, C.c_int, vp]
    L.sd_fcluster.argtypes = [vp, vp, i64, dbl, vp]
    L.sd_segment_chunks.argtypes = [vp, vp, i64, i64, vp, C.POINTER(i32)]
    L.sd_embed_signals.argtypes = [vp, vp, vp, i64, vp]
    L.sd_cluster.argtypes = [vp, vp, i64, C.c_int, dbl, vp]
    L.sd_clustering.argtypes = [vp, vp, i64, C.c_int, vp, C.POINTER(i32)]
    L.sd_clustering_ex.argtypes = [vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(i32)]
    L.sd_reconstruct.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, C.POINTER(C.POINTER(Turn)), C.POINTER(i64)]
    L.sd_diarize.argtypes = [vp, vp, i64, C.POINTER(C.POINTER(Turn)), C.POINTER(i64)]
    L.sd_diarize_dev.argtypes = [vp, vp, i64, C.POINTER(C.POINTER(Turn)), C.POINTER(i64)]
    L.sd_free_turns.argtypes = [C.POINTER(Turn)]
    L.sd_shard_infer_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64, vp, vp]
    L.sd_finalize_dev.argtypes = [vp, vp, vp, i64, i64, C.POINTER(C.POINTER(Turn)), C.POINTER(i64)]
    L.sd_read_wav.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]
    L.sd_free_pcm.argtypes = [C.POINTER(C.c_int16)]
    L.sd_read_wav_f32.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.sd_free_wav.argtypes = [C.POINTER(C.c_float)]
    L.sd_diarize_f32.argtypes = [vp, vp, i64, C.POINTER(C.POINTER(Turn)), C.POINTER(i64)]
    L.sd_write_rttm.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Turn), i64]
    L.sd_set_planted.argtypes = [vp, vp, vp, i64, i64]
    L.sd_comm_unique_id.argtypes = [vp]
    L.sd_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.sd_comm_destroy.argtypes = [vp]
    L.sd_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sd_shard_plan.argtypes = [i64, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.sd_diarize_sharded.argtypes = [vp, vp, i64, i64, i64, C.POINTER(C.POINTER(Turn)), C.POINTER(i64)]
    L.sd_diarize_sharded_dev.argtypes = [vp, vp, i64, i64, i64, C.POINTER(C.POINTER(Turn)), C.POINTER(i64)]
    L.sd_write_rttm_ex.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Turn), i64, C.POINTER(dbl)]
    L.sd_relabel_turns.argtypes = [C.POINTER(Turn), i64]
    L.sd_relabel_turns_ex.argtypes = [C.POINTER(Turn), i64, C.c_int]
    L.sd_last_confidence.argtypes = [vp, C.POINTER(dbl), i64, C.POINTER(i64)]
    L.sd_format_turn.argtypes = [C.POINTER(Turn), C.c_char_p, C.c_int]
    L.sd_stage_ms.argtypes = [vp, C.POINTER(dbl)]
    L.sd_kernel_stats.argtypes = [vp, C.c_char_p, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl)]
    L.sd_reset_stats.argtypes = [vp]
    L.sd_set_option.argtypes = [vp, C.c_char_p, i64]
    L.sd_convert_onnx.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    L.sd_convert_error.restype = C.c_char_p
    L.sd_debug_read_ws.argtypes = [vp, C.c_char_p, i64, vp, i64]
    L.sd_test_pack_split_weights.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.sd_bench_barrier.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(dbl)]
    L.sd_bench_linkage_parts.argtypes = [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(dbl)]
    L.sd_bench_conv.argtypes = [vp, i64] + [C.c_int] * 9 + [C.POINTER(dbl)]
    _lib = L
    return L


def num_chunks(n):
    ll = C.c_int64(0)
    c = lib().sd_num_chunks(n, C.byref(ll))
    return int(c), int(ll.value)


def fcluster(Z, cutoff):
    """sd_fcluster = Clustering::fcluster (clustering.h:9-10): Z [N-1][4] -> 1-based labels [N].  Host arithmetic, no context, no GPU."""
    Z = np.ascontiguousarray(Z, np.float64).reshape(-1, 4)
    N = len(Z) + 1
    T = np.zeros(N, np.int32)
    rc = lib().sd_fcluster(None, _ptr(Z) if len(Z) else None, N, float(cutoff), _ptr(T))
    if rc:
        raise SdError(rc, "sd_fcluster: Z is not a dendrogram")
    return T


def shard_plan(n_total, world, rank0_permille=-1):
    """sd_shard_plan: (slot size in chunks of the padded all-gather, [(lo, hi)] by rank) -- contiguous chunk ranges, each
    starting on a multiple of 32 chunks (= 96 items = 3 reference embedding batches)"""
    arr = (C.c_int64 * (2 * world))()
    per = C.c_int64(0)
    rc = lib().sd_shard_plan(n_total, world, rank0_permille, arr, C.byref(per))
    if rc:
        raise SdError(rc, "sd_shard_plan: bad argument")
    return int(per.value), [(int(arr[2 * r]), int(arr[2 * r + 1])) for r in range(world)]


def plan_shards(n_total, world):
    """equal shares"""
    return shard_plan(n_total, world, -1)


def plan_ranks(n_total, world, rank0_fraction=None):
    """chunk range of every rank when rank 0, which also finalizes (count / clustering / reconstruction), is given a
    smaller share of the chunks: rank0_fraction of all chunks (0 = none), the other ranks split the rest evenly.
    None = equal shares.  Returns (per, [(lo, hi)] by rank): per = slot size of the padded all-gather."""
    if rank0_fraction is None or world == 1:
        return shard_plan(n_total, world, -1)
    return shard_plan(n_total, world, int(round(1000.0 * max(0.0, min(1.0, rank0_fraction)))))


def gather_pieces(per, ranges):
    """[(offset in the gathered buffer, chunks)] in chunk order: what rank 0 concatenates before finalizing"""
    return [(r * per, hi - lo) for r, (lo, hi) in enumerate(ranges) if hi > lo]


def shard_sample_range(lo, hi, n_total):
    """samples a rank must hold for chunks [lo, hi): [lo*8000, min(n, (hi-1)*8000 + 80000))"""
    if hi <= lo:
        return lo * HOP, lo * HOP
    return lo * HOP, min(n_total, (hi - 1) * HOP + CHUNK)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def convert_onnx(onnx_path, kind, out_path):
    """kind: 'segmentation' | 'embedding'.  Host-only (no GPU needed)."""
    rc = lib().sd_convert_onnx(onnx_path.encode(), 0 if kind.startswith('seg') else 1, out_path.encode())
    if rc:
        raise SdError(rc, lib().sd_convert_error().decode())


def format_turn(t):
    tt = Turn(t[0], t[1], t[2], 0)
    buf = C.create_string_buffer(128)
    lib().sd_format_turn(C.byref(tt), buf, 128)
    return buf.value.decode()


def read_wav(path):
    p = C.POINTER(C.c_int16)()
    n, sr, ch = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    rc = lib().sd_read_wav(path.encode(), C.byref(p), C.byref(n), C.byref(sr), C.byref(ch))
    if rc:
        raise SdError(rc, "cannot read wav " + path)
    total = n.value * max(ch.value, 1)
    arr = np.ctypeslib.as_array(p, shape=(total,)).copy()
    lib().sd_free_pcm(p)
    return arr[:n.value], sr.value, ch.value


def read_wav_f32(path):
    """8/16/32-bit PCM -> float32 samples / 32768 (reference scaling), sample_rate, channels, bits"""
    p = C.POINTER(C.c_float)()
    n, sr, ch, bits = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rc = lib().sd_read_wav_f32(path.encode(), C.byref(p), C.byref(n), C.byref(sr), C.byref(ch), C.byref(bits))
    if rc:
        raise SdError(rc, "cannot read wav " + path)
    total = n.value * max(ch.value, 1)
    arr = np.ctypeslib.as_array(p, shape=(total,)).copy()
    lib().sd_free_wav(p)
    return arr[:n.value], sr.value, ch.value, bits.value


def _turn_array(turns):
    arr = (Turn * max(len(turns), 1))()
    for i, t in enumerate(turns):
        arr[i] = Turn(t[0], t[1], t[2], 0)
    return arr


def write_rttm(path, uri, turns, conf=None):
    arr = _turn_array(turns)
    if conf is None:
        rc = lib().sd_write_rttm(path.encode(), uri.encode(), arr, len(turns))
    else:
        cc = (C.c_double * max(len(turns), 1))(*[float(x) for x in conf])
        rc = lib().sd_write_rttm_ex(path.encode(), uri.encode(), arr, len(turns), cc)
    if rc:
        raise SdError(rc, "cannot write " + path)


def relabel_turns(turns, mode="pyannote"):
    """'pyannote': labels that occur, sorted by their string, -> 0, 1, ... (SPEAKER_00 ...); 'first': order of first appearance"""
    arr = _turn_array(turns)
    rc = lib().sd_relabel_turns_ex(arr, len(turns), 1 if mode == "pyannote" else 0)
    if rc:
        raise SdError(rc, "sd_relabel_turns_ex: bad argument")
    return [(arr[i].start, arr[i].end, int(arr[i].label)) for i in range(len(turns))]


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().sd_comm_unique_id(buf)
    if rc:
        raise SdError(rc, "sd_comm_unique_id failed (RCCL needs a GPU)")
    return buf.raw


class Diarizer:
    """one libsdhip context on one GPU (not thread-safe, like the reference's OnnxModel statics)"""

    def __init__(self, seg_model=None, emb_model=None, device=0):
        L = lib()
        self._h = L.sd_create(seg_model.encode() if seg_model else None, emb_model.encode() if emb_model else None, device)
        if not self._h:
            raise SdError(-1, L.sd_create_error().decode())

    def close(self):
        if self._h:
            lib().sd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise SdError(rc, lib().sd_last_error(self._h).decode())

    def set_option(self, key, value):
        self._chk(lib().sd_set_option(self._h, key.encode(), int(value)))

    # ---- a2+a3
    def segment(self, wav):
        wav = np.ascontiguousarray(wav, np.float32)
        c, _ = num_chunks(len(wav))
        out = np.zeros((c, FRAMES, SPEAKERS), np.float32)
        cc = C.c_int64(0)
        self._chk(lib().sd_segment(self._h, _ptr(wav), len(wav), _ptr(out), C.byref(cc)))
        return out[:cc.value]

    # ---- a4-a6
    def postseg(self, seg):
        seg = np.ascontiguousarray(seg, np.float32)
        c = seg.shape[0]
        nb = np.zeros((c, FRAMES, SPEAKERS), np.uint8)
        masks = np.zeros((c * SPEAKERS, FRAMES), np.float32)
        cap = int(lib().sd_count_frames(c))
        count = np.zeros(max(cap, 1), np.int32)
        nc = C.c_int64(0)
        self._chk(lib().sd_postseg(self._h, _ptr(seg), c, _ptr(nb), _ptr(masks), _ptr(count), cap, C.byref(nc)))
        return nb, masks, count[:nc.value]

    def segment_chunks(self, chunks):
        """sd_segment_chunks = SegmentModel::infer as declared (sd.cpp:1352): [rows][T] separate waveforms -> ([rows][293][3], frames)"""
        chunks = np.ascontiguousarray(chunks, np.float32)
        rows, T = chunks.shape
        out = np.zeros((rows, FRAMES, SPEAKERS), np.float32)
        fr = C.c_int32(0)
        self._chk(lib().sd_segment_chunks(self._h, _ptr(chunks), rows, T, _ptr(out), C.byref(fr)))
        return out, int(fr.value)

    def embed_signals(self, signals, wav_lens):
        """sd_embed_signals = EmbeddingModel1::infer as declared (sd.cpp:1977): [B][80000] compacted signals + relative lengths -> [B][192]"""
        signals = np.ascontiguousarray(signals, np.float32)
        wav_lens = np.ascontiguousarray(wav_lens, np.float32)
        B, T = signals.shape
        assert T == 80000 and wav_lens.shape == (B,)
        out = np.zeros((B, EMB_DIM), np.float32)
        self._chk(lib().sd_embed_signals(self._h, _ptr(signals), _ptr(wav_lens), B, _ptr(out)))
        return out

    # ---- a6-a9
    def embed(self, wav, masks):
        wav = np.ascontiguousarray(wav, np.float32)
        masks = np.ascontiguousarray(masks, np.float32)
        items = masks.shape[0]
        out = np.zeros((items, EMB_DIM), np.float32)
        self._chk(lib().sd_embed(self._h, _ptr(wav), len(wav), _ptr(masks), items, _ptr(out)))
        return out

    def frontend(self, wav, masks):
        wav = np.ascontiguousarray(wav, np.float32)
        masks = np.ascontiguousarray(masks, np.float32)
        items = masks.shape[0]
        feats = np.zeros((items, T_FRAMES, N_MELS), np.float32)
        lens = np.zeros(items, np.float32)
        self._chk(lib().sd_frontend(self._h, _ptr(wav), len(wav), _ptr(masks), items, _ptr(feats), _ptr(lens)))
        return feats, lens

    def ecapa(self, feats, lens):
        feats = np.ascontiguousarray(feats, np.float32)
        lens = np.ascontiguousarray(lens, np.float32)
        items = feats.shape[0]
        out = np.zeros((items, EMB_DIM), np.float32)
        self._chk(lib().sd_ecapa(self._h, _ptr(feats), _ptr(lens), items, _ptr(out)))
        return out

    # ---- a12-a14
    def linkage(self, X):
        X = np.ascontiguousarray(X, np.float64)
        N, d = X.shape
        Z = np.zeros((max(N - 1, 0), 4), np.float64)
        self._chk(lib().sd_linkage(self._h, _ptr(X), N, d, _ptr(Z)))
        return Z

    def cluster(self, X, cutoff):
        X = np.ascontiguousarray(X, np.float64)
        N, d = X.shape
        T = np.zeros(N, np.int32)
        self._chk(lib().sd_cluster(self._h, _ptr(X), N, d, float(cutoff), _ptr(T)))
        return T

    def clustering(self, emb, num_clusters=-1, min_clusters=-1, max_clusters=-1):
        emb = np.ascontiguousarray(emb, np.float64)
        c, S, d = emb.shape
        assert S == SPEAKERS
        hard = np.zeros((c, S), np.int32)
        K = C.c_int32(0)
        self._chk(lib().sd_clustering_ex(self._h, _ptr(emb), c, d, num_clusters, min_clusters, max_clusters, _ptr(hard), C.byref(K)))
        return hard, int(K.value)

    # ---- a15-a17
    def reconstruct(self, seg, binarized, hard, count, n_samples):
        seg = np.ascontiguousarray(seg, np.float32)
        binarized = np.ascontiguousarray(binarized, np.uint8)
        hard = np.ascontiguousarray(hard, np.int32)
        count = np.ascontiguousarray(count, np.int32)
        p = C.POINTER(Turn)()
        n = C.c_int64(0)
        self._chk(lib().sd_reconstruct(self._h, _ptr(seg), _ptr(binarized), _ptr(hard), _ptr(count), len(count),
                                       seg.shape[0], n_samples, C.byref(p), C.byref(n)))
        return self._turns(p, n)

    def _turns(self, p, n):
        out = [(p[i].start, p[i].end, int(p[i].label)) for i in range(n.value)]
        if p:
            lib().sd_free_turns(p)
        return out

    # ---- whole path
    def diarize(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        p = C.POINTER(Turn)()
        n = C.c_int64(0)
        self._chk(lib().sd_diarize(self._h, _ptr(pcm), len(pcm), C.byref(p), C.byref(n)))
        return self._turns(p, n)

    def diarize_f32(self, wav):
        wav = np.ascontiguousarray(wav, np.float32)
        p = C.POINTER(Turn)()
        n = C.c_int64(0)
        self._chk(lib().sd_diarize_f32(self._h, _ptr(wav), len(wav), C.byref(p), C.byref(n)))
        return self._turns(p, n)

    def set_dump_dir(self, path, level=1):
        """sd_set_dump_dir: the reference's WRITE_DATA items as <path>/cpp_<item>.txt from the next whole-path call on (None = off)"""
        self._chk(lib().sd_set_dump_dir(self._h, str(path).encode() if path else None, level if path else 0))

    def diarize_wav(self, path, resample=False, downmix=False, assume_16k=False):
        """sd_diarize_wav: reader + sample-rate / channel handling + the whole path (SD_WAV_RESAMPLE = 1, SD_WAV_DOWNMIX = 2, SD_WAV_ASSUME_16K = 4)"""
        p = C.POINTER(Turn)()
        n = C.c_int64(0)
        self._chk(lib().sd_diarize_wav(self._h, str(path).encode(), (1 if resample else 0) | (2 if downmix else 0) | (4 if assume_16k else 0), C.byref(p), C.byref(n)))
        return self._turns(p, n)

    def resample(self, wav, in_sr, out_sr=16000):
        wav = np.ascontiguousarray(wav, np.float32)
        no = C.c_int64(0)
        self._chk(lib().sd_resample(self._h, _ptr(wav), len(wav), in_sr, out_sr, None, 0, C.byref(no)))
        out = np.zeros(max(no.value, 1), np.float32)
        self._chk(lib().sd_resample(self._h, _ptr(wav), len(wav), in_sr, out_sr, _ptr(out), len(out), C.byref(no)))
        return out[:no.value]

    def diarize_dev(self, d_pcm_ptr, n_samples):
        p = C.POINTER(Turn)()
        n = C.c_int64(0)
        self._chk(lib().sd_diarize_dev(self._h, C.c_void_p(d_pcm_ptr), n_samples, C.byref(p), C.byref(n)))
        return self._turns(p, n)

    def shard_infer_dev(self, d_pcm_shard_ptr, first_sample, shard_samples, n_total, chunk_lo, chunk_hi, d_seg_ptr, d_emb_ptr):
        self._chk(lib().sd_shard_infer_dev(self._h, C.c_void_p(d_pcm_shard_ptr), first_sample, shard_samples, n_total,
                                           chunk_lo, chunk_hi, C.c_void_p(d_seg_ptr), C.c_void_p(d_emb_ptr)))

    def comm_init(self, id_bytes, rank, world):
        assert len(id_bytes) == COMM_ID_BYTES
        self._chk(lib().sd_comm_init(self._h, C.c_char_p(id_bytes), rank, world))

    def comm_destroy(self):
        lib().sd_comm_destroy(self._h)

    def comm_info(self):
        r, w = C.c_int(0), C.c_int(0)
        lib().sd_comm_info(self._h, C.byref(r), C.byref(w))
        return r.value, w.value

    def diarize_sharded_dev(self, d_pcm_shard_ptr, first_sample, shard_samples, n_total):
        """collective over the ctx's RCCL communicator; turns on rank 0, [] elsewhere"""
        p = C.POINTER(Turn)()
        n = C.c_int64(0)
        self._chk(lib().sd_diarize_sharded_dev(self._h, C.c_void_p(d_pcm_shard_ptr or None), first_sample, shard_samples, n_total, C.byref(p), C.byref(n)))
        return self._turns(p, n)

    def diarize_sharded(self, pcm_shard, first_sample, n_total):
        pcm_shard = np.ascontiguousarray(pcm_shard, np.int16)
        p = C.POINTER(Turn)()
        n = C.c_int64(0)
        self._chk(lib().sd_diarize_sharded(self._h, _ptr(pcm_shard), first_sample, len(pcm_shard), n_total, C.byref(p), C.byref(n)))
        return self._turns(p, n)

    def last_confidence(self):
        n = C.c_int64(0)
        self._chk(lib().sd_last_confidence(self._h, None, 0, C.byref(n)))
        buf = (C.c_double * max(n.value, 1))()
        self._chk(lib().sd_last_confidence(self._h, buf, n.value, C.byref(n)))
        return np.array(buf[:n.value], np.float64)

    def finalize_dev(self, d_seg_ptr, d_emb_ptr, chunks, n_samples):
        p = C.POINTER(Turn)()
        n = C.c_int64(0)
        self._chk(lib().sd_finalize_dev(self._h, C.c_void_p(d_seg_ptr), C.c_void_p(d_emb_ptr), chunks, n_samples,
                                        C.byref(p), C.byref(n)))
        return self._turns(p, n)

    def set_planted(self, d_scores_ptr, d_emb_ptr, chunk_lo, chunks):
        """planted workload hook: device pointers (0 = none) for chunks [chunk_lo, chunk_lo + chunks)"""
        self._chk(lib().sd_set_planted(self._h, C.c_void_p(d_scores_ptr or None), C.c_void_p(d_emb_ptr or None), chunk_lo, chunks))

    # ---- measurement
    def stage_ms(self):
        a = (C.c_double * 4)()
        self._chk(lib().sd_stage_ms(self._h, a))
        return list(a)

    def kernel_stats(self, name):
        ms, n, fl, by = C.c_double(0), C.c_int64(0), C.c_double(0), C.c_double(0)
        self._chk(lib().sd_kernel_stats(self._h, name.encode(), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        return {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}

    def bench_barrier(self, G, iters=2000, dirty=0):
        us = C.c_double(0)
        self._chk(lib().sd_bench_barrier(self._h, G, iters, dirty, C.byref(us)))
        return us.value

    def bench_conv(self, items, Tp, T, Cin, Cout, KT=1, dil=1, has_x2=0, dbg=0, reps=5):
        ms = C.c_double(0)
        self._chk(lib().sd_bench_conv(self._h, items, Tp, T, Cin, Cout, KT, dil, has_x2, dbg, reps, C.byref(ms)))
        return ms.value

    def read_ws(self, name, dtype, count, offset=0):
        """test hook: `count` elements of the named device workspace"""
        out = np.zeros(count, dtype)
        self._chk(lib().sd_debug_read_ws(self._h, name.encode(), offset, _ptr(out), out.nbytes))
        return out

    def reset_stats(self):
        lib().sd_reset_stats(self._h)
