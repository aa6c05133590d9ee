"""CPU tests of the drop-in boundary: libsdhip.so loads without a GPU, exports every symbol that
include/sdhip.h declares, and its pure-host entry points agree with the oracle."""
import os
import re

import numpy as np
import pytest

import sdhip
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = sdhip.lib()
    decl = {}
    for h in ("sdhip.h", "sdhip_test.h"):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        decl[h] = set(re.findall(r"^(?:[a-z_0-9]+\*?\s+\*?)+(sd_[a-z0-9_]+)\s*\(", hdr, re.M))
    declared = decl["sdhip.h"] | decl["sdhip_test.h"]
    assert not (decl["sdhip.h"] & decl["sdhip_test.h"])
    assert declared == set(sdhip.EXPORTS), declared ^ set(sdhip.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    # the drop-in header holds reference-cited entries only: the test / bench / tuning hooks live in sdhip_test.h
    assert decl["sdhip_test.h"] == {"sd_set_planted", "sd_kernel_stats", "sd_reset_stats", "sd_bench_barrier", "sd_bench_conv", "sd_debug_read_ws", "sd_test_pack_split_weights", "sd_bench_linkage_parts"}


def test_create_fails_loudly_without_gpu_or_model(tmp_path):
    import torch
    if torch.cuda.is_available():
        with pytest.raises(sdhip.SdError) as e:
            sdhip.Diarizer(str(tmp_path / "missing.sdw"), None)
        assert "cannot open" in str(e.value)
    else:
        with pytest.raises(sdhip.SdError) as e:
            sdhip.Diarizer(None, None)
        assert "no HIP device" in str(e.value)        # no CPU fallback in the product path


@pytest.mark.parametrize("n", [1, 2, 79999, 80000, 80001, 88000, 100000, 944000, 9600000, 57600000, 57601234])
def test_chunk_rule_matches_oracle(n):
    assert sdhip.num_chunks(n) == orc.num_chunks(n)


@pytest.mark.parametrize("c", [1, 2, 7, 109, 1191, 7191])
def test_count_frames_matches_oracle(c):
    b = np.zeros((c, 293, 3))
    cnt, _, _ = orc.speaker_count(b)
    assert sdhip.lib().sd_count_frames(c) == len(cnt)


def test_format_turn_matches_reference_stream_format():
    for t in [(5.222812345, 17.74406789, 3), (0.4978125, 39.9516, 0), (1234.56789, 3599.991, 12), (1e-7, 100000.5, 1)]:
        assert sdhip.format_turn(t) == orc.format_turn(t)
    assert sdhip.format_turn((5.222812345, 17.74406789, 3)) == "[5.22281 -- 17.7441] --> Speaker_3"


def test_wav_reader_matches_oracle(golden_dir):
    pcm, sr, ch = sdhip.read_wav(os.path.join(golden_dir, "multi-speaker_1min.wav"))
    w, sr2, ch2, bits = orc.read_wav(os.path.join(golden_dir, "multi-speaker_1min.wav"))
    assert (sr, ch, len(pcm)) == (16000, 1, 944000) and sr2 == sr and bits == 16
    assert np.array_equal(pcm.astype(np.float32) / np.float32(32768.0), w)
    with pytest.raises(sdhip.SdError):
        sdhip.read_wav("/nonexistent.wav")


def test_cli_usage_line():
    import subprocess
    exe = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd", "speakerDiarizer")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0                                              # sd.cpp:3421-3425
    assert out.stdout.strip() == "program [segment model file] [embeding model file] [wave file]"


def test_synth_is_deterministic_and_prefix_consistent():
    import synth
    a = synth.make_pcm(30.0, seed=5)
    assert np.array_equal(a, synth.make_pcm(30.0, seed=5))
    assert np.array_equal(a[:100000], synth.make_pcm(30.0, seed=5, limit=100000))
    assert a.dtype == np.int16 and len(a) == 480000 and np.abs(a).max() > 5000


def test_product_path_never_touches_the_oracle():
    """oracle/ is the checker, never the thing shipped: no source of the package -- library, CLI, ctypes mirror, synthetic-data and weight-pack helpers --
    imports, opens or names anything under oracle/ (bench.py may, in its cpu_baseline leg only: checked there by its structure, not here), the shared
    library has no dependency on liboracle / libref objects, and nothing in the package reads /root/reference"""
    import subprocess
    pkg = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        if "__pycache__" in dirpath:
            continue
        for f in files:
            if not f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                continue
            text = open(os.path.join(dirpath, f), errors="replace").read()
            for needle in ("from oracle", "import oracle", "libsd_oracle", "libref_", "oracle/_ref", "/root/reference"):
                if needle in text:
                    offenders.append((f, needle))
    assert not offenders, offenders
    needed = subprocess.run(["readelf", "-d", os.path.join(pkg, "libsdhip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in needed and "libref" not in needed and "libtorch" not in needed and "onnxruntime" not in needed


def test_header_is_plain_c_and_links_from_a_c_program(tmp_path):
    """include/sdhip.h is the drop-in boundary: it must compile as C99 (no C++-isms), and a C program linked against
    libsdhip.so must be able to call the host-only entry points"""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include "sdhip.h"
#include "sdhip_test.h"
int main(void) {
    int64_t last = 0, ranges[4], per = 0;
    int64_t c = sd_num_chunks(944000, &last);
    sd_turn t[3] = {{5.222812345, 17.74406789, 10, 0}, {18.0, 19.0, 2, 0}, {20.0, 21.0, 10, 0}};
    char buf[128];
    sd_format_turn(&t[0], buf, (int)sizeof buf);
    if (sd_shard_plan(57600000, 2, -1, ranges, &per) != SD_OK) return 2;
    if (sd_relabel_turns_ex(t, 3, 0) != SD_OK) return 3;
    printf("%lld %lld|%s|%lld %lld %lld %lld %lld|%d %d %d|%d\n", (long long)c, (long long)last, buf, (long long)per, (long long)ranges[0],
           (long long)ranges[1], (long long)ranges[2], (long long)ranges[3], t[0].label, t[1].label, t[2].label, SD_COMM_ID_BYTES);
    return 0;
}
''')
    pkg = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", pkg, "-lsdhip", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "109 80000|[5.22281 -- 17.7441] --> Speaker_10|3616 0 3616 3616 7191|0 1 0|128"


def test_reference_side_binding_of_seam4_compiles_against_the_reference_header(tmp_path):
    """INTEGRATION.md seam 4 as a compilable file: Clustering::cluster / ::linkage / ::fcluster re-implemented on sd_cluster / sd_linkage / sd_fcluster must
    match the reference's own clustering.h (signatures, constness) and include/sdhip.h"""
    import subprocess
    ref = "/root/reference/pipeline/src/clustering"
    if not os.path.exists(os.path.join(ref, "clustering.h")):
        pytest.skip("reference tree absent (GPU box)")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-include", "vector", "-include", "algorithm", "-include", "cstdint", "-I", ref,
                           "-I", os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "oracle", "ref_build", "seam4_binding.cpp"), "-o", str(tmp_path / "seam4.o")])


def test_reference_side_bindings_of_the_two_literal_model_seams_compile(tmp_path):
    """INTEGRATION.md seams 2 and 3 in their literal form -- SegmentModel::infer (sd.cpp:1352) on sd_segment_chunks and EmbeddingModel1::infer
    (sd.cpp:1977) on sd_embed_signals, with the reference's own parameter and return types -- compile against include/sdhip.h, alone and
    as the test shims the GPU tests run (tests/test_next_rows.py)"""
    import subprocess
    pkg = os.path.join(ROOT, "pyannote-audio_speaker-diarization_cpp_amd")
    for seam in ("seam2", "seam3"):
        src = os.path.join(ROOT, "oracle", "ref_build", seam + "_binding.cpp")
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-include", "algorithm", "-include", "cstdint", "-I", os.path.join(ROOT, "include"),
                               "-c", src, "-o", str(tmp_path / (seam + ".o"))])
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-include", "algorithm", "-include", "cstdint", "-DSEAM_TEST_SHIM", "-fPIC", "-shared",
                               "-I", os.path.join(ROOT, "include"), src, "-o", str(tmp_path / ("lib" + seam + ".so")), "-L", pkg, "-lsdhip"])


def test_fcluster_entry_point_against_the_oracle_and_the_reference():
    """sd_fcluster = Clustering::fcluster (clustering.h:9-10, clustering.cpp:442-457) is host arithmetic: labels equal to the C oracle's and to
    the reference's own compiled fcluster on dendrograms with nested, flat and tied heights, at cutoffs below, between and above them; a Z
    that is not a dendrogram is refused instead of indexed with"""
    rng = np.random.default_rng(12)
    cases = []
    for N, d in ((2, 4), (3, 4), (40, 8), (700, 16)):
        X = rng.standard_normal((N, d)); X[N // 2:] += 3.0
        cases.append(orc.linkage_centroid(orc.pdist(X), N))
    g = np.stack(np.meshgrid(np.arange(4.0), np.arange(4.0)), -1).reshape(-1, 2)                 # lattice: equal heights everywhere
    cases.append(orc.linkage_centroid(orc.pdist(g), len(g)))
    R = orc.ref()
    for Z in cases:
        N = len(Z) + 1
        hs = np.sort(Z[:, 2])
        for cut in (hs[0] * 0.5, hs[len(hs) // 2], float(np.nextafter(hs[len(hs) // 2], -np.inf)), hs[-1], hs[-1] * 2, orc.THRESH_F32):
            T = sdhip.fcluster(Z, cut)
            assert np.array_equal(T, orc.fcluster_distance(Z, cut)), (N, cut)
            if R is not None:
                Tr = np.zeros(N, np.int32)
                R.ref_fcluster(np.ascontiguousarray(Z), N, float(cut), Tr)
                assert np.array_equal(T, Tr), (N, cut)
    assert np.array_equal(sdhip.fcluster(np.zeros((0, 4)), 1.0), [1])                             # one observation: one cluster
    bad = cases[2].copy(); bad[5, 0] = 10 ** 6
    with pytest.raises(sdhip.SdError):
        sdhip.fcluster(bad, 1.0)
    bad = cases[2].copy(); bad[7, 1] = bad[3, 1]                                                  # a node merged twice
    with pytest.raises(sdhip.SdError):
        sdhip.fcluster(bad, 1.0)


def test_split_weight_packing_of_the_x3_mode_on_the_host():
    """option ecapa_precision = 3 feeds the fp16 MFMA with hi + lo halves of both operands; the weight side is packed once on the host
    (weights.cpp pack_split_weights, exposed as sd_test_pack_split_weights): per layer a power-of-two scale that puts the largest weight
    into [2^13, 2^14), per 32-channel chunk the kernel's LDS row [hi 0..7 | lo 0..7 | hi 8..15 | ...], padding channels zero.
    hi + lo must give the weight back to 2^-21 relative for every weight down to 2^-17 of the largest (their lo halves are normal fp16
    numbers), with an ABSOLUTE error of at most 2^-25 of the scaled range below that -- for small, for huge and for mixed magnitudes."""
    import ctypes as C
    L = sdhip.lib()
    rng = np.random.default_rng(5)
    K, Cout, CinPad, cin = 3, 8, 96, 80
    for mag in (1.0, 0.02, 3e-7, 4e5):
        w = np.zeros((K, Cout, CinPad), np.float32)
        w[:, :, :cin] = (mag * rng.standard_normal((K, Cout, cin)) * np.exp(4.0 * rng.standard_normal((K, Cout, cin)))).astype(np.float32)
        w[0, 0, 0] = 0.0
        w[:, :, cin:] = 7.0                                               # channels beyond cin are ignored, whatever they hold
        out = np.zeros(2 * K * Cout * CinPad, np.uint16)
        inv = C.c_float(0.0)
        assert L.sd_test_pack_split_weights(w.ctypes.data_as(C.c_void_p), K, Cout, CinPad, cin, out.ctypes.data_as(C.c_void_p), C.byref(inv)) == 0
        wmax = np.abs(w[:, :, :cin]).max()
        scale = 1.0 / inv.value
        assert np.log2(scale) == np.round(np.log2(scale)) and 2.0 ** 13 <= wmax * scale < 2.0 ** 14
        h = out.view(np.float16).astype(np.float64).reshape(K, Cout, CinPad // 32, 4, 2, 8)      # [chunk][group of 8][hi | lo][channel]
        hi = h[..., 0, :].reshape(K, Cout, CinPad)
        lo = h[..., 1, :].reshape(K, Cout, CinPad)
        assert (hi[:, :, cin:] == 0).all() and (lo[:, :, cin:] == 0).all()
        assert np.array_equal(hi[:, :, :cin], (w[:, :, :cin].astype(np.float64) * scale).astype(np.float16).astype(np.float64))   # hi = fp16(w 2^e), nearest-even
        back = (hi + lo) * inv.value
        ref = w.astype(np.float64)
        err = np.abs(back - ref)[:, :, :cin]
        big = np.abs(ref[:, :, :cin]) >= wmax * 2.0 ** -17
        assert big.sum() > 100 and (~big).sum() > 5
        assert (err[big] <= 2.0 ** -21 * np.abs(ref[:, :, :cin][big])).all()
        assert (err[~big] <= 2.0 ** -25 * inv.value).all()                # subnormal lo halves: absolute error of half a subnormal step
    assert L.sd_test_pack_split_weights(w.ctypes.data_as(C.c_void_p), K, Cout, 80, 80, out.ctypes.data_as(C.c_void_p), C.byref(inv)) != 0      # CinPad % 32


def test_split_operand_arithmetic_of_the_x3_mode_is_f32_grade():
    """the arithmetic of option ecapa_precision = 3, modelled in numpy (no GPU): both operands of a dot product are split into hi = fp16(v) and
    lo = fp16(v - hi), the weights after the power-of-two scale of sd_test_pack_split_weights; hi*hi + lo*hi + hi*lo is accumulated in f32, the
    lo*lo term dropped.  Per product the error is 2^-21 of |a w| plus -- for activations below 2^-3, whose lo halves are fp16 subnormals --
    an absolute 2^-25 |w|: the bound asserted below.  For activations of order one and larger (what BatchNorm leaves in every ECAPA layer) that
    is the class of the plain f32 dot product, three orders below fp16 operands (mode 1); a layer whose activations were ALL tiny would fall
    back to the absolute term (the "small" case: still 20x better than fp16 operands, no longer f32-grade) -- DESIGN section 4 says so."""
    rng = np.random.default_rng(11)
    K, N = 3072, 64                                    # the MFA layer's contraction length

    def split(v):
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float32)).astype(np.float16)
        return hi.astype(np.float32), lo.astype(np.float32)

    res = {}
    for name, a_scale in (("unit", 1.0), ("small", 2.0 ** -9), ("large", 3.0e3)):
        a = (a_scale * np.abs(rng.standard_normal((N, K)))).astype(np.float32)                   # post-ReLU activations
        w = (rng.standard_normal((K,)) * np.exp(2.0 * rng.standard_normal((K,))) / np.sqrt(K)).astype(np.float32)
        e = int(np.floor(np.log2(np.abs(w).max()))) + 1
        sc = np.float32(2.0 ** (14 - e))                                                         # largest weight into [2^13, 2^14)
        ah, al = split(a)
        wh, wl = split(w * sc)
        exact = a.astype(np.float64) @ w.astype(np.float64)
        acc = np.zeros(N, np.float32)
        for k0 in range(0, K, 16):                                                               # one MFMA k-block at a time, f32 accumulator
            s_ = slice(k0, k0 + 16)
            for x, y in ((ah, wh), (al, wh), (ah, wl)):
                acc = (acc + (x[:, s_].astype(np.float64) @ y[s_].astype(np.float64)).astype(np.float32)).astype(np.float32)
        x3 = acc.astype(np.float64) / float(sc)
        f32 = np.zeros(N, np.float32)
        for k0 in range(0, K, 2):                                                                # v_mfma_f32_32x32x2_f32: two products per accumulation
            f32 = (f32 + (a[:, k0:k0 + 2].astype(np.float64) @ w[k0:k0 + 2].astype(np.float64)).astype(np.float32)).astype(np.float32)
        h16 = (a.astype(np.float16).astype(np.float64) @ w.astype(np.float16).astype(np.float64))
        aw = np.abs(a.astype(np.float64)) @ np.abs(w.astype(np.float64))                         # the magnitude the roundings are relative to
        bound = 2.0 ** -21 * aw + 2.0 ** -25 * np.abs(w.astype(np.float64)).sum() + 2.0 ** -22 * aw      # split terms + the f32 accumulation
        assert (np.abs(x3 - exact) <= bound).all(), name
        res[name] = (np.abs(x3 - exact) / aw).max(), (np.abs(f32 - exact) / aw).max(), (np.abs(h16 - exact) / aw).max()
    for name in ("unit", "large"):
        ex3, ef32, eh16 = res[name]
        assert ex3 < 4e-7 and ex3 < 8 * max(ef32, 2.0 ** -24) and eh16 > 100 * ex3, (name, res[name])
    assert res["small"][0] < 5e-6 and res["small"][2] > 20 * res["small"][0], res["small"]
