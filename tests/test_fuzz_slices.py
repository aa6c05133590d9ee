"""Seeded slices of the differential fuzzers as tests (two linkage defects were found by tools/linkage_fuzz.py alone in round 5;
nothing of the fuzzers ran under `pytest -m gpu`).  Each slice runs the tool itself for a bounded time with a fixed seed -- random sizes, data families
(clustered, lattice = ties everywhere, duplicates, collinear, 1e-150 / 1e120 scales), forced geometries (2 .. 100 workgroups, 128 .. 1024 threads = 2 .. 16
waves), all four linkage kernels -- against the C oracle, Z `array_equal` every time; the non-neural stages likewise."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, seconds, seed, min_runs):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seconds), str(seed)], capture_output=True, text=True, timeout=seconds + 240)
    tail = "\n".join(out.stdout.splitlines()[-12:])
    assert out.returncode == 0, tail + "\n" + out.stderr[-2000:]
    m = re.search(r"runs (\d+) failures (\d+)", out.stdout)
    assert m and int(m.group(2)) == 0 and int(m.group(1)) >= min_runs, tail
    return out.stdout


@pytest.mark.gpu
def test_linkage_fuzz_slice():
    """25 s of tools/linkage_fuzz.py, seed 7: every kernel route of run_linkage (k_linkage_rg, k_linkage_mw, the zero phase, k_linkage_hx in its 16- and
    32-bit forms, k_linkage_heap) is taken at least once"""
    text = _run("linkage_fuzz.py", 25, 7, 40)
    counts = {k: int(v) for k, v in re.findall(r"^(linkage_\w+) (\d+)$", text, re.M)}
    assert counts.get("linkage_rg_launches", 0) > 0 and counts.get("linkage_hx_jobs", 0) > 0 and counts.get("linkage_tie_fallbacks", 0) > 0, counts
    assert counts.get("linkage_hx_failed", 0) == 0, counts


@pytest.mark.gpu
def test_stage_fuzz_slice():
    """14 s of tools/stage_fuzz.py, seed 7: post-segmentation, clustering (NaN rows, duplicates, too few live rows) and reconstruction + annotation"""
    _run("stage_fuzz.py", 14, 7, 16)


@pytest.mark.gpu
def test_linkage_with_duplicates_between_34k_and_65k_rows_takes_the_replay_kernel():
    """from ~34 200 to 65 535 rows the 16-bit form of k_linkage_hx asked for more LDS than a CU has, every cooperative launch was
    refused and a job with ties (a 4-hour recording with duplicated rows) fell through to the one-workgroup heap kernel.  Those sizes now take the 32-bit form:
    40 000 clustered rows with 1 % duplicates go through the zero phase (replay kernel, then k_linkage_rg), nothing is refused, and Z equals the
    one-workgroup replay of the reference's heap bit for bit."""
    import sdhip
    rng = np.random.default_rng(5)
    N, dd = 40000, 32
    cen = rng.standard_normal((4, dd))
    X = cen[rng.integers(0, 4, N)] + 0.6 * rng.standard_normal((N, dd))
    q = rng.integers(0, N, N // 100)
    X[q] = X[rng.integers(0, N, len(q))]
    X = np.ascontiguousarray(X, np.float64)
    d = sdhip.Diarizer(None, None)
    try:
        d.set_option("profile", 1)
        d.reset_stats()
        Z = d.linkage(X)
        assert d.kernel_stats("linkage_hx_failed")["launches"] == 0 and d.kernel_stats("linkage_hx_refused")["launches"] == 0
        assert d.kernel_stats("linkage_zero_phase_jobs")["launches"] + d.kernel_stats("linkage_hx_jobs")["launches"] >= 1
        d.set_option("linkage_tie_kernel", 0)          # the last resort: k_linkage_heap, one workgroup, the reference's heap
        Zh = d.linkage(X)
    finally:
        d.close()
    assert np.array_equal(Z, Zh)
