#!/usr/bin/env python3
"""Mints golden vectors for the clustering stage (a10, a11, a14 and the constrained variants of f3) from the REFERENCE'S OWN Python,
/root/reference/clustering/Clustering.py -- the pyannote.audio pipeline code the C++ `Cluster` class (sd.cpp:2049-2420) was ported from -- run
on the real scipy (`linkage` / `fcluster` / `cdist` / `linear_sum_assignment`: the Python's third-party dependency, present in this image).

The file cannot be imported as a module (its classes derive from pyannote.pipeline's `Pipeline`, which this container lacks, it never imports
`random` / `typing`, and a mis-indented docstring makes it a syntax error as a whole), so the script cuts the method definitions filter_embeddings
(:46-78), set_num_clusters (:21-43), constrained_argmax (:81-94), assign_embeddings (:97-164), __call__ (:167-238) and
AgglomerativeClustering.cluster (:278-428) out of it by their lines, AS THEY STAND (see method_source for the one docstring that has to go), and
executes them in a namespace holding numpy, scipy and einops; they become the methods of a plain object carrying the hyper-parameters
the C++ hard-codes: metric "cosine", method "centroid", threshold 0.7153814381597874, min_cluster_size 15 (sd.cpp:2049-2056).  What runs is
the reference's code, unedited; nothing of it is written anywhere.  Output: tests/golden/ref_clustering.npz + .sha256 (inputs and the
reference's outputs); tests/test_reference_clustering.py checks the oracle (CPU) and sd_clustering_ex (GPU) against it.

Known deviations of the C++ port (sd.cpp:332): it normalises with a float32 L2 norm and takes the small / large cluster
centroids from the UN-normalised embeddings.  On the well-separated sets minted here neither moves a label; soft scores agree to 1e-6.
"""
import ast
import hashlib
import os
import random
from typing import Tuple

import numpy as np
from einops import rearrange
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import cdist, pdist

SRC = "/root/reference/clustering/Clustering.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ref_clustering.npz")
THRESHOLD = 0.7153814381597874             # sd.cpp:2049
MIN_CLUSTER_SIZE = 15                      # sd.cpp:2050


def method_source(lines, name):
    """the text of method `name` as it stands in the reference file: from its `def` line (4 spaces deep) to the line before the next
    definition at that depth.  The file as a whole does not parse -- the docstring of `cluster` (:284-303) sits one column left of the body it
    documents, an IndentationError -- so the methods are cut out by their lines; for `cluster` the docstring lines, and only those, are
    dropped (documentation, no code), everything else runs character for character"""
    start = next(i for i, l in enumerate(lines) if l.startswith("    def %s(" % name))
    end = next((i for i in range(start + 1, len(lines)) if lines[i].startswith("    def ") or lines[i].startswith("class ")), len(lines))
    body = lines[start:end]
    if name == "cluster":
        q = [i for i, l in enumerate(body) if l.strip().startswith('"""')]
        assert len(q) == 2 and q[0] < q[1], "docstring of cluster not where it was"
        assert all("=" not in l and "return" not in l for l in body[q[0]:q[1] + 1]), "docstring of cluster holds code?"
        body = body[:q[0]] + body[q[1] + 1:]
    return "".join(l[4:] if l.startswith("    ") else l for l in body)


def reference_clusterer(constrained):
    lines = open(SRC).read().splitlines(keepends=True)
    want = ["set_num_clusters", "filter_embeddings", "constrained_argmax", "assign_embeddings", "__call__", "cluster"]
    ns = {"np": np, "rearrange": rearrange, "fcluster": fcluster, "linkage": linkage, "linear_sum_assignment": linear_sum_assignment, "cdist": cdist,
          "pdist": pdist, "random": random, "Tuple": Tuple, "SlidingWindowFeature": object}
    for n in want:
        exec(compile(ast.parse(method_source(lines, n), SRC), SRC, "exec"), ns)
    Ref = type("ReferenceAgglomerativeClustering", (), {n: ns[n] for n in want})
    r = Ref()
    r.metric, r.method, r.threshold, r.min_cluster_size = "cosine", "centroid", THRESHOLD, MIN_CLUSTER_SIZE
    r.max_num_embeddings, r.constrained_assignment = np.inf, constrained
    return r


def planted(rng, chunks, sizes, noise, nan_frac, dim=192):
    """sizes[k] items around centre k (orthogonal-ish random directions), values on a 1/64 grid (exact in float32), some rows NaN"""
    n = 3 * chunks
    cen = rng.standard_normal((len(sizes), dim))
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    lab = np.concatenate([np.full(s, k) for k, s in enumerate(sizes)])
    assert len(lab) <= n
    lab = np.concatenate([lab, rng.integers(0, 2, n - len(lab))])           # the rest joins the first two (large) clusters
    rng.shuffle(lab)
    x = 6.0 * cen[lab] + noise * rng.standard_normal((n, dim))
    q = np.round(x * 64.0).astype(np.int16)
    nan = rng.random(n) < nan_frac
    return q.reshape(chunks, 3, dim), nan.reshape(chunks, 3)


def main():
    if not os.path.exists(SRC):
        raise SystemExit("reference tree absent: fixtures can only be minted in the build container")
    rng = np.random.default_rng(20261005)
    out = {}
    cases = []
    # (name, chunks, cluster sizes, noise, NaN fraction, kwargs of __call__, constrained assignment)
    spec = [("four_large_two_small", 90, [70, 60, 50, 40, 9, 5], 0.25, 0.08, {}, False),
            ("recut_to_three", 90, [70, 60, 50, 40, 9, 5], 0.25, 0.08, {"num_clusters": 3}, False),
            ("recut_to_six", 90, [70, 60, 50, 40, 20, 16], 0.25, 0.05, {"num_clusters": 6}, False),
            ("at_most_two", 90, [70, 60, 50, 40, 9, 5], 0.25, 0.08, {"max_clusters": 2}, False),
            ("at_least_five", 90, [70, 60, 50, 40, 9, 5], 0.25, 0.08, {"min_clusters": 5}, False),
            ("constrained_assignment", 60, [60, 50, 40], 0.3, 0.1, {}, True),
            ("tiny_recording", 7, [9, 8], 0.2, 0.1, {}, False),                  # min_cluster_size shrinks to round(0.1 * n)
            ("no_large_cluster", 54, [10] * 16, 0.1, 0.0, {}, False),            # every cluster below min_cluster_size -> one cluster
            ("one_speaker", 40, [120], 0.3, 0.2, {}, False)]
    for name, chunks, sizes, noise, nanf, kw, constrained in spec:
        q, nan = planted(rng, chunks, sizes, noise, nanf)
        emb = q.astype(np.float64) / 64.0
        emb[nan] = np.nan
        ref = reference_clusterer(constrained)
        hard, soft = ref(emb.copy(), **kw)
        hard = np.asarray(hard).astype(np.int64)
        out["%s_q" % name] = q
        out["%s_nan" % name] = nan
        out["%s_hard" % name] = hard
        out["%s_soft" % name] = np.asarray(soft, np.float64)
        out["%s_kw" % name] = np.array([kw.get("num_clusters", -1), kw.get("min_clusters", -1), kw.get("max_clusters", -1), int(constrained)], np.int64)
        cases.append(name)
        print("%-26s items %4d live %4d  K = %d  cluster sizes %s" % (name, hard.size, int((~nan).sum()), soft.shape[2], np.bincount(hard[~nan][hard[~nan] >= 0]).tolist()))
    out["cases"] = np.array(cases)
    np.savez_compressed(OUT, **out)
    h = hashlib.sha256(open(OUT, "rb").read()).hexdigest()
    open(OUT.replace(".npz", ".sha256"), "w").write("%s  ref_clustering.npz\n" % h)
    print("wrote %s (%d bytes), sha256 %s" % (OUT, os.path.getsize(OUT), h))


if __name__ == "__main__":
    main()
