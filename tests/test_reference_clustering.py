"""The clustering stage pinned on the REFERENCE'S OWN Python: tests/golden/ref_clustering.npz was minted by
tools/mint_reference_fixtures_clustering.py from the method definitions of /root/reference/clustering/Clustering.py (filter_embeddings :46-78,
set_num_clusters :21-43, constrained_argmax :81-94, assign_embeddings :97-164, __call__ :167-238, AgglomerativeClustering.cluster :278-428)
executed on the real scipy with the hyper-parameters the C++ hard-codes (cosine / centroid / 0.7153814381597874 / 15, sd.cpp:2049-2056).
CPU: the C oracle against the vectors; GPU: sd_clustering_ex of libsdhip against the same vectors.

Conventions that differ and how the test maps them:
  NaN rows     the Python's argmax over a row of NaN scores is 0 (np.argmax), the C++ port gives them cluster 0 too (a14, sd.cpp:2191-2207);
               with constrained_argmax the Python first replaces NaN by the smallest score
  unassigned   constrained_argmax leaves -2 where a chunk has more speakers than clusters; compared as it is
  soft scores  2 - cosine distance to the centroid of the UN-normalised train embeddings: equal to 1e-9 (the C++ sums in another order)
Known deviations of the C++ (sd.cpp:332: float32 L2 norm in front of the linkage; small / large centroids of the re-assignment
from un-normalised embeddings) do not move a label on these well-separated sets -- which is what is asserted.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import orc


@pytest.fixture(scope="module")
def gold(golden_dir):
    path = os.path.join(golden_dir, "ref_clustering.npz")
    want = open(os.path.join(golden_dir, "ref_clustering.sha256")).read().split()[0]
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == want, "fixture and manifest disagree: re-mint both"
    return np.load(path)


def _case(gold, name):
    emb = gold["%s_q" % name].astype(np.float64) / 64.0
    emb[gold["%s_nan" % name]] = np.nan
    nc, mn, mx, constrained = [int(v) for v in gold["%s_kw" % name]]
    return emb, {"num_clusters": nc, "min_clusters": mn, "max_clusters": mx}, bool(constrained), gold["%s_hard" % name], gold["%s_soft" % name]


CASES = ["four_large_two_small", "recut_to_three", "recut_to_six", "at_most_two", "at_least_five", "constrained_assignment", "tiny_recording",
         "no_large_cluster", "one_speaker"]


def test_fixture_holds_the_cases_the_tests_name(gold):
    assert list(gold["cases"]) == CASES
    # the regimes the cases were built for did occur in the reference's run
    assert gold["four_large_two_small_soft"].shape[2] == 4                     # two small clusters re-assigned to large ones
    assert gold["recut_to_three_soft"].shape[2] == 3 and gold["recut_to_six_soft"].shape[2] == 6 and gold["at_most_two_soft"].shape[2] == 2
    assert gold["at_least_five_soft"].shape[2] == 4                            # "Found only 4 clusters": the dendrogram has no cut with 5 large ones
    assert gold["no_large_cluster_soft"].shape[2] == 1 and gold["one_speaker_soft"].shape[2] == 1 and gold["tiny_recording_soft"].shape[2] == 2
    assert (gold["constrained_assignment_hard"] == -2).sum() == 0 and gold["constrained_assignment_soft"].shape[2] == 3


@pytest.mark.parametrize("name", CASES)
def test_oracle_clustering_against_reference_python(gold, name):
    emb, kw, constrained, hard, soft = _case(gold, name)
    h, K, s = orc.clustering_full(emb, constrained=constrained, **kw)
    assert K == soft.shape[2]
    assert np.array_equal(h, hard), (name, int((h != hard).sum()))
    live = ~np.isnan(emb[:, :, 0])
    np.testing.assert_allclose(s[live], soft[live], rtol=0, atol=1e-9)
    if not constrained:
        # the plain entry point (a10 + a11 + a14) gives the same labels
        h2, K2, _ = orc.clustering(emb, **kw)
        assert K2 == K and np.array_equal(h2, hard)
    if constrained:
        # every chunk's speakers sit in different clusters, and at least one chunk differs from the plain arg-max
        assert all(len(set(r.tolist())) == 3 for r in h)
        assert (h != np.argmax(np.nan_to_num(soft, nan=-1.0), axis=2)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_clustering_against_reference_python(diarizer, gold, name):
    emb, kw, constrained, hard, soft = _case(gold, name)
    diarizer.set_option("constrained_assignment", int(constrained))
    try:
        h, K = diarizer.clustering(emb, **kw)
    finally:
        diarizer.set_option("constrained_assignment", 0)
    assert K == soft.shape[2] and np.array_equal(h, hard)
