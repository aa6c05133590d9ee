// seam4_binding.cpp -- the reference-side binding of INTEGRATION.md "Seam 4", kept compilable: the bodies a maintainer
// puts in place of pipeline/src/clustering/clustering.cpp so that Clustering::cluster / linkage / fcluster
// (clustering.h:7-11) run on libsdhip.  tests/test_abi.py compiles this file against the REFERENCE's own clustering.h
// (when /root/reference is present) and against include/sdhip.h: the signatures on both sides have to agree.
// TEST INFRASTRUCTURE (never linked into libsdhip.so).
#include <vector>
#include "clustering.h"
#include "sdhip.h"

extern sd_ctx* g_ctx;          // created once by the host program (sd_create)

std::vector<int> Clustering::cluster(const std::vector<std::vector<double>>& input, double cutoff)
{
    const int64_t N = (int64_t)input.size();
    const int d = N ? (int)input[0].size() : 0;
    std::vector<double> X((size_t)N * d);
    for (int64_t i = 0; i < N; ++i) std::copy(input[i].begin(), input[i].end(), X.begin() + i * d);
    std::vector<int32_t> T((size_t)N);
    sd_cluster(g_ctx, X.data(), N, d, cutoff, T.data());      // 1-based labels, same numbering as fcluster
    return std::vector<int>(T.begin(), T.end());
}

void Clustering::linkage(const std::vector<std::vector<double>>& input, std::vector<std::vector<double>>& dendrogram)
{
    const int64_t N = (int64_t)input.size();
    const int d = N ? (int)input[0].size() : 0;
    std::vector<double> X((size_t)N * d), Z((size_t)(N > 1 ? N - 1 : 0) * 4);
    for (int64_t i = 0; i < N; ++i) std::copy(input[i].begin(), input[i].end(), X.begin() + i * d);
    sd_linkage(g_ctx, X.data(), N, d, Z.data());
    dendrogram.assign((size_t)(N > 1 ? N - 1 : 0), std::vector<double>(4));
    for (size_t k = 0; k + 1 < (size_t)N; ++k) for (int q = 0; q < 4; ++q) dendrogram[k][q] = Z[k * 4 + q];
}

void Clustering::fcluster(const std::vector<std::vector<double>>& Z, double cutoff, std::vector<int>& clusters)
{
    const int64_t N = (int64_t)Z.size() + 1;                   // clustering.cpp:445
    std::vector<double> flat(Z.size() * 4);
    for (size_t k = 0; k < Z.size(); ++k) for (int q = 0; q < 4; ++q) flat[k * 4 + q] = Z[k][q];
    std::vector<int32_t> T((size_t)N);
    sd_fcluster(g_ctx, flat.data(), N, cutoff, T.data());      // host arithmetic only; 1-based labels in the reference's numbering
    clusters.assign(T.begin(), T.end());
}
