// ref_shim.cpp -- C-ABI driver around the REFERENCE's own clustering.cpp so the
// restatement in ../sd_oracle.c can be validated against the real thing.
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes
// the reference header where it lies (-I/root/reference/pipeline/src/clustering)
// and is linked with the reference's clustering.cpp compiled in place.
// (The header forgets <vector>; the Makefile passes `-include vector`.)
#include <cstddef>
#include <vector>
#include "clustering.h"

extern "C" {

// Clustering::linkage (clustering.cpp:417-440): X[N][d] -> Z[N-1][4]
void ref_linkage(const double* X, long N, int d, double* Z)
{
    std::vector<std::vector<double>> in(N, std::vector<double>(d));
    for (long i = 0; i < N; ++i) for (int q = 0; q < d; ++q) in[i][q] = X[i * d + q];
    std::vector<std::vector<double>> z;
    Clustering::linkage(in, z);
    for (std::size_t i = 0; i < z.size(); ++i) for (int q = 0; q < 4; ++q) Z[i * 4 + q] = z[i][q];
}

// Clustering::fcluster (clustering.cpp:442-457): Z[N-1][4] -> T[N] (1-based)
void ref_fcluster(const double* Z, long N, double cutoff, int* T)
{
    std::vector<std::vector<double>> z(N - 1, std::vector<double>(4));
    for (long i = 0; i < N - 1; ++i) for (int q = 0; q < 4; ++q) z[i][q] = Z[i * 4 + q];
    std::vector<int> t;
    Clustering::fcluster(z, cutoff, t);
    for (long i = 0; i < N; ++i) T[i] = t[i];
}

// Clustering::cluster (clustering.cpp:459-468)
void ref_cluster(const double* X, long N, int d, double cutoff, int* T)
{
    std::vector<std::vector<double>> in(N, std::vector<double>(d));
    for (long i = 0; i < N; ++i) for (int q = 0; q < d; ++q) in[i][q] = X[i * d + q];
    std::vector<int> t = Clustering::cluster(in, cutoff);
    for (long i = 0; i < N; ++i) T[i] = t[i];
}

}
