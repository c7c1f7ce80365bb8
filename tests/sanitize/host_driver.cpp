// Sanitizer driver for the HOST side of the product (set-up only code: csrc/kkt_assembly.cpp, csrc/ordering.cpp,
// csrc/symbolic.cpp), built by tests/test_host_sanitizers.py with g++ -fsanitize=address,undefined (the GPU box offers
// no AddressSanitizer; the kernels are covered by the parity suite, the host code by this).  No HIP, no device.
//
// Input file (little-endian): int64 n, m, ncones, nnzP, nnzA, nd_leaf_size; int64 Pp[n+1], Pi[nnzP]; double Px[nnzP];
// int64 Ap[n+1], Ai[nnzA]; double Ax[nnzA]; int32 kinds[ncones]; int64 dims[ncones].
// For each file and ordering: assemble the KKT pattern and its maps, check the maps against the pattern, run the
// symbolic analysis, check the permutation and the schedule, print one line.
#include "kkt_assembly.hpp"
#include "symbolic.hpp"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

using namespace hipkkt;

template <class T>
static std::vector<T> rd(FILE* f, size_t n)
{
    std::vector<T> v(n);
    if (n && std::fread(v.data(), sizeof(T), n, f) != n) throw std::runtime_error("short read");
    return v;
}

static void require(bool ok, const char* what)
{
    if (!ok) throw std::runtime_error(std::string("check failed: ") + what);
}

int main(int argc, char** argv)
{
    try {
        for (int a = 1; a < argc; ++a) {
            FILE* f = std::fopen(argv[a], "rb");
            if (!f) { std::perror(argv[a]); return 2; }
            const auto h = rd<int64_t>(f, 6);
            const int64_t n = h[0], m = h[1], nc = h[2], nnzP = h[3], nnzA = h[4];
            const auto Pp = rd<int64_t>(f, n + 1), Pi = rd<int64_t>(f, nnzP);
            const auto Px = rd<double>(f, nnzP);
            const auto Ap = rd<int64_t>(f, n + 1), Ai = rd<int64_t>(f, nnzA);
            const auto Ax = rd<double>(f, nnzA);
            const auto kinds = rd<int32_t>(f, nc);
            const auto dims = rd<int64_t>(f, nc);
            std::fclose(f);

            KKTAssembly K;
            assemble_kkt(n, m, Pp.data(), Pi.data(), Px.data(), Ap.data(), Ai.data(), Ax.data(), nc, kinds.data(), dims.data(), 0, K);
            require(K.N == K.n + K.m + K.p && (int64_t)K.colptr.size() == (int64_t)K.N + 1 && K.colptr[K.N] == K.nnzK, "KKT shape");
            require((int64_t)K.mapP.size() == nnzP && (int64_t)K.mapA.size() == nnzA && (int)K.map_diag.size() == K.N, "map sizes");
            for (int64_t j = 0; j < K.N; ++j)
                for (int64_t q = K.colptr[j]; q < K.colptr[j + 1]; ++q)
                    require(K.rowval[q] >= 0 && K.rowval[q] <= j && (q == K.colptr[j] || K.rowval[q - 1] < K.rowval[q]), "triu, sorted columns");
            for (int v : K.mapP) require(v >= 0 && v < K.nnzK, "mapP range");
            for (int v : K.mapA) require(v >= 0 && v < K.nnzK, "mapA range");
            for (int v : K.mapHs) require(v >= 0 && v < K.nnzK, "mapHs range");
            for (int j = 0; j < K.N; ++j) require(K.rowval[K.map_diag[j]] == j && K.map_diag[j] == K.colptr[j + 1] - 1, "map_diag");
            for (int v : K.mapU) require(v >= 0 && v < K.nnzK, "mapU range");
            for (int v : K.mapV) require(v >= 0 && v < K.nnzK, "mapV range");
            for (int v : K.mapD) require(v >= 0 && v < K.nnzK, "mapD range");
            for (int s : K.dsigns) require(s == 1 || s == -1, "dsigns");

            std::vector<int64_t> rowval(K.rowval.begin(), K.rowval.end());
            for (int ordering : {ORDER_ND, ORDER_AMD, ORDER_NATURAL}) {
                if (ordering == ORDER_NATURAL && K.N > 30000) continue;
                SymbolicOptions opt;
                opt.ordering = ordering;
                if (h[5] > 0) opt.nd_leaf_size = (int)h[5];
                Symbolic S;
                analyse(K.N, K.colptr.data(), rowval.data(), 0, opt, S);
                require(S.N == K.N && (int)S.perm.size() == K.N && (int)S.iperm.size() == K.N, "perm size");
                for (int j = 0; j < K.N; ++j) require(S.perm[j] >= 0 && S.perm[j] < K.N && S.iperm[S.perm[j]] == j, "perm is a permutation");
                require((int)S.sn_start.size() == S.nsuper + 1 && S.sn_start[0] == 0 && S.sn_start[S.nsuper] == K.N, "supernode partition");
                int seen = 0;
                for (size_t l = 0; l < S.levels.size(); ++l)
                    for (int t = S.levels[l].begin; t < S.levels[l].end; ++t, ++seen) {
                        const int sn = S.level_sn[t];
                        require(sn >= 0 && sn < S.nsuper && S.sn_level[sn] == (int)l, "level lists");
                        const int par = S.sn_parent[sn];
                        require(par == -1 || (par > sn && S.sn_level[par] > (int)l), "parents later and higher");
                        const int fp = par < 0 ? 0 : (S.sn_start[par + 1] - S.sn_start[par]) + (int)(S.rowptr[par + 1] - S.rowptr[par]);
                        for (int64_t q = S.rowptr[sn]; q < S.rowptr[sn + 1]; ++q) {
                            require(S.rows[q] >= S.sn_start[sn + 1] && S.rows[q] < K.N && (q == S.rowptr[sn] || S.rows[q - 1] < S.rows[q]), "row structure");
                            require(par >= 0 && S.rel[q] >= 0 && S.rel[q] < fp, "relative indices inside the parent's front");
                        }
                    }
                require(seen == S.nsuper, "every supernode scheduled once");
                std::printf("%s ordering %d: N %d nnzK %lld supernodes %d levels %zu nnzL_stored %lld max_front %d\n", argv[a], ordering, K.N,
                            (long long)K.nnzK, S.nsuper, S.levels.size(), (long long)S.nnzL, S.max_front);
            }
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host_driver: %s\n", e.what());
        return 1;
    }
    std::printf("HOST SANITIZER DRIVER OK\n");
    return 0;
}
