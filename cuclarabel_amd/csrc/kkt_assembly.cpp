// See kkt_assembly.hpp.  Built column by column with explicit per-column writers instead of the
// reference's count / fill / back-shift passes over colptr; the resulting layout is the same:
// within each column rows ascend and the diagonal is the last entry
// (directldl_kkt_assembly.jl:161-165).
#include "kkt_assembly.hpp"

#include <stdexcept>
#include <string>

#include "../../include/hipkkt.h"

namespace hipkkt {

static inline int64_t tri(int64_t k) { return k * (k + 1) / 2; }

void assemble_kkt(int64_t n64, int64_t m64, const int64_t* Pp, const int64_t* Pi, const double* Px,
                  const int64_t* Ap, const int64_t* Ai, const double* Ax, int64_t ncones,
                  const int32_t* kinds, const int64_t* dims, int base, KKTAssembly& K)
{
    K = KKTAssembly();
    if (n64 < 0 || m64 < 0 || n64 + m64 > 1900000000) throw std::runtime_error("kkt: bad dimensions");
    const int n = (int)n64, m = (int)m64;
    K.n = n;
    K.m = m;
    // ---- cone layout (compositecone_type.jl:114-141)
    int64_t off = 0, boff = 0;
    for (int64_t c = 0; c < ncones; ++c) {
        ConeInfo ci{};
        ci.kind = kinds[c];
        if (ci.kind < HIPKKT_CONE_ZERO || ci.kind > HIPKKT_CONE_PSD) throw std::runtime_error("kkt: unknown cone kind");
        if (dims[c] < 0) throw std::runtime_error("kkt: negative cone dimension");
        ci.dim = (int)dims[c];
        ci.numel = ci.kind == HIPKKT_CONE_PSD ? (int)tri(dims[c]) : (int)dims[c];
        if (ci.kind == HIPKKT_CONE_SOC && ci.dim < 2) throw std::runtime_error("kkt: second-order cone needs dim >= 2");
        ci.off = (int)off;
        off += ci.numel;
        ci.sparse = (ci.kind == HIPKKT_CONE_SOC && ci.dim > 4) ? 1 : 0;     // cone_types.jl:101-112
        ci.boff = boff;
        bool dense = ci.kind == HIPKKT_CONE_PSD || (ci.kind == HIPKKT_CONE_SOC && !ci.sparse);
        ci.blen = dense ? tri(ci.numel) : ci.numel;
        boff += ci.blen;
        if (ci.sparse) {
            ci.sidx = K.nsparse++;
            ci.soff = K.sparse_len;
            K.sparse_len += ci.numel;
        }
        K.cones.push_back(ci);
    }
    if (off != m) throw std::runtime_error("kkt: cone dimensions do not sum to m");
    K.nHs = boff;
    K.p = 2 * K.nsparse;
    const int N = K.N = n + m + K.p;

    // ---- the caller's CSC arrays: a wrong index_base or a malformed colptr must end as an argument error, not as
    // host writes out of bounds further down
    auto check_csc = [&](const char* what, const int64_t* cp, const int64_t* ri, const double* vx, int64_t nrows) {
        if (!cp) throw std::runtime_error(std::string("kkt: ") + what + " colptr is null");
        if (cp[0] != base) throw std::runtime_error(std::string("kkt: ") + what + " colptr[0] does not equal index_base");
        for (int j = 0; j < n; ++j)
            if (cp[j + 1] < cp[j]) throw std::runtime_error(std::string("kkt: ") + what + " colptr is not non-decreasing");
        const int64_t nnz = cp[n] - base;
        if (nnz >= ((int64_t)1 << 31)) throw std::runtime_error(std::string("kkt: ") + what + " has too many entries");
        if (nnz > 0 && (!ri || !vx)) throw std::runtime_error(std::string("kkt: ") + what + " rowval / nzval is null");
        for (int64_t q = 0; q < nnz; ++q) {
            const int64_t r = ri[q] - base;
            if (r < 0 || r >= nrows) throw std::runtime_error(std::string("kkt: ") + what + " row index out of range");
        }
    };
    check_csc("P", Pp, Pi, Px, n);
    check_csc("A", Ap, Ai, Ax, m);

    // ---- column lengths
    std::vector<int64_t> len((size_t)N, 0);
    const int64_t nnzP = Pp[n] - base, nnzA = Ap[n] - base;
    std::vector<char> has_diag((size_t)n, 0);
    for (int j = 0; j < n; ++j) {
        int64_t b = Pp[j] - base, e = Pp[j + 1] - base;
        for (int64_t q = b; q < e; ++q) {
            int64_t i = Pi[q] - base;
            if (i > j) throw std::runtime_error("kkt: P must be upper triangular");
            if (q > b && Pi[q] <= Pi[q - 1]) throw std::runtime_error("kkt: P rows must ascend within a column");
        }
        has_diag[j] = (e > b && Pi[e - 1] - base == j) ? 1 : 0;     // csc_assembly.jl:36-48
        len[j] = (e - b) + (has_diag[j] ? 0 : 1);
    }
    for (int64_t q = 0; q < nnzA; ++q) {
        int64_t r = Ai[q] - base;
        if (r < 0 || r >= m) throw std::runtime_error("kkt: A row index out of range");
        len[n + r] += 1;
    }
    {
        int pcol = n + m;
        for (const ConeInfo& ci : K.cones) {
            bool dense = ci.kind == HIPKKT_CONE_PSD || (ci.kind == HIPKKT_CONE_SOC && !ci.sparse);
            for (int t = 0; t < ci.numel; ++t) len[n + ci.off + t] += dense ? t + 1 : 1;
            if (ci.sparse) {
                len[pcol] += ci.numel + 1;
                len[pcol + 1] += ci.numel + 1;
                pcol += 2;
            }
        }
    }
    K.colptr.assign((size_t)N + 1, 0);
    for (int j = 0; j < N; ++j) K.colptr[j + 1] = K.colptr[j] + len[j];
    K.nnzK = K.colptr[N];
    if (K.nnzK >= ((int64_t)1 << 31)) throw std::runtime_error("kkt: nnz(K) exceeds int32 indexing");
    K.rowval.assign((size_t)K.nnzK, 0);
    K.nzval.assign((size_t)K.nnzK, 0.0);
    K.mapP.assign((size_t)nnzP, 0);
    K.mapA.assign((size_t)nnzA, 0);
    K.mapHs.assign((size_t)K.nHs, 0);
    K.map_diag.assign((size_t)N, 0);
    K.mapU.assign((size_t)K.sparse_len, 0);
    K.mapV.assign((size_t)K.sparse_len, 0);
    K.mapD.assign((size_t)2 * K.nsparse, 0);

    std::vector<int64_t> nxt(K.colptr.begin(), K.colptr.end() - 1);
    auto put = [&](int col, int row, double v) -> int {
        int64_t d = nxt[col]++;
        K.rowval[d] = row;
        K.nzval[d] = v;
        return (int)d;
    };
    // upper-left block: triu(P) plus structural zeros on a missing diagonal (csc_assembly.jl:207-220)
    for (int j = 0; j < n; ++j) {
        for (int64_t q = Pp[j] - base; q < Pp[j + 1] - base; ++q) K.mapP[q] = put(j, (int)(Pi[q] - base), Px[q]);
        if (!has_diag[j]) put(j, j, 0.0);
    }
    // upper-right block: A transposed (csc_assembly.jl:125-143 with shape :T)
    for (int j = 0; j < n; ++j)
        for (int64_t q = Ap[j] - base; q < Ap[j + 1] - base; ++q) K.mapA[q] = put(n + (int)(Ai[q] - base), j, Ax[q]);
    // lower-right blocks per cone, then the sparse-expansion columns (v first, then u)
    int pcol = n + m;
    for (const ConeInfo& ci : K.cones) {
        int row0 = n + ci.off;
        int* block = K.mapHs.data() + ci.boff;
        bool dense = ci.kind == HIPKKT_CONE_PSD || (ci.kind == HIPKKT_CONE_SOC && !ci.sparse);
        if (dense) {
            int64_t kidx = 0;
            for (int t = 0; t < ci.numel; ++t)
                for (int r = 0; r <= t; ++r) block[kidx++] = put(row0 + t, row0 + r, 0.0);
        } else {
            for (int t = 0; t < ci.numel; ++t) block[t] = put(row0 + t, row0 + t, 0.0);
        }
        if (ci.sparse) {
            for (int t = 0; t < ci.numel; ++t) K.mapV[ci.soff + t] = put(pcol, row0 + t, 0.0);
            for (int t = 0; t < ci.numel; ++t) K.mapU[ci.soff + t] = put(pcol + 1, row0 + t, 0.0);
            K.mapD[2 * ci.sidx] = put(pcol, pcol, 0.0);
            K.mapD[2 * ci.sidx + 1] = put(pcol + 1, pcol + 1, 0.0);
            pcol += 2;
        }
    }
    for (int j = 0; j < N; ++j) {
        if (nxt[j] != K.colptr[j + 1]) throw std::runtime_error("kkt: internal column fill mismatch");
        K.map_diag[j] = (int)(K.colptr[j + 1] - 1);
        if (K.rowval[K.map_diag[j]] != j) throw std::runtime_error("kkt: diagonal is not last in column");
    }
    // expected pivot signs
    K.dsigns.assign((size_t)N, 1);
    for (int i = n; i < n + m; ++i) K.dsigns[i] = -1;
    for (int t = 0; t < K.nsparse; ++t) {
        K.dsigns[n + m + 2 * t] = -1;          // Dsigns(::SOCExpansionMap) = (-1, 1)
        K.dsigns[n + m + 2 * t + 1] = 1;
    }
}

}  // namespace hipkkt
