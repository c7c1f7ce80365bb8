// Host-side assembly of the triu CSC KKT matrix and its data maps (setup only).
// Follows /root/reference/src/kktsolvers/direct-ldl/directldl_kkt_assembly.jl:15-175,
// src/utils/csc_assembly.jl, directldl_datamaps.jl:8-79,170-214 and the cone layout of
// src/cones/compositecone_type.jl:114-141.  All indices 0-based, int32 on the way to the device.
#pragma once
#include <cstdint>
#include <vector>

namespace hipkkt {

struct ConeInfo {
    int kind;          // HIPKKT_CONE_*
    int dim;           // numel, or matrix side for PSD
    int numel;
    int off;           // rng_cones start (0-based) in (s, z)
    int64_t boff;      // rng_blocks start in Hsblocks
    int64_t blen;
    int sparse;        // SOC with dim > 4: sparse expansion
    int sidx;          // index among sparse SOCs
    int soff;          // offset into the concatenated u / v
};

struct KKTAssembly {
    int n = 0, m = 0, p = 0, N = 0;
    int64_t nnzK = 0, nHs = 0;
    int nsparse = 0, sparse_len = 0;
    std::vector<ConeInfo> cones;
    // triu CSC
    std::vector<int64_t> colptr;
    std::vector<int> rowval;
    std::vector<double> nzval;
    // LDLDataMap (directldl_datamaps.jl:170-214)
    std::vector<int> mapP, mapA, mapHs, map_diag, mapU, mapV, mapD;
    std::vector<int> dsigns;       // kktsolver_directldl.jl:112-126
};

// P: triu CSC n x n; A: CSC m x n (any index base).  Throws std::runtime_error on bad input.
void assemble_kkt(int64_t n, int64_t m, const int64_t* Pp, const int64_t* Pi, const double* Px,
                  const int64_t* Ap, const int64_t* Ai, const double* Ax, int64_t ncones,
                  const int32_t* kinds, const int64_t* dims, int base, KKTAssembly& K);

}  // namespace hipkkt
