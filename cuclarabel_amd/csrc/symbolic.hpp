// Host-side symbolic analysis for the supernodal multifrontal LDL^T (setup only).
//
// Replaces what the reference obtains from QDLDL.jl's constructor
// (/root/reference/src/kktsolvers/direct-ldl/directldl_qdldl.jl:6-28: AMD ordering, symmetric
// permutation, elimination tree, column counts, allocation of L).  Because the reference never
// pivots (static regularisation + sign-guided dynamic regularisation), the structure computed
// here is fixed for the whole IPM run and every per-iteration operation runs on the device.
#pragma once
#include <cstdint>
#include <vector>

namespace hipkkt {

struct Graph {            // full symmetric pattern, no diagonal
    int n = 0;
    std::vector<int64_t> ptr;
    std::vector<int> idx;
};

enum OrderingKind { ORDER_AMD = 0, ORDER_ND = 1, ORDER_NATURAL = 2, ORDER_USER = 3 };

// LDS doubles of one row slice of a panel: the top nc x nc triangle (every slice keeps and factors its own copy) plus
// the slice's share of the nb rows below, nc doubles each.  slices = 1 is the whole panel as a trapezoid.
inline int64_t panel_slice_doubles(int64_t nc, int64_t nb, int64_t slices)
{
    return nc * (nc + 1) / 2 + ((nb + slices - 1) / slices) * nc;
}
// fewest row slices with which an f x nc panel fits cap doubles of LDS (0: it does not with max_slices)
inline int panel_slices_needed(int64_t nc, int64_t nb, int64_t cap, int max_slices)
{
    for (int r = 1; r <= max_slices; ++r)
        if (panel_slice_doubles(nc, nb, r) <= cap) return r;
    return 0;
}

struct SymbolicOptions {
    int ordering = ORDER_ND;
    double amd_dense_scale = 1.5;     // directldl_qdldl.jl:24
    int nd_leaf_size = 1000;
    // relaxed supernode amalgamation: merge a child into its parent when the merged supernode
    // has <= relax_cols[k] columns and the fraction of explicit zeros stays <= relax_zeros[k]
    int relax_cols[3] = {8, 32, 128};
    double relax_zeros[4] = {1.0, 0.5, 0.15, 0.05};
    double relax_tall = 2.0;     // the allowance is this much larger for the child with the tallest subtree
    // a supernode's panel (f x nc doubles) is kept LDS-resident while it is factorised: wider
    // supernodes are split into a chain so that the panel's trapezoid f*nc - nc(nc-1)/2 <= panel_cap (0 = no
    // splitting).  The panel kernel's LDS holds
    // 19 374 doubles of panel beside its block buffers (factor_kernels.hip: panel_lds_bytes); every split is one more
    // level of the schedule, so the cap sits just under that (cfg2: 31 levels at 17 344, 29 at 19 200)
    int64_t panel_cap = 19200;
    // A panel too tall for one CU is cut into ROW slices, one workgroup (CU) each: every slice holds the top nc x nc
    // block and factors it redundantly, so the slices never talk to each other (factor_kernels.hip, k_panel SLICED).
    // That keeps tall fronts wide -- a 1531-row front takes 96 columns per level instead of 12.
    int panel_max_slices = 128;      // (r03: 16 until then -- a 14 000-row front then had to take 21-column panels, 670 levels of them)
    int panel_slice_below = 64;  // ... and only for fronts whose unsliced panel would be narrower than this (cfg2's
                                 // 289-row fronts take 70+ columns unsliced: slicing them costs more than it saves)
    // the solve kernels keep a front's vector and its partial sums in LDS: (1 + ceil(nc/8)) * f doubles forward,
    // f + ceil(f/8) * nc backward (solve_kernels.hip: solve_lds_bytes) -- a very tall front must stay narrow for them
    int64_t solve_cap = 20000;
    int panel_max_cols = 96;     // and no panel is wider than this: k_winv stages L11 (nc x nc) in LDS, and with 96
                                 // columns two of its workgroups share a CU (cfg2: same 18 levels as with 128, 2 % faster)
    const int64_t* user_perm = nullptr;
};

// one stage of the numeric schedule: a set of supernodes with no dependencies among them
struct Level {
    int begin, end;       // range into Symbolic::level_sn
};

struct Symbolic {
    int N = 0;
    int64_t nnzK = 0;
    std::vector<int> perm, iperm;        // perm[new] = old ; iperm[old] = new
    // supernodes (columns are contiguous in the permuted order)
    int nsuper = 0;
    std::vector<int> sn_start;           // nsuper+1
    std::vector<int> col2sn;             // N
    std::vector<int> sn_parent;          // assembly tree, -1 for roots
    std::vector<int64_t> rowptr;         // nsuper+1, into rows / rel
    std::vector<int> rows;               // below-diagonal-block row indices (permuted), ascending
    std::vector<int> rel;                // rel[rowptr[s]+t]: local row of rows[..] in the parent's front
    std::vector<int64_t> front_off;      // nsuper+1, offsets (in doubles) into the front store
    std::vector<int> child_ptr, child_idx;   // children of each supernode (ascending)
    // original K entry -> front position
    std::vector<int64_t> kptr;           // nsuper+1
    std::vector<int> ksrc;               // index into the caller's K.nzval (original order)
    std::vector<int> kdst;               // lrow + lcol*f within the front
    std::vector<int> diag_src;           // N: K.nzval index of the diagonal of permuted column j
    // schedule
    std::vector<int> level_sn;           // supernodes grouped by level (leaves first)
    std::vector<Level> levels;
    std::vector<int> sn_level;
    // statistics
    int64_t nnzL = 0;                    // strictly-lower entries of L incl. explicit zeros
    int64_t nnzL_struct = 0;             // structural nnz(L) before amalgamation (QDLDL's count)
    double flops = 0;                    // sum_j (c_j^2 + 3 c_j) on the amalgamated structure
    int etree_height = 0;                // height of the column elimination tree
    int max_front = 0;
    int64_t front_store = 0;             // total doubles in the front store
    int64_t update_store = 0;            // total doubles in the update store
    std::vector<int64_t> upd_off;        // nsuper+1 offsets into the update store
};

void build_graph(int N, const int64_t* colptr, const int64_t* rowval, int index_base, Graph& g);
void amd_order(const Graph& g, double dense_scale, std::vector<int>& perm, const char* halo = nullptr);
void nd_order(const Graph& g, int leaf_size, double dense_scale, std::vector<int>& perm);

// full analysis of a triu CSC pattern (colptr/rowval with the given index base)
void analyse(int N, const int64_t* colptr, const int64_t* rowval, int index_base,
             const SymbolicOptions& opt, Symbolic& S);

}  // namespace hipkkt
