// Device kernels of the supernodal multifrontal LDL^T, the tree solves, the KKT value updates,
// the residual SpMV and the cone scalings.  gfx950 (wave64) only.
//
// Data layout in HBM (all fronts of the tree are resident at once -- 288 GB makes the classic
// multifrontal stack unnecessary, and a static layout lets every per-iteration step be a pure
// function of precomputed index maps):
//   panel store   per supernode s: f_s x nc_s column-major (ld = f_s), f_s = nc_s + nb_s.
//                 after factorisation: strictly-lower part = L, diagonal = D.
//   update store  per supernode s: nb_s x nb_s column-major (ld = nb_s), lower triangle =
//                 the Schur complement s hands to its parent.
//   uvec          per supernode s: nb_s doubles, forward-solve contributions to the ancestors.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>

namespace hipkkt {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };   // -> HIPKKT_ERR_HIP at the C ABI

// Function attributes (hipFuncSetAttribute) are per DEVICE, and a process may hold handles on several devices and
// drive them from several threads: run `fn` once per device ordinal, under a lock; `fn` returns a hipError_t and a
// failure throws (the C ABI turns that into a negative return code) instead of leaving a kernel that cannot launch.
class PerDeviceOnce {
    std::mutex mu_;
    uint64_t done_[4] = {0, 0, 0, 0};       // 256 device ordinals
public:
    template <class F>
    void run(F&& fn)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) throw HipError("hipGetDevice failed");
        std::lock_guard<std::mutex> lk(mu_);
        if (done_[dev >> 6] >> (dev & 63) & 1) return;
        const hipError_t e = fn();
        if (e != hipSuccess) {
            (void)hipGetLastError();
            throw HipError(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
        }
        done_[dev >> 6] |= (uint64_t)1 << (dev & 63);
    }
};
template <class K>
inline hipError_t set_max_lds(K kernel, int bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

struct ExtItem {                 // a piece (<= 64 rows) of one child update column that lands in a panel column; 32 bytes
    int64_t uoff;                // offset in the update store of the piece's first entry U_c(a, b)
    int relstart;                // index into rel[] of the child's row a
    int cnt;                     // rows in the piece
    int rfirst, rlast;           // parent-local target rows of the piece's first and last entry (row slices skip
                                 // pieces that miss their row range without touching rel[])
    int tcol;                    // parent-local column this item lands in
    int pad;
};

struct SubItem {                 // a child update column restricted to one 64-row tile of the parent's U; 16 bytes
    int64_t uoff;                // offset in the update store of the first entry
    int relstart;                // index into rel[] of the first row
    unsigned char cnt;           // rows (1..64)
    unsigned char qcol;          // column inside the tile (0..63)
    unsigned short pad;
};

struct FrontDesc {               // everything a kernel needs to know about one front, in launch order; 64 bytes
    int64_t front_off;           // panel store offset
    int64_t upd_off;             // update store offset
    int64_t w_off;               // solve-matrix store offset
    int64_t rp;                  // rowptr[s]
    int64_t kptr;                // first K entry
    int s;                       // supernode id
    int c0;                      // first column
    int nc;                      // columns
    int nb;                      // rows below
    int nk;                      // K entries
    int pad;                     // row-sliced panels (TreeDev::sdesc): slice << 16 | number of slices
};

// Packed sweep records (hipkkt.hip, build_records): what a sweep kernel needs to know about a front, laid out so that it
// arrives in ONE round of loads.  The fronts of a launch and size class (block-class / one-wave / tiny) have records of one
// size, so a record's address follows from the front's place in the launch alone: [SolveHdr (64 B)] [idx: fmax ints]
// [gather slots: fmax x 32 B, absent on tree level 0] -- a thread fetches the header and its row's slots side by side,
// and the values (b, the children's contributions, the matrix) in the second round.  The legacy layout (FrontDesc ->
// gl_ptr / perm / rows -> gl_src -> values) took four.
//   idx[i]   i < nc: the row of column c0 + i in the caller's order (perm); i >= nc: the permuted index of below-row i (rows)
//   slot i   {count, six sources (indices into uvec, -1 beyond the count), index into gl_src of the seventh source}
struct SolveHdr {
    int64_t mat_off;             // block class: W = [T; M] in the solve-matrix store (W' behind it); one-wave / tiny: the panel in the front store
    int64_t rp;                  // rowptr[s]: first of the nb rows below in rows[] / uvec
    int s, c0, nc, nb;
    int par;                     // parent supernode, -1 for a root
    int nchild;                  // children swept by chained launches (ChainArgs::nchild)
    int64_t pad[3];
};
struct RecSeg {                  // the records of one kernel launch's fronts, by size class: 0 block-class, 1 one-wave, 2 tiny
    int64_t off[3];              // byte offset in SolveArgs::recs of the class's first record
    int stride[3];               // bytes per record
    int fmax[3];                 // row slots per record (a multiple of 4)
};

struct TreeDev {                 // device copies of the symbolic structure
    int nsuper;
    const int* sn_start;         // nsuper+1
    const int64_t* rowptr;       // nsuper+1
    const int* rows;             // sum nb
    const int* rel;              // sum nb: local row in the parent's front
    const int* ncolpar;          // nsuper: how many of s's rows fall in its parent's columns
    const int64_t* front_off;    // nsuper+1
    const int64_t* upd_off;      // nsuper+1
    const int* child_ptr;        // nsuper+1
    const int* child_idx;
    const int64_t* kptr;         // nsuper+1
    const int* ksrc;
    const int* kdst;
    const int* sched;            // supernodes in launch order (level by level, size class inside)
    const int* spos;             // supernode -> position in sched
    const int* sn_parent;        // assembly tree
    const FrontDesc* sdesc;      // row slices of the panels too tall for one CU's LDS, one record per slice (k_panel SLICED)
    const FrontDesc* desc;       // same order: one 64-byte record per launch slot (one scalar load instead of a
                                 // chain of dependent index loads at the head of every kernel)
    const signed char* psign;    // N: expected pivot sign, permuted order
    const int* perm;             // N: perm[new] = old
    // Extend-add work lists of the panel columns: the pieces of child update columns that land in a panel column,
    // grouped by column, children in fixed order; a wave owns whole columns (wave_cut) and applies their pieces
    // one after the other (deterministic).  Local row / column j of front s has index lc = sn_start[s] + rowptr[s] + j
    // (used by the gather lists below).
    const int64_t* item_ptr;     // = gl_ptr
    const ExtItem* items;
    // dense child (hipkkt.hip, upload): offset in upd of the child whose update block is this front's whole front
    // (entry (r, j) -> entry (r, j), leading dimension = this front's size), or -1; block-class fronts only
    const int64_t* dense_off;
    // forward-solve gather lists: local row r of front s (same indexing) receives
    // uvec[gl_src[q]] for q in gl_ptr[lc] .. gl_ptr[lc+1]
    const int64_t* gl_ptr;
    const int* gl_src;
    // inverse of gl_src: contribution entry e of the tree (index into uvec) is gather-list entry udst[e] of its
    // receiver.  The multi-column solve stores contributions in THAT order, so a receiver reads a contiguous run.
    const int* udst;
    // many-column sweeps (solve_kernels.hip, k_*_m): one-column leaves are PULLED -- the parent computes -L(r,0) b_c from
    // the leaf's row of B instead of reading a stored contribution row.  glm_ptr / udst_m are gl_ptr / udst without the
    // pulled leaves' entries (udst_m = -1 for them), with one extra slot at the head of a row's run for the SUM of its pulled
    // terms (k_pull_leaves_m); a term is the leaf's column hp_col (tree order) and the position hp_lidx of L(r,0) in
    // `fronts`.  A pulled leaf's launch record has FrontDesc::pad bit 0 set.
    const int64_t* glm_ptr;
    const int* udst_m;
    const int* hp_col;
    const int* hp_row;           // = perm[hp_col]: the leaf's row in the caller's order
    const int64_t* hp_lidx;
    const int64_t* pr_ptr;       // receiving rows with pulled terms: row k's terms are pr_ptr[k] .. pr_ptr[k+1] of hp_*, their sum
    const int* pr_slot;          // goes to slot pr_slot[k] of the receiver-ordered store (the first of the row's run)
    // per child c: cuts[cut_ptr[c] + t] = first child row index whose parent-local row >= nc_p + 64 t
    const int64_t* cut_ptr;
    const int* cuts;
    // per supernode: 17 item indices splitting its panel items (local columns < nc) into 16 slices
    // that end on column boundaries: a wave takes whole columns, so no two waves share a target column
    const int64_t* wave_cut;     // (nsuper + row slices) * 17: supernode s at s, row slice q of sdesc at nsuper + q (its own item list)
    const int64_t* tinv_off;     // nsuper+1: offset of the solve matrix W = [L11^{-1}; L21 L11^{-1}] (f x nc)
    // Schur tiles: sub-items of tile t are sitems[tile_cut[5t] .. tile_cut[5t+4]), sorted by tile column;
    // wave w of the tile's workgroup takes [tile_cut[5t+w], tile_cut[5t+w+1]) -- whole columns
    const SubItem* sitems;
    const int64_t* tile_cut;     // 5 per tile
};

struct FactorArgs {
    TreeDev T;
    const double* Kval;          // caller's K.nzval (original order, un-regularised)
    const double* eps;           // device scalar: static regulariser to add as eps*sign, or null
    double* fronts;
    double* upd;
    double* Dinv;                // N, permuted order
    int* flags;                  // [0] #dynamic regularisations, [1] non-finite pivot seen, [2] an overlap-mode wait gave up
    // Overlap mode (the narrow top of the tree, factor_kernels.hip "overlap"): a level's Schur tiles run on a second
    // stream BESIDE its panels and the next level's panels, ordered by counters in memory instead of kernel boundaries.
    //   ov_prog[s]    panel columns of supernode s whose L / D entries are published (written through) this factorisation
    //   ov_done[s]    Schur tiles of s completed this factorisation
    //   ov_ntiles[s]  tiles a parent's panel must wait for (0: s is not factorised in overlap mode)
    int* ov_prog;
    int* ov_done;
    const int* ov_ntiles;
    //   ov_sprog[q]   the same progress per ROW SLICE q (position in TreeDev::sdesc) of a front factorised in slices
    //   ov_sbase[s]   first slice of supernode s in sdesc (-1: s is factorised whole)
    int* ov_sprog;
    const int* ov_sbase;
    //   ov_started[q] panel workgroups (whole panels and row slices) of launch q that have begun to run this
    //                 factorisation: the launch's tile kernel is released by a gate (k_ov_gate) once ALL of them are
    //                 resident -- a tile workgroup that waits for its panel then never keeps that panel off a CU
    int* ov_started;
    int ov_slot;                 // this launch's index q
    int ov;                      // this launch runs in overlap mode
    long long ov_limit;          // bound of every overlap-mode wait in 100 MHz ticks (50 ms; the tests set it to 0 to
                                 // force the give-up-and-repeat path: HIPKKT_OV_TEST_LIMIT)
    double dyn_eps, dyn_delta;
    int nbk;                     // block-column width (<= 16), chosen so the LDS buffer fits
    long long* stamps;           // diagnostic only (HIPKKT_STAMPS=1): phase time stamps of block 0, else null
    int stamp_row;
};
#define HIPKKT_STAMP(A, k) do { if ((A).stamps && blockIdx.x == 0 && threadIdx.x == 0) \
        (A).stamps[(A).stamp_row * 16 + (k)] = wall_clock64(); } while (0)

struct SolveArgs {
    TreeDev T;
    const double* fronts;
    const double* tinv;          // per supernode: W = [T; M], f x nc col-major
    const double* Dinv;
    const double* b;             // original order
    double* out;                 // original order
    double* xp;                  // N, permuted work vector
    double* uvec;                // sum nb
    // NR right-hand sides in one launch (the single-column kernels' NR template parameter): column c of b, out, xp,
    // uvec lives at these strides
    long long* top_stamps;       // diagnostic only (HIPKKT_TOP_STAMPS=1): per front of the persistent set and direction eight
                                 // wall-clock stamps of its hop (k_top_solve), else null
    long long top_limit;         // bound of every wait of the persistent kernels in 100 MHz ticks (50 ms; the tests set
                                 // it to 0 to force the give-up-and-repeat path: HIPKKT_TOP_TEST_LIMIT)
    int64_t ld_b, ld_out, ld_xp, ld_uvec;   // (b, out: columns ld apart.  The sweeps' internal vectors xp and uvec keep their
                                            //  NR columns INTERLEAVED -- entry i of column c at i * NR + c -- so that a gather
                                            //  touches one cache line for all columns; ld_xp / ld_uvec are unused)
    // k_top_solve_sliced (sets with very tall fronts): its tasks are (front, slice) pairs -- a front whose W is too
    // large for one CU to stream per hop is cut into R slices (rows of W forward, columns of x backward) that never
    // exchange anything.  Task t works on the front at set position tk_pos[t], slice tk_sl[t] & 0xff of tk_sl[t] >> 8;
    // the tasks of position p are tbase[p] .. tbase[p+1]; xf keeps a front's forward solution while its backward
    // slices overwrite xp.
    const int* tk_pos;
    const int* tk_sl;
    const int* tbase;
    double* xf;
    const double* add;           // many-column sweeps (row-major N x KP): out = solution + add where non-null
    double* tall_ws;             // work space of the tall-front sweep kernels (k_*_tall_*): N doubles + the blocks' partial sums; or null
    const char* recs;            // packed sweep records (SolveHdr ...), or null: the legacy layout (TreeDev::desc, gl_ptr, ...)
    int* chain_cnt;              // nullable; the chained launches' forward counters (ChainArgs::cnt): a sweep that chains the
                                 // levels below the persistent set leaves the set's fronts' counters at their bottom
                                 // children's count, and the persistent kernels put them back to zero
};

constexpr int kSolveChunk = 128;  // diagonal chunk of the triangular solves: one wave, two unknowns per lane
constexpr int kMaxNbk = 16;
size_t solve_lds_bytes(int fmax, int ncmax);
// persistent kernel over the top `count` fronts (schedule positions begin ..): forward then backward sweep
constexpr int kTopMaxFronts = 480;
int top_solve_capacity(size_t lds, bool tall);   // resident workgroups the device guarantees for the persistent kernel
                                                // (tall: its 1024-thread build for sets with very tall fronts)
int top_solve_capacity_nr(size_t lds_total, int nr);     // the same for the 1024-thread build's NR-column instance
// nr right-hand sides (1, 2 or 4); lds = bytes per right-hand side
void launch_top_solve(const SolveArgs& a, int begin, int count, int grid, size_t lds, int* flags, int nflag, int epoch,
                      hipStream_t st, bool tall, int nr = 1);
int top_solve_sliced_capacity(size_t lds, int nr = 1);
void launch_top_solve_sliced(const SolveArgs& a, int begin, int pos0, int task0, int task1, int grid, size_t lds, int* flags, int nflag,
                             int epoch, hipStream_t st, int nr = 1);
// max_blocks > 0: at most that many workgroups (each walks several supernodes); nsmall: how many of the list's supernodes
// have at most winv_small_nc() columns -- enough of them and they go to a launch of their own, four times as many 128-thread
// workgroups (solve_kernels.hip: k_winv)
void launch_tinv(const TreeDev& T, const double* fronts, double* tinv, const int* list, int count, int ncmax,
                 hipStream_t st, int max_blocks = 0, int nsmall = 0);
int winv_small_nc();

// factorisation of one level: small fronts (one wave each), panels (one workgroup each), then the
// update blocks tiled over many workgroups
constexpr int kSmallFrontMax = 64;         // f <= 64 ...
constexpr int kSmallSliceMax = 1536;       // ... and f*nc + nb*nb <= this many doubles of LDS per wave
void launch_front_wave(const FactorArgs& a, int begin, int count, int slice_doubles, hipStream_t st);
void launch_front_tiny(const FactorArgs& a, int begin, int count, hipStream_t st);     // fronts with f <= 8, eight to a wave
void launch_panel(const FactorArgs& a, int begin, int count, int bs, size_t lds, hipStream_t st);
// row slices of the panels too tall for one CU (TreeDev::sdesc[begin ..]), one 1024-thread workgroup each
void launch_panel_sliced(const FactorArgs& a, int begin, int count, size_t lds, hipStream_t st);
// ov_grid: overlap mode only -- the number of tile workgroups that may exist at a time (they walk the launch's tiles)
void launch_schur(const FactorArgs& a, const int2* tiles, int tile_begin, int ntiles, hipStream_t st, int ov_grid = 0, int tile_nc = 0);
// overlap mode: returns (the stream goes on) once *started >= target, i.e. every panel workgroup of the launch is resident
void launch_ov_gate(const int* started, int target, int* abort_word, long long limit, hipStream_t st);
// word2[0] must be 0; afterwards word2[1] = 1 iff a kernel on `second`, submitted behind a waiting kernel on `first`, ran
// while that one waited -- i.e. the two streams do not share a hardware queue
void launch_concurrency_probe(int* word2, hipStream_t first, hipStream_t second);
size_t panel_lds_bytes(int fmax, int panel_max);
// litmus test of the hand-over contract (factor_kernels.hip): 2 * pairs workgroups, producer / consumer pairs
void launch_handover_litmus(int variant, double* payload, int* sig, int* ack, int pairs, int words, int rounds,
                            unsigned long long* mismatches, unsigned long long* timeouts, hipStream_t st);
// nr = 1, 2 or 4 right-hand sides per launch (column strides in SolveArgs::ld_*); lds = bytes per right-hand side
// (rs: the launch's packed records; ignored when SolveArgs::recs is null)
void launch_fwd(const SolveArgs& a, const RecSeg& rs, int begin, int count, int bs, size_t lds, hipStream_t st, int nr = 1);
void launch_bwd(const SolveArgs& a, const RecSeg& rs, int begin, int count, int bs, size_t lds, hipStream_t st, int nr = 1);
// one launch for a level's one-wave fronts [begin, begin + nwave) and the tiny fronts behind them
void launch_fwd_small(const SolveArgs& a, const RecSeg& rs, int begin, int nwave, int ntiny, hipStream_t st, bool leaf = false, int nr = 1);
void launch_bwd_small(const SolveArgs& a, const RecSeg& rs, int begin, int nwave, int ntiny, hipStream_t st, int nr = 1);
// one launch for a whole level: nblock block-class fronts at [begin, ..), then nwave one-wave, then ntiny tiny fronts
void launch_fwd_level(const SolveArgs& a, const RecSeg& rs, int begin, int nblock, int nwave, int ntiny, int bs, size_t lds, hipStream_t st, int nr = 1);
void launch_bwd_level(const SolveArgs& a, const RecSeg& rs, int begin, int nblock, int nwave, int ntiny, int bs, size_t lds, hipStream_t st, int nr = 1);

// several right-hand sides: work vectors row-major N x KP / sum(nb) x KP (KP = columns rounded up to 16)
// iperm[caller's index] = permuted index
void launch_permute_in(const double* B, int64_t ldb, double* Xp, int KP, const int* iperm, int N, int nrhs, hipStream_t st);
// row-major source / destination (N x KP): rows move, columns stay; dir 0: Xp[iperm[o]] = B[o], 1: X[o] = Xp[iperm[o]]
void launch_permute_rows(double* dst, const double* src, int KP, const int* iperm, int N, int dir, hipStream_t st, const double* add = nullptr);
void launch_permute_out(double* X, int64_t ldx, const double* Xp, int KP, const int* iperm, int N, int nrhs, hipStream_t st);
void launch_fwd_multi(const SolveArgs& a, int begin, int count, bool small, int ncmax, int KP, hipStream_t st);
// fronts too tall for the block kernels' LDS: their rows spread over workgroups (fmax: tallest of them; SolveArgs::tall_ws)
void launch_fwd_tall(const SolveArgs& a, int begin, int count, int fmax, hipStream_t st);
void launch_bwd_tall(const SolveArgs& a, int begin, int count, int fmax, int N, hipStream_t st);
size_t tall_ws_doubles(int N, int ntall, int fmax);     // doubles SolveArgs::tall_ws needs for a launch with ntall such fronts
void launch_pull_leaves_multi(const SolveArgs& a, int nrows, int KP, hipStream_t st);
void launch_bwd_multi(const SolveArgs& a, int begin, int count, bool small, int ncmax, int KP, hipStream_t st, bool leaves = false);

// ---- chained levels (chain_kernels.hip): several tree levels of a sweep in ONE launch, ordered by counters in memory
// instead of kernel boundaries.  A segment is one launch of the per-level path (same three size classes); workgroups are
// numbered segment by segment, so that every workgroup depends only on workgroups with a LOWER index.
struct ChainSeg {
    int begin;                   // schedule position of the segment's first front
    int nblock, nwave, ntiny;    // block-class fronts, then one-wave fronts, then tiny fronts (f <= 8)
    int wg0;                     // the segment's first workgroup
    int leaf;                    // forward: the fronts have no children (tree level 0)
    RecSeg rec;                  // the fronts' packed records
};
constexpr int kChainMaxSeg = 20;
struct ChainArgs {
    int nseg;
    int lo, hi;                  // schedule positions covered by this launch (backward: a parent at a position >= hi is complete)
    int* cnt;                    // nsuper: forward, children of supernode s that have handed over their contribution this sweep
    const int* nchild;           // nsuper: how many of s's children are swept by chained launches (the ones that add to cnt[s])
    int* done;                   // nsuper: backward, epoch of supernode s's last finished backward step
    int* abort_word;             // set when a bounded wait expires: everyone leaves, the host repeats the sweep level by level
    int epoch;
    int lo0, nstamp;             // diagnostic (SolveArgs::top_stamps): first position and number of fronts of the sweep's chained part
    ChainSeg seg[kChainMaxSeg];
};
// lds = bytes per right-hand side of the largest block-class front in the segments (0: none)
void launch_fwd_chain(const SolveArgs& a, const ChainArgs& c, int nwg, size_t lds, hipStream_t st, int nr);
void launch_bwd_chain(const SolveArgs& a, const ChainArgs& c, int nwg, size_t lds, hipStream_t st, int nr);
constexpr int kChainBS = 512;    // workgroup size of the chained kernels: one block-class front, 8 one-wave or 64 tiny fronts
inline int chain_seg_wgs(const ChainSeg& s) { return s.nblock + (s.nwave + kChainBS / 64 - 1) / (kChainBS / 64) + (s.ntiny + kChainBS / 8 - 1) / (kChainBS / 8); }

// ---- KKT value updates (kktsolver_directldl.jl:130-188, 211-245, 374-386)
void launch_scatter(double* Kval, const int* idx, const double* vals, int64_t n, double scale, hipStream_t st);
void launch_scale(double* Kval, const int* idx, double scale, int64_t n, hipStream_t st);
// per sparse SOC t: K[mapU] = u * (-eta2), K[mapV] = v * (-eta2), K[mapD] = (-eta2, +eta2)
void launch_soc_columns(double* Kval, const int* mapU, const int* mapV, const int* mapD,
                        const double* u, const double* v, const double* eta2, const int* soc_of_entry,
                        int sparse_len, int nsparse, hipStream_t st);
// eps = c0 + c1 * max_i |Kval[diag[i]]|  (kktsolver_directldl.jl:297-310) -> *eps_out
void launch_regularizer(const double* Kval, const int* diag, int N, double c0, double c1,
                        double* partial, double* eps_out, hipStream_t st);

// ---- residual: e = b - K_sym x on the un-regularised K, and infinity norms
struct SpmvDev {
    const int64_t* ptr;          // N+1, full symmetric CSR
    const int* col;
    const int* vmap;             // entry -> index into Kval
    const double* val;           // the values in CSR order (a refreshed copy of Kval[vmap[.]]), or null
    const int64_t* pend;         // nullable; rows of the x block (row < n): end of the row's entries with column < n (P)
    int N;
    int lanes_per_row;           // 8 or 64
    // rows longer than kLongRow entries (a dense constraint row, say) are cut into chunks of kLongChunk entries,
    // one workgroup each, and their partial sums combined in chunk order: no single wave walks 50 000 entries
    int nlong, nchunks;
    const int* long_rows;        // nlong
    const int64_t* long_chunk_ptr;   // nlong + 1: chunk range of each long row
    const int64_t* chunk_q;      // nchunks + 1 would not do (rows are not adjacent): 2 per chunk, [begin, end)
    double* long_partial;        // nchunks per right-hand side
};
constexpr int kLongRow = 4096;
constexpr int kLongChunk = 2048;
// norm_out[j] = ||e_j||_inf (not finite if any entry is non-finite).  nrhs > 1: column j of b, x, e
// at stride ld; `partial` then needs nrhs * (kNormParts + 1) doubles
constexpr int kNormParts = 2048;   // partial needs nrhs * (kNormParts + 1) doubles
constexpr int kMaxNormbCols = 4;   // ||b|| rides along in the residual pass for up to this many columns: partial then needs
                                   // nrhs * (2 kNormParts + 1) doubles
void launch_update_values(double* K, const int* mapHs, const double* Hs, int nHs, const int* mapU, const int* mapV,
                          const int* mapD, const double* u, const double* v, const double* eta2, const int* soc_of_entry,
                          int sparse_len, int nsparse, double* fval, const int* kpos, hipStream_t st);
// flag_in/flag_out (nullable): *flag_out = (*flag_in != 0) as a double, so that a device status word rides along
// with the norm read-back
void launch_residual(const SpmvDev& A, const double* Kval, const double* b, const double* x, double* e,
                     double* partial, double* norm_out, hipStream_t st, int nrhs = 1, int64_t ld = 0,
                     const int* flag_in = nullptr, double* flag_out = nullptr, double* normb_out = nullptr);
// normb_out (nullable): ||b_j||_inf as well -- in the same pass when there is one column and no long row
// (partial then needs 2 * (kNormParts + 1) doubles), by a separate reduction otherwise
void launch_norm_inf(const double* v, int n, double* partial, double* out, hipStream_t st, int nrhs = 1,
                     int64_t ld = 0);
void launch_axpby_sum(double* y, const double* a, const double* b, int64_t n, hipStream_t st);   // y = a + b
// val[q] = Kval[vmap[q]]: the CSR-ordered copy the residual streams (one gather per value update instead of one
// per residual)
void launch_gather_values(double* val, const double* Kval, const int* vmap, int64_t nnz, hipStream_t st);
// b_j = [rx_j; rz_j; 0]: rx is n x nrhs (ld n), rz is m x nrhs (ld m), b is N x nrhs (ld N)
void launch_pack_rhs(double* b, const double* rx, const double* rz, int n, int m, int p, hipStream_t st,
                     int nrhs = 1);
// ---- row-major (N x KP, KP = columns rounded up to 16, padding columns zero) variants for the multi-column path:
// a row of 16 columns is one 128-byte line, so the SpMV's gather of x moves whole lines
void launch_pack_rhs_rm(double* B, const double* rx, const double* rz, int n, int m, int p, int nrhs, int KP, hipStream_t st);
void launch_unpack_lhs_rm(double* lhsx, double* lhsz, const double* X, int n, int m, int nrhs, int KP, hipStream_t st);
// norm_out[c] = ||e_c||_inf, normb_out[c] = ||b_c||_inf (nullable); partial: 2 * 512 * KP doubles; no long rows (A.nlong == 0)
void launch_residual_rm(const SpmvDev& A, const double* B, const double* X, double* E, double* partial, double* norm_out,
                        double* normb_out, int KP, hipStream_t st);
void launch_accept_columns_rm(double* X, const double* cand, double* E, const double* E2, const int* mask, int N, int KP,
                              hipStream_t st);
// columns j with mask[j] != 0: x_j = cand_j, e_j = e2_j  (N x nrhs, ld N)
void launch_accept_columns(double* x, const double* cand, double* e, const double* e2, const int* mask, int N,
                           int nrhs, hipStream_t st);
void launch_check_finite(const double* v, int n, int* flag, hipStream_t st);
// kktsolver_getlhs! (kktsolver_directldl.jl:329-343) on the device: lhsx = x[0:n], lhsz = x[n:n+m]; either may be null.
// One kernel instead of two device-to-device copies (the copy engine's latency is several kernel launches' worth).
void launch_unpack_lhs(double* lhsx, double* lhsz, const double* x, int n, int m, hipStream_t st);
// The refinement loop's accept / stop rule (kktsolver_directldl.jl:389-449) evaluated on the device after round r
// (kernels.hip, k_ir_round), for nr right-hand side columns at once (column c: state + c state_stride, norme0[c],
// normb[c], cand[c], x + c n, dx + c n, readback + 5 c).  state holds 4 doubles per round {active, rounds, bad, norme};
// readback (nullable): 5 doubles {state of round r, abort} per column for ONE copy to the host; sticky (nullable): the
// deferred-status record {bad, more, abort, rounds, #dyn. regularisations, eps, solves} (several columns: each adds its own).
// IrPartials: the residual kernels' partial maxima left un-finished (launch_residual with norm_out = nullptr, allowed when
// residual_partials_ok): column c's np = residual_grid + 1 partials of ||e0|| at e0 + c np, its np - 1 of ||b|| at
// b0 + c (np - 1), of the candidate's ||e|| at cand + c np -- the kernel takes the maxima itself and stores ||e0||, ||b||
// to norme0 / normb for later rounds.  flag_in: the sweeps' abort word (instead of abort_word, its copy as a double).
// Null pointers: the norms come finished in norme0 / normb / cand.
struct IrPartials { const double* e0 = nullptr; const double* b0 = nullptr; const double* cand = nullptr; int np = 0; const int* flag_in = nullptr; };
void launch_ir_round(double* state, int state_stride, int r, bool first, double* norme0, double* normb,
                     const double* cand, const double* abort_word, double* x, const double* dx, int n, int nr, double abstol,
                     double reltol, double stop_ratio, int max_iter, double* readback, double* sticky, hipStream_t st,
                     const IrPartials& Q = IrPartials{});
int residual_grid(const SpmvDev& A);     // workgroups (= partials per column, less the long rows' one) of the residual kernel
inline bool residual_partials_ok(const SpmvDev& A, int nrhs) { return A.nlong == 0 && nrhs <= kMaxNormbCols; }
void launch_ir_fold(const double* state, int state_stride, int r, int nr, const double* abort_word, double* sticky, hipStream_t st);
void launch_fold_update_status(double* sticky, const double* st4, hipStream_t st);
void launch_fold_flag(double* sticky, const int* flag, hipStream_t st);
// p[0..n) = 0 with a kernel: a small hipMemsetAsync stalls the stream for ~40 us on this stack
void launch_zero_ints(int* p, int n, hipStream_t st);
struct ZeroList {
    int* p[8];
    int n[8];
    int count = 0;
    void add(int* ptr, int len) { if (ptr && len > 0 && count < 8) { p[count] = ptr; n[count] = len; ++count; } }
};
void launch_zero_ints_multi(const ZeroList& Z, hipStream_t st);      // up to eight arrays in one launch
// dst[0..4] = {eps[0], conefail[0], flags[0], flags[1], flags[2]} (null pointers read as 0)
// sticky (nullable): also fold the words into the deferred-status record (what launch_fold_update_status does)
void launch_collect_status(double* dst, const double* eps, const int* conefail, const int* flags, hipStream_t st, double* sticky = nullptr);

// ---- cone scalings on the device (update_scaling! + get_Hs!, src/cones/coneops_*.jl)
struct ConeDev {
    int ncones;
    const int* kind;             // per cone
    const int* off;              // rng_cones start
    const int* numel;
    const int64_t* boff;         // rng_blocks start
    const int* sidx;             // sparse SOC index or -1
    const int* soff;             // offset in concatenated u/v or -1
    // per-element cone id for the elementwise (zero / nonnegative) kernels
    const int* elem_cone;        // m
    // list of second-order cones
    const int* soc_list;
    int nsoc;
    // list of PSD cones; psd_dim = matrix side k, psd_aoff = offset of the cone's k x k matrix R R' in psdA
    const int* psd_list;
    const int* psd_dim;          // per cone (0 for non-PSD)
    const int64_t* psd_aoff;     // per cone
    int npsd;
    int psd_kmax;
};
struct ConeState {
    double* w;                   // m: NN: sqrt(s/z); SOC: normalised w
    double* lam;                 // m: scaled point lambda = W z (NN: sqrt(s z); SOC: coneops_socone.jl:113-123); may be null
    double* eta;                 // per cone (SOC)
    double* u;                   // sparse_len
    double* v;                   // sparse_len
    double* eta2;                // nsparse
    double* Hs;                  // |Hs| positive blocks
    double* psdA;                // per PSD cone: A = R R' (k x k col-major), Hs = A (x)_s A
    double* psdR;                // per PSD cone: R = L1 V Lam^{-1/2} and Rinv = Lam^{-1/2} U' L2' (k x k col-major, same
    double* psdRinv;             //   offsets as psdA); the singular values Lam go to lam[off .. off + k), descending
    int* fail;                   // set to 1 when a point is not interior
};
constexpr int kPsdMaxDim = 48;   // largest PSD side handled by the in-LDS scaling kernel
void launch_cone_scaling(const ConeDev& C, const ConeState& S, const double* s, const double* z, int m,
                         hipStream_t st);
// addend != null: y = -(W'W x + addend) (the Delta_s recovery of kkt_solve!, kktsystem.jl:206-212)
// Publish: a call's status record handed to the host by the call's LAST kernel instead of a copy and a zeroing launch
// behind it (~5 us each): n doubles from rec to dst -- page-locked host memory the device can write -- then the first
// nzero doubles of rec set to zero for the next call.  The host reads dst after synchronising with the stream.
// seq: written behind the record (dst[n]); the host checks it against the number it passed -- a device store to host memory
// that did not arrive (a mapping this code has not been run on) is then seen, not read as an all-zero "no error" record.
struct Publish { double* dst = nullptr; double* rec = nullptr; int n = 0, nzero = 0; double seq = 0.0; };
void launch_publish(const Publish& p, hipStream_t st);            // ... as a kernel of its own
void launch_mul_Hs(const ConeDev& C, const ConeState& S, double* y, const double* x, int m, hipStream_t st, const double* addend = nullptr,
                   const Publish& pub = Publish{});
void launch_psd_A_from_R(const ConeDev& C, const ConeState& S, hipStream_t st);    // psdA = psdR psdR' per PSD cone
// Hs blocks, sparse second-order-cone u / v / eta^2 (and psdA) from the scaling already in S (w, eta, psdR): get_Hs! on the device
void launch_cone_from_scaling(const ConeDev& C, const ConeState& S, int m, hipStream_t st);

// ---- the reduced-system algebra around the three solves of an IPM iteration (kktsystem.jl:135-215), device-resident
// konst = Delta_s_from_Delta_z_offset!(cones, ds, z) (coneops_compositecone.jl:185-202; zero :137-150, nonnegative
// coneops_nncone.jl:140-148, second-order coneops_socone.jl:241-268, PSD coneops_psdtrianglecone.jl:218-228), or a
// copy of s for the affine step; workz = konst - rhs_z.  Returns false if the cone list holds a kind it does not cover.
bool launch_sys_offset(const ConeDev& C, const ConeState& S, double* konst, double* workz, const double* ds,
                       const double* z, const double* rhs_z, int m, bool affine, hipStream_t st);
// y = Symmetric(P) x for the leading n x n block of K (rows / columns < n of the full-CSR image)
void launch_P_spmv(const SpmvDev& A, const double* Kval, const double* x, double* y, int n, hipStream_t st);
// out = a / (*alpha_div) + beta * b  (alpha_div null: a as is; beta from the device scalar if given, else beta_host):
// xi - x2 with xi = x / tau, and x1 + dtau x2
void launch_sys_axpby(double* out, const double* a, const double* alpha_div, const double* b, const double* beta,
                      double beta_host, int n, hipStream_t st);
struct DotPairs { const double* a[8]; const double* b[8]; int len[8]; int npairs; };
// out[p] = sum_i a_p[i] b_p[i], p < npairs (two-stage, fixed order); partial needs 8 * 64 doubles
void launch_dots(const DotPairs& P, double* partial, double* out, hipStream_t st);
// the scalars of kkt_solve! (kktsystem.jl:176-206): dots = {q.x1, b.z1, x.(P x1), xm.(P xm)}, cached = {q.x2, b.z2, x2.(P x2)}
// scal_in = {rhs_tau, rhs_kappa, var_tau, var_kappa}; out = {dtau, dkappa, tau_num, tau_den}
void launch_sys_scalars(const double* dots, const double* cached, const double* scal_in, double* out, hipStream_t st);
void launch_neg_sum(double* y, const double* a, const double* b, int n, hipStream_t st);      // y = -(a + b)
// fused forms of the above for kkt_solve!'s step recovery (fewer dependent launches, scalars by value)
// (pc nullable: P x2 too; P.npairs = 4, or 7 with the x2-only pairs {q.x2, b.z2, x2.(P x2)} behind them, which are then stored to `cached`)
void launch_P_spmv2(const SpmvDev& A, const double* Kval, const double* x1, const double* x, const double* x2, double tau,
                    double* pa, double* pb, double* xm_out, double* pc, int n, hipStream_t st);
void launch_pack_rhs_affine(double* b, const double* negq, const double* bb, const double* rhs_x, const double* s,
                            const double* rhs_z, int n, int m, int p, int ncol, hipStream_t st);
void launch_sys_step(double* dx, double* dz, const double* x1, const double* z1, const double* x2, const double* z2,
                     const double* scal, int n, int m, hipStream_t st);
// the dot products of P, the scalars of kkt_solve! (P.npairs = 4, or 7 with the x2-only pairs {q.x2, b.z2, x2.(P x2)} behind
// them, which are then stored to `cached`) and the step, the scalars formed inside the step kernel (every workgroup adds
// the partial sums up itself): two launches
void launch_dots_sys_step(const DotPairs& P, double* partial, double* cached, double rhs_tau, double rhs_kappa, double tau,
                          double kappa, double* out, double* dx, double* dz, const double* x1, const double* z1, const double* x2,
                          const double* z2, int n, int m, hipStream_t st, double* keep_x2 = nullptr, double* keep_z2 = nullptr);
// (keep_x2 / keep_z2: (x2, z2) copied there on the way -- it came out of this call's own solve and outlives the call)
void launch_neg_copy(double* y, const double* a, int n, hipStream_t st);                    // y = -a


// ---- Ruiz equilibration of (P, A, q, b) (problemdata.jl:133-221, mathutils.jl:129-269), all vectors on the device
struct EquilDev {
    int n, m;
    int64_t nnzP, nnzA;
    const int* Prow; const int* Pcol;   // per entry of triu(P)
    const int* Arow; const int* Acol;   // per entry of A
    double* Pval; double* Aval; double* q; double* b;
    double* d; double* e;               // cumulative scalings
    double* dwork; double* ework;       // this round's scalings
    double* scal;                       // [0] c, [1] ctmp, [2..] scratch
    double* partial;                    // 2 * 256 doubles
};
// one round of the loop at problemdata.jl:163-205
void launch_equil_round(const EquilDev& E, double scale_min, double scale_max, hipStream_t st);
// rectify_equilibration! (coneops_compositecone.jl:28-47; default coneops_defaults.jl:32-44, elementwise cones
// coneops_nncone.jl:8-17 / coneops_zerocone.jl:16-25) followed by the row re-scaling at problemdata.jl:212-216
void launch_equil_rectify(const EquilDev& E, const int* cone_kind, const int* cone_off, const int* cone_numel,
                          const int* elem_cone, int ncones, hipStream_t st);
// values[j] *= cscale * L[row[j]] * R[col[j]]   (lrscale!, mathutils.jl:231-244; data_updating.jl:181-194)
void launch_lrscale(double* values, const int* row, const int* col, int64_t nnz, const double* L, const double* R,
                    double cscale, hipStream_t st);

}  // namespace hipkkt
