// libhipkkt.so: the C ABI of include/hipkkt.h over the symbolic analysis (symbolic.cpp), the KKT
// assembly (kkt_assembly.cpp) and the device kernels (kernels.hip).
#include "../../include/hipkkt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "kkt_assembly.hpp"
#include "symbolic.hpp"
#include "knobs.hpp"

namespace hipkkt {

static thread_local std::string g_last_error;

struct ArgError : std::runtime_error { using std::runtime_error::runtime_error; };

// the panel-shape settings of knobs.hpp over SymbolicOptions' defaults
static void apply_knobs(SymbolicOptions& opt)
{
    if (knobs().panel_cap >= 0) opt.panel_cap = knobs().panel_cap;
    if (knobs().panel_max_cols >= 0) opt.panel_max_cols = knobs().panel_max_cols;
    if (knobs().panel_slice_below >= 0) opt.panel_slice_below = knobs().panel_slice_below;
}

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_));                   \
    } while (0)

template <class T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { if (p) (void)hipFree(p); }
    void alloc(size_t count)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = count;
        HIP_CHECK(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)));
    }
    void zero(hipStream_t st) { HIP_CHECK(hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(T), st)); }
    template <class V>
    void upload(const std::vector<V>& v)
    {
        static_assert(sizeof(V) == sizeof(T), "element size mismatch");
        alloc(v.size());
        if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
};

struct Launch {
    int begin, count;
    bool small;                 // one wave per front
    int bs_panel, nbk, slice;   // panel kernel block size / block-column width; small-front LDS slice
    size_t lds_panel, lds_solve;
    int fmax, ncmax;            // largest front / column count in the launch
    int solve_bs;               // workgroup size of the block solve kernels for this launch (128 or 256 = the default 512-thread one)
    int ntiny;                  // one-wave launches: the last ntiny fronts have f <= 8 (eight to a wave in the solves)
    int tile_begin, ntiles;     // Schur tiles of this launch's fronts
    int tile_nc = 0;            // panel columns per tile, averaged over the launch's tiles (the depth of a tile's product)
    int tinv_begin, tinv_count, tinv_ncmax;   // this launch's supernodes that need T = L11^{-1}
    int nsliced;                // the last nsliced fronts of a block-class launch are factorised in row slices ...
    int slice_begin, slice_count;   // ... their slice records in d_sdesc
    size_t lds_sliced;
    int ntall;                  // the last ntall fronts of a block-class launch are too tall for the block sweep kernels' LDS
                                //   (beyond ~10 000 rows): k_fwd_tall / k_bwd_tall; such a launch stays out of the persistent kernels
    int level;                  // tree level: a level has at most one block-class launch, followed by its one-wave launch
    RecSeg rec;                 // packed sweep records of the launch's fronts (kernels.hpp): class 0 for a block-class launch, 1 and 2 for a one-wave one
};

static constexpr size_t kLdsCap = 160 * 1024 - 512;
static constexpr int kMaxNR = 4;          // right-hand sides the single-column solve path takes in one sweep (1, 2 or 4)
// (environment settings: knobs.hpp -- one table, read once per process)
// grid of the side-stream W formation while the tree is still being factorised: 3/8 of the CUs (96 of 256) unless set


// WHAT MAY RUN BESIDE WHAT ON ONE DEVICE, ACROSS HANDLES.  Three mechanisms of this library put kernels on the device
// whose workgroups WAIT for other workgroups: the factorisation's overlap mode (panel workgroups that hold a CU each
// while they wait for their children's tiles on ANOTHER stream; tiles behind a gate: enqueue_factor), the persistent
// top-of-tree sweep kernel (all its workgroups resident: k_top_solve) and the chained sweep launches (a workgroup waits
// for lower-index workgroups of its own grid: chain_kernels.hip).  Each is safe beside kernels that end by themselves;
// none is safe beside another handle's waiting kernel, and the overlap mode -- whose waits cross streams -- is not even
// safe beside another handle's plain cross-stream dependencies.  All three were seen in round 4 as 50 ms give-ups with
// two and three handles on their own streams:
//   * handle 0's persistent kernel had part of its workgroups resident, the rest waiting for CUs held by handle 1's
//     resident panels, which waited for tiles behind their gate, which waited for handle 1's remaining panels, which needed
//     the CUs handle 0's workgroups sat on;
//   * a chained grid's workgroups are dealt to the eight XCDs round-robin and each XCD starts its share in order, so the
//     grid's lowest unfinished workgroup may be the one NOT yet resident on an XCD whose CUs another handle's waiting
//     workgroups hold, while its own resident workgroups hold the CUs those are waiting for;
//   * HIP maps a process's streams onto a few hardware queues, and a queue runs its packets in order: handle B's main
//     stream waiting (an event) for B's W-formation kernel, parked in the queue behind handle A's tile kernel, which waits
//     for A's panel kernel, which sits in the queue behind B's wait.
// So per device the operations that are not plain single-stream work are classed and admitted under a mutex:
//   X  overlapped factorisation (in-kernel waits across streams)    admitted iff every OTHER handle is idle (no M, S, X in flight)
//                                                                     and has been for 20 ms (it is not in a loop beside this one:
//                                                                     an X in flight takes the others' M away, which costs a busy
//                                                                     handle more than the mode gives -- 64 SOCPs on three handles:
//                                                                     3.7 k problems/s with X admitted on idleness alone, 4.4 k so)
//   S  sweep through the persistent / chained kernels                admitted iff no other handle has an S or X in flight
//   M  factorisation with side-stream W formation (no in-kernel wait) admitted iff no other handle has an X in flight
// "in flight" = enqueued and not yet past the handle's three events (main, tile and W-formation stream); handles that
// share one stream never conflict (the stream serialises them).  An operation that is not admitted is not switched off:
// THAT factorisation runs level by level on the main stream alone, THAT sweep level by level -- kernels and packets that
// depend on earlier work of their own stream only, safe beside anything.  Under contention (bench.py --mode problems:
// three busy handles) X is rarely admitted and the handles settle on M + one S at a time, which is what round 3 reached
// with HIPKKT_FACTOR_OVERLAP=0 set by hand.  A handle alone on its device (the common case) pays a mutex and no event;
// the first time a second handle appears on a device the device is synchronised once, so that no operation without
// events is in flight.
enum DevOp { kOpNone = 0, kOpM = 1, kOpS = 2, kOpX = 3 };
struct DevRec {
    const void* eng = nullptr;
    hipStream_t stream = nullptr;
    int kind = kOpNone;                          // strongest class among the handle's operations still in flight
    bool enqueuing = false;                      // between admission and the recording of its events
    hipEvent_t done_main = nullptr, done_tiles = nullptr, done_side = nullptr;
    std::chrono::steady_clock::time_point last_seen{};   // the handle's last factorisation / sweep call (admitted or not)
};
struct DevState { std::vector<DevRec> recs; };
static std::mutex g_dev_mu;
static std::map<int, DevState> g_dev;

// ------------------------------------------------------------------------------------
//  The numeric engine shared by both API levels
// ------------------------------------------------------------------------------------
class LDLEngine {
public:
    Symbolic S;
    hipStream_t stream = nullptr;
    int device_id = 0;
    double dyn_eps, dyn_delta;

    LDLEngine(int N, const int64_t* colptr, const int64_t* rowval, int base, const std::vector<int>& dsigns,
              const hipkkt_settings& st)
    {
        SymbolicOptions opt;
        opt.ordering = st.ordering;
        opt.amd_dense_scale = st.amd_dense_scale > 0 ? st.amd_dense_scale : 1.5;
        if (st.nd_leaf_size > 0) opt.nd_leaf_size = st.nd_leaf_size;
        opt.user_perm = st.user_perm;
        apply_knobs(opt);
        // user_perm arrives in the caller's index base; analyse() applies `base` to it
        analyse(N, colptr, rowval, base, opt, S);
        panel_cap = opt.panel_cap;
        panel_max_slices = std::max(1, opt.panel_max_slices);
        dyn_eps = st.dynamic_regularization_eps;
        dyn_delta = st.dynamic_regularization_delta;
        build_schedule();
        upload(dsigns);
        if (knobs().verbose) {
            int nblock = 0, nsl_fronts = 0, ov_slices = 0;
            for (const Launch& L : launches) if (!L.small) { nblock += L.count; nsl_fronts += L.nsliced; }
            if (overlap_wanted()) for (size_t q = ov_first; q < launches.size(); ++q) ov_slices += launches[q].slice_count;
            std::fprintf(stderr, "[hipkkt] N %d, %d supernodes in %zu levels (%zu launches), %d block-class fronts, %d of them in "
                         "%zu row slices; persistent solve set: last %zu launches, %d fronts on %d workgroups, %d (front, slice) tasks; "
                         "factorisation overlap: last %zu launches (%d row slices in them)\n",
                         S.N, S.nsuper, S.levels.size(), launches.size(), nblock, nsl_fronts, slice_list.size(), top_launches,
                         top_count, top_ntask > 0 ? top_sgrid : top_grid, top_ntask, overlap_wanted() ? launches.size() - ov_first : (size_t)0,
                         ov_slices);
            {
                int ntall_all = 0;
                for (const Launch& L : launches) ntall_all += L.ntall;
                if (ntall_all) std::fprintf(stderr, "[hipkkt] %d fronts too tall for the block sweep kernels (k_fwd_tall / k_bwd_tall)\n", ntall_all);
            }
            if (knobs().verbose >= 2)
                for (size_t q = 0; q < launches.size(); ++q) {
                    const Launch& L = launches[q];
                    int fmin = 1 << 30, ncmin = 1 << 30;
                    double flops = 0;
                    for (int t = L.begin; t < L.begin + L.count; ++t) {
                        const int sn = sched[(size_t)t], nc = S.sn_start[sn + 1] - S.sn_start[sn], f = front_size(sn);
                        fmin = std::min(fmin, f); ncmin = std::min(ncmin, nc);
                        flops += (double)nc * (f - nc) * (f - nc) + (double)nc * nc * (f - nc) + (double)nc * nc * nc / 3;
                    }
                    std::fprintf(stderr, "[hipkkt] launch %zu level %d: %d fronts (%s), f %d..%d, nc %d..%d, %d sliced into %d, %d tiles (%d columns deep on average), %.3f GF\n",
                                 q, L.level, L.count, L.small ? "one wave" : "block", fmin, L.fmax, ncmin, L.ncmax, L.nsliced, L.slice_count,
                                 L.ntiles, L.tile_nc, flops * 1e-9);
                }
            if (overlap_wanted())
                for (size_t q = ov_first; q < launches.size(); ++q)
                    std::fprintf(stderr, "[hipkkt] overlap admission: launch %zu: %d panel workgroups, %d tiles behind a gate, %d CUs\n", q,
                                 ov_group_of[q] >= 0 ? ov_groups[(size_t)ov_group_of[q]].count : launches[q].count - launches[q].nsliced + launches[q].slice_count,
                                 launches[q].ntiles, n_cus);
        }
    }

    // Both sequences are static (no pivoting, fixed structure), so they can be captured once into
    // hipGraphs and replayed (HIPKKT_GRAPH=1), with the T = L11^{-1} kernels on a parallel branch.
    void factor(const double* d_Kval, const double* d_eps)
    {
        const bool want_stamps = knobs().stamps;
        // measured on MI355X/ROCm 7.2: replay is ~25 % slower than eager launches for these ~100-node
        // chains (the work is GPU-latency-bound, not host-bound), so graphs are opt-in
        const bool no_graph = !knobs().graph;
        if (want_stamps || no_graph || n_factor_calls++ == 0) {
            enqueue_factor(d_Kval, d_eps, stream, nullptr, want_stamps);
            return;
        }
        auto key = std::make_pair((const void*)d_Kval, (const void*)d_eps);
        auto it = factor_graphs.find(key);
        if (it == factor_graphs.end()) {
            ensure_capture_streams();
            HIP_CHECK(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
            enqueue_factor(d_Kval, d_eps, cap_stream, cap_side, false);
            hipGraph_t g;
            HIP_CHECK(hipStreamEndCapture(cap_stream, &g));
            hipGraphExec_t ex;
            HIP_CHECK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            it = factor_graphs.emplace(key, ex).first;
        }
        HIP_CHECK(hipGraphLaunch(it->second, stream));
    }

    // d_b, d_x in the caller's (original) ordering; may alias.  allow_top: the caller will synchronise and ask
    // top_gave_up() afterwards (and repeat the solve if so); otherwise the launch-per-level path is used.
    // nr = 1, 2 or 4 right-hand sides in ONE sweep (column c at d_b + c ldb / d_x + c ldx): see supports_nr().
    void solve(const double* d_b, double* d_x, bool allow_top = false, int nr = 1, int64_t ldb = 0, int64_t ldx = 0)
    {
        const bool no_graph = !knobs().graph;
        if (nr > 1) {
            if (!supports_nr(nr)) throw ArgError("solve: this structure takes one right-hand side per sweep");
            reserve_nr(nr);
        }
        if (no_graph || nr > 1 || n_solve_calls++ == 0) {
            // the persistent kernel AND the chained launches hold workgroups that wait for other workgroups: class S
            // (DevOp); not admitted, this sweep goes level by level
            const bool waits = allow_top && ((!top_disabled && top_launches > 0) || (!chain_disabled && chain_from < launches.size()));
            const bool tok = waits && claim_dev(kOpS);
            TokenRelease rel{this, tok};
            enqueue_solve(d_b, d_x, stream, tok, nr, ldb, ldx, tok);
            if (tok) { rel.on = false; dev_enqueued(true); }
            return;
        }
        auto key = std::make_pair((const void*)d_b, (const void*)d_x);
        auto it = solve_graphs.find(key);
        if (it == solve_graphs.end()) {
            ensure_capture_streams();
            HIP_CHECK(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
            enqueue_solve(d_b, d_x, cap_stream, false, 1, 0, 0);
            hipGraph_t g;
            HIP_CHECK(hipStreamEndCapture(cap_stream, &g));
            hipGraphExec_t ex;
            HIP_CHECK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            it = solve_graphs.emplace(key, ex).first;
        }
        HIP_CHECK(hipGraphLaunch(it->second, stream));
    }
    // The single-column kernels' NR-column instances keep NR times the vectors in LDS: possible when every level's
    // share still fits a CU (the (front, slice) kernel of sets with very tall fronts: two columns at most).
    bool supports_nr(int nr) const
    {
        if (nr == 1) return true;
        if (nr != 2 && nr != 4) return false;
        // A set with very tall fronts ((front, slice) kernel): two columns at most, and only the launches BELOW the set
        // have to fit -- the set's own fronts (1531 x 96: one column's vectors fill a CU's LDS in the per-level kernels)
        // go through the persistent kernel, or, where that is not available (no claim, given up), column by column
        // (enqueue_solve).
        if (top_ntask > 0 && (nr != 2 || top_sgrid2 <= 0)) return false;
        const size_t below = top_ntask > 0 ? launches.size() - top_launches : launches.size();
        for (size_t q = 0; q < below; ++q)
            if (!launches[q].small && (launches[q].ntall > 0 || launches[q].lds_solve * (size_t)nr > kLdsCap)) return false;
        // (fronts too tall for the block kernels take one column in the per-level path; inside the (front, slice) set they
        //  are slices like any other)
        return true;
    }

    // nrhs right-hand sides at once: column j of d_B at stride ldb, of d_X at stride ldx (may alias d_B).
    // Every level is one launch (grid.y = block of 16 columns, 8 for the one-wave fronts): the per-level latency
    // that bounds a single solve is paid once for all columns, and every entry of the solve matrices a workgroup
    // fetches is used for all of its columns (solve_kernels.hip, "several right-hand sides").
    void solve_multi(const double* d_B, int64_t ldb, double* d_X, int64_t ldx, int nrhs)
    {
        if (nrhs <= 0) return;
        if (nrhs == 1) { solve(d_B, d_X); return; }
        const int KP = multi_prepare(nrhs);
        launch_permute_in(d_B, ldb, xp_m.p, KP, d_iperm.p, S.N, nrhs, stream);
        multi_sweeps(KP);
        launch_permute_out(d_X, ldx, xp_m.p, KP, d_iperm.p, S.N, nrhs, stream);
        HIP_CHECK(hipGetLastError());
    }
    // the same with ROW-major vectors, N x KP (KP a multiple of 16, padding columns zero); d_X may alias d_B;
    // d_add (nullable, same layout): X = K^{-1} B + add, formed while the solution is permuted back
    void solve_multi_rm(const double* d_B, double* d_X, int KP, const double* d_add = nullptr)
    {
        if (KP <= 0 || (KP & 15)) throw ArgError("solve_multi_rm: KP must be a positive multiple of 16");
        multi_prepare(KP);
        // (r03: the two row permutations -- 2 x (read + write) of N x KP per sweep pair, 5 GB at 512 columns -- are gone: the
        //  forward kernels read a front's own rows of B through the permutation, the backward kernels store the solution
        //  (+ add) to the caller's rows beside the tree-ordered copy the descendants read)
        multi_sweeps(KP, d_B, d_X, d_add);
        HIP_CHECK(hipGetLastError());
    }

private:
    int multi_prepare(int nrhs)
    {
        const int KP = (nrhs + 15) & ~15;
        if (KP / 8 > 65535) throw ArgError("solve_multi: too many right-hand sides per call");
        {   // the block kernels' grid.y is (fronts of a launch / 8) x (column blocks of 16)
            int64_t widest = 0;
            for (const Launch& L : launches) if (!L.small) widest = std::max<int64_t>(widest, (L.count + 7) / 8);
            if (widest * (KP / 16) > 65535) throw ArgError("solve_multi: too many right-hand sides per call for this structure (split the call)");
        }
        wait_w(stream);
        if ((size_t)KP > multi_cap) {
            xp_m.alloc((size_t)S.N * KP);
            uvec_m.alloc(std::max<size_t>(S.rows.size(), 1) * KP);
            multi_cap = (size_t)KP;
        }
        if (!d_iperm.p) d_iperm.upload(S.iperm);
        return KP;
    }
    void multi_sweeps(int KP, const double* b_rm = nullptr, double* out_rm = nullptr, const double* add_rm = nullptr)
    {
        SolveArgs a;
        a.T = tree();
        a.fronts = fronts.p;
        a.tinv = tinv.p;
        a.Dinv = Dinv.p;
        a.b = b_rm;                  // (row-major N x KP in the caller's row order, or null: xp_m holds the permuted right-hand sides)
        a.out = out_rm;
        a.add = add_rm;
        a.xp = xp_m.p;
        a.uvec = uvec_m.p;
        a.ld_b = a.ld_out = a.ld_xp = a.ld_uvec = 0;
        a.top_limit = 5000000;       // (the multi-column path has no persistent kernel)
        a.tk_pos = nullptr; a.tk_sl = nullptr; a.tbase = nullptr; a.xf = nullptr; a.chain_cnt = nullptr; a.recs = nullptr; a.tall_ws = nullptr;
        launch_pull_leaves_multi(a, n_pull_rows, KP, stream);        // (the pulled leaves' terms, summed per receiving row)
        for (const Launch& L : launches) launch_fwd_multi(a, L.begin, L.count, L.small, L.ncmax, KP, stream);
        for (size_t q = launches.size(); q-- > 0;) {
            const Launch& L = launches[q];
            launch_bwd_multi(a, L.begin, L.count, L.small, L.ncmax, KP, stream, L.level == 0);
        }
    }

public:
    ~LDLEngine()
    {
        engine_gone();
        for (auto& kv : factor_graphs) (void)hipGraphExecDestroy(kv.second);
        for (auto& kv : solve_graphs) (void)hipGraphExecDestroy(kv.second);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (cap_side) (void)hipStreamDestroy(cap_side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_join2) (void)hipEventDestroy(ev_join2);
        if (ev_side) (void)hipEventDestroy(ev_side);
        if (ov_stream) (void)hipStreamDestroy(ov_stream);
        if (ev_ov_fork) (void)hipEventDestroy(ev_ov_fork);
        if (ev_ov_join) (void)hipEventDestroy(ev_ov_join);
        if (ev_dev_done) (void)hipEventDestroy(ev_dev_done);
    }

private:
    std::map<std::pair<const void*, const void*>, hipGraphExec_t> factor_graphs, solve_graphs;
    hipStream_t cap_stream = nullptr, cap_side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_side = nullptr;
    long n_factor_calls = 0, n_solve_calls = 0;

    void ensure_capture_streams()
    {
        if (cap_stream) return;
        HIP_CHECK(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&cap_side, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_join2, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
    }

    // Do two streams really run side by side?  HIP deals a process's streams to a few hardware queues, and two streams on
    // one queue execute in submission order: the side stream's W formation would then sit in the middle of the
    // factorisation (seen: cfg2's 1.75 ms became 2.69 in a process that had created a few streams before), the tile
    // stream's kernels would not overlap their panels.  A kernel on `a` waits up to 2 ms for a kernel submitted AFTER it
    // on `b` (factor_kernels.hip, launch_concurrency_probe).
    bool streams_concurrent(hipStream_t a, hipStream_t b)
    {
        int* w = flags.p + 8;
        launch_zero_ints(w, 2, a);
        HIP_CHECK(hipStreamSynchronize(a));
        HIP_CHECK(hipStreamSynchronize(b));
        launch_concurrency_probe(w, a, b);
        int seen[2] = {0, 0};
        HIP_CHECK(hipMemcpyAsync(seen, w, sizeof(seen), hipMemcpyDeviceToHost, a));
        HIP_CHECK(hipStreamSynchronize(a));
        HIP_CHECK(hipStreamSynchronize(b));
        return seen[1] != 0;
    }
    // A stream that runs beside every stream of `beside`: `have` if it does (or nullptr: create one), else new streams are
    // tried -- up to six, the rejected ones held until the search ends so that the next creation lands on another queue.
    // *ok says whether the search succeeded; without success the last candidate is returned (everything still works
    // on a shared queue, in submission order).
    template <typename Create>
    hipStream_t stream_beside(hipStream_t have, const std::vector<hipStream_t>& beside, Create&& create, bool* ok)
    {
        std::vector<hipStream_t> rejected;
        hipStream_t cand = have;
        *ok = false;
        for (int attempt = 0; attempt < 6; ++attempt) {
            if (!cand && create(&cand) != hipSuccess) { (void)hipGetLastError(); cand = nullptr; break; }
            bool good = true;
            for (hipStream_t o : beside) good = good && streams_concurrent(o, cand);
            if (good) { *ok = true; break; }
            if (attempt == 5) break;
            rejected.push_back(cand);
            cand = nullptr;
        }
        if (!cand && !rejected.empty()) { cand = rejected.back(); rejected.pop_back(); }
        for (hipStream_t r : rejected) { (void)hipStreamSynchronize(r); (void)hipStreamDestroy(r); }
        return cand;
    }
    hipStream_t sides_for = (hipStream_t)(-1);     // the main stream the side / tile streams were chosen for
    bool side_concurrent = false;
    // (eager path only) the W-formation stream and the tile stream, each beside the main stream and beside each other
    void choose_side_streams(hipStream_t st, bool want_tiles)
    {
        if (sides_for == st && (!want_tiles || ov_stream || ov_disabled)) return;
        wait_w(st);
        auto plain = [](hipStream_t* out) { return hipStreamCreateWithFlags(out, hipStreamNonBlocking); };
        const bool new_main = sides_for != st;
        // The tile stream first: created before the capture / W-formation streams, as it always was (with the W-formation
        // stream created first the factorisation of cfg2 measured 13 us longer: which hardware queue a stream lands on
        // follows the order of creation, and the queues are not served alike)
        if (want_tiles && !ov_disabled && (new_main || !ov_stream)) {
            // (HIPKKT_OV_CU_MASK=1 confines the tile stream to every other CU.  Measured: on this stack a CU-masked
            // stream slows EVERY stream of the process down as if all of them were masked -- residual 0.032 -> 0.054 ms,
            // sweep 0.29 -> 0.37 ms -- so the default is a plain stream.)
            const bool cu_mask = knobs().ov_cu_mask;
            hipDeviceProp_t prop;
            HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
            std::vector<uint32_t> mask((size_t)std::max((prop.multiProcessorCount + 31) / 32, 1), 0x55555555u);
            auto masked = [&](hipStream_t* out) { return hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data()); };
            const bool had = ov_stream != nullptr;
            std::vector<hipStream_t> beside{st};
            if (cap_side && !new_main) beside.push_back(cap_side);           // (a W-formation stream already chosen for st)
            if (cu_mask) ov_stream = stream_beside(ov_stream, beside, masked, &ov_concurrent);
            else ov_stream = stream_beside(ov_stream, beside, plain, &ov_concurrent);
            if (!ov_stream) {
                ov_disabled = true;
                std::fprintf(stderr, "[hipkkt] no stream for the Schur tiles: factorisation overlap off\n");
            } else if (!had) {
                HIP_CHECK(hipEventCreateWithFlags(&ev_ov_fork, hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&ev_ov_join, hipEventDisableTiming));
            }
            if (ov_stream && !ov_concurrent && knobs().verbose)
                std::fprintf(stderr, "[hipkkt] the main and the tile stream share a hardware queue: no merged panel kernel for this handle\n");
        }
        ensure_capture_streams();
        if (new_main) {
            std::vector<hipStream_t> beside{st};
            if (ov_stream) beside.push_back(ov_stream);
            cap_side = stream_beside(cap_side, beside, plain, &side_concurrent);
            if (!side_concurrent && knobs().verbose)
                std::fprintf(stderr, "[hipkkt] no stream beside the main stream for the W formation: it will run in submission order\n");
        }
        sides_for = st;
    }

    // side != nullptr: put the T = L11^{-1} kernels on that stream, forked after each level's panels
    void enqueue_factor(const double* d_Kval, const double* d_eps, hipStream_t st, hipStream_t side, bool want_stamps)
    {
        wait_w(st);                  // (a refactorisation without a solve in between: the side stream still reads the fronts)
        if (ov_join_pending) {
            // the previous factorisation's tile stream (see the end of the overlapped launches below) before its counters
            // and update blocks are touched again
            if (side) HIP_CHECK(hipEventSynchronize(ev_ov_join));        // (a stream being captured cannot wait for it)
            else HIP_CHECK(hipStreamWaitEvent(st, ev_ov_join, 0));
            ov_join_pending = false;
        }
        ZeroList zl;
        zl.add(flags.p, 3);
        // overlap mode (eager launches only): the top launches' Schur tiles on their own stream, ordered by counters
        // in memory instead of kernel boundaries.  On by default (HIPKKT_FACTOR_OVERLAP=0 turns it off).  Measured on cfg2:
        // the tiles hide completely behind the panels (a top level costs its panel kernel instead of panel + tiles:
        // factorisation 1.85 -> 1.79 ms); it pays since the solve matrices W are formed early enough on the side stream
        // (3.43 -> 3.365 ms per step; with the round's earlier, slower W formation the first sweep waited for W by as
        // much as the factorisation ended earlier).
        const bool want_ov = overlap_wanted();
        const bool use_ov = want_ov && !side && !want_stamps && !ov_disabled && ov_first < launches.size();
        // Whatever this factorisation does on more than ONE stream needs admission (DevOp): the overlap mode (class X:
        // its kernels wait across streams) or, failing that, the side stream's W formation with its fork / join events
        // (class M).  Not admitted, it keeps to its main stream, where every packet depends on earlier packets of the same
        // stream only.
        const bool no_overlap = knobs().no_overlap;
        const bool side_w = !side && !no_overlap && !want_stamps && late_launches > 0 && late_launches < launches.size();
        const bool tok_x = use_ov && claim_dev(kOpX);
        const bool tok = tok_x || (side_w && claim_dev(kOpM));
        TokenRelease ov_release{this, tok};      // (the "enqueuing" mark ends with this call, however it ends)
        // the side streams (W formation, Schur tiles) are chosen once per main stream, each probed to run BESIDE it.  The
        // tile stream only once an overlapped factorisation has been admitted: a stream that exists changes which
        // hardware queues the process's other streams share (three busy handles, each with an idle tile stream: 3.7 k
        // problems/s instead of 4.2 k)
        if (!side && !want_stamps) choose_side_streams(st, tok_x);
        const bool ov_on = tok_x && !ov_disabled && ov_stream != nullptr;
        // The merged panel kernels of the narrow top (below) wait for tile kernels that are submitted BEHIND them: that
        // needs the two streams on different hardware queues (ov_concurrent: choose_side_streams' probe; seen without it,
        // eight handles in one process: two of them waited for their 50 ms bound).  Without concurrency every level keeps
        // its own panel kernel, whose waits only ever look at work submitted earlier (panel L, tiles L, panel L + 1, ...:
        // correct on one queue as well).
        // (the merged kernels only where the two streams run side by side)
        auto group_of = [&](size_t q) -> const MergeGroup* {
            return (ov_concurrent && q < ov_group_of.size() && ov_group_of[q] >= 0) ? &ov_groups[(size_t)ov_group_of[q]] : nullptr;
        };
        const size_t merge_from = (ov_concurrent && !ov_groups.empty()) ? ov_groups.back().first : ~(size_t)0;
        if (ov_on) {
            zl.add(d_ov_prog.p, S.nsuper);
            zl.add(d_ov_done.p, S.nsuper);
            if (!slice_list.empty()) zl.add(d_ov_sprog.p, (int)slice_list.size());
            zl.add(d_ov_started.p, (int)launches.size());
        }
        launch_zero_ints_multi(zl, st);
        FactorArgs a;
        a.T = tree();
        a.Kval = d_Kval;
        a.eps = d_eps;
        a.fronts = fronts.p;
        a.upd = upd.p;
        a.Dinv = Dinv.p;
        a.flags = flags.p;
        a.dyn_eps = dyn_eps;
        a.dyn_delta = dyn_delta;
        a.ov_prog = d_ov_prog.p; a.ov_done = d_ov_done.p; a.ov_ntiles = d_ov_ntiles.p; a.ov = 0;
        a.ov_sprog = d_ov_sprog.p; a.ov_sbase = d_ov_sbase.p;
        a.ov_started = d_ov_started.p; a.ov_slot = 0;
        const long long ov_limit = knobs().ov_test_limit;
        a.ov_limit = ov_limit;
        a.stamps = nullptr;
        a.stamp_row = 0;
        if (want_stamps) {
            if (!stamps.p) { stamps.alloc(launches.size() * 16); }
            stamps.zero(st);
            a.stamps = (long long*)stamps.p;
        }
        int li = 0;
        bool forked = false;
#ifdef HIPKKT_EXPERIMENTS
        // EXPERIMENT (timing only; without the gate the overlap mode's forward progress rests on submission order again)
        static const bool no_gate = std::getenv("HIPKKT_EXPERIMENT_NO_GATE") != nullptr;
#else
        constexpr bool no_gate = false;
#endif
#ifdef HIPKKT_EXPERIMENTS
        // EXPERIMENT (timing only, wrong results unless K stays the same between factorisations; compiled in only with
        // -DHIPKKT_EXPERIMENTS, never in the default build): skip W formation after n factorisations
        static const int skip_w_after = std::getenv("HIPKKT_EXPERIMENT_SKIP_WINV") ? std::atoi(std::getenv("HIPKKT_EXPERIMENT_SKIP_WINV")) : -1;
        const bool skip_w = skip_w_after >= 0 && n_skipw_calls++ >= skip_w_after;
#else
        const bool skip_w = false;
#endif
        // (every W formation of this factorisation goes through form_w)
        auto form_w = [&](const int* list, int count, int ncmax, hipStream_t on, int max_blocks) {
            const size_t i0 = (size_t)(list - d_tinv_list.p);
            const int nsmall = i0 + (size_t)count < tinv_small_prefix.size() ? tinv_small_prefix[i0 + (size_t)count] - tinv_small_prefix[i0] : 0;
            if (!skip_w) launch_tinv(a.T, fronts.p, tinv.p, list, count, ncmax, on, max_blocks, nsmall);
        };
        // eager mode: once the tree narrows to its top levels most CUs idle, so the solve matrices
        // W = [T; M] of everything below are formed on a side stream meanwhile (HIPKKT_NO_OVERLAP=1 disables)
        const size_t nl = launches.size();
        const size_t first_top = (tok && side_w) ? nl - late_launches : nl;
        bool eager_fork = false;
        int w_done = 0;
        for (size_t q = 0; q < nl; ++q) {
            const Launch& L = launches[q];
            a.stamp_row = li++;
            // fork points of the side stream: a few levels before the narrow top (the bulk of the fronts), at the narrow
            // top, and a few levels before the root -- behind the tree only the last levels' handful of fronts is left,
            // which the next sweep's bottom levels hide
            // (with the top launches' panels merged into one kernel, the last fork sits in front of that kernel: an event
            //  behind it would wait for the whole top of the tree)
            // (with merged panel kernels every run's first launch is a fork point: the W of everything factorised so far is
            //  formed beside the run.  With the last run's start as the only late fork, cfg3 and cfg5 formed the W of their
            //  widest levels behind the tree and the first sweep waited for it: 0.1 and 0.37 ms per step)
            const bool runs = ov_on && merge_from < nl;
            const size_t tail_fork = runs ? nl : (nl >= (size_t)std::max(0, knobs().winv_tail) ? nl - (size_t)std::max(0, knobs().winv_tail) : 0);
            const MergeGroup* gq = runs ? group_of(q) : nullptr;
            const bool run_forks = knobs().winv_run_forks;
            const bool run_first = gq && gq->first == q && (run_forks || q == merge_from);
            // (a fork point inside a merged run moves to the run's first launch: an event recorded behind the run's kernel
            //  would wait for the whole run)
            auto fork_at = [&](size_t want) {
                if (want >= nl) return want;
                const MergeGroup* g = ov_on ? group_of(want) : nullptr;
                return g ? g->first : want;
            };
            const bool fork_here = first_top < nl && (q == fork_at(first_top >= (size_t)std::max(0, knobs().winv_early) ? first_top - (size_t)std::max(0, knobs().winv_early) : nl) ||
                                                      q == fork_at(first_top) || (q > first_top && (run_first || q == tail_fork)));
            if (fork_here && launches[q].tinv_begin > w_done) {
                ensure_capture_streams();
                HIP_CHECK(hipEventRecord(ev_fork, st));
                HIP_CHECK(hipStreamWaitEvent(cap_side, ev_fork, 0));
                // a bounded grid: the top panels need whole CUs (their LDS), which a full-width launch would hold
                form_w(d_tinv_list.p + w_done, launches[q].tinv_begin - w_done, tinv_ncmax, cap_side, side_winv_blocks);
                eager_fork = true;
                w_done = launches[q].tinv_begin;
                // the fronts below the narrow top are used by the next sweep's first launches: the factorisation ends
                // with a wait for this event (normally long past by then)
                if (q <= first_top) HIP_CHECK(hipEventRecord(ev_side, cap_side));
            }
            if (ov_on && q >= ov_first) {
                // Panels on the main stream, the level's Schur tiles on the overlap stream,
                // submitted in this order -- panel L, tiles L, panel L+1, ... -- which is correct even if the two streams
                // share a hardware queue.  Tiles of level L become ready only when the tiles of level L-1 have finished,
                // i.e. after panel L-1 ended and panel L got ready: the panels reach the CUs first.  (A tile workgroup
                // that waits for its panel holds LDS a panel workgroup needs; the first overlapped level therefore
                // starts its tiles only after its panels have finished.)
                a.ov = 1;
                a.nbk = L.nbk;
                // The panels of a run of narrow launches go out as ONE kernel (ov_groups: a few dozen workgroups): every one of
                // them is resident from the start, zeroes its panel, scatters K and fetches its item lists while its
                // children are still being factorised, and then waits for its children's tiles -- a level of the narrow
                // top costs its critical path (children's assembly, block loop, tiles) without a kernel boundary and
                // launch ramp in between.  Workgroups are dispatched in schedule order, i.e. lower levels first.
                const MergeGroup* g = group_of(q);
                const bool merged = g != nullptr;
                a.ov_slot = merged ? (int)g->first : (int)q;
                if (!merged) {
                    launch_panel(a, L.begin, L.count - L.nsliced, L.bs_panel, L.lds_panel, st);
                    launch_panel_sliced(a, L.slice_begin, L.slice_count, L.lds_sliced, st);
                } else if (q == g->first) {
                    if (g->sliced) launch_panel_sliced(a, L.slice_begin, g->count, g->lds, st);
                    else launch_panel(a, L.begin, g->count, 1024, g->lds, st);
                }
                if (q == ov_first) {
                    HIP_CHECK(hipEventRecord(ev_ov_fork, st));
                    HIP_CHECK(hipStreamWaitEvent(ov_stream, ev_ov_fork, 0));
                } else if (L.ntiles > 0 && !no_gate) {
                    // the gate: this launch's tiles are released once all its panel workgroups are resident (k_ov_gate)
                    if (merged) launch_ov_gate(d_ov_started.p + g->first, g->count, flags.p + 2, ov_limit, ov_stream);
                    else launch_ov_gate(d_ov_started.p + q, L.count - L.nsliced + L.slice_count, flags.p + 2, ov_limit, ov_stream);
                }
                launch_schur(a, (const int2*)d_tiles.p, L.tile_begin, L.ntiles, ov_stream, L.ntiles);
                a.ov = 0;
                // (the tile stream's join event: behind the loop -- the last W formation may ride on that stream)
                continue;
            }
            if (L.small) {
                launch_front_wave(a, L.begin, L.count - L.ntiny, L.slice, st);
                launch_front_tiny(a, L.begin + L.count - L.ntiny, L.ntiny, st);
            } else {
                a.nbk = L.nbk;
                launch_panel(a, L.begin, L.count - L.nsliced, L.bs_panel, L.lds_panel, st);
                launch_panel_sliced(a, L.slice_begin, L.slice_count, L.lds_sliced, st);
                if (L.tinv_count > 0 && side) {
                    HIP_CHECK(hipEventRecord(ev_fork, st));
                    HIP_CHECK(hipStreamWaitEvent(side, ev_fork, 0));
                    form_w(d_tinv_list.p + L.tinv_begin, L.tinv_count, L.tinv_ncmax, side, 0);
                    forked = true;
                }
                launch_schur(a, (const int2*)d_tiles.p, L.tile_begin, L.ntiles, st, 0, L.tile_nc);
            }
        }
        if (forked) {
            HIP_CHECK(hipEventRecord(ev_join, side));
            HIP_CHECK(hipStreamWaitEvent(st, ev_join, 0));
        } else if (eager_fork) {
            // the solve matrices of the top fronts are formed on a side stream as well, behind the tree: the next
            // sweep only needs them when it reaches the top of the tree (enqueue_solve waits for ev_join there), so
            // their formation hides behind the sweep's bottom levels instead of ending the factorisation.
            // In the overlap mode it rides on the TILE stream, which has nothing left to do by now (a root has no tiles):
            // on the W stream it queued behind the previous fork's formation -- 170 us on a bounded grid, started when the
            // last run of panels started -- and the first sweep's persistent kernel waited ~50 us for it every step.
            const int done = w_done;
            hipStream_t last_on = ov_on ? ov_stream : cap_side;
            HIP_CHECK(hipStreamWaitEvent(st, ev_side, 0));
            HIP_CHECK(hipEventRecord(ev_fork, st));
            HIP_CHECK(hipStreamWaitEvent(last_on, ev_fork, 0));
            form_w(d_tinv_list.p + done, (int)tinv_list.size() - done, tinv_ncmax, last_on, 0);
            HIP_CHECK(hipEventRecord(ev_join, cap_side));
            if (ov_on) { HIP_CHECK(hipEventRecord(ev_join2, ov_stream)); w2_pending = true; }
            w_pending = true;
        } else {
            // one launch over every supernode, after the tree (all of them independent)
            form_w(d_tinv_list.p, (int)tinv_list.size(), tinv_ncmax, st, 0);
        }
        if (ov_on) {
            // The tile stream is joined by the NEXT factorisation, not by this one's tail: every tile has been counted
            // by its parent's panel (a root has no tiles), so when the main stream's last panel kernel ends all
            // tile data has long been stored; what is left on the tile stream are workgroups on their way out (and the
            // last W formation, which the sweeps wait for by ev_join2).  A cross-stream wait here cost ~17 us between the
            // last panel and the status kernel, every step.
            HIP_CHECK(hipEventRecord(ev_ov_join, ov_stream));
            ov_join_pending = true;
        }
        if (tok) {
            ov_release.on = false;
            dev_enqueued(true);                  // (records the event that, with ev_ov_join / ev_join, says this factorisation has left its streams)
        }
        HIP_CHECK(hipGetLastError());
        if (want_stamps) {
            std::vector<long long> h(launches.size() * 16);
            HIP_CHECK(hipMemcpyAsync(h.data(), stamps.p, h.size() * 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            static int printed = 0;
            if (printed++ == 2) {
                for (size_t r = 0; r < launches.size(); ++r) {
                    const long long* q = &h[r * 16];
                    if (launches[r].small) continue;
                    std::fprintf(stderr, "[stamps2] launch %zu: wave 0's items %.1f us (first batch's loads %.1f us), then all waves %.1f us\n", r, (q[7] - q[6]) * 0.01, (q[15] - q[6]) * 0.01, (q[3] - q[7]) * 0.01);
                    std::fprintf(stderr, "[stamps] launch %zu fronts %d f=%lld nc=%lld kids=%lld | zero %.1f K %.1f kids %.1f factor %.1f "
                                 "(diag %.1f trsm %.1f trail %.1f) store %.1f us  clk %.0f MHz\n", r, launches[r].count, q[11], q[12], q[13],
                                 (q[1] - q[0]) * 0.01, (q[2] - q[1]) * 0.01, (q[3] - q[2]) * 0.01, (q[4] - q[3]) * 0.01,
                                 q[8] * 0.01, q[9] * 0.01, q[10] * 0.01, (q[5] - q[4]) * 0.01,
                                 (double)q[14] / ((q[5] - q[0]) * 0.01));
                }
            }
        }
    }

    void reserve_nr(int nr)
    {
        if ((size_t)nr <= nr_cap) return;
        wait_w(stream);
        HIP_CHECK(hipStreamSynchronize(stream));
        // (solve graphs captured so far have the old xp / uvec baked in)
        for (auto& kv : solve_graphs) (void)hipGraphExecDestroy(kv.second);
        solve_graphs.clear();
        xp.alloc((size_t)S.N * nr);
        uvec.alloc(std::max<size_t>(S.rows.size(), 1) * nr);
        nr_cap = (size_t)nr;
    }
    int top_grid_for(int nr)
    {
        if (nr == 1) return top_grid;
        if (top_ntask > 0) return nr == 2 ? top_sgrid2 : 0;
        int& g = top_grid_nr[nr == 2 ? 0 : 1];
        if (g < 0) g = std::min(top_grid, top_solve_capacity_nr(top_lds * (size_t)nr, nr));
        return g;
    }

    // use_top: the persistent kernel may be used (claimed); allow_chain: the caller reads the abort word afterwards and
    // repeats the solve if a bounded wait expired (the same promise use_top implies), so the chained launches may be used
    void enqueue_solve(const double* d_b, double* d_x, hipStream_t st, bool use_top, int nr, int64_t ldb, int64_t ldx, bool allow_chain = false)
    {
        SolveArgs a;
        a.T = tree();
        a.fronts = fronts.p;
        a.tinv = tinv.p;
        a.Dinv = Dinv.p;
        a.b = d_b;
        a.out = d_x;
        a.xp = xp.p;
        a.uvec = uvec.p;
        const long long top_limit = knobs().top_test_limit;
        a.top_limit = top_limit;
        // diagnostic (HIPKKT_TOP_STAMPS=n): the n-th single-column sweep over the full persistent set records eight time
        // stamps per front and direction, printed per level afterwards (this call then synchronises)
        const int stamp_call = knobs().top_stamps;
        a.top_stamps = nullptr;
        bool stamp_now = false;
        a.ld_b = ldb; a.ld_out = ldx; a.ld_xp = S.N; a.ld_uvec = (int64_t)std::max<size_t>(S.rows.size(), 1);
        a.tk_pos = d_tk_pos.p; a.tk_sl = d_tk_sl.p; a.tbase = d_tbase.p; a.xf = xf.p; a.add = nullptr;
        a.chain_cnt = nullptr;       // (set below when this sweep chains its lower levels)
        a.recs = d_recs.p ? reinterpret_cast<const char*>(d_recs.p) : nullptr;
        a.tall_ws = tall_ws.p;
        const bool no_top = knobs().no_top;
        if (nr > 1 && top_ntask > 0 && (no_top || !use_top || top_disabled || top_sgrid2 <= 0)) {
            // two columns through a set with very tall fronts need its persistent kernel (supports_nr): without it, one
            // column after the other
            for (int c = 0; c < nr; ++c) enqueue_solve(d_b + (int64_t)c * ldb, d_x + (int64_t)c * ldx, st, use_top, 1, 0, 0, allow_chain);
            return;
        }
        const size_t nl = launches.size();
        // the persistent kernel covers the last ntl launches.  Right after a factorisation the W of the narrow top is
        // still being formed on the side stream: that sweep keeps the per-level launches for the levels below the
        // narrow top, so that the formation hides behind them
        // Chained launches (chain_kernels.hip): the launches from chain_from on -- the levels with few enough fronts that
        // a launch per level is one front's latency chain, not throughput -- as segments of ONE grid per direction, ordered
        // by counters in memory instead of kernel boundaries; the wide levels below keep their launches.  The persistent
        // kernel keeps its set -- its 1024-thread workgroups park a whole front's matrix items before the wait, a hop
        // costs ~4 us against ~4.2 forward / ~6.3 backward in the 512-thread chained kernel -- and the launches between
        // chain_from and the set are chained (cfg2: levels 3 and 4, 32 -> 26 us forward).  Both need the device's token
        // (allow_chain / use_top: the caller holds it).  HIPKKT_CHAIN=0: off; HIPKKT_CHAIN_TOP=0: chain to the root
        // instead of the persistent kernel (measured: cfg2's sweep pair 0.2675 against 0.260 ms).
        const bool chain_env = knobs().chain;
        const bool chain_top = knobs().chain_top;
        // (a set with very tall fronts keeps its (front, slice) kernel: such fronts do not fit one workgroup's LDS)
        const bool chain_want = chain_env && allow_chain && !chain_disabled && chain_from < nl && chain_lds * (size_t)nr <= 150 * 1024;
        const bool keep_top = chain_want && (chain_top || top_ntask > 0);
        const int tgrid = (no_top || !use_top || top_disabled || (chain_want && !keep_top)) ? 0 : top_grid_for(nr);
        size_t ntl = tgrid > 0 ? top_launches : 0;
        int ncount = top_count;
        // (two columns through a set with very tall fronts: the whole set or nothing -- its lower levels do not fit the
        //  per-level kernels with two columns; the sweep then waits for W at the set's first level)
        if (ntl > 0 && w_pending && late_launches > 0 && late_launches < ntl && !(nr > 1 && top_ntask > 0)) { ntl = late_launches; ncount = late_count; }
        const size_t first_w = nl - std::min(nl, late_launches);    // fronts from here on get their W late (w_pending)
        const bool chain_on = chain_want && chain_from + 2 <= nl - ntl;
        const size_t nper = chain_on ? chain_from : nl - ntl;        // launches [0, nper) go level by level, [nper, nl - ntl) chained
        // a level's block-class launch and the one-wave launch behind it (sched order) go out as one launch
        const bool no_merge = knobs().no_level_merge;
        auto pair_at = [&](size_t q) {      // launches q (block-class) and q + 1 (one-wave) belong to one level
            return !no_merge && q + 1 < nper && !launches[q].small && launches[q].ntall == 0 && launches[q + 1].small &&
                   launches[q].level == launches[q + 1].level;
        };
        // (a level's block-class launch and the one-wave launch behind it as one kernel launch: the records of both)
        auto pair_rec = [](const Launch& Lb, const Launch& Ls) {
            RecSeg r = Ls.rec;
            r.off[0] = Lb.rec.off[0]; r.stride[0] = Lb.rec.stride[0]; r.fmax[0] = Lb.rec.fmax[0];
            return r;
        };
        bool chain_stamp = false;
        ChainArgs ca;
        int ca_wgs = 0;
        size_t ca_lds = 0;
        auto chain_reset = [&]() {
            ca.nseg = 0; ca.lo = 1 << 30; ca.hi = 0; ca.cnt = d_chain.p; ca.done = d_chain.p + S.nsuper; ca.nchild = d_chain_nchild.p;
            ca.abort_word = top_flags.p + 2 * top_nflag; ca.epoch = chain_epoch;
            ca.lo0 = chain_on ? launches[nper].begin : 0;
            ca.nstamp = chain_on ? launches[nl - ntl - 1].begin + launches[nl - ntl - 1].count - ca.lo0 : 0;
            ca_wgs = 0; ca_lds = 0;
        };
        auto chain_flush = [&](bool fwd) {
            if (ca.nseg > 0) {
                a.top_stamps = chain_stamp ? (long long*)top_stamps.p : nullptr;
                if (fwd) launch_fwd_chain(a, ca, ca_wgs, ca_lds, st, nr);
                else launch_bwd_chain(a, ca, ca_wgs, ca_lds, st, nr);
                a.top_stamps = nullptr;
            }
            chain_reset();
        };
        auto chain_add = [&](const Launch& L, bool fwd) {
            const bool blocks = !L.small;
            if (ca.nseg == kChainMaxSeg) chain_flush(fwd);
            ChainSeg& sg = ca.seg[ca.nseg++];
            sg.begin = L.begin;
            sg.nblock = blocks ? L.count : 0;
            sg.nwave = blocks ? 0 : L.count - L.ntiny;
            sg.ntiny = blocks ? 0 : L.ntiny;
            sg.wg0 = ca_wgs;
            sg.leaf = (L.small && L.level == 0) ? 1 : 0;
            sg.rec = L.rec;
            ca_wgs += chain_seg_wgs(sg);
            ca.lo = std::min(ca.lo, L.begin);
            ca.hi = std::max(ca.hi, L.begin + L.count);
            if (blocks) ca_lds = std::max(ca_lds, L.lds_solve);
        };
        if (chain_on) {
            ++chain_epoch;
            a.chain_cnt = d_chain.p;
            // diagnostic (HIPKKT_TOP_STAMPS=n): the n-th single-column chained sweep records six time stamps per
            // block-class front and direction, printed per launch afterwards
            if (stamp_call > 0 && nr == 1 && !w_pending && ++n_stamp_sweeps == stamp_call) {
                const size_t nst = (size_t)(launches[nl - ntl - 1].begin + launches[nl - ntl - 1].count - launches[nper].begin);
                top_stamps.alloc(2 * nst * 8);
                top_stamps.zero(st);
                chain_stamp = true;
            }
        }
        for (size_t q = 0; q < nper; ++q) {
            const Launch& L = launches[q];
            if (q == first_w) wait_w(st);
            if (pair_at(q)) {
                const Launch& Ls = launches[q + 1];
                if (q + 1 == first_w) wait_w(st);
                launch_fwd_level(a, pair_rec(L, Ls), L.begin, L.count, Ls.count - Ls.ntiny, Ls.ntiny, L.solve_bs, L.lds_solve, st, nr);
                ++q;
            } else if (L.small) {
                launch_fwd_small(a, L.rec, L.begin, L.count - L.ntiny, L.ntiny, st, L.level == 0, nr);
            } else {
                launch_fwd(a, L.rec, L.begin, L.count - L.ntall, L.solve_bs, L.lds_solve, st, nr);
                launch_fwd_tall(a, L.begin + L.count - L.ntall, L.ntall, L.fmax, st);       // (fronts beyond the block kernels' LDS)
            }
        }
        if (chain_on) {
            chain_reset();
            for (size_t q = nper; q + ntl < nl; ++q) {
                if (q == first_w) { chain_flush(true); wait_w(st); }
                chain_add(launches[q], true);
            }
            chain_flush(true);
        }
        wait_w(st);
        if (ntl > 0) {
            const Launch& L0 = launches[nl - ntl];
            if (top_ntask > 0) {
                // the set holds very tall fronts: the (front, slice) kernel; positions count from the full set's first front
                const Launch& Lfull = launches[nl - top_launches];
                const int pos0 = top_count - ncount;
                const int task0 = h_tbase[(size_t)pos0];
                bool sl_stamp = false;
                if (stamp_call > 0 && !chain_on && nr == 1 && ntl == top_launches && ++n_stamp_sweeps == stamp_call) {
                    top_stamps.alloc((size_t)2 * (top_ntask - task0) * 8);
                    top_stamps.zero(st);
                    a.top_stamps = (long long*)top_stamps.p;
                    sl_stamp = true;
                }
                launch_top_solve_sliced(a, Lfull.begin, pos0, task0, top_ntask, nr == 2 ? top_sgrid2 : top_sgrid, top_slds, top_flags.p,
                                        top_nflag, ++top_epoch, st, nr);
                a.top_stamps = nullptr;
                if (sl_stamp) {
                    // (front, slice) tasks: per direction the mean time from a task's start to its flags seen, from there
                    // to its publication, and the sweep's span
                    const size_t nt = (size_t)(top_ntask - task0);
                    std::vector<long long> h(2 * nt * 8);
                    HIP_CHECK(hipMemcpyAsync(h.data(), top_stamps.p, h.size() * 8, hipMemcpyDeviceToHost, st));
                    HIP_CHECK(hipStreamSynchronize(st));
                    for (int dir = 0; dir < 2; ++dir) {
                        double wait = 0, work = 0;
                        long long lo = h[(size_t)dir * nt * 8], hi = 0;
                        std::vector<double> works;
                        for (size_t t = 0; t < nt; ++t) {
                            const long long* e = &h[((size_t)dir * nt + t) * 8];
                            wait += (e[2] - e[0]) * 0.01;
                            work += (e[5] - e[2]) * 0.01;
                            works.push_back((e[5] - e[2]) * 0.01);
                            lo = std::min(lo, e[0]); hi = std::max(hi, e[5]);
                        }
                        std::sort(works.begin(), works.end());
                        std::fprintf(stderr, "[top stamps] sliced %s: %zu tasks, span %.1f us; per task: start->flags seen %.2f us (mean), flags seen->publish "
                                     "%.2f us (mean), %.2f (median), %.2f (max)\n", dir == 0 ? "fwd" : "bwd", nt, (hi - lo) * 0.01, wait / nt, work / nt,
                                     works[nt / 2], works.back());
                    }
                }
            } else {
                const int stamp_nr = knobs().top_stamps_nr;
                if (stamp_call > 0 && !chain_on && nr == stamp_nr && ntl == top_launches && ++n_stamp_sweeps == stamp_call) {
                    top_stamps.alloc((size_t)2 * ncount * 8);
                    top_stamps.zero(st);
                    a.top_stamps = (long long*)top_stamps.p;
                    stamp_now = true;
                }
                launch_top_solve(a, L0.begin, ncount, std::min(tgrid, ncount), top_lds, top_flags.p, top_count, ++top_epoch, st, top_tall, nr);
                a.top_stamps = nullptr;
            }
        }
        if (stamp_now) {
            std::vector<long long> h((size_t)2 * ncount * 8);
            HIP_CHECK(hipMemcpyAsync(h.data(), top_stamps.p, h.size() * 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            long long t0 = h[0];
            for (int p = 0; p < ncount; ++p) t0 = std::min(t0, h[(size_t)p * 8]);
            const int pbase = launches[nl - ntl].begin;
            for (int dir = 0; dir < 2; ++dir) {
                double prev_done = 0.0;
                for (size_t qq = 0; qq < ntl; ++qq) {
                    const size_t q = dir == 0 ? nl - ntl + qq : nl - 1 - qq;
                    const Launch& L = launches[q];
                    double d[5] = {0, 0, 0, 0, 0}, first_start = 1e30, last_seen = 0, last_done = 0, crit[5] = {0, 0, 0, 0, 0};
                    for (int t = L.begin; t < L.begin + L.count; ++t) {
                        const long long* e = &h[((size_t)dir * ncount + (size_t)(t - pbase)) * 8];
                        for (int k = 0; k < 5; ++k) d[k] += (e[k + 1] - e[k]) * 0.01;
                        first_start = std::min(first_start, (e[0] - t0) * 0.01);
                        last_seen = std::max(last_seen, (e[2] - t0) * 0.01);
                        if ((e[5] - t0) * 0.01 > last_done) {
                            last_done = (e[5] - t0) * 0.01;
                            for (int k = 0; k < 5; ++k) crit[k] = (e[k + 1] - e[k]) * 0.01;
                        }
                    }
                    std::fprintf(stderr, "[top stamps] %s level %2d: %4d fronts, first starts %7.2f, last sees its flags %7.2f, last publishes %7.2f "
                                 "(hop %5.2f) us | mean: preload %5.2f wait %6.2f gather %5.2f products %5.2f reduce+store %5.2f | "
                                 "last front: %5.2f %6.2f %5.2f %5.2f %5.2f\n", dir == 0 ? "fwd" : "bwd", L.level, L.count, first_start, last_seen,
                                 last_done, last_done - prev_done, d[0] / L.count, d[1] / L.count, d[2] / L.count, d[3] / L.count,
                                 d[4] / L.count, crit[0], crit[1], crit[2], crit[3], crit[4]);
                    prev_done = last_done;
                }
            }
        }
        if (chain_on) {
            chain_reset();
            for (size_t q = nl - ntl; q-- > nper;) chain_add(launches[q], false);
            chain_flush(false);
            if (chain_stamp) {
                const int lo0 = launches[nper].begin;
                const size_t nst = (size_t)(launches[nl - ntl - 1].begin + launches[nl - ntl - 1].count - lo0);
                std::vector<long long> h(2 * nst * 8);
                HIP_CHECK(hipMemcpyAsync(h.data(), top_stamps.p, h.size() * 8, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                for (int dir = 0; dir < 2; ++dir) {
                    long long t00 = 0;
                    for (size_t k = 0; k < nst; ++k) { const long long v = h[((size_t)dir * nst + k) * 8]; if (v && (!t00 || v < t00)) t00 = v; }
                    for (size_t qq = nper; qq + ntl < nl; ++qq) {
                        const size_t q = dir == 0 ? qq : nl - ntl - 1 - (qq - nper);
                        const Launch& L = launches[q];
                        if (L.small) continue;
                        double d[5] = {0, 0, 0, 0, 0}, first_start = 1e30, last_start = 0, last_seen = 0, first_done = 1e30, last_done = 0;
                        for (int t = L.begin; t < L.begin + L.count; ++t) {
                            const long long* e = &h[((size_t)dir * nst + (size_t)(t - lo0)) * 8];
                            for (int k = 0; k < 5; ++k) d[k] += (e[k + 1] - e[k]) * 0.01;
                            first_start = std::min(first_start, (e[0] - t00) * 0.01);
                            last_start = std::max(last_start, (e[0] - t00) * 0.01);
                            last_seen = std::max(last_seen, (e[2] - t00) * 0.01);
                            first_done = std::min(first_done, (e[5] - t00) * 0.01);
                            last_done = std::max(last_done, (e[5] - t00) * 0.01);
                        }
                        std::fprintf(stderr, "[chain stamps] %s launch %2zu level %2d: %4d fronts, starts %6.2f .. %6.2f, last sees its flag %6.2f, done %6.2f .. %6.2f us | "
                                     "mean: preload %5.2f wait %6.2f gather %5.2f products %5.2f reduce+store %5.2f\n", dir == 0 ? "fwd" : "bwd", q, L.level, L.count,
                                     first_start, last_start, last_seen, first_done, last_done, d[0] / L.count, d[1] / L.count, d[2] / L.count, d[3] / L.count, d[4] / L.count);
                    }
                }
            }
        }
        for (size_t q = nper; q-- > 0;) {
            const Launch& L = launches[q];
            if (q > 0 && pair_at(q - 1)) {
                const Launch& Lb = launches[q - 1];
                launch_bwd_level(a, pair_rec(Lb, L), Lb.begin, Lb.count, L.count - L.ntiny, L.ntiny, Lb.solve_bs, Lb.lds_solve, st, nr);
                --q;
            } else if (L.small) {
                launch_bwd_small(a, L.rec, L.begin, L.count - L.ntiny, L.ntiny, st, nr);
            } else {
                launch_bwd(a, L.rec, L.begin, L.count - L.ntall, L.solve_bs, L.lds_solve, st, nr);
                launch_bwd_tall(a, L.begin + L.count - L.ntall, L.ntall, L.fmax, S.N, st);
            }
        }
        HIP_CHECK(hipGetLastError());
    }

    // the side stream may still be forming the top fronts' solve matrices (enqueue_factor)
    bool w_pending = false, w2_pending = false;     // (w2: the last formation rode on the tile stream: ev_join2)
    void wait_w(hipStream_t st)
    {
        if (!w_pending) return;
        HIP_CHECK(hipStreamWaitEvent(st, ev_join, 0));
        if (w2_pending) HIP_CHECK(hipStreamWaitEvent(st, ev_join2, 0));
        w_pending = false;
        w2_pending = false;
    }

    // ---- chained launches (chain_kernels.hip)
    DBuf<int> d_chain;           // [0, nsuper): forward counters (zero between sweeps); [nsuper, 2 nsuper): backward epoch words
    int chain_epoch = 0;
    bool chain_disabled = false;
    size_t chain_from = ~(size_t)0;  // first chained launch (>= launches.size(): none)
    size_t chain_lds = 0;            // LDS of the largest block-class front in the chained launches, per right-hand side
    DBuf<int> d_chain_nchild;    // per supernode: its children in chained launches (the ones that count themselves in)
    std::vector<int64_t> h_toff; // (upload: solve-matrix offsets and chained-children counts, kept for build_records)
    std::vector<int> h_nch;
    DBuf<int64_t> d_recs;        // packed sweep records (kernels.hpp: SolveHdr); empty: the legacy layout
    DBuf<double> tall_ws;        // work space of the tall-front sweep kernels (allocated when the schedule holds such fronts)

    // ---- admission of this handle's multi-stream / waiting operations on its device (DevOp)
    struct TokenRelease {
        LDLEngine* e; bool on;
        ~TokenRelease() { if (on) e->dev_enqueued(false); }
    };
public:
    void set_device(int d)       // (once, right after construction: the engine joins its device's bookkeeping)
    {
        device_id = d;
        if (born) return;
        born = true;
        std::lock_guard<std::mutex> lk(g_dev_mu);
        DevState& D = g_dev[device_id];
        DevRec r;
        r.eng = this;
        D.recs.push_back(r);
        if (D.recs.size() == 2) (void)hipDeviceSynchronize();     // from now on operations carry events (see DevOp)
    }
private:
    bool born = false;
    void engine_gone()
    {
        if (!born) return;
        std::lock_guard<std::mutex> lk(g_dev_mu);
        auto it = g_dev.find(device_id);
        if (it == g_dev.end()) return;
        auto& v = it->second.recs;
        for (size_t k = 0; k < v.size(); ++k) if (v[k].eng == this) { v.erase(v.begin() + (long)k); break; }
    }
    static bool rec_idle(DevRec& r)
    {
        if (r.kind == kOpNone) return true;
        if (r.enqueuing) return false;
        const bool done = (!r.done_main || hipEventQuery(r.done_main) == hipSuccess) &&
                          (!r.done_tiles || hipEventQuery(r.done_tiles) == hipSuccess) &&
                          (!r.done_side || hipEventQuery(r.done_side) == hipSuccess);
        (void)hipGetLastError();
        if (done) r.kind = kOpNone;
        return done;
    }
    bool claim_dev(int kind)
    {
        if (!born) return true;
        std::lock_guard<std::mutex> lk(g_dev_mu);
        DevState& D = g_dev[device_id];
        DevRec* me = nullptr;
        int busy = kOpNone;
        bool others_quiet = true;                // no other handle has called in during the last 20 ms
        const auto now = std::chrono::steady_clock::now();
        for (DevRec& r : D.recs) {
            if (r.eng == this) { me = &r; continue; }
            if (D.recs.size() > 1 && r.stream != stream) {
                if (!rec_idle(r)) busy = std::max(busy, r.kind);
                if (now - r.last_seen < std::chrono::milliseconds(20)) others_quiet = false;
            }
        }
        if (!me) return true;
        me->last_seen = now;
        const bool ok = kind == kOpX ? (busy == kOpNone && others_quiet) : (kind == kOpS ? busy < kOpS : busy < kOpX);
        if (!ok) { ++(kind == kOpS ? n_top_busy : n_ov_busy); return false; }
        if (D.recs.size() > 1) {
            if (!ev_dev_done) HIP_CHECK(hipEventCreateWithFlags(&ev_dev_done, hipEventDisableTiming));
            const bool mine_idle = rec_idle(*me);
            me->kind = std::max(mine_idle ? (int)kOpNone : me->kind, kind);
            me->done_main = ev_dev_done; me->done_tiles = ev_ov_join; me->done_side = ev_join;
        }
        me->stream = stream;
        me->enqueuing = true;
        claimed_kind = kind;
        return true;
    }
    int claimed_kind = kOpNone;
    // the operation is enqueued (recorded = true: mark its end on the main stream) or the call was abandoned
    void dev_enqueued(bool recorded)
    {
        if (!born) return;
        std::lock_guard<std::mutex> lk(g_dev_mu);
        DevState& D = g_dev[device_id];
        for (DevRec& r : D.recs) {
            if (r.eng != this) continue;
            if (recorded && D.recs.size() > 1) {
                if (!ev_dev_done && hipEventCreateWithFlags(&ev_dev_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); r.enqueuing = false; continue; }
                (void)hipEventRecord(ev_dev_done, stream);
                r.kind = std::max(r.kind, claimed_kind);       // (a second handle may have appeared since the admission)
                r.done_main = ev_dev_done; r.done_tiles = ev_ov_join; r.done_side = ev_join;     // (the streams' events may have been created meanwhile)
            }
            r.enqueuing = false;
        }
    }

    // ---- persistent-kernel bookkeeping
    bool top_disabled = false;

public:
    // device word set by the persistent kernel when one of its bounded waits expired (nullptr: no such kernel)
    const int* top_abort_word() const { return (top_flags.p && (top_launches > 0 || chain_from < launches.size())) ? top_flags.p + 2 * top_nflag : nullptr; }
    // The caller has synchronised and found the abort word set: clear it and never use the kernel again.
    int64_t n_ov_fallbacks = 0, n_top_fallbacks = 0;      // lifetime counts (hipkkt_profile, hipkkt_ldl_fallbacks)
    int64_t n_ov_busy = 0;       // factorisations that were not admitted as class X / M (DevOp) and ran on fewer streams
    int64_t n_top_busy = 0;      // sweeps that were not admitted as class S and went level by level
    void top_gave_up()
    {
        top_disabled = true;
        chain_disabled = true;       // (the chained launches share the abort word: whichever wait expired, both go)
        ++n_top_fallbacks;
        launch_zero_ints(top_flags.p + 2 * top_nflag, 1, stream);
        if (d_chain.p) launch_zero_ints(d_chain.p, 2 * S.nsuper, stream);       // (an abandoned sweep leaves counters behind)
        std::fprintf(stderr, "[hipkkt] persistent top-of-tree kernel gave up waiting (GPU shared with another "
                             "resident kernel?); falling back to one launch per level\n");
    }
    // synchronises; for callers without a read-back of their own
    bool top_aborted_sync()
    {
        const int* w = top_abort_word();
        if (!w) return false;
        int v = 0;
        HIP_CHECK(hipMemcpyAsync(&v, w, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return v != 0;
    }

    // synchronises: {#dynamic regularisations, non-finite flag, an overlap-mode wait gave up}
    void read_flags(int out[3])
    {
        HIP_CHECK(hipMemcpyAsync(out, flags.p, 3 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    // The caller found flags[2] set: a bounded wait of the overlap mode expired (never expected; another process on
    // the GPU, or two streams that could not run side by side).  From now on the factorisation runs level by level;
    // the caller repeats it.
    void ov_gave_up()
    {
        ov_disabled = true;
        ++n_ov_fallbacks;
        if (knobs().verbose) {            // which wait expired first (factor_kernels.hip, ov_wait_ge)
            // (the diagnostic words are plain stores behind the abort's CAS: let the tile stream drain first)
            if (ov_stream) (void)hipStreamSynchronize(ov_stream);
            int w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpy(w, flags.p, sizeof(w), hipMemcpyDeviceToHost);
            const int sn = w[4];
            std::fprintf(stderr, "[hipkkt] expired wait: kind %d (1 panel for a child's tiles, 2 tile for its panel, 3 gate), waited for %d "
                         "(saw %d of %d)", w[3], sn, w[5], w[6]);
            if ((w[3] == 1 || w[3] == 2) && sn >= 0 && sn < S.nsuper) {
                int q = -1;
                for (size_t k = 0; k < sched.size(); ++k) if (sched[k] == sn) q = (int)k;
                size_t lq = 0;
                while (lq < launches.size() && !(q >= launches[lq].begin && q < launches[lq].begin + launches[lq].count)) ++lq;
                std::fprintf(stderr, "; supernode %d is in launch %zu of %zu (overlap from %zu, %zu merged runs, the first from %zu), level %d, parent %d", sn, lq,
                             launches.size(), ov_first, ov_groups.size(), ov_groups.empty() ? launches.size() : ov_groups.front().first, lq < launches.size() ? launches[lq].level : -1, S.sn_parent[sn]);
            }
            std::fprintf(stderr, "\n");
            // the lowest overlapped launch with unfinished fronts: progress of its panels and tiles as the abort left them
            std::vector<int> prog((size_t)S.nsuper), done((size_t)S.nsuper), nt((size_t)S.nsuper), started(launches.size());
            (void)hipMemcpy(prog.data(), d_ov_prog.p, prog.size() * sizeof(int), hipMemcpyDeviceToHost);
            (void)hipMemcpy(done.data(), d_ov_done.p, done.size() * sizeof(int), hipMemcpyDeviceToHost);
            (void)hipMemcpy(nt.data(), d_ov_ntiles.p, nt.size() * sizeof(int), hipMemcpyDeviceToHost);
            (void)hipMemcpy(started.data(), d_ov_started.p, started.size() * sizeof(int), hipMemcpyDeviceToHost);
            int shown = 0;
            for (size_t q = ov_first; q < launches.size() && shown < 12; ++q) {
                const Launch& L = launches[q];
                int unfinished = 0;
                for (int t = L.begin; t < L.begin + L.count; ++t) {
                    const int s2 = sched[(size_t)t];
                    const int ncs = S.sn_start[s2 + 1] - S.sn_start[s2];
                    if (prog[(size_t)s2] < ncs || done[(size_t)s2] < nt[(size_t)s2]) {
                        if (shown < 12) {
                            std::fprintf(stderr, "[hipkkt]   launch %zu (started %d) position %d supernode %d: %d of %d columns published, %d of %d tiles done, "
                                         "%d children:", q, started[q], t - L.begin, s2, prog[(size_t)s2], ncs, done[(size_t)s2], nt[(size_t)s2],
                                         S.child_ptr[s2 + 1] - S.child_ptr[s2]);
                            for (int e = S.child_ptr[s2]; e < S.child_ptr[s2 + 1]; ++e) {
                                const int c = S.child_idx[e];
                                if (nt[(size_t)c] > 0) std::fprintf(stderr, " %d(%d/%d)", c, done[(size_t)c], nt[(size_t)c]);
                            }
                            if (ov_group_of[q] >= 0) {
                                const MergeGroup& g = ov_groups[(size_t)ov_group_of[q]];
                                std::fprintf(stderr, " | merged kernel of launches %zu..%zu: %d of %d workgroups started", g.first, g.end - 1, started[g.first], g.count);
                            }
                            std::fprintf(stderr, "\n");
                            ++shown;
                        }
                        ++unfinished;
                    }
                }
                if (unfinished) std::fprintf(stderr, "[hipkkt]   launch %zu: %d of %d fronts unfinished, %d tiles\n", q, unfinished, L.count, L.ntiles);
            }
        }
        std::fprintf(stderr, "[hipkkt] factorisation overlap gave up waiting; falling back to one level after the other\n");
    }
    static bool overlap_wanted()
    {
        return knobs().factor_overlap;
    }
    bool ov_active() const { return overlap_wanted() && !ov_disabled && ov_first < launches.size(); }

    int* flags_ptr() { return flags.p; }

private:
    DBuf<int> d_sn_start, d_rows, d_rel, d_ncolpar, d_child_ptr, d_child_idx, d_ksrc, d_kdst, d_sched, d_perm;
    DBuf<int64_t> d_rowptr, d_front_off, d_upd_off, d_kptr;
    DBuf<signed char> d_psign;
    DBuf<double> fronts, upd, Dinv, xp, uvec, tinv;
    DBuf<double> xp_m, uvec_m;   // work vectors of solve_multi, row-major N x cap and sum(nb) x cap
    DBuf<int> d_iperm;
    size_t multi_cap = 0;
    DBuf<int64_t> d_tinv_off;
    DBuf<double> top_stamps;
    int n_stamp_sweeps = 0;
#ifdef HIPKKT_EXPERIMENTS
    int n_skipw_calls = 0;
#endif
    DBuf<int> d_tinv_list;
    std::vector<int> tinv_list, tinv_small_prefix;
    int tinv_ncmax = 1;
    DBuf<int> flags;
    DBuf<int64_t> stamps;
    DBuf<int64_t> d_tiles;       // int2 {supernode, ti<<16|tj}
    DBuf<int64_t> d_cut_ptr;     // per supernode (as a child): offset into d_cuts
    DBuf<int> d_cuts;            // first child-row index reaching each 64-row tile boundary of the parent's U
    DBuf<int64_t> d_item_ptr;
    DBuf<int64_t> d_items;       // ExtItem = 4 x int64
    DBuf<int64_t> d_dense_off;   // per supernode: offset of its dense child's update block (-1: none)
    DBuf<int64_t> d_wave_cut;
    DBuf<int64_t> d_sitems;      // SubItem = 2 x int64
    DBuf<int64_t> d_tile_cut;
    DBuf<int64_t> d_desc;        // FrontDesc = 8 x int64
    DBuf<int64_t> d_sdesc;       // FrontDesc per row slice of the sliced panels
    std::vector<std::array<int, 3>> slice_list;   // (supernode, slice, slices) in launch order
    std::vector<int64_t> slice_kptr;              // per slice: its K scatter list in ksrc / kdst
    std::vector<int> slice_nk;
    int64_t panel_cap = 0;
    int panel_max_slices = 1;
    DBuf<int> d_spos, d_sn_parent, top_flags;
    DBuf<int> d_tk_pos, d_tk_sl, d_tbase;      // tasks of k_top_solve_sliced (SolveArgs::tk_*); empty unless the set has tall fronts
    DBuf<double> xf;
    std::vector<int> h_tbase;
    int top_ntask = 0, top_nflag = 0, top_sgrid = 0, top_sgrid2 = 0;
    size_t top_slds = 0;
    size_t top_launches = 0, late_launches = 0, top_lds = 0;
    // overlap mode of the factorisation (factor_kernels.hip): the launches from ov_first on (the narrow top of the tree)
    size_t ov_first = 0;         // == launches.size(): none
    bool ov_disabled = false;
    int n_cus = 256, side_winv_blocks = 96;
    // overlap mode: runs of consecutive launches whose panels go out as ONE kernel each (upload: overlap admission)
    struct MergeGroup {
        size_t first, end;           // launches [first, end)
        int count;                   // panel workgroups of the kernel (whole fronts, or row slices: sliced)
        size_t lds;
        bool sliced;
    };
    std::vector<MergeGroup> ov_groups;        // in launch order
    std::vector<int> ov_group_of;             // per launch: index into ov_groups, -1 = a kernel of its own
    bool ov_concurrent = false;          // the main and the tile stream run side by side (choose_side_streams)
    DBuf<int> d_ov_prog, d_ov_done, d_ov_ntiles, d_ov_sprog, d_ov_sbase, d_ov_started;
    hipStream_t ov_stream = nullptr;
    hipEvent_t ev_ov_fork = nullptr, ev_ov_join = nullptr, ev_dev_done = nullptr;
    bool ov_join_pending = false;    // ev_ov_join recorded, not yet waited for (enqueue_factor)
    size_t nr_cap = 1;           // right-hand sides xp / uvec are sized for
    int top_grid_nr[2] = {-1, -1};   // the persistent kernel's grid for 2 / 4 right-hand sides (asked on first use)
    int top_count = 0, late_count = 0, top_grid = 0, top_epoch = 0;
    bool top_tall = true;        // the persistent kernel's 1024-thread build (default) or its 512-thread one
    std::vector<int64_t> tile_base;   // per supernode: index of its first tile in `tiles` (-1: none)
    DBuf<int> d_gl_src, d_udst;
    DBuf<int64_t> d_glm_ptr, d_hp_lidx, d_pr_ptr;     // many-column sweeps: gather lists without the pulled leaves, and the pulled terms
    DBuf<int> d_udst_m, d_hp_col, d_hp_row, d_pr_slot;
    int n_pull_rows = 0;
    std::vector<Launch> launches;
    std::vector<int> sched;
    std::vector<int64_t> tiles;

    TreeDev tree() const
    {
        TreeDev t;
        t.nsuper = S.nsuper;
        t.sn_start = d_sn_start.p; t.rowptr = d_rowptr.p; t.rows = d_rows.p; t.rel = d_rel.p;
        t.ncolpar = d_ncolpar.p; t.front_off = d_front_off.p; t.upd_off = d_upd_off.p;
        t.child_ptr = d_child_ptr.p; t.child_idx = d_child_idx.p; t.kptr = d_kptr.p;
        t.ksrc = d_ksrc.p; t.kdst = d_kdst.p; t.sched = d_sched.p; t.psign = d_psign.p; t.perm = d_perm.p;
        t.dense_off = d_dense_off.p; t.item_ptr = d_item_ptr.p; t.items = (const ExtItem*)d_items.p; t.gl_ptr = d_item_ptr.p; t.gl_src = d_gl_src.p; t.udst = d_udst.p;
        t.glm_ptr = d_glm_ptr.p; t.udst_m = d_udst_m.p; t.hp_col = d_hp_col.p; t.hp_row = d_hp_row.p; t.hp_lidx = d_hp_lidx.p; t.pr_ptr = d_pr_ptr.p; t.pr_slot = d_pr_slot.p;
        t.cut_ptr = d_cut_ptr.p; t.cuts = d_cuts.p; t.wave_cut = d_wave_cut.p; t.tinv_off = d_tinv_off.p;
        t.sitems = (const SubItem*)d_sitems.p; t.tile_cut = d_tile_cut.p; t.desc = (const FrontDesc*)d_desc.p; t.sdesc = (const FrontDesc*)d_sdesc.p;
        t.spos = d_spos.p; t.sn_parent = d_sn_parent.p;
        return t;
    }

    int front_size(int s) const
    {
        return (S.sn_start[s + 1] - S.sn_start[s]) + (int)(S.rowptr[s + 1] - S.rowptr[s]);
    }

    // too tall for the LDS of the block sweep kernels / of k_top_solve (solve_kernels.hip, k_fwd_tall)
    // (HIPKKT_SOLVE_TALL_ROWS=n: fronts of n rows or more count as tall as well -- the tests' way to those paths)
    bool front_is_tall(int s) const
    {
        const int tall_rows = knobs().solve_tall_rows;
        const int f = front_size(s), nc = S.sn_start[s + 1] - S.sn_start[s];
        return solve_lds_bytes(f, nc) > kLdsCap || (tall_rows > 0 && f >= tall_rows);
    }

    void build_schedule()
    {
        sched.clear();
        launches.clear();
        tiles.clear();
        tinv_list.clear();
        slice_list.clear();
        tile_base.assign(S.nsuper, -1);
        auto ncols = [&](int s) { return S.sn_start[s + 1] - S.sn_start[s]; };
        auto is_small = [&](int s) {
            int f = front_size(s), nc = ncols(s), nb = f - nc;
            return f <= kSmallFrontMax && f * nc + nb * nb <= kSmallSliceMax;
        };
        int sched_cus = 256;
        {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) sched_cus = prop.multiProcessorCount;
            else (void)hipGetLastError();
        }
        int level_no = -1;
        for (const Level& lv : S.levels) {
            ++level_no;
            std::vector<int> small, big;
            for (int t = lv.begin; t < lv.end; ++t) {
                int s = S.level_sn[t];
                (is_small(s) ? small : big).push_back(s);
            }
            // a handful of one-wave fronts beside a block-class launch is not worth launches of its own (one in the
            // factorisation, two per solve, each ~5-45 us of pure latency): they ride with the block-class fronts
            const int merge_small = knobs().merge_small;
            if (!big.empty() && (int)small.size() <= merge_small) {
                big.insert(big.end(), small.begin(), small.end());
                small.clear();
            }
            const int slice_rows = knobs().slice_rows;
            const bool slice_fit = knobs().slice_fit;
            int level_slice_rows = slice_rows;
            auto slices_of = [&](int s) {        // row slices the panel kernel needs for this front (1: fits one CU)
                if (panel_cap <= 0) return 1;
                const int nc = ncols(s), nb = front_size(s) - nc;
                int r = panel_slices_needed(nc, nb, panel_cap, std::min(panel_max_slices, std::max(1, nb)));
                if (r == 0) throw std::runtime_error("panel does not fit LDS even in row slices (panel_cap too large?)");
                // More, shorter slices than LDS needs: a slice's block step is bound by its WORKER waves when it holds many
                // rows (250 rows x 90 columns: 75 trailing tiles per block on 12 waves = 4-5 us against the diagonal
                // chain's 3.7), and its assembly by what one CU can load; the price is one more redundant copy of the
                // diagonal block's factorisation per slice.  128 rows (0 = as few slices as LDS allows): cfg5's
                // factorisation 5.30 -> 5.06 ms, cfg3's 1.21 -> 1.23 (slices have K and item lists of their own, so the
                // per-slice overhead no longer grows with their number).
                // (the preference stops at 16 slices: beyond that only what LDS needs -- a 6289-row front in 49 slices of 128
                //  rows measured 44.9 ms per factorisation against 40.3 in 16)
                if (r > 1 && level_slice_rows > 0) r = std::max(r, std::min(std::min(panel_max_slices, 16), (nb + level_slice_rows - 1) / level_slice_rows));
                return r;
            };
            // One round of panel workgroups where possible: every slice needs a CU to itself, so a level with more slices
            // than CUs runs its panel kernel in two rounds (cfg3's second level: 99 fronts x 3 slices of 114 rows = 297
            // workgroups, 112 us; x 2 slices of 171 rows = 198 workgroups, one round).  Such a level takes the shortest
            // slices (>= the default 128 rows) that bring it down to the CU count, if LDS allows any.
            level_slice_rows = slice_rows;
            if (slice_fit && slice_rows > 0) {
                auto total = [&]() {
                    int t = 0;
                    bool any = false;
                    for (int s : big) { const int r = slices_of(s); t += r; any = any || r > 1; }
                    return any ? t : 0;
                };
                if (total() > sched_cus) {
                    static const int cand[] = {144, 160, 176, 192, 224, 256, 320, 384, 512, 1 << 20};
                    for (int c : cand) {
                        level_slice_rows = c;
                        if (total() <= sched_cus) break;
                    }
                    if (total() > sched_cus) level_slice_rows = slice_rows;
                }
            }
            auto work = [&](int s) { return (double)front_size(s) * front_size(s) * ncols(s); };
            auto by_work = [&](int a, int b) { double wa = work(a), wb = work(b); return wa != wb ? wa > wb : a < b; };
            std::sort(small.begin(), small.end(), by_work);
            std::sort(big.begin(), big.end(), by_work);
            // fronts factorised in row slices go to the end of the block-class launch (own panel kernel launch)
            std::stable_partition(big.begin(), big.end(), [&](int s) { return slices_of(s) == 1; });
            auto is_tall = [&](int s) { return front_is_tall(s); };
            std::stable_partition(big.begin(), big.end(), [&](int s) { return !is_tall(s); });
            // the tiny fronts (f <= 8) go to the end of the one-wave launch: the solves give them their own kernel
            std::stable_partition(small.begin(), small.end(), [&](int s) { return front_size(s) > 8; });
            const int ntiny_level = (int)std::count_if(small.begin(), small.end(), [&](int s) { return front_size(s) <= 8; });
            for (int cls = 0; cls < 2; ++cls) {
                const std::vector<int>& v = cls == 0 ? big : small;
                if (v.empty()) continue;
                Launch L{};
                L.begin = (int)sched.size();
                L.count = (int)v.size();
                L.small = cls == 1;
                L.level = level_no;
                L.ntiny = cls == 1 ? ntiny_level : 0;
                int fmax = 0, slice = 0;
                for (int s : v) {
                    int f = front_size(s), nc = ncols(s), nb = f - nc;
                    fmax = std::max(fmax, f);
                    if (!(cls == 1 && f <= 8)) slice = std::max(slice, f * nc + nb * nb);     // (tiny fronts: own kernel)
                }
                L.slice = (slice + 1) & ~1;
                int pmax = 0;                       // LDS doubles of the largest panel: a trapezoid (panel kernel)
                int64_t smax = 0;
                L.nsliced = 0;
                L.slice_begin = (int)slice_list.size();
                // A launch that holds row-sliced fronts runs ALL its fronts through the sliced panel kernel, a whole front
                // as a front of one slice: two panel kernels one after the other (whole, then sliced) cost a level of
                // cfg5 30-57 us for the one to four whole fronts that sit beside its hundreds of slices.
                bool all_sliced = false;
                if (cls == 0) for (int s : v) all_sliced = all_sliced || slices_of(s) > 1;
                for (int s : v) {
                    const int r = cls == 0 ? slices_of(s) : 1;
                    const int nc = ncols(s), nb = front_size(s) - nc;
                    if (r == 1 && !all_sliced) {
                        pmax = std::max(pmax, front_size(s) * nc - nc * (nc - 1) / 2);
                    } else {
                        ++L.nsliced;
                        smax = std::max(smax, panel_slice_doubles(nc, nb, r));
                        for (int q = 0; q < r; ++q) slice_list.push_back({s, q, r});
                    }
                }
                L.slice_count = (int)slice_list.size() - L.slice_begin;
                L.lds_sliced = L.nsliced ? panel_lds_bytes(0, (int)smax) : 0;
                L.nbk = kMaxNbk;
                // (swept on cfg2 after the tree got shorter and wider: 128 / 192 beat the earlier 96 / 128 by 2 %)
                L.bs_panel = fmax > 192 ? 1024 : (fmax > 128 ? 512 : 256);
                L.lds_panel = panel_lds_bytes(fmax, pmax);
                if (!L.small && L.lds_panel > kLdsCap)
                    throw std::runtime_error("panel does not fit LDS (panel_cap too large?)");
                if (L.lds_sliced > kLdsCap) throw std::runtime_error("panel slice does not fit LDS (panel_cap too large?)");
                int ncmax = 0;
                for (int s : v) ncmax = std::max(ncmax, ncols(s));
                // (per front, then the maximum: the tallest front of a launch is a narrow panel and its widest a short one --
                //  sized from (fmax, ncmax) jointly, a level with a 7000-row panel beside a 96-column one asked for LDS
                //  nobody needs and the structure was refused as "too large")
                L.lds_solve = 0;
                L.ntall = 0;
                if (!L.small) for (int s : v) {
                    if (is_tall(s)) { ++L.ntall; continue; }
                    L.lds_solve = std::max(L.lds_solve, solve_lds_bytes(front_size(s), ncols(s)));
                }
                L.fmax = fmax;
                L.ncmax = ncmax;
                const int small_bs_count = knobs().bs128_count;
                const int small_bs_f = knobs().bs128_f;
                L.solve_bs = (L.count >= small_bs_count && fmax <= small_bs_f) ? 128 : 256;
                L.tinv_begin = (int)tinv_list.size();
                L.tinv_ncmax = 1;
                if (!L.small) for (int s : v) {
                    tinv_list.push_back(s);
                    L.tinv_ncmax = std::max(L.tinv_ncmax, ncols(s));
                    tinv_ncmax = std::max(tinv_ncmax, ncols(s));
                }
                L.tinv_count = (int)tinv_list.size() - L.tinv_begin;
                if (L.lds_solve > kLdsCap) throw std::runtime_error("front too large for the solve kernels");
                L.tile_begin = (int)tiles.size();
                int64_t tile_cols = 0;
                if (!L.small) {
                    for (int s : v) {
                        int nb = front_size(s) - ncols(s);
                        int nt = (nb + 63) / 64;
                        tile_base[s] = (int64_t)tiles.size();
                        tile_cols += (int64_t)ncols(s) * (nt * (nt + 1) / 2);
                        for (int ti = 0; ti < nt; ++ti)
                            for (int tj = 0; tj <= ti; ++tj) {
                                // int2 {x = s, y = ti<<16 | tj}, little endian in one int64
                                uint64_t lo = (uint32_t)s, hi = (uint32_t)((ti << 16) | tj);
                                tiles.push_back((int64_t)(lo | (hi << 32)));
                            }
                    }
                }
                L.ntiles = (int)tiles.size() - L.tile_begin;
                L.tile_nc = L.ntiles > 0 ? (int)(tile_cols / L.ntiles) : 0;
                launches.push_back(L);
                sched.insert(sched.end(), v.begin(), v.end());
            }
        }
    }

    // Packed sweep records (kernels.hpp: SolveHdr / RecSeg): per launch and size class one record size, so that a
    // kernel finds a front's record from its place in the launch and fetches header and row slots in one round of loads.
    // HIPKKT_PACKED=0, or more than HIPKKT_PACKED_MAX_MB (4096) of records -- a wide level with one very tall front pays
    // that front's height for every front --: the legacy layout.
    void build_records(const std::vector<int64_t>& glptr, const std::vector<int>& gsrc)
    {
        const bool packed_on = knobs().packed;
        const int64_t max_mb = knobs().packed_max_mb;
        for (Launch& L : launches) L.rec = RecSeg{};
        if (!packed_on) return;
        static_assert(sizeof(SolveHdr) == 64, "SolveHdr layout");
        int64_t total = 0;
        struct Cls { int first, count, cls; };
        auto classes = [&](const Launch& L) {
            std::vector<Cls> v;
            if (L.small) {
                if (L.count - L.ntiny > 0) v.push_back({L.begin, L.count - L.ntiny, 1});
                if (L.ntiny > 0) v.push_back({L.begin + L.count - L.ntiny, L.ntiny, 2});
            } else if (L.count > 0) {
                v.push_back({L.begin, L.count, 0});
            }
            return v;
        };
        // no gather slots for the one-wave launch of tree level 0 (its fronts have no children, and its kernels are told
        // so: `leaf`) -- unless a block-class launch of that level sits in front of it: the two may then go out as one
        // level kernel, whose one-wave bodies read the slots
        auto no_slots = [&](size_t q) {
            const Launch& L = launches[q];
            return L.small && L.level == 0 && !(q > 0 && !launches[q - 1].small && launches[q - 1].level == 0);
        };
        for (size_t q = 0; q < launches.size(); ++q) {
            Launch& L = launches[q];
            const bool leaf = no_slots(q);
            for (const Cls& c : classes(L)) {
                const int fmax = c.cls == 0 ? ((L.fmax + 3) & ~3) : (c.cls == 1 ? 64 : 8);
                const int64_t stride = ((int64_t)sizeof(SolveHdr) + 4 * (int64_t)fmax + (leaf ? 0 : 32 * (int64_t)fmax) + 63) & ~(int64_t)63;
                if (stride > (1 << 30)) return;
                L.rec.off[c.cls] = total;
                L.rec.stride[c.cls] = (int)stride;
                L.rec.fmax[c.cls] = fmax;
                total += stride * c.count;
            }
        }
        if (total > max_mb * (1 << 20)) {
            for (Launch& L : launches) L.rec = RecSeg{};
            if (knobs().verbose) std::fprintf(stderr, "[hipkkt] packed sweep records would take %.0f MB: legacy layout\n", total / 1048576.0);
            return;
        }
        std::vector<int64_t> store((size_t)(total / 8) + 8, 0);
        char* base = reinterpret_cast<char*>(store.data());
        for (size_t q = 0; q < launches.size(); ++q) {
            const Launch& L = launches[q];
            const bool leaf = no_slots(q);
            for (const Cls& c : classes(L)) {
                const int fmax = L.rec.fmax[c.cls];
                for (int k = 0; k < c.count; ++k) {
                    const int sn = sched[(size_t)(c.first + k)];
                    char* rec = base + L.rec.off[c.cls] + (int64_t)k * L.rec.stride[c.cls];
                    const int nc = S.sn_start[sn + 1] - S.sn_start[sn], nb = (int)(S.rowptr[sn + 1] - S.rowptr[sn]), f = nc + nb;
                    if (f > fmax) throw std::runtime_error("packed record: front larger than its class");
                    SolveHdr h{};
                    h.mat_off = c.cls == 0 ? h_toff[(size_t)sn] : S.front_off[sn];
                    h.rp = S.rowptr[sn];
                    h.s = sn; h.c0 = S.sn_start[sn]; h.nc = nc; h.nb = nb;
                    h.par = S.sn_parent[sn];
                    h.nchild = h_nch.empty() ? 0 : h_nch[(size_t)sn];
                    std::memcpy(rec, &h, sizeof(h));
                    int* idx = reinterpret_cast<int*>(rec + sizeof(SolveHdr));
                    for (int i = 0; i < nc; ++i) idx[i] = S.perm[(size_t)(h.c0 + i)];
                    for (int i = nc; i < f; ++i) idx[i] = S.rows[(size_t)(h.rp + i - nc)];
                    if (leaf) continue;
                    int* slot = idx + fmax;
                    const int64_t lc0 = (int64_t)h.c0 + h.rp;
                    for (int i = 0; i < fmax; ++i) {
                        int* g = slot + 8 * (int64_t)i;
                        g[0] = 0;
                        for (int q = 1; q <= 6; ++q) g[q] = -1;
                        g[7] = 0;
                        if (i >= f) continue;
                        const int64_t g0 = glptr[(size_t)(lc0 + i)], g1 = glptr[(size_t)(lc0 + i + 1)];
                        g[0] = (int)(g1 - g0);
                        for (int q = 0; q < 6 && g0 + q < g1; ++q) g[1 + q] = gsrc[(size_t)(g0 + q)];
                        g[7] = (int)(g0 + 6);
                    }
                }
            }
        }
        store.resize((size_t)(total / 8) + 8);
        d_recs.upload(store);
        if (knobs().verbose) std::fprintf(stderr, "[hipkkt] packed sweep records: %.1f MB\n", total / 1048576.0);
    }

    void upload(const std::vector<int>& dsigns)
    {
        // leaves with one column and a short row list, pulled by their parents in the many-column sweeps (see the gather
        // lists below; HIPKKT_PULL_LEAVES=0: none)
        const bool pull_on = knobs().pull_leaves;
        std::vector<char> pulled((size_t)S.nsuper, 0);
        {
            std::vector<char> in_small((size_t)S.nsuper, 0);       // (one-wave launches only: the block kernels do not know the flag)
            for (const Launch& L : launches)
                if (L.small) for (int t = L.begin; t < L.begin + L.count; ++t) in_small[(size_t)sched[(size_t)t]] = 1;
            for (int s = 0; pull_on && s < S.nsuper; ++s) {
                const int nb = (int)(S.rowptr[s + 1] - S.rowptr[s]);
                pulled[(size_t)s] = in_small[(size_t)s] && S.child_ptr[s + 1] == S.child_ptr[s] && S.sn_start[s + 1] - S.sn_start[s] == 1 &&
                                    S.sn_parent[s] >= 0 && nb >= 1 && nb <= 47;
            }
        }
        d_sn_start.upload(S.sn_start);
        d_rowptr.upload(S.rowptr);
        d_rows.upload(S.rows);
        d_rel.upload(S.rel);
        std::vector<int> ncolpar(S.nsuper, 0);
        for (int s = 0; s < S.nsuper; ++s) {
            int p = S.sn_parent[s];
            if (p < 0) continue;
            int pnc = S.sn_start[p + 1] - S.sn_start[p], k = 0;
            for (int64_t q = S.rowptr[s]; q < S.rowptr[s + 1] && S.rel[q] < pnc; ++q) ++k;
            ncolpar[s] = k;
        }
        d_ncolpar.upload(ncolpar);
        d_front_off.upload(S.front_off);
        d_upd_off.upload(S.upd_off);
        d_child_ptr.upload(S.child_ptr);
        d_child_idx.upload(S.child_idx);
        d_kptr.upload(S.kptr);
        std::vector<int> ks_all(S.ksrc);
        slice_kptr.assign(slice_list.size(), 0);
        slice_nk.assign(slice_list.size(), 0);
        {
            // the panel kernel keeps a block-class front as a trapezoid in LDS (factor_kernels.hip: pcol): its K
            // entries get their packed position; one-wave fronts keep lrow + lcol*f
            std::vector<int> kd(S.kdst);
            for (const Launch& L : launches) {
                if (L.small) continue;
                for (int q = L.begin; q < L.begin + L.count - L.nsliced; ++q) {      // (sliced fronts keep lrow + lcol*f)
                    const int s = sched[q];
                    const int f = front_size(s);
                    for (int64_t e = S.kptr[s]; e < S.kptr[s + 1]; ++e) {
                        const int lcol = kd[e] / f, lrow = kd[e] - lcol * f;
                        kd[e] = lrow + (int)(((int64_t)lcol * (2 * f - 1 - lcol)) >> 1);
                    }
                }
            }
            // A row slice of a sliced front gets a K list of its own, with the positions in ITS LDS image (top block +
            // its rows, as a trapezoid): the panel kernel neither scans the other slices' entries nor divides per entry.
            for (size_t q = 0; q < slice_list.size(); ++q) {
                const int s = slice_list[q][0], sl = slice_list[q][1], nsl = slice_list[q][2];
                const int ff = front_size(s), nc = S.sn_start[s + 1] - S.sn_start[s], nb = ff - nc;
                const int rsmax = (nb + nsl - 1) / nsl, r_lo = nc + sl * rsmax;
                const int rs = std::max(0, std::min(rsmax, ff - r_lo)), f = nc + rs;
                slice_kptr[q] = (int64_t)ks_all.size();
                for (int64_t e = S.kptr[s]; e < S.kptr[s + 1]; ++e) {
                    const int lcol = S.kdst[e] / ff;
                    int r = S.kdst[e] - lcol * ff;
                    if (r >= nc) { if (r < r_lo || r >= r_lo + rs) continue; r = nc + (r - r_lo); }
                    ks_all.push_back(S.ksrc[e]);
                    kd.push_back(r + (int)(((int64_t)lcol * (2 * f - 1 - lcol)) >> 1));
                }
                slice_nk[q] = (int)((int64_t)ks_all.size() - slice_kptr[q]);
            }
            if (ks_all.size() >= ((size_t)1 << 31)) throw std::runtime_error("K scatter lists exceed int32 indexing");
            d_kdst.upload(kd);
            d_ksrc.upload(ks_all);
        }
        d_sched.upload(sched);
        // (d_tiles is uploaded with the tile work lists below: the launch order of a level's tiles may be permuted there)
        {
            // Overlap admission.  In overlap mode a level's PANEL workgroups (each needs a CU to itself: ~150 KB of LDS) run
            // beside the level's TILE workgroups on the overlap stream (53 KB: they fit beside each other, but one of them
            // on a CU is enough to keep a panel out) and the side stream's W formation.  Tiles wait for their panel's
            // blocks and panels wait for their children's tiles, so forward progress needs every panel workgroup of a
            // launch to be RESIDENT before a tile of that launch may wait for it.  That is enforced, not hoped for: the
            // launch's tile kernel sits behind a gate (k_ov_gate) that opens when all its panel workgroups have started.
            // Everything else on the device is work that ends by itself (the previous launch's tiles, whose panels are
            // resident or done; the W formation, which waits for nothing), so the panel workgroups do get their CUs,
            // provided there are enough CUs for all of them plus the gate's wave at once:
            //     panel workgroups of the launch + 1 + margin <= CUs.
            // Below that bound the width is a matter of speed only: wide launches were measured slower in the mode
            // (panels that wait hold whole CUs the tiles could use), so the default admits launches of up to 120 panel
            // workgroups; HIPKKT_OV_MAX_FRONTS moves that, never beyond the bound.  (The bounded waits remain for what
            // this argument cannot see: another process, or another handle's kernels, on the same device.)
            {
                int dev = 0;
                hipDeviceProp_t prop;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cus = prop.multiProcessorCount;
                side_winv_blocks = knobs().winv_blocks > 0 ? knobs().winv_blocks : std::max(8, n_cus * 3 / 8);
            }
            constexpr int kOvMargin = 8;
            // (the environment override is per process, the default per handle: n_cus is this handle's device's)
            const int ov_max_env = knobs().ov_max_fronts;
            const int ov_max = std::min(ov_max_env > 0 ? ov_max_env : 120 * n_cus / 256, n_cus - 1 - kOvMargin);
            auto panel_wgs = [&](const Launch& L) { return L.count - L.nsliced + L.slice_count; };   // whole panels + row slices
            size_t first = launches.size();
            while (first > 0) {
                const Launch& L = launches[first - 1];
                if (L.small || panel_wgs(L) > ov_max) break;
                --first;
            }
            // Launches with thousands of tiles stay out of the mode, and with them the whole handle (the overlapped launches
            // are the schedule's tail): there the tiles ARE the level -- nothing to hide them behind -- and a tile of the mode
            // is the slower one (54 KB of LDS instead of 33: two workgroups per CU instead of four; operands and results
            // past the L2's write-back path).  Measured with the mode on / off, by the largest launch of the region:
            // 1080 tiles 6.25 / 6.45 ms, 1279 tiles 3.17 / 3.65, 1145 tiles 3.74 / 3.72 | 2310 tiles 18.6 / 17.6,
            // 2428 tiles 9.5 / 8.2, 4253 tiles 10.0 / 7.8, 13 041 tiles 54 / 27, cfg2 with 1 % long-range couplings
            // (24 000 tiles) 143 / 61 ms.  (cfg2's overlapped launches have at most 418 tiles, cfg5's 630.)
            const int ov_max_tiles = knobs().ov_max_tiles;
            bool heavy_tiles = false;
            for (size_t q = first; q < launches.size(); ++q) heavy_tiles = heavy_tiles || launches[q].ntiles > ov_max_tiles;
            ov_first = (!heavy_tiles && launches.size() - first >= 3) ? first : launches.size();
            {
                // Runs of narrow launches whose panels share one kernel, found from the root downwards: all launches of a run
                // are of one kind (whole panels, or row slices -- a launch with sliced fronts runs all its fronts as slices),
                // a run has at most ov_merge_max workgroups in all (HIPKKT_OV_MERGE, 0 = off) and at least two launches, a
                // launch in a run has at most ov_merge_wide workgroups (HIPKKT_OV_MERGE_WIDE: the workgroups of a run hold
                // their CUs from the start of the run, which the tiles of a WIDE level below them would miss), and the first
                // overlapped launch is in none (its tiles are released by an event).
                const int ov_merge_max = knobs().ov_merge;
                const int ov_merge_wide = knobs().ov_merge_wide;
                const int ov_merge_groups = knobs().ov_merge_groups;
                ov_groups.clear();
                ov_group_of.assign(launches.size(), -1);
                size_t m = launches.size();
                const int cap = std::min(ov_merge_max, ov_max);
                while (m > ov_first + 1 && (int)ov_groups.size() < ov_merge_groups) {
                    MergeGroup g{m, m, 0, 0, launches[m - 1].nsliced > 0};
                    // (the root's run takes launches of any width, as it always did; the runs below it narrow ones only)
                    const int wide = ov_groups.empty() ? cap : ov_merge_wide;
                    while (g.first > ov_first + 1 && !launches[g.first - 1].small &&
                           (g.sliced ? launches[g.first - 1].nsliced == launches[g.first - 1].count : launches[g.first - 1].nsliced == 0) &&
                           panel_wgs(launches[g.first - 1]) <= wide && g.count + panel_wgs(launches[g.first - 1]) <= cap) {
                        const Launch& L = launches[g.first - 1];
                        g.count += panel_wgs(L);
                        g.lds = std::max(g.lds, g.sliced ? L.lds_sliced : L.lds_panel);
                        --g.first;
                    }
                    if (g.end - g.first < 2) break;
                    ov_groups.insert(ov_groups.begin(), g);
                    m = g.first;
                }
                for (size_t k = 0; k < ov_groups.size(); ++k)
                    for (size_t q = ov_groups[k].first; q < ov_groups[k].end; ++q) ov_group_of[q] = (int)k;
            }
            d_ov_started.alloc(std::max<size_t>(launches.size(), 1));
            HIP_CHECK(hipMemset(d_ov_started.p, 0, std::max<size_t>(launches.size(), 1) * sizeof(int)));
            std::vector<int> nt((size_t)S.nsuper, 0);
            for (size_t q = ov_first; q < launches.size(); ++q) {
                const Launch& L = launches[q];
                for (int t = L.begin; t < L.begin + L.count; ++t) {
                    const int sn = sched[(size_t)t];
                    const int nb = front_size(sn) - (S.sn_start[sn + 1] - S.sn_start[sn]);
                    const int k = (nb + 63) / 64;
                    nt[(size_t)sn] = k * (k + 1) / 2;
                }
            }
            d_ov_ntiles.upload(nt);
            // fronts factorised in row slices publish their progress per slice
            std::vector<int> sbase((size_t)std::max(S.nsuper, 1), -1);
            for (size_t q = slice_list.size(); q-- > 0;) sbase[(size_t)slice_list[q][0]] = (int)q;      // (slices of a front are consecutive)
            d_ov_sbase.upload(sbase);
            d_ov_sprog.alloc(std::max<size_t>(slice_list.size(), 1));
            HIP_CHECK(hipMemset(d_ov_sprog.p, 0, std::max<size_t>(slice_list.size(), 1) * sizeof(int)));
            d_ov_prog.alloc((size_t)S.nsuper);
            d_ov_done.alloc((size_t)S.nsuper);
            HIP_CHECK(hipMemset(d_ov_prog.p, 0, (size_t)std::max(S.nsuper, 1) * sizeof(int)));
            HIP_CHECK(hipMemset(d_ov_done.p, 0, (size_t)std::max(S.nsuper, 1) * sizeof(int)));
        }
        {
            std::vector<int64_t> toff(S.nsuper + 1, 0);
            std::vector<char> in_list(S.nsuper, 0);
            for (int s : tinv_list) in_list[s] = 1;
            // (level by level like the panel and update stores: symbolic.cpp, step 10)
            {
                int64_t wo = 0;
                for (int t = 0; t < S.nsuper; ++t) {
                    const int s = S.level_sn[t];
                    int64_t nc = S.sn_start[s + 1] - S.sn_start[s];
                    int64_t fs = nc + (S.rowptr[s + 1] - S.rowptr[s]);
                    toff[s] = wo;
                    wo += in_list[s] ? 2 * fs * nc : 0;
                }
                toff[S.nsuper] = wo;
            }
            d_tinv_off.upload(toff);
            h_toff = toff;
            static_assert(sizeof(FrontDesc) == 64, "FrontDesc layout");
            std::vector<FrontDesc> desc(sched.size());
            for (size_t q = 0; q < sched.size(); ++q) {
                const int s = sched[q];
                FrontDesc d;
                d.front_off = S.front_off[s]; d.upd_off = S.upd_off[s]; d.w_off = toff[s]; d.rp = S.rowptr[s];
                d.kptr = S.kptr[s]; d.s = s; d.c0 = S.sn_start[s]; d.nc = S.sn_start[s + 1] - S.sn_start[s];
                d.nb = (int)(S.rowptr[s + 1] - S.rowptr[s]); d.nk = (int)(S.kptr[s + 1] - S.kptr[s]);
                d.pad = pulled[(size_t)s] ? 1 : 0;          // (launch records: bit 0 = a pulled leaf of the many-column sweeps, see below)
                desc[q] = d;
            }
            std::vector<int64_t> rawd(desc.size() * 8);
            std::memcpy(rawd.data(), desc.data(), desc.size() * sizeof(FrontDesc));
            d_desc.upload(rawd);
            std::vector<int> pos_of(S.nsuper, -1);
            for (size_t q = 0; q < sched.size(); ++q) pos_of[sched[q]] = (int)q;
            std::vector<int64_t> raws(std::max<size_t>(slice_list.size(), 1) * 8, 0);
            for (size_t q = 0; q < slice_list.size(); ++q) {
                FrontDesc d = desc[(size_t)pos_of[slice_list[q][0]]];
                d.pad = (slice_list[q][1] << 16) | slice_list[q][2];
                d.kptr = slice_kptr[q]; d.nk = slice_nk[q];             // (its own K list, positions in the slice's image)
                std::memcpy(raws.data() + q * 8, &d, sizeof(FrontDesc));
            }
            d_sdesc.upload(raws);
            std::vector<int> spos(S.nsuper, -1);
            for (size_t q = 0; q < sched.size(); ++q) spos[sched[q]] = (int)q;
            d_spos.upload(spos);
            d_sn_parent.upload(S.sn_parent);
            // late_launches: the longest suffix of block-class launches with <= kTopMaxFronts fronts in total -- the
            // narrow top of the tree, whose W is formed behind the factorisation (enqueue_factor / wait_w)
            late_launches = 0;
            {
                int cnt = 0;
                for (size_t q = launches.size(); q-- > 0;) {
                    const Launch& L = launches[q];
                    if (L.small || cnt + L.count > kTopMaxFronts) break;
                    cnt += L.count;
                    ++late_launches;
                }
                late_count = cnt;
                if (late_launches < 3) { late_launches = 0; late_count = 0; }
            }
            // persistent top: the longest suffix of block-class launches none of which holds more than 1.5 x as many
            // fronts as the device keeps resident workgroups of the persistent kernel (a workgroup then has at most two
            // fronts per level; k_top_solve walks its fronts in level order)
            top_launches = 0; top_count = 0; top_lds = 0; top_grid = 0;
            {
                size_t lds = 0;
                for (size_t q = launches.size(); q-- > 0 && !launches[q].small;) lds = std::max(lds, launches[q].lds_solve);
                const int tall_env = knobs().top_tall;
                top_tall = tall_env != 0;           // the 1024-thread build unless HIPKKT_TOP_TALL=0 (solve_kernels.hip)
                const int cap_env = knobs().top_cap;
                const int cap = std::min(std::min(kTopMaxFronts, cap_env), top_solve_capacity(lds, top_tall));
                for (size_t q = launches.size(); q-- > 0;) {
                    const Launch& L = launches[q];
                    // (measured on cfg2 with 240 workgroups: x1 0.313, x1.25-1.7 0.307, x2.5 0.319, x6 0.346 ms per solve)
                    const double mult = knobs().top_mult;
                    if (L.small || L.count > mult * cap) break;
                    top_count += L.count;
                    top_lds = std::max(top_lds, L.lds_solve);
                    ++top_launches;
                }
                top_grid = std::min(cap, top_count);
            }
            if (top_launches < 3) { top_launches = 0; top_count = 0; top_grid = 0; }     // not worth a special kernel
            top_ntask = 0;
            top_nflag = std::max(top_count, 1);
            {
                // Sets with very tall fronts (solve matrix > 3 x slice_kb: far more than a CU should stream per hop) run the
                // (front, slice) kernel: such a front is cut into slices of ~slice_kb (at most slice_max), the others are one task
                // (r03, cfg5: slices of ~80 KB, at most 16, instead of ~120 KB / 8: sweep pair 0.765 -> 0.729 ms; 60 KB / 16 and
                //  40 KB / 32: 0.74 -- a hop is mostly its fixed latencies by then.  A front is sliced when its W exceeds
                //  HIPKKT_SOLVE_SLICE_FROM KB, by default 4.5 slices' worth: cfg3's 395 KB fronts are faster whole)
                // (r04: at most 64 slices instead of 16 -- cfg5's 1.2 MB fronts take 15 either way, the long-range cfg2
                //  variant's 14 154-row panels (11 MB each, 148 of them in a chain) were streamed in 680 KB pieces: sweep pair
                //  4.98 -> 3.85 ms, unit 89.5 -> 81.7 ms; 96 and 128 slices: the same)
                const int slice_kb = knobs().solve_slice_kb;
                const int slice_max = std::max(1, std::min(64, knobs().solve_slice_max));
                const int64_t slice_from = knobs().solve_slice_from >= 0 ? knobs().solve_slice_from * 1024 : (int64_t)slice_kb * 1024 * 9 / 2;
                std::vector<int> tp, ts;
                h_tbase.assign((size_t)top_count + 1, 0);
                const int b0 = top_launches ? launches[launches.size() - top_launches].begin : 0;
                bool any_sliced = false, set_has_tall = false, tall_unsliceable = false;
                size_t slds = 0;
                for (int p = 0; p < top_count; ++p) {
                    const int sn = sched[(size_t)b0 + p];
                    const int f = front_size(sn), nc = S.sn_start[sn + 1] - S.sn_start[sn], nb = f - nc;
                    const int64_t wbytes = (int64_t)f * nc * 8;
                    int R = 1;
                    if (slice_kb > 0 && wbytes > slice_from)
                        R = (int)std::min<int64_t>(slice_max, (wbytes + (int64_t)slice_kb * 1024 - 1) / ((int64_t)slice_kb * 1024));
                    const bool tallf = front_is_tall(sn);                        // (too tall for k_top_solve's LDS: slices only)
                    if (tallf) R = std::max(R, 2);
                    set_has_tall = set_has_tall || tallf;
                    R = std::max(1, std::min(R, std::max(1, nb)));
                    if (tallf && R < 2) tall_unsliceable = true;
                    any_sliced = any_sliced || R > 1;
                    h_tbase[(size_t)p] = (int)tp.size();
                    for (int q = 0; q < R; ++q) { tp.push_back(p); ts.push_back(q | (R << 8)); }
                    const size_t nloc = (size_t)nc + (size_t)((nb + R - 1) / R) + 8;
                    const size_t fwd = (size_t)((nc + 3) & ~3) + nloc * (1 + (size_t)((nc + 7) >> 3));
                    const size_t bwd = (size_t)((f + 3) & ~3) + 16 * 16;
                    slds = std::max(slds, std::max(fwd, bwd) * sizeof(double));
                }
                h_tbase[(size_t)top_count] = (int)tp.size();
                if (any_sliced && top_count > 0) {
                    top_ntask = (int)tp.size();
                    top_nflag = top_ntask;
                    top_slds = slds;
                    top_sgrid = std::min(top_solve_sliced_capacity(slds), top_ntask);
                    // (two right-hand sides per sweep: twice the LDS, the same grid or none)
                    top_sgrid2 = slds * 2 <= 150 * 1024 ? std::min(top_solve_sliced_capacity(slds * 2, 2), top_ntask) : 0;
                    if (top_sgrid2 < top_sgrid) top_sgrid2 = 0;
                    d_tk_pos.upload(tp); d_tk_sl.upload(ts); d_tbase.upload(h_tbase);
                    xf.alloc((size_t)S.N * 2);
                    if (top_sgrid <= 0) top_ntask = 0;
                }
                if (set_has_tall && (top_ntask == 0 || tall_unsliceable)) {
                    // fronts too tall for the one-front-per-workgroup kernel, and the (front, slice) kernel cannot take the set
                    // either (a slice's vectors beyond a CU's LDS: fronts of ~18 000 rows and more): no persistent set at all,
                    // the sweeps go level by level (k_fwd_tall / k_bwd_tall for those fronts)
                    top_launches = 0; top_count = 0; top_grid = 0; top_ntask = 0; top_nflag = 1;
                }
            }
            top_flags.alloc((size_t)2 * std::max(top_nflag, 1) + 4);
            HIP_CHECK(hipMemset(top_flags.p, 0, top_flags.n * sizeof(int)));
            // chained launches: every launch must fit the chained kernels (no front beyond the block kernels' LDS)
            // chain_from: the longest suffix of launches with at most chain_max workgroups each (HIPKKT_CHAIN_MAX; the wide
            // levels below are throughput-bound: a launch each costs them little, while their thousands of waiting
            // workgroups would crowd a chained grid), every front of which fits the chained kernels' LDS
            {
                const int chain_max = knobs().chain_max;
                size_t q = launches.size();
                while (q > 0) {
                    const Launch& L = launches[q - 1];
                    ChainSeg sg{L.begin, L.small ? 0 : L.count, L.small ? L.count - L.ntiny : 0, L.small ? L.ntiny : 0, 0, 0, RecSeg{}};
                    if (L.ntall > 0 || (!L.small && L.lds_solve > 150 * 1024) || chain_seg_wgs(sg) > chain_max) break;
                    if (!L.small) chain_lds = std::max(chain_lds, L.lds_solve);
                    --q;
                }
                chain_from = launches.size() - q >= 2 ? q : launches.size();
                std::vector<int> nch((size_t)std::max(S.nsuper, 1), 0);
                if (chain_from < launches.size()) {
                    const int p0 = launches[chain_from].begin;
                    for (int c = 0; c < S.nsuper; ++c)
                        if (S.sn_parent[c] >= 0 && spos[(size_t)c] >= p0) nch[(size_t)S.sn_parent[c]]++;
                }
                d_chain_nchild.upload(nch);
                h_nch = nch;
            }
            d_chain.alloc((size_t)2 * std::max(S.nsuper, 1));
            HIP_CHECK(hipMemset(d_chain.p, 0, d_chain.n * sizeof(int)));
            tinv.alloc((size_t)toff[S.nsuper]);
            HIP_CHECK(hipMemset(tinv.p, 0, std::max<size_t>(tinv.n, 1) * sizeof(double)));
            d_tinv_list.upload(tinv_list);
            tinv_small_prefix.assign(tinv_list.size() + 1, 0);      // (how many narrow supernodes a stretch of the list holds: launch_tinv)
            for (size_t k = 0; k < tinv_list.size(); ++k) {
                const int s = tinv_list[k];
                tinv_small_prefix[k + 1] = tinv_small_prefix[k] + (S.sn_start[s + 1] - S.sn_start[s] <= winv_small_nc() ? 1 : 0);
            }
        }
        {
            std::vector<int64_t> cut_ptr(S.nsuper + 1, 0);
            std::vector<int> cuts;
            for (int c = 0; c < S.nsuper; ++c) {
                cut_ptr[c] = (int64_t)cuts.size();
                int p = S.sn_parent[c];
                if (p < 0) continue;
                int pnc = S.sn_start[p + 1] - S.sn_start[p];
                int pnb = (int)(S.rowptr[p + 1] - S.rowptr[p]);
                int nt = (pnb + 63) / 64;
                const int* rb = S.rel.data() + S.rowptr[c];
                int nbc = (int)(S.rowptr[c + 1] - S.rowptr[c]);
                for (int t = 0; t <= nt; ++t)
                    cuts.push_back((int)(std::lower_bound(rb, rb + nbc, pnc + 64 * t) - rb));
            }
            cut_ptr[S.nsuper] = (int64_t)cuts.size();
            d_cut_ptr.upload(cut_ptr);
            d_cuts.upload(cuts);
        }
        {
            // extend-add items and forward-solve gather lists, both keyed by the parent's local index
            const int64_t nloc = (int64_t)S.N + (int64_t)S.rows.size();
            std::vector<int64_t> ptr((size_t)nloc + 1, 0);
            auto lbase = [&](int s) { return (int64_t)S.sn_start[s] + S.rowptr[s]; };
            for (int c = 0; c < S.nsuper; ++c) {
                int p = S.sn_parent[c];
                if (p < 0) continue;
                for (int64_t q = S.rowptr[c]; q < S.rowptr[c + 1]; ++q) ptr[lbase(p) + S.rel[q] + 1]++;
            }
            for (int64_t i = 0; i < nloc; ++i) ptr[i + 1] += ptr[i];
            // gather lists of the forward sweep: one entry per child row, keyed by the receiving local row
            std::vector<int> gsrc((size_t)ptr[nloc]);
            {
                std::vector<int64_t> nxt(ptr.begin(), ptr.end() - 1);
                for (int p = 0; p < S.nsuper; ++p)
                    for (int e = S.child_ptr[p]; e < S.child_ptr[p + 1]; ++e) {     // children in fixed order
                        int c = S.child_idx[e];
                        for (int64_t q = S.rowptr[c]; q < S.rowptr[c + 1]; ++q) gsrc[nxt[lbase(p) + S.rel[q]]++] = (int)q;
                    }
            }
            if (S.rows.size() >= ((size_t)1 << 31)) throw std::runtime_error("row structure exceeds int32 indexing");
            if (knobs().verbose >= 2 && top_launches > 0) {
                // gather-list lengths of the rows of the persistent solve set (what k_top_solve's parked indices must cover)
                long hist[6] = {0, 0, 0, 0, 0, 0};
                long fronts_over8 = 0, fronts_over12 = 0, nf = 0;
                int64_t gmax = 0;
                for (size_t q = launches.size() - top_launches; q < launches.size(); ++q)
                    for (int t = launches[q].begin; t < launches[q].begin + launches[q].count; ++t) {
                        const int sn = sched[(size_t)t];
                        const int f = front_size(sn);
                        int64_t fm = 0;
                        for (int i = 0; i < f; ++i) {
                            const int64_t n = ptr[lbase(sn) + i + 1] - ptr[lbase(sn) + i];
                            hist[n == 0 ? 0 : n <= 4 ? 1 : n <= 8 ? 2 : n <= 12 ? 3 : n <= 16 ? 4 : 5]++;
                            fm = std::max(fm, n);
                        }
                        gmax = std::max(gmax, fm);
                        fronts_over8 += fm > 8; fronts_over12 += fm > 12; ++nf;
                    }
                std::fprintf(stderr, "[hipkkt] persistent solve set: gather sources per row: 0: %ld, 1-4: %ld, 5-8: %ld, 9-12: %ld, "
                             "13-16: %ld, more: %ld (max %lld); fronts with a row over 8: %ld, over 12: %ld of %ld\n",
                             hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], (long long)gmax, fronts_over8, fronts_over12, nf);
            }
            d_item_ptr.upload(ptr);
            // extend-add items of the PANEL columns (the update-block columns go through the Schur sub-items):
            // one record per child update column and 64-row piece of it, grouped by the (global, permuted) panel
            // column it lands in, children in fixed order -- a wave handles eight pieces at a time, all of them
            // a single load round, however long the child's column is
            static_assert(sizeof(ExtItem) == 32, "ExtItem layout");
            // DENSE CHILD: a child whose update block IS its parent's whole front (same rows in the same order: the
            // previous panel of a supernode that was cut into panels, the lower front of a chain) needs no work lists at
            // all -- parent entry (r, j) receives child entry (r, j).  The panel kernel and the Schur tiles add it as a
            // dense block (coalesced loads, no descriptors, no row lookups); its pieces stay out of both lists.
            std::vector<int> dense_child((size_t)S.nsuper, -1);
            std::vector<int64_t> dense_off((size_t)S.nsuper, -1);
            const bool dense_on = knobs().dense_child;
            int64_t n_dense = 0;
            for (int p = 0; dense_on && p < S.nsuper; ++p) {
                if (tile_base[p] < 0) continue;                              // block-class parents only
                const int fp = front_size(p);
                for (int e = S.child_ptr[p]; e < S.child_ptr[p + 1]; ++e) {
                    const int c = S.child_idx[e];
                    const int nbc = (int)(S.rowptr[c + 1] - S.rowptr[c]);
                    if (nbc != fp) continue;
                    bool ident = true;
                    for (int q = 0; ident && q < nbc; ++q) ident = S.rel[S.rowptr[c] + q] == q;
                    if (!ident) continue;
                    dense_child[(size_t)p] = c;
                    dense_off[(size_t)p] = S.upd_off[c];
                    ++n_dense;
                    break;
                }
            }
            d_dense_off.upload(dense_off);
            if (knobs().verbose) std::fprintf(stderr, "[hipkkt] dense children: %lld\n", (long long)n_dense);
            std::vector<int64_t> pptr((size_t)S.N + 1, 0);
            for (int c = 0; c < S.nsuper; ++c) {
                int p = S.sn_parent[c];
                if (p < 0 || dense_child[(size_t)p] == c) continue;
                const int pnc = S.sn_start[p + 1] - S.sn_start[p];
                const int nbc = (int)(S.rowptr[c + 1] - S.rowptr[c]);
                for (int b = 0; b < nbc; ++b) {
                    const int r = S.rel[S.rowptr[c] + b];
                    if (r >= pnc) break;                                   // rel ascends: the rest lands in U
                    pptr[(size_t)S.sn_start[p] + r + 1] += (nbc - b + 63) / 64;
                }
            }
            for (int i = 0; i < S.N; ++i) pptr[i + 1] += pptr[i];
            std::vector<ExtItem> items((size_t)pptr[S.N]);
            {
                std::vector<int64_t> nxt(pptr.begin(), pptr.end() - 1);
                for (int p = 0; p < S.nsuper; ++p) {
                    const int pnc = S.sn_start[p + 1] - S.sn_start[p];
                    for (int e = S.child_ptr[p]; e < S.child_ptr[p + 1]; ++e) {
                        int c = S.child_idx[e];
                        if (dense_child[(size_t)p] == c) continue;
                        int nbc = (int)(S.rowptr[c + 1] - S.rowptr[c]);
                        for (int b = 0; b < nbc; ++b) {
                            int64_t q = S.rowptr[c] + b;
                            const int r = S.rel[q];
                            if (r >= pnc) break;
                            for (int a = b; a < nbc; a += 64) {
                                ExtItem it;
                                it.uoff = S.upd_off[c] + (int64_t)b * nbc + a;
                                it.relstart = (int)(S.rowptr[c] + a);
                                it.cnt = std::min(64, nbc - a);
                                it.rfirst = S.rel[S.rowptr[c] + a];
                                it.rlast = S.rel[S.rowptr[c] + a + it.cnt - 1];
                                it.tcol = r;
                                it.pad = 0;
                                items[(size_t)nxt[(size_t)S.sn_start[p] + r]++] = it;
                            }
                        }
                    }
                }
            }
            // 16 slices of each supernode's panel items, cut on column boundaries
            std::vector<int64_t> wcut(((size_t)S.nsuper + slice_list.size()) * 17, 0);
            for (int s = 0; s < S.nsuper; ++s) {
                const int nc = S.sn_start[s + 1] - S.sn_start[s];
                const int64_t* cp = pptr.data() + S.sn_start[s];      // nc + 1 column pointers
                const int64_t I0 = cp[0], I1 = cp[nc];
                int j = 0;
                for (int w = 0; w < 16; ++w) {
                    const int64_t target = I0 + ((I1 - I0) * w) / 16;
                    while (j < nc && cp[j] < target) ++j;
                    wcut[(size_t)s * 17 + w] = cp[j];
                }
                wcut[(size_t)s * 17 + 16] = I1;
            }
            // A row slice of a sliced front gets an item list of its own: the pieces that reach its rows or the top block,
            // in the same order, cut into 16 wave slices of whole columns again -- instead of every slice walking the whole
            // front's list in rounds that are mostly skipped pieces.
            for (size_t q = 0; q < slice_list.size(); ++q) {
                const int s = slice_list[q][0], sl = slice_list[q][1], nsl = slice_list[q][2];
                const int ff = front_size(s), nc = S.sn_start[s + 1] - S.sn_start[s], nb = ff - nc;
                const int rsmax = (nb + nsl - 1) / nsl, r_lo = nc + sl * rsmax;
                const int rs = std::max(0, std::min(rsmax, ff - r_lo));
                const int64_t* cp = pptr.data() + S.sn_start[s];
                std::vector<int64_t> scp((size_t)nc + 1);
                for (int j = 0; j < nc; ++j) {
                    scp[(size_t)j] = (int64_t)items.size();
                    for (int64_t it = cp[j]; it < cp[j + 1]; ++it) {
                        const ExtItem e = items[(size_t)it];
                        if (e.rfirst >= nc && (e.rlast < r_lo || e.rfirst >= r_lo + rs)) continue;
                        items.push_back(e);
                    }
                }
                scp[(size_t)nc] = (int64_t)items.size();
                const int64_t I0 = scp[0], I1 = scp[(size_t)nc];
                int j = 0;
                int64_t* w = wcut.data() + ((size_t)S.nsuper + q) * 17;
                for (int k = 0; k < 16; ++k) {
                    const int64_t target = I0 + ((I1 - I0) * k) / 16;
                    while (j < nc && scp[(size_t)j] < target) ++j;
                    w[k] = scp[(size_t)j];
                }
                w[16] = I1;
            }
            std::vector<int64_t> raw(std::max<size_t>(items.size(), 1) * 4);
            std::memcpy(raw.data(), items.data(), items.size() * sizeof(ExtItem));
            d_items.upload(raw);
            d_wave_cut.upload(wcut);
            if (knobs().verbose) {
                if (knobs().verbose >= 2) {            // the extend-add work of the last fronts of the schedule
                    for (size_t q = sched.size() > 12 ? sched.size() - 12 : 0; q < sched.size(); ++q) {
                        const int sn = sched[q];
                        const int64_t* w = wcut.data() + (size_t)sn * 17;
                        int64_t mx = 0, rows = 0;
                        for (int k = 0; k < 16; ++k) mx = std::max(mx, w[k + 1] - w[k]);
                        for (int64_t it = w[0]; it < w[16]; ++it) rows += items[(size_t)it].cnt;
                        std::fprintf(stderr, "[hipkkt] front %d f %d nc %d kids %d: %lld panel items (%lld entries), largest wave slice %lld\n",
                                     sn, front_size(sn), S.sn_start[sn + 1] - S.sn_start[sn], S.child_ptr[sn + 1] - S.child_ptr[sn],
                                     (long long)(w[16] - w[0]), (long long)rows, (long long)mx);
                    }
                }
            }
            // Schur sub-items: every child update column that lands in a U column, cut at the parent's
            // 64-row tile boundaries, grouped by (tile, tile column), children in fixed order
            static_assert(sizeof(SubItem) == 16, "SubItem layout");
            const int64_t ntile = (int64_t)tiles.size();
            std::vector<int64_t> cnt((size_t)ntile * 64 + 1, 0);
            auto for_each_sub = [&](auto&& fn) {
                for (int p = 0; p < S.nsuper; ++p) {
                    if (tile_base[p] < 0) continue;
                    const int pnc = S.sn_start[p + 1] - S.sn_start[p];
                    for (int e = S.child_ptr[p]; e < S.child_ptr[p + 1]; ++e) {
                        const int c = S.child_idx[e];
                        if (dense_child[(size_t)p] == c) continue;
                        const int nbc = (int)(S.rowptr[c + 1] - S.rowptr[c]);
                        const int* rl = S.rel.data() + S.rowptr[c];
                        for (int b = 0; b < nbc; ++b) {
                            if (rl[b] < pnc) continue;
                            const int j = rl[b] - pnc, tj = j >> 6;
                            int a = b;
                            while (a < nbc) {
                                const int ti = (rl[a] - pnc) >> 6;
                                int a2 = a;
                                while (a2 < nbc && ((rl[a2] - pnc) >> 6) == ti) ++a2;
                                const int64_t tile = tile_base[p] + (int64_t)ti * (ti + 1) / 2 + tj;
                                fn(tile, j & 63, c, b, a, a2 - a, nbc);
                                a = a2;
                            }
                        }
                    }
                }
            };
            for_each_sub([&](int64_t tile, int q, int, int, int, int, int) { cnt[tile * 64 + q + 1]++; });
            for (size_t i = 0; i + 1 < cnt.size(); ++i) cnt[i + 1] += cnt[i];
            std::vector<SubItem> sit((size_t)cnt.back());
            {
                std::vector<int64_t> nx(cnt.begin(), cnt.end() - 1);
                for_each_sub([&](int64_t tile, int q, int c, int b, int a, int n, int nbc) {
                    SubItem si;
                    si.uoff = S.upd_off[c] + (int64_t)b * nbc + a;
                    si.relstart = (int)(S.rowptr[c] + a);
                    si.cnt = (unsigned char)n;
                    si.qcol = (unsigned char)q;
                    si.pad = 0;
                    sit[(size_t)nx[tile * 64 + q]++] = si;
                });
            }
            std::vector<int64_t> tcut((size_t)ntile * 5 + 5, 0);
            for (int64_t t = 0; t < ntile; ++t) {
                const int64_t* cp = cnt.data() + t * 64;       // 65 column pointers of this tile
                const int64_t I0 = cp[0], I1 = cp[64];
                int q = 0;
                for (int w = 0; w < 4; ++w) {
                    const int64_t target = I0 + ((I1 - I0) * w) / 4;
                    while (q < 64 && cp[q] < target) ++q;
                    tcut[t * 5 + w] = cp[q];
                }
                tcut[t * 5 + 4] = I1;
            }
            {
                // XCD-aware launch order of a wide level's tiles (HIPKKT_TILE_XCD=n: launches with at least n fronts;
                // 0 = off): workgroup i of a launch runs on XCD i mod 8, each XCD has its own L2, and every tile of a front
                // reads that front's L21 strips -- a front's tiles are dealt to ONE residue class (eight fronts interleaved,
                // fronts in work order so that a group's fronts have similar tile counts), so its strips come from HBM
                // once instead of once per XCD that happens to hold one of its tiles.
                // Measured on cfg2 (rocprofv3 --pmc FETCH_SIZE, r03): k_schur's reads 838 -> 525 MB per factorisation with the
                // wide levels (>= 150 fronts) permuted, factorisation 1.754 / 1.760 -> 1.743 / 1.751 ms; permuting the narrow
                // levels as well costs time (>= 32: 1.766 / 1.770 -- a handful of fronts' tiles then crowd one XCD).
                const int tile_xcd = knobs().tile_xcd;
                if (tile_xcd > 0) {
                    std::vector<int64_t> nt2(tiles.size()), nc2(tcut.size(), 0);
                    nt2 = tiles;
                    nc2 = tcut;
                    for (const Launch& L : launches) {
                        if (L.small || L.count < tile_xcd || L.ntiles <= 0) continue;
                        // the launch's fronts and their (contiguous) logical tile ranges
                        std::vector<std::pair<int64_t, int64_t>> rng;       // [first, last) per front, launch order
                        for (int t = L.begin; t < L.begin + L.count; ++t) {
                            const int sn = sched[(size_t)t];
                            if (tile_base[sn] < 0) continue;
                            const int nb = front_size(sn) - (S.sn_start[sn + 1] - S.sn_start[sn]);
                            const int k = (nb + 63) / 64;
                            if (k > 0) rng.push_back({tile_base[sn], tile_base[sn] + (int64_t)k * (k + 1) / 2});
                        }
                        int64_t pos = L.tile_begin;
                        for (size_t g0 = 0; g0 < rng.size(); g0 += 8) {
                            const size_t g1 = std::min(rng.size(), g0 + 8);
                            int64_t longest = 0;
                            for (size_t x = g0; x < g1; ++x) longest = std::max(longest, rng[x].second - rng[x].first);
                            for (int64_t j = 0; j < longest; ++j)
                                for (size_t x = g0; x < g1; ++x)
                                    if (rng[x].first + j < rng[x].second) {
                                        const int64_t from = rng[x].first + j;
                                        nt2[(size_t)pos] = tiles[(size_t)from];
                                        for (int w = 0; w < 5; ++w) nc2[(size_t)pos * 5 + w] = tcut[(size_t)from * 5 + w];
                                        ++pos;
                                    }
                        }
                        if (pos != (int64_t)L.tile_begin + L.ntiles) throw std::runtime_error("tile permutation lost a tile");
                    }
                    tiles.swap(nt2);
                    tcut.swap(nc2);
                }
            }
            d_tiles.upload(tiles);
            std::vector<int64_t> raw2(sit.size() * 2);
            std::memcpy(raw2.data(), sit.data(), sit.size() * sizeof(SubItem));
            d_sitems.upload(raw2);
            d_tile_cut.upload(tcut);
            d_gl_src.upload(gsrc);
            build_records(ptr, gsrc);
            // where each contribution entry sits in its receiver's gather list (solve_multi's layout)
            std::vector<int> ud(std::max<size_t>(S.rows.size(), 1), 0);
            for (size_t g = 0; g < gsrc.size(); ++g) ud[(size_t)gsrc[g]] = (int)g;
            d_udst.upload(ud);
            // PULLED LEAVES (many-column sweeps only).  Half of a sparse KKT system's contribution rows come from leaves with
            // ONE column (cfg2: 113 k of them, the slack rows of the linear cone): forward, such a leaf only forwards
            // u_r = -L(r,0) b_c to its parent's rows -- 4 KB per row and 512 columns, written and read once.  In the
            // many-column sweeps the PARENT computes those terms itself from the leaf's row of B (one row instead of nb, and
            // shared by the parent's rows through L2), the leaf does nothing forward, and its backward step is a gather
            // kernel of its own (k_bwd_leaf_m).  The parent's gather list is split: the stored contributions of its other
            // children (contiguous rows of the receiver-ordered store, as before, without the pulled ones) and the pulled
            // terms (leaf column, position of L(r,0) in the fronts).
            {
                std::vector<int> row_sn(std::max<size_t>(S.rows.size(), 1), -1);
                for (int c = 0; c < S.nsuper; ++c)
                    for (int64_t q = S.rowptr[c]; q < S.rowptr[c + 1]; ++q) row_sn[(size_t)q] = c;
                std::vector<int64_t> mptr((size_t)nloc + 1, 0), prp(1, 0);
                std::vector<int> udm(std::max<size_t>(S.rows.size(), 1), -1), hcol, prs;
                std::vector<int64_t> hl;
                int64_t slot = 0;
                for (int64_t lc = 0; lc < nloc; ++lc) {
                    mptr[(size_t)lc] = slot;
                    bool any = false;
                    for (int64_t g = ptr[(size_t)lc]; g < ptr[(size_t)lc + 1] && !any; ++g) any = pulled[(size_t)row_sn[(size_t)gsrc[(size_t)g]]] != 0;
                    if (any) prs.push_back((int)slot++);                  // the sum of this row's pulled terms: first of its run
                    for (int64_t g = ptr[(size_t)lc]; g < ptr[(size_t)lc + 1]; ++g) {
                        const int q = gsrc[(size_t)g], c = row_sn[(size_t)q];
                        if (pulled[(size_t)c]) {
                            hcol.push_back(S.sn_start[c]);
                            hl.push_back(S.front_off[c] + 1 + ((int64_t)q - S.rowptr[c]));     // L(r, 0): column-major, ld f, nc = 1
                        } else {
                            udm[(size_t)q] = (int)slot++;
                        }
                    }
                    if (any) prp.push_back((int64_t)hcol.size());
                }
                mptr[(size_t)nloc] = slot;
                n_pull_rows = (int)prs.size();
                if (hcol.empty()) { hcol.push_back(0); hl.push_back(0); }
                if (prs.empty()) prs.push_back(0);
                std::vector<int> hrow(hcol.size());
                for (size_t k = 0; k < hcol.size(); ++k) hrow[k] = S.perm[(size_t)hcol[k]];
                d_glm_ptr.upload(mptr); d_udst_m.upload(udm); d_hp_col.upload(hcol); d_hp_row.upload(hrow); d_hp_lidx.upload(hl);
                d_pr_ptr.upload(prp); d_pr_slot.upload(prs);
                if (knobs().verbose)
                    std::fprintf(stderr, "[hipkkt] many-column sweeps: %lld of %lld contribution rows pulled from one-column leaves into %d sums\n",
                                 (long long)(hcol.size()), (long long)ptr[(size_t)nloc], n_pull_rows);
            }
        }
        d_perm.upload(S.perm);
        std::vector<signed char> ps(S.N);
        for (int k = 0; k < S.N; ++k) ps[k] = (signed char)(dsigns[S.perm[k]] >= 0 ? 1 : -1);
        d_psign.upload(ps);
        {
            size_t ws = 0;
            for (const Launch& L : launches) if (L.ntall > 0) ws = std::max(ws, tall_ws_doubles(S.N, L.ntall, L.fmax));
            if (ws) { tall_ws.alloc(ws); tall_ws.zero(nullptr); HIP_CHECK(hipStreamSynchronize(nullptr)); }     // (its first N doubles: ticket words, zero between sweeps)
        }
        fronts.alloc((size_t)S.front_store);
        upd.alloc((size_t)S.update_store);
        Dinv.alloc((size_t)S.N);
        xp.alloc((size_t)S.N);
        uvec.alloc(S.rows.size());
        flags.alloc(12);      // [0..2] status words, [3..6] which wait expired (diagnostic), [8..9] concurrency probe
        HIP_CHECK(hipMemset(fronts.p, 0, std::max<size_t>(fronts.n, 1) * sizeof(double)));
        HIP_CHECK(hipMemset(Dinv.p, 0, std::max<size_t>(Dinv.n, 1) * sizeof(double)));
        HIP_CHECK(hipMemset(flags.p, 0, 12 * sizeof(int)));
    }
};

static void fill_info(const Symbolic& S, hipkkt_info* info)
{
    info->N = S.N;
    info->nnzK = S.nnzK;
    info->nnzL = S.nnzL_struct;
    info->nnzL_stored = S.nnzL;
    info->nsuper = S.nsuper;
    info->nlevels = (int64_t)S.levels.size();
    info->max_front = S.max_front;
    info->etree_height = S.etree_height;
    info->factor_flops = S.flops;
    info->front_bytes = (double)S.front_store * 8.0;
    info->update_bytes = (double)S.update_store * 8.0;
}

struct PinnedScalars {
    double* h = nullptr;
    PinnedScalars() { HIP_CHECK(hipHostMalloc((void**)&h, 64 * sizeof(double))); }   // [8..11] update status, [16..35] refinement read-back, [40..47] sticky
    ~PinnedScalars() { if (h) (void)hipHostFree(h); }
};

// phase timing with hipEvents on the handle's stream
struct Profiler {
    bool enabled = false;
    struct Span { hipEvent_t a, b; int phase; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    hipkkt_profile acc{};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    int begin(int phase, hipStream_t st)
    {
        if (!enabled) return -1;
        Span s{get(), get(), phase};
        HIP_CHECK(hipEventRecord(s.a, st));
        spans.push_back(s);
        return (int)spans.size() - 1;
    }
    void end(int id, hipStream_t st)
    {
        if (id < 0) return;
        HIP_CHECK(hipEventRecord(spans[id].b, st));
    }
    void resolve()
    {
        for (Span& s : spans) {
            HIP_CHECK(hipEventSynchronize(s.b));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, s.a, s.b));
            switch (s.phase) {
            case 0: acc.update_ms += ms; acc.n_update++; break;
            case 1: acc.factor_ms += ms; acc.n_factor++; break;
            case 2: acc.trisolve_ms += ms; acc.n_trisolve++; break;
            case 3: acc.residual_ms += ms; acc.n_residual++; break;
            default: acc.other_ms += ms;
            }
            pool.push_back(s.a);
            pool.push_back(s.b);
        }
        spans.clear();
    }
    ~Profiler()
    {
        for (Span& s : spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    }
};

}  // namespace hipkkt

using namespace hipkkt;

// ====================================================================================
//  handles
// ====================================================================================
struct hipkkt_ldl_s {
    int device = 0;
    hipkkt_settings st;
    hipStream_t stream = nullptr;
    int64_t N = 0, nnzK = 0;
    int base = 0;
    std::unique_ptr<LDLEngine> eng;
    DBuf<double> Kval, vals, b, x;
    DBuf<double> mB;             // host-pointer solve_multi staging, N x mcap
    size_t mcap = 0;
    DBuf<int> idx;
    ~hipkkt_ldl_s() { if (stream) (void)hipStreamDestroy(stream); }
};

struct hipkkt_kkt_s {
    int device = 0;
    hipkkt_settings st;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    KKTAssembly K;
    std::unique_ptr<LDLEngine> eng;
    // values and maps
    DBuf<double> Kval, Pval, Aval;
    DBuf<int> mapP, mapA, mapHs, mapDiag, mapU, mapV, mapD, soc_of_entry;
    // residual SpMV
    DBuf<int64_t> fptr;
    DBuf<int> fcol, fmap;
    DBuf<int64_t> fpend;         // rows < n: end of the row's P entries in the image (SpmvDev::pend)
    DBuf<double> fval;               // K values in the CSR image's order: gathered whole after P / A changed (fval_dirty), kept
    bool fval_dirty = true;          //   current by write-through from the cone update otherwise (kpos: two slots per K entry)
    DBuf<int> kpos;
    // long rows of the image (kernels.hpp, SpmvDev)
    DBuf<int> long_rows;
    DBuf<int64_t> long_chunk_ptr, chunk_q;
    DBuf<double> long_partial;
    int nlong = 0, nchunks = 0;
    size_t long_partial_cols = 0;
    int lanes_per_row = 8;
    // vectors
    DBuf<double> b, x, e, dx, rx, rz, sbuf, zbuf, ybuf;
    double *cur_x = nullptr, *cur_dx = nullptr;
    DBuf<double> partial, scal;      // scal: [0] eps, [1] norme, [2] normb
    std::unique_ptr<PinnedScalars> pin;
    // cones
    DBuf<int> c_kind, c_off, c_numel, c_sidx, c_soff, c_elem, c_soclist;
    DBuf<int64_t> c_boff;
    DBuf<double> w, eta, soc_u, soc_v, soc_eta2, Hs;
    DBuf<int> fail;
    int nsoc = 0, npsd = 0, psd_kmax = 1;
    DBuf<int> c_psdlist, c_psddim;
    DBuf<int64_t> c_psdaoff;
    DBuf<double> psdA, psdR, psdRinv;
    bool has_psd = false, psd_too_big = false, scaling_valid = false;
    double last_eps = 0;
    int64_t last_ir = 0;
    // refinement on the device (k_ir_round): state slots (4 doubles per round), a 5-double read-back record, the
    // sticky deferred-status record (8 doubles)
    DBuf<double> irbuf;
    double *ir_state = nullptr, *ir_readback = nullptr, *ir_sticky = nullptr, *ir_norms = nullptr, *sys_out_dev = nullptr;
    int ir_stride = 0;
    int r_spec = 1;                  // refinement rounds enqueued ahead of the first read-back (= what the previous solve took)
    bool publish_ok = true;          // the status record reaches the host by the call's last kernel (kernels.hpp: Publish); false: by a copy
    long long publish_seq = 0;
    int spec_low_calls = 0;          // status records in a row whose solves took fewer rounds than were enqueued (kkt_eval_sticky)
    bool deferred = false;           // hipkkt_kkt_set_deferred_status
    // level C (DefaultKKTSystem on the device, kktsystem.jl:21-215)
    DBuf<double> lam;                                        // scaled point, m
    DBuf<double> sq, snegq, sb, sx2, sz2, sworkx, sworkz, sconic, spa, spb, spc;
    DBuf<double> sys_partial, sys_dots, sys_cached, sys_in;
    bool sys_ready = false;
    bool sys_lazy = false;           // hipkkt_kkt_system_set_lazy: kkt_update! leaves (x2, z2) = K \ (-q, b) to the affine kkt_solve!
    bool sys_const_pending = false;  // ... and that solve is still due
    bool sys_update_unread = false;  // lazy mode: the last kkt_update!'s status sits in the sticky record, read with the next solve's
    DBuf<double> hst;                // staging of the *_host entry points of level C: 3 x (n + 2 m) doubles (rhs, variables, lhs)
    bool host_vars_valid = false;    // hst holds the variables of the previous hipkkt_kkt_system_solve_host call
    // "the caller's host arrays may be reused": an event behind the uploads, waited for at the end of the call -- the
    // kernels enqueued in between keep the device busy meanwhile (a stream synchronise would wait for them as well)
    hipEvent_t ev_upload = nullptr;
    void host_upload_mark(hipStream_t st)
    {
        if (!ev_upload) HIP_CHECK(hipEventCreateWithFlags(&ev_upload, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev_upload, st));
    }
    void host_upload_wait() { if (ev_upload) HIP_CHECK(hipEventSynchronize(ev_upload)); }
    // solve_multi work space, N x mcap each (grown on demand)
    DBuf<double> mB, mX, mC, mE, mE2, mpartial, mnorms;
    DBuf<int> mmask;
    size_t mcap = 0;
    Profiler prof;
    ~hipkkt_kkt_s()
    {
        if (ev_upload) (void)hipEventDestroy(ev_upload);
        if (stream && own_stream) (void)hipStreamDestroy(stream);
    }

    ConeDev cone_dev() const
    {
        ConeDev C;
        C.ncones = (int)K.cones.size();
        C.kind = c_kind.p; C.off = c_off.p; C.numel = c_numel.p; C.boff = c_boff.p; C.sidx = c_sidx.p;
        C.soff = c_soff.p; C.elem_cone = c_elem.p; C.soc_list = c_soclist.p; C.nsoc = nsoc;
        C.psd_list = c_psdlist.p; C.psd_dim = c_psddim.p; C.psd_aoff = c_psdaoff.p; C.npsd = npsd; C.psd_kmax = psd_kmax;
        return C;
    }
    ConeState cone_state()
    {
        ConeState S;
        S.w = w.p; S.eta = eta.p; S.u = soc_u.p; S.v = soc_v.p; S.eta2 = soc_eta2.p; S.Hs = Hs.p; S.fail = fail.p;
        S.psdA = psdA.p; S.psdR = psdR.p; S.psdRinv = psdRinv.p;
        S.lam = lam.p;
        return S;
    }
};

template <class F>
static int guarded(F&& f)
{
    try {
        return f();
    } catch (const ArgError& e) {
        g_last_error = e.what();
        return HIPKKT_ERR_ARG;
    } catch (const HipError& e) {
        g_last_error = e.what();
        return HIPKKT_ERR_HIP;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return HIPKKT_ERR_INTERNAL;
    }
}

static int select_device(const hipkkt_settings& st)
{
    int dev = st.device;
    if (dev < 0) HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipSetDevice(dev));
    return dev;
}

extern "C" {

int hipkkt_available(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return std::strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

const char* hipkkt_last_error(void) { return g_last_error.c_str(); }
const char* hipkkt_version(void) { return "hipkkt 0.1.0 (gfx950)"; }

void hipkkt_default_settings(hipkkt_settings* s)
{
    s->static_regularization_constant = 1e-8;
    s->static_regularization_proportional = DBL_EPSILON * DBL_EPSILON;
    s->dynamic_regularization_eps = 1e-13;
    s->dynamic_regularization_delta = 2e-7;
    s->iterative_refinement_reltol = 1e-13;
    s->iterative_refinement_abstol = 1e-12;
    s->iterative_refinement_stop_ratio = 5.0;
    s->iterative_refinement_max_iter = 10;
    s->static_regularization_enable = 1;
    s->iterative_refinement_enable = 1;
    s->ordering = HIPKKT_ORDER_ND;
    s->nd_leaf_size = 1000;
    s->device = -1;
    s->user_perm = nullptr;
    s->amd_dense_scale = 1.5;
}

int hipkkt_symbolic_analyse(int64_t N, const int64_t* colptr, const int64_t* rowval, int base, int ordering,
                            int nd_leaf_size, int64_t* perm_out, hipkkt_info* info_out)
{
    return guarded([&]() {
        if (!colptr || !rowval || N <= 0 || N > 2000000000 || (base != 0 && base != 1))
            throw ArgError("hipkkt_symbolic_analyse: bad argument");
        if (ordering != HIPKKT_ORDER_AMD && ordering != HIPKKT_ORDER_ND && ordering != HIPKKT_ORDER_NATURAL)
            throw ArgError("hipkkt_symbolic_analyse: ordering must be AMD, ND or NATURAL");
        SymbolicOptions opt;
        opt.ordering = ordering;
        if (nd_leaf_size > 0) opt.nd_leaf_size = nd_leaf_size;
        apply_knobs(opt);
        Symbolic S;
        analyse((int)N, colptr, rowval, base, opt, S);
        if (knobs().dump_levels) {          // diagnostic: the shape of every tree level (host only)
            for (size_t l = 0; l < S.levels.size(); ++l) {
                int cnt = 0, fmax = 0, ncmax = 0, n8 = 0, n64 = 0;
                int lds3 = 0, lds2 = 0, lds1 = 0;        // panels (f > 64) that would fit 3 / 2 / 1 to a CU's LDS
                double flops = 0, panel = 0, upd = 0, cols = 0;
                for (int t = S.levels[l].begin; t < S.levels[l].end; ++t) {
                    const int sn = S.level_sn[t];
                    const int nc = S.sn_start[sn + 1] - S.sn_start[sn], nb = (int)(S.rowptr[sn + 1] - S.rowptr[sn]), f = nc + nb;
                    ++cnt; fmax = std::max(fmax, f); ncmax = std::max(ncmax, nc);
                    n8 += f <= 8; n64 += f <= 64;
                    if (f > 64) {
                        const size_t b = panel_lds_bytes(f, f * nc - nc * (nc - 1) / 2);
                        (b <= 52 * 1024 ? lds3 : b <= 79 * 1024 ? lds2 : lds1)++;
                    }
                    cols += nc; panel += (double)f * nc; upd += (double)nb * nb;
                    for (int j = 0; j < nc; ++j) { const double c = f - 1 - j; flops += c * c + 3 * c; }
                }
                std::fprintf(stderr, "[levels] %2zu: %6d fronts (%6d f<=8, %6d f<=64) fmax %4d ncmax %3d cols %7.0f panel %.2f MB "
                             "upd %.2f MB flops %.1f M | panels by LDS <=52K %d, <=79K %d, more %d\n", l, cnt, n8, n64, fmax, ncmax, cols, panel * 8e-6,
                             upd * 8e-6, flops * 1e-6, lds3, lds2, lds1);
            }
        }
        if (knobs().dump_subtrees) {        // diagnostic: what the subtrees below a cut level look like (host only)
            {
                std::vector<int> hist(8, 0);
                int mx = 0;
                for (int sn = 0; sn < S.nsuper; ++sn) {
                    const int k = S.child_ptr[sn + 1] - S.child_ptr[sn];
                    mx = std::max(mx, k);
                    hist[k == 0 ? 0 : k <= 4 ? 1 : k <= 16 ? 2 : k <= 64 ? 3 : k <= 256 ? 4 : k <= 1024 ? 5 : 6]++;
                }
                std::fprintf(stderr, "[subtrees] children per front: 0: %d, 1-4: %d, 5-16: %d, 17-64: %d, 65-256: %d, 257-1024: %d, more: %d (max %d)\n",
                             hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], mx);
            }
            for (int cut = 0; cut < (int)S.levels.size() && cut < 10; ++cut) {
                std::vector<int> root_of((size_t)S.nsuper, -1);
                // level_sn is leaves first: walk from the top so that a parent's root is known before its children's
                for (int t = S.nsuper - 1; t >= 0; --t) {
                    const int sn = S.level_sn[(size_t)t];
                    if (S.sn_level[(size_t)sn] > cut) continue;
                    const int p = S.sn_parent[(size_t)sn];
                    root_of[(size_t)sn] = (p >= 0 && S.sn_level[(size_t)p] <= cut) ? root_of[(size_t)p] : sn;
                }
                struct St { double fronts = 0, cols = 0, snb = 0, bytes = 0, border = 0; int fmax = 0; };
                std::map<int, St> st;
                for (int sn = 0; sn < S.nsuper; ++sn) {
                    if (root_of[(size_t)sn] < 0) continue;
                    St& x = st[root_of[(size_t)sn]];
                    const int nc = S.sn_start[sn + 1] - S.sn_start[sn], nb = (int)(S.rowptr[sn + 1] - S.rowptr[sn]);
                    x.fronts += 1; x.cols += nc; x.snb += nb; x.bytes += 8.0 * (nc + nb) * nc * (nc + nb > 64 ? 2 : 1);
                    x.fmax = std::max(x.fmax, nc + nb);
                    if (root_of[(size_t)sn] == sn) x.border = nb;
                }
                St mx, sum;
                for (auto& kv : st) {
                    const St& x = kv.second;
                    mx.fronts = std::max(mx.fronts, x.fronts); mx.cols = std::max(mx.cols, x.cols); mx.snb = std::max(mx.snb, x.snb);
                    mx.bytes = std::max(mx.bytes, x.bytes); mx.border = std::max(mx.border, x.border); mx.fmax = std::max(mx.fmax, x.fmax);
                    sum.fronts += x.fronts; sum.cols += x.cols; sum.snb += x.snb; sum.bytes += x.bytes;
                }
                const double k = st.empty() ? 1.0 : (double)st.size();
                std::fprintf(stderr, "[subtrees] cut %d: %zu subtrees; fronts %.0f total, %.0f mean, %.0f max; cols %.0f mean %.0f max; "
                             "sum nb %.0f mean %.0f max; matrix %.1f MB total, %.0f KB mean, %.0f KB max; border max %.0f, fmax %d\n",
                             cut, st.size(), sum.fronts, sum.fronts / k, mx.fronts, sum.cols / k, mx.cols, sum.snb / k, mx.snb,
                             sum.bytes * 1e-6, sum.bytes / k * 1e-3, mx.bytes * 1e-3, mx.border, mx.fmax);
            }
        }
        if (perm_out) for (int64_t i = 0; i < N; ++i) perm_out[i] = S.perm[i];
        if (info_out) { std::memset(info_out, 0, sizeof(*info_out)); fill_info(S, info_out); }
        return HIPKKT_OK;
    });
}

// ------------------------------------------------------------------------ level A
int hipkkt_ldl_create(hipkkt_ldl_t* out, int64_t N, const int64_t* colptr, const int64_t* rowval,
                      const double* nzval, const int64_t* dsigns, const hipkkt_settings* settings, int base)
{
    return guarded([&]() {
        if (!out || !colptr || !rowval || !nzval || !dsigns || N <= 0 || N > 2000000000)
            throw ArgError("hipkkt_ldl_create: bad argument");
        if (base != 0 && base != 1) throw ArgError("index_base must be 0 or 1");
        std::unique_ptr<hipkkt_ldl_s> h(new hipkkt_ldl_s);
        if (settings) h->st = *settings; else hipkkt_default_settings(&h->st);
        h->device = select_device(h->st);
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->N = N;
        h->nnzK = colptr[N] - base;
        h->base = base;
        std::vector<int> ds((size_t)N);
        for (int64_t i = 0; i < N; ++i) ds[i] = dsigns[i] >= 0 ? 1 : -1;
        h->eng.reset(new LDLEngine((int)N, colptr, rowval, base, ds, h->st));
        h->eng->stream = h->stream;
        h->eng->set_device(h->device);
        h->Kval.alloc((size_t)h->nnzK);
        HIP_CHECK(hipMemcpy(h->Kval.p, nzval, (size_t)h->nnzK * sizeof(double), hipMemcpyHostToDevice));
        h->b.alloc((size_t)N);
        h->x.alloc((size_t)N);
        *out = h.release();
        return HIPKKT_OK;
    });
}

void hipkkt_ldl_destroy(hipkkt_ldl_t h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
}

static void ldl_stage_index(hipkkt_ldl_t h, const int64_t* index, int64_t k)
{
    std::vector<int> idx((size_t)k);
    for (int64_t i = 0; i < k; ++i) {
        int64_t v = index[i] - h->base;
        if (v < 0 || v >= h->nnzK) throw ArgError("value index out of range");
        idx[i] = (int)v;
    }
    if (h->idx.n < (size_t)k) h->idx.alloc((size_t)k);
    HIP_CHECK(hipMemcpyAsync(h->idx.p, idx.data(), (size_t)k * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
}

int hipkkt_ldl_update_values(hipkkt_ldl_t h, const int64_t* index, const double* values, int64_t k)
{
    return guarded([&]() {
        if (!h || k < 0 || (k > 0 && (!index || !values))) throw ArgError("hipkkt_ldl_update_values: bad argument");
        if (k == 0) return HIPKKT_OK;
        HIP_CHECK(hipSetDevice(h->device));
        ldl_stage_index(h, index, k);
        if (h->vals.n < (size_t)k) h->vals.alloc((size_t)k);
        HIP_CHECK(hipMemcpyAsync(h->vals.p, values, (size_t)k * sizeof(double), hipMemcpyHostToDevice, h->stream));
        launch_scatter(h->Kval.p, h->idx.p, h->vals.p, k, 1.0, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_ldl_scale_values(hipkkt_ldl_t h, const int64_t* index, double scale, int64_t k)
{
    return guarded([&]() {
        if (!h || k < 0 || (k > 0 && !index)) throw ArgError("hipkkt_ldl_scale_values: bad argument");
        if (k == 0) return HIPKKT_OK;
        HIP_CHECK(hipSetDevice(h->device));
        ldl_stage_index(h, index, k);
        launch_scale(h->Kval.p, h->idx.p, scale, k, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_ldl_refactor(hipkkt_ldl_t h)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        HIP_CHECK(hipSetDevice(h->device));
        h->eng->factor(h->Kval.p, nullptr);
        int fl[3];
        h->eng->read_flags(fl);
        if (fl[2]) {                                            // never expected; see LDLEngine::ov_gave_up
            h->eng->ov_gave_up();
            h->eng->factor(h->Kval.p, nullptr);
            h->eng->read_flags(fl);
        }
        return fl[1] ? HIPKKT_NUMERIC_FAILURE : HIPKKT_OK;      // directldl_qdldl.jl:79
    });
}

int hipkkt_ldl_solve_dev(hipkkt_ldl_t h, double* d_x, const double* d_b)
{
    return guarded([&]() {
        if (!h || !d_x || !d_b) throw ArgError("hipkkt_ldl_solve_dev: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        h->eng->solve(d_b, d_x);
        return HIPKKT_OK;
    });
}

int hipkkt_ldl_solve(hipkkt_ldl_t h, double* x, const double* b)
{
    return guarded([&]() {
        if (!h || !x || !b) throw ArgError("hipkkt_ldl_solve: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        HIP_CHECK(hipMemcpyAsync(h->b.p, b, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
        h->eng->solve(h->b.p, h->x.p, true);
        if (h->eng->top_aborted_sync()) {                  // never expected; see TopOwner
            h->eng->top_gave_up();
            h->eng->solve(h->b.p, h->x.p, false);
        }
        HIP_CHECK(hipMemcpyAsync(x, h->x.p, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_ldl_solve_multi_dev(hipkkt_ldl_t h, int64_t nrhs, double* d_X, int64_t ldx, const double* d_B, int64_t ldb)
{
    return guarded([&]() {
        if (!h || nrhs < 0 || (nrhs > 0 && (!d_X || !d_B || ldx < h->N || ldb < h->N)))
            throw ArgError("hipkkt_ldl_solve_multi_dev: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        h->eng->solve_multi(d_B, ldb, d_X, ldx, (int)nrhs);
        return HIPKKT_OK;
    });
}

int hipkkt_ldl_solve_multi(hipkkt_ldl_t h, int64_t nrhs, double* X, const double* B)
{
    return guarded([&]() {
        if (!h || nrhs < 0 || (nrhs > 0 && (!X || !B))) throw ArgError("hipkkt_ldl_solve_multi: bad argument");
        if (nrhs == 0) return HIPKKT_OK;
        HIP_CHECK(hipSetDevice(h->device));
        if ((size_t)nrhs > h->mcap) { h->mB.alloc((size_t)h->N * nrhs); h->mcap = (size_t)nrhs; }
        const size_t bytes = (size_t)h->N * nrhs * sizeof(double);
        HIP_CHECK(hipMemcpyAsync(h->mB.p, B, bytes, hipMemcpyHostToDevice, h->stream));
        h->eng->solve_multi(h->mB.p, h->N, h->mB.p, h->N, (int)nrhs);     // in place: every entry is read before its slot is written
        HIP_CHECK(hipMemcpyAsync(X, h->mB.p, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_ldl_info(hipkkt_ldl_t h, hipkkt_info* info)
{
    return guarded([&]() {
        if (!h || !info) throw ArgError("hipkkt_ldl_info: bad argument");
        std::memset(info, 0, sizeof(*info));
        fill_info(h->eng->S, info);
        return HIPKKT_OK;
    });
}

int hipkkt_ldl_get_perm(hipkkt_ldl_t h, int64_t* perm)
{
    return guarded([&]() {
        if (!h || !perm) throw ArgError("hipkkt_ldl_get_perm: bad argument");
        for (int64_t i = 0; i < h->N; ++i) perm[i] = h->eng->S.perm[i];
        return HIPKKT_OK;
    });
}

int hipkkt_ldl_fallbacks(hipkkt_ldl_t h, int64_t out[2])
{
    return guarded([&]() {
        if (!h || !out) throw ArgError("hipkkt_ldl_fallbacks: bad argument");
        out[0] = h->eng->n_ov_fallbacks;
        out[1] = h->eng->n_top_fallbacks;
        return HIPKKT_OK;
    });
}

// ------------------------------------------------------------------------ level B
int hipkkt_kkt_create(hipkkt_kkt_t* out, int64_t n, int64_t m, const int64_t* Pcolptr, const int64_t* Prowval,
                      const double* Pnzval, const int64_t* Acolptr, const int64_t* Arowval, const double* Anzval,
                      int64_t ncones, const int32_t* kinds, const int64_t* dims, const hipkkt_settings* settings,
                      int base)
{
    return guarded([&]() {
        if (!out || !Pcolptr || !Acolptr || n < 0 || m < 0 || ncones < 0 || (ncones > 0 && (!kinds || !dims)))
            throw ArgError("hipkkt_kkt_create: bad argument");
        if (base != 0 && base != 1) throw ArgError("index_base must be 0 or 1");
        if (n + m == 0) throw ArgError("empty problem");
        std::unique_ptr<hipkkt_kkt_s> h(new hipkkt_kkt_s);
        if (settings) h->st = *settings; else hipkkt_default_settings(&h->st);
        // host-side assembly first: a malformed (P, A, cones) is an argument error whether or not a device is there
        try {
            assemble_kkt(n, m, Pcolptr, Prowval, Pnzval, Acolptr, Arowval, Anzval, ncones, kinds, dims, base, h->K);
        } catch (const std::runtime_error& e) {
            throw ArgError(e.what());
        }
        h->device = select_device(h->st);
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        KKTAssembly& K = h->K;
        hipkkt_settings st = h->st;
        if (st.ordering == HIPKKT_ORDER_USER && !st.user_perm) throw ArgError("ORDER_USER needs user_perm");
        std::vector<int64_t> rv(K.rowval.begin(), K.rowval.end());
        // K is 0-based here; a user permutation stays in the caller's base
        std::vector<int64_t> uperm;
        if (st.ordering == HIPKKT_ORDER_USER) {
            uperm.resize(K.N);
            for (int i = 0; i < K.N; ++i) uperm[i] = st.user_perm[i] - base;
            st.user_perm = uperm.data();
        }
        h->eng.reset(new LDLEngine(K.N, K.colptr.data(), rv.data(), 0, K.dsigns, st));
        h->eng->stream = h->stream;
        h->eng->set_device(h->device);
        h->st.user_perm = nullptr;

        h->Kval.upload(K.nzval);
        h->mapP.upload(K.mapP);
        h->mapA.upload(K.mapA);
        h->mapHs.upload(K.mapHs);
        h->mapDiag.upload(K.map_diag);
        h->mapU.upload(K.mapU);
        h->mapV.upload(K.mapV);
        h->mapD.upload(K.mapD);
        h->Pval.alloc(K.mapP.size());
        h->Aval.alloc(K.mapA.size());
        // full symmetric CSR image of K for the residual
        {
            const int N = K.N;
            std::vector<int64_t> ptr((size_t)N + 1, 0);
            for (int j = 0; j < N; ++j)
                for (int64_t q = K.colptr[j]; q < K.colptr[j + 1]; ++q) {
                    int i = K.rowval[q];
                    ptr[i + 1]++;
                    if (i != j) ptr[j + 1]++;
                }
            for (int i = 0; i < N; ++i) ptr[i + 1] += ptr[i];
            std::vector<int> col((size_t)ptr[N]), vmap((size_t)ptr[N]);
            std::vector<int64_t> nx(ptr.begin(), ptr.end() - 1);
            // row i gets its upper-triangle partners (i, j>i) from column scans in ascending j and its
            // lower partners from its own column; fill lower part first so columns ascend within a row
            for (int j = 0; j < N; ++j) {            // entries (i<j) stored in column j: row j, col i
                for (int64_t q = K.colptr[j]; q < K.colptr[j + 1]; ++q) {
                    int i = K.rowval[q];
                    int64_t d = nx[j]++;
                    col[d] = i;
                    vmap[d] = (int)q;
                }
            }
            for (int j = 0; j < N; ++j)              // mirrored entries: row i, col j (j > i), ascending j
                for (int64_t q = K.colptr[j]; q < K.colptr[j + 1]; ++q) {
                    int i = K.rowval[q];
                    if (i == j) continue;
                    int64_t d = nx[i]++;
                    col[d] = j;
                    vmap[d] = (int)q;
                }
            h->fptr.upload(ptr);
            h->fcol.upload(col);
            h->fmap.upload(vmap);
            {
                // rows of the x block: where their P entries end (columns ascend, so those are a prefix of the row) -- the
                // reduced-system layer's P x products stop there instead of walking the row's A' entries as well (cfg3:
                // ~250 of them per row behind a handful of P entries)
                std::vector<int64_t> pend((size_t)std::max<int64_t>(K.n, 1), 0);
                for (int64_t i = 0; i < K.n; ++i)
                    pend[(size_t)i] = std::lower_bound(col.begin() + ptr[(size_t)i], col.begin() + ptr[(size_t)i + 1], (int)K.n) - col.begin();
                h->fpend.upload(pend);
            }
            h->fval.alloc(vmap.size());
            {
                std::vector<int> kp((size_t)2 * K.nnzK, -1);
                for (size_t q = 0; q < vmap.size(); ++q) {
                    const size_t e = (size_t)vmap[q];
                    if (kp[2 * e] < 0) kp[2 * e] = (int)q; else kp[2 * e + 1] = (int)q;
                }
                h->kpos.upload(kp);
            }
            {
                std::vector<int> lrows;
                std::vector<int64_t> lptr{0}, cq;
                for (int i = 0; i < N; ++i) {
                    const int64_t len = ptr[i + 1] - ptr[i];
                    if (len <= kLongRow) continue;
                    lrows.push_back(i);
                    for (int64_t q = ptr[i]; q < ptr[i + 1]; q += kLongChunk) {
                        cq.push_back(q);
                        cq.push_back(std::min<int64_t>(q + kLongChunk, ptr[i + 1]));
                    }
                    lptr.push_back((int64_t)cq.size() / 2);
                }
                h->nlong = (int)lrows.size();
                h->nchunks = (int)(cq.size() / 2);
                if (h->nlong) {
                    h->long_rows.upload(lrows);
                    h->long_chunk_ptr.upload(lptr);
                    h->chunk_q.upload(cq);
                    h->long_partial.alloc((size_t)h->nchunks * kMaxNR);
                    h->long_partial_cols = kMaxNR;
                }
            }
            double avg = (double)ptr[N] / std::max(N, 1);
            h->lanes_per_row = avg > 24.0 ? 64 : 8;
        }
        const size_t N = (size_t)K.N;
        // right-hand side, solution, residual and candidate: up to kMaxNR columns share a sweep (kkt_solve_core)
        h->b.alloc(N * kMaxNR); h->x.alloc(N * kMaxNR); h->e.alloc(N * kMaxNR); h->dx.alloc(N * kMaxNR);
        HIP_CHECK(hipMemset(h->b.p, 0, N * kMaxNR * sizeof(double)));
        HIP_CHECK(hipMemset(h->x.p, 0, N * kMaxNR * sizeof(double)));
        h->cur_x = h->x.p;
        h->cur_dx = h->dx.p;
        h->rx.alloc((size_t)K.n); h->rz.alloc((size_t)K.m);
        h->sbuf.alloc((size_t)K.m); h->zbuf.alloc((size_t)K.m); h->ybuf.alloc((size_t)K.m);
        h->partial.alloc(2 * ((size_t)std::max(2, kMaxNR) * (2 * kNormParts + 1) + 8));      // (two halves: kkt_partials)
        h->scal.alloc(16);               // [0] eps, [1] norme, [2] normb, [3] abort, [4] speculative norme, [8..11] update status
        HIP_CHECK(hipMemset(h->scal.p, 0, 16 * sizeof(double)));
        h->pin.reset(new PinnedScalars);
        {
            h->ir_stride = 4 * (std::max(h->st.iterative_refinement_max_iter, 0) + 2);
            const size_t nstate = (size_t)h->ir_stride * kMaxNR;
            const size_t total = nstate + 5 * kMaxNR + 4 + 8 + 4 + 3 * kMaxNR;
            h->irbuf.alloc(total);
            HIP_CHECK(hipMemset(h->irbuf.p, 0, total * sizeof(double)));
            h->ir_state = h->irbuf.p;
            h->ir_readback = h->irbuf.p + nstate;
            h->ir_sticky = h->ir_readback + 5 * kMaxNR + 4;
            h->sys_out_dev = h->ir_sticky + 8;          // {dtau, dkappa, tau_num, tau_den} of kkt_solve!: read back with the record in ONE copy
            h->ir_norms = h->sys_out_dev + 4;           // several columns: norme0[kMaxNR], normb[kMaxNR], cand[kMaxNR]
        }
        // cones
        {
            size_t nc = K.cones.size();
            std::vector<int> kind(nc), off(nc), numel(nc), sidx(nc), soff(nc), elem((size_t)K.m), soclist, soc_of;
            std::vector<int64_t> boff(nc);
            soc_of.assign((size_t)K.sparse_len, 0);
            for (size_t c = 0; c < nc; ++c) {
                const ConeInfo& ci = K.cones[c];
                kind[c] = ci.kind; off[c] = ci.off; numel[c] = ci.numel; boff[c] = ci.boff;
                sidx[c] = ci.sparse ? ci.sidx : -1;
                soff[c] = ci.sparse ? ci.soff : -1;
                for (int t = 0; t < ci.numel; ++t) elem[ci.off + t] = (int)c;
                if (ci.kind == HIPKKT_CONE_SOC) soclist.push_back((int)c);
                if (ci.kind == HIPKKT_CONE_PSD) {
                    h->has_psd = true;
                    if (ci.dim > kPsdMaxDim) h->psd_too_big = true;
                }
                if (ci.sparse) for (int t = 0; t < ci.numel; ++t) soc_of[ci.soff + t] = ci.sidx;
            }
            h->nsoc = (int)soclist.size();
            {
                std::vector<int> plist, pdim(nc, 0);
                std::vector<int64_t> paoff(nc, 0);
                int64_t ao = 0;
                for (size_t c = 0; c < nc; ++c) {
                    const ConeInfo& ci = K.cones[c];
                    if (ci.kind != HIPKKT_CONE_PSD) continue;
                    plist.push_back((int)c);
                    pdim[c] = ci.dim;
                    paoff[c] = ao;
                    ao += (int64_t)ci.dim * ci.dim;
                    h->psd_kmax = std::max(h->psd_kmax, ci.dim);
                }
                h->npsd = (int)plist.size();
                h->c_psdlist.upload(plist); h->c_psddim.upload(pdim); h->c_psdaoff.upload(paoff);
                h->psdA.alloc((size_t)ao); h->psdR.alloc((size_t)ao); h->psdRinv.alloc((size_t)ao);
            }
            h->c_kind.upload(kind); h->c_off.upload(off); h->c_numel.upload(numel); h->c_boff.upload(boff);
            h->c_sidx.upload(sidx); h->c_soff.upload(soff); h->c_elem.upload(elem); h->c_soclist.upload(soclist);
            h->soc_of_entry.upload(soc_of);
            h->w.alloc((size_t)K.m); h->eta.alloc(nc); h->lam.alloc((size_t)K.m);
            h->soc_u.alloc((size_t)K.sparse_len); h->soc_v.alloc((size_t)K.sparse_len);
            h->soc_eta2.alloc((size_t)K.nsparse); h->Hs.alloc((size_t)K.nHs);
            h->fail.alloc(1);
            HIP_CHECK(hipMemset(h->fail.p, 0, sizeof(int)));
        }
        *out = h.release();
        return HIPKKT_OK;
    });
}

void hipkkt_kkt_destroy(hipkkt_kkt_t h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
}

int hipkkt_kkt_info(hipkkt_kkt_t h, hipkkt_info* info)
{
    return guarded([&]() {
        if (!h || !info) throw ArgError("hipkkt_kkt_info: bad argument");
        std::memset(info, 0, sizeof(*info));
        fill_info(h->eng->S, info);
        info->n = h->K.n; info->m = h->K.m; info->p = h->K.p;
        info->nHs = h->K.nHs; info->nsparse_soc = h->K.nsparse; info->sparse_soc_len = h->K.sparse_len;
        return HIPKKT_OK;
    });
}

// scatter of -Hs and the sparse-cone columns, static regulariser, numeric factorisation
// (kktsolver_directldl.jl:211-294).  Hs/u/v/eta2 already on the device.
static int kkt_update_device(hipkkt_kkt_t h, bool deferred = false)
{
    KKTAssembly& K = h->K;
    int pu = h->prof.begin(0, h->stream);
    // -Hs (:225-228) and the sparse cones' columns (:235-241) in one launch, written through to the residual's copy of K
    // when that copy is current (otherwise it is gathered whole before the next residual: kkt_spmv)
    launch_update_values(h->Kval.p, h->mapHs.p, h->Hs.p, (int)K.nHs, h->mapU.p, h->mapV.p, h->mapD.p, h->soc_u.p, h->soc_v.p,
                         h->soc_eta2.p, h->soc_of_entry.p, K.sparse_len, K.nsparse, h->fval_dirty ? nullptr : h->fval.p,
                         h->kpos.p, h->stream);
    const double* eps_ptr = nullptr;
    if (h->st.static_regularization_enable) {                                      // :259-279
        launch_regularizer(h->Kval.p, h->mapDiag.p, K.N, h->st.static_regularization_constant,
                           h->st.static_regularization_proportional, h->partial.p, h->scal.p, h->stream);
        eps_ptr = h->scal.p;
    }
    h->prof.end(pu, h->stream);
    int pf = h->prof.begin(1, h->stream);
    h->eng->factor(h->Kval.p, eps_ptr);          // K itself stays un-regularised (:283-291)
    h->prof.end(pf, h->stream);
    // read back: flags, eps, cone failure
    // (one small kernel gathers the four words, one copy into pinned memory brings them over: three separate
    // copies, two of them into pageable memory, cost ~60 us of idle GPU per update)
    // (deferred: no read-back, the status joins the sticky record in the same launch -- hipkkt_kkt_deferred_status)
    launch_collect_status(h->scal.p + 8, h->scal.p, h->fail.p, h->eng->flags_ptr(), h->stream, deferred ? h->ir_sticky : nullptr);
    if (deferred) return HIPKKT_OK;
    HIP_CHECK(hipMemcpyAsync(h->pin->h + 8, h->scal.p + 8, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->pin->h[12] != 0.0) {                      // never expected; see LDLEngine::ov_gave_up: repeat level by level
        h->eng->ov_gave_up();
        h->eng->factor(h->Kval.p, eps_ptr);
        launch_collect_status(h->scal.p + 8, h->scal.p, h->fail.p, h->eng->flags_ptr(), h->stream);
        HIP_CHECK(hipMemcpyAsync(h->pin->h + 8, h->scal.p + 8, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    h->last_eps = h->st.static_regularization_enable ? h->pin->h[8] : 0.0;
    h->prof.acc.dynamic_regularizations += (int64_t)h->pin->h[10];
    if (h->pin->h[9] != 0.0) return HIPKKT_NUMERIC_FAILURE;
    return h->pin->h[11] != 0.0 ? HIPKKT_NUMERIC_FAILURE : HIPKKT_OK;
}

int hipkkt_kkt_update_cones(hipkkt_kkt_t h, const double* Hs, const double* soc_u, const double* soc_v,
                            const double* soc_eta2)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        KKTAssembly& K = h->K;
        if ((K.nHs > 0 && !Hs) || (K.nsparse > 0 && (!soc_u || !soc_v || !soc_eta2)))
            throw ArgError("hipkkt_kkt_update_cones: missing cone data");
        HIP_CHECK(hipSetDevice(h->device));
        if (K.nHs) HIP_CHECK(hipMemcpyAsync(h->Hs.p, Hs, (size_t)K.nHs * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (K.nsparse) {
            HIP_CHECK(hipMemcpyAsync(h->soc_u.p, soc_u, (size_t)K.sparse_len * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(h->soc_v.p, soc_v, (size_t)K.sparse_len * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(h->soc_eta2.p, soc_eta2, (size_t)K.nsparse * sizeof(double), hipMemcpyHostToDevice, h->stream));
        }
        launch_zero_ints(h->fail.p, 1, h->stream);
        h->scaling_valid = false;
        return kkt_update_device(h);
    });
}

static int kkt_update_from_sz_dev_impl(hipkkt_kkt_t h, const double* d_s, const double* d_z, bool deferred)
{
    return guarded([&]() {
        if (!h || (h->K.m > 0 && (!d_s || !d_z))) throw ArgError("hipkkt_kkt_update_from_sz: bad argument");
        if (h->psd_too_big)
            throw ArgError("update_from_sz: PSD cones with side > 48 are scaled by the caller; use hipkkt_kkt_update_cones");
        HIP_CHECK(hipSetDevice(h->device));
        launch_zero_ints(h->fail.p, 1, h->stream);
        int pu = h->prof.begin(0, h->stream);
        launch_cone_scaling(h->cone_dev(), h->cone_state(), d_s, d_z, h->K.m, h->stream);
        h->prof.end(pu, h->stream);
        h->scaling_valid = true;
        return kkt_update_device(h, deferred);
    });
}
int hipkkt_kkt_update_from_sz_dev(hipkkt_kkt_t h, const double* d_s, const double* d_z)
{
    return kkt_update_from_sz_dev_impl(h, d_s, d_z, h && h->deferred);
}

int hipkkt_kkt_update_from_sz(hipkkt_kkt_t h, const double* s, const double* z)
{
    return guarded([&]() {
        if (!h || (h->K.m > 0 && (!s || !z))) throw ArgError("hipkkt_kkt_update_from_sz: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        size_t bytes = (size_t)h->K.m * sizeof(double);
        if (bytes) {
            HIP_CHECK(hipMemcpyAsync(h->sbuf.p, s, bytes, hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(h->zbuf.p, z, bytes, hipMemcpyHostToDevice, h->stream));
        }
        return hipkkt_kkt_update_from_sz_dev(h, h->sbuf.p, h->zbuf.p);
    });
}

int hipkkt_kkt_update_P(hipkkt_kkt_t h, const double* Pnzval)
{
    return guarded([&]() {
        if (!h || (h->K.mapP.size() && !Pnzval)) throw ArgError("hipkkt_kkt_update_P: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        size_t k = h->K.mapP.size();
        if (!k) return HIPKKT_OK;
        HIP_CHECK(hipMemcpyAsync(h->Pval.p, Pnzval, k * sizeof(double), hipMemcpyHostToDevice, h->stream));
        launch_scatter(h->Kval.p, h->mapP.p, h->Pval.p, (int64_t)k, 1.0, h->stream);
        h->fval_dirty = true;
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_update_A(hipkkt_kkt_t h, const double* Anzval)
{
    return guarded([&]() {
        if (!h || (h->K.mapA.size() && !Anzval)) throw ArgError("hipkkt_kkt_update_A: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        size_t k = h->K.mapA.size();
        if (!k) return HIPKKT_OK;
        HIP_CHECK(hipMemcpyAsync(h->Aval.p, Anzval, k * sizeof(double), hipMemcpyHostToDevice, h->stream));
        launch_scatter(h->Kval.p, h->mapA.p, h->Aval.p, (int64_t)k, 1.0, h->stream);
        h->fval_dirty = true;
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_setrhs_dev(hipkkt_kkt_t h, const double* d_rx, const double* d_rz)
{
    return guarded([&]() {
        if (!h || (h->K.n && !d_rx) || (h->K.m && !d_rz)) throw ArgError("hipkkt_kkt_setrhs: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        launch_pack_rhs(h->b.p, d_rx, d_rz, h->K.n, h->K.m, h->K.p, h->stream);
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_setrhs(hipkkt_kkt_t h, const double* rx, const double* rz)
{
    return guarded([&]() {
        if (!h || (h->K.n && !rx) || (h->K.m && !rz)) throw ArgError("hipkkt_kkt_setrhs: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        if (h->K.n) HIP_CHECK(hipMemcpyAsync(h->rx.p, rx, (size_t)h->K.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (h->K.m) HIP_CHECK(hipMemcpyAsync(h->rz.p, rz, (size_t)h->K.m * sizeof(double), hipMemcpyHostToDevice, h->stream));
        launch_pack_rhs(h->b.p, h->rx.p, h->rz.p, h->K.n, h->K.m, h->K.p, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));     // rx/rz staging may be overwritten by the next call
        return HIPKKT_OK;
    });
}

// the full-symmetric CSR image of K with its values streamed in CSR order
static SpmvDev kkt_spmv(hipkkt_kkt_t h)
{
    if (h->fval_dirty) {
        launch_gather_values(h->fval.p, h->Kval.p, h->fmap.p, (int64_t)h->fmap.n, h->stream);
        h->fval_dirty = false;
    }
    SpmvDev A;
    A.ptr = h->fptr.p; A.col = h->fcol.p; A.vmap = h->fmap.p; A.val = h->fval.p; A.N = h->K.N;
    A.pend = h->fpend.p;
    A.lanes_per_row = h->lanes_per_row;
    A.nlong = h->nlong; A.nchunks = h->nchunks; A.long_rows = h->long_rows.p; A.long_chunk_ptr = h->long_chunk_ptr.p;
    A.chunk_q = h->chunk_q.p; A.long_partial = h->long_partial.p;
    return A;
}

// e = b - K xi for nr columns (ld N); ||e_c||_inf -> norm_out[c], optionally ||b_c||_inf -> normb_out[c]; scal[3] carries
// the persistent solve kernel's abort word along.  Nothing is read back here.
// The norms' second stage: a finishing kernel behind the residual, or -- kkt_partials_mode -- left to k_ir_round, which
// reduces the partial maxima itself (kernels.hpp: IrPartials).  The first residual of a solve leaves its partials in the
// first half of h->partial, the candidates' residuals in the second half (both are read by the same k_ir_round).
static bool kkt_partials_mode(hipkkt_kkt_t h, int nr) { return residual_partials_ok(kkt_spmv(h), nr); }
static double* kkt_partials(hipkkt_kkt_t h, bool candidate) { return h->partial.p + (candidate ? h->partial.n / 2 : 0); }
static void kkt_enqueue_refine_error(hipkkt_kkt_t h, const double* xi, double* norm_out, double* normb_out, int nr, bool candidate)
{
    const SpmvDev A = kkt_spmv(h);
    const bool unfinished = kkt_partials_mode(h, nr);
    int pr = h->prof.begin(3, h->stream);
    // (||b|| rides along with the FIRST residual only; its slot is still named so that the b partials are written)
    launch_residual(A, h->Kval.p, h->b.p, xi, h->e.p, kkt_partials(h, candidate), unfinished ? nullptr : norm_out, h->stream, nr,
                    nr > 1 ? (int64_t)h->K.N : 0, h->eng->top_abort_word(), h->scal.p + 3, normb_out);
    h->prof.end(pr, h->stream);
}
static void kkt_trisolve(hipkkt_kkt_t h, const double* rhs, double* out, bool allow_top = true, int nr = 1)
{
    int ps = h->prof.begin(2, h->stream);
    h->eng->solve(rhs, out, allow_top, nr, h->K.N, h->K.N);
    h->prof.end(ps, h->stream);
}

// kktsolver_solve! with _iterative_refinement (kktsolver_directldl.jl:346-449); leaves the solution in h->x.
// The accept / stop rule of the loop runs on the device (k_ir_round): the host enqueues the first solve, its
// residual and r_spec refinement rounds (what the previous solve on this handle took) without waiting, then reads ONE
// 5-double record back -- or nothing at all in deferred-status mode.  Only when the device reports that the
// reference's loop would go on does the host add rounds, one read-back each.  A candidate is computed into h->dx and
// copied over h->x when accepted, so the solution always sits in the same buffer.
// Returns HIPKKT_OK or HIPKKT_NUMERIC_FAILURE.
struct IrNorms { double *norme0, *normb, *cand; };
static IrNorms kkt_ir_norms(hipkkt_kkt_t h, int nr)
{
    // one column: the residual kernel's single-column slots in scal; several: the per-column arrays
    if (nr == 1) return IrNorms{h->scal.p + 1, h->scal.p + 2, h->scal.p + 4};
    return IrNorms{h->ir_norms, h->ir_norms + kMaxNR, h->ir_norms + 2 * kMaxNR};
}
static void kkt_launch_ir(hipkkt_kkt_t h, int r, bool first, bool readback, bool fold, int nr)
{
    const hipkkt_settings& st = h->st;
    const IrNorms nm = kkt_ir_norms(h, nr);
    const double* abortw = h->eng->top_abort_word() ? h->scal.p + 3 : nullptr;
    IrPartials Q;
    const bool unfinished = kkt_partials_mode(h, nr);
    if (unfinished) {
        Q.np = residual_grid(kkt_spmv(h)) + 1;
        Q.e0 = kkt_partials(h, false);
        Q.b0 = Q.e0 + (size_t)nr * Q.np;
        Q.cand = r > 0 ? kkt_partials(h, true) : nullptr;
        Q.flag_in = h->eng->top_abort_word();
    }
    launch_ir_round(h->ir_state, h->ir_stride, r, first, nm.norme0, nm.normb, nm.cand, abortw, h->x.p, h->dx.p, h->K.N, nr,
                    st.iterative_refinement_abstol, st.iterative_refinement_reltol, st.iterative_refinement_stop_ratio,
                    std::max(st.iterative_refinement_max_iter, 0), readback ? h->ir_readback : nullptr,
                    (fold && (nr == 1 || unfinished)) ? h->ir_sticky : nullptr, h->stream, Q);
    if (fold && nr > 1 && !unfinished) launch_ir_fold(h->ir_state, h->ir_stride, r, nr, abortw, h->ir_sticky, h->stream);
}
static void kkt_enqueue_round(hipkkt_kkt_t h, int r, bool first, bool readback, bool fold, int nr)
{
    kkt_trisolve(h, h->e.p, h->dx.p, true, nr);                                   // dx = K^{-1} e
    launch_axpby_sum(h->dx.p, h->dx.p, h->x.p, (int64_t)h->K.N * nr, h->stream);  // prospective solution x + dx
    kkt_enqueue_refine_error(h, h->dx.p, kkt_ir_norms(h, nr).cand, nullptr, nr, true);  // e <- b - K (x + dx), its norm -> cand
    kkt_launch_ir(h, r, first, readback, fold, nr);
}
// nr = 1, 2 or 4 right-hand sides (columns of h->b, N apart) share every sweep; each column goes through the
// reference's loop on its own.  ir_out (nullable, nr entries): rounds per column (-1 in deferred mode).
static int kkt_solve_core(hipkkt_kkt_t h, bool may_defer = false, int nr = 1, int64_t* ir_out = nullptr, bool force_defer = false)
{
    const hipkkt_settings& st = h->st;
    const bool deferred = force_defer || (may_defer && h->deferred);
    h->cur_x = h->x.p;
    h->cur_dx = h->dx.p;
    h->last_ir = 0;
    if (ir_out) for (int c = 0; c < nr; ++c) ir_out[c] = deferred ? -1 : 0;
    if (!st.iterative_refinement_enable) {
        kkt_trisolve(h, h->b.p, h->x.p, false, nr);   // nothing reads the abort word back on this path
        int bad = 0;
        launch_zero_ints(h->fail.p, 1, h->stream);
        launch_check_finite(h->x.p, h->K.N * nr, h->fail.p, h->stream);
        if (deferred) { launch_fold_flag(h->ir_sticky, h->fail.p, h->stream); return HIPKKT_OK; }
        HIP_CHECK(hipMemcpyAsync(&bad, h->fail.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return bad ? HIPKKT_NUMERIC_FAILURE : HIPKKT_OK;
    }
    const int max_iter = std::max(st.iterative_refinement_max_iter, 0);
    int R = std::min(max_iter, std::max(h->r_spec, deferred ? 1 : 0));
    const IrNorms nm = kkt_ir_norms(h, nr);
    kkt_trisolve(h, h->b.p, h->x.p, true, nr);        // (deferred mode reads the persistent kernel's abort word at the status query)
    kkt_enqueue_refine_error(h, h->x.p, nm.norme0, nm.normb, nr, false);     // e = b - K x, ||e||, ||b||
    if (R == 0) kkt_launch_ir(h, 0, true, !deferred, deferred, nr);
    for (int r = 1; r <= R; ++r) kkt_enqueue_round(h, r, r == 1, r == R && !deferred, r == R && deferred, nr);
    if (deferred) { h->last_ir = -1; return HIPKKT_OK; }
    int64_t total = 0;
    for (;;) {
        HIP_CHECK(hipMemcpyAsync(h->pin->h + 16, h->ir_readback, 5 * (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        bool any_active = false, any_bad = false, aborted = false;
        total = 0;
        for (int c = 0; c < nr; ++c) {
            const double* rb = h->pin->h + 16 + 5 * c;
            any_active = any_active || rb[0] != 0.0;
            any_bad = any_bad || rb[2] != 0.0;
            aborted = aborted || rb[4] != 0.0;
            total += (int64_t)rb[1];
            if (ir_out) ir_out[c] = (int64_t)rb[1];
        }
        if (aborted && h->eng->top_abort_word() != nullptr) {           // never expected; see TopOwner
            h->eng->top_gave_up();                                      // the per-level path from now on: repeat the solve
            return kkt_solve_core(h, false, nr, ir_out);
        }
        h->last_ir = total;
        if (any_bad) { h->prof.acc.ir_iterations += total; return HIPKKT_NUMERIC_FAILURE; }
        if (!any_active || R >= max_iter) break;
        ++R;                                                            // the reference's loop goes on: one more round
        kkt_enqueue_round(h, R, false, true, false, nr);
    }
    h->prof.acc.ir_iterations += total;
    int64_t most = 0;
    for (int c = 0; c < nr; ++c) most = std::max<int64_t>(most, (int64_t)h->pin->h[16 + 5 * c + 1]);
    h->r_spec = (int)std::min<int64_t>(most, max_iter);
    return HIPKKT_OK;
}

int hipkkt_kkt_solve_dev(hipkkt_kkt_t h, double* d_lhsx, double* d_lhsz)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        HIP_CHECK(hipSetDevice(h->device));
        int rc = kkt_solve_core(h, true);
        if (rc != HIPKKT_OK) return rc;
        launch_unpack_lhs(d_lhsx, d_lhsz, h->cur_x, h->K.n, h->K.m, h->stream);
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_speculative_rounds(hipkkt_kkt_t h, int set)
{
    if (!h) return -1;
    if (set >= 0) { h->r_spec = std::min(set, std::max(h->st.iterative_refinement_max_iter, 0)); h->spec_low_calls = 0; }
    return h->r_spec;
}

int hipkkt_kkt_set_deferred_status(hipkkt_kkt_t h, int defer)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        HIP_CHECK(hipSetDevice(h->device));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        launch_zero_ints((int*)h->ir_sticky, 16, h->stream);
        h->deferred = defer != 0;
        return HIPKKT_OK;
    });
}

// what the sticky record {bad, more, abort, rounds, #dyn. regularisations, eps, solves, overlap gave up} read back into
// pinned memory says about everything enqueued since the last query: OK / NUMERIC_FAILURE / REFINEMENT_INCOMPLETE
// gave_up_out (nullable): a bounded wait expired -- the factor (or the solution) is void, as opposed to "a solve would
// have refined further".  will_repeat: the caller repeats the whole attempt synchronously whenever this returns
// HIPKKT_REFINEMENT_INCOMPLETE, so that attempt's counts stay out of the profile (the repeat adds its own).
static int kkt_eval_sticky(hipkkt_kkt_t h, const double* s, bool* gave_up_out = nullptr, bool will_repeat = false)
{
        const int max_iter = std::max(h->st.iterative_refinement_max_iter, 0);
        const bool void_step = s[7] != 0.0 || (s[2] != 0.0 && h->eng->top_abort_word() != nullptr);
        const bool repeat = will_repeat && (void_step || (s[0] == 0.0 && s[1] != 0.0));
        if (gave_up_out) *gave_up_out = false;
        if (!repeat) h->prof.acc.ir_iterations += (int64_t)s[3];
        if (!(will_repeat && void_step)) h->prof.acc.dynamic_regularizations += (int64_t)s[4];   // (a void factorisation is repeated too)
        if (s[5] != 0.0 || !h->st.static_regularization_enable) h->last_eps = h->st.static_regularization_enable ? s[5] : 0.0;
        if (s[6] > 0.0) h->last_ir = (int64_t)(s[3] / s[6] + 0.5);      // mean rounds per solve since the last query
        // The give-up words come first (as on the synchronous paths: kkt_update_device, kkt_solve_core): after a bounded
        // wait expired the factor or the solution is void and may well be non-finite -- that is not a numeric failure of
        // the problem but a step to repeat, with the mechanism switched off.
        bool gave_up = false;
        if (s[7] != 0.0) { h->eng->ov_gave_up(); gave_up = true; }      // never expected; see LDLEngine::ov_gave_up
        if (s[2] != 0.0 && h->eng->top_abort_word() != nullptr) { h->eng->top_gave_up(); gave_up = true; }   // see TopOwner
        if (gave_up_out) *gave_up_out = gave_up;
        if (gave_up) return HIPKKT_REFINEMENT_INCOMPLETE;               // (the step's results are void: repeat it)
        if (s[0] != 0.0) return HIPKKT_NUMERIC_FAILURE;
        if (s[1] != 0.0) {                                              // some solve would have gone on refining
            h->r_spec = std::min(max_iter, h->r_spec + 1);
            h->spec_low_calls = 0;
            return HIPKKT_REFINEMENT_INCOMPLETE;
        }
        // The speculation depth comes down again: one solve that took two rounds (a borderline accept / stop decision)
        // would otherwise cost every later solve of the run a wasted sweep pair (seen once as 3.77 instead of 3.36 ms per
        // step).  Eight records in a row whose solves could all have done with one round fewer: one round fewer.
        if (s[6] > 0.0 && h->r_spec > 1 && s[3] <= s[6] * (double)(h->r_spec - 1)) {
            if (++h->spec_low_calls >= 8) { --h->r_spec; h->spec_low_calls = 0; }
        } else if (s[6] > 0.0) {
            h->spec_low_calls = 0;
        }
        return HIPKKT_OK;
}

int hipkkt_kkt_deferred_status(hipkkt_kkt_t h)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        HIP_CHECK(hipSetDevice(h->device));
        HIP_CHECK(hipMemcpyAsync(h->pin->h + 40, h->ir_sticky, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        launch_zero_ints((int*)h->ir_sticky, 16, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->sys_update_unread = false;
        return kkt_eval_sticky(h, h->pin->h + 40);
    });
}

int hipkkt_kkt_solve(hipkkt_kkt_t h, double* lhsx, double* lhsz)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        HIP_CHECK(hipSetDevice(h->device));
        int rc = kkt_solve_core(h);
        if (rc != HIPKKT_OK) return rc;
        if (lhsx && h->K.n)
            HIP_CHECK(hipMemcpyAsync(lhsx, h->cur_x, (size_t)h->K.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (lhsz && h->K.m)
            HIP_CHECK(hipMemcpyAsync(lhsz, h->cur_x + h->K.n, (size_t)h->K.m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

// ---- several right-hand sides against one factorisation (SURVEY.md 8b "solve_multi", 8e(ii)).
// Per column exactly the sequence of kkt_solve_core: the refinement rounds run in lockstep over the
// columns that still need them; a column that has met its stopping rule is frozen (its candidate is
// not accepted any more), so every column ends where its own single solve would.
static void kkt_multi_reserve(hipkkt_kkt_t h, size_t k)
{
    k = (k + 15) & ~(size_t)15;                      // the row-major path pads the columns to a multiple of 16
    if (k <= h->mcap) return;
    const size_t N = (size_t)h->K.N;
    h->mB.alloc(N * k); h->mX.alloc(N * k); h->mC.alloc(N * k); h->mE.alloc(N * k); h->mE2.alloc(N * k);
    h->mpartial.alloc(std::max((size_t)(kNormParts + 1) * k, (size_t)2 * 512 * k));
    if (h->nlong && k > h->long_partial_cols) { h->long_partial.alloc((size_t)h->nchunks * k); h->long_partial_cols = k; }
    h->mnorms.alloc(2 * k);
    h->mmask.alloc(k);                               // (k is padded: enough for the row-major path's KP entries)
    h->mcap = k;
}

static int kkt_solve_multi_core(hipkkt_kkt_t h, int k, int64_t* ir_out)
{
    const hipkkt_settings& st = h->st;
    const int N = (int)h->K.N;
    const int64_t ld = N;
    std::vector<int64_t> ir((size_t)k, 0);
    auto trisolve = [&](const double* rhs, double* out) {
        int ps = h->prof.begin(2, h->stream);
        h->eng->solve_multi(rhs, ld, out, ld, k);
        h->prof.end(ps, h->stream);
    };
    trisolve(h->mB.p, h->mX.p);
    if (!st.iterative_refinement_enable) {
        int bad = 0;
        launch_zero_ints(h->fail.p, 1, h->stream);
        for (int j = 0; j < k; ++j) launch_check_finite(h->mX.p + (size_t)j * N, N, h->fail.p, h->stream);
        HIP_CHECK(hipMemcpyAsync(&bad, h->fail.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (ir_out) std::copy(ir.begin(), ir.end(), ir_out);
        return bad ? HIPKKT_NUMERIC_FAILURE : HIPKKT_OK;
    }
    const SpmvDev A = kkt_spmv(h);
    std::vector<double> hn(2 * (size_t)k), norme((size_t)k), normb((size_t)k);
    std::vector<int> active((size_t)k, 1), mask((size_t)k, 0);
    {
        int pr = h->prof.begin(3, h->stream);
        launch_residual(A, h->Kval.p, h->mB.p, h->mX.p, h->mE.p, h->mpartial.p, h->mnorms.p, h->stream, k, ld);
        launch_norm_inf(h->mB.p, N, h->mpartial.p, h->mnorms.p + k, h->stream, k, ld);
        h->prof.end(pr, h->stream);
        HIP_CHECK(hipMemcpyAsync(hn.data(), h->mnorms.p, 2 * (size_t)k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    for (int j = 0; j < k; ++j) {
        norme[j] = hn[j];
        normb[j] = hn[k + j];
        if (!std::isfinite(norme[j])) return HIPKKT_NUMERIC_FAILURE;
    }
    for (int i = 0; i < st.iterative_refinement_max_iter; ++i) {
        bool any = false;
        for (int j = 0; j < k; ++j) {
            if (active[j] && norme[j] <= st.iterative_refinement_abstol + st.iterative_refinement_reltol * normb[j]) active[j] = 0;
            any = any || active[j];
        }
        if (!any) break;
        trisolve(h->mE.p, h->mC.p);                                            // dx_j = K^{-1} e_j
        launch_axpby_sum(h->mC.p, h->mC.p, h->mX.p, (int64_t)N * k, h->stream); // prospective x_j + dx_j
        int pr = h->prof.begin(3, h->stream);
        launch_residual(A, h->Kval.p, h->mB.p, h->mC.p, h->mE2.p, h->mpartial.p, h->mnorms.p, h->stream, k, ld);
        h->prof.end(pr, h->stream);
        HIP_CHECK(hipMemcpyAsync(hn.data(), h->mnorms.p, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        for (int j = 0; j < k; ++j) {
            mask[j] = 0;
            if (!active[j]) continue;
            ir[j]++;
            h->prof.acc.ir_iterations++;
            if (!std::isfinite(hn[j])) return HIPKKT_NUMERIC_FAILURE;
            const double ratio = norme[j] / hn[j];
            if (ratio < st.iterative_refinement_stop_ratio) {
                if (ratio > 1.0) mask[j] = 1;
                active[j] = 0;
            } else {
                mask[j] = 1;
            }
            if (mask[j]) norme[j] = hn[j];
        }
        HIP_CHECK(hipMemcpyAsync(h->mmask.p, mask.data(), (size_t)k * sizeof(int), hipMemcpyHostToDevice, h->stream));
        launch_accept_columns(h->mX.p, h->mC.p, h->mE.p, h->mE2.p, h->mmask.p, N, k, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));      // `mask` (pageable) must outlive the copy
    }
    h->last_ir = 0;
    for (int j = 0; j < k; ++j) h->last_ir += ir[j];
    if (ir_out) std::copy(ir.begin(), ir.end(), ir_out);
    return HIPKKT_OK;
}

// The same on ROW-major work vectors (N x KP): used when the CSR image has no long rows.  A row of 16 columns is one
// cache line, so the residual's gather of x, the permutations and the accept step all move whole lines.
static int kkt_solve_multi_core_rm(hipkkt_kkt_t h, int k, int KP, int64_t* ir_out)
{
    const hipkkt_settings& st = h->st;
    const int N = (int)h->K.N;
    std::vector<int64_t> ir((size_t)k, 0);
    auto trisolve = [&](const double* rhs, double* out, const double* add = nullptr) {
        int ps = h->prof.begin(2, h->stream);
        h->eng->solve_multi_rm(rhs, out, KP, add);
        h->prof.end(ps, h->stream);
    };
    trisolve(h->mB.p, h->mX.p);
    if (!st.iterative_refinement_enable) {
        int bad = 0;
        launch_zero_ints(h->fail.p, 1, h->stream);
        launch_check_finite(h->mX.p, N * KP, h->fail.p, h->stream);
        HIP_CHECK(hipMemcpyAsync(&bad, h->fail.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (ir_out) std::copy(ir.begin(), ir.end(), ir_out);
        return bad ? HIPKKT_NUMERIC_FAILURE : HIPKKT_OK;
    }
    const SpmvDev A = kkt_spmv(h);
    std::vector<double> hn(2 * (size_t)KP), norme((size_t)k), normb((size_t)k);
    std::vector<int> active((size_t)k, 1), mask((size_t)KP, 0);
    {
        int pr = h->prof.begin(3, h->stream);
        launch_residual_rm(A, h->mB.p, h->mX.p, h->mE.p, h->mpartial.p, h->mnorms.p, h->mnorms.p + KP, KP, h->stream);
        h->prof.end(pr, h->stream);
        HIP_CHECK(hipMemcpyAsync(hn.data(), h->mnorms.p, 2 * (size_t)KP * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    for (int j = 0; j < k; ++j) {
        norme[j] = hn[j];
        normb[j] = hn[KP + j];
        if (!std::isfinite(norme[j])) return HIPKKT_NUMERIC_FAILURE;
    }
    for (int i = 0; i < st.iterative_refinement_max_iter; ++i) {
        bool any = false;
        for (int j = 0; j < k; ++j) {
            if (active[j] && norme[j] <= st.iterative_refinement_abstol + st.iterative_refinement_reltol * normb[j]) active[j] = 0;
            any = any || active[j];
        }
        if (!any) break;
        trisolve(h->mE.p, h->mC.p, h->mX.p);                                      // prospective x_j + dx_j, dx_j = K^{-1} e_j
        int pr = h->prof.begin(3, h->stream);
        launch_residual_rm(A, h->mB.p, h->mC.p, h->mE2.p, h->mpartial.p, h->mnorms.p, nullptr, KP, h->stream);
        h->prof.end(pr, h->stream);
        HIP_CHECK(hipMemcpyAsync(hn.data(), h->mnorms.p, (size_t)KP * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        for (int j = 0; j < k; ++j) {
            mask[j] = 0;
            if (!active[j]) continue;
            ir[j]++;
            h->prof.acc.ir_iterations++;
            if (!std::isfinite(hn[j])) return HIPKKT_NUMERIC_FAILURE;
            const double ratio = norme[j] / hn[j];
            if (ratio < st.iterative_refinement_stop_ratio) {
                if (ratio > 1.0) mask[j] = 1;
                active[j] = 0;
            } else {
                mask[j] = 1;
            }
            if (mask[j]) norme[j] = hn[j];
        }
        bool all_accept = true;
        for (int j = 0; j < k; ++j) all_accept = all_accept && mask[j] != 0;
        if (all_accept) {
            // every column takes its candidate (the usual first round): the buffers change roles, nothing is copied
            // (the padding columns are zero in both)
            std::swap(h->mX.p, h->mC.p);
            std::swap(h->mE.p, h->mE2.p);
            continue;
        }
        HIP_CHECK(hipMemcpyAsync(h->mmask.p, mask.data(), (size_t)KP * sizeof(int), hipMemcpyHostToDevice, h->stream));
        launch_accept_columns_rm(h->mX.p, h->mC.p, h->mE.p, h->mE2.p, h->mmask.p, N, KP, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));      // `mask` (pageable) must outlive the copy
    }
    h->last_ir = 0;
    for (int j = 0; j < k; ++j) h->last_ir += ir[j];
    if (ir_out) std::copy(ir.begin(), ir.end(), ir_out);
    return HIPKKT_OK;
}

static void kkt_multi_unpack(hipkkt_kkt_t h, int k, double* lhsx, double* lhsz, hipMemcpyKind kind)
{
    const size_t n = (size_t)h->K.n, m = (size_t)h->K.m, N = (size_t)h->K.N;
    if (lhsx && n)
        HIP_CHECK(hipMemcpy2DAsync(lhsx, n * sizeof(double), h->mX.p, N * sizeof(double), n * sizeof(double), (size_t)k, kind, h->stream));
    if (lhsz && m)
        HIP_CHECK(hipMemcpy2DAsync(lhsz, m * sizeof(double), h->mX.p + n, N * sizeof(double), m * sizeof(double), (size_t)k, kind, h->stream));
}

// A handful of right-hand sides (2 .. kSmallBatch): the single-column kernels' 2- and 4-column instances, every
// sweep shared by up to four columns -- the tree's dependency chain is paid once per sweep, not once per column (the
// 16-column MFMA path below only pays off from ~a dozen columns on).  Device pointers, column-major, contiguous.
static constexpr int kSmallBatch = 8;
static int kkt_solve_small_batch(hipkkt_kkt_t h, int nrhs, const double* d_rhsx, const double* d_rhsz, double* d_lhsx,
                                 double* d_lhsz, int64_t* ir_iterations, bool may_defer)
{
    const int n = h->K.n, m = h->K.m;
    const size_t N = (size_t)h->K.N;
    const int chunk = h->eng->supports_nr(4) ? 4 : 2;
    kkt_multi_reserve(h, kMaxNR);
    std::swap(h->b.p, h->mB.p);                   // a borrowed right-hand-side buffer: the one set by setrhs! stays as it is
    int rc = HIPKKT_OK;
    int64_t total = 0;
    try {
        for (int j0 = 0; j0 < nrhs && rc == HIPKKT_OK; j0 += chunk) {
            const int k = std::min(chunk, nrhs - j0);
            const int nr = k == 1 ? 1 : (k == 2 ? 2 : 4);
            launch_pack_rhs(h->b.p, d_rhsx + (size_t)j0 * n, d_rhsz + (size_t)j0 * m, n, m, h->K.p, h->stream, k);
            if (k < nr) HIP_CHECK(hipMemsetAsync(h->b.p + (size_t)k * N, 0, (size_t)(nr - k) * N * sizeof(double), h->stream));
            int64_t ir[kMaxNR] = {0, 0, 0, 0};
            rc = kkt_solve_core(h, may_defer, nr, ir);
            if (rc != HIPKKT_OK) break;
            for (int c = 0; c < k; ++c) {
                launch_unpack_lhs(d_lhsx ? d_lhsx + (size_t)(j0 + c) * n : nullptr, d_lhsz ? d_lhsz + (size_t)(j0 + c) * m : nullptr,
                                  h->x.p + (size_t)c * N, n, m, h->stream);
                if (ir_iterations) ir_iterations[j0 + c] = ir[c];
                total += std::max<int64_t>(ir[c], 0);
            }
        }
    } catch (...) { std::swap(h->b.p, h->mB.p); throw; }
    std::swap(h->b.p, h->mB.p);
    h->last_ir = total;
    return rc;
}

int hipkkt_kkt_solve_multi_dev(hipkkt_kkt_t h, int64_t nrhs, const double* d_rhsx, const double* d_rhsz, double* d_lhsx,
                               double* d_lhsz, int64_t* ir_iterations)
{
    return guarded([&]() {
        if (!h || nrhs < 0 || nrhs > 65535 || (nrhs > 0 && ((h->K.n && !d_rhsx) || (h->K.m && !d_rhsz))))
            throw ArgError("hipkkt_kkt_solve_multi_dev: bad argument");
        if (nrhs == 0) return HIPKKT_OK;
        HIP_CHECK(hipSetDevice(h->device));
        if (nrhs == 1) {
            // one column: the single-column path (persistent top kernel, no padding to 16 columns), on a borrowed
            // right-hand-side buffer so that the one set by hipkkt_kkt_setrhs stays as it is
            kkt_multi_reserve(h, 1);
            std::swap(h->b.p, h->mB.p);
            launch_pack_rhs(h->b.p, d_rhsx, d_rhsz, h->K.n, h->K.m, h->K.p, h->stream);
            int rc;
            try { rc = kkt_solve_core(h, true); } catch (...) { std::swap(h->b.p, h->mB.p); throw; }
            std::swap(h->b.p, h->mB.p);
            if (ir_iterations) ir_iterations[0] = h->last_ir;      // (-1 in deferred-status mode, as for 2..8 columns)
            if (rc != HIPKKT_OK) return rc;
            launch_unpack_lhs(d_lhsx, d_lhsz, h->cur_x, h->K.n, h->K.m, h->stream);
            return HIPKKT_OK;
        }
        if (nrhs <= kSmallBatch && h->eng->supports_nr(2))
            return kkt_solve_small_batch(h, (int)nrhs, d_rhsx, d_rhsz, d_lhsx, d_lhsz, ir_iterations, true);
        kkt_multi_reserve(h, (size_t)nrhs);
        if (h->nlong == 0) {
            const int KP = ((int)nrhs + 15) & ~15;
            launch_pack_rhs_rm(h->mB.p, d_rhsx, d_rhsz, h->K.n, h->K.m, h->K.p, (int)nrhs, KP, h->stream);
            int rc = kkt_solve_multi_core_rm(h, (int)nrhs, KP, ir_iterations);
            if (rc != HIPKKT_OK) return rc;
            launch_unpack_lhs_rm(d_lhsx, d_lhsz, h->mX.p, h->K.n, h->K.m, (int)nrhs, KP, h->stream);
            return HIPKKT_OK;
        }
        launch_pack_rhs(h->mB.p, d_rhsx, d_rhsz, h->K.n, h->K.m, h->K.p, h->stream, (int)nrhs);
        int rc = kkt_solve_multi_core(h, (int)nrhs, ir_iterations);
        if (rc != HIPKKT_OK) return rc;
        kkt_multi_unpack(h, (int)nrhs, d_lhsx, d_lhsz, hipMemcpyDeviceToDevice);
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_solve_multi(hipkkt_kkt_t h, int64_t nrhs, const double* rhsx, const double* rhsz, double* lhsx,
                           double* lhsz, int64_t* ir_iterations)
{
    return guarded([&]() {
        if (!h || nrhs < 0 || nrhs > 65535 || (nrhs > 0 && ((h->K.n && !rhsx) || (h->K.m && !rhsz))))
            throw ArgError("hipkkt_kkt_solve_multi: bad argument");
        if (nrhs == 0) return HIPKKT_OK;
        HIP_CHECK(hipSetDevice(h->device));
        kkt_multi_reserve(h, (size_t)nrhs);
        const size_t n = (size_t)h->K.n, m = (size_t)h->K.m, k = (size_t)nrhs;
        // stage [rhsx | rhsz] in the candidate buffer (N k >= (n + m) k doubles, not needed before the first round)
        double* sx = h->mC.p;
        double* sz = h->mC.p + n * k;
        if (n) HIP_CHECK(hipMemcpyAsync(sx, rhsx, n * k * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (m) HIP_CHECK(hipMemcpyAsync(sz, rhsz, m * k * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (nrhs >= 2 && nrhs <= kSmallBatch && h->eng->supports_nr(2)) {
            // staged in the candidate buffer of the many-column path (unused by this one); solutions come back through it too
            double* ox = h->mE.p;
            double* oz = h->mE.p + n * k;
            int rc = kkt_solve_small_batch(h, (int)nrhs, sx, sz, lhsx ? ox : nullptr, lhsz ? oz : nullptr, ir_iterations, false);
            if (rc != HIPKKT_OK) return rc;
            if (lhsx && n) HIP_CHECK(hipMemcpyAsync(lhsx, ox, n * k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (lhsz && m) HIP_CHECK(hipMemcpyAsync(lhsz, oz, m * k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_CHECK(hipStreamSynchronize(h->stream));
            return HIPKKT_OK;
        }
        if (h->nlong == 0) {
            const int KP = ((int)nrhs + 15) & ~15;
            launch_pack_rhs_rm(h->mB.p, sx, sz, h->K.n, h->K.m, h->K.p, (int)nrhs, KP, h->stream);
            int rc = kkt_solve_multi_core_rm(h, (int)nrhs, KP, ir_iterations);
            if (rc != HIPKKT_OK) return rc;
            // column-major staging in the candidate buffer again, then to the host
            double* ox = h->mC.p;
            double* oz = h->mC.p + n * k;
            launch_unpack_lhs_rm(lhsx ? ox : nullptr, lhsz ? oz : nullptr, h->mX.p, h->K.n, h->K.m, (int)nrhs, KP, h->stream);
            if (lhsx && n) HIP_CHECK(hipMemcpyAsync(lhsx, ox, n * k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (lhsz && m) HIP_CHECK(hipMemcpyAsync(lhsz, oz, m * k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_CHECK(hipStreamSynchronize(h->stream));
            return HIPKKT_OK;
        }
        launch_pack_rhs(h->mB.p, sx, sz, h->K.n, h->K.m, h->K.p, h->stream, (int)nrhs);
        int rc = kkt_solve_multi_core(h, (int)nrhs, ir_iterations);
        if (rc != HIPKKT_OK) return rc;
        kkt_multi_unpack(h, (int)nrhs, lhsx, lhsz, hipMemcpyDeviceToHost);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

// ------------------------------------------------------------------------------------------------
//  Level C: the reduced-system layer (DefaultKKTSystem, /root/reference/src/kktsystem.jl:21-215) with every
//  vector resident in HBM -- right-hand-side construction and the recovery of (dtau, dx, dz, ds, dkappa)
//  around the solves run on the device, so nothing but two scalars per solve crosses PCIe.
// ------------------------------------------------------------------------------------------------
static SpmvDev sys_spmv(hipkkt_kkt_t h) { return kkt_spmv(h); }

int hipkkt_kkt_system_init(hipkkt_kkt_t h, const double* q, const double* b)
{
    return guarded([&]() {
        if (!h || (h->K.n && !q) || (h->K.m && !b)) throw ArgError("hipkkt_kkt_system_init: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        const size_t n = (size_t)h->K.n, m = (size_t)h->K.m;
        h->sq.alloc(n); h->snegq.alloc(n); h->sb.alloc(m);
        h->sx2.alloc(n); h->sworkx.alloc(n); h->spa.alloc(n); h->spb.alloc(n); h->spc.alloc(n);
        h->sz2.alloc(m); h->sworkz.alloc(m); h->sconic.alloc(m);
        h->sys_partial.alloc(8 * 64); h->sys_dots.alloc(8); h->sys_cached.alloc(4); h->sys_in.alloc(4); 
        if (n) HIP_CHECK(hipMemcpyAsync(h->sq.p, q, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (m) HIP_CHECK(hipMemcpyAsync(h->sb.p, b, m * sizeof(double), hipMemcpyHostToDevice, h->stream));
        launch_neg_copy(h->snegq.p, h->sq.p, (int)n, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->sys_ready = true;
        return HIPKKT_OK;
    });
}

// setrhs! + solve! with device vectors; either output may be null
static int sys_solve_into(hipkkt_kkt_t h, const double* rx, const double* rz, double* outx, double* outz)
{
    launch_pack_rhs(h->b.p, rx, rz, h->K.n, h->K.m, h->K.p, h->stream);
    int rc = kkt_solve_core(h);
    if (rc != HIPKKT_OK) return rc;
    launch_unpack_lhs(outx, outz, h->cur_x, h->K.n, h->K.m, h->stream);
    return HIPKKT_OK;
}

static void sys_cache_constant_terms(hipkkt_kkt_t h);
static int sys_flush_update_status(hipkkt_kkt_t h);
static int sys_flush_startup(hipkkt_kkt_t h);
static int sys_constant_rhs(hipkkt_kkt_t h)
{
    // _kkt_solve_constant_rhs! (kktsystem.jl:80-92): (x2, z2) = K \ (-q, b)
    int rc = sys_solve_into(h, h->snegq.p, h->sb.p, h->sx2.p, h->sz2.p);
    if (rc != HIPKKT_OK) return rc;
    sys_cache_constant_terms(h);
    return HIPKKT_OK;
}

int hipkkt_kkt_system_solve_constant_rhs(hipkkt_kkt_t h)
{
    return guarded([&]() {
        if (!h || !h->sys_ready) throw ArgError("hipkkt_kkt_system_*: call hipkkt_kkt_system_init first");
        HIP_CHECK(hipSetDevice(h->device));
        h->sys_const_pending = false;
        const int rc = sys_flush_update_status(h);      // (an enqueued-only kkt_update! before it: its status is due here)
        if (rc != HIPKKT_OK) return rc;
        return sys_constant_rhs(h);
    });
}

int hipkkt_kkt_system_set_lazy(hipkkt_kkt_t h, int lazy)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        if (!lazy && h->sys_const_pending) {           // leaving the mode with a solve still due: run it now
            if (!h->sys_ready) throw ArgError("hipkkt_kkt_system_*: call hipkkt_kkt_system_init first");
            HIP_CHECK(hipSetDevice(h->device));
            h->sys_const_pending = false;
            h->sys_lazy = false;
            const int rc = sys_flush_update_status(h);
            if (rc != HIPKKT_OK) return rc;
            return sys_constant_rhs(h);
        }
        if (!lazy && h->sys_ready) {                   // (an enqueued-only update's record does not outlive the mode)
            HIP_CHECK(hipSetDevice(h->device));
            const int rc = sys_flush_update_status(h);
            if (rc != HIPKKT_OK) { h->sys_lazy = false; return rc; }
        }
        h->sys_lazy = lazy != 0;
        return HIPKKT_OK;
    });
}

// the constant-RHS part of kkt_update! (kktsystem.jl:74-77): at once, or noted for the affine kkt_solve! (lazy mode)
static int sys_after_update(hipkkt_kkt_t h)
{
    if (!h->sys_ready) throw ArgError("hipkkt_kkt_system_*: call hipkkt_kkt_system_init first");
    if (h->deferred) throw ArgError("hipkkt_kkt_system_*: level C reads its scalars back (deferred status is for level B)");
    if (h->sys_lazy) { h->sys_const_pending = true; return HIPKKT_OK; }
    h->sys_const_pending = false;
    return sys_constant_rhs(h);
}

int hipkkt_kkt_system_update(hipkkt_kkt_t h, const double* d_s, const double* d_z)
{
    // kkt_update! (kktsystem.jl:62-78)
    if (h && h->deferred) {
        g_last_error = "hipkkt_kkt_system_*: level C reads its scalars back (deferred status is for level B)";
        return HIPKKT_ERR_ARG;         // (before anything is enqueued: a deferred factorisation's status would be left unread)
    }
    // Lazy mode: the update is only ENQUEUED (its status -- cone points, pivots, the overlap mode's waits -- joins the
    // sticky record on the device) and is read back together with the affine kkt_solve!'s own result: one host round
    // trip for kkt_update! + kkt_solve!(:affine) instead of three.  A failed factorisation therefore surfaces at the
    // affine kkt_solve! -- `is_kkt_solve_success = kkt_update!(...)` and `is_kkt_solve_success && kkt_solve!(...)`
    // (solver.jl:279-295) reach the same branch either way.
    const bool enqueue_only = h && h->sys_lazy && h->sys_ready;
    int rc = kkt_update_from_sz_dev_impl(h, d_s, d_z, enqueue_only);
    if (rc != HIPKKT_OK) return rc;                    // "bail if the factorization has failed" (:71)
    if (enqueue_only) h->sys_update_unread = true;
    return guarded([&]() { return sys_after_update(h); });
}

int hipkkt_kkt_system_solve_initial_point(hipkkt_kkt_t h, double* d_x, double* d_s, double* d_z)
{
    return guarded([&]() {
        if (!h || !h->sys_ready) throw ArgError("hipkkt_kkt_system_*: call hipkkt_kkt_system_init first");
        if ((h->K.n && !d_x) || (h->K.m && (!d_s || !d_z))) throw ArgError("hipkkt_kkt_system_solve_initial_point: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        const int n = h->K.n, m = h->K.m;
        int rc = sys_flush_startup(h);
        if (rc != HIPKKT_OK) return rc;
        if (h->mapP.n == 0) {
            // LP initialisation (kktsystem.jl:107-128): [0; b] -> (x, -s), then [-q; 0] -> z
            if (n) HIP_CHECK(hipMemsetAsync(h->sworkx.p, 0, (size_t)n * sizeof(double), h->stream));
            rc = sys_solve_into(h, h->sworkx.p, h->sb.p, d_x, h->sworkz.p);
            if (rc != HIPKKT_OK) return rc;
            launch_neg_copy(d_s, h->sworkz.p, m, h->stream);
            if (m) HIP_CHECK(hipMemsetAsync(h->sworkz.p, 0, (size_t)m * sizeof(double), h->stream));
            rc = sys_solve_into(h, h->snegq.p, h->sworkz.p, nullptr, d_z);
            if (rc != HIPKKT_OK) return rc;
        } else {
            // QP initialisation (:129-137): [-q; b] -> (x, z), s = -z
            rc = sys_solve_into(h, h->snegq.p, h->sb.p, d_x, d_z);
            if (rc != HIPKKT_OK) return rc;
            launch_neg_copy(d_s, d_z, m, h->stream);
        }
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

// the part of kkt_solve! behind the solve for (x1, z1) (kktsystem.jl:175-212): dtau, (dx, dz), ds, dkappa
// with_const: (x2, z2) is new as well (it came out of the same 2-column solve): its terms of tau_den are formed here too
// addend: Delta_s constant term (sconic, or variables.s itself for the affine step)
// (x1, z1): the solve's solution where the sweeps left it (h->x: no copy); it is consumed before the next solve
static void sys_finish_step(hipkkt_kkt_t h, const double* x1, const double* z1, double* d_lhs_x, double* d_lhs_s, double* d_lhs_z,
                            double* lhs_tau_kappa, double rhs_tau, double rhs_kappa, const double* d_var_x, double var_tau,
                            double var_kappa, bool with_const = false, const double* addend = nullptr, const Publish& pub = Publish{})
{
    const int n = h->K.n, m = h->K.m;
    hipStream_t st = h->stream;
    // (x2, z2): the handle's copy -- or, fresh out of this call's 2-column solve, column 0 where the sweeps left it; the copy
    // that the iteration's later solves read is then written by the step kernel on its way
    const double* x2 = with_const ? h->x.p : h->sx2.p;
    const double* z2 = with_const ? h->x.p + n : h->sz2.p;
    // P x1 and P (xi - x2), xi = x / tau, in one pass (xi - x2 kept in workx for its dot product)
    launch_P_spmv2(sys_spmv(h), h->Kval.p, x1, d_var_x, x2, var_tau, h->spa.p, h->spb.p, h->sworkx.p,
                   with_const ? h->spc.p : nullptr, n, st);
    DotPairs P{};
    P.npairs = with_const ? 7 : 4;
    if (with_const) {
        P.a[4] = h->sq.p; P.b[4] = x2; P.len[4] = n;
        P.a[5] = h->sb.p; P.b[5] = z2; P.len[5] = m;
        P.a[6] = x2; P.b[6] = h->spc.p; P.len[6] = n;
    }
    P.a[0] = h->sq.p; P.b[0] = x1; P.len[0] = n;
    P.a[1] = h->sb.p; P.b[1] = z1; P.len[1] = m;
    P.a[2] = d_var_x; P.b[2] = h->spa.p; P.len[2] = n;
    P.a[3] = h->sworkx.p; P.b[3] = h->spb.p; P.len[3] = n;
    // ... the scalars (:185-196, :206) and (dx, dz) = (x1, z1) + dtau (x2, z2)   (:200-203)
    launch_dots_sys_step(P, h->sys_partial.p, h->sys_cached.p, rhs_tau, rhs_kappa, var_tau, var_kappa, h->sys_out_dev, d_lhs_x, d_lhs_z,
                         x1, z1, x2, z2, n, m, st, with_const ? h->sx2.p : nullptr, with_const ? h->sz2.p : nullptr);
    // ds = -(Hs dz + const)                                               (:206-212)
    launch_mul_Hs(h->cone_dev(), h->cone_state(), d_lhs_s, d_lhs_z, m, st, addend ? addend : h->sconic.p, pub);
    if (!lhs_tau_kappa) return;                      // (the caller reads sys_out back with its own status record)
    // (dtau, dkappa) through the handle's pinned block: a copy into pageable memory would be staged
    HIP_CHECK(hipMemcpyAsync(h->pin->h + 48, h->sys_out_dev, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    lhs_tau_kappa[0] = h->pin->h[48];
    lhs_tau_kappa[1] = h->pin->h[49];
}
// an enqueued-only kkt_update! whose status nobody has read yet (lazy mode), before a call that reads back on its own
static int sys_flush_update_status(hipkkt_kkt_t h)
{
    if (!h->sys_update_unread) return HIPKKT_OK;
    h->sys_update_unread = false;
    HIP_CHECK(hipMemcpyAsync(h->pin->h + 40, h->ir_sticky, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    launch_zero_ints((int*)h->ir_sticky, 16, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    bool gave_up = false;
    int rc = kkt_eval_sticky(h, h->pin->h + 40, &gave_up);
    if (rc == HIPKKT_REFINEMENT_INCOMPLETE) rc = gave_up ? kkt_update_device(h) : HIPKKT_OK;   // a bounded wait gave up: level by level, synchronously
    return rc;
}
// ... before the calls of the start-up sequence (solver.jl:389-393: kkt_update!, then kkt_solve_initial_point!, the
// update's Bool ignored): the record is read so that a give-up is repaired and no stale word reaches the first
// iteration; a numeric failure of that update is the reference's to ignore -- the solve that follows reports its own
static int sys_flush_startup(hipkkt_kkt_t h)
{
    const int rc = sys_flush_update_status(h);
    return rc == HIPKKT_NUMERIC_FAILURE ? HIPKKT_OK : rc;
}
// the x2-only terms of tau_den (kktsystem.jl:194-196) are the same for both solves of the iteration
static void sys_cache_constant_terms(hipkkt_kkt_t h)
{
    launch_P_spmv(sys_spmv(h), h->Kval.p, h->sx2.p, h->spa.p, h->K.n, h->stream);
    DotPairs P{};
    P.npairs = 3;
    P.a[0] = h->sq.p; P.b[0] = h->sx2.p; P.len[0] = h->K.n;
    P.a[1] = h->sb.p; P.b[1] = h->sz2.p; P.len[1] = h->K.m;
    P.a[2] = h->sx2.p; P.b[2] = h->spa.p; P.len[2] = h->K.n;
    launch_dots(P, h->sys_partial.p, h->sys_cached.p, h->stream);
}

// kkt_solve! (kktsystem.jl:145-215).  With the constant-RHS solve of the preceding kkt_update! still due (lazy mode)
// an affine step sends both right-hand sides through the sweeps together.
static int sys_solve_step(hipkkt_kkt_t h, double* d_lhs_x, double* d_lhs_s, double* d_lhs_z, double* lhs_tau_kappa,
                          const double* d_rhs_x, const double* d_rhs_s, const double* d_rhs_z, double rhs_tau,
                          double rhs_kappa, const double* d_var_x, const double* d_var_s, const double* d_var_z,
                          double var_tau, double var_kappa, int steptype)
{
    if (!h->sys_ready) throw ArgError("hipkkt_kkt_system_*: call hipkkt_kkt_system_init first");
    const int n = h->K.n, m = h->K.m;
    const bool affine = steptype == 0;
    // (the affine step does not read rhs.s: kktsystem.jl:157-158)
    if ((n && (!d_lhs_x || !d_rhs_x || !d_var_x)) || (m && (!d_lhs_s || !d_lhs_z || !d_rhs_z || !d_var_s || !d_var_z)) ||
        (m && !affine && !d_rhs_s) || !lhs_tau_kappa || (steptype != 0 && steptype != 1))
        throw ArgError("hipkkt_kkt_system_solve: bad argument");
    if (!h->scaling_valid) throw ArgError("hipkkt_kkt_system_solve: needs the cone scaling of hipkkt_kkt_system_update");
    if (h->deferred) throw ArgError("hipkkt_kkt_system_*: level C reads its scalars back (deferred status is for level B)");
    HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const bool pair = h->sys_const_pending && affine && h->eng->supports_nr(2);
    if (h->sys_const_pending && !pair) {
        // (x2, z2) is due and cannot ride with this solve: by itself first, as kkt_update! would have done it
        h->sys_const_pending = false;
        int rc = sys_flush_update_status(h);
        if (rc == HIPKKT_OK) rc = sys_constant_rhs(h);
        if (rc != HIPKKT_OK) return rc;
    }
    h->sys_const_pending = false;
    const size_t N = (size_t)h->K.N;
    // everything of this call, enqueued; `defer`: the solve's refinement decisions stay on the device, nothing is read back here
    auto run = [&](bool defer) -> int {
        // Delta_s constant term and the z part of the right-hand side (kktsystem.jl:150-166).  Affine step: the constant
        // term is variables.s itself (:157-158), so the right-hand side is packed straight from (rhs.x, s - rhs.z)
        if (!affine && !launch_sys_offset(h->cone_dev(), h->cone_state(), h->sconic.p, h->sworkz.p, d_rhs_s, d_var_z, d_rhs_z,
                                          m, false, st))
            throw ArgError("hipkkt_kkt_system_solve: unsupported cone kind");
        if (pair) {
            // _kkt_solve_constant_rhs! (:80-92) and this solve (:170-173) as ONE 2-column solve: column 0 = (-q, b),
            // column 1 = (rhs.x, s - rhs.z); each column is refined by the reference's rule on its own
            launch_pack_rhs_affine(h->b.p, h->snegq.p, h->sb.p, d_rhs_x, d_var_s, d_rhs_z, n, m, h->K.p, 2, st);
            int rc = kkt_solve_core(h, false, 2, nullptr, defer);
            if (rc != HIPKKT_OK) return rc;
            // ((x2, z2) = column 0 outlives this solve: sys_finish_step copies it on its way)
        } else {
            // (x1, z1) = K \ (rhs.x, const - rhs.z)                              (:170-173)
            if (affine) launch_pack_rhs_affine(h->b.p, nullptr, nullptr, d_rhs_x, d_var_s, d_rhs_z, n, m, h->K.p, 1, st);
            else launch_pack_rhs(h->b.p, d_rhs_x, h->sworkz.p, n, m, h->K.p, st);
            int rc = kkt_solve_core(h, false, 1, nullptr, defer);
            if (rc != HIPKKT_OK) return rc;
        }
        const double* x1 = h->x.p + (pair ? N : 0);
        // (deferred: the status record and (dtau, dkappa) behind it reach the host with the call's last kernel)
        sys_finish_step(h, x1, x1 + n, d_lhs_x, d_lhs_s, d_lhs_z, defer ? nullptr : lhs_tau_kappa, rhs_tau, rhs_kappa, d_var_x, var_tau, var_kappa,
                        pair, affine ? d_var_s : nullptr,
                        (defer && h->publish_ok) ? Publish{h->pin->h + 40, h->ir_sticky, 10, 8, (double)++h->publish_seq} : Publish{});
        return HIPKKT_OK;
    };
    if (h->sys_lazy && h->st.iterative_refinement_enable) {
        // Lazy mode: ONE read-back per call -- the sticky status record (this call's solve, and the kkt_update! before it
        // if that was enqueued only) together with (dtau, dkappa).  If the record says that the reference's refinement loop
        // would have gone on, or that a bounded wait gave up, the call is repeated with the synchronous sequence.
        const bool update_unread = h->sys_update_unread;
        h->sys_update_unread = false;
        int rc = run(true);
        if (rc != HIPKKT_OK) return rc;
        if (h->publish_ok) {
            HIP_CHECK(hipStreamSynchronize(st));       // (pin->h[40..49]: the record, then dtau, dkappa -- published by run's last kernel)
            if (h->pin->h[50] != (double)h->publish_seq || knobs().test_publish_fail) {
                // The device's stores to the page-locked block did not arrive (never seen; a mapping this code has not been
                // run on): the record -- zeroed on the device by the kernel that published it -- is lost for this call, so
                // the call is repeated with the synchronous sequence, its update included; copies from now on.
                h->publish_ok = false;
                if (knobs().verbose) std::fprintf(stderr, "[hipkkt] status record not published to host memory: copying it from now on\n");
                if (update_unread) {
                    rc = kkt_update_device(h);
                    if (rc != HIPKKT_OK) return rc;
                }
                return run(false);
            }
        } else {
            HIP_CHECK(hipMemcpyAsync(h->pin->h + 40, h->ir_sticky, 10 * sizeof(double), hipMemcpyDeviceToHost, st));   // (the record, then dtau, dkappa)
            launch_zero_ints((int*)h->ir_sticky, 16, st);
            HIP_CHECK(hipStreamSynchronize(st));
        }
        bool gave_up = false;
        rc = kkt_eval_sticky(h, h->pin->h + 40, &gave_up, true);
        h->last_ir = (int64_t)h->pin->h[43];           // this call's refinement rounds, summed over its columns (as the synchronous path reports)
        if (rc == HIPKKT_OK) {
            lhs_tau_kappa[0] = h->pin->h[48];
            lhs_tau_kappa[1] = h->pin->h[49];
            return HIPKKT_OK;
        }
        if (rc != HIPKKT_REFINEMENT_INCOMPLETE) return rc;
        if (update_unread && gave_up) {                // (the factorisation itself may be void: a wait of the overlap mode gave up;
            rc = kkt_update_device(h);                 //  a solve that merely wanted another refinement round keeps its factor)
            if (rc != HIPKKT_OK) return rc;
        }
    } else {
        int rc = sys_flush_update_status(h);
        if (rc != HIPKKT_OK) return rc;
    }
    return run(false);
}

int hipkkt_kkt_system_solve(hipkkt_kkt_t h, double* d_lhs_x, double* d_lhs_s, double* d_lhs_z, double* lhs_tau_kappa,
                            const double* d_rhs_x, const double* d_rhs_s, const double* d_rhs_z, double rhs_tau,
                            double rhs_kappa, const double* d_var_x, const double* d_var_s, const double* d_var_z,
                            double var_tau, double var_kappa, int steptype)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        return sys_solve_step(h, d_lhs_x, d_lhs_s, d_lhs_z, lhs_tau_kappa, d_rhs_x, d_rhs_s, d_rhs_z, rhs_tau, rhs_kappa,
                              d_var_x, d_var_s, d_var_z, var_tau, var_kappa, steptype);
    });
}

int hipkkt_kkt_system_update_and_solve_affine(hipkkt_kkt_t h, double* d_lhs_x, double* d_lhs_s, double* d_lhs_z,
                                              double* lhs_tau_kappa, const double* d_rhs_x, const double* d_rhs_z,
                                              double rhs_tau, double rhs_kappa, const double* d_var_x, const double* d_var_s,
                                              const double* d_var_z, double var_tau, double var_kappa)
{
    // kkt_update! in lazy mode followed by the affine kkt_solve!: exactly what the two separate calls do
    if (!h) { g_last_error = "null handle"; return HIPKKT_ERR_ARG; }
    const bool was_lazy = h->sys_lazy;
    h->sys_lazy = true;
    int rc = hipkkt_kkt_system_update(h, d_var_s, d_var_z);
    h->sys_lazy = was_lazy;
    if (rc != HIPKKT_OK) return rc;
    return guarded([&]() {
        return sys_solve_step(h, d_lhs_x, d_lhs_s, d_lhs_z, lhs_tau_kappa, d_rhs_x, nullptr, d_rhs_z, rhs_tau, rhs_kappa,
                              d_var_x, d_var_s, d_var_z, var_tau, var_kappa, 0);
    });
}

// kkt_update! from the caller's cone objects (the Julia glue): kktsolver_update!'s data plus the NT scaling that
// kkt_solve!'s right-hand-side construction and step recovery read from the same cones
int hipkkt_kkt_system_update_cones(hipkkt_kkt_t h, const double* Hs, const double* soc_u, const double* soc_v,
                                   const double* soc_eta2, const double* w, const double* eta, const double* lambda,
                                   const double* psd_R, const double* psd_Rinv)
{
    if (h && h->deferred) {
        g_last_error = "hipkkt_kkt_system_*: level C reads its scalars back (deferred status is for level B)";
        return HIPKKT_ERR_ARG;
    }
    int rc = hipkkt_kkt_update_cones(h, Hs, soc_u, soc_v, soc_eta2);
    if (rc != HIPKKT_OK) return rc;
    return guarded([&]() {
        const size_t m = (size_t)h->K.m, nc = h->K.cones.size();
        if ((m && (!w || !lambda)) || (h->nsoc > 0 && !eta) || (h->npsd > 0 && (!psd_R || !psd_Rinv)))
            throw ArgError("hipkkt_kkt_system_update_cones: missing scaling data");
        if (h->psd_too_big) throw ArgError("hipkkt_kkt_system_*: PSD cones with side > 48 are not covered by level C");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        if (m) {
            HIP_CHECK(hipMemcpyAsync(h->w.p, w, m * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(h->lam.p, lambda, m * sizeof(double), hipMemcpyHostToDevice, st));
        }
        if (eta && nc) HIP_CHECK(hipMemcpyAsync(h->eta.p, eta, nc * sizeof(double), hipMemcpyHostToDevice, st));
        if (h->npsd > 0) {
            HIP_CHECK(hipMemcpyAsync(h->psdR.p, psd_R, h->psdR.n * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(h->psdRinv.p, psd_Rinv, h->psdRinv.n * sizeof(double), hipMemcpyHostToDevice, st));
            launch_psd_A_from_R(h->cone_dev(), h->cone_state(), st);       // A = R R' (mul_Hs!, coneops_psdtrianglecone.jl:164-187)
        }
        HIP_CHECK(hipStreamSynchronize(st));           // the caller's arrays may go away
        h->scaling_valid = true;
        return sys_after_update(h);
    });
}

int hipkkt_kkt_system_update_scaling(hipkkt_kkt_t h, const double* w, const double* eta, const double* lambda,
                                     const double* psd_R, const double* psd_Rinv)
{
    if (h && h->deferred) {
        g_last_error = "hipkkt_kkt_system_*: level C reads its scalars back (deferred status is for level B)";
        return HIPKKT_ERR_ARG;
    }
    bool enqueue_only = false;
    int rc = guarded([&]() {
        if (!h) throw ArgError("null handle");
        const size_t m = (size_t)h->K.m, nc = h->K.cones.size();
        if ((m && (!w || !lambda)) || (h->nsoc > 0 && !eta) || (h->npsd > 0 && (!psd_R || !psd_Rinv)))
            throw ArgError("hipkkt_kkt_system_update_scaling: missing scaling data");
        if (h->psd_too_big) throw ArgError("hipkkt_kkt_system_*: PSD cones with side > 48 are not covered by level C");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        if (m) {
            HIP_CHECK(hipMemcpyAsync(h->w.p, w, m * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(h->lam.p, lambda, m * sizeof(double), hipMemcpyHostToDevice, st));
        }
        if (eta && nc) HIP_CHECK(hipMemcpyAsync(h->eta.p, eta, nc * sizeof(double), hipMemcpyHostToDevice, st));
        if (h->npsd > 0) {
            HIP_CHECK(hipMemcpyAsync(h->psdR.p, psd_R, h->psdR.n * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(h->psdRinv.p, psd_Rinv, h->psdRinv.n * sizeof(double), hipMemcpyHostToDevice, st));
        }
        // the caller's arrays are free again once these copies have run; the kernels behind them need not have
        h->host_upload_mark(st);
        launch_zero_ints(h->fail.p, 1, st);
        int pu = h->prof.begin(0, st);
        launch_cone_from_scaling(h->cone_dev(), h->cone_state(), h->K.m, st);      // Hs, u, v, eta^2 (and R R') on the device
        h->prof.end(pu, st);
        h->scaling_valid = true;
        enqueue_only = h->sys_lazy && h->sys_ready;
        const int r = kkt_update_device(h, enqueue_only);
        h->host_upload_wait();
        return r;
    });
    if (rc != HIPKKT_OK) return rc;
    if (enqueue_only) h->sys_update_unread = true;
    return guarded([&]() { return sys_after_update(h); });
}

int hipkkt_selftest_handover(int variant, int pairs, int words, int rounds, int device, int64_t out[2])
{
    return guarded([&]() {
        if (variant < 0 || variant > 2 || pairs <= 0 || pairs > 96 || words <= 0 || rounds <= 0 || !out)
            throw ArgError("hipkkt_selftest_handover: bad argument");     // (<= 96 pairs: every workgroup resident, whatever else runs)
        HIP_CHECK(hipSetDevice(device < 0 ? 0 : device));
        DBuf<double> payload;
        DBuf<int> words_i;
        DBuf<int64_t> counts;
        payload.alloc((size_t)pairs * words);
        words_i.alloc((size_t)2 * pairs);
        counts.alloc(2);
        HIP_CHECK(hipMemset(payload.p, 0, payload.n * sizeof(double)));
        HIP_CHECK(hipMemset(words_i.p, 0, words_i.n * sizeof(int)));
        HIP_CHECK(hipMemset(counts.p, 0, 2 * sizeof(int64_t)));
        launch_handover_litmus(variant, payload.p, words_i.p, words_i.p + pairs, pairs, words, rounds,
                               reinterpret_cast<unsigned long long*>(counts.p), reinterpret_cast<unsigned long long*>(counts.p) + 1, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out, counts.p, 2 * sizeof(int64_t), hipMemcpyDeviceToHost));
        return HIPKKT_OK;
    });
}

int hipkkt_host_register(void* ptr, int64_t bytes)
{
    return guarded([&]() {
        if (!ptr || bytes <= 0) throw ArgError("hipkkt_host_register: bad argument");
        HIP_CHECK(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
        return HIPKKT_OK;
    });
}
int hipkkt_host_unregister(void* ptr)
{
    return guarded([&]() {
        if (!ptr) throw ArgError("hipkkt_host_unregister: bad argument");
        HIP_CHECK(hipHostUnregister(ptr));
        return HIPKKT_OK;
    });
}

// ---- host-vector variants: the iterate, right-hand side and step live in host memory (DefaultVariables)
static double* sys_host_stage(hipkkt_kkt_t h)
{
    const size_t len = (size_t)h->K.n + 2 * (size_t)h->K.m;
    if (h->hst.n < 3 * len) { h->hst.alloc(3 * len); h->host_vars_valid = false; }
    return h->hst.p;
}

int hipkkt_kkt_system_update_host(hipkkt_kkt_t h, const double* s, const double* z)
{
    int rc = guarded([&]() {
        if (!h || (h->K.m && (!s || !z))) throw ArgError("hipkkt_kkt_system_update_host: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        const size_t bytes = (size_t)h->K.m * sizeof(double);
        if (bytes) {
            HIP_CHECK(hipMemcpyAsync(h->sbuf.p, s, bytes, hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(h->zbuf.p, z, bytes, hipMemcpyHostToDevice, h->stream));
        }
        return HIPKKT_OK;
    });
    if (rc != HIPKKT_OK) return rc;
    return hipkkt_kkt_system_update(h, h->sbuf.p, h->zbuf.p);
}

int hipkkt_kkt_system_solve_initial_point_host(hipkkt_kkt_t h, double* x, double* s, double* z)
{
    if (!h) { g_last_error = "null handle"; return HIPKKT_ERR_ARG; }
    const size_t n = (size_t)h->K.n, m = (size_t)h->K.m;
    double* d = nullptr;
    int rc = guarded([&]() {
        if ((n && !x) || (m && (!s || !z))) throw ArgError("hipkkt_kkt_system_solve_initial_point_host: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        d = sys_host_stage(h);
        return HIPKKT_OK;
    });
    if (rc != HIPKKT_OK) return rc;
    rc = hipkkt_kkt_system_solve_initial_point(h, d, d + n, d + n + m);
    if (rc != HIPKKT_OK) return rc;
    return guarded([&]() {
        if (n) HIP_CHECK(hipMemcpyAsync(x, d, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (m) HIP_CHECK(hipMemcpyAsync(s, d + n, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (m) HIP_CHECK(hipMemcpyAsync(z, d + n + m, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_system_solve_host(hipkkt_kkt_t h, double* lhs_x, double* lhs_s, double* lhs_z, double* lhs_tau_kappa,
                                 const double* rhs_x, const double* rhs_s, const double* rhs_z, double rhs_tau,
                                 double rhs_kappa, const double* var_x, const double* var_s, const double* var_z,
                                 double var_tau, double var_kappa, int steptype)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        const size_t n = (size_t)h->K.n, m = (size_t)h->K.m, len = n + 2 * m;
        const bool affine = steptype == 0;
        // (var_x = var_s = var_z = NULL: the variables uploaded by the previous call -- the combined step's are the affine step's)
        const bool reuse_vars = !var_x && !var_s && !var_z && h->host_vars_valid;
        if ((n && (!lhs_x || !rhs_x || (!var_x && !reuse_vars))) || (m && (!lhs_s || !lhs_z || !rhs_z || ((!var_s || !var_z) && !reuse_vars))) ||
            (m && !affine && !rhs_s) || !lhs_tau_kappa)
            throw ArgError("hipkkt_kkt_system_solve_host: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        double* d = sys_host_stage(h);
        double *dr = d, *dv = d + len, *dl = d + 2 * len;
        hipStream_t st = h->stream;
        auto up = [&](double* dst, const double* src, size_t k) {
            if (k && src) HIP_CHECK(hipMemcpyAsync(dst, src, k * sizeof(double), hipMemcpyHostToDevice, st));
        };
        up(dr, rhs_x, n); up(dr + n, affine ? nullptr : rhs_s, m); up(dr + n + m, rhs_z, m);
        if (!reuse_vars) { up(dv, var_x, n); up(dv + n, var_s, m); up(dv + n + m, var_z, m); h->host_vars_valid = true; }
        int rc = sys_solve_step(h, dl, dl + n, dl + n + m, lhs_tau_kappa, dr, dr + n, dr + n + m, rhs_tau, rhs_kappa,
                                dv, dv + n, dv + n + m, var_tau, var_kappa, steptype);
        if (rc != HIPKKT_OK) return rc;
        if (n) HIP_CHECK(hipMemcpyAsync(lhs_x, dl, n * sizeof(double), hipMemcpyDeviceToHost, st));
        if (m) HIP_CHECK(hipMemcpyAsync(lhs_s, dl + n, m * sizeof(double), hipMemcpyDeviceToHost, st));
        if (m) HIP_CHECK(hipMemcpyAsync(lhs_z, dl + n + m, m * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return HIPKKT_OK;
    });
}

// ------------------------------------------------------------------------------------------------
//  Ruiz equilibration of the problem data (data_equilibrate!, problemdata.jl:133-221) -- SURVEY.md 8 f4.
//  Stand-alone: runs before the KKT solver is built from the scaled (P, A).
// ------------------------------------------------------------------------------------------------
static void expand_csc(int64_t ncols, const int64_t* colptr, const int64_t* rowval, int base, std::vector<int>& row,
                       std::vector<int>& col)
{
    const int64_t nnz = colptr[ncols] - base;
    row.resize((size_t)nnz);
    col.resize((size_t)nnz);
    for (int64_t j = 0; j < ncols; ++j)
        for (int64_t q = colptr[j] - base; q < colptr[j + 1] - base; ++q) {
            row[(size_t)q] = (int)(rowval[q] - base);
            col[(size_t)q] = (int)j;
        }
}

int hipkkt_equilibrate(int64_t n, int64_t m, const int64_t* Pcolptr, const int64_t* Prowval, double* Pnzval,
                       const int64_t* Acolptr, const int64_t* Arowval, double* Anzval, double* q, double* b,
                       int64_t ncones, const int32_t* cone_kinds, const int64_t* cone_dims, int32_t max_iter,
                       double min_scaling, double max_scaling, double* d, double* e, double* c, int index_base,
                       int device)
{
    return guarded([&]() {
        if (n < 0 || m < 0 || !Pcolptr || !Acolptr || (n && (!q || !d)) || (m && (!b || !e)) || !c || max_iter < 0 ||
            ncones < 0 || (ncones && (!cone_kinds || !cone_dims)))
            throw ArgError("hipkkt_equilibrate: bad argument");
        hipkkt_settings st{};
        st.device = device;
        select_device(st);
        hipStream_t stream;
        HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
        std::vector<int> prow, pcol, arow, acol;
        expand_csc(n, Pcolptr, Prowval, index_base, prow, pcol);
        expand_csc(n, Acolptr, Arowval, index_base, arow, acol);
        for (size_t j = 0; j < prow.size(); ++j)
            if (prow[j] < 0 || prow[j] >= n) throw ArgError("hipkkt_equilibrate: P row index out of range");
        for (size_t j = 0; j < arow.size(); ++j)
            if (arow[j] < 0 || arow[j] >= m) throw ArgError("hipkkt_equilibrate: A row index out of range");
        // cone layout (numel per cone: k(k+1)/2 for a PSD cone of side k)
        std::vector<int> ckind((size_t)ncones), coff((size_t)ncones), cnumel((size_t)ncones);
        int64_t off = 0;
        for (int64_t k = 0; k < ncones; ++k) {
            const int64_t dim = cone_dims[k];
            const int64_t ne = cone_kinds[k] == HIPKKT_CONE_PSD ? dim * (dim + 1) / 2 : dim;
            ckind[(size_t)k] = cone_kinds[k];
            coff[(size_t)k] = (int)off;
            cnumel[(size_t)k] = (int)ne;
            off += ne;
        }
        if (off != m) throw ArgError("hipkkt_equilibrate: cone dimensions do not add up to m");

        DBuf<int> dprow, dpcol, darow, dacol, dkind, doff, dnumel;
        dprow.upload(prow); dpcol.upload(pcol); darow.upload(arow); dacol.upload(acol);
        dkind.upload(ckind); doff.upload(coff); dnumel.upload(cnumel);
        DBuf<double> dP, dA, dq, db, dd, de, dw, ew, scal, partial;
        const size_t nnzP = prow.size(), nnzA = arow.size();
        dP.alloc(nnzP); dA.alloc(nnzA); dq.alloc((size_t)n); db.alloc((size_t)m);
        dd.alloc((size_t)n); de.alloc((size_t)m); dw.alloc((size_t)n); ew.alloc((size_t)m);
        scal.alloc(8); partial.alloc(512);
        if (nnzP) HIP_CHECK(hipMemcpyAsync(dP.p, Pnzval, nnzP * sizeof(double), hipMemcpyHostToDevice, stream));
        if (nnzA) HIP_CHECK(hipMemcpyAsync(dA.p, Anzval, nnzA * sizeof(double), hipMemcpyHostToDevice, stream));
        if (n) HIP_CHECK(hipMemcpyAsync(dq.p, q, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
        if (m) HIP_CHECK(hipMemcpyAsync(db.p, b, (size_t)m * sizeof(double), hipMemcpyHostToDevice, stream));
        std::vector<double> ones((size_t)std::max<int64_t>(std::max(n, m), 8), 1.0);
        if (n) HIP_CHECK(hipMemcpyAsync(dd.p, ones.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
        if (m) HIP_CHECK(hipMemcpyAsync(de.p, ones.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(scal.p, ones.data(), 8 * sizeof(double), hipMemcpyHostToDevice, stream));
        EquilDev E;
        E.n = (int)n; E.m = (int)m; E.nnzP = (int64_t)nnzP; E.nnzA = (int64_t)nnzA;
        E.Prow = dprow.p; E.Pcol = dpcol.p; E.Arow = darow.p; E.Acol = dacol.p;
        E.Pval = dP.p; E.Aval = dA.p; E.q = dq.p; E.b = db.p; E.d = dd.p; E.e = de.p; E.dwork = dw.p; E.ework = ew.p;
        E.scal = scal.p; E.partial = partial.p;
        for (int it = 0; it < max_iter; ++it) launch_equil_round(E, min_scaling, max_scaling, stream);
        if (max_iter > 0) launch_equil_rectify(E, dkind.p, doff.p, dnumel.p, nullptr, (int)ncones, stream);
        HIP_CHECK(hipGetLastError());
        if (nnzP) HIP_CHECK(hipMemcpyAsync(Pnzval, dP.p, nnzP * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (nnzA) HIP_CHECK(hipMemcpyAsync(Anzval, dA.p, nnzA * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (n) HIP_CHECK(hipMemcpyAsync(q, dq.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (m) HIP_CHECK(hipMemcpyAsync(b, db.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (n) HIP_CHECK(hipMemcpyAsync(d, dd.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (m) HIP_CHECK(hipMemcpyAsync(e, de.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(c, scal.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return HIPKKT_OK;
    });
}

// _update_matrix (data_updating.jl:169-194): values <- cscale * L[row] * R[col] * values for a CSC matrix
int hipkkt_scale_matrix_values(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowval, double* nzval,
                               const double* lscale, const double* rscale, double cscale, int index_base, int device)
{
    return guarded([&]() {
        if (nrows < 0 || ncols < 0 || !colptr || !lscale || !rscale) throw ArgError("hipkkt_scale_matrix_values: bad argument");
        hipkkt_settings st{};
        st.device = device;
        select_device(st);
        std::vector<int> row, col;
        expand_csc(ncols, colptr, rowval, index_base, row, col);
        if (row.empty()) return HIPKKT_OK;
        if (!nzval) throw ArgError("hipkkt_scale_matrix_values: bad argument");
        for (int r : row) if (r < 0 || r >= nrows) throw ArgError("hipkkt_scale_matrix_values: row index out of range");
        DBuf<int> drow, dcol;
        drow.upload(row); dcol.upload(col);
        DBuf<double> v, L, R;
        v.alloc(row.size()); L.alloc((size_t)nrows); R.alloc((size_t)ncols);
        HIP_CHECK(hipMemcpy(v.p, nzval, row.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(L.p, lscale, (size_t)nrows * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(R.p, rscale, (size_t)ncols * sizeof(double), hipMemcpyHostToDevice));
        launch_lrscale(v.p, drow.p, dcol.p, (int64_t)row.size(), L.p, R.p, cscale, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(nzval, v.p, row.size() * sizeof(double), hipMemcpyDeviceToHost));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_mul_Hs(hipkkt_kkt_t h, double* y, const double* x)
{
    return guarded([&]() {
        if (!h || !y || !x) throw ArgError("hipkkt_kkt_mul_Hs: bad argument");
        if (!h->scaling_valid) throw ArgError("mul_Hs needs a device-side scaling (hipkkt_kkt_update_from_sz)");
        HIP_CHECK(hipSetDevice(h->device));
        size_t bytes = (size_t)h->K.m * sizeof(double);
        HIP_CHECK(hipMemcpyAsync(h->sbuf.p, x, bytes, hipMemcpyHostToDevice, h->stream));
        launch_mul_Hs(h->cone_dev(), h->cone_state(), h->ybuf.p, h->sbuf.p, h->K.m, h->stream);
        HIP_CHECK(hipMemcpyAsync(y, h->ybuf.p, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_get_pattern(hipkkt_kkt_t h, int64_t* colptr, int64_t* rowval)
{
    return guarded([&]() {
        if (!h || !colptr || !rowval) throw ArgError("hipkkt_kkt_get_pattern: bad argument");
        std::copy(h->K.colptr.begin(), h->K.colptr.end(), colptr);
        for (int64_t q = 0; q < h->K.nnzK; ++q) rowval[q] = h->K.rowval[q];
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_get_values(hipkkt_kkt_t h, double* nzval)
{
    return guarded([&]() {
        if (!h || !nzval) throw ArgError("hipkkt_kkt_get_values: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        HIP_CHECK(hipMemcpyAsync(nzval, h->Kval.p, (size_t)h->K.nnzK * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_get_maps(hipkkt_kkt_t h, int64_t* mapP, int64_t* mapA, int64_t* mapHs, int64_t* map_diag,
                        int64_t* mapU, int64_t* mapV, int64_t* mapD, int64_t* dsigns)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        auto cp = [](const std::vector<int>& v, int64_t* o) { if (o) for (size_t i = 0; i < v.size(); ++i) o[i] = v[i]; };
        cp(h->K.mapP, mapP); cp(h->K.mapA, mapA); cp(h->K.mapHs, mapHs); cp(h->K.map_diag, map_diag);
        cp(h->K.mapU, mapU); cp(h->K.mapV, mapV); cp(h->K.mapD, mapD); cp(h->K.dsigns, dsigns);
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_get_perm(hipkkt_kkt_t h, int64_t* perm)
{
    return guarded([&]() {
        if (!h || !perm) throw ArgError("hipkkt_kkt_get_perm: bad argument");
        for (int i = 0; i < h->K.N; ++i) perm[i] = h->eng->S.perm[i];
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_get_Hs(hipkkt_kkt_t h, double* Hs)
{
    return guarded([&]() {
        if (!h || (h->K.nHs && !Hs)) throw ArgError("hipkkt_kkt_get_Hs: bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        if (h->K.nHs) HIP_CHECK(hipMemcpyAsync(Hs, h->Hs.p, (size_t)h->K.nHs * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_get_scaling(hipkkt_kkt_t h, double* lambda, double* psd_R, double* psd_Rinv)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        if (!h->scaling_valid) throw ArgError("hipkkt_kkt_get_scaling needs a device-side scaling (hipkkt_kkt_update_from_sz)");
        HIP_CHECK(hipSetDevice(h->device));
        if (lambda && h->K.m) HIP_CHECK(hipMemcpyAsync(lambda, h->lam.p, (size_t)h->K.m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (psd_R && h->psdR.n) HIP_CHECK(hipMemcpyAsync(psd_R, h->psdR.p, h->psdR.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (psd_Rinv && h->psdRinv.n) HIP_CHECK(hipMemcpyAsync(psd_Rinv, h->psdRinv.p, h->psdRinv.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_get_scaling_w(hipkkt_kkt_t h, double* w, double* eta)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        if (!h->scaling_valid) throw ArgError("hipkkt_kkt_get_scaling_w needs a device-side scaling (hipkkt_kkt_update_from_sz)");
        HIP_CHECK(hipSetDevice(h->device));
        if (w && h->K.m) HIP_CHECK(hipMemcpyAsync(w, h->w.p, (size_t)h->K.m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (eta && h->eta.n) HIP_CHECK(hipMemcpyAsync(eta, h->eta.p, h->eta.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

double hipkkt_kkt_last_regularizer(hipkkt_kkt_t h) { return h ? h->last_eps : 0.0; }
int64_t hipkkt_kkt_last_ir_iterations(hipkkt_kkt_t h) { return h ? h->last_ir : 0; }

int hipkkt_kkt_set_stream(hipkkt_kkt_t h, void* stream)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        HIP_CHECK(hipSetDevice(h->device));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->own_stream && h->stream) HIP_CHECK(hipStreamDestroy(h->stream));
        h->stream = (hipStream_t)stream;
        h->own_stream = false;
        h->eng->stream = h->stream;
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_synchronize(hipkkt_kkt_t h)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        HIP_CHECK(hipSetDevice(h->device));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_profile_enable(hipkkt_kkt_t h, int enable)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        h->prof.resolve();
        h->prof.enabled = enable != 0;
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_profile_reset(hipkkt_kkt_t h)
{
    return guarded([&]() {
        if (!h) throw ArgError("null handle");
        h->prof.resolve();
        h->prof.acc = hipkkt_profile{};
        return HIPKKT_OK;
    });
}

int hipkkt_kkt_profile_get(hipkkt_kkt_t h, hipkkt_profile* out)
{
    return guarded([&]() {
        if (!h || !out) throw ArgError("hipkkt_kkt_profile_get: bad argument");
        h->prof.resolve();
        *out = h->prof.acc;
        out->overlap_fallbacks = h->eng->n_ov_fallbacks;
        out->top_fallbacks = h->eng->n_top_fallbacks;
        out->overlap_deferrals = h->eng->n_ov_busy;
        out->top_deferrals = h->eng->n_top_busy;
        return HIPKKT_OK;
    });
}

}  // extern "C"
