// Fill-reducing orderings for the symmetric KKT pattern (host side, setup only).
//
// The reference gets its ordering from AMD.jl -> SuiteSparse AMD inside QDLDL.jl
// (call site /root/reference/src/kktsolvers/direct-ldl/directldl_qdldl.jl:18-25, with
// amd_dense_scale = 1.5); neither package is in the reference tree nor installed here, so
// this file implements the published approximate-minimum-degree algorithm (Amestoy, Davis,
// Duff 1996: quotient graph, element absorption, approximate external degree, mass
// elimination, supervariables, dense-row postponement) from scratch, plus a nested-dissection
// driver that produces the bushy elimination trees a GPU factorisation wants
// (SURVEY.md section 7.3 items 1-2).  Orderings differ from SuiteSparse's; parity is defined on the
// solution of K x = b, never on L.
#include "symbolic.hpp"
#include "knobs.hpp"

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <queue>
#include <vector>

namespace hipkkt {

// ---------------------------------------------------------------- AMD
// g: full symmetric pattern without diagonal.  Returns perm (perm[k] = node eliminated k-th).
//    halo (optional): nodes flagged 1 take part in the graph (their presence counts in every
//    degree) but are never chosen as pivots and do not appear in perm -- used by nested
//    dissection so that a sub-domain's ordering sees its already-placed separators.
void amd_order(const Graph& g, double dense_scale, std::vector<int>& perm, const char* halo)
{
    const int n = g.n;
    perm.clear();
    perm.reserve(n);
    if (n == 0) return;

    std::vector<std::vector<int>> vadj(n), eadj(n), emem(n);
    std::vector<int> nv(n, 1), degree(n, 0), esize(n, 0);
    std::vector<char> state(n, 0);          // 0 live variable, 1 live element, 2 dead/merged, 3 dense
    std::vector<int64_t> w(n, 0);
    std::vector<int> head(n + 1, -1), next(n, -1), prev(n, -1);
    std::vector<int> merged_head(n, -1), merged_next(n, -1);   // variables merged into / eliminated with i
    std::vector<int> pivots;
    pivots.reserve(n);

    // dense rows are postponed to the end (SuiteSparse default 10*sqrt(n), scaled by the
    // reference's amd_dense_scale)
    double dthr = std::min((double)n, std::max(16.0, dense_scale * 10.0 * std::sqrt((double)n)));
    int dense_thr = (int)dthr;
    std::vector<int> dense_nodes;
    for (int i = 0; i < n; ++i) {
        int d = (int)(g.ptr[i + 1] - g.ptr[i]);
        if (d > dense_thr && !(halo && halo[i])) { state[i] = 3; dense_nodes.push_back(i); }
    }
    int nlive = 0;
    for (int i = 0; i < n; ++i) {
        if (state[i] == 3) continue;
        auto& a = vadj[i];
        a.reserve(g.ptr[i + 1] - g.ptr[i]);
        for (int64_t q = g.ptr[i]; q < g.ptr[i + 1]; ++q) {
            int j = g.idx[q];
            if (j != i && state[j] != 3) a.push_back(j);
        }
        degree[i] = (int)a.size();
        if (!(halo && halo[i])) ++nlive;
    }
    auto dl_insert = [&](int i) {
        int d = degree[i];
        next[i] = head[d]; prev[i] = -1;
        if (head[d] >= 0) prev[head[d]] = i;
        head[d] = i;
    };
    auto dl_remove = [&](int i) {
        int d = degree[i];
        if (prev[i] >= 0) next[prev[i]] = next[i]; else head[d] = next[i];
        if (next[i] >= 0) prev[next[i]] = prev[i];
    };
    auto attach = [&](int child, int to) {     // order `child` right after `to`
        merged_next[child] = merged_head[to];
        merged_head[to] = child;
    };
    auto is_halo = [&](int i) { return halo && halo[i]; };
    for (int i = 0; i < n; ++i) if (state[i] == 0 && !is_halo(i)) dl_insert(i);

    int64_t wflg = 2;
    int mindeg = 0, nel = 0;
    std::vector<int> Lp, bucket_head(n, -1), bucket_next(n, -1), hashval(n, 0);
    std::vector<int64_t> mark(n, 0);
    int64_t mstamp = 1;

    while (nel < nlive) {
        while (mindeg < n && head[mindeg] < 0) ++mindeg;
        int p = head[mindeg];
        dl_remove(p);
        pivots.push_back(p);
        int nvp = nv[p];
        nel += nvp;
        // ---- build L_p; members flagged by negated nv
        Lp.clear();
        nv[p] = -nvp;
        int degme = 0;
        for (int v : vadj[p]) {
            if (nv[v] > 0 && state[v] == 0) { degme += nv[v]; nv[v] = -nv[v]; Lp.push_back(v); }
        }
        for (int e : eadj[p]) {
            if (state[e] != 1) continue;
            for (int v : emem[e]) {
                if (nv[v] > 0 && state[v] == 0) { degme += nv[v]; nv[v] = -nv[v]; Lp.push_back(v); }
            }
            state[e] = 2;                       // absorbed into p
            std::vector<int>().swap(emem[e]);
        }
        std::vector<int>().swap(vadj[p]);
        std::vector<int>().swap(eadj[p]);
        state[p] = 1;
        for (int i : Lp) if (!is_halo(i)) dl_remove(i);

        // ---- scan 1: |L_e \ L_p| for every element adjacent to a member
        if (wflg > (int64_t)1 << 60) { std::fill(w.begin(), w.end(), 0); wflg = 2; }
        for (int i : Lp) {
            int nvi = -nv[i];
            for (int e : eadj[i]) {
                if (state[e] != 1) continue;
                if (w[e] >= wflg) w[e] -= nvi;
                else w[e] = esize[e] + wflg - nvi;
            }
        }
        // ---- scan 2: prune lists, approximate degrees, hashes
        for (size_t t = 0; t < Lp.size(); ++t) {
            int i = Lp[t];
            int nvi = -nv[i];
            int64_t deg = 0;
            unsigned hash = 0;
            auto& ei = eadj[i];
            size_t wr = 0;
            for (int e : ei) {
                if (state[e] != 1) continue;
                int64_t dext = w[e] - wflg;
                if (dext > 0) { deg += dext; ei[wr++] = e; hash += (unsigned)e; }
                else { state[e] = 2; std::vector<int>().swap(emem[e]); }   // aggressive absorption
            }
            ei.resize(wr);
            ei.push_back(p);
            hash += (unsigned)p;
            auto& vi = vadj[i];
            wr = 0;
            for (int v : vi) {
                if (nv[v] > 0 && state[v] == 0) { deg += nv[v]; vi[wr++] = v; hash += (unsigned)v; }
            }
            vi.resize(wr);
            if (ei.size() == 1 && vi.empty() && !is_halo(i)) {
                // mass elimination: i has no neighbours outside L_p
                degme -= nvi;
                nel += nvi;
                nv[i] = 0;
                state[i] = 2;
                attach(i, p);
                std::vector<int>().swap(ei);
                Lp[t] = -1;
            } else {
                degree[i] = (int)std::min<int64_t>(degree[i], deg);
                hashval[i] = (int)(hash % (unsigned)n);
            }
        }
        wflg += (int64_t)n + 1;      // invalidates every w[e] set in scan 1 (esize <= n)

        // ---- supervariable detection among the members of L_p
        for (int i : Lp) {
            if (i < 0) continue;
            int h = hashval[i];
            bucket_next[i] = bucket_head[h];
            bucket_head[h] = i;
        }
        for (int i0 : Lp) {
            if (i0 < 0) continue;
            int h = hashval[i0];
            if (bucket_head[h] < 0) continue;
            // compare all pairs in this bucket
            for (int i = bucket_head[h]; i >= 0; i = bucket_next[i]) {
                if (nv[i] == 0) continue;
                ++mstamp;
                for (int e : eadj[i]) mark[e] = mstamp;
                for (int v : vadj[i]) mark[v] = mstamp;
                int prevj = i;
                for (int j = bucket_next[i]; j >= 0; j = bucket_next[j]) {
                    bool same = nv[j] != 0 && eadj[j].size() == eadj[i].size() &&
                                vadj[j].size() == vadj[i].size() && is_halo(i) == is_halo(j);
                    if (same) for (int e : eadj[j]) if (mark[e] != mstamp) { same = false; break; }
                    if (same) for (int v : vadj[j]) if (mark[v] != mstamp) { same = false; break; }
                    if (same) {
                        nv[i] += nv[j];          // both negative: sizes add
                        nv[j] = 0;
                        state[j] = 2;
                        attach(j, i);
                        std::vector<int>().swap(eadj[j]);
                        std::vector<int>().swap(vadj[j]);
                        bucket_next[prevj] = bucket_next[j];
                    } else {
                        prevj = j;
                    }
                }
            }
            bucket_head[h] = -1;
        }
        // ---- finalise: restore nv, final degrees, new element p
        auto& mem = emem[p];
        mem.clear();
        for (int i : Lp) {
            if (i < 0 || nv[i] == 0) continue;
            int nvi = -nv[i];
            nv[i] = nvi;
            int64_t deg = (int64_t)degree[i] + degme - nvi;
            deg = std::min<int64_t>(deg, (int64_t)n - nel - nvi);
            if (deg < 0) deg = 0;
            degree[i] = (int)deg;
            if (!is_halo(i)) {
                dl_insert(i);
                if (degree[i] < mindeg) mindeg = degree[i];
            }
            mem.push_back(i);
        }
        nv[p] = nvp;
        esize[p] = degme;
        if (mem.empty()) state[p] = 2;
    }
    // ---- emit: each pivot followed by everything merged into / eliminated with it
    std::vector<int> stack;
    for (int p : pivots) {
        stack.push_back(p);
        while (!stack.empty()) {
            int v = stack.back();
            stack.pop_back();
            perm.push_back(v);
            for (int c = merged_head[v]; c >= 0; c = merged_next[c]) stack.push_back(c);
        }
    }
    std::sort(dense_nodes.begin(), dense_nodes.end(), [&](int a, int b) {
        int64_t da = g.ptr[a + 1] - g.ptr[a], db = g.ptr[b + 1] - g.ptr[b];
        return da != db ? da < db : a < b;
    });
    for (int v : dense_nodes) perm.push_back(v);
}

// ------------------------------------------------------- nested dissection
namespace {

struct NDWork {
    const Graph* g;
    std::vector<int> part;      // current sub-problem id of each node, -1 = already ordered
    std::vector<int> level;     // BFS scratch
    std::vector<int> local;     // global -> local index scratch
};

// BFS from `root` restricted to nodes with part == pid; returns the level structure
static void bfs_levels(NDWork& W, int pid, int root, const std::vector<int>& nodes,
                       std::vector<int>& order, std::vector<int>& lvl_ptr)
{
    const Graph& g = *W.g;
    order.clear();
    lvl_ptr.clear();
    for (int v : nodes) W.level[v] = -1;
    order.push_back(root);
    W.level[root] = 0;
    lvl_ptr.push_back(0);
    size_t headq = 0;
    int cur = 0;
    while (headq < order.size()) {
        int v = order[headq];
        if (W.level[v] != cur) { lvl_ptr.push_back((int)headq); cur = W.level[v]; }
        ++headq;
        for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
            int u = g.idx[q];
            if (W.part[u] == pid && W.level[u] < 0) { W.level[u] = cur + 1; order.push_back(u); }
        }
    }
    lvl_ptr.push_back((int)order.size());
}

}  // namespace

// Recursive bisection by level-structure vertex separators (George's automatic nested
// dissection with pseudo-peripheral roots); leaves and separators are ordered by AMD on
// their induced subgraphs.  Ordering: [left][right][separator], recursively.
void nd_order(const Graph& g, int leaf_size, double dense_scale, std::vector<int>& perm)
{
    const int n = g.n;
    perm.assign(n, -1);
    if (n == 0) return;
    NDWork W;
    W.g = &g;
    W.part.assign(n, 0);
    W.level.assign(n, -1);
    W.local.assign(n, -1);

    // dense rows out first (ordered last), as AMD does
    int dense_thr = (int)std::min((double)n, std::max(16.0, dense_scale * 10.0 * std::sqrt((double)n)));
    std::vector<int> dense_nodes;
    for (int i = 0; i < n; ++i)
        if ((int)(g.ptr[i + 1] - g.ptr[i]) > dense_thr) { dense_nodes.push_back(i); W.part[i] = -3; }

    struct Task { std::vector<int> nodes; int pos_begin; };   // nodes get positions [pos_begin, +size)
    int next_pid = 1;
    std::vector<Task> stack;
    {
        Task t;
        for (int i = 0; i < n; ++i) if (W.part[i] == 0) t.nodes.push_back(i);
        t.pos_begin = 0;
        stack.push_back(std::move(t));
    }
    std::vector<char> halo_flag;
    std::vector<int> ext;
    auto order_block_amd = [&](const std::vector<int>& nodes, int pos_begin) {
        // AMD on the induced subgraph plus its halo of already-placed separator nodes
        int k = (int)nodes.size();
        for (int t = 0; t < k; ++t) W.local[nodes[t]] = t;
        ext.assign(nodes.begin(), nodes.end());
        for (int t = 0; t < k; ++t) {
            int v = nodes[t];
            for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
                int u = g.idx[q];
                if (W.local[u] < 0 && W.part[u] == -1) { W.local[u] = (int)ext.size(); ext.push_back(u); }
            }
        }
        int kk = (int)ext.size();
        halo_flag.assign(kk, 0);
        for (int t = k; t < kk; ++t) halo_flag[t] = 1;
        Graph sub;
        sub.n = kk;
        sub.ptr.assign(kk + 1, 0);
        for (int t = 0; t < kk; ++t) {
            int v = ext[t];
            for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
                int l = W.local[g.idx[q]];
                if (l >= 0 && !(t >= k && l >= k)) sub.ptr[t + 1]++;     // no halo-halo edges
            }
        }
        for (int t = 0; t < kk; ++t) sub.ptr[t + 1] += sub.ptr[t];
        sub.idx.resize(sub.ptr[kk]);
        for (int t = 0; t < kk; ++t) {
            int v = ext[t];
            int64_t w = sub.ptr[t];
            for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
                int l = W.local[g.idx[q]];
                if (l >= 0 && !(t >= k && l >= k)) sub.idx[w++] = l;
            }
        }
        std::vector<int> p;
        amd_order(sub, 1e9, p, halo_flag.data());
        int w = 0;
        for (int t = 0; t < (int)p.size(); ++t) if (p[t] < k) perm[pos_begin + w++] = nodes[p[t]];
        for (int t = 0; t < kk; ++t) W.local[ext[t]] = -1;
    };

    std::vector<int> order, lvl_ptr, order2, lvl_ptr2;
    while (!stack.empty()) {
        Task task = std::move(stack.back());
        stack.pop_back();
        int sz = (int)task.nodes.size();
        if (sz == 0) continue;
        if (sz <= leaf_size) { order_block_amd(task.nodes, task.pos_begin); for (int v : task.nodes) W.part[v] = -1; continue; }
        int pid = next_pid++;
        for (int v : task.nodes) W.part[v] = pid;
        // pseudo-peripheral root: a few BFS sweeps from the last level's min-degree node
        int root = task.nodes[0];
        bfs_levels(W, pid, root, task.nodes, order, lvl_ptr);
        if ((int)order.size() < sz) {
            // disconnected: every component becomes a task of its own, found in ONE pass over the task's nodes (peeling
            // them off one at a time is quadratic in their number: a separable problem of 60 000 two-node components
            // took 7.7 s).  The permutation is the one the one-at-a-time split produced: components are discovered from
            // the first unvisited node in task order and listed breadth first, and the trailing components that
            // together fit a leaf go to AMD as ONE block in task order (the old split's last remainder).
            std::vector<Task> comps;
            int pos = task.pos_begin;
            for (int v0 : task.nodes) {
                if (W.part[v0] != pid) continue;
                Task c;
                c.pos_begin = pos;
                c.nodes.push_back(v0);
                W.part[v0] = -2;
                W.level[v0] = (int)comps.size();
                for (size_t h = 0; h < c.nodes.size(); ++h) {
                    const int v = c.nodes[h];
                    for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
                        const int u = g.idx[q];
                        if (W.part[u] == pid) { W.part[u] = -2; W.level[u] = (int)comps.size(); c.nodes.push_back(u); }
                    }
                }
                pos += (int)c.nodes.size();
                comps.push_back(std::move(c));
            }
            // first component of the remainder that is small enough to be a leaf as a whole (never the first one:
            // the old split always peeled one component before looking at what was left)
            int tail = (int)comps.size(), tail_sz = 0;
            while (tail > 1 && tail_sz + (int)comps[tail - 1].nodes.size() <= leaf_size) tail_sz += (int)comps[--tail].nodes.size();
            if (tail < (int)comps.size()) {
                Task t;
                t.pos_begin = comps[tail].pos_begin;
                t.nodes.reserve(tail_sz);
                for (int v : task.nodes) if (W.level[v] >= tail) t.nodes.push_back(v);
                comps.resize(tail);
                comps.push_back(std::move(t));
            }
            for (int v : task.nodes) W.part[v] = 0;
            for (auto& c : comps) stack.push_back(std::move(c));
            continue;
        }
        for (int sweep = 0; sweep < 3; ++sweep) {
            int nl = (int)lvl_ptr.size() - 1;
            // min-degree node of the last level
            int best = order[lvl_ptr[nl - 1]];
            int64_t bd = INT64_MAX;
            for (int t = lvl_ptr[nl - 1]; t < lvl_ptr[nl]; ++t) {
                int v = order[t];
                int64_t d = g.ptr[v + 1] - g.ptr[v];
                if (d < bd) { bd = d; best = v; }
            }
            bfs_levels(W, pid, best, task.nodes, order2, lvl_ptr2);
            if (lvl_ptr2.size() > lvl_ptr.size()) { order.swap(order2); lvl_ptr.swap(lvl_ptr2); root = best; }
            else break;
        }
        int nl = (int)lvl_ptr.size() - 1;
        if (nl < 3) { order_block_amd(task.nodes, task.pos_begin); for (int v : task.nodes) W.part[v] = -1; continue; }
        // separator = the smallest level among those that leave both sides within [35%, 65%]
        int bestl = -1;
        int64_t bestsz = INT64_MAX;
        for (int l = 1; l < nl - 1; ++l) {
            int before = lvl_ptr[l], after = sz - lvl_ptr[l + 1];
            if (before < 0.30 * sz || after < 0.30 * sz) continue;
            int64_t s = lvl_ptr[l + 1] - lvl_ptr[l];
            if (s < bestsz) { bestsz = s; bestl = l; }
        }
        if (bestl < 0) {
            // fall back to the level containing the median node
            for (int l = 1; l < nl - 1; ++l) if (lvl_ptr[l + 1] > sz / 2) { bestl = l; break; }
            if (bestl < 0) bestl = nl / 2;
        }
        Task left, right;
        std::vector<int> sep;
        left.nodes.assign(order.begin(), order.begin() + lvl_ptr[bestl]);
        sep.assign(order.begin() + lvl_ptr[bestl], order.begin() + lvl_ptr[bestl + 1]);
        right.nodes.assign(order.begin() + lvl_ptr[bestl + 1], order.end());
        // thin the separator: a separator node with no neighbour on the right can move left
        {
            for (int v : right.nodes) W.level[v] = -7;
            std::vector<int> keep;
            for (int v : sep) {
                bool touches_right = false;
                for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q)
                    if (W.part[g.idx[q]] == pid && W.level[g.idx[q]] == -7) { touches_right = true; break; }
                if (touches_right) keep.push_back(v); else left.nodes.push_back(v);
            }
            sep.swap(keep);
        }
        // A level-structure separator is only worth its name on graphs with some geometry.  On a graph with long-range
        // edges (small diameter) a breadth-first level holds a large share of the nodes, and dissecting along it is
        // catastrophic -- cfg2 with 1 % of A's entries re-drawn over all columns: nnz(L) 3.0e9 against AMD's 1.4e8 (n = 20 000:
        // 2.8e8 against 9e6).  A subgraph of more than 5000 nodes whose separator is beyond a tenth of it goes to AMD as a
        // whole.  (Not the small ones: subgraphs with dense cliques -- cfg5's PSD blocks -- have relatively large level
        // separators and are still better off dissected; with these thresholds the orderings of cfg1-cfg5 are unchanged.
        // HIPKKT_ND_SEP_RATIO / HIPKKT_ND_SEP_MIN move them.)
        const double sep_ratio = knobs().nd_sep_ratio;
        const int sep_min = knobs().nd_sep_min;
        if ((double)sep.size() > sep_ratio * (double)sz && sz > sep_min) {
            for (int v : task.nodes) W.part[v] = 0;
            order_block_amd(task.nodes, task.pos_begin);
            for (int v : task.nodes) W.part[v] = -1;
            continue;
        }
        left.pos_begin = task.pos_begin;
        right.pos_begin = task.pos_begin + (int)left.nodes.size();
        int sep_begin = right.pos_begin + (int)right.nodes.size();
        for (int v : left.nodes) W.part[v] = 0;
        for (int v : right.nodes) W.part[v] = 0;
        order_block_amd(sep, sep_begin);
        for (int v : sep) W.part[v] = -1;
        stack.push_back(std::move(left));
        stack.push_back(std::move(right));
    }
    int pos = n - (int)dense_nodes.size();
    std::sort(dense_nodes.begin(), dense_nodes.end(), [&](int a, int b) {
        int64_t da = g.ptr[a + 1] - g.ptr[a], db = g.ptr[b + 1] - g.ptr[b];
        return da != db ? da < db : a < b;
    });
    for (int v : dense_nodes) perm[pos++] = v;
}

}  // namespace hipkkt
