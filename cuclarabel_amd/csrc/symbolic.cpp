// Symbolic analysis: ordering -> elimination tree -> postorder -> column counts ->
// supernodes (+ relaxed amalgamation) -> row structures, assembly maps, level schedule.
// See symbolic.hpp for what this replaces in the reference.
#include "symbolic.hpp"
#include "knobs.hpp"
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <stdexcept>

namespace hipkkt {

void build_graph(int N, const int64_t* colptr, const int64_t* rowval, int base, Graph& g)
{
    g.n = N;
    g.ptr.assign((size_t)N + 1, 0);
    for (int j = 0; j < N; ++j)
        for (int64_t q = colptr[j] - base; q < colptr[j + 1] - base; ++q) {
            int i = (int)(rowval[q] - base);
            if (i == j) continue;
            g.ptr[i + 1]++;
            g.ptr[j + 1]++;
        }
    for (int j = 0; j < N; ++j) g.ptr[j + 1] += g.ptr[j];
    g.idx.resize((size_t)g.ptr[N]);
    std::vector<int64_t> nxt(g.ptr.begin(), g.ptr.end() - 1);
    for (int j = 0; j < N; ++j)
        for (int64_t q = colptr[j] - base; q < colptr[j + 1] - base; ++q) {
            int i = (int)(rowval[q] - base);
            if (i == j) continue;
            g.idx[nxt[i]++] = j;
            g.idx[nxt[j]++] = i;
        }
}

namespace {

// elimination tree of the permuted matrix (Liu, with path compression)
void etree_of(const Graph& g, const std::vector<int>& perm, const std::vector<int>& iperm,
              std::vector<int>& parent)
{
    int n = g.n;
    parent.assign(n, -1);
    std::vector<int> anc(n, -1);
    for (int k = 0; k < n; ++k) {
        int v = perm[k];
        for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
            int i = iperm[g.idx[q]];
            while (i != -1 && i < k) {
                int nx = anc[i];
                anc[i] = k;
                if (nx == -1) parent[i] = k;
                i = nx;
            }
        }
    }
}

void postorder(const std::vector<int>& parent, std::vector<int>& post)
{
    int n = (int)parent.size();
    std::vector<int> head(n, -1), next(n, -1);
    for (int j = n - 1; j >= 0; --j)
        if (parent[j] >= 0) { next[j] = head[parent[j]]; head[parent[j]] = j; }
    post.clear();
    post.reserve(n);
    std::vector<int> stack;
    for (int r = 0; r < n; ++r) {
        if (parent[r] >= 0) continue;
        stack.push_back(r);
        while (!stack.empty()) {
            int v = stack.back();
            int c = head[v];
            if (c >= 0) { head[v] = next[c]; stack.push_back(c); }
            else { post.push_back(v); stack.pop_back(); }
        }
    }
}

}  // namespace

void analyse(int N, const int64_t* colptr, const int64_t* rowval, int base,
             const SymbolicOptions& opt, Symbolic& S)
{
    S = Symbolic();
    S.N = N;
    S.nnzK = colptr[N] - base;
    Graph g;
    build_graph(N, colptr, rowval, base, g);

    // ---- 1. fill-reducing ordering
    std::vector<int> perm0;
    switch (opt.ordering) {
    case ORDER_AMD: amd_order(g, opt.amd_dense_scale, perm0, nullptr); break;
    case ORDER_ND: nd_order(g, opt.nd_leaf_size, opt.amd_dense_scale, perm0); break;
    case ORDER_USER:
        if (!opt.user_perm) throw std::runtime_error("ORDER_USER without a permutation");
        perm0.resize(N);
        for (int i = 0; i < N; ++i) perm0[i] = (int)(opt.user_perm[i] - base);
        break;
    default:
        perm0.resize(N);
        std::iota(perm0.begin(), perm0.end(), 0);
    }
    {
        std::vector<char> seen(N, 0);
        if ((int)perm0.size() != N) throw std::runtime_error("ordering: wrong length");
        for (int v : perm0) {
            if (v < 0 || v >= N || seen[v]) throw std::runtime_error("ordering: not a permutation");
            seen[v] = 1;
        }
    }
    std::vector<int> iperm0(N);
    for (int k = 0; k < N; ++k) iperm0[perm0[k]] = k;

    // ---- 2. etree, postorder, relabel
    std::vector<int> parent0, post;
    etree_of(g, perm0, iperm0, parent0);
    postorder(parent0, post);
    std::vector<int> perm1(N), iperm1(N), parent(N);
    {
        std::vector<int> ipost(N);
        for (int k = 0; k < N; ++k) ipost[post[k]] = k;
        for (int k = 0; k < N; ++k) perm1[k] = perm0[post[k]];
        for (int k = 0; k < N; ++k) iperm1[perm1[k]] = k;
        for (int k = 0; k < N; ++k) parent[k] = parent0[post[k]] >= 0 ? ipost[parent0[post[k]]] : -1;
    }
    // ---- 3. column counts of L (row-subtree traversal, O(nnz(L)))
    std::vector<int> cc(N, 0);
    {
        std::vector<int> mark(N, -1);
        for (int i = 0; i < N; ++i) {
            mark[i] = i;
            int v = perm1[i];
            for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
                int k = iperm1[g.idx[q]];
                while (k < i && mark[k] != i) { cc[k]++; mark[k] = i; k = parent[k]; }
            }
        }
        int64_t s = 0;
        for (int j = 0; j < N; ++j) s += cc[j];
        S.nnzL_struct = s;
        // height of the column etree
        std::vector<int> h(N, 1);
        int hm = 0;
        for (int j = 0; j < N; ++j) {
            if (parent[j] >= 0) h[parent[j]] = std::max(h[parent[j]], h[j] + 1);
            hm = std::max(hm, h[j]);
        }
        S.etree_height = hm;
    }
    // ---- 4. maximal supernodes in the postordered labelling
    std::vector<int> sn_of(N), fs_start;     // fundamental/maximal supernode of each column
    {
        std::vector<int> nchild(N, 0);
        for (int j = 0; j < N; ++j) if (parent[j] >= 0) nchild[parent[j]]++;
        for (int j = 0; j < N; ++j) {
            bool join = j > 0 && parent[j - 1] == j && cc[j - 1] == cc[j] + 1;
            if (!join) fs_start.push_back(j);
            sn_of[j] = (int)fs_start.size() - 1;
        }
        fs_start.push_back(N);
    }
    int nfs = (int)fs_start.size() - 1;
    // supernodal tree + per-node stats
    std::vector<int> fs_parent(nfs, -1), fs_nc(nfs), fs_nb(nfs);
    std::vector<double> fs_zeros(nfs, 0.0);
    for (int s = 0; s < nfs; ++s) {
        int last = fs_start[s + 1] - 1;
        fs_nc[s] = fs_start[s + 1] - fs_start[s];
        fs_nb[s] = cc[last];
        fs_parent[s] = parent[last] >= 0 ? sn_of[parent[last]] : -1;
    }
    // ---- 5. relaxed amalgamation (bottom-up); merged[s] = representative it was merged into
    std::vector<int> merged(nfs);
    std::iota(merged.begin(), merged.end(), 0);
    {
        std::vector<std::vector<int>> kids(nfs);
        for (int s = 0; s < nfs; ++s) if (fs_parent[s] >= 0) kids[fs_parent[s]].push_back(s);
        auto trap = [](double nc, double nb) { return nc * (nc + 1) / 2 + nc * nb; };
        // Every level of the tree is a round of dependent launches in the numeric phase, so the HEIGHT of the
        // amalgamated tree matters more than its zeros: the child with the tallest subtree is tried first and with
        // relax_tall times the usual zero allowance (absorbing it takes one level off the path through this node);
        // the other children follow, widest first, with the plain allowance -- after the tall child has widened the
        // parent they are less likely to pile up zeros than if they had gone first.
        std::vector<int> hgt(nfs, 1);         // height of the amalgamated subtree under each node
        for (int p = 0; p < nfs; ++p) {       // postorder: children have smaller ids
            auto& ks = kids[p];
            std::sort(ks.begin(), ks.end(), [&](int a, int b) { return hgt[a] != hgt[b] ? hgt[a] > hgt[b] : fs_nc[a] > fs_nc[b]; });
            std::vector<int> newkids;
            for (int c : ks) {
                double nc = (double)fs_nc[c] + fs_nc[p], nb = fs_nb[p];
                double total = trap(nc, nb);
                double truennz = (trap(fs_nc[c], fs_nb[c]) - fs_zeros[c]) + (trap(fs_nc[p], fs_nb[p]) - fs_zeros[p]);
                double zeros = total - truennz;
                double frac = zeros / total;
                const double mult = (c == ks[0]) ? opt.relax_tall : 1.0;
                bool ok;
                if (nc <= opt.relax_cols[0]) ok = frac <= opt.relax_zeros[0] * mult;
                else if (nc <= opt.relax_cols[1]) ok = frac <= opt.relax_zeros[1] * mult;
                else if (nc <= opt.relax_cols[2]) ok = frac <= opt.relax_zeros[2] * mult;
                else ok = frac <= opt.relax_zeros[3] * mult;
                if (ok) {
                    merged[c] = p;
                    fs_nc[p] = (int)nc;
                    fs_zeros[p] = zeros;
                    for (int gc : kids[c]) { newkids.push_back(gc); }
                } else {
                    newkids.push_back(c);
                }
            }
            ks.swap(newkids);
            // SIBLING BUNDLES.  A supernode left with very many children -- the multiplier of a dense equality row whose
            // 3000 singleton columns hang off the root, a hub variable of a transportation LP, the n-column block under
            // the 840 one-column slack leaves of a dense A -- is assembled by ONE workgroup (or one wave) that walks its
            // children in turn, and gathered from by one workgroup in the forward sweep: 2.3 ms for those 3000 children
            // where the rest of the factorisation takes 0.2.  Children are therefore put together, in the order of
            // their row counts, into supernodes of their own ("bundles": siblings side by side, explicit zeros
            // between them -- the same layout a parent gets when it absorbs two children) as long as a bundle stays
            // a one-wave front (f <= 64), or beyond that, up to the widest panel, where the bundle's panel and update block
            // are smaller than its members' together (children that share a tall row structure: 96 one-column leaves of
            // 600 rows are one 600 x 600 update block instead of 96).  The row count of a bundle is not known here; its
            // bound -- the members' sum, at most the parent's front -- errs on the side of not bundling.
            //   WHEN.  Measured per child (MI355X): 0.77 us in the one-wave front kernel, 0.13 us in the forward sweep's
            // gather, 0.055 us in the 1024-thread panel kernel, against ~0.1 ms that a level of bundles costs (a launch
            // class more at the bottom of the tree, larger blocks to add).  So: a parent that is a one-wave front itself
            // (f <= 64) from bundle_kids children (200; cfg2's second-order-cone columns have up to ~190 and are left
            // alone: bundling them cost 1.3 % of a step), a wider parent from bundle_kids_panel (400) children per panel of
            // panel_max_cols columns (cfg3's 500 one-row slack leaves per dense block stay as they are: bundled they
            // made its factorisation 12 % slower), and any parent whose children's update blocks together are more than
            // 16 times its own front (the dense-A case: storage and traffic, not the count).
            const Knobs& kn = knobs();
            const double umax = (double)fs_nc[p] + fs_nb[p];
            bool bundle = false;
            if (kn.bundle_kids > 0 && ks.size() >= 2) {
                const double K = (double)ks.size();
                if (umax <= 64) bundle = K > kn.bundle_kids;
                else bundle = K * std::min(1.0, (double)opt.panel_max_cols / std::max(1, fs_nc[p])) > kn.bundle_kids_panel;
                if (!bundle && kn.bundle_cost) {
                    double upd = 0;
                    for (int c : ks) upd += (double)fs_nb[c] * fs_nb[c];
                    bundle = upd > 16.0 * umax * umax;
                }
            }
            if (bundle) {
                const bool by_cost = kn.bundle_cost;
                auto cost = [&](double nc, double nb) { return trap(nc, nb) + nb * nb; };
                // sqrt(K) members each: the bundles assemble side by side one level down, the parent walks K / g of
                // them, and the explicit zeros grow with g -- K (g / 2 + rows) entries in all
                const int per_bundle = std::max(8, (int)std::ceil(std::sqrt((double)ks.size())));
                std::vector<int> order(ks);
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fs_nb[a] > fs_nb[b]; });
                std::vector<int> out, cur;
                double cur_nc = 0, cur_nbsum = 0, cur_cost = 0, cur_true = 0;
                auto close = [&]() {
                    if (cur.size() > 1) {
                        const int r = *std::max_element(cur.begin(), cur.end());      // the representative: the last in postorder
                        const double u = std::min(cur_nbsum, umax);
                        std::vector<int> gk;
                        int h = 1;
                        for (int c : cur) {
                            h = std::max(h, hgt[c]);
                            gk.insert(gk.end(), kids[c].begin(), kids[c].end());
                            if (c != r) merged[c] = r;
                        }
                        kids[r].swap(gk);
                        hgt[r] = h;
                        fs_nc[r] = (int)cur_nc;
                        fs_nb[r] = (int)u;
                        fs_zeros[r] = trap(cur_nc, u) - cur_true;
                        out.push_back(r);
                    } else if (cur.size() == 1) {
                        out.push_back(cur[0]);
                    }
                    cur.clear();
                    cur_nc = cur_nbsum = cur_cost = cur_true = 0;
                };
                for (int c : order) {
                    const double nc_c = fs_nc[c], nb_c = fs_nb[c];
                    if (!cur.empty()) {
                        const double nc2 = cur_nc + nc_c, u2 = std::min(cur_nbsum + nb_c, umax);
                        const bool fits = nc2 <= (double)opt.panel_max_cols && (opt.panel_cap <= 0 || (nc2 + u2) * nc2 <= (double)opt.panel_cap);   // one panel, no chain
                        const bool one_wave = nc2 + u2 <= 64;
                        const bool cheaper = by_cost && cost(nc2, u2) <= cur_cost + cost(nc_c, nb_c);
                        if (!(fits && (one_wave || cheaper)) || (int)cur.size() >= per_bundle) close();
                    }
                    cur.push_back(c);
                    cur_nc += nc_c;
                    cur_nbsum += nb_c;
                    cur_cost += cost(nc_c, nb_c);
                    cur_true += trap(nc_c, nb_c) - fs_zeros[c];
                }
                close();
                ks.swap(out);
            }
            for (int c : ks) { fs_parent[c] = p; hgt[p] = std::max(hgt[p], hgt[c] + 1); }
        }
    }
    auto rep = [&](int s) { while (merged[s] != s) s = merged[s]; return s; };
    // ---- 6. final ordering: DFS postorder over the amalgamated tree, member columns kept in
    //         their (topological) postorder index order
    std::vector<int> final_of_fs(nfs, -1);
    std::vector<std::vector<int>> members(nfs);      // fundamental supernodes of each representative
    for (int s = 0; s < nfs; ++s) members[rep(s)].push_back(s);
    std::vector<int> aparent(nfs, -1);
    for (int s = 0; s < nfs; ++s) {
        if (merged[s] != s) continue;
        int p = fs_parent[s];
        aparent[s] = p >= 0 ? rep(p) : -1;
    }
    std::vector<int> apost;
    {
        std::vector<int> ap(nfs, -1);
        // postorder() wants a parent array over a dense id range: use representatives only
        std::vector<int> ids;
        for (int s = 0; s < nfs; ++s) if (merged[s] == s) ids.push_back(s);
        std::vector<int> dense(nfs, -1);
        for (size_t t = 0; t < ids.size(); ++t) dense[ids[t]] = (int)t;
        std::vector<int> par(ids.size());
        for (size_t t = 0; t < ids.size(); ++t) par[t] = aparent[ids[t]] >= 0 ? dense[aparent[ids[t]]] : -1;
        std::vector<int> po;
        postorder(par, po);
        for (int t : po) apost.push_back(ids[t]);
    }
    S.nsuper = (int)apost.size();
    S.perm.resize(N);
    S.iperm.resize(N);
    S.sn_start.assign(S.nsuper + 1, 0);
    S.col2sn.resize(N);
    {
        int pos = 0;
        for (int t = 0; t < S.nsuper; ++t) {
            int r = apost[t];
            final_of_fs[r] = t;
            S.sn_start[t] = pos;
            for (int fsn : members[r])            // ascending fundamental ids = ascending columns
                for (int j = fs_start[fsn]; j < fs_start[fsn + 1]; ++j) {
                    S.perm[pos] = perm1[j];
                    S.col2sn[pos] = t;
                    ++pos;
                }
        }
        S.sn_start[S.nsuper] = pos;
        for (int k = 0; k < N; ++k) S.iperm[S.perm[k]] = k;
    }
    S.sn_parent.assign(S.nsuper, -1);
    for (int t = 0; t < S.nsuper; ++t) {
        int p = aparent[apost[t]];
        S.sn_parent[t] = p >= 0 ? final_of_fs[p] : -1;
    }
    S.child_ptr.assign(S.nsuper + 1, 0);
    for (int t = 0; t < S.nsuper; ++t) if (S.sn_parent[t] >= 0) S.child_ptr[S.sn_parent[t] + 1]++;
    for (int t = 0; t < S.nsuper; ++t) S.child_ptr[t + 1] += S.child_ptr[t];
    S.child_idx.resize(S.child_ptr[S.nsuper]);
    {
        std::vector<int> nxt(S.child_ptr.begin(), S.child_ptr.end() - 1);
        for (int t = 0; t < S.nsuper; ++t) if (S.sn_parent[t] >= 0) S.child_idx[nxt[S.sn_parent[t]]++] = t;
    }
    // ---- 7. row structure of every supernode (bottom-up union)
    S.rowptr.assign(S.nsuper + 1, 0);
    {
        std::vector<int> mark(N, -1);
        std::vector<std::vector<int>> rws(S.nsuper);
        for (int s = 0; s < S.nsuper; ++s) {
            int c0 = S.sn_start[s], c1 = S.sn_start[s + 1];
            auto& r = rws[s];
            for (int j = c0; j < c1; ++j) {
                int v = S.perm[j];
                for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
                    int i = S.iperm[g.idx[q]];
                    if (i >= c1 && mark[i] != s) { mark[i] = s; r.push_back(i); }
                }
            }
            for (int e = S.child_ptr[s]; e < S.child_ptr[s + 1]; ++e) {
                int c = S.child_idx[e];
                for (int i : rws[c]) if (i >= c1 && mark[i] != s) { mark[i] = s; r.push_back(i); }
            }
            std::sort(r.begin(), r.end());
            S.rowptr[s + 1] = S.rowptr[s] + (int64_t)r.size();
            // children's lists are no longer needed once the parent is done -- but a child is only
            // consumed by its own parent, so free them here
            for (int e = S.child_ptr[s]; e < S.child_ptr[s + 1]; ++e) {
                int c = S.child_idx[e];
                // keep: copied out below
                (void)c;
            }
        }
        S.rows.resize((size_t)S.rowptr[S.nsuper]);
        for (int s = 0; s < S.nsuper; ++s) std::copy(rws[s].begin(), rws[s].end(), S.rows.begin() + S.rowptr[s]);
    }
    // ---- 7b. split supernodes whose panel (f x nc doubles) exceeds the LDS-resident cap into a chain
    //          of narrower ones: chunk i keeps columns [a_i, a_i+1) and gets the later columns as rows
    if (opt.panel_cap > 0) {
        std::vector<int> nstart, nparent_old;      // new supernode starts; old id each chunk came from
        std::vector<int> first_new(S.nsuper), last_new(S.nsuper);
        std::vector<int64_t> nrowptr(1, 0);
        std::vector<int> nrows;
        nrows.reserve(S.rows.size());
        for (int s0 = 0; s0 < S.nsuper; ++s0) {
            int c0 = S.sn_start[s0], c1 = S.sn_start[s0 + 1];
            int64_t nb = S.rowptr[s0 + 1] - S.rowptr[s0];
            first_new[s0] = (int)nstart.size();
            int a = c0;
            while (a < c1) {
                int64_t fa = (c1 - a) + nb;
                // widest panel (<= panel_max_cols) whose LDS image fits: a trapezoid in one workgroup, or row slices
                // in up to panel_max_slices workgroups (symbolic.hpp: panel_slice_doubles)
                int64_t wd = std::min<int64_t>(c1 - a, opt.panel_max_cols);
                auto solve_fits = [&](int64_t w) {
                    const int64_t fp = (fa + 3) & ~(int64_t)3, wp = (w + 3) & ~(int64_t)3;
                    return fp * (1 + (w + 7) / 8) <= opt.solve_cap && fp + ((fa + 7) / 8) * wp <= opt.solve_cap;
                };
                // (a front too tall for the block sweep kernels at ANY width -- beyond ~10 000 rows -- goes to the tall-front
                //  kernels, solve_kernels.hip: its width is the panel kernel's business alone)
                // (r03: the same for a front that does not fit them at the widest panel the factorisation would take -- beyond
                //  ~1540 rows at 96 columns.  Narrowing its panels to fit cost a level per 17 columns at 6000 rows, and every
                //  level of a chain passes the whole trailing block through HBM; such fronts go through the sliced
                //  persistent kernel anyway, the tall kernels are their per-level fallback.)
                const bool tall = !solve_fits(std::min<int64_t>(wd, c1 - a));
                while (wd > 1 && (panel_slices_needed(wd, fa - wd, opt.panel_cap, std::max(1, opt.panel_max_slices)) == 0 ||
                                  (!tall && !solve_fits(wd)))) --wd;
                // slices cost a second launch per level and redundant diagonal work: only where they save a level,
                // i.e. where the unsliced width would need more chunks for the remaining columns
                int64_t w1 = wd;
                while (w1 > 1 && panel_slices_needed(w1, fa - w1, opt.panel_cap, 1) == 0) --w1;
                if (w1 >= opt.panel_slice_below || (c1 - a + w1 - 1) / w1 <= (c1 - a + wd - 1) / wd) wd = w1;
                // avoid a sliver at the end: balance the remaining columns over the remaining chunks
                int64_t rem = c1 - a;
                if (wd < rem) { int64_t parts = (rem + wd - 1) / wd; wd = (rem + parts - 1) / parts; }
                nstart.push_back(a);
                nparent_old.push_back(s0);
                int e = a + (int)wd;
                for (int j = e; j < c1; ++j) nrows.push_back(j);
                nrows.insert(nrows.end(), S.rows.begin() + S.rowptr[s0], S.rows.begin() + S.rowptr[s0 + 1]);
                nrowptr.push_back((int64_t)nrows.size());
                a = e;
            }
            last_new[s0] = (int)nstart.size() - 1;
        }
        int nnew = (int)nstart.size();
        if (nnew != S.nsuper) {
            std::vector<int> nparent(nnew, -1);
            for (int t = 0; t < nnew; ++t) {
                int s0 = nparent_old[t];
                if (t < last_new[s0]) nparent[t] = t + 1;
                else nparent[t] = S.sn_parent[s0] >= 0 ? first_new[S.sn_parent[s0]] : -1;
            }
            nstart.push_back(N);
            S.nsuper = nnew;
            S.sn_start.swap(nstart);
            S.sn_parent.swap(nparent);
            S.rowptr.swap(nrowptr);
            S.rows.swap(nrows);
            for (int t = 0; t < nnew; ++t)
                for (int j = S.sn_start[t]; j < S.sn_start[t + 1]; ++j) S.col2sn[j] = t;
            S.child_ptr.assign(S.nsuper + 1, 0);
            for (int t = 0; t < S.nsuper; ++t) if (S.sn_parent[t] >= 0) S.child_ptr[S.sn_parent[t] + 1]++;
            for (int t = 0; t < S.nsuper; ++t) S.child_ptr[t + 1] += S.child_ptr[t];
            S.child_idx.resize(S.child_ptr[S.nsuper]);
            std::vector<int> nxt(S.child_ptr.begin(), S.child_ptr.end() - 1);
            for (int t = 0; t < S.nsuper; ++t) if (S.sn_parent[t] >= 0) S.child_idx[nxt[S.sn_parent[t]]++] = t;
        }
    }
    // sanity: every below-row of a child lies in the parent's columns or rows
    S.rel.assign(S.rows.size(), -1);
    {
        std::vector<int> loc(N, -1);
        for (int p = 0; p < S.nsuper; ++p) {
            int c0 = S.sn_start[p], c1 = S.sn_start[p + 1], nc = c1 - c0;
            for (int j = c0; j < c1; ++j) loc[j] = j - c0;
            for (int64_t q = S.rowptr[p]; q < S.rowptr[p + 1]; ++q) loc[S.rows[q]] = nc + (int)(q - S.rowptr[p]);
            for (int e = S.child_ptr[p]; e < S.child_ptr[p + 1]; ++e) {
                int c = S.child_idx[e];
                for (int64_t q = S.rowptr[c]; q < S.rowptr[c + 1]; ++q) {
                    int l = loc[S.rows[q]];
                    if (l < 0) throw std::runtime_error("symbolic: child row missing from parent front");
                    S.rel[q] = l;
                }
            }
            for (int j = c0; j < c1; ++j) loc[j] = -1;
            for (int64_t q = S.rowptr[p]; q < S.rowptr[p + 1]; ++q) loc[S.rows[q]] = -1;
        }
    }
    // ---- 8. storage offsets, K scatter map, stats
    S.front_off.assign(S.nsuper + 1, 0);
    S.upd_off.assign(S.nsuper + 1, 0);
    S.nnzL = 0;
    S.flops = 0;
    S.max_front = 0;
    for (int s = 0; s < S.nsuper; ++s) {
        int64_t nc = S.sn_start[s + 1] - S.sn_start[s], nb = S.rowptr[s + 1] - S.rowptr[s], f = nc + nb;
        S.front_off[s + 1] = S.front_off[s] + f * nc;
        S.upd_off[s + 1] = S.upd_off[s] + nb * nb;
        S.nnzL += nc * (nc - 1) / 2 + nc * nb;
        for (int64_t k = 0; k < nc; ++k) { double c = (double)(nc - 1 - k + nb); S.flops += c * c + 3 * c; }
        S.max_front = std::max<int>(S.max_front, (int)f);
        if (f * nc >= (int64_t)1 << 31) throw std::runtime_error("symbolic: panel too large for int32 offsets");
    }
    S.front_store = S.front_off[S.nsuper];
    S.update_store = S.upd_off[S.nsuper];
    // K entries: triu entry (i,j) of the original -> lower entry (max,min) of the permuted matrix
    S.kptr.assign(S.nsuper + 1, 0);
    S.diag_src.assign(N, -1);
    std::vector<int> ecol((size_t)S.nnzK), erow((size_t)S.nnzK);
    for (int j = 0; j < N; ++j)
        for (int64_t q = colptr[j] - base; q < colptr[j + 1] - base; ++q) {
            int i = (int)(rowval[q] - base);
            int a = S.iperm[i], b = S.iperm[j];
            int col = std::min(a, b), row = std::max(a, b);
            ecol[q] = col; erow[q] = row;
            S.kptr[S.col2sn[col] + 1]++;
            if (i == j) S.diag_src[col] = (int)q;
        }
    for (int j = 0; j < N; ++j) if (S.diag_src[j] < 0) throw std::runtime_error("symbolic: KKT column without a structural diagonal");
    for (int s = 0; s < S.nsuper; ++s) S.kptr[s + 1] += S.kptr[s];
    S.ksrc.resize((size_t)S.nnzK);
    S.kdst.resize((size_t)S.nnzK);
    {
        std::vector<int64_t> nxt(S.kptr.begin(), S.kptr.end() - 1);
        for (int64_t q = 0; q < S.nnzK; ++q) {
            int s = S.col2sn[ecol[q]];
            int c0 = S.sn_start[s], c1 = S.sn_start[s + 1], nc = c1 - c0;
            int64_t nb = S.rowptr[s + 1] - S.rowptr[s], f = nc + nb;
            int lrow;
            if (erow[q] < c1) lrow = erow[q] - c0;
            else {
                const int* rb = S.rows.data() + S.rowptr[s];
                const int* it = std::lower_bound(rb, rb + nb, erow[q]);
                if (it == rb + nb || *it != erow[q]) throw std::runtime_error("symbolic: K entry outside the front structure");
                lrow = nc + (int)(it - rb);
            }
            int64_t d = nxt[s]++;
            S.ksrc[d] = (int)q;
            S.kdst[d] = (int)(lrow + (int64_t)(ecol[q] - c0) * f);
        }
    }
    // ---- 9. level schedule (leaves = level 0)
    S.sn_level.assign(S.nsuper, 0);
    int nlev = 0;
    for (int s = 0; s < S.nsuper; ++s) {
        int p = S.sn_parent[s];
        if (p >= 0) S.sn_level[p] = std::max(S.sn_level[p], S.sn_level[s] + 1);
        nlev = std::max(nlev, S.sn_level[s] + 1);
    }
    std::vector<int> cnt(nlev + 1, 0);
    for (int s = 0; s < S.nsuper; ++s) cnt[S.sn_level[s] + 1]++;
    for (int l = 0; l < nlev; ++l) cnt[l + 1] += cnt[l];
    S.level_sn.resize(S.nsuper);
    S.levels.resize(nlev);
    for (int l = 0; l < nlev; ++l) { S.levels[l].begin = cnt[l]; S.levels[l].end = cnt[l + 1]; }
    {
        std::vector<int> nxt(cnt.begin(), cnt.end() - 1);
        for (int s = 0; s < S.nsuper; ++s) S.level_sn[nxt[S.sn_level[s]]++] = s;
    }
    // ---- 10. the stores are laid out LEVEL BY LEVEL (a level's fronts side by side, leaves first), not in supernode
    // order: a front's children sit (mostly) one level below it, so the update blocks a panel gathers from, and the
    // panels / solve matrices one launch touches, share a few large pages instead of one page each -- a front near
    // the root has a dozen children whose blocks were megabytes apart in postorder, and its assembly spent 10-20 us
    // on what are ~30 scattered first touches (address translation, not bytes).  Offsets stay per supernode;
    // entry [nsuper] holds the total.  (HIPKKT_POSTORDER_LAYOUT=1 keeps the postorder layout, for comparison.)
    if (!knobs().postorder_layout) {
        // CHAINS share two update blocks.  In a chain of panels (a supernode cut into panels, 7b: the child's update block
        // IS its parent's whole front) the block of link i is read by link i + 1 alone -- by its panel and by its tiles --
        // and is dead once link i + 1's tiles are done, which is before link i + 2's tiles start (a stream's kernels run in
        // order; in overlap mode the tiles of a launch start when the previous launch's have finished).  Links alternate
        // between two buffers of the sizes of the first two (the blocks shrink along the chain): a 14 000-row root cut into
        // 148 panels keeps 2 x 1.6 GB instead of 148 blocks, 150 GB.  (HIPKKT_UPD_PINGPONG=0: every block its own.)
        const bool pingpong = knobs().upd_pingpong;
        std::vector<int> pred((size_t)S.nsuper, -1), next((size_t)S.nsuper, -1);
        auto nbof = [&](int s) { return (int64_t)(S.rowptr[s + 1] - S.rowptr[s]); };
        auto fof = [&](int s) { return (int64_t)(S.sn_start[s + 1] - S.sn_start[s]) + nbof(s); };
        for (int c = 0; pingpong && c < S.nsuper; ++c) {
            const int p = S.sn_parent[c];
            if (p >= 0 && pred[(size_t)p] < 0 && nbof(c) == fof(p) && nbof(c) > 0) { pred[(size_t)p] = c; next[(size_t)c] = p; }
        }
        std::vector<int64_t> chain_a((size_t)S.nsuper, -1), chain_b((size_t)S.nsuper, -1);     // per chain head: its two buffers
        std::vector<int> head((size_t)S.nsuper, -1), link((size_t)S.nsuper, 0);
        for (int h = 0; h < S.nsuper; ++h) {
            if (pred[(size_t)h] >= 0 || next[(size_t)h] < 0) continue;
            int len = 0;
            for (int m = h; m >= 0; m = next[(size_t)m]) ++len;
            if (len < 3) continue;
            int i = 0;
            for (int m = h; m >= 0; m = next[(size_t)m], ++i) { head[(size_t)m] = h; link[(size_t)m] = i; }
        }
        int64_t fo = 0, uo = 0;
        for (int t = 0; t < S.nsuper; ++t) {
            const int s = S.level_sn[t];
            const int64_t nc = S.sn_start[s + 1] - S.sn_start[s], nb = S.rowptr[s + 1] - S.rowptr[s], f = nc + nb;
            S.front_off[s] = fo;
            fo += f * nc;
            const int h = head[(size_t)s];
            if (h < 0) {
                S.upd_off[s] = uo;
                uo += nb * nb;
            } else {
                if (chain_a[(size_t)h] < 0) {              // (the head comes first in level order: it is the lowest link)
                    const int second = next[(size_t)h];
                    chain_a[(size_t)h] = uo;
                    uo += nbof(h) * nbof(h);
                    chain_b[(size_t)h] = uo;
                    uo += nbof(second) * nbof(second);
                }
                S.upd_off[s] = (link[(size_t)s] & 1) ? chain_b[(size_t)h] : chain_a[(size_t)h];
            }
        }
        S.front_off[S.nsuper] = fo;
        S.upd_off[S.nsuper] = uo;
        S.update_store = uo;
    }
}

}  // namespace hipkkt
