// Every environment setting the library reads, in ONE place: names, defaults and what they do.  They are for experiments
// and tests -- none of them is part of the ABI (include/hipkkt.h: hipkkt_settings is) -- and are read ONCE per process,
// when the first handle is created (or the first host-only symbolic call is made): knobs() below.  Defaults that scale
// with the device (CU count) are 0 / -1 here and resolved per handle where they are used.
#pragma once
#include <cstdlib>

namespace hipkkt {

//   KNOB(type, field, "NAME", default)  -- value of the variable, or the default when it is not set
//   FLAG_SET(field, "NAME")             -- true when the variable exists, whatever its value
//   FLAG_ON(field, "NAME")              -- true unless the variable is set to 0
#define HIPKKT_KNOBS(KNOB, FLAG_SET, FLAG_ON)                                                                                         \
    /* ---- diagnostics */                                                                                                             \
    KNOB(int, verbose, "HIPKKT_VERBOSE", 0)                 /* 1: schedule summary, admissions, fall-back details on stderr; 2: per-launch detail */ \
    FLAG_SET(stamps, "HIPKKT_STAMPS")                       /* per-phase time stamps of the panel kernel (synchronises) */             \
    KNOB(int, top_stamps, "HIPKKT_TOP_STAMPS", 0)           /* the n-th single-column sweep prints the persistent / chained kernels' hop timings */ \
    KNOB(int, top_stamps_nr, "HIPKKT_TOP_STAMPS_NR", 1)     /* ... for sweeps of this many columns */                                  \
    FLAG_SET(dump_levels, "HIPKKT_DUMP_LEVELS")             /* host-only symbolic call: shape of every tree level */                   \
    FLAG_SET(dump_subtrees, "HIPKKT_DUMP_SUBTREES")         /* host-only symbolic call: subtrees below each cut level */               \
    /* ---- ordering and symbolic structure (host) */                                                                                  \
    KNOB(double, nd_sep_ratio, "HIPKKT_ND_SEP_RATIO", 0.10) /* a subgraph whose separator exceeds this share of it goes to AMD whole */ \
    KNOB(int, nd_sep_min, "HIPKKT_ND_SEP_MIN", 5000)        /* ... if it has more nodes than this */                                   \
    KNOB(long long, panel_cap, "HIPKKT_PANEL_CAP", -1)      /* LDS doubles of a panel's trapezoid (-1: SymbolicOptions' 19200); small values force row slices */ \
    KNOB(int, panel_max_cols, "HIPKKT_PANEL_MAX_COLS", -1)  /* widest panel (-1: 96) */                                                \
    KNOB(int, panel_slice_below, "HIPKKT_PANEL_SLICE_BELOW", -1) /* a front whose unsliced panel would be narrower than this is cut into row slices (-1: 64) */ \
    FLAG_SET(postorder_layout, "HIPKKT_POSTORDER_LAYOUT")   /* stores in supernode order instead of level order */                    \
    FLAG_ON(upd_pingpong, "HIPKKT_UPD_PINGPONG")            /* chains of panels share two update blocks */                            \
    KNOB(int, bundle_kids, "HIPKKT_BUNDLE_KIDS", 200)       /* a one-wave supernode (f <= 64) left with more children than this has them bundled into sibling supernodes (0: no bundles at all) */ \
    KNOB(int, bundle_kids_panel, "HIPKKT_BUNDLE_KIDS_PANEL", 400)  /* ... a wider one with more than this many per panel */ \
    FLAG_ON(bundle_cost, "HIPKKT_BUNDLE_COST")              /* ... and children whose update blocks dwarf the parent's front; bundles beyond one-wave size where that shrinks panel + update-block storage */ \
    /* ---- schedule (handle creation) */                                                                                              \
    KNOB(int, merge_small, "HIPKKT_MERGE_SMALL", 128)       /* up to this many one-wave fronts ride with their level's block-class launch */ \
    KNOB(int, slice_rows, "HIPKKT_SLICE_ROWS", 128)         /* row-sliced panels: rows per slice (0: as few slices as LDS allows) */   \
    FLAG_ON(slice_fit, "HIPKKT_SLICE_FIT")                  /* taller slices for a level with more slices than CUs */                  \
    KNOB(int, bs128_count, "HIPKKT_BS128_COUNT", 1024)      /* a level with at least this many fronts ... */                           \
    KNOB(int, bs128_f, "HIPKKT_BS128_F", 128)               /* ... none taller than this sweeps with 128-thread workgroups */          \
    KNOB(int, solve_tall_rows, "HIPKKT_SOLVE_TALL_ROWS", 0) /* fronts of this many rows count as too tall for the block sweep kernels (tests) */ \
    KNOB(int, tall_block_rows, "HIPKKT_TALL_BLOCK_ROWS", 256) /* rows of such a front per workgroup of the k_*_tall sweep kernels (64..4096; 25 088-row root, two kernels per direction: 1024 35.6, 512 28.6, 256 26.6, 128 28.6 ms per sweep pair; fused: 256 22.0) */ \
    FLAG_ON(dense_child, "HIPKKT_DENSE_CHILD")              /* a child whose update block is its parent's whole front is added as a dense block */ \
    KNOB(int, tile_xcd, "HIPKKT_TILE_XCD", 150)             /* launches of at least this many fronts deal a front's tiles to one XCD (0: off) */ \
    FLAG_ON(pull_leaves, "HIPKKT_PULL_LEAVES")              /* many-column sweeps: one-column leaves are pulled by their parents */     \
    FLAG_ON(packed, "HIPKKT_PACKED")                        /* packed sweep records (kernels.hpp: SolveHdr); 0: the legacy layout */    \
    KNOB(long long, packed_max_mb, "HIPKKT_PACKED_MAX_MB", 4096) /* ... unless they would exceed this many MB */                      \
    /* ---- factorisation */                                                                                                           \
    FLAG_SET(graph, "HIPKKT_GRAPH")                         /* replay both launch chains as hipGraphs (measured slower; opt-in) */      \
    FLAG_SET(no_overlap, "HIPKKT_NO_OVERLAP")               /* no side-stream W formation */                                          \
    KNOB(int, winv_tail, "HIPKKT_WINV_TAIL", 4)             /* fork points of the W-formation stream: launches before the root ... */   \
    KNOB(int, winv_early, "HIPKKT_WINV_EARLY", 1)           /* ... and before the narrow top */                                        \
    KNOB(int, winv_blocks, "HIPKKT_WINV_BLOCKS", 0)         /* grid of the side-stream W kernel (0: 3/8 of the CUs) */                  \
    FLAG_ON(winv_split, "HIPKKT_WINV_SPLIT")                /* bounded W formation: narrow supernodes on 128-thread workgroups, four times as many */ \
    FLAG_ON(winv_run_forks, "HIPKKT_WINV_RUN_FORKS")        /* W formation forked at every merged run's first launch */                \
    FLAG_ON(factor_overlap, "HIPKKT_FACTOR_OVERLAP")        /* overlap mode: the narrow top's Schur tiles beside its panels */         \
    FLAG_SET(ov_cu_mask, "HIPKKT_OV_CU_MASK")               /* CU-masked tile stream (measured and dropped) */                         \
    KNOB(long long, ov_test_limit, "HIPKKT_OV_TEST_LIMIT", 5000000) /* bound of the overlap mode's waits in 10 ns ticks (tests: 0) */  \
    KNOB(int, ov_max_fronts, "HIPKKT_OV_MAX_FRONTS", 0)     /* widest overlapped launch in panel workgroups (0: 120 per 256 CUs; never beyond CUs - 9) */ \
    KNOB(int, ov_max_tiles, "HIPKKT_OV_MAX_TILES", 1600)    /* a handle with a launch of more tiles in its overlap region stays out of the mode */ \
    KNOB(int, ov_merge, "HIPKKT_OV_MERGE", 100)             /* panel workgroups per merged panel kernel (0: a kernel per level) */      \
    KNOB(int, ov_merge_wide, "HIPKKT_OV_MERGE_WIDE", 24)    /* widest launch a merged run below the root's may hold */                 \
    KNOB(int, ov_merge_groups, "HIPKKT_OV_MERGE_GROUPS", 8) /* merged panel kernels per factorisation */                               \
    KNOB(int, schur_pipe_tiles, "HIPKKT_SCHUR_PIPE_TILES", 1600) /* outside the mode: launches of more tiles ... */                    \
    KNOB(int, schur_pipe_nc, "HIPKKT_SCHUR_PIPE_NC", 64)    /* ... or at least this deep on average pipeline their chunk loop */        \
    /* ---- sweeps */                                                                                                                  \
    FLAG_SET(no_top, "HIPKKT_NO_TOP")                       /* no persistent top-of-tree kernel */                                     \
    KNOB(int, top_tall, "HIPKKT_TOP_TALL", -1)              /* 0: the persistent kernel's 512-thread build */                          \
    KNOB(int, top_cap, "HIPKKT_TOP_CAP", 1 << 30)           /* cap of its grid (tests force tiny grids) */                             \
    KNOB(double, top_mult, "HIPKKT_TOP_MULT", 1.5)          /* fronts per level it takes, as a multiple of the grid */                  \
    KNOB(long long, top_test_limit, "HIPKKT_TOP_TEST_LIMIT", 5000000) /* bound of the persistent / chained kernels' waits in 10 ns ticks (tests: 0) */ \
    FLAG_SET(no_level_merge, "HIPKKT_NO_LEVEL_MERGE")       /* a level's block-class and one-wave launches stay two launches */        \
    KNOB(int, solve_slice_kb, "HIPKKT_SOLVE_SLICE_KB", 80)  /* (front, slice) kernel: slice size ... */                                \
    KNOB(int, solve_slice_max, "HIPKKT_SOLVE_SLICE_MAX", 64) /* ... slices per front (1..64; 16 until r04: a 14 154-row front's 11 MB want more) ... */                                    \
    KNOB(long long, solve_slice_from, "HIPKKT_SOLVE_SLICE_FROM", -1) /* ... and the W size in KB from which a front is sliced (-1: 4.5 slices' worth) */ \
    FLAG_ON(chain, "HIPKKT_CHAIN")                          /* chained launches (chain_kernels.hip) */                                 \
    FLAG_ON(chain_top, "HIPKKT_CHAIN_TOP")                  /* ... below the persistent kernel's set; 0: up to the root instead of it */ \
    KNOB(int, chain_max, "HIPKKT_CHAIN_MAX", 640)           /* widest chained launch in workgroups */                                  \
    FLAG_SET(test_publish_fail, "HIPKKT_TEST_PUBLISH_FAIL") /* tests: a handle's first published status record counts as not arrived (level C falls back to copies) */ \
    KNOB(int, multi_vec, "HIPKKT_MULTI_VEC", 4)             /* many-column kernels: columns per lane (1, 2, 4) */                       \
    KNOB(int, multi_ct, "HIPKKT_MULTI_CT", 0)               /* 1: 16 instead of 32 columns per block-kernel workgroup */

struct Knobs {
#define HIPKKT_K_FIELD(type, field, name, dflt) type field = dflt;
#define HIPKKT_K_FLAG(field, name) bool field = false;
#define HIPKKT_K_FLAGON(field, name) bool field = true;
    HIPKKT_KNOBS(HIPKKT_K_FIELD, HIPKKT_K_FLAG, HIPKKT_K_FLAGON)
#undef HIPKKT_K_FIELD
#undef HIPKKT_K_FLAG
#undef HIPKKT_K_FLAGON
    static void get(int& v, const char* e) { v = std::atoi(e); }
    static void get(long long& v, const char* e) { v = std::atoll(e); }
    static void get(double& v, const char* e) { v = std::atof(e); }
    Knobs()
    {
#define HIPKKT_K_READ(type, field, name, dflt) if (const char* e_ = std::getenv(name)) get(field, e_);
#define HIPKKT_K_READF(field, name) field = std::getenv(name) != nullptr;
#define HIPKKT_K_READON(field, name) if (const char* e_ = std::getenv(name)) field = std::atoi(e_) != 0;
        HIPKKT_KNOBS(HIPKKT_K_READ, HIPKKT_K_READF, HIPKKT_K_READON)
#undef HIPKKT_K_READ
#undef HIPKKT_K_READF
#undef HIPKKT_K_READON
        if (std::getenv("HIPKKT_VERBOSE") && verbose < 1) verbose = 1;      // (set at all means on)
    }
};
inline const Knobs& knobs()
{
    static const Knobs k;
    return k;
}

}  // namespace hipkkt
