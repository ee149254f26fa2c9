// Every environment hook of the library, in ONE table (csrc/hooks.hpp; the same list as DESIGN.md section 9, kept in step by
// tests/test_abi.py).  They exist for measurements (A/B of kernel forms inside one process) and tests (forcing a path a size
// heuristic would not pick, rehearsing a failure); none changes results beyond rounding order.  A deployment that wants no
// environment-dependent behaviour builds with -DHSD_NO_ENV_HOOKS: hsd_hook() is then a constant nullptr and every default holds.
// The two ablation hooks (HSD_DEBUG, HSD_TOPK_DEBUG: wrong results on purpose) exist only in a -DHSD_DEBUG_HOOKS build.
// HSD_HOOKS_LIST=1 prints the table (name, current value, meaning) to stderr when a context is created.
#pragma once

#include <cstdlib>

enum HsdHook {
    H_FORCE_COMM,
    H_COMM_DRYRUN,
    H_CHEBY_LONGROWS,
    H_CHEBY_STORE,
    H_CHEBY_C,
    H_HEAT_SORT_LDS,
    H_HEAT_TILED,
    H_HEAT_GROUPS,
    H_HEAT_TT,
    H_HEAT_BCAP,
    H_HEAT_ACAP,
    H_HEAT_ROWLDS,
    H_KHOP_MS,
    H_KHOP_BOTTOMUP,
    H_KHOP_MS_NOPLANES,
    H_TILE,
    H_SUPERTILE,
    H_XCD_DEAL,
    H_PLAN_VERBOSE,
    H_WAVES_PER_CU,
    H_STREAM,
    H_GRAN_SORT,
    H_ROWBLOCK,
    H_SAD16,
    H_RANK,
    H_PLAN_THREADS,
    H_PLAN_THREADS_FAIL,
    H_U8,
    H_SHARD_BALANCE,
    H_KL_POINT,
    H_KL_TRIM,
    H_PREP_STAGED,
    H_DEBUG,
    H_SIDE_STREAMS,
    H_TOPK_DEBUG,
    H_TOPK_PRUNE,
    H_TOPK_GROUPS,
    H_TOPK_BAND,
    H_TOPK_ORDER,
    H_TOPK_PHASES,
    H_TOPK_GEOM,
    H_TOPK_CAP,
    H_TOPK_VERBOSE,
    H_TOPK_NOMEM,
    H_COUNT
};

struct HsdHookInfo { const char *name, *doc; };
extern const HsdHookInfo hsd_hook_table[H_COUNT];

// Hooks that make a call return WRONG results or fail on purpose (ablations for timing: kernels without their stores, without their arithmetic,
// a sweep without its candidate epilogue) are read only by a measurement build (-DHSD_DEBUG_HOOKS: `make debug` -> libhsd_amd_dbg.so, loaded
// through HSD_LIB by the A/B tools of tools_dev/); the shipped library ignores them, whatever the environment says.
constexpr bool hsd_hook_is_ablation(HsdHook h) { return h == H_DEBUG || h == H_TOPK_DEBUG; }

#ifdef HSD_NO_ENV_HOOKS
inline const char *hsd_hook(HsdHook) { return nullptr; }
#else
inline const char *hsd_hook(HsdHook h) {
#ifndef HSD_DEBUG_HOOKS
    if (hsd_hook_is_ablation(h)) return nullptr;
#endif
    return getenv(hsd_hook_table[h].name);
}
#endif
inline int hsd_hook_int(HsdHook h, int dflt) { const char *e = hsd_hook(h); return e ? atoi(e) : dflt; }
