// Shared by the tile kernels: work-item lookup and record access.
#pragma once
#include "common.h"

namespace gridhip {

struct WorkItem {
    int tile, v_lo, v_hi;
};

// Map blockIdx -> work item.  Work items of w-group g are the blocks with blockIdx % ngroups
// == g, so with the dispatcher's round-robin over the 8 XCDs a group's kernel planes stay in
// one XCD's L2 (speed only; any placement is correct).
__device__ __forceinline__ bool find_work(const Geom &g, const int32_t *__restrict__ bin_start,
                                          const int32_t *__restrict__ work_start, WorkItem *w)
{
    const int grp = blockIdx.x % g.ngroups;
    const int k = blockIdx.x / g.ngroups;
    const int32_t *ws = work_start + (size_t)grp * (g.ntiles + 1);
    if (k >= ws[g.ntiles]) return false;
    int lo = 0, hi = g.ntiles;  // largest t with ws[t] <= k
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ws[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    const int bin = grp * g.ntiles + lo;
    const int c = k - ws[lo], nch = ws[lo + 1] - ws[lo];
    const int b0 = bin_start[bin], cnt = bin_start[bin + 1] - b0;
    w->tile = lo;
    w->v_lo = b0 + (int)(((int64_t)cnt * c) / nch);
    w->v_hi = b0 + (int)(((int64_t)cnt * (c + 1)) / nch);
    return true;
}

// All lanes of the wave read the same 16 bytes (one broadcast transaction).  Vector loads on
// purpose: scalar loads share the lgkmcnt counter with the LDS atomics and would make every
// record fetch wait for the wave's outstanding ds_add_f64s.
__device__ __forceinline__ VisRec load_rec(const VisRec *__restrict__ recs, int idx)
{
    const int4 a = *reinterpret_cast<const int4 *>(recs + idx);
    VisRec r;
    r.lxy = a.x;
    r.kslice = a.y;
    r.orig = a.z;
    r.pad = 0;
    return r;
}

// upper bound on the number of work items (grid size of the tile kernels)
static inline int work_blocks(const Geom &g, int64_t n)
{
    // per group: every tile may add one partial chunk
    int64_t per_group = n / g.chunk + g.ntiles + 1;
    return (int)(per_group * g.ngroups);
}

template <typename K>
static int raise_lds(gridhip_ctx *ctx, K kernel, uint32_t bit)
{
    if (ctx->attr_mask & bit) return GRIDHIP_OK;
    GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          ctx->max_lds));
    ctx->attr_mask |= bit;
    return GRIDHIP_OK;
}

}  // namespace gridhip
