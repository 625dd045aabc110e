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
__device__ __forceinline__ bool find_work_at(const Geom &g, const int32_t *__restrict__ bin_start,
                                             const int32_t *__restrict__ work_start, int grp, int k, WorkItem *w)
{
    const int32_t *ws = work_start + (size_t)grp * (g.ntiles + 1);
    // (a group never has more items than every tile's partial chunk plus the full chunks the records make: whatever
    // the table says, the queue ends)
    if (k >= ws[g.ntiles] || k > g.nrec / g.chunk + g.ntiles) return false;
    int lo = 0, hi = g.ntiles;  // largest t with ws[t] <= k
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ws[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    const int bin = grp * g.ntiles + lo;
    const int nch = max(ws[lo + 1] - ws[lo], 1), c = min(max(k - ws[lo], 0), nch - 1);
    // (the tables are the pre-pass's own, but nothing read from memory becomes an index before it is in range)
    const int b0 = min(max(bin_start[bin], 0), g.nrec), cnt = min(max(bin_start[bin + 1] - b0, 0), g.nrec - b0);
    w->tile = lo;
    w->v_lo = b0 + (int)(((int64_t)cnt * c) / nch);
    w->v_hi = b0 + (int)(((int64_t)cnt * (c + 1)) / nch);
    return true;
}

__device__ __forceinline__ bool find_work(const Geom &g, const int32_t *__restrict__ bin_start,
                                          const int32_t *__restrict__ work_start, WorkItem *w)
{
    return find_work_at(g, bin_start, work_start, blockIdx.x % g.ngroups, blockIdx.x / g.ngroups, w);
}

// All lanes of the wave read the same 8 bytes (one broadcast transaction), or consecutive records (coalesced).
// Vector loads on purpose: scalar loads share the lgkmcnt counter with the LDS atomics and would make every
// record fetch wait for the wave's outstanding ds_add_f64s.
// Nothing read from a record becomes an address before it has been brought into range: a record slot the pre-pass
// never wrote (only possible when the caller's arrays change between its sweeps) holds stale bytes, and those may
// cost a wrong sum but never a wild access.
__device__ __forceinline__ VisRec load_rec(const RecWord *__restrict__ recs, int idx, const Geom &g, bool *clamped = nullptr)
{
    // (streamed once or twice and never again: non-temporal, so that the records do not push kernel taps out of L2)
    const RecWord w = __builtin_nontemporal_load(recs + idx);
    const int sk = g.ob, sx = g.ob + g.kb;
    const int32_t o = (int32_t)(w & ((1ull << sk) - 1)), k = (int32_t)((w >> sk) & ((1ull << g.kb) - 1));
    const int32_t lx = (int32_t)(w >> sx) & 0x7f, ly = (int32_t)(w >> (sx + 7)) & 0x7f;
    VisRec r;
    r.lxy = (min(ly, g.Ty - 1) << 16) | min(lx, g.Tx - 1);
    r.kslice = min(k, g.nslices - 1);
    r.orig = min(o, g.nvis - 1);
    // (bits above the fields must be clear; the fields may fill the word exactly, and a shift by 64 is undefined)
    if (clamped) *clamped = r.lxy != ((ly << 16) | lx) || r.kslice != k || r.orig != o || (sx + 14 < 64 && (w >> (sx + 14)) != 0);
    return r;
}

// A 16-byte load with the non-temporal hint: the gather of visibility values, each used once (measured together with
// the records' hint when the sorter still did the gather: tile kernel 11.5 -> 11.3 ms, HBM traffic 34.5 -> 31.8 GB per
// launch, TCC hit rate 53 -> 60 %; the same hint on the walkers' reads of the sorted list measured slower, on its
// stores 4 % slower)
typedef double dvec2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_nt(const double2 *p)
{
    const dvec2_t v = __builtin_nontemporal_load(reinterpret_cast<const dvec2_t *>(p));
    return make_double2(v.x, v.y);
}

// The slot of this lane's record among the records of its key in the LDS histogram `hist` (key < 0: none, returns 0).
// Neighbouring lanes hold neighbouring records of the stream, and real streams come in runs - the channels of one
// baseline and time fall into the same bin - so a run of equal keys in neighbouring lanes is reserved by its first lane
// with ONE atomic and the others take their place behind it: 64 returning atomics on one LDS address serialise (a
// stream of 64-sample tracks took the pre-pass's two scatter levels from 0.36 + 0.47 to 0.54 + 0.65 ms; used in bin.hip).  On a random stream every
// lane is the head of its own run: the same atomics as before plus a dozen integer instructions.
__device__ __forceinline__ int run_rank(int32_t *hist, int key)
{
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1);
    const bool valid = key >= 0, head = valid && (lane == 0 || prev != key);
    const unsigned long long hm = __ballot(head), vm = __ballot(valid);
    const unsigned long long upto = (2ull << lane) - 1;            // lanes 0 .. lane (lane 63: all)
    const unsigned long long below = hm & upto;                    // a valid lane has its run's head at or below it
    const int hl = valid && below ? 63 - __clzll((long long)below) : lane;
    const unsigned long long stops = (hm | ~vm) & ~upto;           // where the run ends: the next head, or the next lane without a key
    const int end = stops ? __ffsll((long long)stops) - 1 : 64;
    int base = 0;
    if (head) base = atomicAdd(&hist[key], end - lane);
    base = __shfl(base, hl);
    return valid ? base + (lane - hl) : 0;
}

// Sum of x over the 64 lanes, valid in lane 63.  Data-parallel-primitive moves only (no LDS
// traffic, unlike __shfl which goes through ds_bpermute): an inclusive scan inside each row of 16
// lanes (row_shr 1,2,4,8; lanes without a source add 0), then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_step(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return x + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_sum_lane63(double x)
{
    x = dpp_add_step<0x111, 0xf>(x);  // row_shr:1
    x = dpp_add_step<0x112, 0xf>(x);  // row_shr:2
    x = dpp_add_step<0x114, 0xf>(x);  // row_shr:4
    x = dpp_add_step<0x118, 0xf>(x);  // row_shr:8
    x = dpp_add_step<0x142, 0xa>(x);  // row_bcast:15 -> rows 1, 3
    x = dpp_add_step<0x143, 0xc>(x);  // row_bcast:31 -> rows 2, 3
    return x;
}

// Sums of four values over the 64 lanes at once: x[r] summed over all lanes arrives in lane 16*r + 15.
// v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / rows between two registers, so each
// level adds two registers into one that carries twice as many different sums: 21 VALU instructions for four
// sums instead of 4 x 18 with one reduction each.
__device__ __forceinline__ void swap32_f64(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap16_f64(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double wave_sum4_rows(double x0, double x1, double x2, double x3)
{
    swap32_f64(x0, x2);  // x0 = [x0 low half | x2 low half], x2 = [x0 high half | x2 high half]
    swap32_f64(x1, x3);
    double y02 = x0 + x2, y13 = x1 + x3;  // low half: sums 0 / 1, high half: sums 2 / 3
    swap16_f64(y02, y13);                 // odd rows of y02 <-> even rows of y13
    double z = y02 + y13;                 // row r: sum r, spread over its 16 lanes
    z = dpp_add_step<0x111, 0xf>(z);      // row_shr:1
    z = dpp_add_step<0x112, 0xf>(z);      // row_shr:2
    z = dpp_add_step<0x114, 0xf>(z);      // row_shr:4
    z = dpp_add_step<0x118, 0xf>(z);      // row_shr:8
    return z;
}

// upper bound on the number of work items (grid size of the tile kernels)
static inline int work_blocks(const Geom &g, int64_t n)
{
    // per group: every tile may add one partial chunk
    int64_t per_group = n / g.chunk + g.ntiles + 1;
    return (int)(per_group * g.ngroups);
}

// Zeroes up to three int32 ranges in one launch.  Kernels, not hipMemsetAsync: what a gridding call enqueues is then
// nothing but kernel launches, which a caller may capture into a HIP graph and replay.
__global__ void __launch_bounds__(256) clear_ints_kernel(int32_t *a, int na, int32_t *b, int nb, int32_t *c, int nc);
static inline void launch_clear(gridhip_ctx *ctx, int32_t *a, int na, int32_t *b = nullptr, int nb = 0, int32_t *c = nullptr,
                                int nc = 0)
{
    const int most = na > nb ? (na > nc ? na : nc) : (nb > nc ? nb : nc);
    int blocks = (most + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(clear_ints_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a, na, b, nb, c, nc);
}

template <typename K>
static int raise_lds(gridhip_ctx *ctx, K kernel)
{
    const void *f = (const void *)kernel;
    if (ctx->lds_raised.count(f)) return GRIDHIP_OK;
    GH_CHECK_HIP(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
    ctx->lds_raised.insert(f);
    return GRIDHIP_OK;
}

}  // namespace gridhip
