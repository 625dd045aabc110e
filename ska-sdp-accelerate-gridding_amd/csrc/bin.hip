// Binning pre-pass: turn the caller's (u, v, wbin, vis) stream into tile-ordered VisRec
// records so that the tile kernels can accumulate a whole grid tile in LDS.
//
//   bin_count   : histogram of visibilities per bin (bin = w-group x grid tile), LDS-privatised; each
//                 work-group also keeps its own histogram
//   bin_scan    : exclusive scans -> bin_start[], per-group work_start[] (chunks of <=chunk vis)
//   bin_offsets : per-work-group histograms -> each work-group's first slot in every bin
//   bin_scatter : second sweep writes each visibility's VisRec into its work-group's slot range
//
// Large streams take the scatter in two levels instead (option "prepass": 0 = auto, 1 = one level,
// 2 = two levels).  A record written straight to its bin is a lone 16-byte store into one of ~10^5
// open regions, i.e. one partial-line HBM write per visibility, which is what bounds the one-level
// sweep (2.6 ms for 10^8 records against 0.45 ms for the counting sweep over the same input).
//   coarse_scatter : chunks of 8192 visibilities are counting-sorted by COARSE bin (2^k consecutive
//                    bins) in LDS and written as runs into a temporary array laid out like the final
//                    one at coarse granularity
//   fine_scatter   : segments of the temporary array (a handful of coarse bins each, so few open
//                    lines per work-group: the L2 merges them) are distributed to their bins
//
// Coordinates follow frac_coords / convgrid2 of src/Gridding.hs:126-151,212-218: the footprint
// origin is (x - gw/2, y - gh/2); a visibility none of whose taps can land inside the grid is
// dropped here (fixoutofbounds would drop every one of its taps, :883-891).
#include "common.h"

namespace gridhip {

#ifndef LIGHT_PRIO
#define LIGHT_PRIO 3  // wave priority of the kernels that run beside a tile kernel (async_prepass): they finish in
                      // 5.8 ms instead of 9.8 ms, the tile kernel loses the same 1.6 ms either way
#endif

// streaming accesses of the kernels that run beside a tile kernel carry the non-temporal hint: they should not
// displace the kernel table (L2, Infinity Cache) the tile kernel lives on
typedef int nt_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 ld_nt(const int4 *p)
{
    const nt_i4 a = __builtin_nontemporal_load(reinterpret_cast<const nt_i4 *>(p));
    return make_int4(a.x, a.y, a.z, a.w);
}
__device__ __forceinline__ void st_nt(int4 *p, int4 v)
{
    nt_i4 a;
    a.x = v.x;
    a.y = v.y;
    a.z = v.z;
    a.w = v.w;
    __builtin_nontemporal_store(a, reinterpret_cast<nt_i4 *>(p));
}

struct BinOut {
    int32_t bin;  // -1: no tap in the grid, -2: wbin outside [0,W)
    int32_t lxy, kslice;
};

__device__ __forceinline__ BinOut vis_bin(const Geom &g, double pu, double pv, int64_t wb, int64_t k)
{
    BinOut o;
    int64_t x, y;
    int32_t xf, yf;
    frac_coord_dev(g.Wd, g.Q, pu, &x, &xf);
    frac_coord_dev(g.H, g.Q, pv, &y, &yf);
    const int64_t x0 = x - g.gw / 2, y0 = y - g.gh / 2;
    // NaN coordinates compare false everywhere below and are dropped by the first test
    if (!(pu == pu) || !(pv == pv) || x0 <= -(int64_t)g.gw || x0 >= g.Wd || y0 <= -(int64_t)g.gh ||
        y0 >= g.H) {
        o.bin = -1;
        o.lxy = 0;
        o.kslice = 0;
        return o;
    }
    if (wb < 0 || wb >= g.W) {
        o.bin = -2;
        o.lxy = 0;
        o.kslice = 0;
        return o;
    }
    const int32_t X = (int32_t)x0 + g.offx, Y = (int32_t)y0 + g.offy;
    const int32_t tx = X >> g.tshift, ty = Y >> g.tshift;  // X, Y >= 0 by construction of offx/offy
    const int32_t lx = X & (g.T - 1), ly = Y & (g.T - 1);
    const int32_t grp = ((int32_t)wb * g.ngroups) / g.W;  // 32-bit: W * ngroups < 2^31
    o.bin = grp * g.ntiles + ty * g.ntx + tx;
    o.lxy = (ly << 16) | lx;
    o.kslice = g.per_vis ? (int32_t)k : ((int32_t)wb * g.Q + yf) * g.Q + xf;
    return o;
}

// Each block owns one contiguous slice of the stream (the same slice in both sweeps).
__device__ __forceinline__ void block_range(int64_t n, int64_t *lo, int64_t *hi)
{
    int64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + 255) & ~(int64_t)255;
    *lo = (int64_t)blockIdx.x * per;
    *hi = *lo + per < n ? *lo + per : n;
    if (*lo > n) *lo = n;
}

template <bool LDS_HIST>
__global__ void __launch_bounds__(1024) bin_count_kernel(Geom g, int64_t n, const double *__restrict__ u,
                                                         const double *__restrict__ v, int64_t stride,
                                                         const int64_t *__restrict__ wbin,
                                                         int32_t *__restrict__ bin_count,
                                                         int32_t *__restrict__ block_hist,
                                                         int32_t *__restrict__ scalars, int bin_lo, int bin_hi)
{
    // LDS_HIST: this launch handles the bins [bin_lo, bin_hi) only (a window that fits in LDS); grids
    // with more bins than that are covered by several launches, each a full sweep of the stream.
    extern __shared__ int32_t hist[];
    const int nwin = bin_hi - bin_lo;
    // 512-thread launches run beside a tile kernel (async_prepass): few instructions, all of them feeding memory
    // requests, so they go first at the SIMD's arbiter (which otherwise favours the tile kernel's older waves)
    if (LIGHT_PRIO && blockDim.x == 512) __builtin_amdgcn_s_setprio(LIGHT_PRIO);
    if (LDS_HIST) {
        for (int i = threadIdx.x; i < nwin; i += blockDim.x) hist[i] = 0;
        __syncthreads();
    }
    int64_t lo, hi;
    block_range(n, &lo, &hi);
    int dropped = 0;
    for (int64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) {
        BinOut b = vis_bin(g, u[k * stride], v[k * stride], wbin ? wbin[k] : 0, k);
        if (b.bin >= 0) {
            if (LDS_HIST) {
                if (b.bin >= bin_lo && b.bin < bin_hi) atomicAdd(&hist[b.bin - bin_lo], 1);
            } else
                atomicAdd(&bin_count[b.bin], 1);
        } else if (b.bin == -2)
            ++dropped;
    }
    if (dropped && bin_lo == 0) atomicAdd(&scalars[0], dropped);
    if (LDS_HIST) {
        __syncthreads();
        // keep this block's histogram: bin_offsets_kernel turns it into the block's first slot per
        // bin, so the scatter pass needs neither a recount nor slot-reservation atomics
        int32_t *mine = block_hist ? block_hist + (size_t)blockIdx.x * g.nbins + bin_lo : nullptr;
        for (int i = threadIdx.x; i < nwin; i += blockDim.x) {
            int c = hist[i];
            if (mine) mine[i] = c;  // (the two-level scatter reserves its slots differently)
            if (c) atomicAdd(&bin_count[bin_lo + i], c);
        }
    }
}

// block_hist[b][bin] (count) -> first slot of block b inside bin: bin_start[bin] + counts of the
// blocks before it.  One thread per bin, coalesced across bins.
__global__ void __launch_bounds__(256) bin_offsets_kernel(int nbins, int nblocks, const int32_t *__restrict__ bin_start,
                                                          int32_t *__restrict__ block_hist)
{
    const int bin = blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= nbins) return;
    int run = bin_start[bin];
    for (int b = 0; b < nblocks; ++b) {
        int32_t *p = block_hist + (size_t)b * nbins + bin;
        const int c = *p;
        *p = run;
        run += c;
    }
}

// Single work-group scans (nbins is at most a few hundred thousand).
// bin_start  : exclusive scan of bin_count over all bins (group-major order)
// work_start : per group, exclusive scan of ceil(count/chunk) over that group's tiles
// Each trip covers 4 x NT entries, four consecutive ones per thread (a wave reads 1 KB contiguous), with a
// wave-level scan and one LDS exchange of the 16 wave totals.
template <int NT>
__device__ __forceinline__ int block_exclusive_scan(int sum, int32_t *wtot, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();  // wtot free again
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = incl - sum, all = 0;
    for (int w = 0; w < NT / 64; ++w) {
        const int t = wtot[w];
        if (w < wave) base += t;
        all += t;
    }
    *total = all;
    return base;
}

template <int NT>
__global__ void __launch_bounds__(NT) bin_scan_kernel(Geom g, const int32_t *__restrict__ bin_count,
                                                        int32_t *__restrict__ bin_start,
                                                        int32_t *__restrict__ work_start,
                                                        int32_t *__restrict__ cursor)
{
    __shared__ int32_t wtot[16];
    const int tid = threadIdx.x;
    // ---- bin_start over all bins
    {
        int carry = 0;
        for (int base = 0; base < g.nbins; base += 4 * NT) {
            const int i0 = base + tid * 4;
            int c[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = i0 + q < g.nbins ? bin_count[i0 + q] : 0;
            int total;
            int acc = carry + block_exclusive_scan<NT>(c[0] + c[1] + c[2] + c[3], wtot, &total);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i0 + q < g.nbins) {
                    bin_start[i0 + q] = acc;
                    cursor[i0 + q] = 0;
                }
                acc += c[q];
            }
            carry += total;
        }
        if (tid == 0) bin_start[g.nbins] = carry;
    }
    // ---- work_start per group
    for (int grp = 0; grp < g.ngroups; ++grp) {
        const int32_t *cnt = bin_count + (size_t)grp * g.ntiles;
        int32_t *ws = work_start + (size_t)grp * (g.ntiles + 1);
        int carry = 0;
        for (int base = 0; base < g.ntiles; base += 4 * NT) {
            const int i0 = base + tid * 4;
            int c[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = i0 + q < g.ntiles ? (cnt[i0 + q] + g.chunk - 1) / g.chunk : 0;
            int total;
            int acc = carry + block_exclusive_scan<NT>(c[0] + c[1] + c[2] + c[3], wtot, &total);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i0 + q < g.ntiles) ws[i0 + q] = acc;
                acc += c[q];
            }
            carry += total;
        }
        if (tid == 0) ws[g.ntiles] = carry;
    }
}

template <bool LDS_HIST>
__global__ void __launch_bounds__(1024) bin_scatter_kernel(Geom g, int64_t n, const double *__restrict__ u,
                                                           const double *__restrict__ v, int64_t stride,
                                                           const int64_t *__restrict__ wbin,
                                                           const int32_t *__restrict__ bin_start,
                                                           int32_t *__restrict__ cursor,
                                                           const int32_t *__restrict__ block_hist,
                                                           VisRec *__restrict__ recs, int bin_lo, int bin_hi)
{
    extern __shared__ int32_t hist[];
    int64_t lo, hi;
    block_range(n, &lo, &hi);
    if (LDS_HIST) {
        // hist[bin] = this block's next free slot in the bin (its range was fixed by bin_offsets_kernel)
        const int32_t *mine = block_hist + (size_t)blockIdx.x * g.nbins + bin_lo;
        for (int i = threadIdx.x; i < bin_hi - bin_lo; i += blockDim.x) hist[i] = mine[i];
        __syncthreads();
    }
    // write the records
    for (int64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) {
        BinOut b = vis_bin(g, u[k * stride], v[k * stride], wbin ? wbin[k] : 0, k);
        if (b.bin < 0) continue;
        if (LDS_HIST && (b.bin < bin_lo || b.bin >= bin_hi)) continue;
        int slot;
        if (LDS_HIST)
            slot = atomicAdd(&hist[b.bin - bin_lo], 1);
        else
            slot = bin_start[b.bin] + atomicAdd(&cursor[b.bin], 1);
        VisRec r;
        r.lxy = b.lxy;
        r.kslice = b.kslice;
        r.orig = (int32_t)k;
        r.pad = 0;
        *reinterpret_cast<int4 *>(recs + slot) = *reinterpret_cast<const int4 *>(&r);  // one 16-B store
    }
}

// ---- pre-pass beside a tile kernel (async_prepass) -------------------------------------------
// The coordinate arithmetic (fp64 floor / round / conversions, ~100 instructions per visibility) is what a
// co-resident pre-pass costs the tile kernel most - issue slots, not bandwidth - and the LDS the tile kernel
// leaves (44 KB) holds a quarter of the bin histogram.  So this form does the arithmetic ONCE, leaving unbinned
// records and a compact array of bin numbers; the histogram is then counted from the bin numbers in four
// windows (4 x 0.4 GB, no arithmetic) and the coarse scatter reads the records instead of the coordinates.
__global__ void __launch_bounds__(512) light_records_kernel(Geom g, int64_t n, const double *__restrict__ u,
                                                            const double *__restrict__ v, int64_t stride,
                                                            const int64_t *__restrict__ wbin,
                                                            VisRec *__restrict__ raw, int32_t *__restrict__ bins,
                                                            int32_t *__restrict__ scalars)
{
    if (LIGHT_PRIO) __builtin_amdgcn_s_setprio(LIGHT_PRIO);
    int dropped = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const BinOut b = vis_bin(g, __builtin_nontemporal_load(u + k * stride), __builtin_nontemporal_load(v + k * stride),
                                 wbin ? __builtin_nontemporal_load(wbin + k) : 0, k);
        if (b.bin == -2) ++dropped;
        st_nt(reinterpret_cast<int4 *>(raw + k), make_int4(b.lxy, b.kslice, (int32_t)k, b.bin));
        __builtin_nontemporal_store(b.bin, bins + k);
    }
    if (dropped) atomicAdd(&scalars[0], dropped);
}

__global__ void __launch_bounds__(512) light_count_kernel(int64_t n, const int32_t *__restrict__ bins,
                                                          int32_t *__restrict__ bin_count, int bin_lo, int bin_hi)
{
    extern __shared__ int32_t hist[];
    if (LIGHT_PRIO) __builtin_amdgcn_s_setprio(LIGHT_PRIO);
    const int nwin = bin_hi - bin_lo;
    for (int i = threadIdx.x; i < nwin; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int64_t n4 = n / 4;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x) {
        const int4 b = ld_nt(reinterpret_cast<const int4 *>(bins) + k);
        if (b.x >= bin_lo && b.x < bin_hi) atomicAdd(&hist[b.x - bin_lo], 1);
        if (b.y >= bin_lo && b.y < bin_hi) atomicAdd(&hist[b.y - bin_lo], 1);
        if (b.z >= bin_lo && b.z < bin_hi) atomicAdd(&hist[b.z - bin_lo], 1);
        if (b.w >= bin_lo && b.w < bin_hi) atomicAdd(&hist[b.w - bin_lo], 1);
    }
    if (blockIdx.x == 0)
        for (int64_t k = n4 * 4 + threadIdx.x; k < n; k += blockDim.x) {
            const int b = bins[k];
            if (b >= bin_lo && b < bin_hi) atomicAdd(&hist[b - bin_lo], 1);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < nwin; i += blockDim.x) {
        const int c = hist[i];
        if (c) atomicAdd(&bin_count[bin_lo + i], c);
    }
}

// Coarse-bin counts of every work-group's slice of the stream (the slices of coarse_scatter_kernel), and their
// conversion into each work-group's first slot per coarse bin.  With these the coarse scatter beside a tile kernel
// reserves nothing at run time: its reservations would be ~25 M returning atomics on ~500 addresses, and hot-spot
// atomics in L2 are what a co-resident kernel must not do - 24 M of them slow the tile kernel from 14 to 20 ms,
// while the same number spread over 34 K addresses, or 17 GB of plain copies, cost it nothing
// (tools/coresidency_probe.py).
__global__ void __launch_bounds__(512) light_coarse_count_kernel(int64_t n, const int32_t *__restrict__ bins, int shift,
                                                                 int ncoarse, int32_t *__restrict__ wcnt)
{
    extern __shared__ int32_t hist[];
    if (LIGHT_PRIO) __builtin_amdgcn_s_setprio(LIGHT_PRIO);
    for (int i = threadIdx.x; i < ncoarse; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    int64_t lo, hi;
    block_range(n, &lo, &hi);
    for (int64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) {
        const int b = __builtin_nontemporal_load(bins + k);
        if (b >= 0) atomicAdd(&hist[b >> shift], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncoarse; i += blockDim.x) wcnt[(size_t)blockIdx.x * ncoarse + i] = hist[i];
}

// wcnt[w][c] (count) -> first slot of work-group w inside coarse bin c's region.  One thread per coarse bin.
__global__ void __launch_bounds__(256) light_coarse_offsets_kernel(int ncoarse, int nwg, int shift,
                                                                   const int32_t *__restrict__ bin_start,
                                                                   int32_t *__restrict__ wcnt)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncoarse) return;
    int run = bin_start[c << shift];
    for (int w = 0; w < nwg; ++w) {
        int32_t *p = wcnt + (size_t)w * ncoarse + c;
        const int cnt = *p;
        *p = run;
        run += cnt;
    }
}

// ---- two-level scatter ----------------------------------------------------------------------
// Both levels exist in two sizes: <1024 threads, 8192-record chunks> (128 KB of records in LDS, one work-group per
// CU; 4096-record chunks at two per CU measured 0.1 ms slower) for a pre-pass that has the chip to itself, and <512, 2048> (37-41 KB of LDS, 64 registers) for one that runs
// on a side stream BESIDE the previous call's tile kernel, whose persistent work-groups leave 44 KB of LDS, half
// the wave slots and a quarter of the registers of every CU free (option "async_prepass").

// In-place exclusive scan of hist[0..nent), nent <= 1024, by NT threads (consecutive entries per thread);
// reserve(e, count, base) is called for every non-empty entry; wtot[16] receives the total.
template <int NT, typename F>
__device__ __forceinline__ void scan_entries(int32_t *hist, int nent, int32_t *wtot, F &&reserve)
{
    constexpr int EPT = 1024 / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c[EPT], sum = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid * EPT + k;
        c[k] = e < nent ? hist[e] : 0;
        sum += c[k];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid * EPT + k;
        if (e < nent) {
            hist[e] = base;
            if (c[k]) reserve(e, c[k], base);
        }
        base += c[k];
    }
    if (tid == NT - 1) wtot[16] = base;
}

// Level 1.  tmp is laid out like the final record array at coarse granularity: coarse bin c owns
// [bin_start[c << shift], bin_start[min((c + 1) << shift, nbins)]).  Each chunk reserves, per coarse bin, a
// contiguous range there (one global atomic per chunk and non-empty coarse bin) and writes its records as
// runs; the record's spare word carries its bin for level 2.
template <int NT, int CHUNK, bool FROM_RECS = false>
__global__ void __launch_bounds__(NT, (NT == 512 ? 8 : 4)) coarse_scatter_kernel(Geom g, int64_t n, const double *__restrict__ u,
                                                            const double *__restrict__ v, int64_t stride,
                                                            const int64_t *__restrict__ wbin,
                                                            const int32_t *__restrict__ bin_start,
                                                            int32_t *__restrict__ ccur, int shift, int ncoarse,
                                                            VisRec *__restrict__ tmp, const VisRec *__restrict__ raw,
                                                            const int32_t *__restrict__ woff)
{
    extern __shared__ int32_t smem[];
    VisRec *sorted = reinterpret_cast<VisRec *>(smem);  // [CHUNK]
    int32_t *hist = smem + CHUNK * 4;                   // [ncoarse]: count, then the coarse bin's first slot in `sorted`
    int32_t *gbase = hist + ncoarse;                    // [ncoarse]: tmp position of sorted[0] if it were in this bin
    int32_t *wtot = gbase + ncoarse;                    // [16] per-wave totals of the scan, [16] = chunk total
    int32_t *lcur = wtot + 32;                          // [ncoarse] (FROM_RECS): this work-group's next slot per coarse bin
    const int tid = threadIdx.x;
    constexpr int PER = CHUNK / NT;
    if (LIGHT_PRIO && NT == 512) __builtin_amdgcn_s_setprio(LIGHT_PRIO);  // beside a tile kernel: see bin_count_kernel
    int64_t lo, hi;
    block_range(n, &lo, &hi);
    if (FROM_RECS)
        for (int i = tid; i < ncoarse; i += NT) lcur[i] = woff[(size_t)blockIdx.x * ncoarse + i];
    for (int64_t c0 = lo; c0 < hi; c0 += CHUNK) {
        for (int i = tid; i < ncoarse; i += NT) hist[i] = 0;
        __syncthreads();
        BinOut b[PER];
        int rank[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t k = c0 + q * NT + tid;
            b[q].bin = -1;
            if (k < hi) {
                if (FROM_RECS) {  // (light_records_kernel has done the arithmetic)
                    const int4 r = ld_nt(reinterpret_cast<const int4 *>(raw + k));
                    b[q].lxy = r.x;
                    b[q].kslice = r.y;
                    b[q].bin = r.w;
                } else
                    b[q] = vis_bin(g, u[k * stride], v[k * stride], wbin ? wbin[k] : 0, k);
            }
        }
#pragma unroll
        for (int q = 0; q < PER; ++q)
            rank[q] = b[q].bin >= 0 ? atomicAdd(&hist[b[q].bin >> shift], 1) : 0;
        __syncthreads();
        // exclusive scan of the counts and the global reservations
        scan_entries<NT>(hist, ncoarse, wtot, [&](int e, int c, int base) {
            if (FROM_RECS) {  // slots fixed beforehand (light_coarse_offsets_kernel): no atomics beside a tile kernel
                const int first = lcur[e];
                lcur[e] = first + c;
                gbase[e] = first - base;
            } else
                gbase[e] = bin_start[e << shift] + atomicAdd(&ccur[e], c) - base;
        });
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (b[q].bin < 0) continue;
            VisRec r;
            r.lxy = b[q].lxy;
            r.kslice = b[q].kslice;
            r.orig = (int32_t)(c0 + q * NT + tid);
            r.pad = b[q].bin;
            sorted[hist[b[q].bin >> shift] + rank[q]] = r;
        }
        __syncthreads();
        const int total = wtot[16];
        for (int i = tid; i < total; i += NT) {
            const int4 r = *reinterpret_cast<const int4 *>(sorted + i);
            int4 *dst = reinterpret_cast<int4 *>(tmp + gbase[r.w >> shift] + i);  // neighbouring lanes: neighbouring slots
            if (NT == 512)
                st_nt(dst, r);
            else
                *dst = r;
        }
        __syncthreads();
    }
}

// Level 2.  Work-group w takes the w-th equal share of tmp in chunks of CHUNK records.  tmp is ordered by
// coarse bin, so a chunk's bins lie between the coarse bins of its first and last record: normally one or two
// coarse bins, i.e. at most a few hundred bins.  The chunk is counting-sorted by bin in LDS exactly as level 1 sorts
// by coarse bin, each bin's range is reserved with one global atomic, and the records leave as runs.  A chunk that
// spans more than 1024 bins (very sparse regions) falls back to one global atomic per record.
// `tmp` is ordered by (bin >> in_shift); this level orders by key = bin >> kshift (kshift < in_shift; 0 = the final
// level), writing key k's records to [bin_start[k << kshift], ...) of `out`, with `cursor` (one entry per key,
// zeroed) handing out the ranges.
template <int NT, int CHUNK>
__global__ void __launch_bounds__(NT, (NT == 512 ? 8 : 4)) fine_scatter_kernel(Geom g, const int32_t *__restrict__ bin_start,
                                                          int32_t *__restrict__ cursor, int in_shift, int kshift,
                                                          const VisRec *__restrict__ tmp, VisRec *__restrict__ out)
{
    extern __shared__ int32_t smem[];
    VisRec *sorted = reinterpret_cast<VisRec *>(smem);  // [CHUNK]
    int32_t *hist = smem + CHUNK * 4;                   // [1024]
    int32_t *gbase = hist + 1024;                       // [1024]
    int32_t *wtot = gbase + 1024;                       // [17]
    const int tid = threadIdx.x;
    constexpr int PER = CHUNK / NT;
    if (LIGHT_PRIO && NT == 512) __builtin_amdgcn_s_setprio(LIGHT_PRIO);  // beside a tile kernel: see bin_count_kernel
    const int64_t ntot = bin_start[g.nbins];
    int64_t per = (ntot + gridDim.x - 1) / gridDim.x;
    per = (per + CHUNK - 1) / CHUNK * CHUNK;
    const int64_t lo = min((int64_t)blockIdx.x * per, ntot), hi = min(lo + per, ntot);
    const int up = in_shift - kshift;
    for (int64_t c0 = lo; c0 < hi; c0 += CHUNK) {
        const int64_t c1 = min(c0 + CHUNK, hi);
        const int cb_first = tmp[c0].pad >> in_shift, cb_last = tmp[c1 - 1].pad >> in_shift;
        const int k0 = cb_first << up, span = (cb_last - cb_first + 1) << up;  // keys k0 .. k0 + span
        if (span > 1024) {  // rare: one global atomic per record
            for (int64_t i = c0 + tid; i < c1; i += NT) {
                const int4 r = *reinterpret_cast<const int4 *>(tmp + i);
                const int k = r.w >> kshift;
                const int slot = bin_start[k << kshift] + atomicAdd(&cursor[k], 1);
                *reinterpret_cast<int4 *>(out + slot) = r;
            }
            continue;
        }
        for (int i = tid; i < span; i += NT) hist[i] = 0;
        __syncthreads();
        int4 r[PER];
        int rank[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t i = c0 + q * NT + tid;
            r[q] = make_int4(0, 0, 0, -1);
            if (i < c1) r[q] = NT == 512 ? ld_nt(reinterpret_cast<const int4 *>(tmp + i)) : *reinterpret_cast<const int4 *>(tmp + i);
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) rank[q] = r[q].w >= 0 ? atomicAdd(&hist[(r[q].w >> kshift) - k0], 1) : 0;
        __syncthreads();
        scan_entries<NT>(hist, span, wtot, [&](int e, int c, int base) {
            gbase[e] = bin_start[(k0 + e) << kshift] + atomicAdd(&cursor[k0 + e], c) - base;
        });
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (r[q].w >= 0) *reinterpret_cast<int4 *>(sorted + hist[(r[q].w >> kshift) - k0] + rank[q]) = r[q];
        __syncthreads();
        const int total = (int)(c1 - c0);
        for (int i = tid; i < total; i += NT) {
            const int4 x = *reinterpret_cast<const int4 *>(sorted + i);
            int4 *dst = reinterpret_cast<int4 *>(out + gbase[(x.w >> kshift) - k0] + i);
            if (NT == 512)
                st_nt(dst, x);
            else
                *dst = x;
        }
        __syncthreads();
    }
}

int launch_bin(gridhip_ctx *ctx, const Geom &g, int64_t n, const double *u, const double *v,
               int64_t uv_stride, const int64_t *wbin)
{
    GH_CHECK(ws_reserve(ctx, ctx->tables, tables_bytes(g)));
    GH_CHECK(ws_reserve(ctx, ctx->recs, (size_t)(n > 0 ? n : 1) * sizeof(VisRec)));
    Tables t = tables_of(ctx, g);
    int32_t *block_hist = nullptr;
    GH_CHECK_HIP(ctx, hipMemsetAsync(t.bin_count, 0, (size_t)g.nbins * sizeof(int32_t), ctx->stream));
    if (!ctx->pre_light) {
        ctx->bin_scalars = ctx->d_scalars;
        ctx->main_binned = true;  // (a later pipelined pre-pass shares recs_tmp / blockhist with this one)
    }
    t.scalars = ctx->bin_scalars;  // [0] = dropped (wbin out of range)
    GH_CHECK_HIP(ctx, hipMemsetAsync(t.scalars, 0, 16 * sizeof(int32_t), ctx->stream));

    // The histogram of one launch lives in LDS; when there are more bins than fit (large grids x 8
    // w-groups) the bins are covered in several windows, each a full sweep of the stream.  Beyond 8
    // windows the re-reads cost more than plain global atomics.
    // `light`: this pre-pass runs beside a tile kernel (async_prepass): 512-thread work-groups with at most
    // 40 KB of LDS, whatever that costs in extra sweeps - it has the tile kernel's whole duration.
    const bool light = ctx->pre_light;
    const int cap = light ? 10240 : (int)(((size_t)ctx->max_lds - 8192) / sizeof(int32_t));
    const int windows = (g.nbins + cap - 1) / cap;
    const bool lds_hist = (windows <= 8 || light) && ctx->opt.prepass != 3;  // prepass = 3: global atomics, no LDS
    const int win = lds_hist ? (g.nbins + windows - 1) / windows : g.nbins;
    const size_t hist_bytes = (size_t)win * sizeof(int32_t);
    const int threads = light ? 512 : ctx->opt.prepass == 3 ? 256 : 1024;
    // one block per CU with an LDS histogram; more, smaller slices when counting in global memory
    int blocks = lds_hist ? ctx->num_cu * (hist_bytes <= 64 * 1024 && !light ? 2 : 1) : ctx->num_cu * 8;
    // at least 16 K visibilities per work-group: below that the per-work-group histogram traffic
    // (and the serial walk over work-groups in bin_offsets_kernel) outweighs the parallelism
    int64_t need = (n + 16383) / 16384;
    if (need < 1) need = 1;
    if (blocks > need) blocks = (int)need;

    // two-level scatter for large streams (only the counting sweep needs the histogram windows)
    int shift = light ? 8 : 6;  // beside a tile kernel: fewer, longer runs per level and one level more
    while (((g.nbins + (1 << shift) - 1) >> shift) > 1024) ++shift;
    const int ncoarse = (g.nbins + (1 << shift) - 1) >> shift;
    const bool two_level =
        lds_hist && (light || ctx->opt.prepass == 2 || (ctx->opt.prepass == 0 && n >= ((int64_t)1 << 22)));
    if (two_level) {
        GH_CHECK(ws_reserve(ctx, ctx->recs_tmp, (size_t)(n > 0 ? n : 1) * sizeof(VisRec)));
        // coarse cursors; beside a tile kernel also every work-group's slots per coarse bin
        GH_CHECK(ws_reserve(ctx, ctx->blockhist,
                            ((size_t)(light ? ctx->num_cu * 2 + 1 : 1) * ncoarse + (light ? (size_t)g.nbins + 64 : 0)) *
                                sizeof(int32_t)));
        int32_t *ccur = (int32_t *)ctx->blockhist.ptr;
        VisRec *tmp = (VisRec *)ctx->recs_tmp.ptr;
        if (!(ctx->attr_mask & 2u)) {
            GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)bin_count_kernel<true>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
            GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)coarse_scatter_kernel<1024, 8192>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
            GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)fine_scatter_kernel<1024, 8192>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
            ctx->attr_mask |= 2u;
        }
        GH_CHECK_HIP(ctx, hipMemsetAsync(ccur, 0, (size_t)ncoarse * sizeof(int32_t), ctx->stream));
        const int chunk = light ? 2048 : 8192;
        const size_t coarse_lds = (size_t)chunk * sizeof(VisRec) + (size_t)(2 * ncoarse + 32) * sizeof(int32_t);
        const size_t fine_lds = (size_t)chunk * sizeof(VisRec) + (size_t)(2 * 1024 + 32) * sizeof(int32_t);
        int cblocks = light ? ctx->num_cu * 2 : ctx->num_cu;  // (8192-record chunks: one work-group per CU)
        int64_t cneed = (n + 4 * chunk - 1) / (4 * chunk);
        if (cblocks > cneed) cblocks = (int)(cneed < 1 ? 1 : cneed);
        if (light) {
            GH_CHECK(ws_reserve(ctx, ctx->recs_raw, (size_t)(n > 0 ? n : 1) * (sizeof(VisRec) + sizeof(int32_t)) + 256));
            VisRec *raw = (VisRec *)ctx->recs_raw.ptr;
            int32_t *bins = (int32_t *)(raw + (n > 0 ? n : 1));
            int sblocks = ctx->num_cu * 2;
            if (sblocks > need) sblocks = (int)need;
            hipLaunchKernelGGL(light_records_kernel, dim3(sblocks), dim3(512), 0, ctx->stream, g, n, u, v, uv_stride,
                               wbin, raw, bins, t.scalars);
            for (int wdw = 0; wdw < windows; ++wdw) {
                const int b_lo = wdw * win, b_hi = b_lo + win < g.nbins ? b_lo + win : g.nbins;
                hipLaunchKernelGGL(light_count_kernel, dim3(blocks), dim3(512), hist_bytes, ctx->stream, n, bins,
                                   t.bin_count, b_lo, b_hi);
            }
            hipLaunchKernelGGL(bin_scan_kernel<512>, dim3(1), dim3(512), 0, ctx->stream, g, t.bin_count, t.bin_start,
                               t.work_start, t.cursor);
            // every work-group's slots per coarse bin, fixed beforehand (no reservations at run time)
            int32_t *woff = (int32_t *)ctx->blockhist.ptr + ncoarse;
            hipLaunchKernelGGL(light_coarse_count_kernel, dim3(cblocks), dim3(512), (size_t)ncoarse * sizeof(int32_t),
                               ctx->stream, n, bins, shift, ncoarse, woff);
            hipLaunchKernelGGL(light_coarse_offsets_kernel, dim3((ncoarse + 255) / 256), dim3(256), 0, ctx->stream,
                               ncoarse, cblocks, shift, t.bin_start, woff);
            hipLaunchKernelGGL((coarse_scatter_kernel<512, 2048, true>), dim3(cblocks), dim3(512),
                               coarse_lds + (size_t)ncoarse * sizeof(int32_t), ctx->stream, g, n, u, v, uv_stride, wbin,
                               t.bin_start, ccur, shift, ncoarse, tmp, raw, woff);
            // runs shorter than 128 B are what slows a tile kernel next door (tools/coresidency_probe.py), so the
            // levels below the coarse one go in steps of 16 keys: with 2 048-record chunks every run is >= 1 KB
            const VisRec *src = tmp;
            VisRec *spare = raw;  // (its records have been consumed by the coarse level)
            for (int in_shift = shift; in_shift > 0;) {
                const int kshift = in_shift > 4 ? in_shift - 4 : 0;
                VisRec *dst = kshift == 0 ? (VisRec *)ctx->recs.ptr : spare;
                int32_t *cur = t.cursor;
                if (kshift > 0) {  // cursors of an intermediate level: behind the work-group offsets
                    cur = (int32_t *)ctx->blockhist.ptr + (size_t)(ctx->num_cu * 2 + 1) * ncoarse;
                    GH_CHECK_HIP(ctx, hipMemsetAsync(cur, 0, (size_t)((g.nbins >> kshift) + 1) * sizeof(int32_t), ctx->stream));
                }
                hipLaunchKernelGGL((fine_scatter_kernel<512, 2048>), dim3(cblocks), dim3(512), fine_lds, ctx->stream, g,
                                   t.bin_start, cur, in_shift, kshift, src, dst);
                spare = const_cast<VisRec *>(src);
                src = dst;
                in_shift = kshift;
            }
        } else {
            for (int wdw = 0; wdw < windows; ++wdw) {
                const int b_lo = wdw * win, b_hi = b_lo + win < g.nbins ? b_lo + win : g.nbins;
                hipLaunchKernelGGL(bin_count_kernel<true>, dim3(blocks), dim3(threads), hist_bytes, ctx->stream, g, n, u,
                                   v, uv_stride, wbin, t.bin_count, (int32_t *)nullptr, t.scalars, b_lo, b_hi);
            }
            hipLaunchKernelGGL(bin_scan_kernel<1024>, dim3(1), dim3(1024), 0, ctx->stream, g, t.bin_count, t.bin_start,
                               t.work_start, t.cursor);
            hipLaunchKernelGGL((coarse_scatter_kernel<1024, 8192>), dim3(cblocks), dim3(1024), coarse_lds, ctx->stream,
                               g, n, u, v, uv_stride, wbin, t.bin_start, ccur, shift, ncoarse, tmp, (const VisRec *)nullptr,
                               (const int32_t *)nullptr);
            hipLaunchKernelGGL((fine_scatter_kernel<1024, 8192>), dim3(cblocks), dim3(1024), fine_lds, ctx->stream, g,
                               t.bin_start, t.cursor, shift, 0, tmp, (VisRec *)ctx->recs.ptr);
        }
        GH_CHECK_HIP(ctx, hipGetLastError());
        return GRIDHIP_OK;
    }

    if (lds_hist) {
        GH_CHECK(ws_reserve(ctx, ctx->blockhist, (size_t)blocks * g.nbins * sizeof(int32_t)));
        block_hist = (int32_t *)ctx->blockhist.ptr;
        if (!(ctx->attr_mask & 1u)) {
            GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)bin_count_kernel<true>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
            GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)bin_scatter_kernel<true>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
            ctx->attr_mask |= 1u;
        }
        for (int wdw = 0; wdw < windows; ++wdw) {
            const int b_lo = wdw * win, b_hi = b_lo + win < g.nbins ? b_lo + win : g.nbins;
            hipLaunchKernelGGL(bin_count_kernel<true>, dim3(blocks), dim3(threads), hist_bytes, ctx->stream, g, n, u,
                               v, uv_stride, wbin, t.bin_count, block_hist, t.scalars, b_lo, b_hi);
        }
    } else {
        hipLaunchKernelGGL(bin_count_kernel<false>, dim3(blocks), dim3(threads), 0, ctx->stream, g, n, u, v,
                           uv_stride, wbin, t.bin_count, block_hist, t.scalars, 0, g.nbins);
    }
    hipLaunchKernelGGL(bin_scan_kernel<1024>, dim3(1), dim3(1024), 0, ctx->stream, g, t.bin_count, t.bin_start,
                       t.work_start, t.cursor);
    if (lds_hist) {
        hipLaunchKernelGGL(bin_offsets_kernel, dim3((g.nbins + 255) / 256), dim3(256), 0, ctx->stream, g.nbins, blocks,
                           t.bin_start, block_hist);
        for (int wdw = 0; wdw < windows; ++wdw) {
            const int b_lo = wdw * win, b_hi = b_lo + win < g.nbins ? b_lo + win : g.nbins;
            hipLaunchKernelGGL(bin_scatter_kernel<true>, dim3(blocks), dim3(threads), hist_bytes, ctx->stream, g, n,
                               u, v, uv_stride, wbin, t.bin_start, t.cursor, block_hist, (VisRec *)ctx->recs.ptr,
                               b_lo, b_hi);
        }
    } else
        hipLaunchKernelGGL(bin_scatter_kernel<false>, dim3(blocks), dim3(threads), 0, ctx->stream, g, n, u, v,
                           uv_stride, wbin, t.bin_start, t.cursor, block_hist, (VisRec *)ctx->recs.ptr, 0, g.nbins);
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

}  // namespace gridhip
