// Binning pre-pass: turn the caller's (u, v, wbin) stream into tile-ordered records (RecWord) so that
// the tile kernels can accumulate a whole grid tile in LDS.
//
//   bin_count   : the ONLY sweep over the caller's stream (24 B per visibility) and the only place the fp64
//                 coordinate arithmetic is done.  Histogram of visibilities per bin (bin = w-group x grid tile),
//                 LDS-privatised; for large streams it also leaves an 8-byte PRE-RECORD per visibility
//                 (bin | lx | ly | kslice) for the scatter
//   bin_scan    : exclusive scans -> bin_start[], per-group work_start[] (chunks of <= chunk visibilities)
//
// Small streams (< 2^22 visibilities) then write each record straight to its bin:
//   bin_offsets : per-work-group histograms -> each work-group's first slot in every bin
//   bin_scatter : a second sweep over the stream writes each visibility's record into its work-group's range
// A record written straight to its bin is a lone 8-byte store into one of ~10^5 open regions, i.e. one
// partial-line HBM write per visibility (2.6 ms for 10^8 records against 0.45 ms for the counting sweep), so
// large streams scatter in two levels, both through an LDS counting sort so that records leave the CU as
// contiguous runs (option "prepass": 0 = auto, 1 = one level, 2 = two levels, 3 = one level with global atomics
// only, 4 = two levels recomputing from the stream instead of reading pre-records, 5 / 6 = two levels with 16- /
// 12-byte intermediate records):
//   coarse_scatter : chunks of 8192 pre-records are counting-sorted by COARSE bin (2^k consecutive bins) in LDS
//                    and written as runs into a temporary array laid out like the final one at coarse
//                    granularity (one global atomic per chunk and non-empty coarse bin reserves the run's place)
//   fine_scatter   : equal shares of the temporary array (a chunk spans one or two coarse bins, so few open
//                    lines per work-group) are sorted by bin the same way and written to the final array
// Records are 8-byte words (RecWord, common.h), between the levels as well when the fields fit (FMT 8: the word's
// spare bits carry the bin's index inside its coarse bin, the coarse bin follows from the record's position).
// Bytes per visibility: 24 read + 8 written (count), 8 + 8 (coarse), 8 + 8 (fine) = 64 B; 76 B with the 12-byte
// records of early round 2, 98 B when both scatter levels recomputed from the stream and records were 16 B.  Because the scatter no longer reads the
// caller's arrays, the count and the scatter cannot disagree (a caller overwriting u, v, wbin during the call
// changes which records are produced, never where they are written); every record store is bounds-checked
// against the array all the same and violations are counted (option "errors").
//
// Coordinates follow frac_coords / convgrid2 of src/Gridding.hs:126-151,212-218: the footprint origin is
// (x - gw/2, y - gh/2); a visibility none of whose taps can land inside the grid is dropped here (fixoutofbounds
// would drop every one of its taps, :883-891).
#include "tile_common.h"

namespace gridhip {

struct BinOut {
    int32_t bin;  // -1: no tap in the grid, -2: wbin outside [0,W)
    int32_t lxy, kslice;
};

// Element e of the pre-pass's stream is part (e % P) of visibility e / P (P = 1: the visibility itself).
__device__ __forceinline__ void elem_of(const Geom &g, int64_t e, int64_t *k, int *part)
{
    if (g.P == 1) {
        *k = e;
        *part = 0;
    } else {
        const uint32_t kk = udiv_magic((uint32_t)e, g.mP, g.sP);  // (e < 2^31)
        *k = kk;
        *part = (int)((uint32_t)e - kk * (uint32_t)g.P);
    }
}

// frac_coord of src/Gridding.hs:126-140 for the counting sweep, whose time is its instruction count: the same
// operations in the same order as frac_coord_dev (common.h) - rounded multiply, rounded add, floor, subtract,
// multiply, round - but the cell stays a double (`fl`, integer-valued) until it is known to be inside the int32
// range: frac_coord_dev's (int64) fl and (double) f conversions are a dozen instructions each on this hardware, and
// for |fl| < 2^31 the second gives back fl exactly, so x - (double)(int64)fl == x - fl bit for bit.
__device__ __forceinline__ void frac_coord_cell(double nf, double halfnf, double qpxf, double qpxfrac, int32_t qpx, double p,
                                                double *fl, int32_t *fr)
{
#pragma clang fp contract(off)
    const double pn = p * nf;
    const double x = halfnf + pn;
    const double f = floor(x + qpxfrac);
    const double xd = x - f;
    const double dd = xd * qpxf;
    int32_t r = (int32_t)round(dd);
    r = r < 0 ? 0 : r;
    r = r > qpx - 1 ? qpx - 1 : r;
    *fl = f;
    *fr = r;
}

__device__ __forceinline__ BinOut vis_bin(const Geom &g, double pu, double pv, int64_t wb, int64_t k, int part)
{
    BinOut o;
    double xfl, yfl;
    int32_t xf, yf;
    const double qf = (double)g.Q, qfrac = 0.5 / qf;  // (as frac_coord_dev: exact for power-of-two Q, correctly rounded otherwise)
    frac_coord_cell((double)g.Wd, (double)(g.Wd / 2), qf, qfrac, g.Q, pu, &xfl, &xf);
    frac_coord_cell((double)g.H, (double)(g.H / 2), qf, qfrac, g.Q, pv, &yfl, &yf);
    // footprint origin (this part's corner of it), still in doubles: small integers, exact
    int qy = 0, qx = 0;
    if (g.P > 1) {
        qy = (int)udiv_magic((uint32_t)part, g.mPx, g.sPx);
        qx = part - qy * g.px;
    }
    const double x0 = xfl - (double)(g.fgw / 2 - qx * g.gw), y0 = yfl - (double)(g.fgh / 2 - qy * g.gh);
    // no tap inside the grid (fixoutofbounds would drop every one).  NaN and infinite coordinates compare false here
    // and are dropped; everything that passes is far inside the int32 range.
    if (!(x0 > -(double)g.gw && x0 < (double)g.Wd && y0 > -(double)g.gh && y0 < (double)g.H)) {
        o.bin = -1;
        o.lxy = 0;
        o.kslice = 0;
        return o;
    }
    if (wb < 0 || wb >= g.W) {
        o.bin = part == 0 ? -2 : -1;  // (counted once per visibility)
        o.lxy = 0;
        o.kslice = 0;
        return o;
    }
    const int32_t X = (int32_t)x0 + g.offx, Y = (int32_t)y0 + g.offy;
    // (X, Y >= 0 by construction of offx / offy)
    const int32_t tx = (int32_t)udiv_magic((uint32_t)X, g.mTx, g.sTx), ty = (int32_t)udiv_magic((uint32_t)Y, g.mTy, g.sTy);
    const int32_t lx = X - tx * g.Tx, ly = Y - ty * g.Ty;
    const int32_t grp = g.ngroups > 1 ? (int32_t)udiv_magic((uint32_t)((int32_t)wb * g.ngroups), g.mW, g.sW) : 0;  // W * ngroups < 2^31
    o.bin = grp * g.ntiles + ty * g.ntx + tx;
    o.lxy = (ly << 16) | lx;
    o.kslice = g.per_vis ? (int32_t)k : (((int32_t)wb * g.Q + yf) * g.Q + xf) * g.P + part;
    return o;
}

// ---- pre-records ------------------------------------------------------------------------------
// What the counting sweep computed, one 8-byte word per visibility (bin | lx | ly | kslice, the bin's width from
// the geometry; all ones = dropped); the visibility's index is its position.
struct PreFmt {
    int bin_bits;  // bits of the bin field (bin_count_kernel: negated = count FROM the pre-records)
};
__device__ __forceinline__ unsigned long long pre_pack(const PreFmt f, const BinOut &b)
{
    if (b.bin < 0) return ~0ull;
    const unsigned long long lx = (unsigned)b.lxy & 0x7f, ly = ((unsigned)b.lxy >> 16) & 0x7f;
    return (unsigned long long)(unsigned)b.bin | lx << f.bin_bits | ly << (f.bin_bits + 7) |
           (unsigned long long)(unsigned)b.kslice << (f.bin_bits + 14);
}
__device__ __forceinline__ BinOut pre_unpack(const PreFmt f, unsigned long long p)
{
    BinOut b;
    if (p == ~0ull) {
        b.bin = -1;
        b.lxy = 0;
        b.kslice = 0;
        return b;
    }
    b.bin = (int32_t)(p & ((1ull << f.bin_bits) - 1));
    const int32_t lx = (int32_t)(p >> f.bin_bits) & 0x7f, ly = (int32_t)(p >> (f.bin_bits + 7)) & 0x7f;
    b.lxy = (ly << 16) | lx;
    b.kslice = (int32_t)(p >> (f.bin_bits + 14));
    return b;
}

// ---- records between the two scatter levels -------------------------------------------------------
// FMT 8: the final 8-byte word with, in the bits above its fields (from bit ob + kb + 14), the bin's index inside its
// coarse bin (bin & (2^shift - 1)); which coarse bin follows from where the record lies in the coarse-ordered array.
// Level 2 keeps a 10-bit sort key in the same place while a chunk is in LDS, hence the condition ob + kb <= 40.
// Otherwise the unpacked record plus its bin - T12: 12 bytes, the bin in the 18 bits of lxy that lx and ly (7 bits
// each, tiles are at most 128 cells) leave free - bins < 2^18; or 16 bytes with the bin in a fourth word.
constexpr int TMP12_MAX_BINS = 1 << 18;
template <bool T12>
struct TmpRec;
template <>
struct TmpRec<true> {
    int32_t lxy, kslice, orig;
    __device__ __forceinline__ void set(int32_t lxy_, int32_t ks, int32_t o, int32_t bin)
    {
        lxy = lxy_ | ((bin & 0x1ff) << 7) | ((bin >> 9) << 23);
        kslice = ks;
        orig = o;
    }
    __device__ __forceinline__ int32_t bin() const
    {
        return (int32_t)(((uint32_t)lxy >> 7) & 0x1ff) | (int32_t)(((uint32_t)lxy >> 23) << 9);
    }
    __device__ __forceinline__ RecWord final_rec(const Geom &g) const { return rec_pack(g, lxy & 0x007f007f, kslice, orig); }
};
template <>
struct TmpRec<false> {
    int32_t lxy, kslice, orig, b;
    __device__ __forceinline__ void set(int32_t lxy_, int32_t ks, int32_t o, int32_t bin)
    {
        lxy = lxy_;
        kslice = ks;
        orig = o;
        b = bin;
    }
    __device__ __forceinline__ int32_t bin() const { return b; }
    __device__ __forceinline__ RecWord final_rec(const Geom &g) const { return rec_pack(g, lxy, kslice, orig); }
};
static_assert(sizeof(TmpRec<true>) == 12 && sizeof(TmpRec<false>) == 16, "record sizes");

// Each block owns one contiguous slice of the stream (the same slice in both sweeps).
__device__ __forceinline__ void block_range(int64_t n, int64_t *lo, int64_t *hi)
{
    int64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + 255) & ~(int64_t)255;
    *lo = (int64_t)blockIdx.x * per;
    *hi = *lo + per < n ? *lo + per : n;
    if (*lo > n) *lo = n;
}

template <bool LDS_HIST, int UN = 1, bool V2 = false>
__global__ void __launch_bounds__(1024) bin_count_kernel(Geom g, int64_t n, const double *__restrict__ u,
                                                         const double *__restrict__ v, int64_t stride,
                                                         const int64_t *__restrict__ wbin,
                                                         int32_t *__restrict__ bin_count,
                                                         int32_t *__restrict__ block_hist,
                                                         int32_t *__restrict__ scalars, int bin_lo, int bin_hi,
                                                         PreFmt pf, unsigned long long *__restrict__ pre,
                                                         double2 *__restrict__ zero_out)
{
    // LDS_HIST: this launch handles the bins [bin_lo, bin_hi) only (a window that fits in LDS); grids
    // with more bins than that are covered by several launches.  pre != null: the first window's launch leaves
    // the pre-record of every visibility, the others (pf.bin_bits < 0) count from those.
    extern __shared__ int32_t hist[];
    const int nwin = bin_hi - bin_lo;
    if (LDS_HIST) {
        for (int i = threadIdx.x; i < nwin; i += blockDim.x) hist[i] = 0;
        __syncthreads();
    }
    int64_t lo, hi;
    block_range(n, &lo, &hi);
    int dropped = 0;
    const bool from_pre = pre && pf.bin_bits < 0;
    const PreFmt rf = {from_pre ? -pf.bin_bits : pf.bin_bits};
    auto tally = [&](const BinOut &b) {
        if (b.bin >= 0) {
            if (LDS_HIST) {
                if (b.bin >= bin_lo && b.bin < bin_hi) atomicAdd(&hist[b.bin - bin_lo], 1);
            } else
                atomicAdd(&bin_count[b.bin], 1);
        } else if (b.bin == -2)
            ++dropped;
    };
    // V2 (unit stride, one record per visibility, 16-byte aligned arrays - the launcher checks): a thread takes two
    // consecutive visibilities, so every access of the sweep is a 16-byte one (u, v, wbin read, pre-records written),
    // and the work-groups take the stream grid-stride instead of one contiguous slice each.  The wider accesses alone
    // measured nothing; the grid-stride order took the sweep from 0.70 to 0.62 ms (256 x 4 concurrent streams become
    // four narrow windows: HBM page locality).  The same order in the coarse scatter measured no difference.
    if (V2 && !from_pre) {
        hi = n;
        for (int64_t k0 = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); k0 < hi; k0 += 2 * (int64_t)gridDim.x * blockDim.x) {
            const bool two = k0 + 1 < hi;
            double2 pu, pv;
            longlong2 wb = make_longlong2(0, 0);
            if (two) {
                pu = *reinterpret_cast<const double2 *>(u + k0);
                pv = *reinterpret_cast<const double2 *>(v + k0);
                if (wbin) wb = *reinterpret_cast<const longlong2 *>(wbin + k0);
            } else {
                pu = make_double2(u[k0], 0.0);
                pv = make_double2(v[k0], 0.0);
                if (wbin) wb.x = wbin[k0];
            }
            const BinOut b0 = vis_bin(g, pu.x, pv.x, wb.x, k0, 0);
            BinOut b1 = vis_bin(g, pu.y, pv.y, wb.y, k0 + 1, 0);
            if (!two) b1.bin = -1;
            if (pre) {
                if (two)
                    *reinterpret_cast<ulonglong2 *>(pre + k0) = make_ulonglong2(pre_pack(rf, b0), pre_pack(rf, b1));
                else
                    pre[k0] = pre_pack(rf, b0);
            }
            if (zero_out) {
                if (b0.bin < 0) zero_out[k0] = make_double2(0.0, 0.0);
                if (two && b1.bin < 0) zero_out[k0 + 1] = make_double2(0.0, 0.0);
            }
            tally(b0);
            tally(b1);
        }
    } else {
    // UN visibilities per thread and trip: their loads are in flight together (one work-group per CU has
    // nothing else to cover the memory latency with).
    // (two-level scatter, block_hist == null: nothing ties a work-group to a slice of the stream, so this loop too runs
    // grid-stride; the one-level scatter needs each work-group to see the same slice in both of its sweeps)
    if (!block_hist) {
        lo = (int64_t)blockIdx.x * UN * blockDim.x;
        hi = n;
    }
    const int64_t kstep = block_hist ? (int64_t)UN * blockDim.x : (int64_t)gridDim.x * UN * blockDim.x;
    for (int64_t k0 = lo + threadIdx.x; k0 < hi; k0 += kstep) {
        BinOut b[UN];
        if (from_pre) {
            unsigned long long p[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int64_t k = k0 + (int64_t)q * blockDim.x;
                p[q] = k < hi ? pre[k] : ~0ull;
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) b[q] = pre_unpack(rf, p[q]);
        } else {
            double pu[UN], pv[UN];
            int64_t wb[UN], kk[UN];
            int part[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int64_t e = min(k0 + (int64_t)q * blockDim.x, hi - 1);  // (loads unconditional, indices clamped)
                elem_of(g, e, &kk[q], &part[q]);
                pu[q] = u[kk[q] * stride];
                pv[q] = v[kk[q] * stride];
                wb[q] = wbin ? wbin[kk[q]] : 0;
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int64_t e = k0 + (int64_t)q * blockDim.x;
                b[q] = vis_bin(g, pu[q], pv[q], wb[q], kk[q], part[q]);
                if (e >= hi)
                    b[q].bin = -1;
                else {
                    if (pre) pre[e] = pre_pack(rf, b[q]);
                    if (zero_out && b[q].bin < 0) zero_out[kk[q]] = make_double2(0.0, 0.0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) tally(b[q]);
    }
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
    // (eight blocks' counts are read together: the walk is a chain of memory latencies otherwise - 16 -> 5 us at
    // 61 work-groups, which is 6 % of a 10^6-visibility call)
    for (int b0 = 0; b0 < nblocks; b0 += 8) {
        int c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = b0 + q < nblocks ? block_hist[(size_t)(b0 + q) * nbins + bin] : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (b0 + q < nblocks) block_hist[(size_t)(b0 + q) * nbins + bin] = run;
            run += c[q];
        }
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

// Work-groups 0 .. SCAN_SEGS-1: bin_start over an equal share of the bins each (and the cursors cleared) - a share's
// first value is the sum of all counts before it, which its work-group adds up itself (a coalesced read of at most the
// whole histogram out of the L2: nothing waits for another work-group); work-group SCAN_SEGS + grp: that w-group's
// work_start.  All parts run side by side (one work-group doing them one after the other took 0.115 ms for the 110 000
// bins of an 8192^2 grid, 0.104 ms with only the w-groups' parts split off).
constexpr int SCAN_SEGS = 16;
template <int NT>
__global__ void __launch_bounds__(NT) bin_scan_kernel(Geom g, const int32_t *__restrict__ bin_count,
                                                        int32_t *__restrict__ bin_start,
                                                        int32_t *__restrict__ work_start,
                                                        int32_t *__restrict__ cursor)
{
    __shared__ int32_t wtot[16];
    const int tid = threadIdx.x;
    const bool starts = (int)blockIdx.x < SCAN_SEGS;  // (uniform per work-group)
    const int grp = (int)blockIdx.x - SCAN_SEGS;
    if (!starts && grp >= g.ngroups) return;
    // this work-group's range [lo, hi) of `cnt` -> `dst`
    const int seg = ((g.nbins + SCAN_SEGS - 1) / SCAN_SEGS + 3) & ~3;
    const int lo = starts ? min((int)blockIdx.x * seg, g.nbins) : 0;
    const int hi = starts ? min(lo + seg, g.nbins) : g.ntiles;
    const int32_t *cnt = starts ? bin_count : bin_count + (size_t)grp * g.ntiles;
    int32_t *dst = starts ? bin_start : work_start + (size_t)grp * (g.ntiles + 1);
    int carry = 0;
    if (starts && lo > 0) {  // everything before the share
        int sum = 0;
        for (int i = tid; i < lo; i += NT) sum += bin_count[i];
        int total;
        (void)block_exclusive_scan<NT>(sum, wtot, &total);
        carry = total;
    }
    for (int base = lo; base < hi; base += 4 * NT) {
        const int i0 = base + tid * 4;
        int c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[q] = i0 + q < hi ? cnt[i0 + q] : 0;
            if (!starts) c[q] = (c[q] + g.chunk - 1) / g.chunk;  // work items of the tile: chunks of <= chunk visibilities
        }
        int total;
        int acc = carry + block_exclusive_scan<NT>(c[0] + c[1] + c[2] + c[3], wtot, &total);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (i0 + q < hi) {
                dst[i0 + q] = acc;
                if (starts) cursor[i0 + q] = 0;
            }
            acc += c[q];
        }
        carry += total;
    }
    if (tid == 0 && (!starts || (hi == g.nbins && (lo < hi || blockIdx.x == 0)))) dst[hi] = carry;
}

// One-level scatter (small streams).  `cap` = record slots the array holds: a slot outside it is never written
// (it can only arise if the caller's arrays changed between the two sweeps) and is counted in scalars[2].
template <bool LDS_HIST>
__global__ void __launch_bounds__(1024) bin_scatter_kernel(Geom g, int64_t n, const double *__restrict__ u,
                                                           const double *__restrict__ v, int64_t stride,
                                                           const int64_t *__restrict__ wbin,
                                                           const int32_t *__restrict__ bin_start,
                                                           int32_t *__restrict__ cursor,
                                                           const int32_t *__restrict__ block_hist,
                                                           RecWord *__restrict__ recs, int bin_lo, int bin_hi,
                                                           int32_t cap, int32_t *__restrict__ scalars)
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
    int bad = 0;
    for (int64_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
        int64_t k;
        int part;
        elem_of(g, e, &k, &part);
        BinOut b = vis_bin(g, u[k * stride], v[k * stride], wbin ? wbin[k] : 0, k, part);
        if (b.bin < 0) continue;
        if (LDS_HIST && (b.bin < bin_lo || b.bin >= bin_hi)) continue;
        int slot;
        if (LDS_HIST)
            slot = atomicAdd(&hist[b.bin - bin_lo], 1);
        else
            slot = bin_start[b.bin] + atomicAdd(&cursor[b.bin], 1);
        if ((uint32_t)slot >= (uint32_t)cap) {
            ++bad;
            continue;
        }
        recs[slot] = rec_pack(g, b.lxy, b.kslice, (int32_t)k);
    }
    if (bad) atomicAdd(&scalars[2], bad);
}

// ---- two-level scatter ----------------------------------------------------------------------
// In-place exclusive scan of hist[0..nent), nent <= 1024, by NT threads (consecutive entries per thread);
// reserve(e, count, base) is called for every non-empty entry and its result kept in res[] (one slot per entry of
// the thread) - the caller publishes the results later, so that a returning global atomic issued in `reserve` is not
// waited for here; wtot[16] receives the total.
// Barrier between the LDS phases of the scatter kernels.  __syncthreads() also waits for every outstanding global
// access of the wave (s_waitcnt vmcnt(0): on this hardware stores and loads share that counter), which would expose the
// latency of the next chunk's prefetch and of the previous chunk's stores at every phase boundary.  The phases only
// hand LDS data to each other; registers fed by global loads and returning atomics are waited for where they are used.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT, typename F>
__device__ __forceinline__ void scan_entries(int32_t *hist, int nent, int32_t *wtot, int (&res)[1024 / NT], F &&reserve)
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
    lds_barrier();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid * EPT + k;
        res[k] = 0;
        if (e < nent) {
            hist[e] = base;
            if (c[k]) res[k] = reserve(e, c[k], base);
        }
        base += c[k];
    }
    if (tid == NT - 1) wtot[16] = base;
}
// the second half: gbase[e] = res (after the work that did not need it)
template <int NT>
__device__ __forceinline__ void publish_entries(int32_t *gbase, int nent, const int (&res)[1024 / NT])
{
    constexpr int EPT = 1024 / NT;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = threadIdx.x * EPT + k;
        if (e < nent) gbase[e] = res[k];
    }
}

// Level 1.  tmp is laid out like the final record array at coarse granularity: coarse bin c owns
// [bin_start[c << shift], bin_start[min((c + 1) << shift, nbins)]).  Each chunk reserves, per coarse bin, a
// contiguous range there (one global atomic per chunk and non-empty coarse bin) and writes its records as
// runs; the record carries its bin for level 2 (FMT 8: the bin's low `shift` bits).
// FROM_PRE: the chunk is read from the counting sweep's pre-records; otherwise it is recomputed from the stream.
template <int NT, int CHUNK, bool FROM_PRE, int FMT>
__global__ void __launch_bounds__(NT, 4) coarse_scatter_kernel(Geom g, int64_t n, const double *__restrict__ u,
                                                               const double *__restrict__ v, int64_t stride,
                                                               const int64_t *__restrict__ wbin,
                                                               const int32_t *__restrict__ bin_start,
                                                               int32_t *__restrict__ ccur, int shift, int ncoarse,
                                                               void *__restrict__ tmp_,
                                                               const unsigned long long *__restrict__ pre, PreFmt pf,
                                                               int32_t cap, int32_t *__restrict__ scalars)
{
    using Tmp = TmpRec<FMT != 16>;
    constexpr int RW = FMT == 8 ? 2 : (int)sizeof(Tmp) / 4;  // 32-bit words per staged record
    extern __shared__ int32_t smem[];
    Tmp *sorted = reinterpret_cast<Tmp *>(smem);              // [CHUNK] (FMT 12, 16)
    RecWord *sorted8 = reinterpret_cast<RecWord *>(smem);     // [CHUNK] (FMT 8)
    uint16_t *scoarse = reinterpret_cast<uint16_t *>(smem + CHUNK * RW);  // [CHUNK] (FMT 8): coarse bin of sorted8[i]
    int32_t *hist = smem + CHUNK * RW + (FMT == 8 ? CHUNK / 2 : 0);  // [ncoarse]: count, then the coarse bin's first slot in `sorted`
    int32_t *gbase = hist + ncoarse;                              // [ncoarse]: tmp position of sorted[0] if it were in this bin
    int32_t *wtot = gbase + ncoarse;                              // [16] per-wave totals of the scan, [16] = chunk total
    const int tid = threadIdx.x;
    constexpr int PER = CHUNK / NT;
    const int fbit = g.ob + g.kb + 14;                            // (FMT 8) where the fine index sits in the word
    const RecWord fmask = (1ull << shift) - 1;
    int64_t lo, hi;
    block_range(n, &lo, &hi);
    int bad = 0;
    // The next chunk's pre-records travel while this chunk goes through its LDS phases (histogram, scan, sort):
    // with one work-group per CU nothing else would keep the memory system busy meanwhile.
    unsigned long long nxt[PER];
    if (FROM_PRE) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t k = lo + q * NT + tid;
            nxt[q] = k < hi ? pre[k] : ~0ull;
        }
    }
    for (int64_t c0 = lo; c0 < hi; c0 += CHUNK) {
        for (int i = tid; i < ncoarse; i += NT) hist[i] = 0;
        lds_barrier();
        BinOut b[PER];
        int rank[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t k = c0 + q * NT + tid;
            b[q].bin = -1;
            if (FROM_PRE) {
                b[q] = pre_unpack(pf, nxt[q]);
                const int64_t kn = k + CHUNK;
                nxt[q] = kn < hi ? pre[kn] : ~0ull;
            } else if (k < hi) {
                int64_t kv;
                int part;
                elem_of(g, k, &kv, &part);
                b[q] = vis_bin(g, u[kv * stride], v[kv * stride], wbin ? wbin[kv] : 0, kv, part);
            }
            if (b[q].bin >= g.nbins) b[q].bin = -1;  // (cannot happen)
        }
#pragma unroll
        for (int q = 0; q < PER; ++q)
            rank[q] = run_rank(hist, b[q].bin >= 0 ? b[q].bin >> shift : -1);
        lds_barrier();
        // exclusive scan of the counts and the global reservations; the returning atomics travel while the chunk
        // is sorted in LDS (which needs the local offsets only)
        int res[1024 / NT];
        scan_entries<NT>(hist, ncoarse, wtot, res, [&](int e, int c, int base) {
            return bin_start[e << shift] + atomicAdd(&ccur[e], c) - base;
        });
        lds_barrier();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (b[q].bin < 0) continue;
            int64_t kv;
            int part;
            elem_of(g, c0 + q * NT + tid, &kv, &part);
            const int pos = hist[b[q].bin >> shift] + rank[q];
            if (FMT == 8) {
                sorted8[pos] = rec_pack(g, b[q].lxy, b[q].kslice, (int32_t)kv) | ((RecWord)(uint32_t)b[q].bin & fmask) << fbit;
                scoarse[pos] = (uint16_t)(b[q].bin >> shift);
            } else {
                Tmp r;
                r.set(b[q].lxy, b[q].kslice, (int32_t)kv, b[q].bin);
                sorted[pos] = r;
            }
        }
        publish_entries<NT>(gbase, ncoarse, res);
        lds_barrier();
        const int total = wtot[16];
        for (int i = tid; i < total; i += NT) {
            // neighbouring lanes: neighbouring slots
            if (FMT == 8) {
                const int slot = gbase[scoarse[i]] + i;
                if ((uint32_t)slot >= (uint32_t)cap) {
                    ++bad;
                    continue;
                }
                static_cast<RecWord *>(tmp_)[slot] = sorted8[i];
            } else {
                const Tmp r = sorted[i];
                const int slot = gbase[r.bin() >> shift] + i;
                if ((uint32_t)slot >= (uint32_t)cap) {
                    ++bad;
                    continue;
                }
                static_cast<Tmp *>(tmp_)[slot] = r;
            }
        }
        lds_barrier();
    }
    if (bad) atomicAdd(&scalars[2], bad);
}

// Level 2.  The work-groups take tmp in chunks of CHUNK records, dealt round-robin.  tmp is ordered by
// coarse bin, so a chunk's bins lie between the coarse bins of its first and last record: normally one or two
// coarse bins, i.e. at most a few hundred bins.  The chunk is counting-sorted by bin in LDS exactly as level 1 sorts
// by coarse bin, each bin's range is reserved with one global atomic, and the records leave as runs.  A chunk that
// spans more than 1024 bins (very sparse regions) falls back to one global atomic per record.
// What one staged record is, per format: its sort key (bin - k0) and the final word it becomes.
template <int FMT>
struct FineRec;
template <>
struct FineRec<8> {
    RecWord w;  // (in LDS: the 10-bit key where the fine index was)
};
template <>
struct FineRec<12> {
    TmpRec<true> r;
};
template <>
struct FineRec<16> {
    TmpRec<false> r;
};

template <int NT, int CHUNK, int FMT>
__global__ void __launch_bounds__(NT, 4) fine_scatter_kernel(Geom g, const int32_t *__restrict__ bin_start,
                                                             int32_t *__restrict__ cursor, int shift, int ncoarse,
                                                             const void *__restrict__ tmp_,
                                                             RecWord *__restrict__ out, int32_t cap,
                                                             int32_t *__restrict__ scalars)
{
    using Rec = FineRec<FMT>;
    constexpr int RW = (int)sizeof(Rec) / 4;
    extern __shared__ int32_t smem[];
    Rec *sorted = reinterpret_cast<Rec *>(smem);  // [CHUNK]
    int32_t *hist = smem + CHUNK * RW;            // [1024]
    int32_t *gbase = hist + 1024;                 // [1024]
    int32_t *wtot = gbase + 1024;                 // [32]: [0..16] scan, [24], [25] the chunk's first / last bin or coarse bin
    int32_t *cend = wtot + 32;                    // (FMT 8) [ncoarse]: where each coarse bin ends in tmp
    const Rec *tmp = static_cast<const Rec *>(tmp_);
    const int tid = threadIdx.x;
    constexpr int PER = CHUNK / NT;
    const int fbit = g.ob + g.kb + 14;
    const RecWord lowmask = (1ull << fbit) - 1, fmask = (1ull << shift) - 1;
    int64_t ntot = bin_start[g.nbins];
    if (ntot > cap) ntot = cap;  // (level 1 has written nothing beyond `cap`)
    // chunks are dealt round-robin to the work-groups: at any moment they read one narrow window of tmp and write into
    // the output range of the one or two coarse bins it belongs to (1.50 -> 1.45 ms for the pre-pass against one
    // contiguous share per work-group; the bins' cursors see no more contention than they did)
    const int64_t lo = min((int64_t)blockIdx.x * CHUNK, ntot), hi = ntot, fstep = (int64_t)gridDim.x * CHUNK;
    int bad = 0;
    if (FMT == 8) {
        for (int t = tid; t < ncoarse; t += NT) cend[t] = bin_start[min((t + 1) << shift, g.nbins)];
        lds_barrier();
    }
    // (FMT 8) the coarse bin position i of tmp lies in, searching upwards from c (positions come in rising order)
    auto coarse_at = [&](int64_t i, int c) {
        while (c < ncoarse - 1 && i >= cend[c]) ++c;
        return c;
    };
    auto bin_of = [&](const Rec &x, int64_t i, int &c) -> int {
        if constexpr (FMT == 8) {
            c = coarse_at(i, c);
            return (c << shift) | (int)((x.w >> fbit) & fmask);
        } else
            return x.r.bin();
    };
    auto final_of = [&](const Rec &x) -> RecWord {
        if constexpr (FMT == 8)
            return x.w & lowmask;
        else
            return x.r.final_rec(g);
    };
    // The next chunk's records travel while this chunk goes through its LDS phases, as in level 1.  (Two work-groups
    // per CU instead - 8-byte records leave the LDS for it, at 64 registers per thread and without this prefetch -
    // measured slower: 1.63 against 1.55 ms for the whole pre-pass.)
    constexpr bool PREF = true;
    Rec nxt[PER];
    if (PREF && lo < hi) {
#pragma unroll
        for (int q = 0; q < PER; ++q) nxt[q] = tmp[min(lo + q * NT + tid, hi - 1)];  // (unconditional: indices clamped)
    }
    for (int64_t c0 = lo; c0 < hi; c0 += fstep) {
        const int64_t c1 = min(c0 + CHUNK, hi);
        if (!PREF) {
#pragma unroll
            for (int q = 0; q < PER; ++q) nxt[q] = tmp[min(c0 + q * NT + tid, hi - 1)];
        }
        int b_first, b_last, cfirst = 0;
        if (FMT == 8) {
            // the coarse bins of the chunk's first and last position: exactly one (non-empty) coarse bin holds each
            for (int t = tid; t < ncoarse; t += NT) {
                const int64_t cs = t ? cend[t - 1] : 0, ce = cend[t];
                if (cs <= c0 && c0 < ce) wtot[24] = t;
                if (cs <= c1 - 1 && c1 - 1 < ce) wtot[25] = t;
            }
            lds_barrier();
            cfirst = min(max(wtot[24], 0), ncoarse - 1);
            const int clast = min(max(wtot[25], cfirst), ncoarse - 1);
            b_first = cfirst << shift;
            b_last = min(((clast + 1) << shift) - 1, g.nbins - 1);
        } else {
            // the chunk's first and last record are in two threads' prefetch registers: hand them round through LDS
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int64_t i = c0 + q * NT + tid;
                int c = 0;
                if (i == c0) wtot[24] = bin_of(nxt[q], i, c);
                if (i == c1 - 1) wtot[25] = bin_of(nxt[q], i, c);
            }
            lds_barrier();
            b_first = wtot[24], b_last = wtot[25];
            b_first = min(max(b_first, 0), g.nbins - 1);
            b_last = min(max(b_last, b_first), g.nbins - 1);
        }
        const int k0 = (b_first >> shift) << shift, span = (((b_last >> shift) + 1) << shift) - k0;  // bins k0 .. k0 + span
        if (span > 1024) {  // rare: one global atomic per record
            int c = cfirst;
            for (int64_t i = c0 + tid; i < c1; i += NT) {
                const Rec r = tmp[i];
                const int k = bin_of(r, i, c);
                if ((uint32_t)k >= (uint32_t)g.nbins) {
                    ++bad;
                    continue;
                }
                const int slot = bin_start[k] + atomicAdd(&cursor[k], 1);
                if ((uint32_t)slot >= (uint32_t)cap) {
                    ++bad;
                    continue;
                }
                out[slot] = final_of(r);
            }
            if (PREF) {
#pragma unroll
                for (int q = 0; q < PER; ++q) nxt[q] = tmp[min(c0 + fstep + q * NT + tid, hi - 1)];
            }
            lds_barrier();  // (wtot[24], [25] are rewritten by the next chunk)
            continue;
        }
        for (int i = tid; i < span; i += NT) hist[i] = 0;
        lds_barrier();
        Rec r[PER];
        int key[PER], rank[PER];
        int c = cfirst;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t i = c0 + q * NT + tid;
            key[q] = -1;
            if (i < c1) {
                r[q] = nxt[q];
                key[q] = bin_of(r[q], i, c) - k0;
                if ((uint32_t)key[q] >= (uint32_t)span || k0 + key[q] >= g.nbins) {  // tmp not ordered (cannot happen)
                    key[q] = -1;
                    ++bad;
                }
            }
        }
        if (PREF) {
#pragma unroll
            for (int q = 0; q < PER; ++q) nxt[q] = tmp[min(c0 + fstep + q * NT + tid, hi - 1)];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) rank[q] = run_rank(hist, key[q]);
        lds_barrier();
        int res[1024 / NT];
        scan_entries<NT>(hist, span, wtot, res, [&](int e, int c_, int base) {
            return bin_start[k0 + e] + atomicAdd(&cursor[k0 + e], c_) - base;
        });
        lds_barrier();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (key[q] < 0) continue;
            if constexpr (FMT == 8) r[q].w = (r[q].w & lowmask) | (RecWord)key[q] << fbit;  // the key replaces the fine index
            sorted[hist[key[q]] + rank[q]] = r[q];
        }
        publish_entries<NT>(gbase, span, res);
        lds_barrier();
        const int total = wtot[16];
        for (int i = tid; i < total; i += NT) {
            const Rec x = sorted[i];
            int kx;
            if constexpr (FMT == 8)
                kx = (int)(x.w >> fbit);
            else
                kx = x.r.bin() - k0;
            const int slot = gbase[kx] + i;
            if ((uint32_t)slot >= (uint32_t)cap) {
                ++bad;
                continue;
            }
            out[slot] = final_of(x);
        }
        lds_barrier();
    }
    if (bad) atomicAdd(&scalars[2], bad);
}

// NT threads take chunks of CHUNK records: <1024, 8192> is one work-group per CU (LDS), <512, 4096> two, so that one
// drains its stores and waits for its reservations while the other sorts.
template <bool FROM_PRE, int FMT, int NT, int CHUNK>
static int launch_two_level(gridhip_ctx *ctx, const Geom &g, const Tables &t, int64_t n, const double *u, const double *v,
                            int64_t uv_stride, const int64_t *wbin, int shift, int ncoarse, int cblocks,
                            const unsigned long long *pre, PreFmt pf, int32_t cap)
{
    const size_t rec_bytes = FMT;
    const size_t coarse_lds = (size_t)CHUNK * (rec_bytes + (FMT == 8 ? 2 : 0)) + (size_t)(2 * ncoarse + 32) * sizeof(int32_t);
    const size_t fine_lds = (size_t)CHUNK * rec_bytes + (size_t)(2 * 1024 + 32 + (FMT == 8 ? ncoarse : 0)) * sizeof(int32_t);
    auto coarse = coarse_scatter_kernel<NT, CHUNK, FROM_PRE, FMT>;
    auto fine = fine_scatter_kernel<NT, CHUNK, FMT>;
    const uint32_t bit = 2u << ((FROM_PRE ? 1 : 0) + (FMT == 12 ? 2 : FMT == 8 ? 8 : 0) + (NT == 512 ? 4 : 0));
    if (!(ctx->attr_mask & bit)) {
        GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)coarse, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
        GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)fine, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
        ctx->attr_mask |= bit;
    }
    int32_t *ccur = (int32_t *)ctx->blockhist.ptr;
    void *tmp = ctx->recs_tmp.ptr;
    hipLaunchKernelGGL(coarse, dim3(cblocks), dim3(NT), coarse_lds, ctx->stream, g, n, u, v, uv_stride, wbin, t.bin_start,
                       ccur, shift, ncoarse, tmp, pre, pf, cap, t.scalars);
    hipLaunchKernelGGL(fine, dim3(cblocks), dim3(NT), fine_lds, ctx->stream, g, t.bin_start, t.cursor, shift, ncoarse,
                       (const void *)tmp, (RecWord *)ctx->recs.ptr, cap, t.scalars);
    return GRIDHIP_OK;
}

int launch_bin(gridhip_ctx *ctx, const Geom &g, int64_t n, const double *u, const double *v,
               int64_t uv_stride, const int64_t *wbin, double2 *zero_out)
{
    GH_CHECK(ws_reserve(ctx, ctx->tables, tables_bytes(g)));
    if (!rec_fits(g)) return fail(ctx, GRIDHIP_EUNSUPPORTED, "record fields need %d bits", g.ob + g.kb + 14);  // (api.hip cuts such calls)
    GH_CHECK(ws_reserve(ctx, ctx->recs, (size_t)(n > 0 ? n : 1) * sizeof(RecWord)));
    Tables t = tables_of(ctx, g);
    int32_t *block_hist = nullptr;
    // scalars: [0] = dropped (wbin out of range), [2] = errors
    // (cleared together with the histogram and the coarse cursors by one small kernel further down)
    // record slots the scatter may write: all n of them - or fewer under the test hook "fault_inject", which
    // hides the array's last slots so that the bounds checks have something to catch
    const int64_t cap64 = n - (ctx->opt.fault_inject > 0 ? ctx->opt.fault_inject : 0);
    const int32_t cap = (int32_t)(cap64 < 0 ? 0 : cap64);

    // The histogram of one launch lives in LDS; when there are more bins than fit (large grids x 8
    // w-groups) the bins are covered in several windows.  Beyond 8 windows the re-reads cost more than plain
    // global atomics.
    const int lds_cap = (int)(((size_t)ctx->max_lds - 8192) / sizeof(int32_t));
    const int windows = (g.nbins + lds_cap - 1) / lds_cap;
    const int64_t p = ctx->opt.prepass;
    const bool lds_hist = windows <= 8 && p != 3;  // prepass = 3: global atomics, no LDS
    const int win = lds_hist ? (g.nbins + windows - 1) / windows : g.nbins;
    const size_t hist_bytes = (size_t)win * sizeof(int32_t);
    const int threads = p == 3 ? 256 : 1024;
    // one block per CU with an LDS histogram; more, smaller slices when counting in global memory
    int blocks = lds_hist ? ctx->num_cu * (hist_bytes <= 64 * 1024 ? 2 : 1) : ctx->num_cu * 8;
    // at least 16 K visibilities per work-group: below that the per-work-group histogram traffic
    // (and the serial walk over work-groups in bin_offsets_kernel) outweighs the parallelism
    int64_t need = (n + 16383) / 16384;
    if (need < 1) need = 1;
    if (blocks > need) blocks = (int)need;
    if (lds_hist && !(ctx->attr_mask & 1u)) {
        GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)bin_count_kernel<true, 1>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
        GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)bin_count_kernel<true, 4>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
        GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)bin_count_kernel<true, 1, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
        GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)bin_scatter_kernel<true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
        ctx->attr_mask |= 1u;
    }

    // two-level scatter for large streams (only the counting sweep needs the histogram windows)
    // bins per coarse bin = 2^shift.  Level 1 writes runs of chunk / ncoarse records, level 2 of chunk / 2^shift:
    // balanced (both ~ sqrt) keeps every run above 256 B
    int shift = (int)ctx->opt.coarse_shift;
    if (shift <= 0) {
        shift = 6;
        while (shift < 9 && (1 << (2 * shift)) < g.nbins) ++shift;
    }
    if (shift > 10) shift = 10;  // (level 2 sorts at most 1024 bins per coarse bin in LDS)
    while (((g.nbins + (1 << shift) - 1) >> shift) > 1024) ++shift;
    const int ncoarse = (g.nbins + (1 << shift) - 1) >> shift;
    const bool two_level = lds_hist && (p == 2 || p == 4 || p == 5 || p == 6 || (p == 0 && n >= ((int64_t)1 << 22)));
    if (two_level) {
        // record format between the levels: 8 bytes when the fields leave 10 bits for level 2's sort key, else 12
        // (bins < 2^18), else 16 (options prepass = 5, 6: 16 and 12 bytes regardless, for comparison)
        const int fmt = (p != 5 && p != 6 && g.ob + g.kb <= 40) ? 8 : (g.nbins <= TMP12_MAX_BINS && p != 5) ? 12 : 16;
        GH_CHECK(ws_reserve(ctx, ctx->recs_tmp, (size_t)(n > 0 ? n : 1) * fmt));
        GH_CHECK(ws_reserve(ctx, ctx->blockhist, (size_t)ncoarse * sizeof(int32_t)));  // coarse cursors
        launch_clear(ctx, t.bin_count, g.nbins, t.scalars, 3, (int32_t *)ctx->blockhist.ptr, ncoarse);
        // chunk size: 8192 records, one work-group per CU (option scatter_chunk = 4096: two per CU; measured no faster)
        const bool small_chunk = ctx->opt.scatter_chunk == 4096;
        const int chunk = small_chunk ? 4096 : 8192;
        int cblocks = ctx->num_cu * (small_chunk ? 2 : 1);
        int64_t cneed = (n + 4 * chunk - 1) / (4 * chunk);
        if (cblocks > cneed) cblocks = (int)(cneed < 1 ? 1 : cneed);
        // pre-records: when bin | lx | ly | kslice fit one 64-bit word
        const int bb = bits_for(g.nbins);
        const int64_t nslices = g.nslices;
        const bool use_pre = p != 4 && bb + 14 + bits_for(nslices) <= 63;
        unsigned long long *pre = nullptr;
        if (use_pre) {
            GH_CHECK(ws_reserve(ctx, ctx->recs_raw, (size_t)(n > 0 ? n : 1) * sizeof(unsigned long long)));
            pre = (unsigned long long *)ctx->recs_raw.ptr;
        }
        for (int wdw = 0; wdw < windows; ++wdw) {
            const int b_lo = wdw * win, b_hi = b_lo + win < g.nbins ? b_lo + win : g.nbins;
            // four visibilities per thread and trip when the bins take several windows: the windows after the first
            // read nothing but 8-byte pre-records and gain from more loads in flight (8192^2, three windows: 2.8 -> 2.6 ms;
            // one window, 4096^2: no difference).  Option count_unroll: 4 = always, 1 = never.
            const bool v2 = wdw == 0 && pre && uv_stride == 1 && g.P == 1 && ctx->opt.count_unroll != 1 &&
                            (((uintptr_t)u | (uintptr_t)v | (uintptr_t)wbin | (uintptr_t)pre) & 15) == 0;
            if (v2)
                hipLaunchKernelGGL((bin_count_kernel<true, 1, true>), dim3(blocks), dim3(threads), hist_bytes, ctx->stream, g, n, u, v,
                                   uv_stride, wbin, t.bin_count, (int32_t *)nullptr, t.scalars, b_lo, b_hi,
                                   PreFmt{bb}, pre, zero_out);
            else if (ctx->opt.count_unroll == 4 || (ctx->opt.count_unroll == 0 && windows > 1))
                hipLaunchKernelGGL((bin_count_kernel<true, 4>), dim3(blocks), dim3(threads), hist_bytes, ctx->stream, g, n, u, v,
                                   uv_stride, wbin, t.bin_count, (int32_t *)nullptr, t.scalars, b_lo, b_hi,
                                   PreFmt{wdw == 0 ? bb : -bb}, pre, zero_out);
            else
                hipLaunchKernelGGL((bin_count_kernel<true, 1>), dim3(blocks), dim3(threads), hist_bytes, ctx->stream, g, n, u, v,
                                   uv_stride, wbin, t.bin_count, (int32_t *)nullptr, t.scalars, b_lo, b_hi,
                                   PreFmt{wdw == 0 ? bb : -bb}, pre, zero_out);
        }
        hipLaunchKernelGGL(bin_scan_kernel<1024>, dim3(SCAN_SEGS + g.ngroups), dim3(1024), 0, ctx->stream, g, t.bin_count, t.bin_start,
                           t.work_start, t.cursor);
        const PreFmt pf{bb};
#define GH_TWO(P_, T_)                                                                                                  \
    (small_chunk ? launch_two_level<P_, T_, 512, 4096>(ctx, g, t, n, u, v, uv_stride, wbin, shift, ncoarse, cblocks, pre, \
                                                        pf, cap)                                                          \
                 : launch_two_level<P_, T_, 1024, 8192>(ctx, g, t, n, u, v, uv_stride, wbin, shift, ncoarse, cblocks,   \
                                                         pre, pf, cap))
        if (use_pre && fmt == 8)
            GH_CHECK(GH_TWO(true, 8));
        else if (use_pre && fmt == 12)
            GH_CHECK(GH_TWO(true, 12));
        else if (use_pre)
            GH_CHECK(GH_TWO(true, 16));
        else if (fmt == 8)
            GH_CHECK(GH_TWO(false, 8));
        else if (fmt == 12)
            GH_CHECK(GH_TWO(false, 12));
        else
            GH_CHECK(GH_TWO(false, 16));
#undef GH_TWO
        GH_CHECK_HIP(ctx, hipGetLastError());
        return GRIDHIP_OK;
    }

    launch_clear(ctx, t.bin_count, g.nbins, t.scalars, 3);
    if (lds_hist) {
        GH_CHECK(ws_reserve(ctx, ctx->blockhist, (size_t)blocks * g.nbins * sizeof(int32_t)));
        block_hist = (int32_t *)ctx->blockhist.ptr;
        for (int wdw = 0; wdw < windows; ++wdw) {
            const int b_lo = wdw * win, b_hi = b_lo + win < g.nbins ? b_lo + win : g.nbins;
            hipLaunchKernelGGL(bin_count_kernel<true>, dim3(blocks), dim3(threads), hist_bytes, ctx->stream, g, n, u,
                               v, uv_stride, wbin, t.bin_count, block_hist, t.scalars, b_lo, b_hi, PreFmt{0},
                               (unsigned long long *)nullptr, zero_out);
        }
    } else {
        hipLaunchKernelGGL(bin_count_kernel<false>, dim3(blocks), dim3(threads), 0, ctx->stream, g, n, u, v,
                           uv_stride, wbin, t.bin_count, block_hist, t.scalars, 0, g.nbins, PreFmt{0},
                           (unsigned long long *)nullptr, zero_out);
    }
    hipLaunchKernelGGL(bin_scan_kernel<1024>, dim3(SCAN_SEGS + g.ngroups), dim3(1024), 0, ctx->stream, g, t.bin_count, t.bin_start,
                       t.work_start, t.cursor);
    if (lds_hist) {
        hipLaunchKernelGGL(bin_offsets_kernel, dim3((g.nbins + 255) / 256), dim3(256), 0, ctx->stream, g.nbins, blocks,
                           t.bin_start, block_hist);
        for (int wdw = 0; wdw < windows; ++wdw) {
            const int b_lo = wdw * win, b_hi = b_lo + win < g.nbins ? b_lo + win : g.nbins;
            hipLaunchKernelGGL(bin_scatter_kernel<true>, dim3(blocks), dim3(threads), hist_bytes, ctx->stream, g, n,
                               u, v, uv_stride, wbin, t.bin_start, t.cursor, block_hist, (RecWord *)ctx->recs.ptr,
                               b_lo, b_hi, cap, t.scalars);
        }
    } else
        hipLaunchKernelGGL(bin_scatter_kernel<false>, dim3(blocks), dim3(threads), 0, ctx->stream, g, n, u, v,
                           uv_stride, wbin, t.bin_start, t.cursor, block_hist, (RecWord *)ctx->recs.ptr, 0, g.nbins,
                           cap, t.scalars);
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

}  // namespace gridhip
