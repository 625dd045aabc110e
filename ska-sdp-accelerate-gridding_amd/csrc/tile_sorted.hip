// Tap-reusing variant of the tile gridder ("sort" option).
//
// The plain tile kernel streams one [S][S] kernel slice (3.6 KB at 15x15) from L2 for every
// visibility; at 10^8 visibilities that L2->CU stream (360 GB) is what bounds it.  Within one
// work item (w-group x tile, <= ~3000 visibilities) only ~1000 distinct slices occur, so this
// variant first orders the work item's records by slice inside LDS (counting sort: histogram,
// scan, scatter) and stages the visibility values next to them; each wave then walks a
// contiguous piece of the sorted list run by run, keeping a run's taps in registers and fetching
// the next run's taps while the current run is accumulated.  Tap traffic drops by the mean run
// length (~3x on the uniform benchmark; far more on real, w-coherent data).
//
// LDS plan (T=64, 15x15): tile 98.6 KB | histogram 4.4 KB | values 16 B x batch | meta 4 B x batch
// (batch = 3008).  The tile stays resident for the whole work item (flushed once); the work item's
// records pass through in batches.
#include "tile_common.h"

namespace gridhip {

// DEGRID = false: accumulate vis * taps into the tile and flush it onto `grid` (convgrid2).
// DEGRID = true : the tile is loaded from `grid` once and each visibility's sum over taps of
//                 taps * tile is written to vis_out[orig] (degrid2); `vis` is then the output.
template <int S, bool DEGRID>
__global__ void __launch_bounds__(1024) tile_grid_sorted_kernel(Geom g, const VisRec *__restrict__ recs,
                                                                const int32_t *__restrict__ bin_start,
                                                                const int32_t *__restrict__ work_start,
                                                                const double2 *__restrict__ gcf,
                                                                double2 *__restrict__ vis,
                                                                double *__restrict__ grid, int nkeys, int batch,
                                                                int32_t *__restrict__ scalars)
{
    extern __shared__ double lds[];
    constexpr int S2 = S * S;
    constexpr int NSTEP = (S2 + 63) / 64;
    constexpr int TAIL = S2 - (NSTEP - 1) * 64;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int plane = g.lrows * g.ldw;
    // LDS: tile (re plane, im plane) | histogram | staged values | staged meta
    double *lre = lds, *lim = lds + plane;
    int32_t *hist = reinterpret_cast<int32_t *>(lds + 2 * plane);
    const int hist_words = (nkeys + 1 + 3) & ~3;  // keep what follows 16-byte aligned
    double2 *vals = reinterpret_cast<double2 *>(hist + hist_words);
    uint32_t *meta = reinterpret_cast<uint32_t *>(vals + batch);
    int32_t *wsum = reinterpret_cast<int32_t *>(vals);  // per-wave totals of the scan (before vals is filled)
    int32_t *s_item = reinterpret_cast<int32_t *>(meta + batch);  // 16 B past the staging (sorted_plan)

    // Persistent work-groups: the launch has one work-group per CU slot; each pulls work items of
    // "its" w-group (blockIdx % ngroups: the group whose kernel planes this XCD's L2 holds) from a
    // queue counter, then helps the other groups when its own queue is empty.  Every wave leaves
    // the loop once all queues are exhausted; there is no waiting on other work-groups.
    int32_t *queue = scalars + 4;
    for (int turn = 0; turn < g.ngroups;) {
    const int grp = (blockIdx.x + turn) % g.ngroups;
    if (tid == 0) *s_item = atomicAdd(&queue[grp], 1);
    __syncthreads();
    const int item = *s_item;
    __syncthreads();
    WorkItem w;
    if (!find_work_at(g, bin_start, work_start, grp, item, &w)) {
        ++turn;  // this group's queue is exhausted (uniform across the work-group)
        continue;
    }
    const int first_plane = (grp * g.W + g.ngroups - 1) / g.ngroups;  // smallest wb with wb*ng/W == grp
    const int first_slice = first_plane * g.Q * g.Q;

    const int tx = w.tile % g.ntx, ty = w.tile / g.ntx;
    const int64_t ox = (int64_t)tx * g.T - g.offx, oy = (int64_t)ty * g.T - g.offy;
    const int ncell = g.lrows * g.lcols;
    if (!DEGRID) {
        double2 *z = reinterpret_cast<double2 *>(lds);
        for (int i = tid; i < plane; i += nthr) z[i] = make_double2(0.0, 0.0);
    } else {
        const double2 *gsrc = reinterpret_cast<const double2 *>(grid);
        for (int c = tid; c < ncell; c += nthr) {
            const int r_ = c / g.lcols, c_ = c - r_ * g.lcols;
            const int64_t gx = ox + c_, gy = oy + r_;
            double2 v = make_double2(0.0, 0.0);
            if (gx >= 0 && gy >= 0 && gx < g.Wd && gy < g.H) v = gsrc[gy * g.Wd + gx];
            lre[r_ * g.ldw + c_] = v.x;
            lim[r_ * g.ldw + c_] = v.y;
        }
    }

    int loff[NSTEP];
    const bool tail_ok = lane < TAIL || TAIL == 64;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        int t = s * 64 + lane;
        if (s == NSTEP - 1 && !tail_ok) t = lane - 32;  // adds 0.0 on an idle bank pair (see tile_grid.hip)
        loff[s] = (t / S) * g.ldw + (t % S);
    }
    const int ttail = tail_ok ? (NSTEP - 1) * 64 + lane : 0;

    // The tile stays in LDS for the whole work item; its records go through in batches that fit
    // next to it.
    for (int b_lo = w.v_lo; b_lo < w.v_hi; b_lo += batch) {
        const int cnt = min(batch, w.v_hi - b_lo);

        // ---- counting sort of this batch's records by kernel slice ---------------------------
        __syncthreads();  // previous batch fully consumed (and, first time, the tile zeroed)
        for (int i = tid; i <= nkeys; i += nthr) hist[i] = 0;
        __syncthreads();
        for (int r = tid; r < cnt; r += nthr) {
            const VisRec rec = load_rec(recs, b_lo + r);
            const int key = rec.kslice - first_slice;
            if ((unsigned)key < (unsigned)nkeys) atomicAdd(&hist[key], 1);
        }
        __syncthreads();
        {   // exclusive scan of hist[0..nkeys): each thread owns a contiguous strip
            const int per = (nkeys + nthr - 1) / nthr;
            const int lo = min(tid * per, nkeys), hi = min(lo + per, nkeys);
            int s = 0;
            for (int i = lo; i < hi; ++i) s += hist[i];
            int incl = s;  // inclusive scan across the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int base = incl - s;
            for (int ww = 0; ww < wave; ++ww) base += wsum[ww];
            __syncthreads();  // wsum is about to be overwritten by vals
            for (int i = lo; i < hi; ++i) {
                const int c = hist[i];
                hist[i] = base;
                base += c;
            }
        }
        __syncthreads();
        int bad = 0;
        for (int r = tid; r < cnt; r += nthr) {
            const VisRec rec = load_rec(recs, b_lo + r);
            const int key = rec.kslice - first_slice;
            if ((unsigned)key >= (unsigned)nkeys) {  // cannot happen unless binning and kernel disagree
                ++bad;
                continue;
            }
            const int pos = atomicAdd(&hist[key], 1);
            if ((unsigned)pos < (unsigned)batch) {
                meta[pos] = ((uint32_t)key << 16) | (uint32_t)((rec.lxy >> 16) << 8) | (uint32_t)(rec.lxy & 0xff);
                if (DEGRID)
                    reinterpret_cast<int32_t *>(vals)[pos] = rec.orig;
                else
                    vals[pos] = vis[rec.orig];
            } else
                ++bad;
        }
        if (bad) atomicAdd(&scalars[2], bad);
        __syncthreads();
        // records that were rejected leave holes at the end of the sorted list: hist[nkeys-1] is now
        // the number actually staged
        const int staged = hist[nkeys - 1];

        // ---- each wave walks a contiguous piece of the sorted list, run by run ---------------
        const int seg_lo = (int)(((int64_t)staged * wave) / nw), seg_hi = (int)(((int64_t)staged * (wave + 1)) / nw);
        if (seg_lo < seg_hi) {
            auto issue = [&](double2(&k)[NSTEP], int key) {
                key = min(max(key, 0), nkeys - 1);  // never form an address outside the kernel table
                const double2 *kp = gcf + (size_t)(first_slice + key) * S2;
#pragma unroll
                for (int s = 0; s < NSTEP - 1; ++s) k[s] = kp[s * 64 + lane];
                k[NSTEP - 1] = kp[ttail];
            };
            // run starting at q0 with key key0: its length (<= 63) and the key that follows it
            auto scan_run = [&](int q0, int key0, int &len, int &nextkey) {
                const int idx = min(q0 + lane, seg_hi - 1);
                const int k = (int)(meta[idx] >> 16);
                const bool diff = (k != key0) && (q0 + lane < seg_hi);
                const unsigned long long b = __ballot(diff);
                int first = b ? (int)__builtin_ctzll(b) : 63;
                first = min(first, 63);
                len = min(first, seg_hi - q0);
                len = max(len, 0);
                nextkey = __builtin_amdgcn_readfirstlane(__shfl(k, min(len, 63), 64));
                len = __builtin_amdgcn_readfirstlane(len);
            };
            auto process = [&](const double2(&k)[NSTEP], int q0, int len) {
                uint32_t m = meta[min(q0, seg_hi - 1)];
                if (!DEGRID) {
                    double2 v = vals[min(q0, seg_hi - 1)];
                    // Consume the two reads here: otherwise their wait lands in the loop header, where it
                    // merges with the back edge into lgkmcnt(0) and every trip would drain the previous
                    // visibility's eight ds_add_f64 before starting (checked in the ISA).
                    asm volatile("" ::"v"(m), "v"(v.x), "v"(v.y));
                    for (int i = 0; i < len; ++i) {
                        const int qn = min(q0 + i + 1, seg_hi - 1);
                        const uint32_t mn = meta[qn];  // next record's LDS reads go out before this one's atomics
                        const double2 vn = vals[qn];
                        const int lbase = (int)((m >> 8) & 0xff) * g.ldw + (int)(m & 0xff);
#pragma unroll
                        for (int s = 0; s < NSTEP; ++s) {
                            double re = v.x * k[s].x - v.y * k[s].y;
                            double im = v.x * k[s].y + v.y * k[s].x;
                            if (s == NSTEP - 1 && TAIL != 64) {
                                re = tail_ok ? re : 0.0;
                                im = tail_ok ? im : 0.0;
                            }
                            const int a = lbase + loff[s];
                            __hip_atomic_fetch_add(&lre[a], re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(&lim[a], im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        m = mn;
                        v = vn;
                    }
                } else {
                    const int32_t *origs = reinterpret_cast<const int32_t *>(vals);
                    int32_t o = origs[min(q0, seg_hi - 1)];
                    asm volatile("" ::"v"(m), "v"(o));  // same reason as above
                    for (int i = 0; i < len; ++i) {
                        const int qn = min(q0 + i + 1, seg_hi - 1);
                        const uint32_t mn = meta[qn];
                        const int32_t on = origs[qn];
                        const int lbase = (int)((m >> 8) & 0xff) * g.ldw + (int)(m & 0xff);
                        double sr = 0.0, si = 0.0;
#pragma unroll
                        for (int s = 0; s < NSTEP; ++s) {
                            const int a = lbase + loff[s];
                            const double gr = lre[a], gi = lim[a];
                            double pr = k[s].x * gr - k[s].y * gi;
                            double pi = k[s].x * gi + k[s].y * gr;
                            if (s == NSTEP - 1 && TAIL != 64) {
                                pr = tail_ok ? pr : 0.0;
                                pi = tail_ok ? pi : 0.0;
                            }
                            sr += pr;
                            si += pi;
                        }
                        sr = wave_sum_lane63(sr);
                        si = wave_sum_lane63(si);
                        if (lane == 63) vis[o] = make_double2(sr, si);
                        m = mn;
                        o = on;
                    }
                }
            };

            // Three tap sets used in strict rotation: while run r is accumulated from one set, the
            // taps of runs r+1 and r+2 are in flight into the other two.  The only loop exit sits
            // after a full A+B+C trip and every set is "used" after it, so LLVM cannot sink a
            // prefetch past the exit test; runs of length 0 pad the tail.
            double2 kA[NSTEP], kB[NSTEP], kC[NSTEP];
            int q = seg_lo, lenA, lenB, lenC, keyA, keyB, keyC;
            keyA = __builtin_amdgcn_readfirstlane((int)(meta[q] >> 16));
            issue(kA, keyA);
            scan_run(q, keyA, lenA, keyB);
            issue(kB, keyB);
            scan_run(q + lenA, keyB, lenB, keyC);
            for (;;) {
                issue(kC, keyC);
                asm volatile("" ::: "memory");  // compiler fence: the prefetch may not sink below this point
                __builtin_amdgcn_sched_barrier(0);
                process(kA, q, lenA);
                __builtin_amdgcn_sched_barrier(0);
                q += lenA;
                scan_run(q + lenB, keyC, lenC, keyA);

                issue(kA, keyA);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                process(kB, q, lenB);
                __builtin_amdgcn_sched_barrier(0);
                q += lenB;
                scan_run(q + lenC, keyA, lenA, keyB);

                issue(kB, keyB);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                process(kC, q, lenC);
                __builtin_amdgcn_sched_barrier(0);
                q += lenC;
                if (q >= seg_hi) break;
                scan_run(q + lenA, keyB, lenB, keyC);
            }
            process(kA, q, 0);  // keep the last prefetches "used" on the exit path
            process(kB, q, 0);
        }
    }
    __syncthreads();  // all waves done with the tile (and with this item's staging)
    if (DEGRID) continue;

    // ---- flush the cells that exist in the grid
    // Consecutive lanes take (re, im) of consecutive cells, so one atomic instruction covers a
    // contiguous 512-byte run of the interleaved grid row (memory-side fp64 atomics run at full rate
    // on contiguous runs and at half of it on the stride-16 pattern of one component at a time).
    for (int e = tid; e < 2 * ncell; e += nthr) {
        const int c = e >> 1, comp = e & 1;
        const int r_ = c / g.lcols, c_ = c - r_ * g.lcols;
        const int64_t gx = ox + c_, gy = oy + r_;
        if (gx < 0 || gy < 0 || gx >= g.Wd || gy >= g.H) continue;
        const double val = comp ? lim[r_ * g.ldw + c_] : lre[r_ * g.ldw + c_];
        if (val == 0.0) continue;
        unsafeAtomicAdd(grid + 2 * (gy * g.Wd + gx) + comp, val);
    }
    __syncthreads();  // the tile is re-initialised by the next work item
    }  // persistent loop
}

// Can the sorted variant run this geometry?  Needs a compile-time support, the per-group slice
// count to fit the 16-bit key, and room next to the tile for the histogram and a useful batch of
// staged records.  On success returns the LDS bytes and the batch size.
bool sorted_plan(const gridhip_ctx *ctx, const Geom &g, int block, int *nkeys, int *batch, size_t *lds_bytes)
{
    // square supports with a compile-time instantiation below
    if (g.gh != g.gw || g.gh < 5 || g.gh > 16) return false;
    if (g.per_vis || g.T > 128) return false;
    const int planes = (g.W + g.ngroups - 1) / g.ngroups + 1;  // groups differ by at most one plane
    const int64_t keys = (int64_t)planes * g.Q * g.Q;
    if (keys >= 65536) return false;
    const size_t tile = (size_t)g.lrows * g.ldw * 16;
    const size_t hist = (size_t)((keys + 1 + 3) & ~3) * 4;
    if (tile + hist + 512 * 20 + 512 > (size_t)ctx->max_lds) return false;
    int c = (int)(((size_t)ctx->max_lds - 1024 - tile - hist) / 20);
    (void)block;
    c &= ~63;
    if (c > 8192) c = 8192;
    if (c < 512) return false;
    *nkeys = (int)keys;
    *batch = c;
    *lds_bytes = tile + hist + (size_t)c * 20 + 16;  // + the work-queue slot
    return true;
}

int launch_tile_grid_sorted(gridhip_ctx *ctx, const Geom &g, int block, size_t lds_bytes, int nkeys, int batch,
                            int64_t n, const double *gcf, const double *vis, double *grid, bool degrid)
{
    Tables t = tables_of(ctx, g);
    const VisRec *recs = (const VisRec *)ctx->recs.ptr;
    // persistent work-groups: as many as can be resident (LDS-limited), pulling items from per-group queues
    int per_cu = (int)((size_t)ctx->max_lds / lds_bytes);
    per_cu = per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu;
    if (per_cu * block > 2048) per_cu = 2048 / block > 0 ? 2048 / block : 1;
    int nblk = ctx->num_cu * per_cu;
    nblk = ((nblk + g.ngroups - 1) / g.ngroups) * g.ngroups;
    const int most = work_blocks(g, n);
    if (nblk > most) nblk = most > g.ngroups ? (most / g.ngroups) * g.ngroups : g.ngroups;
    const dim3 gr(nblk), bl(block);
    GH_CHECK_HIP(ctx, hipMemsetAsync(t.scalars + 4, 0, 8 * sizeof(int32_t), ctx->stream));
#define GH_LAUNCH(S_, D_)                                                                                        \
    do {                                                                                                         \
        GH_CHECK(raise_lds(ctx, tile_grid_sorted_kernel<S_, D_>));                                               \
        hipLaunchKernelGGL((tile_grid_sorted_kernel<S_, D_>), gr, bl, lds_bytes, ctx->stream, g, recs,           \
                           t.bin_start, t.work_start, (const double2 *)gcf, (double2 *)vis, grid, nkeys, batch, \
                           t.scalars);                                                                           \
    } while (0)
#define GH_CASE(S_)             \
    case S_:                    \
        if (degrid)             \
            GH_LAUNCH(S_, true); \
        else                    \
            GH_LAUNCH(S_, false); \
        break;
    switch (g.gh) {
        GH_CASE(5) GH_CASE(6) GH_CASE(7) GH_CASE(8) GH_CASE(9) GH_CASE(10) GH_CASE(11) GH_CASE(12) GH_CASE(13)
        GH_CASE(14) GH_CASE(15) GH_CASE(16)
        default: return fail(ctx, GRIDHIP_EUNSUPPORTED, "no sorted instantiation for support %d", g.gh);
    }
#undef GH_CASE
#undef GH_LAUNCH
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

}  // namespace gridhip
