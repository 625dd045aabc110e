// Tile kernels: the w-projection gridder (convgrid / convgrid2, src/Gridding.hs:153-244) and
// its gather twin (degrid2) on CDNA4.
//
// One work item = (w-group, grid tile, chunk of <=chunk binned visibilities).  A work-group
// keeps the tile plus its kernel-support halo — (T+gh-1) x (T+gw-1) complex cells, planar
// re/im, 35 KB at T=32 / 100 KB at T=64 for 15x15 — in LDS (160 KB per CU).  Each wave takes
// one visibility at a time: its 8-byte record and its value arrive by broadcast loads, lanes map
// to kernel taps (64 consecutive taps per step) so the tap read from the [gh][gw] slice is one
// coalesced run and the accumulate is a conflict-free ds_add_f64 (row pitch chosen in ctx.hip).  The grid read-modify-write of the reference's `permute (+)`
// therefore never leaves the CU; HBM sees the tile once, when the work-group flushes it with
// global_atomic_add_f64 (neighbouring tiles overlap in their halos, and several chunks/groups
// may share a tile).  The next visibility's taps are fetched while the current one is being
// accumulated.
//
// fixoutofbounds (:883-891): the LDS region may hang over the grid edge; cells outside the
// grid are simply not flushed, i.e. out-of-range taps are dropped, never wrapped.
#include "tile_common.h"

namespace gridhip {

__global__ void __launch_bounds__(256) clear_ints_kernel(int32_t *a, int na, int32_t *b, int nb, int32_t *c, int nc)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na || i < nb || i < nc; i += gridDim.x * blockDim.x) {
        if (i < na) a[i] = 0;
        if (i < nb) b[i] = 0;
        if (i < nc) c[i] = 0;
    }
}

// Lane -> tap mapping: tap t = step*64 + lane, (i, j) = (t / gw, t % gw).  Taps of one slice
// are contiguous in memory, so a step reads one coalesced 1 KiB run; with the LDS row pitch
// congruent to gw modulo 32 (ctx.hip) tap t falls on 8-byte bank pair t mod 32, so the 32
// lanes an LDS instruction services together never conflict, whatever the footprint origin.
//
// S > 0: square S x S support known at compile time — fully unrolled, software-pipelined per
// wave (record two visibilities ahead, taps + value one ahead, accumulate the current one).
// S == 0: any gh x gw.
// DBG (tuning builds only): 1 = skip the LDS atomics, 2 = every visibility reads slice 0, 3 = both.
template <int S, int DBG>
__global__ void __launch_bounds__(1024) tile_grid_kernel(Geom g, const RecWord *__restrict__ recs,
                                                         const int32_t *__restrict__ bin_start,
                                                         const int32_t *__restrict__ work_start,
                                                         const double2 *__restrict__ gcf,
                                                         const double2 *__restrict__ vis,
                                                         double *__restrict__ grid)
{
    extern __shared__ double lds[];
    WorkItem w;
    if (!find_work(g, bin_start, work_start, &w)) return;

    const int tid = threadIdx.x;
    const int plane = g.lrows * g.ldw;  // doubles per plane
    {
        double2 *z = reinterpret_cast<double2 *>(lds);
        for (int i = tid; i < plane; i += blockDim.x) z[i] = make_double2(0.0, 0.0);  // 2*plane doubles
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    const int gh = S ? S : g.gh, gw = S ? S : g.gw;
    const int S2 = gh * gw;
    double *lre = lds, *lim = lds + plane;

    if (S > 0) {
        constexpr int NSTEP = S ? (S * S + 63) / 64 : 1;
        constexpr int TAIL = S * S - (NSTEP - 1) * 64;  // lanes with a tap in the last step
        // Per-lane LDS offsets of this lane's taps.  The loop below is branch-free: a lane
        // without a tap in the last step (lane >= TAIL) reads tap 0 (valid memory) and adds 0.0
        // to the cell of tap lane-32, which sits on a bank pair none of the step's real taps
        // use (tails of 32 taps or fewer: to one of the tail's own cells).  (A divergent tail block would hide its s_waitcnt from the other path and make
        // the compiler drain every outstanding load at the loop head.)
        int loff[NSTEP];
        const bool tail_ok = lane < TAIL || TAIL == 64;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            int t = s * 64 + lane;
            if (s == NSTEP - 1 && !tail_ok) t = TAIL > 32 ? lane - 32 : lane % TAIL;  // (a cell of the footprint)
            loff[s] = (t / S) * g.ldw + (t % S);
        }
        const int ttail = tail_ok ? (NSTEP - 1) * 64 + lane : 0;
        const int last = w.v_hi - 1;
        int vi = w.v_lo + wave;
        if (vi <= last) {
            // taps + value of one visibility -> registers (NSTEP+1 vector loads, nothing waited on here)
            auto issue = [&](const VisRec &r, double2 &v, double2(&k)[NSTEP]) {
                const double2 *kp = gcf + (size_t)(DBG >= 2 ? 0 : r.kslice) * S2;
                v = vis[r.orig];
#pragma unroll
                for (int s = 0; s < NSTEP - 1; ++s) k[s] = kp[s * 64 + lane];
                k[NSTEP - 1] = kp[ttail];
            };
            auto add = [&](int a, double re, double im) {
                if (DBG == 1 || DBG == 3) {
                    if (re == 12345.678 && im == 9.0) lre[a] = re;  // keeps loads + multiply alive
                } else {
                    __hip_atomic_fetch_add(&lre[a], re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&lim[a], im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            auto accum = [&](const VisRec &r, const double2 &v, const double2(&k)[NSTEP]) {
                const int lbase = (r.lxy >> 16) * g.ldw + (r.lxy & 0xffff);
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) {
                    double re = v.x * k[s].x - v.y * k[s].y;
                    double im = v.x * k[s].y + v.y * k[s].x;
                    if (s == NSTEP - 1 && TAIL != 64) {
                        re = tail_ok ? re : 0.0;
                        im = tail_ok ? im : 0.0;
                    }
                    add(lbase + loff[s], re, im);
                }
            };
            // Two register sets (A, B) used alternately so no tap register is ever copied (a copy
            // would have to wait for the loads just issued).  Two visibilities per trip and no
            // exit between a prefetch and its use: LLVM otherwise sinks the loads past the exit
            // test and the pipeline collapses.  The odd one out at the end is a repeat of the
            // last visibility with its value forced to zero (adds 0.0 to the same cells).
            double2 kA[NSTEP], kB[NSTEP], valA, valB;
            VisRec r0 = load_rec(recs, vi, g);                   // visibility t
            VisRec r1 = load_rec(recs, min(vi + nw, last), g);   // t+1
            issue(r0, valA, kA);
            for (; vi <= last; vi += 2 * nw) {
                const VisRec r2 = load_rec(recs, min(vi + 2 * nw, last), g);  // t+2
                issue(r1, valB, kB);
                __builtin_amdgcn_sched_barrier(0);
                accum(r0, valA, kA);
                __builtin_amdgcn_sched_barrier(0);
                const VisRec r3 = load_rec(recs, min(vi + 3 * nw, last), g);  // t+3
                issue(r2, valA, kA);
                __builtin_amdgcn_sched_barrier(0);
                if (vi + nw > last) valB = make_double2(0.0, 0.0);
                accum(r1, valB, kB);
                __builtin_amdgcn_sched_barrier(0);
                r0 = r2;
                r1 = r3;
            }
        }
    } else {
        // general shape: the taps of the next 64-tap step (and, at a visibility's last step, the first
        // step of the next visibility) are in flight while the current step is accumulated; loads are
        // unconditional (indices clamped), only the adds are predicated.
        const int dj = 64 % gw, di = 64 / gw;
        const int nstep = (S2 + 63) >> 6;
        const int last = w.v_hi - 1;
        int vi = w.v_lo + wave;
        if (vi <= last) {
            VisRec r = load_rec(recs, vi, g);
            VisRec rn = load_rec(recs, min(vi + nw, last), g);
            double2 val = vis[r.orig];
            double2 kv_next = gcf[(size_t)r.kslice * S2 + min(lane, S2 - 1)];
            for (; vi <= last; vi += nw) {
                const VisRec rnn = load_rec(recs, min(vi + 2 * nw, last), g);
                const double2 valn = vis[rn.orig];
                const double2 *kp = gcf + (size_t)r.kslice * S2;
                const double2 *kpn = gcf + (size_t)rn.kslice * S2;
                const int lbase = (r.lxy >> 16) * g.ldw + (r.lxy & 0xffff);
                int i = lane / gw, j = lane - i * gw;
                for (int s = 0; s < nstep; ++s) {
                    const int t = s * 64 + lane;
                    const double2 kv = kv_next;
                    kv_next = (s + 1 < nstep) ? kp[min(t + 64, S2 - 1)] : kpn[min(lane, S2 - 1)];
                    if (t < S2) {
                        const double re = val.x * kv.x - val.y * kv.y;
                        const double im = val.x * kv.y + val.y * kv.x;
                        const int a = lbase + i * g.ldw + j;
                        __hip_atomic_fetch_add(&lre[a], re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&lim[a], im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    j += dj;
                    i += di;
                    if (j >= gw) {
                        j -= gw;
                        ++i;
                    }
                }
                r = rn;
                rn = rnn;
                val = valn;
            }
        }
    }
    __syncthreads();

    // flush the cells that exist in the grid
    const int tx = w.tile % g.ntx, ty = w.tile / g.ntx;
    const int64_t ox = (int64_t)tx * g.Tx - g.offx, oy = (int64_t)ty * g.Ty - g.offy;
    const int ncell = g.lrows * g.lcols;
    // Consecutive lanes take (re, im) of consecutive cells, so one atomic instruction covers a
    // contiguous 512-byte run of the interleaved grid row (memory-side fp64 atomics run at full rate
    // on contiguous runs and at half of it on the stride-16 pattern of one component at a time).
    for (int e = tid; e < 2 * ncell; e += blockDim.x) {
        const int c = e >> 1, comp = e & 1;
        const int r_ = c / g.lcols, c_ = c - r_ * g.lcols;
        const int64_t gx = ox + c_, gy = oy + r_;
        if (gx < 0 || gy < 0 || gx >= g.Wd || gy >= g.H) continue;
        const double val = comp ? lim[r_ * g.ldw + c_] : lre[r_ * g.ldw + c_];
        if (val == 0.0) continue;
        unsafeAtomicAdd(grid + 2 * (gy * g.Wd + gx) + comp, val);
    }
}

// Gather twin: vis_out[orig] = sum_ij gcf[kslice][i][j] * G[y0+i][x0+j]; the tile (zero outside
// the grid) is staged in LDS once per work item, taps are summed across the wave.
__global__ void __launch_bounds__(1024) tile_degrid_kernel(Geom g, const RecWord *__restrict__ recs,
                                                           const int32_t *__restrict__ bin_start,
                                                           const int32_t *__restrict__ work_start,
                                                           const double2 *__restrict__ gcf,
                                                           const double2 *__restrict__ grid,
                                                           double2 *__restrict__ vis_out)
{
    extern __shared__ double lds[];
    WorkItem w;
    if (!find_work(g, bin_start, work_start, &w)) return;

    const int tid = threadIdx.x;
    const int plane = g.lrows * g.ldw;
    double *lre = lds, *lim = lds + plane;
    const int tx = w.tile % g.ntx, ty = w.tile / g.ntx;
    const int64_t ox = (int64_t)tx * g.Tx - g.offx, oy = (int64_t)ty * g.Ty - g.offy;
    const int ncell = g.lrows * g.lcols;
    for (int c = tid; c < ncell; c += blockDim.x) {
        const int r_ = c / g.lcols, c_ = c - r_ * g.lcols;
        const int64_t gx = ox + c_, gy = oy + r_;
        double2 v = make_double2(0.0, 0.0);
        if (gx >= 0 && gy >= 0 && gx < g.Wd && gy < g.H) v = grid[gy * g.Wd + gx];
        lre[r_ * g.ldw + c_] = v.x;
        lim[r_ * g.ldw + c_] = v.y;
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    const int gw = g.gw, S2 = g.gh * g.gw;
    const int dj = 64 % gw, di = 64 / gw;

    const int nstep = (S2 + 63) >> 6;
    const int last = w.v_hi - 1;
    int vi = w.v_lo + wave;
    if (vi <= last) {
        VisRec r = load_rec(recs, vi, g);
        VisRec rn = load_rec(recs, min(vi + nw, last), g);
        double2 kv_next = gcf[(size_t)r.kslice * S2 + min(lane, S2 - 1)];
        for (; vi <= last; vi += nw) {
            const VisRec rnn = load_rec(recs, min(vi + 2 * nw, last), g);
            const double2 *kp = gcf + (size_t)r.kslice * S2;
            const double2 *kpn = gcf + (size_t)rn.kslice * S2;
            const int lbase = (r.lxy >> 16) * g.ldw + (r.lxy & 0xffff);
            double sr = 0.0, si = 0.0;
            int i = lane / gw, j = lane - i * gw;
            for (int s = 0; s < nstep; ++s) {
                const int t = s * 64 + lane;
                const double2 kv = kv_next;
                kv_next = (s + 1 < nstep) ? kp[min(t + 64, S2 - 1)] : kpn[min(lane, S2 - 1)];
                if (t < S2) {
                    const int a = lbase + i * g.ldw + j;
                    const double gr = lre[a], gi = lim[a];
                    sr = fma(kv.x, gr, sr);
                    sr = fma(-kv.y, gi, sr);
                    si = fma(kv.x, gi, si);
                    si = fma(kv.y, gr, si);
                }
                j += dj;
                i += di;
                if (j >= gw) {
                    j -= gw;
                    ++i;
                }
            }
            sr = wave_sum_lane63(sr);
            si = wave_sum_lane63(si);
            if (lane == 63) vis_out[r.orig] = make_double2(sr, si);
            r = rn;
            rn = rnn;
        }
    }
}

int launch_tile_grid(gridhip_ctx *ctx, const Geom &g, int block, size_t lds_bytes, int64_t n,
                     const double *gcf, const double *vis, double *grid)
{
    Tables t = tables_of(ctx, g);
    const RecWord *recs = (const RecWord *)ctx->recs.ptr;
    const dim3 gr(work_blocks(g, n)), bl(block);
#define GH_LAUNCH(S_, D_)                                                                                   \
    do {                                                                                                    \
        GH_CHECK(raise_lds(ctx, tile_grid_kernel<S_, D_>));                                                 \
        hipLaunchKernelGGL((tile_grid_kernel<S_, D_>), gr, bl, lds_bytes, ctx->stream, g, recs, t.bin_start, \
                           t.work_start, (const double2 *)gcf, (const double2 *)vis, grid);                 \
    } while (0)
#define GH_CASE(S_) \
    case S_: GH_LAUNCH(S_, 0); break;
#ifdef GRIDHIP_TUNING
    if (g.gh == 15 && g.gw == 15 && g.dbg == 1)
        GH_LAUNCH(15, 1);
    else if (g.gh == 15 && g.gw == 15 && g.dbg == 2)
        GH_LAUNCH(15, 2);
    else if (g.gh == 15 && g.gw == 15 && g.dbg == 3)
        GH_LAUNCH(15, 3);
    else
#endif
    if (g.gh == g.gw)
        switch (g.gh) {
            GH_CASE(5) GH_CASE(6) GH_CASE(7) GH_CASE(8) GH_CASE(9) GH_CASE(10) GH_CASE(11) GH_CASE(12) GH_CASE(13)
            GH_CASE(14) GH_CASE(15) GH_CASE(16)
            default: GH_LAUNCH(0, 0);
        }
    else
        GH_LAUNCH(0, 0);
#undef GH_CASE
#undef GH_LAUNCH
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

int launch_tile_degrid(gridhip_ctx *ctx, const Geom &g, int block, size_t lds_bytes, int64_t n,
                       const double *gcf, const double *grid, double *vis_out)
{
    Tables t = tables_of(ctx, g);
    const RecWord *recs = (const RecWord *)ctx->recs.ptr;
    const dim3 gr(work_blocks(g, n)), bl(block);
    GH_CHECK(raise_lds(ctx, tile_degrid_kernel));
    hipLaunchKernelGGL(tile_degrid_kernel, gr, bl, lds_bytes, ctx->stream, g, recs, t.bin_start, t.work_start,
                       (const double2 *)gcf, (const double2 *)grid, (double2 *)vis_out);
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

}  // namespace gridhip
