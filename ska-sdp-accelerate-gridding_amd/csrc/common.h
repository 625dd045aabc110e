// libgridhip internal declarations (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/gridhip.h"

namespace gridhip {

// One visibility after the binning pre-pass: where its footprint starts inside the tile, which
// kernel slice it uses and where its value lives in the caller's array.
// Records do not depend on the visibility values: the image and PSF passes of do_imaging
// (src/Gridding.hs:538,541) can share one binning.
// In memory a record is one 64-bit word (RecWord): orig | kslice << ob | lx << (ob + kb) | ly << (ob + kb + 7), with
// the field widths ob, kb taken from the call's sizes (Geom; 27 + 13 + 14 = 54 bits in the headline case).  VisRec is
// the unpacked form the kernels work with.
struct VisRec {
    int32_t lxy;     // ly0 << 16 | lx0 : footprint origin relative to the tile's LDS region
    int32_t kslice;  // (wbin*Q + yf)*Q + xf : which [gh][gw] kernel slice
    int32_t orig;    // index in the caller's arrays (vis is gathered from / degrid writes there)
};
typedef unsigned long long RecWord;

// Geometry of one gridding call, shared by host and device code.
struct Geom {
    int64_t H, Wd;        // grid rows, columns
    int32_t W, Q, gh, gw; // kernel table dims
    int32_t Tx, Ty;       // tile width / height (cells, each <= 128; any size: the binning divides)
    int32_t offx, offy;   // coordinate offsets so tile indices start at 0
    int32_t ntx, nty;     // tiles per row / column
    int32_t ntiles;       // ntx*nty
    int32_t ngroups;      // w-plane groups (bins = ngroups*ntiles)
    int32_t nbins;
    int32_t ldw;          // LDS row pitch in cells
    int32_t lrows, lcols; // valid LDS region: T+gh-1 rows, T+gw-1 columns
    int32_t chunk;        // max visibilities per work item
    int32_t dbg;          // ablation switch for tuning runs (0 = off)
    int32_t per_vis;      // 1: the kernel table holds one [gh][gw] slice per visibility (aw gridders)
    int32_t nrec;         // record slots the pre-pass may have filled (n * P): work items stay inside them
    int32_t nvis;         // visibilities of the call: a record's `orig` is below it
    int32_t nslices;      // [gh][gw] slices in the kernel table: a record's `kslice` is below it
    // Sub-footprints.  A kernel the tap-reusing tile kernel has no instantiation for (supports above 16, non-square
    // ones) is cut into py x px square parts of side gh = gw = sub <= 16 (zero-padded: pad_kernels): every
    // visibility becomes P = py * px records, each with its own footprint origin, tile and slice
    // (kslice * P + part), and the kernels see nothing but more visibilities of a small square support.
    // fgh x fgw is the kernel the caller passed (the footprint origin is x - fgw / 2, y - fgh / 2).
    int32_t fgh, fgw, px, py, P;
    int32_t ob, kb;       // record layout: bits of the orig and kslice fields (set_rec_bits)
    int32_t dense;        // which table of walker weights (tile_sorted.hip): 1 = the steep one (make_geom)
    int32_t npersist;     // tap-reusing kernel: work-groups [0, npersist) run until the queues are empty; the others
    int32_t budget;       // take at most `budget` work items and leave (option "yield_cus", tile_sorted.hip)
    int32_t imoff;        // tap-reusing kernel: bytes from the re to the im plane of the LDS tile; 0 = the fixed 65528 that a DS
                          // instruction's offset field holds ("bigtile": a tile that uses all of the LDS, tile_sorted.hip)
    // division by Tx, Ty, W, P, px in the counting sweep without a divide: x / d == umulhi(x, m) >> s for 0 <= x < 2^31
    // (set_div_magic; the sweep is bound by its instruction count, and a 32-bit divide is ~30 instructions)
    uint32_t mTx, mTy, mW, mP, mPx;
    int32_t sTx, sTy, sW, sP, sPx;
};

// m, s with x / d == (x * m) >> (32 + s) for every 0 <= x < 2^31 (round-up method: m = ceil(2^(31 + l) / d),
// l = ceil(log2 d), which fits 32 bits because d > 2^(l - 1)); d = 1 is s = -1 (the quotient is x)
static inline void div_magic(int64_t d, uint32_t *m, int32_t *s)
{
    if (d <= 1) {
        *m = 0;
        *s = -1;
        return;
    }
    int l = 0;
    while (((int64_t)1 << l) < d) ++l;
    const unsigned __int128 num = (unsigned __int128)1 << (31 + l);
    *m = (uint32_t)((num + (unsigned __int128)d - 1) / (unsigned __int128)d);
    *s = l - 1;
}
__device__ __forceinline__ uint32_t udiv_magic(uint32_t x, uint32_t m, int32_t s)
{
    return s < 0 ? x : __umulhi(x, m) >> s;
}
static inline void set_div_magic(Geom *g)
{
    div_magic(g->Tx, &g->mTx, &g->sTx);
    div_magic(g->Ty, &g->mTy, &g->sTy);
    div_magic(g->W, &g->mW, &g->sW);
    div_magic(g->P, &g->mP, &g->sP);
    div_magic(g->px, &g->mPx, &g->sPx);
}

static inline int bits_for(int64_t count)  // bits that hold 0 .. count-1
{
    int b = 1;
    while (((int64_t)1 << b) < count) ++b;
    return b;
}
// field widths of the packed record from the call's sizes; a call whose fields do not fit 64 bits is cut into
// several (api.hip: more than 2^50 / slices visibilities)
static inline void set_rec_bits(Geom *g)
{
    g->ob = bits_for(g->nvis);
    int64_t ks = g->nslices;
    if (g->per_vis && g->nvis > ks) ks = g->nvis;  // (aw gridders: kslice starts as the visibility's index)
    g->kb = bits_for(ks);
}
static inline bool rec_fits(const Geom &g, int limit = 64) { return g.ob + g.kb + 14 <= limit; }
__host__ __device__ __forceinline__ RecWord rec_pack(const Geom &g, int32_t lxy, int32_t kslice, int32_t orig)
{
    return (RecWord)(uint32_t)orig | (RecWord)(uint32_t)kslice << g.ob | (RecWord)((uint32_t)lxy & 0x7f) << (g.ob + g.kb) |
           (RecWord)(((uint32_t)lxy >> 16) & 0x7f) << (g.ob + g.kb + 7);
}

struct Options {
    int64_t tile = 0, block = 0, chunk = 0, wgroups = 0, variant = 0, sort = 0, dbg = 0, prepass = 0, fault_inject = 0, aw_cache = 1, tile_x = 0, tile_y = 0, coarse_shift = 0, scatter_chunk = 0, count_unroll = 0, rec_bits = 0, wtable = 0, reserve_cus = 0, subfoot = 0, bigtile = 0, yield_cus = 0;
};

struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace gridhip

struct gridhip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t order_ev = nullptr;  // orders a newly selected stream after the previous one (gridhip_set_stream)
    std::string err;
    gridhip::Options opt;
    // device scratch, grown on demand (never inside a timed/captured region after warm-up)
    gridhip::Workspace recs;    // RecWord[n]
    gridhip::Workspace tables;  // bin_count / bin_start / work_start / cursors / scalars
    gridhip::Workspace stage;   // staging for the host-pointer entry points
    gridhip::Workspace sorted;  // sorted-list scratch of the tap-reusing tile kernel (tile_sorted.hip)
    gridhip::Workspace recs_raw;   // 8-byte pre-records the counting sweep leaves for the scatter (bin.hip)
    gridhip::Workspace recs_tmp;   // coarse-binned records between the two scatter levels of the pre-pass (bin.hip)
    gridhip::Workspace blockhist;  // [pre-pass work-groups][nbins] histograms -> first slots
    gridhip::Workspace ktab;       // kernel table cut into zero-padded square parts (sub-footprints, api.hip)
    gridhip::Workspace aw;         // aw gridders: pair slots, pair kernels, key hash table, table of distinct kernels
    int32_t *d_scalars = nullptr;  // [0]=dropped (wbin out of range), [2]=errors, [4..19] work queues,
                                   // [20..27] clock stamps of the last sorted tile kernel, [28..30] aw gridders, [32..] profile
    int num_cu = 256;
    int max_lds = 160 * 1024;
    bool timing = false;
    // HIP events of the last EV_RING timed device calls (3 per call: start, pre-pass done, kernel done), so that a
    // benchmark loop can read every step's device times afterwards instead of synchronising inside the loop
    static constexpr int EV_RING = 64;
    hipEvent_t ev[EV_RING * 3] = {};
    int64_t ev_calls = 0;  // timed calls recorded so far; call c uses slots (c % EV_RING) * 3 ...
    bool ev_open = false;  // a call has recorded its first event and not yet its last
    uint32_t attr_mask = 0;  // pre-pass kernels whose dynamic-LDS limit has been raised
    std::unordered_set<const void *> lds_raised;  // tile kernels whose dynamic-LDS limit has been raised
    // cached hipFFT Z2Z plans (imaging.hip): do_imaging with w_cache_imaging alternates between the kernel generator's
    // size and the image's, and creating a plan costs milliseconds
    void *fft_plan[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t fft_n[4] = {0, 0, 0, 0};
    int fft_next = 0;  // the slot the next new size replaces
    // device blocks of the imaging entry points (imaging.hip: DevBuf), kept between calls: a resident imaging call
    // (gridhip_do_imaging_dev) then neither allocates nor frees - hipFree synchronises the whole device.  All of a
    // context's work is ordered on one stream (gridhip_set_stream orders a new one after the old), so a block handed
    // back by one call may be handed out to the next without waiting.
    std::vector<std::pair<void *, size_t>> pool_free;
    // the last w-kernel table w_cache_imaging built (imaging.hip): the table depends on the field of view, the w-planes
    // and the kernel's shape only, and an imaging run calls with the same ones again and again (image and PSF, every
    // major cycle) - a call whose planes match takes the table as it is (21 planes: 1.6 ms of small kernels saved)
    struct {
        double theta = 0.0;
        int64_t wstep = 0, wmin = 0, nplanes = 0, npixFF = 0, S = 0, Q = 0;
        void *ptr = nullptr;
        size_t bytes = 0;
    } wk_cache;
    // which gridder the last convgrid / convgrid2 / degrid2 / plan call used (read-only option "last_path"):
    // 1 = tap-reusing tile kernel, 2 = the same through sub-footprints (one record per spatial part of the kernel),
    // 3 = general tile kernel (small problems; shapes or sizes the tap-reusing kernel does not take),
    // 4 = direct global-atomic scatter (option variant = 1, supports too large for an LDS tile)
    int last_path = 0;
    // the geometry that call ran with (read-only options "last_wgroups", "last_tile_x", "last_tile_y", "last_bigtile")
    int last_geom[4] = {0, 0, 0, 0};
};

namespace gridhip {

int fail(gridhip_ctx *ctx, int code, const char *fmt, ...);

#define GH_CHECK_HIP(ctx, call)                                                          \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess)                                                           \
            return gridhip::fail((ctx), e__ == hipErrorOutOfMemory ? GRIDHIP_ENOMEM      \
                                                                   : GRIDHIP_EHIP,       \
                                 "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                                 __FILE__, __LINE__);                                    \
    } while (0)

#define GH_CHECK(expr)              \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != GRIDHIP_OK)     \
            return rc__;            \
    } while (0)

int ws_reserve(gridhip_ctx *ctx, Workspace &ws, size_t bytes);

// HIP events of a timed call: i = 0: it starts, 1: its pre-pass is enqueued, 2: its dominant kernel is enqueued (the
// call is then readable with gridhip_timing)
static inline void mark(gridhip_ctx *ctx, int i)
{
    if (!ctx->timing) return;
    if (i == 0) ctx->ev_open = true;
    if (!ctx->ev_open) return;
    (void)hipEventRecord(ctx->ev[(ctx->ev_calls % gridhip_ctx::EV_RING) * 3 + i], ctx->stream);
    if (i == 2) {
        ctx->ev_open = false;
        ++ctx->ev_calls;
    }
}

// geometry / option resolution (host)
// parts: the kernel slices per (plane, sub-pixel offset) the tap-reusing kernel's sort will count (sub-footprints: P; 1
// otherwise) - the big tile must leave room for that histogram; 0 = never a big tile (aw gridders)
int make_geom(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t W, int64_t Q, int64_t gh,
              int64_t gw, int64_t n, Geom *g, int *block, size_t *lds_bytes, int parts = 1);
// ---- device-side coordinate math ---------------------------------------------------------
// frac_coord of src/Gridding.hs:126-140, bit-for-bit with the oracle: contraction is switched
// off for this block so `halfn + p*n` stays a rounded multiply followed by a rounded add
// (HIP's __dmul_rn/__dadd_rn are plain operators and would still be fused; checked in the ISA).
__device__ __forceinline__ void frac_coord_dev(int64_t n, int32_t qpx, double p, int64_t *flx,
                                               int32_t *fr)
{
#pragma clang fp contract(off)
    const double halfnf = (double)(n / 2);
    const double nf = (double)n;
    const double qpxf = (double)qpx;
    const double qpxfrac = 0.5 / qpxf;  // exact for power-of-two qpx, correctly rounded otherwise
    const double pn = p * nf;
    const double x = halfnf + pn;
    const double fl = floor(x + qpxfrac);
    const int64_t f = (int64_t)fl;
    const double xd = x - (double)f;
    const double d = xd * qpxf;
    int32_t r = (int32_t)round(d);
    r = r < 0 ? 0 : r;
    r = r > qpx - 1 ? qpx - 1 : r;
    *flx = f;
    *fr = r;
}

// kernel launchers (each enqueues on ctx->stream)
// zero_out (degrid2): the counting sweep writes a zero prediction for every visibility it drops (no tap inside the
// grid, wbin out of range), so that the caller's array needs no clearing pass
int launch_bin(gridhip_ctx *ctx, const Geom &g, int64_t n, const double *u, const double *v,
               int64_t uv_stride, const int64_t *wbin, double2 *zero_out = nullptr);
int launch_tile_grid(gridhip_ctx *ctx, const Geom &g, int block, size_t lds_bytes, int64_t n,
                     const double *gcf, const double *vis, double *grid);
bool sorted_plan(const gridhip_ctx *ctx, const Geom &g, int block, int *nkeys, int *maxchunk, size_t *lds_bytes);
int launch_tile_grid_sorted(gridhip_ctx *ctx, const Geom &g, int block, size_t lds_bytes, int nkeys, int maxchunk,
                            int64_t n, const double *gcf, const double *vis, double *grid, bool degrid);
int launch_tile_degrid(gridhip_ctx *ctx, const Geom &g, int block, size_t lds_bytes, int64_t n,
                       const double *gcf, const double *grid, double *vis_out);
int launch_direct_grid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n,
                       int64_t W, int64_t Q, int64_t gh, int64_t gw, const double *gcf,
                       const double *u, const double *v, int64_t uv_stride,
                       const int64_t *wbin, const double *vis);
// Everything a gridding call decides before it enqueues: geometry, which tile kernel, scratch sizes.
struct Prep {
    Geom g;
    int block = 0;
    size_t lds = 0, lds_sorted = 0;
    int nkeys = 0, batch = 0;
    bool sorted = false;  // the tap-reusing tile kernel (tile_sorted.hip)
    bool direct = false;  // no LDS tile possible: direct global-atomic scatter
    int64_t nrec = 0;     // records the pre-pass produces at most: n * g.P
};
int prepare(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t W, int64_t Q, int64_t gh, int64_t gw, int64_t n, Prep *p);
// the kernel table the tile kernels read: gcf itself, or (sub-footprints) its zero-padded parts in ctx->ktab
int tile_kernels(gridhip_ctx *ctx, const Prep &p, const double *gcf, const double **out);
int launch_simple_grid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n,
                       const double *u, const double *v, int64_t uv_stride, const double *vis);

// layout of the `tables` workspace (all int32 unless noted); see bin.hip
struct Tables {
    int32_t *bin_count;   // [nbins]
    int32_t *bin_start;   // [nbins+1] exclusive scan of bin_count
    int32_t *work_start;  // [ngroups][ntiles+1] exclusive scan of chunks per bin, per group
    int32_t *cursor;      // [nbins] scatter cursors
    int32_t *scalars;     // = ctx->d_scalars
};
void fft_release(gridhip_ctx *ctx);
// a plan whose records live in the context's own scratch (plan.hip): for callers that keep it for one call's passes
int plan_create_borrowed(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t n, int64_t W, int64_t Q, int64_t gh, int64_t gw,
                         const double *u, const double *v, int64_t uv_stride, const int64_t *wbin, gridhip_plan **out);
Tables tables_of(gridhip_ctx *ctx, const Geom &g);
size_t tables_bytes(const Geom &g);

}  // namespace gridhip
