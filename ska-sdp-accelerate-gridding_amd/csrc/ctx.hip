// Context, options, scratch memory and per-call geometry of libgridhip (gfx950).
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include "common.h"

namespace gridhip {

int fail(gridhip_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int ws_reserve(gridhip_ctx *ctx, Workspace &ws, size_t bytes)
{
    if (bytes <= ws.bytes) return GRIDHIP_OK;
    // growing scratch: earlier calls (possibly on another stream) may still be reading the old block
    GH_CHECK_HIP(ctx, hipDeviceSynchronize());
    if (ws.ptr) GH_CHECK_HIP(ctx, hipFree(ws.ptr));
    ws.ptr = nullptr;
    ws.bytes = 0;
    size_t want = bytes + bytes / 8 + 4096;  // head-room so nearby sizes do not reallocate
    GH_CHECK_HIP(ctx, hipMalloc(&ws.ptr, want));
    ws.bytes = want;
    return GRIDHIP_OK;
}

static int ilog2(int x)
{
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

// LDS row pitch (in cells).  Lanes take consecutive taps t -> (t / gw, t % gw); 64-bit LDS
// accesses are serviced 32 lanes at a time over 32 eight-byte bank pairs, so with
// pitch == gw (mod 32) tap t lands on bank pair (origin + t) mod 32 and any 32 consecutive taps
// are conflict-free, row breaks included (MI355X_MICROARCH.md §LDS).
static int lds_pitch(int lcols, int gw)
{
    int p = lcols;
    while (p % 32 != gw % 32) ++p;
    return p;
}

size_t tables_bytes(const Geom &g)
{
    size_t ints = (size_t)g.nbins + ((size_t)g.nbins + 1) +
                  (size_t)g.ngroups * ((size_t)g.ntiles + 1) + (size_t)g.nbins;
    return ints * sizeof(int32_t);
}

Tables tables_of(gridhip_ctx *ctx, const Geom &g)
{
    Tables t;
    int32_t *p = (int32_t *)ctx->tables.ptr;
    t.bin_count = p;
    p += g.nbins;
    t.bin_start = p;
    p += g.nbins + 1;
    t.work_start = p;
    p += (size_t)g.ngroups * (g.ntiles + 1);
    t.cursor = p;
    t.scalars = ctx->d_scalars;
    return t;
}

static int make_geom1(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t W, int64_t Q, int64_t gh, int64_t gw, int64_t n,
                      Geom *g, int *block, size_t *lds_bytes, int parts);

int make_geom(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t W, int64_t Q, int64_t gh,
              int64_t gw, int64_t n, Geom *g, int *block, size_t *lds_bytes, int parts)
{
    GH_CHECK(make_geom1(ctx, H, Wd, W, Q, gh, gw, n, g, block, lds_bytes, parts));
    if (g->imoff > 0) {
        // the big tile was sized with an estimate of the sort's histogram (8 w-groups); with the groups actually chosen
        // it must still fit beside the two planes, or the tap-reusing kernel would refuse the geometry: then the classic tile
        const int64_t planes = (W + g->ngroups - 1) / g->ngroups + 1;
        const size_t hist = (size_t)((planes * Q * Q * parts + 1 + 3) & ~(int64_t)3) * 4;
        if (2 * (size_t)g->imoff + hist + 128 > (size_t)ctx->max_lds) GH_CHECK(make_geom1(ctx, H, Wd, W, Q, gh, gw, n, g, block, lds_bytes, 0));
    }
    return GRIDHIP_OK;
}

static int make_geom1(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t W, int64_t Q, int64_t gh, int64_t gw, int64_t n,
                      Geom *g, int *block, size_t *lds_bytes, int parts)
{
    if (H <= 0 || Wd <= 0 || W <= 0 || Q <= 0 || gh <= 0 || gw <= 0)
        return fail(ctx, GRIDHIP_EINVAL, "non-positive dimension");
    if (H > (1 << 30) || Wd > (1 << 30) || W * Q * Q > (1LL << 30) || gh > 1024 || gw > 1024)
        return fail(ctx, GRIDHIP_EUNSUPPORTED, "shape outside tile-kernel limits");
    memset(g, 0, sizeof *g);
    g->H = H;
    g->Wd = Wd;
    g->W = (int32_t)W;
    g->Q = (int32_t)Q;
    g->gh = (int32_t)gh;
    g->gw = (int32_t)gw;
    g->nvis = (int32_t)(n > 0 ? n : 1);
    g->nrec = g->nvis;
    g->nslices = (int32_t)(W * Q * Q);
    g->fgh = (int32_t)gh;
    g->fgw = (int32_t)gw;
    g->px = g->py = g->P = 1;
    g->per_vis = 0;
    set_rec_bits(g);
    // test hook: rec_bits = 100 + t widens the kslice field until the record's fields take t bits (t = 64: they fill
    // the word exactly, as they do in every part of a call that had to be cut)
    if (ctx->opt.rec_bits >= 100 && ctx->opt.rec_bits <= 164 && g->ob + g->kb + 14 < ctx->opt.rec_bits - 100)
        g->kb = (int32_t)(ctx->opt.rec_bits - 100) - 14 - g->ob;

    const size_t lds_cap = (size_t)ctx->max_lds - 1024;
    // one plane (re or im) of the tap-reusing kernel's tile must fit below the fixed re / im distance (tile_sorted.hip) -
    // unless the tile is to use all of the LDS (the im plane then follows the re plane directly, at a distance the
    // kernel adds per tap step instead of carrying it in the instruction's offset field); what the sorter's histogram
    // needs (one counter per kernel slice of a w-group) is left free
    // ("bigtile": 0 = auto - where items are sparse, i.e. fewer than two visibilities per kernel slice and work item with
    // the classic tile: measured +2 % on the 8192^2 share of configuration 5, -4 % where the LDS unit binds; 1 = on;
    // 2 = off)
    const bool bt_ok = parts > 0 && ctx->opt.bigtile != 2 && ctx->opt.tile == 0 && ctx->opt.tile_x == 0 && gh == gw && gh >= 5 && gh <= 32;
    auto plane_for = [&](int tx, int ty) {
        return (size_t)lds_pitch(tx + (int)gw - 1, (int)gw) * (size_t)(ty + (int)gh - 1) * 8;
    };
    // The largest tile whose planes fit: tap reuse per work item grows with the tile's area (visibilities per
    // distinct kernel slice = n / (tiles W Q^2)) and the halo's share shrinks.  The row pitch comes in steps of
    // 32 cells (bank-conflict rule, lds_pitch), so the candidates are the widest tile of each pitch with the
    // tallest height that fits: 65 x 89 for a 15 x 15 kernel, against 64 x 64 as a square power of two.
    // how many w-groups a call gets when option "wgroups" does not say (see below, where it is applied): by the work
    // per tile in visibilities of 15 x 15 taps
    auto tiles_of = [&](int tx, int ty) {
        const int ox = (((int)gw - 1 + tx - 1) / tx) * tx, oy = (((int)gh - 1 + ty - 1) / ty) * ty;
        return (int64_t)((Wd - 1 + ox) / tx + 1) * (int64_t)((H - 1 + oy) / ty + 1);
    };
    auto auto_groups = [&](int64_t ntiles) {
        const double work = (double)n / (double)ntiles * ((double)gh * (double)gw / 225.0);
        return W < 8 ? 1 : work >= 8000.0 ? 8 : work >= 1200.0 ? 4 : work >= 600.0 ? 2 : 1;
    };
    // the sort's histogram: one counter per slice of a w-group - sized for 8 groups (or what option "wgroups" says) in
    // the classic tile, whose 24 KB of room hold fewer groups' too; for the groups the call will get in the big tile
    int64_t ng_est = ctx->opt.wgroups ? ctx->opt.wgroups : (W >= 8 ? 8 : 1);
    auto largest_tile = [&](bool big, int *otx, int *oty) {
        const size_t hist_need = (size_t)(((W + ng_est - 1) / ng_est + 2) * Q * Q * (parts > 0 ? parts : 1)) * 4 + 1024;
        const size_t hist_room = big ? hist_need : (hist_need > 24576 && hist_need < 65536 * 4 ? hist_need : 24576);
        const size_t plane_cap = big ? lds_cap : 65528;
        size_t best = 0;
        if (hist_room + 8192 > lds_cap) return best;
        for (int tx = 8; tx <= 128; ++tx) {
            const int pitch = lds_pitch(tx + (int)gw - 1, (int)gw);
            if (tx < 128 && lds_pitch(tx + 1 + (int)gw - 1, (int)gw) == pitch) continue;  // (not the widest of its pitch)
            int rows = (int)(plane_cap / ((size_t)pitch * 8));
            if ((size_t)rows * pitch * 16 > lds_cap - hist_room) rows = (int)((lds_cap - hist_room) / ((size_t)pitch * 16));
            int ty = rows - ((int)gh - 1);
            if (ty > 128) ty = 128;
            if (ty < 8) continue;
            if ((size_t)tx * ty > best) {
                best = (size_t)tx * ty;
                *otx = tx;
                *oty = ty;
            }
        }
        return best;
    };
    bool bigtile = false;
    int Tx = (int)ctx->opt.tile_x, Ty = (int)ctx->opt.tile_y;
    if (ctx->opt.tile) Tx = Ty = (int)ctx->opt.tile;  // option "tile": a square tile
    if ((Tx == 0) != (Ty == 0)) return fail(ctx, GRIDHIP_EINVAL, "tile_x and tile_y go together");
    if (Tx == 0) {
        size_t best = largest_tile(false, &Tx, &Ty);
        if (best > 0 && bt_ok) {
            const int64_t tiles = ((H + Ty - 1) / Ty) * ((Wd + Tx - 1) / Tx);
            const bool sparse = n >= ((int64_t)1 << 22) && n < 2 * tiles * W * Q * Q;
            int bx = 0, by = 0;
            if (!ctx->opt.wgroups) ng_est = auto_groups(tiles_of(Tx, Ty));  // (the big tile has fewer tiles: at least as many groups, a histogram no larger)
            if ((ctx->opt.bigtile == 1 || sparse) && largest_tile(true, &bx, &by) > best) {
                best = (size_t)bx * by;
                Tx = bx;
                Ty = by;
                bigtile = true;
            }
        }
        if (best == 0) {
            Tx = Ty = 8;
            if (plane_for(8, 8) * 2 > lds_cap)
                return fail(ctx, GRIDHIP_EUNSUPPORTED, "a %lldx%lld kernel does not fit an LDS tile", (long long)gh, (long long)gw);
        }
        // small grids: keep enough tiles to occupy the chip (shrink the longer side, a cell at a time)
        while ((Tx > 16 || Ty > 16) && ((H + Ty - 1) / Ty) * ((Wd + Tx - 1) / Tx) < 1024) {
            if (Ty >= Tx)
                --Ty;
            else
                --Tx;
        }
    }
    if (Tx < 8 || Tx > 128 || Ty < 8 || Ty > 128) return fail(ctx, GRIDHIP_EINVAL, "tile sides must be in 8..128");
    if (plane_for(Tx, Ty) * 2 > lds_cap)
        return fail(ctx, GRIDHIP_EUNSUPPORTED, "tile %dx%d with %lldx%lld kernel needs %zu B of LDS", Tx, Ty,
                    (long long)gh, (long long)gw, plane_for(Tx, Ty) * 2);
    g->Tx = Tx;
    g->Ty = Ty;
    g->imoff = bigtile && plane_for(Tx, Ty) > 65528 ? (int32_t)plane_for(Tx, Ty) : 0;
    g->lcols = Tx + (int)gw - 1;
    g->lrows = Ty + (int)gh - 1;
    g->ldw = lds_pitch(g->lcols, (int)gw);
    g->offx = (((int)gw - 1 + Tx - 1) / Tx) * Tx;
    g->offy = (((int)gh - 1 + Ty - 1) / Ty) * Ty;
    g->ntx = (int)((Wd - 1 + g->offx) / Tx) + 1;
    g->nty = (int)((H - 1 + g->offy) / Ty) + 1;
    if ((int64_t)g->ntx * g->nty > (1 << 24)) return fail(ctx, GRIDHIP_EUNSUPPORTED, "too many tiles");
    g->ntiles = g->ntx * g->nty;

    // w-plane groups: work items of group g run on XCD g (blockIdx round-robin), so that XCD's
    // 4 MiB L2 only has to hold W/8 planes of the kernel table instead of all of them (measured:
    // 1.7x on the 128-plane 15x15 case).  Every (group, tile) pair flushes its tile once, so it
    // only pays when there are enough visibilities per pair to amortise that: the number of groups follows the work per
    // tile, in visibilities of 15 x 15 taps (measured on the 4096^2 / 128-plane shape, tools/r3_wg_sweep.sh and
    // profiles/r03_wgroups_by_size.txt: 1.5 x 10^6 visibilities are fastest with 1 group, 3 x 10^6 with 2, 6 - 25 x 10^6
    // with 4 - by 24 %, 17 %, 5 % over 8 - from 5 x 10^7 on with 8; 7 x 7 kernels need 4.6 times the visibilities)
    int ng = (int)ctx->opt.wgroups;
    if (ng == 0) {
        ng = auto_groups(g->ntiles);
        // very large grids: the pre-pass counts the bins in LDS-histogram windows; each window after the first
        // re-reads the 8-byte pre-records (0.2 ms per 10^8).  Up to four windows 8 groups still win
        // (8192^2, 1.25 x 10^8 visibilities: 21.2 ms against 21.9 with 4 groups, 25.5 with 16)
        const int64_t cap = ((int64_t)ctx->max_lds - 8192) / 4;
        if (ng == 8 && (int64_t)g->ntiles * 8 > 4 * cap) ng = 4;
    }
    if (ng > W) ng = (int)W;
    if (W * ng >= ((int64_t)1 << 31)) ng = 1;  // (the group of a plane is computed in 32 bits)
    if (ng < 1 || ng > 16) return fail(ctx, GRIDHIP_EINVAL, "wgroups must be in 1..16");
    g->ngroups = ng;
    g->nbins = ng * g->ntiles;
    set_div_magic(g);
    // which table of walker weights (tile_sorted.hip): the steeper one where the LDS unit binds - supports 15 and 16 with two
    // or more visibilities per slice and tile; measured per support, option "wtable" forces 1 = flat, 2 = steep
    g->dense = ctx->opt.wtable ? (ctx->opt.wtable >= 3 ? 2 : ctx->opt.wtable == 2 ? 1 : 0)
                               : (gh >= 15 && n >= 2 * (int64_t)g->ntiles * W * Q * Q) ? 1 : (g->imoff > 0 && gh == 15) ? 2 : 0;

    int chunk = (int)ctx->opt.chunk;
    if (chunk == 0) chunk = 8192;
    if (chunk < 64) chunk = 64;
    g->chunk = chunk;
    g->dbg = (int32_t)ctx->opt.dbg;

    *lds_bytes = plane_for(Tx, Ty) * 2;
    int b = (int)ctx->opt.block;
    if (b == 0) {
        // one work-group per CU at T=64 (LDS-limited): use all 16 waves; smaller tiles
        // co-reside, so give each fewer waves.
        size_t per_cu = (size_t)ctx->max_lds / *lds_bytes;
        b = per_cu >= 4 ? 256 : per_cu >= 2 ? 512 : 1024;
    }
    if (b < 64 || b > 1024 || (b & 63)) return fail(ctx, GRIDHIP_EINVAL, "block must be a multiple of 64 in 64..1024");
    *block = b;
    (void)n;
    return GRIDHIP_OK;
}

}  // namespace gridhip

using namespace gridhip;

extern "C" {

int gridhip_version(void) { return GRIDHIP_VERSION; }

const char *gridhip_strerror(int code)
{
    switch (code) {
        case GRIDHIP_OK: return "ok";
        case GRIDHIP_EINVAL: return "invalid argument";
        case GRIDHIP_ENOMEM: return "out of memory";
        case GRIDHIP_EHIP: return "HIP runtime error";
        case GRIDHIP_ENODEV: return "no usable device";
        case GRIDHIP_EUNSUPPORTED: return "unsupported shape";
        default: return "unknown error";
    }
}

int gridhip_device_count(int *count)
{
    if (!count) return GRIDHIP_EINVAL;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return GRIDHIP_ENODEV;
    }
    *count = c;
    return GRIDHIP_OK;
}

int gridhip_create(int device, gridhip_ctx **out)
{
    if (!out) return GRIDHIP_EINVAL;
    *out = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return GRIDHIP_ENODEV;
    if (device < 0 || device >= cnt) return GRIDHIP_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return GRIDHIP_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GRIDHIP_ENODEV;
    gridhip_ctx *ctx = new (std::nothrow) gridhip_ctx();
    if (!ctx) return GRIDHIP_ENOMEM;
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int maxlds = 0;
    if (hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess &&
        maxlds > 0)
        ctx->max_lds = maxlds;
    else
        ctx->max_lds = 64 * 1024;
    // CDNA4: 160 KiB of LDS per CU, all of it available to one work-group
    if (strstr(prop.gcnArchName, "gfx950") && ctx->max_lds < 160 * 1024) ctx->max_lds = 160 * 1024;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return GRIDHIP_EHIP;
    }
    ctx->stream = ctx->own_stream;
    if (hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming) != hipSuccess) {
        gridhip_destroy(ctx);
        return GRIDHIP_EHIP;
    }
    if (hipMalloc((void **)&ctx->d_scalars, 128 * sizeof(int32_t)) != hipSuccess ||
        hipMemset(ctx->d_scalars, 0, 128 * sizeof(int32_t)) != hipSuccess) {
        gridhip_destroy(ctx);
        return GRIDHIP_ENOMEM;
    }
    for (int i = 0; i < gridhip_ctx::EV_RING * 3; ++i)
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) {
            gridhip_destroy(ctx);
            return GRIDHIP_EHIP;
        }
    *out = ctx;
    return GRIDHIP_OK;
}

int gridhip_destroy(gridhip_ctx *ctx)
{
    if (!ctx) return GRIDHIP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    fft_release(ctx);
    Workspace *all[] = {&ctx->recs, &ctx->tables, &ctx->stage, &ctx->blockhist, &ctx->sorted, &ctx->recs_tmp, &ctx->recs_raw, &ctx->ktab, &ctx->aw};
    for (Workspace *w : all)
        if (w->ptr) (void)hipFree(w->ptr);
    for (auto &b : ctx->pool_free) (void)hipFree(b.first);
    ctx->pool_free.clear();
    if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
    for (int i = 0; i < gridhip_ctx::EV_RING * 3; ++i)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->order_ev) (void)hipEventDestroy(ctx->order_ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return GRIDHIP_OK;
}

const char *gridhip_last_error(const gridhip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// Every call on a context shares its scratch (records, tables, sorted lists, padded kernels, aw tables): work
// enqueued on the stream selected next must not start before what the previous stream still has queued is done with
// them.  An event recorded on the old stream, waited for by the new one - unless either is being captured into a
// graph (a capture may not depend on work outside it; the caller keeps one stream per context there).
static int switch_stream(gridhip_ctx *ctx, hipStream_t next)
{
    if (next == ctx->stream) return GRIDHIP_OK;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    hipStreamCaptureStatus a = hipStreamCaptureStatusNone, b = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &a);
    (void)hipStreamIsCapturing(next, &b);
    if (a == hipStreamCaptureStatusNone && b == hipStreamCaptureStatusNone) {
        GH_CHECK_HIP(ctx, hipEventRecord(ctx->order_ev, ctx->stream));
        GH_CHECK_HIP(ctx, hipStreamWaitEvent(next, ctx->order_ev, 0));
    }
    ctx->stream = next;
    return GRIDHIP_OK;
}

int gridhip_set_stream(gridhip_ctx *ctx, void *s)
{
    if (!ctx) return GRIDHIP_EINVAL;
    // NULL is a real stream: HIP's default ("null") stream, which is what torch.cuda.current_stream()
    // is until the caller switches streams.  Work is enqueued exactly where the caller's own work is.
    return switch_stream(ctx, (hipStream_t)s);
}

int gridhip_reset_stream(gridhip_ctx *ctx)
{
    if (!ctx) return GRIDHIP_EINVAL;
    return switch_stream(ctx, ctx->own_stream);
}

void *gridhip_get_stream(gridhip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int gridhip_synchronize(gridhip_ctx *ctx)
{
    if (!ctx) return GRIDHIP_EINVAL;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

static int64_t *opt_slot(gridhip_ctx *ctx, const char *key)
{
    if (!strcmp(key, "tile")) return &ctx->opt.tile;
    if (!strcmp(key, "tile_x")) return &ctx->opt.tile_x;
    if (!strcmp(key, "tile_y")) return &ctx->opt.tile_y;
    if (!strcmp(key, "block")) return &ctx->opt.block;
    if (!strcmp(key, "chunk")) return &ctx->opt.chunk;
    if (!strcmp(key, "wgroups")) return &ctx->opt.wgroups;
    if (!strcmp(key, "variant")) return &ctx->opt.variant;
    if (!strcmp(key, "sort")) return &ctx->opt.sort;
#ifdef GRIDHIP_TUNING
    if (!strcmp(key, "dbg")) return &ctx->opt.dbg;  // ablation / profiling switch: tuning builds only (make tuning)
#endif
    if (!strcmp(key, "prepass")) return &ctx->opt.prepass;
    if (!strcmp(key, "fault_inject")) return &ctx->opt.fault_inject;
    if (!strcmp(key, "aw_cache")) return &ctx->opt.aw_cache;
    if (!strcmp(key, "coarse_shift")) return &ctx->opt.coarse_shift;
    if (!strcmp(key, "scatter_chunk")) return &ctx->opt.scatter_chunk;
    if (!strcmp(key, "count_unroll")) return &ctx->opt.count_unroll;
    if (!strcmp(key, "rec_bits")) return &ctx->opt.rec_bits;
    if (!strcmp(key, "wtable")) return &ctx->opt.wtable;
    if (!strcmp(key, "reserve_cus")) return &ctx->opt.reserve_cus;
    if (!strcmp(key, "subfoot")) return &ctx->opt.subfoot;
    if (!strcmp(key, "bigtile")) return &ctx->opt.bigtile;
    if (!strcmp(key, "yield_cus")) return &ctx->opt.yield_cus;
    return nullptr;
}

int gridhip_set_option(gridhip_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key) return GRIDHIP_EINVAL;
    int64_t *s = opt_slot(ctx, key);
    if (!s) return fail(ctx, GRIDHIP_EINVAL, "unknown option '%s'", key);
    if (value < 0) return fail(ctx, GRIDHIP_EINVAL, "option '%s' must be >= 0", key);
    *s = value;
    return GRIDHIP_OK;
}

int gridhip_get_option(gridhip_ctx *ctx, const char *key, int64_t *value)
{
    if (!ctx || !key || !value) return GRIDHIP_EINVAL;
    if (!strcmp(key, "errors")) {
        // read-only: internal consistency failures counted by the last tile-kernel launch (expected 0)
        GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
        int32_t h = 0;
        GH_CHECK_HIP(ctx, hipMemcpyAsync(&h, ctx->d_scalars + 2, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *value = h;
        return GRIDHIP_OK;
    }
    {
        static const char *const names[4] = {"last_wgroups", "last_tile_x", "last_tile_y", "last_bigtile"};
        for (int i = 0; i < 4; ++i)
            if (!strcmp(key, names[i])) {
                *value = ctx->last_geom[i];
                return GRIDHIP_OK;
            }
    }
    if (!strcmp(key, "last_path")) {
        *value = ctx->last_path;
        return GRIDHIP_OK;
    }
    if (!strcmp(key, "clock_khz") || !strcmp(key, "aw_clock_khz")) {
        // read-only: shader clock held during the last sorted tile kernel (aw_clock_khz: the last aw kernel-build
        // launch), from the s_memtime / s_memrealtime (100 MHz) stamps its first work-group takes when it starts
        // and when its queues are empty
        GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
        int64_t h[4] = {0, 0, 0, 0};
        GH_CHECK_HIP(ctx, hipMemcpyAsync(h, ctx->d_scalars + (key[0] == 'a' ? 96 : 20), sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int64_t cyc = h[2] - h[0], ticks = h[3] - h[1];
        *value = ticks > 0 && cyc > 0 ? (int64_t)((double)cyc / (double)ticks * 1e5) : 0;
        return GRIDHIP_OK;
    }
    if (!strncmp(key, "prof", 4) && key[4] >= '0' && key[4] <= '9') {
        // read-only: cycles summed over work-groups by a dbg=16 tuning launch of the sorted kernel
        // (0..6: per phase, thread 0; 8..23: the accumulate walk of wave 0..15)
        const int slot = atoi(key + 4);
        if (slot < 0 || slot >= 32) return fail(ctx, GRIDHIP_EINVAL, "unknown option '%s'", key);
        GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
        int64_t h = 0;
        GH_CHECK_HIP(ctx, hipMemcpyAsync(&h, ctx->d_scalars + 32 + 2 * slot, sizeof h, hipMemcpyDeviceToHost,
                                         ctx->stream));
        GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *value = h;
        return GRIDHIP_OK;
    }
    int64_t *s = opt_slot(ctx, key);
    if (!s) return fail(ctx, GRIDHIP_EINVAL, "unknown option '%s'", key);
    *value = *s;
    return GRIDHIP_OK;
}

int gridhip_malloc(gridhip_ctx *ctx, void **dptr, int64_t bytes)
{
    if (!ctx || !dptr || bytes < 0) return GRIDHIP_EINVAL;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    *dptr = nullptr;
    if (bytes == 0) return GRIDHIP_OK;
    GH_CHECK_HIP(ctx, hipMalloc(dptr, (size_t)bytes));
    return GRIDHIP_OK;
}

int gridhip_free(gridhip_ctx *ctx, void *dptr)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (!dptr) return GRIDHIP_OK;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GH_CHECK_HIP(ctx, hipFree(dptr));
    return GRIDHIP_OK;
}

int gridhip_memcpy_h2d(gridhip_ctx *ctx, void *dst, const void *src, int64_t bytes)
{
    if (!ctx || bytes < 0 || (bytes && (!dst || !src))) return GRIDHIP_EINVAL;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

int gridhip_memcpy_d2h(gridhip_ctx *ctx, void *dst, const void *src, int64_t bytes)
{
    if (!ctx || bytes < 0 || (bytes && (!dst || !src))) return GRIDHIP_EINVAL;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

int gridhip_memset(gridhip_ctx *ctx, void *dptr, int value, int64_t bytes)
{
    if (!ctx || bytes < 0 || (bytes && !dptr)) return GRIDHIP_EINVAL;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    GH_CHECK_HIP(ctx, hipMemsetAsync(dptr, value, (size_t)bytes, ctx->stream));
    return GRIDHIP_OK;
}

int gridhip_enable_timing(gridhip_ctx *ctx, int enable)
{
    if (!ctx) return GRIDHIP_EINVAL;
    ctx->timing = enable != 0;
    ctx->ev_calls = 0;
    ctx->ev_open = false;
    return GRIDHIP_OK;
}

int gridhip_timing(gridhip_ctx *ctx, int back, double *ms_total, double *ms_prepass, double *ms_kernel)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (back < 0 || back >= gridhip_ctx::EV_RING || back >= ctx->ev_calls)
        return fail(ctx, GRIDHIP_EINVAL, "no timed call recorded %d calls back (gridhip_enable_timing; the last %d are kept)",
                    back, gridhip_ctx::EV_RING);
    hipEvent_t *e = ctx->ev + ((ctx->ev_calls - 1 - back) % gridhip_ctx::EV_RING) * 3;
    GH_CHECK_HIP(ctx, hipEventSynchronize(e[2]));
    float pre = 0.f, ker = 0.f;
    GH_CHECK_HIP(ctx, hipEventElapsedTime(&pre, e[0], e[1]));
    GH_CHECK_HIP(ctx, hipEventElapsedTime(&ker, e[1], e[2]));
    if (ms_prepass) *ms_prepass = pre;
    if (ms_kernel) *ms_kernel = ker;
    if (ms_total) *ms_total = (double)pre + (double)ker;
    return GRIDHIP_OK;
}

int gridhip_last_timing(gridhip_ctx *ctx, double *ms_total, double *ms_prepass, double *ms_kernel)
{
    return gridhip_timing(ctx, 0, ms_total, ms_prepass, ms_kernel);
}

int gridhip_last_dropped(gridhip_ctx *ctx, int64_t *dropped)
{
    if (!ctx || !dropped) return GRIDHIP_EINVAL;
    *dropped = 0;
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int32_t h = 0;
    GH_CHECK_HIP(ctx, hipMemcpyAsync(&h, ctx->d_scalars, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *dropped = h;
    return GRIDHIP_OK;
}

}  // extern "C"
