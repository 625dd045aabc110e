// Plans: bin a set of baselines once, grid / degrid against it many times.
//
// The binning pre-pass depends only on the coordinates (u, v, wbin) and the kernel-table shape,
// never on the visibility values or the kernel values.  do_imaging grids the same baselines twice
// (image and PSF, src/Gridding.hs:538,541) and imaging major cycles alternate degrid / grid over
// the same baselines many times; a plan keeps the tile-ordered records resident in HBM so each
// further pass costs the tile kernel only (17 ms instead of 21 ms on the 10^8-visibility case).
#include <utility>

#include "common.h"

struct gridhip_plan {
    gridhip_ctx *ctx = nullptr;
    gridhip::Prep p;
    gridhip::Workspace recs, tables;
    int64_t n = 0;
    bool all_binned = false;  // no visibility was dropped: degrid writes every element of its output
    // the records live in the context's own scratch instead of the plan's (imaging.hip: a plan that lasts for the two
    // passes of one call - no hipMalloc, no hipFree with its device-wide synchronisation); valid until the context's
    // next gridding call that is not this plan's
    bool borrowed = false;
};

using namespace gridhip;

namespace {
// the launchers read the binned data from the context's scratch slots: lend them the plan's
struct Lend {
    gridhip_plan *p;
    explicit Lend(gridhip_plan *pl) : p(pl)
    {
        if (p->borrowed) return;
        std::swap(p->ctx->recs, p->recs);
        std::swap(p->ctx->tables, p->tables);
    }
    ~Lend()
    {
        if (p->borrowed) return;
        std::swap(p->ctx->recs, p->recs);
        std::swap(p->ctx->tables, p->tables);
    }
};
}  // namespace

static int plan_create(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t n, int64_t W, int64_t Q, int64_t gh, int64_t gw,
                       const double *u, const double *v, int64_t uv_stride, const int64_t *wbin, gridhip_plan **out,
                       bool borrowed);

namespace gridhip {
int plan_create_borrowed(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t n, int64_t W, int64_t Q, int64_t gh, int64_t gw,
                         const double *u, const double *v, int64_t uv_stride, const int64_t *wbin, gridhip_plan **out)
{
    return plan_create(ctx, H, Wd, n, W, Q, gh, gw, u, v, uv_stride, wbin, out, true);
}
}  // namespace gridhip

extern "C" {

int gridhip_plan_create_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t n, int64_t W, int64_t Q, int64_t gh,
                            int64_t gw, const double *u, const double *v, int64_t uv_stride, const int64_t *wbin,
                            gridhip_plan **out)
{
    return plan_create(ctx, H, Wd, n, W, Q, gh, gw, u, v, uv_stride, wbin, out, false);
}

}  // extern "C"

static int plan_create(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t n, int64_t W, int64_t Q, int64_t gh, int64_t gw,
                       const double *u, const double *v, int64_t uv_stride, const int64_t *wbin, gridhip_plan **out,
                       bool borrowed)
{
    if (!ctx || !out) return GRIDHIP_EINVAL;
    *out = nullptr;
    if (H <= 0 || Wd <= 0 || n < 0 || uv_stride < 1 || W <= 0 || Q <= 0 || gh <= 0 || gw <= 0 ||
        (n > 0 && (!u || !v)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    if (n > (int64_t)0x7fffff00) return fail(ctx, GRIDHIP_EUNSUPPORTED, "n must be < 2^31 per plan");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    gridhip_plan *p = new (std::nothrow) gridhip_plan();
    if (!p) return GRIDHIP_ENOMEM;
    p->ctx = ctx;
    p->n = n;
    p->borrowed = borrowed;
    int rc = prepare(ctx, H, Wd, W, Q, gh, gw, n, &p->p);
    if (rc == GRIDHIP_OK && p->p.direct) rc = fail(ctx, GRIDHIP_EUNSUPPORTED, "plan: support too large for an LDS tile");
    if (rc == GRIDHIP_OK && !rec_fits(p->p.g)) rc = fail(ctx, GRIDHIP_EUNSUPPORTED, "plan: slices x visibilities above 2^50");
    if (rc != GRIDHIP_OK) {
        delete p;
        return rc;
    }
    {
        Lend lend(p);
        rc = ws_reserve(ctx, ctx->tables, tables_bytes(p->p.g));
        if (rc == GRIDHIP_OK) rc = ws_reserve(ctx, ctx->recs, (size_t)(p->p.nrec > 0 ? p->p.nrec : 1) * sizeof(RecWord));
        if (rc == GRIDHIP_OK) rc = launch_bin(ctx, p->p.g, p->p.nrec, u, v, uv_stride, wbin);
        if (rc == GRIDHIP_OK && p->p.g.P == 1 && !(p->p.sorted && p->p.g.gh > 16) && n > 0) {
            // how many visibilities found a bin: when all did, degrid passes skip clearing their output
            int32_t binned = 0;
            const Tables t = tables_of(ctx, p->p.g);
            if (hipMemcpyAsync(&binned, t.bin_start + p->p.g.nbins, sizeof(binned), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                hipStreamSynchronize(ctx->stream) == hipSuccess)
                p->all_binned = binned == (int32_t)n && ctx->opt.fault_inject == 0;
        }
    }
    if (rc != GRIDHIP_OK) {
        gridhip_plan_destroy(p);
        return rc;
    }
    *out = p;
    return GRIDHIP_OK;
}

extern "C" {

int gridhip_plan_destroy(gridhip_plan *p)
{
    if (!p) return GRIDHIP_OK;
    if (p->borrowed) {  // (nothing of its own on the device)
        delete p;
        return GRIDHIP_OK;
    }
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    if (p->recs.ptr) (void)hipFree(p->recs.ptr);
    if (p->tables.ptr) (void)hipFree(p->tables.ptr);
    delete p;
    return GRIDHIP_OK;
}

// G += scatter(vis x kernels) over the plan's baselines (convgrid2 semantics); asynchronous on the
// context's stream, like the other _dev entry points.
int gridhip_plan_grid_dev(gridhip_plan *p, const double *gcf, const double *vis, double *grid)
{
    if (!p) return GRIDHIP_EINVAL;
    gridhip_ctx *ctx = p->ctx;
    if (!gcf || !grid || (p->n > 0 && !vis)) return fail(ctx, GRIDHIP_EINVAL, "null pointer");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (p->n == 0) return GRIDHIP_OK;
    const double *tk = gcf;
    GH_CHECK(tile_kernels(ctx, p->p, gcf, &tk));
    Lend lend(p);
    const Prep &q = p->p;
    if (q.sorted)
        return launch_tile_grid_sorted(ctx, q.g, q.block, q.lds_sorted, q.nkeys, q.batch, q.nrec, tk, vis, grid, false);
    return launch_tile_grid(ctx, q.g, q.block, q.lds, q.nrec, tk, vis, grid);
}

// vis_out[k] = gather(kernels x G) for every baseline of the plan (degrid2 semantics)
int gridhip_plan_degrid_dev(gridhip_plan *p, const double *gcf, const double *grid, double *vis_out)
{
    if (!p) return GRIDHIP_EINVAL;
    gridhip_ctx *ctx = p->ctx;
    if (!gcf || !grid || (p->n > 0 && !vis_out)) return fail(ctx, GRIDHIP_EINVAL, "null pointer");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (p->n == 0) return GRIDHIP_OK;
    if (!p->all_binned) GH_CHECK_HIP(ctx, hipMemsetAsync(vis_out, 0, (size_t)p->n * 16, ctx->stream));
    const double *tk = gcf;
    GH_CHECK(tile_kernels(ctx, p->p, gcf, &tk));
    Lend lend(p);
    const Prep &q = p->p;
    if (q.sorted)
        return launch_tile_grid_sorted(ctx, q.g, q.block, q.lds_sorted, q.nkeys, q.batch, q.nrec, tk, vis_out,
                                       const_cast<double *>(grid), true);
    return launch_tile_degrid(ctx, q.g, q.block, q.lds, q.nrec, tk, grid, vis_out);
}

}  // extern "C"
