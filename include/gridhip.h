/*
 * gridhip.h — C ABI of libgridhip.so: MI355X (gfx950 / CDNA4) native convolutional
 * w-projection gridder / degridder.
 *
 * This is the drop-in boundary for the gridding hot path of
 * sakehl/SKA-SDP-Accelerate-gridding.  Each entry point replaces the *body* of one
 * Accelerate function of /root/reference/src/Gridding.hs (cited per function); the Haskell
 * signatures stay, the `run` over the array program becomes a `foreign import ccall` into
 * this library (binding shown in INTEGRATION.md).
 *
 * Conventions (all follow the reference's own FFI style, hdf5/hdf5.cc:59-186 and
 * src/Hdf5.hs:30-67: extern "C", plain pointers + integers, caller allocates outputs):
 *   F = double, Int = int64_t, Antenna = int64_t            (src/Types.hs:7-16)
 *   Complex Double arrays = interleaved (re,im) doubles     (src/Hdf5.hs:113-137, hdf5/hdf5.cc:14-17)
 *   Vector (F,F,F) = separate u / v / w pointers; `uv_stride` is the element stride between
 *       consecutive visibilities (1 for struct-of-arrays as Accelerate stores tuples, 3 when u
 *       and v point into the (n,3) row-major /vis/uvw matrix, src/ImageDataset.hs:94-97)
 *   grids are row-major [y][x] (y <-> v axis), H rows x Wd columns (src/Gridding.hs:106-109)
 *   gcf   = [W][Q][Q][gh][gw] complex, index order (wbin, yf, xf, i, j) (src/Gridding.hs:243)
 *   grids are ACCUMULATED INTO (permute (+) a ..., src/Gridding.hs:99,197,244), never overwritten
 *   the library never retains or frees a caller pointer past the call
 *
 * Every function returns GRIDHIP_OK (0) or a negative GRIDHIP_E* code; the message for the
 * last failure on a context is available from gridhip_last_error().  (The reference's FFI
 * reports nothing; this is the one deliberate departure)
 *
 * Two flavours per operation:
 *   gridhip_<op>      host pointers, synchronous — the drop-in form;
 *   gridhip_<op>_dev  device pointers, asynchronous on the context's stream — what the
 *                     benchmark and multi-GPU drivers use so that H2D is outside the timing.
 *                     `grid` and `vis_out` must be ordinary device allocations (hipMalloc: coarse-grained
 *                     memory): the tile kernels flush with hardware fp64 atomics (global_atomic_add_f64),
 *                     which fine-grained / host-coherent mappings do not support.
 * A context is bound to one device and one stream and is not thread-safe; contexts are independent of each other (one
 * per host thread: tests/test_gpu_limits.py runs two on one GPU concurrently).
 */
#ifndef GRIDHIP_H
#define GRIDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRIDHIP_VERSION 110 /* 0.1.1 */

#define GRIDHIP_OK 0
#define GRIDHIP_EINVAL (-1)       /* bad argument (null pointer, negative size, ...) */
#define GRIDHIP_ENOMEM (-2)       /* device or host allocation failed */
#define GRIDHIP_EHIP (-3)         /* HIP runtime / library error, see gridhip_last_error */
#define GRIDHIP_ENODEV (-4)       /* no usable gfx950 device */
#define GRIDHIP_EUNSUPPORTED (-5) /* shape outside what the kernels support */

typedef struct gridhip_ctx gridhip_ctx;

/* ---- context --------------------------------------------------------------------------- */
int gridhip_version(void);
const char *gridhip_strerror(int code);
int gridhip_device_count(int *count);
/* Create a context on HIP device `device` with its own non-blocking stream. */
int gridhip_create(int device, gridhip_ctx **ctx);
int gridhip_destroy(gridhip_ctx *ctx);
const char *gridhip_last_error(const gridhip_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream).  NULL means HIP's default
 * (null) stream, NOT "no stream": device-pointer calls must be ordered with the caller's own
 * kernels and copies, and those run on the null stream unless the caller created another.
 * gridhip_reset_stream() goes back to the context's private non-blocking stream (which does not
 * synchronise with the null stream: only use it when the inputs are known to be complete). */
/* A context has ONE set of scratch buffers (records, tables, sorted lists, padded kernels): when the stream changes,
 * the new stream is made to wait (event) for what was enqueued on the old one, so a caller that alternates streams
 * between calls cannot have one call overwrite scratch another is still reading.  (Not while either stream is being
 * captured into a graph: a capture must not depend on work outside it - use one stream per context there.) */
int gridhip_set_stream(gridhip_ctx *ctx, void *hip_stream);
int gridhip_reset_stream(gridhip_ctx *ctx);
void *gridhip_get_stream(gridhip_ctx *ctx);
int gridhip_synchronize(gridhip_ctx *ctx);
/* Tuning knobs (all have defaults chosen per shape):
 *   "tile"      side of a square grid tile in cells (8..128; 0 = auto: the largest rectangle whose planes fit the
 *               LDS layout, 65 x 89 cells for a 15 x 15 kernel); "tile_x" / "tile_y": a rectangular tile
 *   "block"     threads per work-group of the tile kernels (multiple of 64, <=1024; 0 = auto)
 *   "chunk"     max visibilities per work item (0 = auto; the sorted kernel takes at most 16384)
 *   "wgroups"   number of w-plane groups work items are split into for XCD/L2 locality (1..16; 0 = auto)
 *   "variant"   0 = LDS-tile accumulate (default), 1 = direct global-atomic scatter (baseline)
 *   "sort"      order each work item's records by kernel slice so that runs of visibilities
 *               reuse their taps from registers: 0 = auto, 1 = on (when the shape allows), 2 = off
 *   "prepass"   scatter of the binning pre-pass: 0 = auto (two levels from 2^22 visibilities), 1 = one level,
 *               2 = two levels (the counting sweep leaves 8-byte pre-records, which are LDS-sorted into runs per
 *               coarse bin and then per bin), 3 = one level with global atomics only (no LDS; a measured
 *               baseline), 4 = two levels recomputing from the stream instead of reading pre-records,
 *               5, 6 = two levels with 16-byte / 12-byte instead of 8-byte intermediate records
 *   "coarse_shift", "scatter_chunk", "count_unroll"  details of that scatter kept for comparison runs: bins per
 *               coarse bin = 2^coarse_shift (0 = balanced); scatter_chunk = 4096: half-size chunks, two work-groups
 *               per CU; count_unroll = 4: four visibilities per thread and trip in the counting sweep, 1: one, and
 *               no 16-byte grid-stride form either (0 = auto)
 *   "aw_cache"  aw gridders: 1 (default) = build each distinct (a1, a2, wbin, yf, xf) kernel once per call and let
 *               the visibilities that share it reuse it; 0 = one kernel per visibility (as the reference evaluates)
 *   "fault_inject"  TEST HOOK: hides the last k slots of the record array from the pre-pass's scatter so that its
 *               bounds checks have something to reject (counted in "errors"; results are then incomplete)
 *   "reserve_cus"  compute units the persistent tile kernel leaves free (0 = none): it launches one work-group per
 *               remaining CU, so that a collective queued on another stream (RCCL's kernel, a copy) finds CUs to start
 *               on while the tile kernel runs instead of waiting ~10 ms for it to end; costs the tile kernel
 *               k / num_cu of its throughput (profiles/r03_reserve_cus.txt)
 *   "yield_cus"  the cheaper way to the same end (0 = off; ignored while "reserve_cus" is set): k CUs' worth of the tile
 *               kernel's work-groups are not persistent - each takes eight work items and leaves, and up to 2 048
 *               further ones are launched to follow them - so a kernel queued on another stream gets a CU within a few
 *               hundred microseconds while nothing idles when none is queued (+0.5 % on the tile kernel instead of
 *               +10 %).  k is rounded up to a multiple of 32 (one CU per shader engine of every XCD): with fewer the
 *               dispatcher's rotation over the shader engines stops at one without a free CU and the CUs given up stay
 *               idle (profiles/r03_yield_cus.txt)
 *   "bigtile"   the tap-reusing kernel's tile uses all of the LDS (65 x 110 cells at 15 x 15 instead of 65 x 89): a
 *               quarter more visibilities per kernel slice and work item, for an address add per tap step; pays
 *               where items are sparse (fewer than two visibilities per slice and item), not where the LDS atomic
 *               unit binds.  0 = auto (sparse streams of at least 2^22 visibilities), 1 = on, 2 = off
 *   "subfoot"   1 = supports 17..32 take the sub-footprint path (one record per spatial part of the kernel, as
 *               supports above 32 and non-square kernels always do) instead of the tap-reusing kernel's own parts of
 *               the tap list (one record per visibility): kept for comparison runs
 *   "wtable"    which table of walker weights the tap-reusing kernel uses: 0 = auto, 1 = flat, 2 = steep (tile_sorted.hip)
 *   "rec_bits"  TEST HOOK: pretend the 64-bit record word has this many bits (16..63), so that small calls take the
 *               path that grids a call in several parts (taken for real above 2^50 slices x visibilities);
 *               100 + t: widen the record's kernel-slice field until its fields take t <= 64 bits
 *   ("dbg", the ablation / profiling switch of tuning runs, exists only in the tuning build of the library,
 *   `make -C csrc tuning` -> lib/libgridhip_tuning.so; the shipped library rejects the key)
 * Read-only (gridhip_get_option): "last_wgroups", "last_tile_x", "last_tile_y", "last_bigtile" = the geometry the last
 * convgrid / convgrid2 / degrid2 call chose (w-groups, the LDS tile's interior, whether the tile uses all of the LDS);
 * "last_path" = which gridder the last convgrid / convgrid2 / degrid2 / plan call used:
 * 1 = the tap-reusing tile kernel (square supports 5..32 with enough visibilities per work item), 2 = the same through
 * sub-footprints (other shapes: one record per spatial part of the kernel), 3 = the general tile kernel (small
 * problems, and the sizes listed under "Limits" below: 2 - 3 x slower per visibility at scale), 4 = direct
 * global-atomic scatter; "errors" = internal consistency failures counted by the last tile-kernel
 * launch (expected 0); "clock_khz" = shader clock held during the last tap-reusing tile kernel (in-kernel
 * s_memtime / s_memrealtime stamps), "aw_clock_khz" = the same for the last launch of the aw gridders' kernel builder; "prof0".."prof31" = cycle counters of a dbg=16 launch of the tuning build (tools/phase_profile.py).
 */
int gridhip_set_option(gridhip_ctx *ctx, const char *key, int64_t value);
int gridhip_get_option(gridhip_ctx *ctx, const char *key, int64_t *value);
/* Visibilities skipped by the last gridding call because `wbin` was outside [0,W) (the
 * reference would read the kernel out of range).  Synchronises the stream. */
int gridhip_last_dropped(gridhip_ctx *ctx, int64_t *dropped);

/* ---- Limits ------------------------------------------------------------------------------------
 * Per call: n < 2^31 - 256 visibilities (GRIDHIP_EUNSUPPORTED above; cut the stream), H, Wd <= 2^30, W * Q * Q <= 2^30,
 * gh, gw <= 1024.  A kernel that is not a square of side 5..32 is gridded as P = py * px sub-footprints, one binned
 * record per part, which needs n * P < 2^31 - 256 and W * Q * Q * P < 2^30: a call beyond either takes the general
 * tile kernel instead (read-only option "last_path" = 3 says so; nothing is lost but speed) - cut the stream into
 * calls of fewer visibilities to stay on the fast path.  Slices x visibilities above 2^50 are gridded in several
 * parts internally.  degrid2 has no direct form: a support too large for an LDS tile is GRIDHIP_EUNSUPPORTED. */

/* ---- gridders: host pointers (drop-in) -------------------------------------------------- */

/* grid  — src/Gridding.hs:95-112.   G[N/2+floor(.5+N*v), N/2+floor(.5+N*u)] += vis
 * (N = H as in the reference, :101-103; cells outside the grid are dropped). */
int gridhip_grid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n,
                 const double *u, const double *v, int64_t uv_stride, const double *vis);

/* convgrid — src/Gridding.hs:153-197.  gcf is [Q][Q][gh][gw]. */
int gridhip_convgrid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n,
                     int64_t Q, int64_t gh, int64_t gw, const double *gcf, const double *u,
                     const double *v, int64_t uv_stride, const double *vis);

/* convgrid2 — src/Gridding.hs:199-244 (the w-projection kernel).  gcf is [W][Q][Q][gh][gw]. */
int gridhip_convgrid2(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n,
                      int64_t W, int64_t Q, int64_t gh, int64_t gw, const double *gcf,
                      const double *u, const double *v, int64_t uv_stride,
                      const int64_t *wbin, const double *vis);

/* degrid2 — the gather with convgrid2's coordinates (north_star "degrid"; absent from the
 * reference, defined in SURVEY.md §8a): vis_out[k] = sum_ij gcf[wbin,yf,xf,i,j]*G[y0+i,x0+j]. */
int gridhip_degrid2(gridhip_ctx *ctx, int64_t H, int64_t Wd, const double *grid, int64_t n,
                    int64_t W, int64_t Q, int64_t gh, int64_t gw, const double *gcf,
                    const double *u, const double *v, int64_t uv_stride,
                    const int64_t *wbin, double *vis_out);

/* ---- gridders: device pointers, asynchronous on the context's stream --------------------- */
int gridhip_grid_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n,
                     const double *u, const double *v, int64_t uv_stride, const double *vis);
int gridhip_convgrid_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n,
                         int64_t Q, int64_t gh, int64_t gw, const double *gcf,
                         const double *u, const double *v, int64_t uv_stride,
                         const double *vis);
int gridhip_convgrid2_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n,
                          int64_t W, int64_t Q, int64_t gh, int64_t gw, const double *gcf,
                          const double *u, const double *v, int64_t uv_stride,
                          const int64_t *wbin, const double *vis);
int gridhip_degrid2_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, const double *grid,
                        int64_t n, int64_t W, int64_t Q, int64_t gh, int64_t gw,
                        const double *gcf, const double *u, const double *v,
                        int64_t uv_stride, const int64_t *wbin, double *vis_out);

/* ---- plans: bin the baselines once, grid / degrid many times (device pointers) ----------------
 * The binning pre-pass depends on (u, v, wbin) and the kernel-table SHAPE only.  do_imaging grids
 * the same baselines twice (image and PSF, src/Gridding.hs:538,541) and major cycles alternate
 * degrid / grid over them; a plan keeps the tile-ordered records resident so each further pass is
 * the tile kernel alone.  The coordinate arrays may be released after gridhip_plan_create_dev
 * returns and the stream has run; vis / grid / gcf are per call.  A plan belongs to its context
 * (same device, same stream, not thread-safe) and must be destroyed before it. */
typedef struct gridhip_plan gridhip_plan;
int gridhip_plan_create_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t n, int64_t W, int64_t Q,
                            int64_t gh, int64_t gw, const double *u, const double *v,
                            int64_t uv_stride, const int64_t *wbin, gridhip_plan **plan);
int gridhip_plan_grid_dev(gridhip_plan *plan, const double *gcf, const double *vis, double *grid);
int gridhip_plan_degrid_dev(gridhip_plan *plan, const double *gcf, const double *grid,
                            double *vis_out);
int gridhip_plan_destroy(gridhip_plan *plan);

/* ---- callers either side of the gridder (host pointers) -------------------------
 * All follow src/Gridding.hs; uvw are in wavelengths where the reference takes them so. */

/* N = round (theta * lam) as the imaging functions compute it (:87,:118,:416; Prelude round). */
int64_t gridhip_image_size(double theta, int64_t lam);
/* w-bin rule of w_cache_imaging, :426-432: wbin = (wstep*round(w/wstep) - min) div wstep. */
int gridhip_wbins(gridhip_ctx *ctx, int64_t n, const double *w, int64_t wstep, int64_t *wbin,
                  int64_t *wmin, int64_t *nplanes);
/* findClosest, :895-907, for each w (out of range reads clamped as ImageDataset.hs:150-168). */
int gridhip_find_closest(gridhip_ctx *ctx, int64_t nws, const double *ws, int64_t n,
                         const double *w, int64_t *out);
/* mirror_uvw, :551-562, in place (w and vis may be NULL). */
int gridhip_mirror_uvw(gridhip_ctx *ctx, int64_t n, double *u, double *v, double *w, double *vis);
/* doweight, :564-583: vis /= number of visibilities in its grid cell; u, v in wavelengths. */
int gridhip_doweight(gridhip_ctx *ctx, double theta, int64_t lam, int64_t n, const double *u,
                     const double *v, double *vis);
/* make_grid_hermitian, :585-605, in place. */
int gridhip_make_grid_hermitian(gridhip_ctx *ctx, int64_t N, double *grid);
/* fft / ifft, :815-829: shift2D . fft2D . ishift2D on an N x N complex array (hipFFT);
 * inverse != 0 is scaled by 1/N^2 as accelerate-fft's Inverse mode. */
int gridhip_fft2_centered(gridhip_ctx *ctx, int64_t N, const double *in, double *out, int inverse);
/* w_kernel, :610-728: out is [qpx][qpx][npixKern][npixKern]. */
int gridhip_w_kernel(gridhip_ctx *ctx, double theta, double w, int64_t npixFF, int64_t npixKern,
                     int64_t qpx, double *out);
/* ImagingFunctions (:76-81): grid is N x N with N = gridhip_image_size(theta, lam), overwritten. */
int gridhip_simple_imaging(gridhip_ctx *ctx, double theta, int64_t lam, int64_t n, const double *u,
                           const double *v, int64_t uv_stride, const double *vis, double *grid);
int gridhip_conv_imaging(gridhip_ctx *ctx, int64_t Q, int64_t gh, int64_t gw, const double *kv,
                         double theta, int64_t lam, int64_t n, const double *u, const double *v,
                         int64_t uv_stride, const double *vis, double *grid);
/* w_cache_imaging, :399-449: builds one conjugated w_kernel per plane, then convgrid2. */
int gridhip_w_cache_imaging(gridhip_ctx *ctx, int64_t wstep, int64_t qpx, int64_t npixFF,
                            int64_t npixKern, double theta, int64_t lam, int64_t n, const double *u,
                            const double *v, const double *w, int64_t uv_stride, const double *vis,
                            double *grid);
/* convgrid3 / convgrid4, :246-396 (both produce this grid): per visibility
 * awkern = conj(aw_kernel_fn2 yf xf wkerns[wbin] akerns[a1] akerns[a2]) (:761-775, convolve2d :795-811 with
 * its transposing pad), G[y0+i,x0+j] += vis*awkern[i,j].  wkerns [W][Q][Q][S][S], akerns [A][S][S];
 * the index triple (wbin, a1, a2) is passed as three arrays (Accelerate's struct-of-arrays).
 * The kernel of every distinct (a1, a2, wbin, yf, xf) is built once per call (option "aw_cache"). */
int gridhip_awgrid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W,
                   int64_t Q, int64_t S, int64_t A, const double *wkerns, const double *akerns,
                   const double *u, const double *v, int64_t uv_stride, const int64_t *wbin,
                   const int64_t *a1, const int64_t *a2, const double *vis);
int gridhip_awgrid_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W,
                       int64_t Q, int64_t S, int64_t A, const double *wkerns, const double *akerns,
                       const double *u, const double *v, int64_t uv_stride, const int64_t *wbin,
                       const int64_t *a1, const int64_t *a2, const double *vis);
/* What the last gridhip_awgrid / gridhip_awgrid_dev call on this context did (synchronises): visibilities that
 * received a kernel, and distinct (a1, a2, wbin, yf, xf) kernels built for them (equal with "aw_cache" = 0). */
int gridhip_aw_last_stats(gridhip_ctx *ctx, int64_t *vis_keyed, int64_t *kernels_built);
/* aw_imaging / aw_imagingOld, :452-506: wvals are the W plane w-values searched by findClosest. */
int gridhip_aw_imaging(gridhip_ctx *ctx, double theta, int64_t lam, int64_t W, int64_t Q, int64_t S,
                       int64_t A, const double *wkerns, const double *wvals, const double *akerns,
                       int64_t n, const double *u, const double *v, const double *w,
                       int64_t uv_stride, const int64_t *a1, const int64_t *a2, const double *vis,
                       double *grid);
/* do_imaging, :509-549: mirror -> weight -> grid(vis*wt), grid(wt) -> hermitian -> ifft -> /max(psf).
 * kind: 0 simple_imaging; 1 conv_imaging (Q, gh, gw, kv); 2 w_cache_imaging (wstep, Q, npixFF, gh = npixKern).
 * image and psf are N x N doubles. */
int gridhip_do_imaging(gridhip_ctx *ctx, int kind, int64_t wstep, int64_t Q, int64_t npixFF,
                       int64_t gh, int64_t gw, const double *kv, double theta, int64_t lam, int64_t n,
                       const double *u, const double *v, const double *w, int64_t uv_stride,
                       const double *vis, double *image, double *psf, double *pmax);
/* The same with every array argument resident on the device (kv, u, v, w, vis in; image, psf out: ordinary device
 * allocations): nothing crosses PCIe, and after the first call of a shape nothing is allocated or freed (the scratch
 * comes from a pool the context keeps).  pmax stays a HOST pointer (may be NULL).  The call synchronises the stream
 * (the w-bin rule reads min / max back to the host exactly as the reference's nested CPU.run does, :430). */
int gridhip_do_imaging_dev(gridhip_ctx *ctx, int kind, int64_t wstep, int64_t Q, int64_t npixFF, int64_t gh,
                           int64_t gw, const double *kv, double theta, int64_t lam, int64_t n, const double *u,
                           const double *v, const double *w, int64_t uv_stride, const double *vis, double *image,
                           double *psf, double *pmax);
/* w_cache_imaging with device-resident u, v, w (wavelengths), vis and N x N grid (overwritten). */
int gridhip_w_cache_imaging_dev(gridhip_ctx *ctx, int64_t wstep, int64_t qpx, int64_t npixFF, int64_t npixKern,
                                double theta, int64_t lam, int64_t n, const double *u, const double *v,
                                const double *w, int64_t uv_stride, const double *vis, double *grid);

/* ---- multi-GPU: visibility-sharded gridding + one RCCL fp64 sum all-reduce of the partial grids ------
 * Gridding is linear in the visibility set, so the path shards by visibility with no data-path exchange; the
 * partial N x N grids are summed with ncclAllReduce(ncclDouble, ncclSum) over xGMI.  The
 * reference has no counterpart (single device: app/Main.hs:46-53 picks one (run, runN) pair); these entry points
 * are what its `aw_gridding` / `do_imaging` callers (src/ImageDataset.hs:72-77, src/Gridding.hs:538-541) would
 * bind to use a whole node.  RCCL (librccl.so.1) is loaded on first use.
 *   gridhip_comm_create       ONE process drives ndev devices (ncclCommInitAll); dev_ids NULL = 0..ndev-1.
 *                             The communicator owns one context per device (gridhip_comm_ctx).
 *   gridhip_comm_create_rank  one process per GPU (ncclCommInitRank): rank 0 obtains a 128-byte id with
 *                             gridhip_comm_unique_id and hands it to the other ranks by its own means;
 *                             `ctx` stays the caller's.
 * Failures are described by gridhip_comm_last_error (NULL: the last failure before a communicator existed). */
typedef struct gridhip_comm gridhip_comm;
int gridhip_comm_create(int ndev, const int *dev_ids, gridhip_comm **comm);
int gridhip_comm_unique_id(void *id128);
int gridhip_comm_create_rank(gridhip_ctx *ctx, int nranks, int rank, const void *id128, gridhip_comm **comm);
int gridhip_comm_destroy(gridhip_comm *comm);
const char *gridhip_comm_last_error(const gridhip_comm *comm);
int gridhip_comm_ndev(const gridhip_comm *comm);   /* devices driven by this process */
int gridhip_comm_nranks(const gridhip_comm *comm); /* devices in the communicator */
gridhip_ctx *gridhip_comm_ctx(gridhip_comm *comm, int i);
/* In-place sum over all devices of the communicator of grids[i] (device pointer on this process's i-th device,
 * `cells` complex cells each); enqueued on each context's stream, asynchronous to the host. */
int gridhip_comm_allreduce_grids(gridhip_comm *comm, int64_t cells, double *const *grids);
int gridhip_comm_allreduce_grid(gridhip_comm *comm, int64_t cells, double *grid); /* rank form */
/* The same for rows [y0, y1) of grids of Wd columns only.  A stream that went through mirror_uvw (src/Gridding.hs:551-562:
 * v >= 0) leaves every row below H/2 - gh/2 - 1 of every partial grid exactly zero; reducing from that row on halves the
 * bytes that cross xGMI.  (Rows outside the range keep each device's own partial content.) */
int gridhip_comm_allreduce_rows(gridhip_comm *comm, int64_t Wd, int64_t y0, int64_t y1, double *const *grids);
int gridhip_comm_allreduce_grid_rows(gridhip_comm *comm, int64_t Wd, int64_t y0, int64_t y1, double *grid); /* rank form */
/* Communicator options:
 *   "collective"  0 (default) = one ncclAllReduce; 1 = ncclReduceScatter + ncclAllGather, both in place (rank r owns
 *                 the r-th of nranks equal chunks; the remainder goes through a small all-reduce): the direct schedule
 *                 on xGMI's point-to-point links - every GPU exchanges one chunk with each peer at once */
int gridhip_comm_set_option(gridhip_comm *comm, const char *key, int64_t value);
int gridhip_comm_get_option(gridhip_comm *comm, const char *key, int64_t *value);
/* The hipStream_t device i's collectives are enqueued on.  Default: the context's own stream, i.e. ordered after its
 * gridding calls and before the next one.  A host that wants step i's all-reduce to run beside step i+1's gridding
 * passes a side stream here and orders the two itself (an event recorded after the gridding, waited for by the side
 * stream; python/gridhip/distributed.py: OverlappedCommReducer).  A collective's kernel that becomes ready at the
 * boundary between two steps takes its CUs as the tile kernel's work-groups retire; one that becomes ready in the middle
 * of a persistent tile kernel (the all-gather after the reduce-scatter) waits for its end unless the context options
 * "yield_cus" or "reserve_cus" make CUs come free.
 * gridhip_comm_convgrid2 (synchronous) always reduces on the gridding streams. */
int gridhip_comm_set_stream(gridhip_comm *comm, int i, void *hip_stream);
int gridhip_comm_reset_stream(gridhip_comm *comm, int i);
/* convgrid2 over the communicator, host pointers, synchronous (the drop-in form).  Single-process form: the n
 * visibilities are cut into contiguous shards, one per device, gridded concurrently, the partial grids all-reduced
 * and `grid` (accumulated into) returned.  Rank form: every process passes its own shard; the incoming grid is
 * summed over ranks too, so it should be non-zero on one rank only.
 * Failures: a device whose shard fails locally still takes part in the all-reduce (the other devices / ranks would
 * wait in it for ever otherwise) and the call then returns that device's error with `grid` untouched; the internal
 * consistency counter ("errors") of every local device is checked before the grid is handed back.  In the rank form
 * both are LOCAL verdicts: the other ranks have summed this rank's incomplete grid and return GRIDHIP_OK, so the host
 * must agree on the outcome across ranks before it uses the result. */
int gridhip_comm_convgrid2(gridhip_comm *comm, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W,
                           int64_t Q, int64_t gh, int64_t gw, const double *gcf, const double *u,
                           const double *v, int64_t uv_stride, const int64_t *wbin, const double *vis);

/* ---- device memory helpers (so a non-HIP host language can stage buffers) ----------------- */
int gridhip_malloc(gridhip_ctx *ctx, void **dptr, int64_t bytes);
int gridhip_free(gridhip_ctx *ctx, void *dptr);
int gridhip_memcpy_h2d(gridhip_ctx *ctx, void *dst, const void *src, int64_t bytes);
int gridhip_memcpy_d2h(gridhip_ctx *ctx, void *dst, const void *src, int64_t bytes);
int gridhip_memset(gridhip_ctx *ctx, void *dptr, int value, int64_t bytes);

/* ---- timing of the last device call (HIP events on the context's stream) ------------------
 * ms_total covers binning pre-pass + tile kernel; ms_kernel the dominant kernel only.
 * Synchronises on the recorded events. */
int gridhip_last_timing(gridhip_ctx *ctx, double *ms_total, double *ms_prepass,
                        double *ms_kernel);
/* the same for the timed call `back` calls before the last one (0 = the last; the last 64 are kept), so that a
 * benchmark loop can collect every step's device times after the loop instead of synchronising inside it */
int gridhip_timing(gridhip_ctx *ctx, int back, double *ms_total, double *ms_prepass, double *ms_kernel);
int gridhip_enable_timing(gridhip_ctx *ctx, int enable);

#ifdef __cplusplus
}
#endif
#endif /* GRIDHIP_H */
