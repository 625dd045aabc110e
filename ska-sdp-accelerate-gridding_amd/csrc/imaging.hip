// The callers either side of the gridder, on the device: uvw scaling (div3), the w-bin rule and
// findClosest, mirror_uvw, doweight, make_grid_hermitian, the centred FFT, the w-kernel
// generator, and the imaging functions / do_imaging that string them together
// (src/Gridding.hs:84-124, 399-449, 509-605, 610-728, 815-839, 895-907).
//
// Every kernel here is HBM-streaming or tiny; the hot path stays the tile kernel.
#include <dlfcn.h>
#include <math.h>
#include <string.h>


#include "common.h"

namespace gridhip {

// ---------------------------------------------------------------------------------------------
// small kernels

// div3 (src/Gridding.hs:838-839): a true division, not a multiply by the reciprocal
__global__ void scale_kernel(int64_t n, const double *__restrict__ x, int64_t stride, double lam,
                             double *__restrict__ out)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        out[k] = x[k * stride] / lam;
}

// w-bin rule, src/Gridding.hs:426-432: roundedw = wstep * round(w / wstep)
__global__ void wround_kernel(int64_t n, const double *__restrict__ w, int64_t stride, int64_t wstep,
                              int64_t *__restrict__ rw, long long *__restrict__ minmax)
{
    long long mn = 0x7fffffffffffffffLL, mx = -0x7fffffffffffffffLL - 1;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const long long r = (long long)wstep * (long long)round(w[k * stride] / (double)wstep);
        rw[k] = r;
        mn = r < mn ? r : mn;
        mx = r > mx ? r : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        long long a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    // one pair of atomics per work-group (the two 64-bit counters are the same for everybody: one pair per wave was
    // 3 x 10^4 serialised atomics, most of this kernel's 0.39 ms at 10^7 visibilities)
    __shared__ long long smn[16], smx[16];
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        smn[wave] = mn;
        smx[wave] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) {
            mn = smn[i] < mn ? smn[i] : mn;
            mx = smx[i] > mx ? smx[i] : mx;
        }
        atomicMin(&minmax[0], mn);
        atomicMax(&minmax[1], mx);
    }
}

__global__ void wbin_finish_kernel(int64_t n, int64_t *__restrict__ rw, int64_t wstep,
                                   const long long *__restrict__ minmax)
{
    const long long mn = minmax[0];
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        rw[k] = (rw[k] - mn) / wstep;  // non-negative: `div` and C division agree
}

// findClosest, src/Gridding.hs:895-907 (hi clamped to len-1 as the host twin does, ImageDataset.hs:150-168)
__global__ void find_closest_kernel(int64_t nws, const double *__restrict__ ws, int64_t n,
                                    const double *__restrict__ w, int64_t stride, int64_t *__restrict__ out)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const double x = w[k * stride];
        int64_t lo = 0, hi = nws;
        while ((hi - lo) / 2 >= 1) {
            const int64_t mid = (hi + lo) / 2;
            if (x > ws[mid])
                lo = mid;
            else
                hi = mid;
        }
        const int64_t hc = hi > nws - 1 ? nws - 1 : hi;
        out[k] = fabs(x - ws[lo]) < fabs(x - ws[hc]) ? lo : hc;
    }
}

// mirror_uvw, src/Gridding.hs:551-562
__global__ void mirror_kernel(int64_t n, double *__restrict__ u, double *__restrict__ v, double *__restrict__ w,
                              double2 *__restrict__ vis)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        if (v[k] < 0) {
            u[k] = -u[k];
            v[k] = -v[k];
            if (w) w[k] = -w[k];
            if (vis) vis[k].y = -vis[k].y;
        }
    }
}

// doweight, src/Gridding.hs:564-583: frac_coords (N,N) 1 p -> cell histogram -> v / count
__device__ __forceinline__ int64_t weight_cell(int64_t N, double pu, double pv)
{
    int64_t x, y;
    int32_t f;
    frac_coord_dev(N, 1, pu, &x, &f);
    frac_coord_dev(N, 1, pv, &y, &f);
    if (!(pu == pu) || !(pv == pv) || x < 0 || y < 0 || x >= N || y >= N) return -1;
    return y * N + x;
}

__global__ void weight_hist_kernel(int64_t N, int64_t n, const double *__restrict__ pu, const double *__restrict__ pv,
                                   unsigned int *__restrict__ cnt)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = weight_cell(N, pu[k], pv[k]);
        if (c >= 0) atomicAdd(&cnt[c], 1u);
    }
}

__global__ void weight_apply_kernel(int64_t N, int64_t n, const double *__restrict__ pu, const double *__restrict__ pv,
                                    const unsigned int *__restrict__ cnt, double2 *__restrict__ vis)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = weight_cell(N, pu[k], pv[k]);
        if (c < 0) continue;
        const double wgt = (double)cnt[c];
        double2 v = vis[k];
        v.x /= wgt;
        v.y /= wgt;
        vis[k] = v;
    }
}

// make_grid_hermitian, src/Gridding.hs:585-605 (out of place).  `s`: the output is the Hermitian grid rolled by s both
// ways (out[y][x] = H[(y+s) mod N][(x+s) mod N]) - the ishift2D the centred transform starts with, written at once.
__global__ void hermitian_kernel(int64_t N, const double2 *__restrict__ in, double2 *__restrict__ out, int64_t s)
{
    const bool even = (N % 2) == 0;
    const int64_t cells = N * N;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (int64_t)gridDim.x * blockDim.x) {
        int64_t y = c / N, x = c - y * N;
        y += s;
        x += s;
        y -= y >= N ? N : 0;
        x -= x >= N ? N : 0;
        double2 a = make_double2(0.0, 0.0);
        if (even) {
            if (x != 0 && y != 0) a = in[(N - y) * N + (N - x)];
        } else {
            a = in[(N - 1 - y) * N + (N - 1 - x)];
        }
        const double2 g = in[y * N + x];
        out[c] = make_double2(g.x + a.x, g.y - a.y);
    }
}

// out[y][x] = in[(y+s) mod N][(x+s) mod N] * scale   (shift2D: s = ceil(N/2), ishift2D: s = floor(N/2))
__global__ void roll_kernel(int64_t N, const double2 *__restrict__ in, double2 *__restrict__ out, int64_t s,
                            double scale)
{
    const int64_t cells = N * N;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = c / N, x = c - y * N;
        int64_t sy = y + s, sx = x + s;
        sy -= sy >= N ? N : 0;
        sx -= sx >= N ? N : 0;
        const double2 v = in[sy * N + sx];
        out[c] = make_double2(v.x * scale, v.y * scale);
    }
}

// N > 0: `in` is a transform's raw N x N output and the cell read for c = (y, x) is in[(y+s) mod N][(x+s) mod N] * scale -
// the shift2D and the 1 / N^2 the centred inverse transform ends with, applied while the real part is taken instead
// of in a pass of their own (the same multiplication: bit-identical).
__global__ void real_max_kernel(int64_t cells, const double2 *__restrict__ in, double *__restrict__ real_out,
                                unsigned long long *__restrict__ maxbits, int64_t N, int64_t s, double scale)
{
    double m = -INFINITY;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (int64_t)gridDim.x * blockDim.x) {
        double r;
        if (N > 0) {
            int64_t y = c / N, x = c - y * N;
            y += s;
            x += s;
            y -= y >= N ? N : 0;
            x -= x >= N ? N : 0;
            r = in[y * N + x].x * scale;
        } else
            r = in[c].x;
        if (real_out) real_out[c] = r;
        m = r > m ? r : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    // one atomic per work-group (one per wave was 1.6 x 10^4 serialised 64-bit atomics on one address: most of this
    // kernel's 0.11 ms at 2400^2 cells)
    __shared__ double sm[16];
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) sm[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0 && maxbits) {
        for (int i = 1; i < nw; ++i) m = sm[i] > m ? sm[i] : m;
        // order-preserving map of doubles onto unsigned integers so atomicMax works
        unsigned long long b = (unsigned long long)__double_as_longlong(m);
        b = (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
        atomicMax(maxbits, b);
    }
}

__global__ void divide_kernel(int64_t cells, double *__restrict__ x, const unsigned long long *__restrict__ maxbits)
{
    unsigned long long b = *maxbits;
    b = (b & 0x8000000000000000ULL) ? (b & 0x7fffffffffffffffULL) : ~b;
    const double m = __longlong_as_double((long long)b);
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (int64_t)gridDim.x * blockDim.x)
        x[c] /= m;
}

__global__ void cmul_real_kernel(int64_t n, const double2 *__restrict__ a, const double2 *__restrict__ b,
                                 double2 *__restrict__ out)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const double2 x = a[k], y = b[k];
        out[k] = make_double2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x);
    }
}

__global__ void fill_ones_kernel(int64_t n, double2 *__restrict__ a)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        a[k] = make_double2(1.0, 0.0);
}

// w_kernel far field, padded (src/Gridding.hs:610-667, pad_mid :682-691 via padder :863-877).
// padder reads `array ! index2 oldx oldy`: the far field is transposed while it is padded.
// `s`: the output is the padded far field rolled by s both ways (out[y][x] = field[(y+s) mod na][(x+s) mod na]) - the
// ishift2D the centred transform starts with, written at once instead of by a pass of its own.
__global__ void wkern_farfield_kernel(int64_t n, int64_t na, double theta, double w, double2 *__restrict__ out, int64_t s)
{
    const int64_t p0 = na / 2 - n / 2;
    const double step = 1.0 / (double)n;
    const double start = (double)(-(n / 2)) * step;
    const int64_t cells = na * na;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (int64_t)gridDim.x * blockDim.x) {
        int64_t y = c / na, x = c - y * na;
        y += s;
        x += s;
        y -= y >= na ? na : 0;
        x -= x >= na ? na : 0;
        int64_t oldx = x - p0, oldy = y - p0;
        double2 v = make_double2(0.0, 0.0);
        if (n == na) {  // pad_mid returns ff untouched
            oldx = y;
            oldy = x;
        }
        if (oldx >= 0 && oldx < n && oldy >= 0 && oldy < n) {
            // ff[row = oldx][col = oldy]: l = base[col] * theta, m = base[row] * theta
#pragma clang fp contract(off)
            const double l = (start + (double)oldy * step) * theta;
            const double m = (start + (double)oldx * step) * theta;
            const double r2 = l * l + m * m;
            const double ph = 1.0 - sqrt(1.0 - r2);
            const double arg = 2.0 * M_PI * w * ph;
            double sn, cs;
            sincos(arg, &sn, &cs);
            v = make_double2(cs, sn);
        }
        out[c] = v;
    }
}

// extract_oversampled, src/Gridding.hs:709-728: K[yf,xf,y,x] = af[c - yf + Q*y, c - xf + Q*x] * Q^2
// `s`, `scale`: af is the transform's raw output; the cell the reference reads is af[(row+s) mod na][(col+s) mod na] *
// scale - the shift2D and the 1 / na^2 the centred inverse transform ends with, applied to the Q^2 S^2 cells that are
// used instead of to all na^2 (same two multiplications in the same order: bit-identical).
__global__ void wkern_extract_kernel(int64_t na, int64_t Q, int64_t S, const double2 *__restrict__ af,
                                     double2 *__restrict__ out, int conj, int64_t s, double scale)
{
    const int64_t c0 = na / 2 - Q * (S / 2);
    const double q2 = (double)(Q * Q);
    const int64_t total = Q * Q * S * S;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t;
        const int64_t x = r % S;
        r /= S;
        const int64_t y = r % S;
        r /= S;
        const int64_t xf = r % Q, yf = r / Q;
        int64_t row = c0 - yf + Q * y + s, col = c0 - xf + Q * x + s;
        row -= row >= na ? na : 0;
        col -= col >= na ? na : 0;
        double2 v = af[row * na + col];
        v = make_double2(v.x * scale, v.y * scale);
        out[t] = make_double2(v.x * q2, conj ? -(v.y * q2) : v.y * q2);
    }
}

static inline dim3 grid_for(gridhip_ctx *ctx, int64_t n, int block = 256)
{
    int64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (int64_t)ctx->num_cu * 16) b = (int64_t)ctx->num_cu * 16;
    return dim3((unsigned)b);
}

// ---------------------------------------------------------------------------------------------
// hipFFT, loaded on first use so the gridder itself has no dependency on it

typedef int (*fn_plan2d)(void **, int, int, int);
typedef int (*fn_setstream)(void *, hipStream_t);
typedef int (*fn_exec)(void *, void *, void *, int);
typedef int (*fn_destroy)(void *);
static struct {
    void *h = nullptr;
    fn_plan2d plan2d;
    fn_setstream setstream;
    fn_exec exec;
    fn_destroy destroy;
    bool tried = false;
} g_fft;

static int load_hipfft(gridhip_ctx *ctx)
{
    if (g_fft.h) return GRIDHIP_OK;
    if (!g_fft.tried) {
        g_fft.tried = true;
        const char *names[] = {"libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so"};
        for (const char *nm : names) {
            g_fft.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (g_fft.h) break;
        }
        if (g_fft.h) {
            g_fft.plan2d = (fn_plan2d)dlsym(g_fft.h, "hipfftPlan2d");
            g_fft.setstream = (fn_setstream)dlsym(g_fft.h, "hipfftSetStream");
            g_fft.exec = (fn_exec)dlsym(g_fft.h, "hipfftExecZ2Z");
            g_fft.destroy = (fn_destroy)dlsym(g_fft.h, "hipfftDestroy");
            if (!g_fft.plan2d || !g_fft.setstream || !g_fft.exec || !g_fft.destroy) {
                dlclose(g_fft.h);
                g_fft.h = nullptr;
            }
        }
    }
    if (!g_fft.h) return fail(ctx, GRIDHIP_EHIP, "cannot load libhipfft.so: %s", dlerror());
    return GRIDHIP_OK;
}

// centred transform (src/Gridding.hs:815-829): shift2D . fft2D mode . ishift2D.
// accelerate-fft: Forward = exp(-i..) unnormalised, Inverse = exp(+i..) scaled by 1/N^2.
// `in` is preserved, `tmp` and `out` are N*N scratch/output (out may not alias in).
// the context's cached N x N Z2Z plan, bound to its stream
static int fft_plan_for(gridhip_ctx *ctx, int64_t N, void **out_plan)
{
    GH_CHECK(load_hipfft(ctx));
    if (N > 0x7fffffff) return fail(ctx, GRIDHIP_EUNSUPPORTED, "fft size");
    void *plan = nullptr;
    for (int i = 0; i < 4; ++i)
        if (ctx->fft_plan[i] && ctx->fft_n[i] == N) plan = ctx->fft_plan[i];
    if (!plan) {
        const int slot = ctx->fft_next;
        if (ctx->fft_plan[slot]) {
            // (a plan may still be in use by work queued on the stream)
            GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            g_fft.destroy(ctx->fft_plan[slot]);
        }
        ctx->fft_plan[slot] = nullptr;
        int rc = g_fft.plan2d(&plan, (int)N, (int)N, 0x69 /* HIPFFT_Z2Z */);
        if (rc) return fail(ctx, GRIDHIP_EHIP, "hipfftPlan2d(%lld) failed: %d", (long long)N, rc);
        ctx->fft_plan[slot] = plan;
        ctx->fft_n[slot] = N;
        ctx->fft_next = (slot + 1) % 4;
    }
    if (int rc = g_fft.setstream(plan, ctx->stream)) return fail(ctx, GRIDHIP_EHIP, "hipfftSetStream: %d", rc);
    *out_plan = plan;
    return GRIDHIP_OK;
}

int dev_fft2c(gridhip_ctx *ctx, int64_t N, const double2 *in, double2 *out, double2 *tmp, bool inverse)
{
    void *plan = nullptr;
    GH_CHECK(fft_plan_for(ctx, N, &plan));
    hipLaunchKernelGGL(roll_kernel, grid_for(ctx, N * N), dim3(256), 0, ctx->stream, N, in, tmp, N / 2, 1.0);
    if (int rc = g_fft.exec(plan, tmp, tmp, inverse ? 1 /* HIPFFT_BACKWARD */ : -1 /* HIPFFT_FORWARD */))
        return fail(ctx, GRIDHIP_EHIP, "hipfftExecZ2Z: %d", rc);
    const double sc = inverse ? 1.0 / ((double)N * (double)N) : 1.0;
    hipLaunchKernelGGL(roll_kernel, grid_for(ctx, N * N), dim3(256), 0, ctx->stream, N, tmp, out, (N + 1) / 2, sc);
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

void fft_release(gridhip_ctx *ctx)
{
    if (ctx->wk_cache.ptr) (void)hipFree(ctx->wk_cache.ptr);
    ctx->wk_cache.ptr = nullptr;
    ctx->wk_cache.bytes = 0;
    ctx->wk_cache.nplanes = 0;
    for (int i = 0; i < 4; ++i) {
        if (ctx->fft_plan[i] && g_fft.h) g_fft.destroy(ctx->fft_plan[i]);
        ctx->fft_plan[i] = nullptr;
    }
}

// one plane of the w-kernel table: out[Q][Q][S][S] (conjugated when the caller is w_cache_imaging, :441)
int dev_w_kernel(gridhip_ctx *ctx, double theta, double w, int64_t npixFF, int64_t S, int64_t Q, double2 *out,
                 bool conj, double2 *pad, double2 *af, double2 *tmp)
{
    const int64_t na = npixFF * Q;
    // centred inverse transform = shift2D . ifft2D . ishift2D (dev_fft2c); its two rolls are folded into the far-field
    // kernel's stores and the extraction's loads: three passes over na^2 cells fewer per plane
    (void)af;
    (void)tmp;
    void *plan = nullptr;
    GH_CHECK(fft_plan_for(ctx, na, &plan));
    hipLaunchKernelGGL(wkern_farfield_kernel, grid_for(ctx, na * na), dim3(256), 0, ctx->stream, npixFF, na, theta, w,
                       pad, na / 2);
    if (int rc = g_fft.exec(plan, pad, pad, 1 /* HIPFFT_BACKWARD */)) return fail(ctx, GRIDHIP_EHIP, "hipfftExecZ2Z: %d", rc);
    hipLaunchKernelGGL(wkern_extract_kernel, grid_for(ctx, Q * Q * S * S), dim3(256), 0, ctx->stream, na, Q, S, pad,
                       out, conj ? 1 : 0, (na + 1) / 2, 1.0 / ((double)na * (double)na));
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

// Device buffer of one imaging call, drawn from and returned to the context's pool (gridhip_ctx::pool_free): the
// smallest pooled block of at least the size asked for and at most twice it, else a new one.  Nothing is freed
// before gridhip_destroy, so the second call of a given shape allocates nothing.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    gridhip_ctx *owner = nullptr;
    ~DevBuf()
    {
        if (p && owner) owner->pool_free.emplace_back(p, cap);
    }
    int alloc(gridhip_ctx *ctx, size_t bytes)
    {
        if (bytes < 256) bytes = 256;
        owner = ctx;
        int best = -1;
        for (int i = 0; i < (int)ctx->pool_free.size(); ++i) {
            const size_t c = ctx->pool_free[i].second;
            if (c >= bytes && c <= 2 * bytes && (best < 0 || c < ctx->pool_free[best].second)) best = i;
        }
        if (best >= 0) {
            p = ctx->pool_free[best].first;
            cap = ctx->pool_free[best].second;
            ctx->pool_free.erase(ctx->pool_free.begin() + best);
            return GRIDHIP_OK;
        }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipErrorOutOfMemory && !ctx->pool_free.empty()) {  // give the pooled blocks back and try once more
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
            for (auto &b : ctx->pool_free) (void)hipFree(b.first);
            ctx->pool_free.clear();
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) {
            p = nullptr;
            return fail(ctx, e == hipErrorOutOfMemory ? GRIDHIP_ENOMEM : GRIDHIP_EHIP, "hipMalloc(%zu) failed: %s", bytes,
                        hipGetErrorString(e));
        }
        cap = bytes;
        return GRIDHIP_OK;
    }
    template <typename T>
    T *as()
    {
        return reinterpret_cast<T *>(p);
    }
};

// inputs / outputs of an imaging call: host arrays (the drop-in forms) or device-resident ones (the _dev forms)
static int copy_in(gridhip_ctx *ctx, void *d, const void *src, size_t bytes, bool dev)
{
    if (bytes) GH_CHECK_HIP(ctx, hipMemcpyAsync(d, src, bytes, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    return GRIDHIP_OK;
}
static int copy_out(gridhip_ctx *ctx, void *dst, const void *d, size_t bytes, bool dev)
{
    if (bytes) GH_CHECK_HIP(ctx, hipMemcpyAsync(dst, d, bytes, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return GRIDHIP_OK;
}

static int h2d(gridhip_ctx *ctx, void *d, const void *h, size_t bytes)
{
    if (bytes) GH_CHECK_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GRIDHIP_OK;
}
static int d2h(gridhip_ctx *ctx, void *h, const void *d, size_t bytes)
{
    if (bytes) GH_CHECK_HIP(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return GRIDHIP_OK;
}
static int sync(gridhip_ctx *ctx)
{
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

// Prelude `round` on the host (N = round (theta * lam), src/Gridding.hs:87,118,416): half to even
static int64_t haskell_round(double x) { return (int64_t)nearbyint(x); }

// w-bins on the device; returns min and plane count to the host (the reference does the same
// round-trip with a nested CPU.run, :430)
static int dev_wbins(gridhip_ctx *ctx, int64_t n, const double *w, int64_t stride, int64_t wstep, int64_t *wbin,
                     int64_t *wmin, int64_t *nplanes)
{
    DevBuf mm;
    GH_CHECK(mm.alloc(ctx, 16));
    const long long init[2] = {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1};
    GH_CHECK(h2d(ctx, mm.p, init, 16));
    if (n > 0) {
        hipLaunchKernelGGL(wround_kernel, dim3(grid_for(ctx, n).x > (unsigned)ctx->num_cu * 4 ? (unsigned)ctx->num_cu * 4 : grid_for(ctx, n).x), dim3(256), 0, ctx->stream, n, w, stride, wstep, wbin,
                           mm.as<long long>());
        hipLaunchKernelGGL(wbin_finish_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, wbin, wstep,
                           mm.as<long long>());
    }
    long long res[2];
    GH_CHECK(d2h(ctx, res, mm.p, 16));
    GH_CHECK(sync(ctx));
    GH_CHECK_HIP(ctx, hipGetLastError());
    *wmin = n > 0 ? res[0] : 0;
    *nplanes = n > 0 ? (res[1] - res[0]) / wstep + 1 : 0;
    return GRIDHIP_OK;
}

// mirror + weights + the two gridding passes + hermitian + iFFT + normalise, all on the device.
// `imgfn` grids (u1, v1, w1, vis) -> zeroed N x N grid.
template <typename ImgFn>
static int do_imaging_impl(gridhip_ctx *ctx, double theta, int64_t lam, int64_t n, const double *u, const double *v,
                           const double *w, int64_t stride, const double *vis, double *image, double *psf,
                           double *pmax, bool dev, ImgFn imgfn)
{
    const int64_t N = haskell_round(theta * (double)lam);
    if (N <= 0) return fail(ctx, GRIDHIP_EINVAL, "theta*lam rounds to %lld", (long long)N);
    const size_t cells = (size_t)N * N;
    DevBuf du, dv, dw, dvis, dwt, dpu, dpv, dg, dh, dtmp, dreal, dmax;
    GH_CHECK(du.alloc(ctx, n * 8));
    GH_CHECK(dv.alloc(ctx, n * 8));
    GH_CHECK(dw.alloc(ctx, n * 8));
    GH_CHECK(dvis.alloc(ctx, n * 16));
    GH_CHECK(dwt.alloc(ctx, n * 16));
    GH_CHECK(dpu.alloc(ctx, n * 8));
    GH_CHECK(dpv.alloc(ctx, n * 8));
    GH_CHECK(dg.alloc(ctx, cells * 16));
    GH_CHECK(dh.alloc(ctx, cells * 16));
    GH_CHECK(dtmp.alloc(ctx, cells * 16));
    GH_CHECK(dreal.alloc(ctx, cells * 8));
    GH_CHECK(dmax.alloc(ctx, 8));
    // slice the columns (src/Gridding.hs:524-526) while uploading (device-resident inputs: straight from them)
    {
        const size_t span = n > 0 ? (size_t)(n - 1) * stride + 1 : 0;
        DevBuf s0, s1, s2;
        const double *su = u, *sv = v, *sw = w;
        if (!dev) {
            GH_CHECK(s0.alloc(ctx, span * 8));
            GH_CHECK(s1.alloc(ctx, span * 8));
            GH_CHECK(s2.alloc(ctx, span * 8));
            GH_CHECK(h2d(ctx, s0.p, u, span * 8));
            GH_CHECK(h2d(ctx, s1.p, v, span * 8));
            GH_CHECK(h2d(ctx, s2.p, w, span * 8));
            su = s0.as<double>(), sv = s1.as<double>(), sw = s2.as<double>();
        }
        if (n > 0) {
            hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, su, stride, 1.0, du.as<double>());
            hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, sv, stride, 1.0, dv.as<double>());
            hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, sw, stride, 1.0, dw.as<double>());
        }
        if (!dev) GH_CHECK(sync(ctx));  // (the staging blocks go back to the pool; all later work is stream-ordered after this)
    }
    GH_CHECK(copy_in(ctx, dvis.p, vis, n * 16, dev));
    if (n > 0) {
        // mirror baselines such that v >= 0 (:531)
        hipLaunchKernelGGL(mirror_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, du.as<double>(),
                           dv.as<double>(), dw.as<double>(), dvis.as<double2>());
        // weights (:534-535): doweight theta lam uvw1 ones
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, du.as<double>(), (int64_t)1,
                           (double)lam, dpu.as<double>());
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, dv.as<double>(), (int64_t)1,
                           (double)lam, dpv.as<double>());
        hipLaunchKernelGGL(fill_ones_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, dwt.as<double2>());
        GH_CHECK_HIP(ctx, hipMemsetAsync(dtmp.p, 0, cells * 4, ctx->stream));
        hipLaunchKernelGGL(weight_hist_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, N, n, dpu.as<double>(),
                           dpv.as<double>(), dtmp.as<unsigned int>());
        hipLaunchKernelGGL(weight_apply_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, N, n, dpu.as<double>(),
                           dpv.as<double>(), dtmp.as<unsigned int>(), dwt.as<double2>());
        // wt * vis1 (:538)
        hipLaunchKernelGGL(cmul_real_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, dwt.as<double2>(),
                           dvis.as<double2>(), dvis.as<double2>());
    }
    const unsigned long long neg_inf_bits = ~0xfff0000000000000ULL;  // ordered image of -inf
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: image from wt*vis (:538-539); pass 1: PSF from wt (:541-542)
        GH_CHECK_HIP(ctx, hipMemsetAsync(dg.p, 0, cells * 16, ctx->stream));
        GH_CHECK(imgfn(N, du.as<double>(), dv.as<double>(), dw.as<double>(),
                       pass == 0 ? dvis.as<double>() : dwt.as<double>(), dg.as<double>()));
        // centred inverse transform (shift2D . ifft2D . ishift2D, dev_fft2c) with its two rolls folded into the Hermitian
        // fill's stores and the real part's loads: four passes over the N^2 grid fewer per call
        void *plan = nullptr;
        GH_CHECK(fft_plan_for(ctx, N, &plan));
        hipLaunchKernelGGL(hermitian_kernel, grid_for(ctx, cells), dim3(256), 0, ctx->stream, N, dg.as<double2>(),
                           dh.as<double2>(), N / 2);
        if (int rc = g_fft.exec(plan, dh.p, dh.p, 1 /* HIPFFT_BACKWARD */)) return fail(ctx, GRIDHIP_EHIP, "hipfftExecZ2Z: %d", rc);
        if (pass == 1) GH_CHECK(h2d(ctx, dmax.p, &neg_inf_bits, 8));
        hipLaunchKernelGGL(real_max_kernel, dim3((unsigned)ctx->num_cu * 4), dim3(256), 0, ctx->stream, (int64_t)cells,
                           dh.as<double2>(), pass == 0 ? dreal.as<double>() : dtmp.as<double>(),
                           pass == 1 ? dmax.as<unsigned long long>() : (unsigned long long *)nullptr, N, (N + 1) / 2,
                           1.0 / ((double)N * (double)N));
    }
    // normalise both by max(psf) (:544-548); dtmp holds the real PSF, dreal the real image
    hipLaunchKernelGGL(divide_kernel, grid_for(ctx, cells), dim3(256), 0, ctx->stream, (int64_t)cells,
                       dreal.as<double>(), dmax.as<unsigned long long>());
    unsigned long long mb = 0;
    GH_CHECK(d2h(ctx, &mb, dmax.p, 8));
    GH_CHECK(sync(ctx));
    mb = (mb & 0x8000000000000000ULL) ? (mb & 0x7fffffffffffffffULL) : ~mb;
    double m;
    memcpy(&m, &mb, 8);
    if (pmax) *pmax = m;
    hipLaunchKernelGGL(divide_kernel, grid_for(ctx, cells), dim3(256), 0, ctx->stream, (int64_t)cells,
                       dtmp.as<double>(), dmax.as<unsigned long long>());
    GH_CHECK_HIP(ctx, hipGetLastError());
    if (image) GH_CHECK(copy_out(ctx, image, dreal.p, cells * 8, dev));
    if (psf) GH_CHECK(copy_out(ctx, psf, dtmp.p, cells * 8, dev));
    GH_CHECK(sync(ctx));
    return GRIDHIP_OK;
}

// builds [W][Q][Q][S][S] conjugated w-kernels for planes w = i*wstep + wmin (:434-448)
static int build_wkernels(gridhip_ctx *ctx, double theta, int64_t wstep, int64_t wmin, int64_t nplanes,
                          int64_t npixFF, int64_t S, int64_t Q, double2 *table)
{
    const int64_t na = npixFF * Q;
    DevBuf pad, af, tmp;
    GH_CHECK(pad.alloc(ctx, na * na * 16));
    GH_CHECK(af.alloc(ctx, na * na * 16));
    GH_CHECK(tmp.alloc(ctx, na * na * 16));
    for (int64_t i = 0; i < nplanes; ++i)
        GH_CHECK(dev_w_kernel(ctx, theta, (double)(i * wstep + wmin), npixFF, S, Q, table + i * Q * Q * S * S, true,
                              pad.as<double2>(), af.as<double2>(), tmp.as<double2>()));
    GH_CHECK(sync(ctx));
    return GRIDHIP_OK;
}

// What w_cache_imaging (src/Gridding.hs:399-449) derives from the baselines alone: scaled u, v, the
// w-bins and one conjugated w-kernel per plane.  do_imaging calls the imaging function twice with the
// same baselines (image and PSF, :538,541); the reference rebuilds everything both times ("no cache
// despite the name", :405-411) — here the second call reuses it.
struct WCache {
    DevBuf pu, pv, wb;
    double2 *table = nullptr;  // the context's cached table (gridhip_ctx::wk_cache): not owned
    int64_t nplanes = 0;
    bool ready = false;
    gridhip_plan *plan = nullptr;  // the baselines binned once for both passes
    ~WCache() { gridhip_plan_destroy(plan); }
};

static int w_cache_prepare(gridhip_ctx *ctx, WCache &c, double theta, int64_t lam, int64_t wstep, int64_t Q,
                           int64_t npixFF, int64_t S, int64_t n, const double *u, const double *v, const double *w)
{
    GH_CHECK(c.pu.alloc(ctx, n * 8));
    GH_CHECK(c.pv.alloc(ctx, n * 8));
    GH_CHECK(c.wb.alloc(ctx, n * 8));
    if (n > 0) {
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, u, (int64_t)1, (double)lam,
                           c.pu.as<double>());
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, v, (int64_t)1, (double)lam,
                           c.pv.as<double>());
    }
    int64_t wmin = 0;
    GH_CHECK(dev_wbins(ctx, n, w, 1, wstep, c.wb.as<int64_t>(), &wmin, &c.nplanes));
    c.ready = true;
    if (n == 0 || c.nplanes == 0) return GRIDHIP_OK;
    if (c.nplanes > 65536) return fail(ctx, GRIDHIP_EUNSUPPORTED, "%lld w-planes", (long long)c.nplanes);
    auto &k = ctx->wk_cache;
    const size_t bytes = (size_t)c.nplanes * Q * Q * S * S * 16;
    if (!(k.ptr && k.theta == theta && k.wstep == wstep && k.wmin == wmin && k.nplanes == c.nplanes && k.npixFF == npixFF &&
          k.S == S && k.Q == Q)) {
        k.nplanes = 0;  // (not valid while it is rebuilt)
        if (k.bytes < bytes) {
            if (k.ptr) {
                GH_CHECK(sync(ctx));  // (the old table may still be read by work queued on the stream)
                (void)hipFree(k.ptr);
                k.ptr = nullptr;
                k.bytes = 0;
            }
            if (hipMalloc(&k.ptr, bytes) != hipSuccess) {
                k.ptr = nullptr;
                return fail(ctx, GRIDHIP_ENOMEM, "w-kernel table: %zu bytes", bytes);
            }
            k.bytes = bytes;
        }
        GH_CHECK(build_wkernels(ctx, theta, wstep, wmin, c.nplanes, npixFF, S, Q, (double2 *)k.ptr));
        k.theta = theta;
        k.wstep = wstep;
        k.wmin = wmin;
        k.nplanes = c.nplanes;
        k.npixFF = npixFF;
        k.S = S;
        k.Q = Q;
    }
    c.table = (double2 *)k.ptr;
    return GRIDHIP_OK;
}

// u,v,w in wavelengths, grid zeroed N x N
static int w_cache_grid_dev(gridhip_ctx *ctx, WCache &c, double theta, int64_t lam, int64_t wstep, int64_t Q,
                            int64_t npixFF, int64_t S, int64_t N, int64_t n, const double *u, const double *v,
                            const double *w, const double *vis, double *grid)
{
    if (!c.ready) GH_CHECK(w_cache_prepare(ctx, c, theta, lam, wstep, Q, npixFF, S, n, u, v, w));
    if (n == 0 || c.nplanes == 0) return GRIDHIP_OK;
    if (!c.plan)
        GH_CHECK(plan_create_borrowed(ctx, N, N, n, c.nplanes, Q, S, S, c.pu.as<double>(), c.pv.as<double>(), 1,
                                      c.wb.as<int64_t>(), &c.plan));  // (nothing else grids on this context before both passes are done)
    GH_CHECK(gridhip_plan_grid_dev(c.plan, (const double *)c.table, vis, grid));
    return sync(ctx);
}

}  // namespace gridhip

using namespace gridhip;

extern "C" {

int64_t gridhip_image_size(double theta, int64_t lam) { return haskell_round(theta * (double)lam); }

int gridhip_wbins(gridhip_ctx *ctx, int64_t n, const double *w, int64_t wstep, int64_t *wbin, int64_t *wmin,
                  int64_t *nplanes)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (n < 0 || wstep <= 0 || (n > 0 && (!w || !wbin)) || !wmin || !nplanes)
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf dw, db;
    GH_CHECK(dw.alloc(ctx, n * 8));
    GH_CHECK(db.alloc(ctx, n * 8));
    GH_CHECK(h2d(ctx, dw.p, w, n * 8));
    GH_CHECK(dev_wbins(ctx, n, dw.as<double>(), 1, wstep, db.as<int64_t>(), wmin, nplanes));
    GH_CHECK(d2h(ctx, wbin, db.p, n * 8));
    return sync(ctx);
}

int gridhip_find_closest(gridhip_ctx *ctx, int64_t nws, const double *ws, int64_t n, const double *w, int64_t *out)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (nws <= 0 || n < 0 || !ws || (n > 0 && (!w || !out))) return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf dws, dw, dout;
    GH_CHECK(dws.alloc(ctx, nws * 8));
    GH_CHECK(dw.alloc(ctx, n * 8));
    GH_CHECK(dout.alloc(ctx, n * 8));
    GH_CHECK(h2d(ctx, dws.p, ws, nws * 8));
    GH_CHECK(h2d(ctx, dw.p, w, n * 8));
    if (n > 0)
        hipLaunchKernelGGL(find_closest_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, nws, dws.as<double>(), n,
                           dw.as<double>(), (int64_t)1, dout.as<int64_t>());
    GH_CHECK_HIP(ctx, hipGetLastError());
    GH_CHECK(d2h(ctx, out, dout.p, n * 8));
    return sync(ctx);
}

int gridhip_mirror_uvw(gridhip_ctx *ctx, int64_t n, double *u, double *v, double *w, double *vis)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (n < 0 || (n > 0 && (!u || !v))) return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf du, dv, dw, dvis;
    GH_CHECK(du.alloc(ctx, n * 8));
    GH_CHECK(dv.alloc(ctx, n * 8));
    GH_CHECK(dw.alloc(ctx, n * 8));
    GH_CHECK(dvis.alloc(ctx, n * 16));
    GH_CHECK(h2d(ctx, du.p, u, n * 8));
    GH_CHECK(h2d(ctx, dv.p, v, n * 8));
    if (w) GH_CHECK(h2d(ctx, dw.p, w, n * 8));
    if (vis) GH_CHECK(h2d(ctx, dvis.p, vis, n * 16));
    if (n > 0)
        hipLaunchKernelGGL(mirror_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, du.as<double>(),
                           dv.as<double>(), w ? dw.as<double>() : nullptr, vis ? dvis.as<double2>() : nullptr);
    GH_CHECK_HIP(ctx, hipGetLastError());
    GH_CHECK(d2h(ctx, u, du.p, n * 8));
    GH_CHECK(d2h(ctx, v, dv.p, n * 8));
    if (w) GH_CHECK(d2h(ctx, w, dw.p, n * 8));
    if (vis) GH_CHECK(d2h(ctx, vis, dvis.p, n * 16));
    return sync(ctx);
}

int gridhip_doweight(gridhip_ctx *ctx, double theta, int64_t lam, int64_t n, const double *u, const double *v,
                     double *vis)
{
    if (!ctx) return GRIDHIP_EINVAL;
    const int64_t N = haskell_round(theta * (double)lam);
    if (N <= 0 || n < 0 || (n > 0 && (!u || !v || !vis))) return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf du, dv, dpu, dpv, dvis, cnt;
    GH_CHECK(du.alloc(ctx, n * 8));
    GH_CHECK(dv.alloc(ctx, n * 8));
    GH_CHECK(dpu.alloc(ctx, n * 8));
    GH_CHECK(dpv.alloc(ctx, n * 8));
    GH_CHECK(dvis.alloc(ctx, n * 16));
    GH_CHECK(cnt.alloc(ctx, (size_t)N * N * 4));
    GH_CHECK(h2d(ctx, du.p, u, n * 8));
    GH_CHECK(h2d(ctx, dv.p, v, n * 8));
    GH_CHECK(h2d(ctx, dvis.p, vis, n * 16));
    GH_CHECK_HIP(ctx, hipMemsetAsync(cnt.p, 0, (size_t)N * N * 4, ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, du.as<double>(), (int64_t)1,
                           (double)lam, dpu.as<double>());
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, dv.as<double>(), (int64_t)1,
                           (double)lam, dpv.as<double>());
        hipLaunchKernelGGL(weight_hist_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, N, n, dpu.as<double>(),
                           dpv.as<double>(), cnt.as<unsigned int>());
        hipLaunchKernelGGL(weight_apply_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, N, n, dpu.as<double>(),
                           dpv.as<double>(), cnt.as<unsigned int>(), dvis.as<double2>());
    }
    GH_CHECK_HIP(ctx, hipGetLastError());
    GH_CHECK(d2h(ctx, vis, dvis.p, n * 16));
    return sync(ctx);
}

int gridhip_make_grid_hermitian(gridhip_ctx *ctx, int64_t N, double *grid)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (N <= 0 || !grid) return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)N * N;
    DevBuf a, b;
    GH_CHECK(a.alloc(ctx, cells * 16));
    GH_CHECK(b.alloc(ctx, cells * 16));
    GH_CHECK(h2d(ctx, a.p, grid, cells * 16));
    hipLaunchKernelGGL(hermitian_kernel, grid_for(ctx, cells), dim3(256), 0, ctx->stream, N, a.as<double2>(),
                       b.as<double2>(), (int64_t)0);
    GH_CHECK_HIP(ctx, hipGetLastError());
    GH_CHECK(d2h(ctx, grid, b.p, cells * 16));
    return sync(ctx);
}

int gridhip_fft2_centered(gridhip_ctx *ctx, int64_t N, const double *in, double *out, int inverse)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (N <= 0 || !in || !out) return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)N * N;
    DevBuf a, b, t;
    GH_CHECK(a.alloc(ctx, cells * 16));
    GH_CHECK(b.alloc(ctx, cells * 16));
    GH_CHECK(t.alloc(ctx, cells * 16));
    GH_CHECK(h2d(ctx, a.p, in, cells * 16));
    GH_CHECK(dev_fft2c(ctx, N, a.as<double2>(), b.as<double2>(), t.as<double2>(), inverse != 0));
    GH_CHECK(d2h(ctx, out, b.p, cells * 16));
    return sync(ctx);
}

int gridhip_w_kernel(gridhip_ctx *ctx, double theta, double w, int64_t npixFF, int64_t npixKern, int64_t qpx,
                     double *out)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (npixFF <= 0 || npixKern <= 0 || qpx <= 0 || !out || npixKern > npixFF)
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t na = npixFF * qpx;
    const size_t kel = (size_t)qpx * qpx * npixKern * npixKern;
    DevBuf pad, af, tmp, k;
    GH_CHECK(pad.alloc(ctx, na * na * 16));
    GH_CHECK(af.alloc(ctx, na * na * 16));
    GH_CHECK(tmp.alloc(ctx, na * na * 16));
    GH_CHECK(k.alloc(ctx, kel * 16));
    GH_CHECK(dev_w_kernel(ctx, theta, w, npixFF, npixKern, qpx, k.as<double2>(), false, pad.as<double2>(),
                          af.as<double2>(), tmp.as<double2>()));
    GH_CHECK(d2h(ctx, out, k.p, kel * 16));
    return sync(ctx);
}

// ---- imaging functions (ImagingFunction, src/Gridding.hs:76-81): uvw in wavelengths, returns the N x N grid ----

static int stage3(gridhip_ctx *ctx, int64_t n, const double *u, const double *v, const double *w, int64_t stride,
                  const double *vis, DevBuf &du, DevBuf &dv, DevBuf &dw, DevBuf &dvis)
{
    const size_t span = n > 0 ? (size_t)(n - 1) * stride + 1 : 0;
    DevBuf s0, s1, s2;
    GH_CHECK(s0.alloc(ctx, span * 8));
    GH_CHECK(s1.alloc(ctx, span * 8));
    GH_CHECK(s2.alloc(ctx, span * 8));
    GH_CHECK(du.alloc(ctx, n * 8));
    GH_CHECK(dv.alloc(ctx, n * 8));
    GH_CHECK(dw.alloc(ctx, n * 8));
    GH_CHECK(dvis.alloc(ctx, n * 16));
    GH_CHECK(h2d(ctx, s0.p, u, span * 8));
    GH_CHECK(h2d(ctx, s1.p, v, span * 8));
    if (w) GH_CHECK(h2d(ctx, s2.p, w, span * 8));
    GH_CHECK(h2d(ctx, dvis.p, vis, n * 16));
    if (n > 0) {
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, s0.as<double>(), stride, 1.0,
                           du.as<double>());
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, s1.as<double>(), stride, 1.0,
                           dv.as<double>());
        if (w)
            hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, s2.as<double>(), stride,
                               1.0, dw.as<double>());
    }
    GH_CHECK_HIP(ctx, hipGetLastError());
    return sync(ctx);
}

static int simple_grid_dev(gridhip_ctx *ctx, int64_t lam, int64_t N, int64_t n, const double *u, const double *v,
                           const double *vis, double *grid)
{
    DevBuf pu, pv;
    GH_CHECK(pu.alloc(ctx, n * 8));
    GH_CHECK(pv.alloc(ctx, n * 8));
    if (n > 0) {
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, u, (int64_t)1, (double)lam,
                           pu.as<double>());
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, v, (int64_t)1, (double)lam,
                           pv.as<double>());
    }
    GH_CHECK(gridhip_grid_dev(ctx, N, N, grid, n, pu.as<double>(), pv.as<double>(), 1, vis));
    return sync(ctx);
}

static int conv_grid_dev(gridhip_ctx *ctx, int64_t lam, int64_t N, int64_t Q, int64_t gh, int64_t gw,
                         const double *dkv, int64_t n, const double *u, const double *v, const double *vis,
                         double *grid)
{
    DevBuf pu, pv;
    GH_CHECK(pu.alloc(ctx, n * 8));
    GH_CHECK(pv.alloc(ctx, n * 8));
    if (n > 0) {
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, u, (int64_t)1, (double)lam,
                           pu.as<double>());
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, v, (int64_t)1, (double)lam,
                           pv.as<double>());
    }
    GH_CHECK(gridhip_convgrid_dev(ctx, N, N, grid, n, Q, gh, gw, dkv, pu.as<double>(), pv.as<double>(), 1, vis));
    return sync(ctx);
}

// simple_imaging, src/Gridding.hs:84-93
int gridhip_simple_imaging(gridhip_ctx *ctx, double theta, int64_t lam, int64_t n, const double *u, const double *v,
                           int64_t uv_stride, const double *vis, double *grid)
{
    if (!ctx) return GRIDHIP_EINVAL;
    const int64_t N = haskell_round(theta * (double)lam);
    if (N <= 0 || n < 0 || uv_stride < 1 || !grid || (n > 0 && (!u || !v || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf du, dv, dw, dvis, dg;
    GH_CHECK(stage3(ctx, n, u, v, nullptr, uv_stride, vis, du, dv, dw, dvis));
    GH_CHECK(dg.alloc(ctx, (size_t)N * N * 16));
    GH_CHECK_HIP(ctx, hipMemsetAsync(dg.p, 0, (size_t)N * N * 16, ctx->stream));
    GH_CHECK(simple_grid_dev(ctx, lam, N, n, du.as<double>(), dv.as<double>(), dvis.as<double>(), dg.as<double>()));
    GH_CHECK(d2h(ctx, grid, dg.p, (size_t)N * N * 16));
    return sync(ctx);
}

// conv_imaging kv, src/Gridding.hs:115-124 ; kv is [Q][Q][gh][gw]
int gridhip_conv_imaging(gridhip_ctx *ctx, int64_t Q, int64_t gh, int64_t gw, const double *kv, double theta,
                         int64_t lam, int64_t n, const double *u, const double *v, int64_t uv_stride,
                         const double *vis, double *grid)
{
    if (!ctx) return GRIDHIP_EINVAL;
    const int64_t N = haskell_round(theta * (double)lam);
    if (N <= 0 || n < 0 || uv_stride < 1 || !grid || !kv || Q <= 0 || gh <= 0 || gw <= 0 ||
        (n > 0 && (!u || !v || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf du, dv, dw, dvis, dg, dk;
    GH_CHECK(stage3(ctx, n, u, v, nullptr, uv_stride, vis, du, dv, dw, dvis));
    GH_CHECK(dg.alloc(ctx, (size_t)N * N * 16));
    GH_CHECK(dk.alloc(ctx, (size_t)Q * Q * gh * gw * 16));
    GH_CHECK(h2d(ctx, dk.p, kv, (size_t)Q * Q * gh * gw * 16));
    GH_CHECK_HIP(ctx, hipMemsetAsync(dg.p, 0, (size_t)N * N * 16, ctx->stream));
    GH_CHECK(conv_grid_dev(ctx, lam, N, Q, gh, gw, dk.as<double>(), n, du.as<double>(), dv.as<double>(),
                           dvis.as<double>(), dg.as<double>()));
    GH_CHECK(d2h(ctx, grid, dg.p, (size_t)N * N * 16));
    return sync(ctx);
}

// w_cache_imaging, src/Gridding.hs:399-449 (wstep default 2000, :412)
int gridhip_w_cache_imaging(gridhip_ctx *ctx, int64_t wstep, int64_t qpx, int64_t npixFF, int64_t npixKern,
                            double theta, int64_t lam, int64_t n, const double *u, const double *v, const double *w,
                            int64_t uv_stride, const double *vis, double *grid)
{
    if (!ctx) return GRIDHIP_EINVAL;
    const int64_t N = haskell_round(theta * (double)lam);
    if (wstep <= 0) wstep = 2000;
    if (N <= 0 || n < 0 || uv_stride < 1 || !grid || qpx <= 0 || npixFF <= 0 || npixKern <= 0 || npixKern > npixFF ||
        (n > 0 && (!u || !v || !w || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf du, dv, dw, dvis, dg;
    GH_CHECK(stage3(ctx, n, u, v, w, uv_stride, vis, du, dv, dw, dvis));
    GH_CHECK(dg.alloc(ctx, (size_t)N * N * 16));
    GH_CHECK_HIP(ctx, hipMemsetAsync(dg.p, 0, (size_t)N * N * 16, ctx->stream));
    WCache cache;
    GH_CHECK(w_cache_grid_dev(ctx, cache, theta, lam, wstep, qpx, npixFF, npixKern, N, n, du.as<double>(), dv.as<double>(),
                              dw.as<double>(), dvis.as<double>(), dg.as<double>()));
    GH_CHECK(d2h(ctx, grid, dg.p, (size_t)N * N * 16));
    return sync(ctx);
}

// aw_imaging / aw_imagingOld, src/Gridding.hs:452-506: p = uvw/lam, wbin = findClosest wbins w,
// index = (wbin, a1, a2), then convgrid4 / convgrid3 (same grid).  wvals are the W plane w-values.
int gridhip_aw_imaging(gridhip_ctx *ctx, double theta, int64_t lam, int64_t W, int64_t Q, int64_t S, int64_t A,
                       const double *wkerns, const double *wvals, const double *akerns, int64_t n, const double *u,
                       const double *v, const double *w, int64_t uv_stride, const int64_t *a1, const int64_t *a2,
                       const double *vis, double *grid)
{
    if (!ctx) return GRIDHIP_EINVAL;
    const int64_t N = haskell_round(theta * (double)lam);
    if (N <= 0 || n < 0 || uv_stride < 1 || W <= 0 || Q <= 0 || S <= 0 || A <= 0 || !grid || !wkerns || !wvals ||
        !akerns || (n > 0 && (!u || !v || !w || !a1 || !a2 || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf du, dv, dw, dvis, dg, dpu, dpv, dwb, dws, da1, da2, dwk, dak;
    GH_CHECK(stage3(ctx, n, u, v, w, uv_stride, vis, du, dv, dw, dvis));
    const size_t wel = (size_t)W * Q * Q * S * S, ael = (size_t)A * S * S;
    GH_CHECK(dg.alloc(ctx, (size_t)N * N * 16));
    GH_CHECK(dpu.alloc(ctx, n * 8));
    GH_CHECK(dpv.alloc(ctx, n * 8));
    GH_CHECK(dwb.alloc(ctx, n * 8));
    GH_CHECK(dws.alloc(ctx, W * 8));
    GH_CHECK(da1.alloc(ctx, n * 8));
    GH_CHECK(da2.alloc(ctx, n * 8));
    GH_CHECK(dwk.alloc(ctx, wel * 16));
    GH_CHECK(dak.alloc(ctx, ael * 16));
    GH_CHECK(h2d(ctx, dws.p, wvals, W * 8));
    GH_CHECK(h2d(ctx, da1.p, a1, n * 8));
    GH_CHECK(h2d(ctx, da2.p, a2, n * 8));
    GH_CHECK(h2d(ctx, dwk.p, wkerns, wel * 16));
    GH_CHECK(h2d(ctx, dak.p, akerns, ael * 16));
    GH_CHECK_HIP(ctx, hipMemsetAsync(dg.p, 0, (size_t)N * N * 16, ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, du.as<double>(), (int64_t)1,
                           (double)lam, dpu.as<double>());
        hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, dv.as<double>(), (int64_t)1,
                           (double)lam, dpv.as<double>());
        // NB the reference searches with w in wavelengths, not w/lam (:473-474)
        hipLaunchKernelGGL(find_closest_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, W, dws.as<double>(), n,
                           dw.as<double>(), (int64_t)1, dwb.as<int64_t>());
    }
    GH_CHECK_HIP(ctx, hipGetLastError());
    GH_CHECK(gridhip_awgrid_dev(ctx, N, N, dg.as<double>(), n, W, Q, S, A, dwk.as<double>(), dak.as<double>(),
                                dpu.as<double>(), dpv.as<double>(), 1, dwb.as<int64_t>(), da1.as<int64_t>(),
                                da2.as<int64_t>(), dvis.as<double>()));
    GH_CHECK(d2h(ctx, grid, dg.p, (size_t)N * N * 16));
    return sync(ctx);
}

// do_imaging, src/Gridding.hs:509-549.  kind selects the ImagingFunction:
//   0 simple_imaging ; 1 conv_imaging kv (Q, gh, gw, kv) ; 2 w_cache_imaging (wstep, Q=qpx, npixFF, gh=npixKern)
// dev: every array argument (kv, u, v, w, vis, image, psf) is device-resident; pmax stays a host pointer.
static int do_imaging_any(gridhip_ctx *ctx, bool dev, int kind, int64_t wstep, int64_t Q, int64_t npixFF, int64_t gh,
                          int64_t gw, const double *kv, double theta, int64_t lam, int64_t n, const double *u,
                          const double *v, const double *w, int64_t uv_stride, const double *vis, double *image,
                          double *psf, double *pmax)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (n < 0 || uv_stride < 1 || (n > 0 && (!u || !v || !w || !vis))) return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (kind == 0) {
        return do_imaging_impl(ctx, theta, lam, n, u, v, w, uv_stride, vis, image, psf, pmax, dev,
                               [&](int64_t N, const double *uu, const double *vv, const double *, const double *vs,
                                   double *g) { return simple_grid_dev(ctx, lam, N, n, uu, vv, vs, g); });
    } else if (kind == 1) {
        if (!kv || Q <= 0 || gh <= 0 || gw <= 0) return fail(ctx, GRIDHIP_EINVAL, "bad kernel");
        DevBuf dk;
        const double *k = kv;
        if (!dev) {
            GH_CHECK(dk.alloc(ctx, (size_t)Q * Q * gh * gw * 16));
            GH_CHECK(h2d(ctx, dk.p, kv, (size_t)Q * Q * gh * gw * 16));
            k = dk.as<double>();
        }
        return do_imaging_impl(ctx, theta, lam, n, u, v, w, uv_stride, vis, image, psf, pmax, dev,
                               [&](int64_t N, const double *uu, const double *vv, const double *, const double *vs,
                                   double *g) { return conv_grid_dev(ctx, lam, N, Q, gh, gw, k, n, uu, vv, vs, g); });
    } else if (kind == 2) {
        if (wstep <= 0) wstep = 2000;
        if (Q <= 0 || npixFF <= 0 || gh <= 0 || gh > npixFF) return fail(ctx, GRIDHIP_EINVAL, "bad kernel options");
        WCache cache;  // built by the image pass, reused by the PSF pass
        return do_imaging_impl(ctx, theta, lam, n, u, v, w, uv_stride, vis, image, psf, pmax, dev,
                               [&](int64_t N, const double *uu, const double *vv, const double *ww, const double *vs,
                                   double *g) {
                                   return w_cache_grid_dev(ctx, cache, theta, lam, wstep, Q, npixFF, gh, N, n, uu, vv,
                                                           ww, vs, g);
                               });
    }
    return fail(ctx, GRIDHIP_EINVAL, "unknown imaging function %d", kind);
}

int gridhip_do_imaging(gridhip_ctx *ctx, int kind, int64_t wstep, int64_t Q, int64_t npixFF, int64_t gh, int64_t gw,
                       const double *kv, double theta, int64_t lam, int64_t n, const double *u, const double *v,
                       const double *w, int64_t uv_stride, const double *vis, double *image, double *psf,
                       double *pmax)
{
    return do_imaging_any(ctx, false, kind, wstep, Q, npixFF, gh, gw, kv, theta, lam, n, u, v, w, uv_stride, vis, image,
                          psf, pmax);
}

int gridhip_do_imaging_dev(gridhip_ctx *ctx, int kind, int64_t wstep, int64_t Q, int64_t npixFF, int64_t gh, int64_t gw,
                           const double *kv, double theta, int64_t lam, int64_t n, const double *u, const double *v,
                           const double *w, int64_t uv_stride, const double *vis, double *image, double *psf,
                           double *pmax)
{
    return do_imaging_any(ctx, true, kind, wstep, Q, npixFF, gh, gw, kv, theta, lam, n, u, v, w, uv_stride, vis, image,
                          psf, pmax);
}

// w_cache_imaging with device-resident uvw (wavelengths), vis and N x N grid (overwritten)
int gridhip_w_cache_imaging_dev(gridhip_ctx *ctx, int64_t wstep, int64_t qpx, int64_t npixFF, int64_t npixKern,
                                double theta, int64_t lam, int64_t n, const double *u, const double *v, const double *w,
                                int64_t uv_stride, const double *vis, double *grid)
{
    if (!ctx) return GRIDHIP_EINVAL;
    const int64_t N = haskell_round(theta * (double)lam);
    if (wstep <= 0) wstep = 2000;
    if (N <= 0 || n < 0 || uv_stride < 1 || !grid || qpx <= 0 || npixFF <= 0 || npixKern <= 0 || npixKern > npixFF ||
        (n > 0 && (!u || !v || !w || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf du, dv, dw;
    const double *pu = u, *pv = v, *pw = w;
    if (uv_stride != 1) {  // the (n, 3) matrix: slice its columns
        GH_CHECK(du.alloc(ctx, n * 8));
        GH_CHECK(dv.alloc(ctx, n * 8));
        GH_CHECK(dw.alloc(ctx, n * 8));
        if (n > 0) {
            hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, u, uv_stride, 1.0, du.as<double>());
            hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, v, uv_stride, 1.0, dv.as<double>());
            hipLaunchKernelGGL(scale_kernel, grid_for(ctx, n), dim3(256), 0, ctx->stream, n, w, uv_stride, 1.0, dw.as<double>());
        }
        pu = du.as<double>(), pv = dv.as<double>(), pw = dw.as<double>();
    }
    GH_CHECK_HIP(ctx, hipMemsetAsync(grid, 0, (size_t)N * N * 16, ctx->stream));
    WCache cache;
    return w_cache_grid_dev(ctx, cache, theta, lam, wstep, qpx, npixFF, npixKern, N, n, pu, pv, pw, vis, grid);
}

}  // extern "C"
