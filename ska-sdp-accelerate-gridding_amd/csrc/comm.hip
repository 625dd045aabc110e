// Multi-GPU: visibility-sharded gridding with one RCCL fp64 sum all-reduce of the partial grids.
//
// Gridding is linear in the visibility set, so the path shards by visibility with no data-path
// exchange: every device grids a contiguous range of the stream onto a private grid and ONE
// ncclAllReduce(ncclDouble, ncclSum, 2 * cells) over xGMI combines them (the
// reference itself is single-device: app/Main.hs:46-53 only picks a (run, runN) pair).
//
// Two forms of communicator:
//   gridhip_comm_create       one process drives `ndev` devices (ncclCommInitAll) - what a
//                             Haskell host would bind: one foreign call shards, grids and reduces
//   gridhip_comm_create_rank  one process per GPU (ncclCommInitRank), the id travelling by whatever
//                             the host has (torch.distributed, MPI, a file)
// RCCL is loaded on first use (librccl.so.1, the copy already mapped into the process if there is
// one, e.g. PyTorch's), so libgridhip.so itself loads on machines without it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

using namespace gridhip;

struct gridhip_comm {
    int nranks = 0;                  // devices in the communicator
    int rank0 = 0;                   // communicator rank of this process's first device
    bool own_ctx = false;            // single-process form: the contexts belong to the communicator
    std::vector<gridhip_ctx *> ctx;  // this process's devices
    std::vector<ncclComm_t> comms;
    // where device i's collectives are enqueued: its context's stream (the gridding stream: ordered after the
    // gridding calls, no overlap) unless the host has set another one (gridhip_comm_set_stream: a side stream whose
    // collective runs beside the next step's gridding - the host orders the two with events)
    std::vector<hipStream_t> cstream;
    std::vector<char> has_cstream;
    int collective = 0;              // 0 = ncclAllReduce, 1 = ncclReduceScatter + ncclAllGather (option "collective")
    std::string err;
    hipStream_t stream_of(size_t i) const { return has_cstream[i] ? cstream[i] : ctx[i]->stream; }
};

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char *(*GetErrorString)(ncclResult_t);
} g_rccl;

std::string g_comm_err;  // failures before a communicator exists (written under g_err_mu)
std::mutex g_err_mu;
std::once_flag g_rccl_once;

bool load_rccl_once()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        g_comm_err = std::string("cannot load librccl.so: ") + dlerror();
        return false;
    }
#define GH_SYM(field, name)                                      \
    *(void **)(&g_rccl.field) = dlsym(h, name);                  \
    if (!g_rccl.field) {                                         \
        g_comm_err = std::string("librccl.so lacks ") + name;    \
        dlclose(h);                                              \
        return false;                                            \
    }
    GH_SYM(GetUniqueId, "ncclGetUniqueId")
    GH_SYM(CommInitRank, "ncclCommInitRank")
    GH_SYM(CommInitAll, "ncclCommInitAll")
    GH_SYM(CommDestroy, "ncclCommDestroy")
    GH_SYM(AllReduce, "ncclAllReduce")
    GH_SYM(ReduceScatter, "ncclReduceScatter")
    GH_SYM(AllGather, "ncclAllGather")
    GH_SYM(GroupStart, "ncclGroupStart")
    GH_SYM(GroupEnd, "ncclGroupEnd")
    GH_SYM(GetErrorString, "ncclGetErrorString")
#undef GH_SYM
    g_rccl.h = h;
    return true;
}

// (several host threads may create their communicators at once: the library is loaded by exactly one of them)
bool load_rccl()
{
    std::call_once(g_rccl_once, [] {
        std::lock_guard<std::mutex> lk(g_err_mu);
        (void)load_rccl_once();
    });
    return g_rccl.h != nullptr;
}

int comm_fail(gridhip_comm *c, int code, const std::string &msg)
{
    if (c)
        c->err = msg;
    else {
        std::lock_guard<std::mutex> lk(g_err_mu);
        g_comm_err = msg;
    }
    return code;
}

#define GH_NCCL(c, call)                                                                                 \
    do {                                                                                                 \
        ncclResult_t r__ = (call);                                                                       \
        if (r__ != ncclSuccess)                                                                          \
            return comm_fail((c), GRIDHIP_EHIP, std::string(#call " failed: ") + g_rccl.GetErrorString(r__)); \
    } while (0)

}  // namespace

extern "C" {

const char *gridhip_comm_last_error(const gridhip_comm *comm) { return comm ? comm->err.c_str() : g_comm_err.c_str(); }

int gridhip_comm_create(int ndev, const int *dev_ids, gridhip_comm **out)
{
    if (!out) return GRIDHIP_EINVAL;
    *out = nullptr;
    if (ndev < 1 || ndev > 64) return comm_fail(nullptr, GRIDHIP_EINVAL, "ndev must be in 1..64");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return comm_fail(nullptr, GRIDHIP_ENODEV, "no HIP device");
    std::vector<int> ids(ndev);
    for (int i = 0; i < ndev; ++i) {
        ids[i] = dev_ids ? dev_ids[i] : i;
        if (ids[i] < 0 || ids[i] >= have) return comm_fail(nullptr, GRIDHIP_EINVAL, "device id out of range");
        for (int j = 0; j < i; ++j)
            if (ids[j] == ids[i]) return comm_fail(nullptr, GRIDHIP_EINVAL, "device listed twice");
    }
    if (!load_rccl()) return GRIDHIP_EHIP;
    gridhip_comm *c = new (std::nothrow) gridhip_comm();
    if (!c) return GRIDHIP_ENOMEM;
    c->nranks = ndev;
    c->own_ctx = true;
    for (int i = 0; i < ndev; ++i) {
        gridhip_ctx *x = nullptr;
        const int rc = gridhip_create(ids[i], &x);
        if (rc != GRIDHIP_OK) {
            gridhip_comm_destroy(c);
            return comm_fail(nullptr, rc, "gridhip_create failed for a device of the communicator");
        }
        c->ctx.push_back(x);
    }
    c->comms.assign(ndev, nullptr);
    c->cstream.assign(ndev, nullptr);
    c->has_cstream.assign(ndev, 0);
    const ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), ndev, ids.data());
    if (r != ncclSuccess) {
        c->comms.clear();
        gridhip_comm_destroy(c);
        return comm_fail(nullptr, GRIDHIP_EHIP, std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(r));
    }
    *out = c;
    return GRIDHIP_OK;
}

int gridhip_comm_unique_id(void *id128)
{
    if (!id128) return GRIDHIP_EINVAL;
    if (!load_rccl()) return GRIDHIP_EHIP;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI promises a 128-byte id");
    ncclUniqueId id;
    GH_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return GRIDHIP_OK;
}

int gridhip_comm_create_rank(gridhip_ctx *ctx, int nranks, int rank, const void *id128, gridhip_comm **out)
{
    if (!out) return GRIDHIP_EINVAL;
    *out = nullptr;
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
        return comm_fail(nullptr, GRIDHIP_EINVAL, "bad rank / nranks / id");
    if (!load_rccl()) return GRIDHIP_EHIP;
    if (hipSetDevice(ctx->device) != hipSuccess) return comm_fail(nullptr, GRIDHIP_EHIP, "hipSetDevice failed");
    gridhip_comm *c = new (std::nothrow) gridhip_comm();
    if (!c) return GRIDHIP_ENOMEM;
    c->nranks = nranks;
    c->rank0 = rank;
    c->ctx.push_back(ctx);
    c->cstream.assign(1, nullptr);
    c->has_cstream.assign(1, 0);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t nc = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&nc, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return comm_fail(nullptr, GRIDHIP_EHIP, std::string("ncclCommInitRank failed: ") + g_rccl.GetErrorString(r));
    }
    c->comms.push_back(nc);
    *out = c;
    return GRIDHIP_OK;
}

int gridhip_comm_destroy(gridhip_comm *c)
{
    if (!c) return GRIDHIP_OK;
    for (size_t i = 0; i < c->comms.size(); ++i) {
        if (i < c->ctx.size() && c->ctx[i]) {
            (void)hipSetDevice(c->ctx[i]->device);
            (void)hipStreamSynchronize(c->ctx[i]->stream);
            if (i < c->has_cstream.size() && c->has_cstream[i]) (void)hipStreamSynchronize(c->cstream[i]);
        }
        if (c->comms[i]) (void)g_rccl.CommDestroy(c->comms[i]);
    }
    if (c->own_ctx)
        for (gridhip_ctx *x : c->ctx) gridhip_destroy(x);
    delete c;
    return GRIDHIP_OK;
}

int gridhip_comm_ndev(const gridhip_comm *c) { return c ? (int)c->ctx.size() : 0; }
int gridhip_comm_nranks(const gridhip_comm *c) { return c ? c->nranks : 0; }
gridhip_ctx *gridhip_comm_ctx(gridhip_comm *c, int i) { return (c && i >= 0 && i < (int)c->ctx.size()) ? c->ctx[i] : nullptr; }

// In-place sum over the communicator of `count` doubles at bufs[i] on this process's i-th device, enqueued on that
// device's collective stream.  collective = 0: one ncclAllReduce (RCCL picks ring / tree).  collective = 1:
// ncclReduceScatter + ncclAllGather, both in place - rank r owns doubles [r * chunk, (r + 1) * chunk) - which on
// xGMI's point-to-point links is the direct schedule: every GPU exchanges one chunk with each of its 7 peers at once
// instead of passing 2 (p - 1) / p of the buffer round a ring; what `count` leaves over after the
// nranks equal chunks goes through a small all-reduce in the first group.
static int comm_sum_doubles(gridhip_comm *c, size_t count, double *const *bufs)
{
    const size_t nd = c->ctx.size();
    if (count == 0) return GRIDHIP_OK;
    if (c->collective == 0 || count < (size_t)c->nranks) {
        GH_NCCL(c, g_rccl.GroupStart());
        for (size_t i = 0; i < nd; ++i) {
            (void)hipSetDevice(c->ctx[i]->device);  // (one thread drives several devices: each call is made on its own)
            const ncclResult_t r = g_rccl.AllReduce(bufs[i], bufs[i], count, ncclDouble, ncclSum, c->comms[i], c->stream_of(i));
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                return comm_fail(c, GRIDHIP_EHIP, std::string("ncclAllReduce failed: ") + g_rccl.GetErrorString(r));
            }
        }
        GH_NCCL(c, g_rccl.GroupEnd());
        return GRIDHIP_OK;
    }
    const size_t chunk = count / (size_t)c->nranks, rem = count - chunk * (size_t)c->nranks;
    GH_NCCL(c, g_rccl.GroupStart());
    for (size_t i = 0; i < nd; ++i) {
        (void)hipSetDevice(c->ctx[i]->device);
        double *p = bufs[i];
        const size_t r = (size_t)c->rank0 + i;
        ncclResult_t e = g_rccl.ReduceScatter(p, p + r * chunk, chunk, ncclDouble, ncclSum, c->comms[i], c->stream_of(i));
        if (e == ncclSuccess && rem)
            e = g_rccl.AllReduce(p + chunk * c->nranks, p + chunk * c->nranks, rem, ncclDouble, ncclSum, c->comms[i], c->stream_of(i));
        if (e != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return comm_fail(c, GRIDHIP_EHIP, std::string("ncclReduceScatter failed: ") + g_rccl.GetErrorString(e));
        }
    }
    GH_NCCL(c, g_rccl.GroupEnd());
    GH_NCCL(c, g_rccl.GroupStart());
    for (size_t i = 0; i < nd; ++i) {
        (void)hipSetDevice(c->ctx[i]->device);
        double *p = bufs[i];
        const size_t r = (size_t)c->rank0 + i;
        const ncclResult_t e = g_rccl.AllGather(p + r * chunk, p, chunk, ncclDouble, c->comms[i], c->stream_of(i));
        if (e != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return comm_fail(c, GRIDHIP_EHIP, std::string("ncclAllGather failed: ") + g_rccl.GetErrorString(e));
        }
    }
    GH_NCCL(c, g_rccl.GroupEnd());
    return GRIDHIP_OK;
}

// grids[i]: device pointer on this process's i-th device, `cells` complex cells; summed in place over
// all devices of the communicator.  Enqueued on each device's collective stream (the context's own stream unless
// gridhip_comm_set_stream chose another), asynchronous to the host.
int gridhip_comm_allreduce_grids(gridhip_comm *c, int64_t cells, double *const *grids)
{
    if (!c || !grids || cells < 0) return GRIDHIP_EINVAL;
    for (size_t i = 0; i < c->ctx.size(); ++i)
        if (!grids[i] && cells > 0) return comm_fail(c, GRIDHIP_EINVAL, "null grid pointer");
    return comm_sum_doubles(c, (size_t)cells * 2, grids);
}

int gridhip_comm_allreduce_grid(gridhip_comm *c, int64_t cells, double *grid)
{
    if (!c || c->ctx.size() != 1) return comm_fail(c, GRIDHIP_EINVAL, "rank form only: this communicator drives several devices");
    double *g[1] = {grid};
    return gridhip_comm_allreduce_grids(c, cells, g);
}

// Rows [y0, y1) of grids of Wd columns only.  A stream that went through mirror_uvw (src/Gridding.hs:551-562:
// v >= 0) leaves every row below H / 2 - gh / 2 - 1 of every partial grid exactly zero, and summing zeros over xGMI
// is half of the collective's bytes: the caller that knows its stream is mirrored reduces from that row on.
int gridhip_comm_allreduce_rows(gridhip_comm *c, int64_t Wd, int64_t y0, int64_t y1, double *const *grids)
{
    if (!c || !grids) return GRIDHIP_EINVAL;
    if (Wd <= 0 || y0 < 0 || y1 < y0) return comm_fail(c, GRIDHIP_EINVAL, "bad row range");
    std::vector<double *> p(c->ctx.size());
    for (size_t i = 0; i < c->ctx.size(); ++i) {
        if (!grids[i] && y1 > y0) return comm_fail(c, GRIDHIP_EINVAL, "null grid pointer");
        p[i] = grids[i] + 2 * (size_t)y0 * (size_t)Wd;
    }
    return comm_sum_doubles(c, (size_t)(y1 - y0) * (size_t)Wd * 2, p.data());
}

int gridhip_comm_allreduce_grid_rows(gridhip_comm *c, int64_t Wd, int64_t y0, int64_t y1, double *grid)
{
    if (!c || c->ctx.size() != 1) return comm_fail(c, GRIDHIP_EINVAL, "rank form only: this communicator drives several devices");
    double *g[1] = {grid};
    return gridhip_comm_allreduce_rows(c, Wd, y0, y1, g);
}

int gridhip_comm_set_option(gridhip_comm *c, const char *key, int64_t value)
{
    if (!c || !key) return GRIDHIP_EINVAL;
    if (!strcmp(key, "collective")) {
        if (value != 0 && value != 1) return comm_fail(c, GRIDHIP_EINVAL, "collective: 0 = all-reduce, 1 = reduce-scatter + all-gather");
        c->collective = (int)value;
        return GRIDHIP_OK;
    }
    return comm_fail(c, GRIDHIP_EINVAL, std::string("unknown communicator option '") + key + "'");
}

int gridhip_comm_get_option(gridhip_comm *c, const char *key, int64_t *value)
{
    if (!c || !key || !value) return GRIDHIP_EINVAL;
    if (!strcmp(key, "collective")) {
        *value = c->collective;
        return GRIDHIP_OK;
    }
    return comm_fail(c, GRIDHIP_EINVAL, std::string("unknown communicator option '") + key + "'");
}

int gridhip_comm_set_stream(gridhip_comm *c, int i, void *hip_stream)
{
    if (!c || i < 0 || i >= (int)c->ctx.size()) return GRIDHIP_EINVAL;
    c->cstream[i] = (hipStream_t)hip_stream;
    c->has_cstream[i] = 1;
    return GRIDHIP_OK;
}

int gridhip_comm_reset_stream(gridhip_comm *c, int i)
{
    if (!c || i < 0 || i >= (int)c->ctx.size()) return GRIDHIP_EINVAL;
    c->has_cstream[i] = 0;
    return GRIDHIP_OK;
}

// convgrid2 (src/Gridding.hs:199-244) over the communicator's devices, host pointers, synchronous: the drop-in
// form.  Single-process form: the n visibilities are cut into contiguous shards, one per device (device 0 also
// receives the incoming grid, which is accumulated into as permute (+) does); one host thread per device stages
// and grids its shard, the partial grids are all-reduced and device 0's copy is returned.  Rank form: every
// process passes ITS shard; the incoming grid should be non-zero on one rank only (it is summed over ranks).
int gridhip_comm_convgrid2(gridhip_comm *c, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W, int64_t Q,
                           int64_t gh, int64_t gw, const double *gcf, const double *u, const double *v,
                           int64_t uv_stride, const int64_t *wbin, const double *vis)
{
    if (!c) return GRIDHIP_EINVAL;
    if (H <= 0 || Wd <= 0 || n < 0 || uv_stride < 1 || W <= 0 || Q <= 0 || gh <= 0 || gw <= 0)
        return comm_fail(c, GRIDHIP_EINVAL, "bad size");
    if (!grid || !gcf || (n > 0 && (!u || !v || !vis))) return comm_fail(c, GRIDHIP_EINVAL, "null pointer");
    const int nd = (int)c->ctx.size();
    const size_t cells = (size_t)H * Wd, kel = (size_t)W * Q * Q * gh * gw;
    std::vector<double *> dgrid(nd, nullptr);
    std::vector<int> rcs(nd, GRIDHIP_OK);
    auto work = [&](int i) {
        gridhip_ctx *x = c->ctx[i];
        // contiguous, balanced shard of the stream (the rank form grids everything it was given)
        const int64_t base = n / nd, rem = n % nd;
        const int64_t lo = i * base + (i < rem ? i : rem), cnt = base + (i < rem ? 1 : 0);
        auto run = [&]() -> int {
            GH_CHECK_HIP(x, hipSetDevice(x->device));
            const size_t span = cnt > 0 ? (size_t)(cnt - 1) * uv_stride + 1 : 1;
            auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
            GH_CHECK(ws_reserve(x, x->stage, al(cells * 16) + 2 * al(span * 8) + al((size_t)cnt * 16 + 16) +
                                                 al((size_t)cnt * 8 + 8) + al(kel * 16)));
            char *p = (char *)x->stage.ptr;
            double *dg = (double *)p;
            p += al(cells * 16);
            double *du = (double *)p;
            p += al(span * 8);
            double *dv = (double *)p;
            p += al(span * 8);
            double *dvis = (double *)p;
            p += al((size_t)cnt * 16 + 16);
            int64_t *dwb = (int64_t *)p;
            p += al((size_t)cnt * 8 + 8);
            double *dk = (double *)p;
            dgrid[i] = dg;
            if (i == 0)
                GH_CHECK_HIP(x, hipMemcpyAsync(dg, grid, cells * 16, hipMemcpyHostToDevice, x->stream));
            else
                GH_CHECK_HIP(x, hipMemsetAsync(dg, 0, cells * 16, x->stream));
            GH_CHECK_HIP(x, hipMemcpyAsync(dk, gcf, kel * 16, hipMemcpyHostToDevice, x->stream));
            if (cnt > 0) {
                GH_CHECK_HIP(x, hipMemcpyAsync(du, u + lo * uv_stride, span * 8, hipMemcpyHostToDevice, x->stream));
                GH_CHECK_HIP(x, hipMemcpyAsync(dv, v + lo * uv_stride, span * 8, hipMemcpyHostToDevice, x->stream));
                GH_CHECK_HIP(x, hipMemcpyAsync(dvis, vis + 2 * lo, (size_t)cnt * 16, hipMemcpyHostToDevice, x->stream));
                if (wbin)
                    GH_CHECK_HIP(x, hipMemcpyAsync(dwb, wbin + lo, (size_t)cnt * 8, hipMemcpyHostToDevice, x->stream));
            }
            GH_CHECK(gridhip_convgrid2_dev(x, H, Wd, dg, cnt, W, Q, gh, gw, dk, du, dv, uv_stride, wbin ? dwb : nullptr,
                                           dvis));
            return GRIDHIP_OK;
        };
        rcs[i] = run();
    };
    if (nd == 1)
        work(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nd; ++i) th.emplace_back(work, i);
        for (auto &t : th) t.join();
    }
    // A shard that failed locally must not leave the other devices (or, in the rank form, the other PROCESSES) waiting
    // in the collective for ever: every device still takes part in the all-reduce - with whatever its grid buffer
    // holds - and the failure is reported afterwards; the caller's grid is then left untouched.
    int first_bad = -1;
    for (int i = 0; i < nd; ++i)
        if (rcs[i] != GRIDHIP_OK && first_bad < 0) first_bad = i;
    bool staged = true;
    for (int i = 0; i < nd; ++i) staged = staged && dgrid[i] != nullptr;
    // (the synchronous form reduces on the gridding streams, whatever gridhip_comm_set_stream chose for the async one)
    const std::vector<char> keep = c->has_cstream;
    c->has_cstream.assign(nd, 0);
    int rc_coll = GRIDHIP_OK;
    if (c->nranks > 1 && staged) rc_coll = gridhip_comm_allreduce_grids(c, (int64_t)cells, dgrid.data());
    c->has_cstream = keep;
    if (first_bad >= 0) {
        for (int i = 0; i < nd; ++i) {
            (void)hipSetDevice(c->ctx[i]->device);
            (void)hipStreamSynchronize(c->ctx[i]->stream);
        }
        return comm_fail(c, rcs[first_bad], std::string("device shard failed: ") + c->ctx[first_bad]->err);
    }
    GH_CHECK(rc_coll);
    // internal consistency failures of any shard (records that did not fit, slices outside the table: "errors") are
    // reported instead of a silently incomplete grid, as gridhip_convgrid2 does
    for (int i = 0; i < nd; ++i) {
        gridhip_ctx *x = c->ctx[i];
        GH_CHECK_HIP(x, hipSetDevice(x->device));
        int32_t e = 0;
        GH_CHECK_HIP(x, hipMemcpyAsync(&e, x->d_scalars + 2, sizeof e, hipMemcpyDeviceToHost, x->stream));
        GH_CHECK_HIP(x, hipStreamSynchronize(x->stream));
        if (e) return comm_fail(c, GRIDHIP_EINVAL, "internal consistency check failed on device " + std::to_string(x->device) +
                                                       " for " + std::to_string(e) + " records (inputs modified during the call?)");
    }
    gridhip_ctx *x0 = c->ctx[0];
    GH_CHECK_HIP(x0, hipSetDevice(x0->device));
    GH_CHECK_HIP(x0, hipMemcpyAsync(grid, dgrid[0], cells * 16, hipMemcpyDeviceToHost, x0->stream));
    GH_CHECK_HIP(x0, hipStreamSynchronize(x0->stream));
    return GRIDHIP_OK;
}

}  // extern "C"
