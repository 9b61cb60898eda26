// Multi-GPU reductions of the frame-sharded coadd on RCCL, inside libzudsmi (SURVEY.md section 8(b):
// zm_comm_init / zm_coadd_reduce): one process per GPU, the collectives enqueued on the context's
// stream behind the kernels that produce the partial products.
//
//   zm_coadd_reduce_dev   S1 = sum(w v), S0 = sum(w): the two planes of ONE buffer, one ncclAllReduce
//                         (a reduce-scatter and an all-gather over the 7 xGMI links of a GPU)
//   zm_mask_reduce_dev    RCCL has no bitwise reductions: rank g receives row band g of every rank's
//                         partial mask (grouped ncclSend / ncclRecv, rows of a C-contiguous plane sent
//                         in place), folds it (k_mask_accum), the folded bands are all-gathered:
//                         2 x 4 B / px per rank whatever the world size
//
// The reference shards by job (zuds/mpi.py:36-64) and has no collective; this is the exchange step
// BASELINE config 4 adds.  librccl is opened on first use (dlopen): a single-GPU installation
// needs no RCCL, and libzudsmi.so carries no load-time dependency on it.  The same call pattern
// through torch.distributed (backend "nccl" IS RCCL) is parallel.py; ZM_NATIVE_RCCL=1 selects this.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>

#include "zm_internal.h"

namespace {
struct rccl_api {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
rccl_api R;
std::mutex R_lock;

int rccl_load() {
    std::lock_guard<std::mutex> g(R_lock);
    if (R.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    ZM_CHECK(h != nullptr, "zm_comm: librccl.so not found (%s)", dlerror());
#define ZM_SYM(field, sym)                                                    \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(h, #sym));            \
    if (!R.field) { zm_set_error("zm_comm: %s missing from librccl", #sym); dlclose(h); return 2; }
    ZM_SYM(GetUniqueId, ncclGetUniqueId)
    ZM_SYM(CommInitRank, ncclCommInitRank)
    ZM_SYM(CommDestroy, ncclCommDestroy)
    ZM_SYM(AllReduce, ncclAllReduce)
    ZM_SYM(AllGather, ncclAllGather)
    ZM_SYM(Send, ncclSend)
    ZM_SYM(Recv, ncclRecv)
    ZM_SYM(GroupStart, ncclGroupStart)
    ZM_SYM(GroupEnd, ncclGroupEnd)
    ZM_SYM(GetErrorString, ncclGetErrorString)
#undef ZM_SYM
    R.lib = h;
    return 0;
}
}  // namespace

#define ZM_NCCL(call)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            zm_set_error("%s failed: %s (%s:%d)", #call, R.GetErrorString(r_), __FILE__, __LINE__); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

struct zm_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    bool broken = false;   // a call failed inside a group: peers hold unmatched calls, nothing more may be queued
};

// Inside ncclGroupStart / ncclGroupEnd a plain `return` would leave this thread's group open: every
// later collective on the communicator is then queued and never launched.  Close the group, mark
// the communicator unusable, then report.
#define ZM_NCCL_IN_GROUP(comm_, call)                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            (void)R.GroupEnd();                                                                \
            (comm_)->broken = true;                                                            \
            zm_set_error("%s failed inside a group: %s (%s:%d); communicator unusable", #call, \
                         R.GetErrorString(r_), __FILE__, __LINE__);                            \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)
#define ZM_COMM_USABLE(comm_, who) \
    ZM_CHECK(!(comm_)->broken, who ": communicator marked unusable by an earlier failure; zm_comm_destroy it")

extern "C" int zm_comm_unique_id(void* id128) {
    ZM_CHECK(id128 != nullptr, "zm_comm_unique_id: null argument");
    ZM_TRY(rccl_load());
    static_assert(sizeof(ncclUniqueId) == ZM_COMM_ID_BYTES, "ncclUniqueId is not 128 bytes");
    ncclUniqueId id;
    ZM_NCCL(R.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int zm_comm_init(zm_ctx* ctx, int nranks, int rank, const void* id128, zm_comm** out) {
    ZM_CHECK(ctx && id128 && out, "zm_comm_init: null argument");
    ZM_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "zm_comm_init: rank %d of %d", rank, nranks);
    // (the banded mask reduction runs the fixed-size plan of zm_comm_mask_plan - refuse here, with the
    // reason, a communicator whose mask reduction would fail later; 64 ranks are eight 8-GPU nodes)
    ZM_CHECK(nranks <= ZM_COMM_MAX_RANKS, "zm_comm_init: %d ranks, this library plans its band exchanges for at most %d "
             "(ZM_COMM_MAX_RANKS, include/zudsmi.h)", nranks, ZM_COMM_MAX_RANKS);
    ZM_TRY(rccl_load());
    ZM_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    zm_comm* c = new zm_comm;
    c->nranks = nranks;
    c->rank = rank;
    ncclResult_t r = R.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        zm_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, R.GetErrorString(r));
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

extern "C" int zm_comm_destroy(zm_comm* comm) {
    if (!comm) return 0;
    if (comm->comm && R.CommDestroy) (void)R.CommDestroy(comm->comm);
    delete comm;
    return 0;
}

extern "C" int zm_coadd_reduce_dev(zm_ctx* ctx, zm_comm* comm, float* s1s0, int64_t npix) {
    ZM_CHECK(ctx && comm && s1s0 && npix > 0, "zm_coadd_reduce_dev: bad argument");
    ZM_COMM_USABLE(comm, "zm_coadd_reduce_dev");
    ZM_HIP(hipSetDevice(ctx->device));
    zm_scope_timer t(ctx, "rccl_planes");
    ZM_NCCL(R.AllReduce(s1s0, s1s0, (size_t)(2 * npix), ncclFloat, ncclSum, comm->comm, ctx->stream));
    return 0;
}

// rows [b[g], b[g + 1]) of the plane belong to rank g (np.array_split, as parallel.band_bounds)
static void band_bounds(int nrows, int world, std::vector<int>* b) {
    b->assign(1, 0);
    const int base = nrows / world, extra = nrows % world;
    for (int r = 0; r < world; ++r) b->push_back(b->back() + base + (r < extra ? 1 : 0));
}

extern "C" int zm_comm_band_bounds(int nrows, int world, int32_t* bounds) {
    ZM_CHECK(bounds && nrows >= 0 && world >= 1, "zm_comm_band_bounds: bad argument");
    std::vector<int> b;
    band_bounds(nrows, world, &b);
    for (int g = 0; g <= world; ++g) bounds[g] = b[g];
    return 0;
}

// The banded schedule of zm_mask_reduce_dev for one rank, as element offsets / counts (host arithmetic only;
// the send / recv / gather calls below are issued from exactly these numbers, and tests replay the plans of
// all ranks on the CPU: what rank r sends to g is what g expects from r, slots do not overlap).
extern "C" int zm_comm_mask_plan(int nx, int ny, int world, int rank, zm_mask_plan* plan) {
    ZM_CHECK(plan && nx > 0 && ny > 0 && world >= 1 && world <= ZM_COMM_MAX_RANKS && rank >= 0 && rank < world,
             "zm_comm_mask_plan: bad argument (rank %d of %d, at most %d ranks)", rank, world, ZM_COMM_MAX_RANKS);
    std::vector<int> b;
    band_bounds(ny, world, &b);
    int maxr = 0;
    for (int g = 0; g < world; ++g) maxr = std::max(maxr, b[g + 1] - b[g]);
    memset(plan, 0, sizeof(*plan));
    plan->world = world;
    plan->rank = rank;
    plan->band_px = (int64_t)maxr * nx;
    plan->my_px = (int64_t)(b[rank + 1] - b[rank]) * nx;
    for (int g = 0; g < world; ++g) {
        plan->send_off[g] = (int64_t)b[g] * nx;                   // into this rank's mask plane
        plan->send_cnt[g] = (int64_t)(b[g + 1] - b[g]) * nx;       // (g == rank: the local copy)
        plan->recv_off[g] = (int64_t)g * plan->band_px;            // into the receive buffer: slot of rank g
        plan->recv_cnt[g] = plan->my_px;
        plan->gather_off[g] = (int64_t)g * plan->band_px;          // folded band of rank g in the gather buffer
    }
    return 0;
}

extern "C" int zm_mask_reduce_dev(zm_ctx* ctx, zm_comm* comm, int32_t* mask, int nx, int ny, int kind,
                                  float* cov) {
    ZM_CHECK(ctx && comm && mask && nx > 0 && ny > 0, "zm_mask_reduce_dev: bad argument");
    ZM_CHECK(kind == ZM_MASK_AND || kind == ZM_MASK_OR, "zm_mask_reduce_dev: unknown mask combine %d", kind);
    ZM_COMM_USABLE(comm, "zm_mask_reduce_dev");
    ZM_HIP(hipSetDevice(ctx->device));
    const int world = comm->nranks, rank = comm->rank;
    const int64_t npix = (int64_t)nx * ny;
    if (world == 1) return zm_launch_mask_finalize(ctx, mask, cov, npix);
    zm_mask_plan P;
    ZM_TRY(zm_comm_mask_plan(nx, ny, world, rank, &P));
    const size_t bandpx = (size_t)P.band_px;
    int32_t *recv = nullptr, *folded = nullptr, *gathered = nullptr;
    ZM_TRY(ctx->get("comm_recv", sizeof(int32_t) * bandpx * world, (void**)&recv));
    ZM_TRY(ctx->get("comm_fold", sizeof(int32_t) * bandpx, (void**)&folded));
    ZM_TRY(ctx->get("comm_gather", sizeof(int32_t) * bandpx * world, (void**)&gathered));
    zm_scope_timer t(ctx, "rccl_masks");
    hipStream_t st = ctx->stream;
    ZM_NCCL(R.GroupStart());
    for (int g = 0; g < world; ++g) {
        if (g == rank) continue;
        if (P.send_cnt[g])
            ZM_NCCL_IN_GROUP(comm, R.Send(mask + P.send_off[g], (size_t)P.send_cnt[g], ncclInt32, g, comm->comm, st));
        if (P.my_px)
            ZM_NCCL_IN_GROUP(comm, R.Recv(recv + P.recv_off[g], (size_t)P.recv_cnt[g], ncclInt32, g, comm->comm, st));
    }
    {
        ncclResult_t r_ = R.GroupEnd();
        if (r_ != ncclSuccess) {
            comm->broken = true;
            zm_set_error("ncclGroupEnd failed: %s; communicator unusable", R.GetErrorString(r_));
            return 1;
        }
    }
    if (P.my_px) {
        ZM_HIP(hipMemcpyAsync(recv + P.recv_off[rank], mask + P.send_off[rank], sizeof(int32_t) * (size_t)P.my_px,
                              hipMemcpyDeviceToDevice, st));
        for (int g = 0; g < world; ++g)
            ZM_TRY(zm_launch_mask_accum(ctx, folded, recv + P.recv_off[g], P.my_px, kind, g == 0));
    }
    // bands may differ by one row: gathered through slots of the largest
    ZM_NCCL(R.AllGather(folded, gathered, bandpx, ncclInt32, comm->comm, st));
    for (int g = 0; g < world; ++g) {
        if (P.send_cnt[g])
            ZM_HIP(hipMemcpyAsync(mask + P.send_off[g], gathered + P.gather_off[g], sizeof(int32_t) * (size_t)P.send_cnt[g],
                                  hipMemcpyDeviceToDevice, st));
    }
    return zm_launch_mask_finalize(ctx, mask, cov, npix);
}
