// halo.hip -- ghost-row exchange of a row-block decomposition behind the C ABI (SURVEY.md section 8(b)
// "nsdg_halo_exchange", section 8(e)): nearest-neighbour send/recv of contiguous row blocks, no collective.
//
// The reference has no parallelism at all; the seam this sits behind is the batched model
// step, IModelStep::iterate (core/src/include/IModelStep.hpp:16-34): a multi-rank step implementation owns one
// context per GPU and exchanges ghost rows through the plans below between kernel launches.
//
// One exchange = pack kernel -> transport -> unpack kernel, all on the context's COMMUNICATION stream, ordered
// against the compute stream by events only (the host never waits):
//     nsdg_halo_start : comm stream waits for what the compute stream has enqueued so far; everything that
//                       travels to one neighbour is gathered into one buffer (one launch for both directions);
//                       the transport is posted
//     nsdg_halo_finish: the received buffers are scattered into the ghost rows (one launch); the compute
//                       stream waits for that
// Kernels launched on the compute stream between start and finish overlap with the exchange.
//
// Two transports share everything but the middle step:
//   * RCCL (one process per GPU): ncclGroupStart; ncclSend x<=2; ncclRecv x<=2; ncclGroupEnd on the comm stream.
//     librccl.so.1 is resolved with dlopen at nsdg_comm_init, so the library loads and every other entry point
//     works on a machine without RCCL, and a Python caller shares the RCCL instance torch has already mapped.
//   * local (all ranks are threads of ONE process, on one or several devices of it): the receiver copies the
//     sender's packed buffer device-to-device after waiting on the sender's event; host-side hand-shake through
//     a mailbox.  This is what the one-GPU tests run the multi-rank driver on, and a single-process multi-GPU mode.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include "nsdg_internal.h"

// ---------------------------------------------------------------------------------------------- RCCL by dlopen
namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle)
        return NSDG_OK;
    void* h = nullptr;
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h)
            break;
    }
    if (!h) {
        nsdg_set_error("nsdg_comm: librccl.so.1 not found (%s)", dlerror());
        return NSDG_ERR_COMM;
    }
    RcclApi a;
    a.handle = h;
#define NSDG_RCCL_SYM(field, sym)                                          \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym));          \
    if (!a.field) {                                                        \
        nsdg_set_error("nsdg_comm: librccl lacks the symbol %s", sym);     \
        return NSDG_ERR_COMM;                                              \
    }
    NSDG_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    NSDG_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    NSDG_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    NSDG_RCCL_SYM(CommAbort, "ncclCommAbort")
    NSDG_RCCL_SYM(GroupStart, "ncclGroupStart")
    NSDG_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    NSDG_RCCL_SYM(Send, "ncclSend")
    NSDG_RCCL_SYM(Recv, "ncclRecv")
    NSDG_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef NSDG_RCCL_SYM
    g_rccl = a;
    return NSDG_OK;
}

#define NSDG_CHECK_RCCL(expr)                                                                       \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            nsdg_set_error("%s: %s failed: %s", __func__, #expr, g_rccl.GetErrorString(r_));        \
            return NSDG_ERR_COMM;                                                                   \
        }                                                                                           \
    } while (0)

// ---------------------------------------------------------------------------------------------- local transport
// All ranks of a local group are threads of this process.  A message is the sender's packed buffer plus the event
// recorded after its pack kernel; the receiver enqueues a device-to-device copy behind that event and answers
// with an event recorded after the copy (the sender waits for it before it packs into the buffer again).
// Messages and acknowledgements are matched by (source, destination, plan index): every rank creates its plans
// in the same order (the drivers are SPMD), exactly as communicators must be created in the same order.
struct LocalMsg {
    const double* buf;
    int64_t count;
    hipEvent_t ready;
};

struct LocalGroup {
    int world = 0;
    int refs = 0;
    std::mutex m;
    std::condition_variable cv;
    bool failed = false;
    std::map<std::tuple<int, int, int>, std::deque<LocalMsg>> box; // (src, dst, plan)
    std::map<std::tuple<int, int, int>, std::deque<hipEvent_t>> ack; // (src, dst, plan): dst has consumed src's buffer
};

std::mutex g_groups_mutex;
std::map<int64_t, LocalGroup*> g_groups;

// pops the oldest entry of map[key], waiting for it; the map is only touched under the group's mutex
template <class M>
bool local_wait_pop(LocalGroup* g, M& map, const typename M::key_type& key, typename M::mapped_type::value_type& out, double deadline_s)
{
    std::unique_lock<std::mutex> lock(g->m);
    auto& queue = map[key];
    const auto ready = [&] { return g->failed || !queue.empty(); };
    bool ok = true;
    if (deadline_s > 0.)
        ok = g->cv.wait_for(lock, std::chrono::duration<double>(deadline_s), ready);
    else
        g->cv.wait(lock, ready);
    if (!ok || g->failed) {
        g->failed = true;
        g->cv.notify_all();
        return false;
    }
    out = queue.front();
    queue.pop_front();
    return true;
}

} // namespace

struct nsdg_comm {
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr; // RCCL transport
    LocalGroup* local = nullptr; // local transport
    hipStream_t stream = nullptr; // communication stream
    int nplans = 0; // plans created so far (their indices match across the ranks of a group)
    bool broken = false; // a wait ran into the deadline: the streams may never drain, abort instead of draining
    // rehearsal aid (nsdg_comm_simulate_wire): simulated transfer time per exchange, fixed part and bandwidth per direction
    double sim_delay_us = 0., sim_gbs = 0.;
};

// ---------------------------------------------------------------------------------------------- plans
// per direction pair (a pack or unpack launch covers both directions): the transport plan of NSDG_RB_MAX_FIELDS DG2 fields
// moves 6 planes per field and direction = 48 blocks; the table travels as a kernel argument (64 entries = 1.8 KB)
constexpr int HALO_MAX_SEGS = 64;
static_assert(HALO_MAX_SEGS >= 2 * 6 * NSDG_RB_MAX_FIELDS, "segment table too small for the transport plan");

struct SegTable {
    double* ptr[HALO_MAX_SEGS]; // the row block in the caller's array
    long off[HALO_MAX_SEGS]; // its offset in the packed buffer of its direction
    long count[HALO_MAX_SEGS];
    int dir[HALO_MAX_SEGS]; // 0: first buffer (up / from above), 1: second buffer (down / from below)
    int n;
};

struct nsdg_halo {
    nsdg_ctx* ctx = nullptr;
    int index = 0; // creation index on the context's communicator
    int below = -1, above = -1; // neighbour ranks (-1: physical boundary)
    bool loopback = false; // both neighbours are this rank (one-GPU rehearsal)
    SegTable send, recv; // send.dir 0 = up, 1 = down; recv.dir 0 = from above, 1 = from below
    long n_up = 0, n_down = 0, n_above = 0, n_below = 0; // doubles per direction
    double *b_up = nullptr, *b_down = nullptr, *b_above = nullptr, *b_below = nullptr;
    long max_send = 0, max_recv = 0; // longest segment (grid sizing)
    hipEvent_t ev_ready = nullptr, ev_packed = nullptr, ev_arrived = nullptr, ev_done = nullptr;
    hipEvent_t ev_ack[2] = { nullptr, nullptr }; // local transport: "I have copied your buffer" for from-above / from-below
    bool local_sent[2] = { false, false }; // an acknowledgement is outstanding for up / down
    bool started = false;
    // exchange timing (nsdg_halo_stats_get): a ring of event pairs on the communication stream -- the host runs many
    // exchanges ahead of the device, so a pair is read when the ring comes round to it again (or at the query)
    static constexpr int RING = 32;
    hipEvent_t t0[RING] = {}, t1[RING] = {};
    bool pending[RING] = {};
    int slot = 0;
    long n_timed = 0, n_untimed = 0;
    double ms_total = 0.;
};

namespace {

// one launch gathers (PACK) or scatters (!PACK) every segment of both directions: blockIdx.y = segment
template <bool PACK>
__global__ __launch_bounds__(256) void halo_copy_kernel(SegTable T, double* __restrict__ buf0, double* __restrict__ buf1)
{
    const int s = blockIdx.y;
    const long n = T.count[s];
    double* __restrict__ a = T.ptr[s];
    double* __restrict__ b = (T.dir[s] == 0 ? buf0 : buf1) + T.off[s];
    // 16-byte accesses where both sides are 16-byte aligned (row blocks of the tiled arrays always are; node rows
    // have an odd length, so theirs depends on the row)
    const bool vec = ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const long n2 = n >> 1;
        double2* __restrict__ a2 = reinterpret_cast<double2*>(a);
        double2* __restrict__ b2 = reinterpret_cast<double2*>(b);
        for (long k = i; k < n2; k += stride) {
            if (PACK)
                b2[k] = a2[k];
            else
                a2[k] = b2[k];
        }
        if ((n & 1) && i == 0) {
            if (PACK)
                b[n - 1] = a[n - 1];
            else
                a[n - 1] = b[n - 1];
        }
    } else {
        for (long k = i; k < n; k += stride) {
            if (PACK)
                b[k] = a[k];
            else
                a[k] = b[k];
        }
    }
}

// Rehearsal aid (tools/rank_share_timing.py): a loopback exchange on one GPU has no wire time, so a kernel that spins
// for the time a real transfer would take can be put between pack and transport to see how much of it the overlap with
// the interior launch hides.  NSDG_HALO_DELAY_US = fixed time per exchange (latency); NSDG_HALO_SIM_GBS = link bandwidth
// per direction in GB/s, the time of the LARGER of the two directions is added (the two directions use different links).
// Never active unless one of the two is set in the environment.
__global__ void halo_delay_kernel(long ticks)
{
    const long t0 = (long)__builtin_amdgcn_s_memrealtime(); // 100 MHz
    while ((long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    }
}
long halo_delay_ticks(const nsdg_comm* c, long bytes_up, long bytes_down)
{
    double us = c->sim_delay_us;
    if (c->sim_gbs > 0.)
        us += 1e-3 * (double)std::max(bytes_up, bytes_down) / c->sim_gbs; // bytes / (GB/s) = ns
    return (long)(100. * us);
}
// the environment is read ONCE, when the communicator is created (NSDG_HALO_DELAY_US, NSDG_HALO_SIM_GBS: the rehearsals
// of tools/rank_share_timing.py and bench.py); a running process changes the values through nsdg_comm_simulate_wire
void wire_simulation_from_env(nsdg_comm* c)
{
    const char* v = std::getenv("NSDG_HALO_DELAY_US");
    const char* b = std::getenv("NSDG_HALO_SIM_GBS");
    c->sim_delay_us = (v && *v) ? std::max(std::atof(v), 0.) : 0.;
    c->sim_gbs = (b && *b) ? std::max(std::atof(b), 0.) : 0.;
}

// adds the time of the exchange recorded in ring slot k to the plan's totals; wait: block until it has finished
void harvest_slot(nsdg_halo* p, int k, bool wait)
{
    if (!p->pending[k])
        return;
    if (wait)
        (void)hipEventSynchronize(p->t1[k]);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p->t0[k], p->t1[k]) == hipSuccess) {
        p->ms_total += ms;
        ++p->n_timed;
    } else {
        (void)hipGetLastError(); // not ready: give the slot up
        ++p->n_untimed;
    }
    p->pending[k] = false;
}

int launch_copy(bool pack, const SegTable& T, long longest, double* buf0, double* buf1, hipStream_t stream)
{
    if (T.n == 0)
        return NSDG_OK;
    const int bx = (int)std::min<long>(std::max<long>(nsdg_div_up(longest, 256 * 8), 1), 512);
    if (pack)
        hipLaunchKernelGGL(halo_copy_kernel<true>, dim3(bx, T.n), dim3(256), 0, stream, T, buf0, buf1);
    else
        hipLaunchKernelGGL(halo_copy_kernel<false>, dim3(bx, T.n), dim3(256), 0, stream, T, buf0, buf1);
    NSDG_CHECK_LAUNCH();
    return NSDG_OK;
}

int add_segments(SegTable& T, int dir, int n, const nsdg_halo_seg* segs, long& total, long& longest)
{
    total = 0;
    for (int k = 0; k < n; ++k) {
        NSDG_CHECK_ARG(segs[k].ptr != nullptr && segs[k].count > 0, "halo segment with a null pointer or a non-positive count");
        NSDG_CHECK_ARG(T.n < HALO_MAX_SEGS, "too many halo segments in one plan");
        T.ptr[T.n] = segs[k].ptr;
        // keep every segment 16-byte aligned inside the packed buffer
        T.off[T.n] = total;
        T.count[T.n] = segs[k].count;
        T.dir[T.n] = dir;
        total += (segs[k].count + 1) & ~1L;
        longest = std::max<long>(longest, segs[k].count);
        ++T.n;
    }
    return NSDG_OK;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------- communicator
int nsdg_comm_unique_id(void* id)
{
    NSDG_CHECK_ARG(id != nullptr, "null id buffer");
    static_assert(NSDG_COMM_ID_BYTES == sizeof(ncclUniqueId), "NSDG_COMM_ID_BYTES must match ncclUniqueId");
    const int rc = rccl_load();
    if (rc != NSDG_OK)
        return rc;
    NSDG_CHECK_RCCL(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id)));
    return NSDG_OK;
}

static int comm_common(nsdg_ctx* ctx, int32_t rank, int32_t world)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "rank must be in [0, world)");
    if (ctx->comm) {
        nsdg_set_error("nsdg_comm_init: the context already has a communicator");
        return NSDG_ERR_STATE;
    }
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    nsdg_comm* c = new nsdg_comm();
    c->rank = rank;
    c->world = world;
    wire_simulation_from_env(c);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        nsdg_set_error("nsdg_comm_init: hipStreamCreateWithFlags failed");
        return NSDG_ERR_HIP;
    }
    ctx->comm = c;
    return NSDG_OK;
}

int nsdg_comm_init(nsdg_ctx* ctx, int32_t rank, int32_t world, const void* id)
{
    NSDG_CHECK_ARG(id != nullptr, "null id");
    int rc = rccl_load();
    if (rc != NSDG_OK)
        return rc;
    rc = comm_common(ctx, rank, world);
    if (rc != NSDG_OK)
        return rc;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    const ncclResult_t r = g_rccl.CommInitRank(&ctx->comm->nccl, world, uid, rank);
    if (r != ncclSuccess) {
        nsdg_set_error("nsdg_comm_init: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        (void)hipStreamDestroy(ctx->comm->stream);
        delete ctx->comm;
        ctx->comm = nullptr;
        return NSDG_ERR_COMM;
    }
    return NSDG_OK;
}

int nsdg_comm_init_local(nsdg_ctx* ctx, int64_t group, int32_t rank, int32_t world)
{
    const int rc = comm_common(ctx, rank, world);
    if (rc != NSDG_OK)
        return rc;
    std::lock_guard<std::mutex> lock(g_groups_mutex);
    LocalGroup*& g = g_groups[group];
    if (!g) {
        g = new LocalGroup();
        g->world = world;
    }
    if (g->world != world) {
        (void)hipStreamDestroy(ctx->comm->stream);
        delete ctx->comm;
        ctx->comm = nullptr;
        nsdg_set_error("nsdg_comm_init_local: group %ld exists with a different world size", (long)group);
        return NSDG_ERR_ARG;
    }
    ++g->refs;
    ctx->comm->local = g;
    ctx->comm_group = group;
    return NSDG_OK;
}

int nsdg_comm_finalize(nsdg_ctx* ctx)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    nsdg_comm* c = ctx->comm;
    if (!c)
        return NSDG_OK;
    (void)hipSetDevice(ctx->device);
    if (!c->broken)
        (void)hipStreamSynchronize(c->stream);
    if (c->nccl) {
        if (c->broken)
            (void)g_rccl.CommAbort(c->nccl); // ends the transfers a dead neighbour never answers
        else
            (void)g_rccl.CommDestroy(c->nccl);
    }
    if (c->local) {
        std::lock_guard<std::mutex> lock(g_groups_mutex);
        if (--c->local->refs == 0) {
            g_groups.erase(ctx->comm_group);
            delete c->local;
        }
    }
    if (!c->broken) // hipStreamDestroy waits for the stream's work: a broken communicator's stream is left to the process exit
        (void)hipStreamDestroy(c->stream);
    delete c;
    ctx->comm = nullptr;
    return NSDG_OK;
}

int nsdg_comm_rank(nsdg_ctx* ctx, int32_t* rank, int32_t* world)
{
    NSDG_CHECK_ARG(ctx && rank && world, "null argument");
    *rank = ctx->comm ? ctx->comm->rank : 0;
    *world = ctx->comm ? ctx->comm->world : 1;
    return NSDG_OK;
}

// ---------------------------------------------------------------------------------------------- plans
int nsdg_halo_plan_create(nsdg_ctx* ctx, int32_t rank_below, int32_t rank_above, int32_t n_up, const nsdg_halo_seg* up_send, int32_t n_down,
    const nsdg_halo_seg* down_send, int32_t n_above, const nsdg_halo_seg* from_above, int32_t n_below, const nsdg_halo_seg* from_below,
    nsdg_halo** out)
{
    NSDG_CHECK_ARG(ctx != nullptr && out != nullptr, "null argument");
    *out = nullptr;
    if (!ctx->comm) {
        nsdg_set_error("nsdg_halo_plan_create: nsdg_comm_init was not called on this context");
        return NSDG_ERR_STATE;
    }
    const nsdg_comm* c = ctx->comm;
    NSDG_CHECK_ARG(rank_below >= -1 && rank_below < c->world && rank_above >= -1 && rank_above < c->world, "neighbour rank out of range");
    NSDG_CHECK_ARG(n_up >= 0 && n_down >= 0 && n_above >= 0 && n_below >= 0, "negative segment count");
    NSDG_CHECK_ARG((n_up == 0 || up_send) && (n_down == 0 || down_send) && (n_above == 0 || from_above) && (n_below == 0 || from_below),
        "null segment list");
    NSDG_CHECK_ARG(rank_above >= 0 || (n_up == 0 && n_above == 0), "segments towards a neighbour above that does not exist");
    NSDG_CHECK_ARG(rank_below >= 0 || (n_down == 0 && n_below == 0), "segments towards a neighbour below that does not exist");
    const bool self_above = rank_above == c->rank, self_below = rank_below == c->rank;
    NSDG_CHECK_ARG(self_above == self_below || (rank_above < 0 || rank_below < 0) , "loopback needs both neighbours to be this rank");
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    nsdg_halo* p = new nsdg_halo();
    p->ctx = ctx;
    p->index = ctx->comm->nplans++;
    p->below = rank_below, p->above = rank_above;
    p->loopback = self_above || self_below;
    p->send.n = p->recv.n = 0;
    int rc = add_segments(p->send, 0, n_up, up_send, p->n_up, p->max_send);
    if (rc == NSDG_OK)
        rc = add_segments(p->send, 1, n_down, down_send, p->n_down, p->max_send);
    if (rc == NSDG_OK)
        rc = add_segments(p->recv, 0, n_above, from_above, p->n_above, p->max_recv);
    if (rc == NSDG_OK)
        rc = add_segments(p->recv, 1, n_below, from_below, p->n_below, p->max_recv);
    if (rc != NSDG_OK) {
        delete p;
        return rc;
    }
    auto alloc = [&](double** b, long n) { return n ? hipMalloc(reinterpret_cast<void**>(b), n * sizeof(double)) : hipSuccess; };
    bool ok = alloc(&p->b_up, p->n_up) == hipSuccess && alloc(&p->b_down, p->n_down) == hipSuccess
        && alloc(&p->b_above, p->n_above) == hipSuccess && alloc(&p->b_below, p->n_below) == hipSuccess;
    for (hipEvent_t* e : { &p->ev_ready, &p->ev_packed, &p->ev_arrived, &p->ev_done, &p->ev_ack[0], &p->ev_ack[1] })
        ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < nsdg_halo::RING; ++k)
        ok = ok && hipEventCreate(&p->t0[k]) == hipSuccess && hipEventCreate(&p->t1[k]) == hipSuccess;
    if (!ok) {
        nsdg_halo_plan_destroy(p);
        nsdg_set_error("nsdg_halo_plan_create: device allocation failed");
        return NSDG_ERR_HIP;
    }
    *out = p;
    return NSDG_OK;
}

int nsdg_halo_plan_destroy(nsdg_halo* p)
{
    if (!p)
        return NSDG_OK;
    (void)hipSetDevice(p->ctx->device);
    if (p->ctx->comm && p->ctx->comm->broken) {
        // the communication stream may never drain: hipFree would wait for it.  The buffers and events are left to the
        // process exit that follows a broken communicator.
        delete p;
        return NSDG_OK;
    }
    if (p->ctx->comm)
        (void)hipStreamSynchronize(p->ctx->comm->stream);
    for (int k = 0; k < nsdg_halo::RING; ++k) {
        if (p->t0[k])
            (void)hipEventDestroy(p->t0[k]);
        if (p->t1[k])
            (void)hipEventDestroy(p->t1[k]);
    }
    for (double* b : { p->b_up, p->b_down, p->b_above, p->b_below })
        if (b)
            (void)hipFree(b);
    for (hipEvent_t e : { p->ev_ready, p->ev_packed, p->ev_arrived, p->ev_done, p->ev_ack[0], p->ev_ack[1] })
        if (e)
            (void)hipEventDestroy(e);
    delete p;
    return NSDG_OK;
}

int nsdg_halo_counts(const nsdg_halo* p, int64_t* up, int64_t* down, int64_t* above, int64_t* below)
{
    NSDG_CHECK_ARG(p && up && down && above && below, "null argument");
    *up = p->n_up, *down = p->n_down, *above = p->n_above, *below = p->n_below;
    return NSDG_OK;
}

// ---------------------------------------------------------------------------------------------- the exchange
int nsdg_halo_start(nsdg_ctx* ctx, nsdg_halo* p)
{
    NSDG_CHECK_ARG(ctx && p && p->ctx == ctx, "plan does not belong to this context");
    if (!ctx->comm) {
        nsdg_set_error("nsdg_halo_start: the context has no communicator");
        return NSDG_ERR_STATE;
    }
    if (p->started) {
        nsdg_set_error("nsdg_halo_start: the previous exchange of this plan was not finished");
        return NSDG_ERR_STATE;
    }
    nsdg_comm* c = ctx->comm;
    if (c->broken) {
        nsdg_set_error("nsdg_halo_start: the communicator is broken (an earlier wait ran into the deadline)");
        return NSDG_ERR_COMM;
    }
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    // the exchange sees everything the compute stream has been given so far
    NSDG_CHECK_HIP(hipEventRecord(p->ev_ready, ctx->stream));
    NSDG_CHECK_HIP(hipStreamWaitEvent(c->stream, p->ev_ready, 0));
    // the ring has come round.  NEVER block here: the host runs many exchanges ahead of the device, and with a dead RCCL
    // neighbour the event of an exchange 32 starts ago never fires -- an exchange that has not finished gives its slot
    // up and is counted as untimed
    harvest_slot(p, p->slot, false);
    NSDG_CHECK_HIP(hipEventRecord(p->t0[p->slot], c->stream)); // "the data to send is ready"
    if (c->local && !p->loopback) {
        // the neighbours must have copied the previous contents of the send buffers out
        const int peers[2] = { p->above, p->below };
        for (int d = 0; d < 2; ++d)
            if (p->local_sent[d]) {
                hipEvent_t ack;
                if (!local_wait_pop(c->local, c->local->ack, std::make_tuple(c->rank, peers[d], p->index), ack, ctx->comm_deadline_s)) {
                    nsdg_set_error("nsdg_halo_start: a rank of the local group failed or did not answer within %g s", ctx->comm_deadline_s);
                    return NSDG_ERR_COMM;
                }
                NSDG_CHECK_HIP(hipStreamWaitEvent(c->stream, ack, 0));
                p->local_sent[d] = false;
            }
    }
    int rc = launch_copy(true, p->send, p->max_send, p->b_up, p->b_down, c->stream);
    if (rc != NSDG_OK)
        return rc;
    if (p->n_up || p->n_down) { // rehearsal only: simulated transfer time
        const long ticks = halo_delay_ticks(c, p->n_up * (long)sizeof(double), p->n_down * (long)sizeof(double));
        if (ticks > 0)
            hipLaunchKernelGGL(halo_delay_kernel, dim3(1), dim3(1), 0, c->stream, ticks);
    }
    if (c->nccl) {
        NSDG_CHECK_RCCL(g_rccl.GroupStart());
        ncclResult_t r = ncclSuccess;
        if (p->n_up && r == ncclSuccess)
            r = g_rccl.Send(p->b_up, (size_t)p->n_up, ncclDouble, p->above, c->nccl, c->stream);
        if (p->n_down && r == ncclSuccess)
            r = g_rccl.Send(p->b_down, (size_t)p->n_down, ncclDouble, p->below, c->nccl, c->stream);
        // loopback: both neighbours are this rank, so the send upwards (posted first) must meet the receive from below
        for (int k = 0; k < 2; ++k) {
            const bool from_below = p->loopback ? k == 0 : k == 1;
            if (from_below && p->n_below && r == ncclSuccess)
                r = g_rccl.Recv(p->b_below, (size_t)p->n_below, ncclDouble, p->below, c->nccl, c->stream);
            if (!from_below && p->n_above && r == ncclSuccess)
                r = g_rccl.Recv(p->b_above, (size_t)p->n_above, ncclDouble, p->above, c->nccl, c->stream);
        }
        const ncclResult_t e = g_rccl.GroupEnd(); // always close the group
        NSDG_CHECK_RCCL(r);
        NSDG_CHECK_RCCL(e);
    } else if (p->loopback) {
        // local loopback: what goes up arrives from below and vice versa (sizes checked at run time)
        if (p->n_up != p->n_below || p->n_down != p->n_above) {
            nsdg_set_error("nsdg_halo_start: loopback needs matching send and receive sizes");
            return NSDG_ERR_ARG;
        }
        if (p->n_up)
            NSDG_CHECK_HIP(hipMemcpyAsync(p->b_below, p->b_up, p->n_up * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        if (p->n_down)
            NSDG_CHECK_HIP(hipMemcpyAsync(p->b_above, p->b_down, p->n_down * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else {
        // local transport: publish the packed buffers; the receiving side pulls them in nsdg_halo_finish
        NSDG_CHECK_HIP(hipEventRecord(p->ev_packed, c->stream));
        LocalGroup* g = c->local;
        std::lock_guard<std::mutex> lock(g->m);
        if (p->n_up) {
            g->box[std::make_tuple(c->rank, p->above, p->index)].push_back({ p->b_up, p->n_up, p->ev_packed });
            p->local_sent[0] = true;
        }
        if (p->n_down) {
            g->box[std::make_tuple(c->rank, p->below, p->index)].push_back({ p->b_down, p->n_down, p->ev_packed });
            p->local_sent[1] = true;
        }
        g->cv.notify_all();
    }
    p->started = true;
    return NSDG_OK;
}

int nsdg_halo_finish(nsdg_ctx* ctx, nsdg_halo* p)
{
    NSDG_CHECK_ARG(ctx && p && p->ctx == ctx, "plan does not belong to this context");
    if (!p->started) {
        nsdg_set_error("nsdg_halo_finish: nsdg_halo_start was not called");
        return NSDG_ERR_STATE;
    }
    nsdg_comm* c = ctx->comm;
    if (c->broken) {
        nsdg_set_error("nsdg_halo_finish: the communicator is broken (an earlier wait ran into the deadline)");
        return NSDG_ERR_COMM;
    }
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    p->started = false;
    if (c->local && !p->loopback) {
        LocalGroup* g = c->local;
        const int peers[2] = { p->above, p->below };
        double* bufs[2] = { p->b_above, p->b_below };
        const long counts[2] = { p->n_above, p->n_below };
        for (int d = 0; d < 2; ++d) {
            if (!counts[d])
                continue;
            LocalMsg msg;
            if (!local_wait_pop(g, g->box, std::make_tuple(peers[d], c->rank, p->index), msg, ctx->comm_deadline_s)) {
                nsdg_set_error("nsdg_halo_finish: a rank of the local group failed or did not answer within %g s", ctx->comm_deadline_s);
                return NSDG_ERR_COMM;
            }
            if (msg.count != counts[d]) {
                std::lock_guard<std::mutex> lock(g->m);
                g->failed = true;
                g->cv.notify_all();
                nsdg_set_error("nsdg_halo_finish: rank %d sent %ld doubles where %ld are expected", peers[d], (long)msg.count, counts[d]);
                return NSDG_ERR_COMM;
            }
            NSDG_CHECK_HIP(hipStreamWaitEvent(c->stream, msg.ready, 0));
            NSDG_CHECK_HIP(hipMemcpyAsync(bufs[d], msg.buf, counts[d] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            NSDG_CHECK_HIP(hipEventRecord(p->ev_ack[d], c->stream));
            std::lock_guard<std::mutex> lock(g->m);
            g->ack[std::make_tuple(peers[d], c->rank, p->index)].push_back(p->ev_ack[d]);
            g->cv.notify_all();
        }
    }
    const int rc = launch_copy(false, p->recv, p->max_recv, p->b_above, p->b_below, c->stream);
    if (rc != NSDG_OK)
        return rc;
    NSDG_CHECK_HIP(hipEventRecord(p->ev_done, c->stream));
    NSDG_CHECK_HIP(hipStreamWaitEvent(ctx->stream, p->ev_done, 0));
    NSDG_CHECK_HIP(hipEventRecord(p->t1[p->slot], c->stream)); // "the ghost rows are written"
    p->pending[p->slot] = true;
    p->slot = (p->slot + 1) % nsdg_halo::RING;
    return NSDG_OK;
}

int nsdg_halo_stats_get(nsdg_ctx* ctx, nsdg_halo* p, nsdg_halo_stats* out, int32_t reset)
{
    NSDG_CHECK_ARG(ctx && p && p->ctx == ctx && out, "plan does not belong to this context");
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    if (!(ctx->comm && ctx->comm->broken)) {
        // the query waits for the last exchanges -- but not beyond the communicator's deadline
        const int rc = ctx->comm ? nsdg_comm_bounded_drain(ctx) : NSDG_OK;
        if (rc != NSDG_OK)
            return rc;
        for (int k = 0; k < nsdg_halo::RING; ++k)
            harvest_slot(p, k, false);
    }
    out->exchanges = p->n_timed + p->n_untimed;
    out->untimed = p->n_untimed;
    out->ms = p->ms_total;
    out->bytes_sent = (p->n_up + p->n_down) * (int64_t)sizeof(double);
    out->bytes_received = (p->n_above + p->n_below) * (int64_t)sizeof(double);
    if (reset) {
        p->n_timed = p->n_untimed = 0;
        p->ms_total = 0.;
    }
    return NSDG_OK;
}

int nsdg_comm_simulate_wire(nsdg_ctx* ctx, double delay_us, double gbs)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(delay_us >= 0. && gbs >= 0., "delay and bandwidth must be >= 0 (0 = off)");
    if (!ctx->comm) {
        nsdg_set_error("nsdg_comm_simulate_wire: the context has no communicator");
        return NSDG_ERR_STATE;
    }
    ctx->comm->sim_delay_us = delay_us;
    ctx->comm->sim_gbs = gbs;
    return NSDG_OK;
}

int nsdg_comm_deadline_set(nsdg_ctx* ctx, double seconds)
{
    NSDG_CHECK_ARG(ctx != nullptr, "null context");
    NSDG_CHECK_ARG(seconds >= 0., "the deadline must be >= 0 (0 = wait for ever)");
    ctx->comm_deadline_s = seconds;
    return NSDG_OK;
}

} // extern "C"

// Drains the context's compute and communication streams, but not for longer than the communicator's deadline: with a
// neighbour rank gone, an ncclRecv (or a kernel ordered behind it) never completes and a plain hipStreamSynchronize
// would block this rank for ever.  On expiry the communicator is marked broken and NSDG_ERR_COMM is returned.
int nsdg_comm_bounded_drain(nsdg_ctx* ctx)
{
    nsdg_comm* c = ctx->comm;
    NSDG_CHECK_HIP(hipSetDevice(ctx->device));
    if (c->broken) {
        nsdg_set_error("nsdg_ctx_synchronize: the communicator is broken (an earlier wait ran into the deadline)");
        return NSDG_ERR_COMM;
    }
    if (ctx->comm_deadline_s <= 0.) {
        NSDG_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        NSDG_CHECK_HIP(hipStreamSynchronize(c->stream));
        return NSDG_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t streams[2] = { ctx->stream, c->stream };
    for (hipStream_t st : streams) {
        for (long spins = 0;; ++spins) {
            const hipError_t e = hipStreamQuery(st);
            if (e == hipSuccess)
                break;
            if (e != hipErrorNotReady) {
                nsdg_set_error("nsdg_ctx_synchronize: hipStreamQuery failed: %s", hipGetErrorString(e));
                return NSDG_ERR_HIP;
            }
            (void)hipGetLastError();
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > ctx->comm_deadline_s) {
                c->broken = true;
                if (c->local) {
                    std::lock_guard<std::mutex> lock(c->local->m);
                    c->local->failed = true;
                    c->local->cv.notify_all();
                }
                nsdg_set_error("nsdg_ctx_synchronize: the streams of rank %d did not drain within %g s -- a neighbour rank has probably died; "
                               "the communicator is broken, leave the process without synchronising the device",
                    c->rank, ctx->comm_deadline_s);
                return NSDG_ERR_COMM;
            }
            // short waits spin (a step is a few ms), long ones sleep
            if (spins > 2000)
                std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    return NSDG_OK;
}
