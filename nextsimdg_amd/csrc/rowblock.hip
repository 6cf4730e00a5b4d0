// rowblock.hip -- the per-step driver of one rank's row block behind the C ABI: the mEVP sub-cycle and the
// SSP-RK3 transport with their ghost-row exchanges (halo.hip), one call each per model step.
//
// Why here and not in the callers: both hosts -- the C++ host layer (host/src/DynamicsStep.cpp, multi-rank) and the
// Python driver (nextsimdg_amd/rowblock.py) -- run the same sequence; issued from C it costs ~0.1 ms of host time
// per step instead of ~2 ms of Python + torch.distributed, which at 8 row blocks (a 7 ms step) is what decides
// whether the GPU or the host sets the pace.  The launches between two exchanges can be replayed as one hipGraph.
//
// The sequence (identical to rowblock.py DynamicsCore.subcycle / .transport, which remains the specification
// that the gloo tests pin to the single-domain run bit for bit):
//   kernels with v = 3 or 2 sub-iterations per pass need ghost zones of (d, d-1) element rows, d = v k, and
//   exchange them after every GROUP of k passes; pass i of a group of m covers the owned rows extended by
//   v (m-i) ghost rows on each side (advanced redundantly, bit-identically to their owners).  The last pass of a
//   group is split: the rows whose results travel are launched first, the exchange is posted, the interior follows.
//   What is left of nsub runs as a two-iteration pass and / or single sub-iterations (exchange after each).
//   Transport: with >= 3 ghost rows per side the first two RK stages also advance 2 / 1 ghost rows and the
//   advected fields are exchanged once per step; otherwise after every stage.
#include <algorithm>
#include <cstring>
#include <initializer_list>
#include <map>
#include <tuple>
#include <vector>

#include "mevp_common.h"

using nsdg_mevp_detail::tiles_per_row;

namespace {

struct Range {
    int j0, j1;
};

struct Geometry {
    int nx, ny, j0, j1, depth_below, depth_above, below, above;
    bool has_below() const { return below >= 0; }
    bool has_above() const { return above >= 0; }
    bool multi() const { return below >= 0 || above >= 0; }
    int gb() const { return has_below() ? depth_below : 0; } // ghost element rows actually present
    int gt() const { return has_above() ? depth_above : 0; }
};

int check_geometry(const Geometry& g)
{
    NSDG_CHECK_ARG(g.nx > 0 && g.ny > 0, "empty local array");
    NSDG_CHECK_ARG(g.depth_below >= 0 && g.depth_above >= 0, "negative ghost depth");
    NSDG_CHECK_ARG(g.j0 == g.gb() && g.j1 == g.ny - g.gt() && g.j0 < g.j1,
        "owned rows must be the local array minus the ghost rows towards existing neighbours");
    NSDG_CHECK_ARG(!g.multi() || (g.depth_below >= 1), "a block with neighbours needs at least one ghost row below");
    // depth_below element rows travel upwards and depth_above + 1 element rows (2 depth_above + 1 node rows) downwards:
    // a block with fewer owned rows would pack rows of its own ghost zone, which are stale after the last pass of a group
    NSDG_CHECK_ARG(!g.multi() || g.j1 - g.j0 >= std::max(g.depth_below, g.depth_above + 1),
        "a block with neighbours must own at least max(depth_below, depth_above + 1) rows (the rows it sends)");
    return NSDG_OK;
}

// row blocks of the arrays a plan moves
struct SegList {
    std::vector<nsdg_halo_seg> up, down, above, below;
};

void add_tiled_rows(const Geometry& g, double* f, int nc, SegList& L)
{ // element rows of a tiled array: depth_below rows travel upwards, depth_above rows downwards
    const long row = (long)tiles_per_row(g.nx) * nc * 64;
    auto seg = [&](int a, int c) { return nsdg_halo_seg { f + a * row, (c - a) * row }; };
    if (g.has_above()) {
        L.up.push_back(seg(g.j1 - g.depth_below, g.j1));
        if (g.gt())
            L.above.push_back(seg(g.j1, g.j1 + g.gt()));
    }
    if (g.has_below()) {
        if (g.depth_above)
            L.down.push_back(seg(g.j0, g.j0 + g.depth_above));
        L.below.push_back(seg(g.j0 - g.gb(), g.j0));
    }
}

void add_plane_rows(const Geometry& g, double* f, int nc, SegList& L)
{ // the same for coefficient-major planes [nc][ny][nx]: one block per plane
    const long plane = (long)g.nx * g.ny;
    for (int c = 0; c < nc; ++c) {
        double* p = f + c * plane;
        auto seg = [&](int a, int b) { return nsdg_halo_seg { p + (long)a * g.nx, (long)(b - a) * g.nx }; };
        if (g.has_above()) {
            L.up.push_back(seg(g.j1 - g.depth_below, g.j1));
            if (g.gt())
                L.above.push_back(seg(g.j1, g.j1 + g.gt()));
        }
        if (g.has_below()) {
            if (g.depth_above)
                L.down.push_back(seg(g.j0, g.j0 + g.depth_above));
            L.below.push_back(seg(g.j0 - g.gb(), g.j0));
        }
    }
}

void add_node_rows(const Geometry& g, double* f, int rows_down, SegList& L)
{ // CG2 nodal array: 2*depth_below node rows travel upwards, rows_down rows downwards
    const long nn = 2L * g.nx + 1;
    const int up = 2 * g.depth_below;
    auto seg = [&](int a, int n) { return nsdg_halo_seg { f + a * nn, n * nn }; };
    if (g.has_above()) {
        L.up.push_back(seg(2 * g.j1 - up, up));
        L.above.push_back(seg(2 * g.j1, rows_down));
    }
    if (g.has_below()) {
        L.down.push_back(seg(2 * g.j0, rows_down));
        L.below.push_back(seg(2 * g.j0 - up, up));
    }
}

int make_plan(nsdg_ctx* ctx, const Geometry& g, const SegList& L, nsdg_halo** out)
{
    *out = nullptr;
    if (L.up.empty() && L.down.empty() && L.above.empty() && L.below.empty())
        return NSDG_OK;
    return nsdg_halo_plan_create(ctx, g.below, g.above, (int)L.up.size(), L.up.data(), (int)L.down.size(), L.down.data(), (int)L.above.size(),
        L.above.data(), (int)L.below.size(), L.below.data(), out);
}

// ---- hipGraph replay of a fixed launch sequence -------------------------------------------------------------
// A captured launch bakes in the launch constants the kernels take from the context (K = rho beta / dt, 1/alpha,
// Delta_min^2, the cell size, the strip height, the kernel variant): the cache remembers the values its graphs were
// recorded with and is dropped when any of them has changed, so a replay never mixes two parameter sets.
struct GraphStamp {
    double v[8];
    int k[6];
    bool operator==(const GraphStamp& o) const { return std::memcmp(this, &o, sizeof *this) == 0; }
};

GraphStamp graph_stamp(const nsdg_ctx* c)
{
    GraphStamp s;
    std::memset(&s, 0, sizeof s);
    const nsdg_mevp_params& P = c->mevp;
    const double v[8] = { c->pack_dt, P.alpha, P.beta, P.rho_ice, P.fc, P.delta_min, c->hx, c->hy };
    const int k[6] = { c->strip_rows, c->mevp_variant, c->fused_min_waves, c->nx, c->ny, c->num_cus };
    std::memcpy(s.v, v, sizeof v);
    std::memcpy(s.k, k, sizeof k);
    return s;
}

struct GraphCache {
    hipStream_t capture = nullptr; // launches are recorded on this stream (the context's own stream may be the null stream)
    hipEvent_t fork = nullptr, join = nullptr;
    std::map<std::tuple<int, int, int, int>, hipGraphExec_t> execs;
    GraphStamp stamp; // launch constants of the recorded graphs
    void drop_execs()
    {
        for (auto& kv : execs)
            (void)hipGraphExecDestroy(kv.second);
        execs.clear();
    }
    void destroy()
    {
        drop_execs();
        if (capture)
            (void)hipStreamDestroy(capture);
        if (fork)
            (void)hipEventDestroy(fork);
        if (join)
            (void)hipEventDestroy(join);
        capture = nullptr, fork = join = nullptr;
    }
};

} // namespace

// =================================================================================================== mEVP sub-cycle
struct nsdg_rb_mevp {
    nsdg_ctx* ctx;
    Geometry g;
    nsdg_rb_mevp_desc d;
    int per_pass, group_passes;
    nsdg_halo* rows_plan[2] = { nullptr, nullptr }; // ghost zones of the multi-iteration passes, destination buffer 0 / 1
    nsdg_halo* node_plan[2] = { nullptr, nullptr }; // velocity node rows of the single-iteration kernel
    GraphCache graphs;
};

namespace {

int pass_launch(nsdg_rb_mevp* p, int v, Range r, int par)
{ // v sub-iterations on rows r: buffers `par` -> 1 - par
    const nsdg_rb_mevp_desc& d = p->d;
    const int q = 1 - par;
    if (r.j0 >= r.j1)
        return NSDG_OK;
    if (v == 4)
        return nsdg_mevp_iterate4(p->ctx, r.j0, r.j1, d.s11[par], d.s12[par], d.s22[par], d.s11[q], d.s12[q], d.s22[q], d.u[par], d.v[par], d.u[q],
            d.v[q], d.packed, d.pg);
    if (v == 3)
        return nsdg_mevp_iterate3(p->ctx, r.j0, r.j1, d.s11[par], d.s12[par], d.s22[par], d.s11[q], d.s12[q], d.s22[q], d.u[par], d.v[par], d.u[q],
            d.v[q], d.packed, d.pg);
    return nsdg_mevp_iterate2(p->ctx, r.j0, r.j1, d.s11[par], d.s12[par], d.s22[par], d.s11[q], d.s12[q], d.s22[q], d.u[par], d.v[par], d.u[q], d.v[q],
        d.packed, d.pg);
}

int single_launch(nsdg_rb_mevp* p, int k0, int j0, int j1, int par)
{
    const nsdg_rb_mevp_desc& d = p->d;
    const int q = 1 - par;
    return nsdg_mevp_iterate(p->ctx, k0, j0, j1, d.s11[par], d.s12[par], d.s22[par], d.s11[q], d.s12[q], d.s22[q], d.u[par], d.v[par], d.u[q], d.v[q],
        d.packed, d.pg);
}

// the launches of one pass of a group: ext > 0 -> one launch over the owned rows extended by v*ext ghost rows per side;
// ext == 0 and split -> the rows whose results travel first, the interior last
void pass_ranges(const Geometry& g, int v, bool split, int ext, std::vector<Range>& out)
{
    out.clear();
    int lo = std::max(g.j0 - v * ext, 0), hi = std::min(g.j1 + v * ext, g.ny);
    if (split && ext == 0) {
        if (g.has_above()) {
            out.push_back({ g.j1 - g.depth_below, g.j1 });
            hi = g.j1 - g.depth_below;
        }
        if (g.has_below()) {
            out.push_back({ g.j0, g.j0 + g.depth_above + 1 });
            lo = g.j0 + g.depth_above + 1;
        }
    }
    out.push_back({ lo, hi });
}

// Runs `body` (kernel launches on ctx->stream only) directly, or -- with graphs on -- records it once per key on the
// capture stream and replays the instantiated graph behind the context's stream.
template <class F>
int run_or_replay(nsdg_rb_mevp* p, std::tuple<int, int, int, int> key, F&& body)
{
    nsdg_ctx* ctx = p->ctx;
    if (!p->d.use_graph)
        return body();
    GraphCache& G = p->graphs;
    if (!G.capture) {
        NSDG_CHECK_HIP(hipStreamCreateWithFlags(&G.capture, hipStreamNonBlocking));
        NSDG_CHECK_HIP(hipEventCreateWithFlags(&G.fork, hipEventDisableTiming));
        NSDG_CHECK_HIP(hipEventCreateWithFlags(&G.join, hipEventDisableTiming));
    }
    const GraphStamp now = graph_stamp(ctx);
    if (!G.execs.empty() && !(G.stamp == now)) {
        // parameters, time step, grid or launch geometry changed since the graphs were recorded
        NSDG_CHECK_HIP(hipStreamSynchronize(G.capture)); // no replay may still be running when its exec is destroyed
        G.drop_execs();
    }
    G.stamp = now;
    auto it = G.execs.find(key);
    if (it == G.execs.end()) {
        hipStream_t user = ctx->stream;
        ctx->stream = G.capture;
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(G.capture, hipStreamCaptureModeThreadLocal);
        int rc = NSDG_OK;
        if (e == hipSuccess) {
            rc = body();
            e = hipStreamEndCapture(G.capture, &graph);
        }
        ctx->stream = user;
        if (rc != NSDG_OK || e != hipSuccess) {
            if (graph)
                (void)hipGraphDestroy(graph);
            if (rc != NSDG_OK)
                return rc;
        }
        NSDG_CHECK_HIP(e);
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        NSDG_CHECK_HIP(e);
        it = G.execs.emplace(key, exec).first;
    }
    // replay on the capture stream, ordered behind and before the context's stream
    NSDG_CHECK_HIP(hipEventRecord(G.fork, ctx->stream));
    NSDG_CHECK_HIP(hipStreamWaitEvent(G.capture, G.fork, 0));
    NSDG_CHECK_HIP(hipGraphLaunch(it->second, G.capture));
    NSDG_CHECK_HIP(hipEventRecord(G.join, G.capture));
    NSDG_CHECK_HIP(hipStreamWaitEvent(ctx->stream, G.join, 0));
    return NSDG_OK;
}

} // namespace

extern "C" {

int nsdg_rb_mevp_create(nsdg_ctx* ctx, const nsdg_rb_mevp_desc* desc, nsdg_rb_mevp** out)
{
    NSDG_CHECK_ARG(ctx && desc && out, "null argument");
    *out = nullptr;
    const nsdg_rb_mevp_desc& d = *desc;
    Geometry g = { d.nx, d.ny, d.j0, d.j1, d.depth_below, d.depth_above, d.rank_below, d.rank_above };
    int rc = check_geometry(g);
    if (rc != NSDG_OK)
        return rc;
    NSDG_CHECK_ARG(d.nsub >= 0, "negative sub-iteration count");
    for (int k = 0; k < 2; ++k)
        NSDG_CHECK_ARG(d.s11[k] && d.s12[k] && d.s22[k] && d.u[k] && d.v[k], "null ping-pong buffer");
    NSDG_CHECK_ARG(d.packed && d.pg, "null field pointer");
    if (g.multi() && !ctx->comm) {
        nsdg_set_error("nsdg_rb_mevp_create: a block with neighbours needs nsdg_comm_init* first");
        return NSDG_ERR_STATE;
    }
    nsdg_rb_mevp* p = new nsdg_rb_mevp();
    p->ctx = ctx;
    p->g = g;
    p->d = d;
    // v sub-iterations per kernel pass need a (v k, v k - 1) ghost depth; a block without neighbours needs none
    p->per_pass = 1;
    for (int v : { 4, 3, 2 }) {
        const bool deep = g.depth_below >= v && g.depth_below % v == 0 && g.depth_above == g.depth_below - 1;
        if (ctx->mevp_variant >= v && (!g.multi() || deep)) {
            p->per_pass = v;
            break;
        }
    }
    p->group_passes = (p->per_pass >= 2 && g.multi()) ? g.depth_below / p->per_pass : 1;
    if (g.multi()) {
        for (int q = 0; q < 2 && rc == NSDG_OK; ++q) { // q: the buffer the pass has just written
            SegList L;
            if (p->per_pass >= 2) {
                add_tiled_rows(g, d.s11[q], 8, L);
                add_tiled_rows(g, d.s12[q], 8, L);
                add_tiled_rows(g, d.s22[q], 8, L);
                add_node_rows(g, d.u[q], 2 * g.depth_above + 1, L);
                add_node_rows(g, d.v[q], 2 * g.depth_above + 1, L);
                rc = make_plan(ctx, g, L, &p->rows_plan[q]);
            } else {
                add_node_rows(g, d.u[q], 1, L);
                add_node_rows(g, d.v[q], 1, L);
                rc = make_plan(ctx, g, L, &p->node_plan[q]);
            }
        }
    }
    if (rc != NSDG_OK) {
        nsdg_rb_mevp_destroy(p);
        return rc;
    }
    *out = p;
    return NSDG_OK;
}

int nsdg_rb_mevp_destroy(nsdg_rb_mevp* p)
{
    if (!p)
        return NSDG_OK;
    (void)hipSetDevice(p->ctx->device);
    p->graphs.destroy();
    for (int q = 0; q < 2; ++q) {
        nsdg_halo_plan_destroy(p->rows_plan[q]);
        nsdg_halo_plan_destroy(p->node_plan[q]);
    }
    delete p;
    return NSDG_OK;
}

int nsdg_rb_mevp_info(const nsdg_rb_mevp* p, int32_t* per_pass, int32_t* group_passes)
{
    NSDG_CHECK_ARG(p && per_pass && group_passes, "null argument");
    *per_pass = p->per_pass;
    *group_passes = p->group_passes;
    return NSDG_OK;
}

static int sum_stats(nsdg_ctx* ctx, std::initializer_list<nsdg_halo*> plans, nsdg_halo_stats* out, int32_t reset)
{
    std::memset(out, 0, sizeof *out);
    for (nsdg_halo* h : plans) {
        if (!h)
            continue;
        nsdg_halo_stats one;
        const int rc = nsdg_halo_stats_get(ctx, h, &one, reset);
        if (rc != NSDG_OK)
            return rc;
        out->exchanges += one.exchanges, out->untimed += one.untimed, out->ms += one.ms;
        out->bytes_sent = std::max(out->bytes_sent, one.bytes_sent);
        out->bytes_received = std::max(out->bytes_received, one.bytes_received);
    }
    return NSDG_OK;
}

int nsdg_rb_mevp_stats(nsdg_ctx* ctx, nsdg_rb_mevp* p, nsdg_halo_stats* out, int32_t reset)
{
    NSDG_CHECK_ARG(ctx && p && p->ctx == ctx && out, "plan does not belong to this context");
    return sum_stats(ctx, { p->rows_plan[0], p->rows_plan[1], p->node_plan[0], p->node_plan[1] }, out, reset);
}

int nsdg_rb_mevp_run(nsdg_ctx* ctx, nsdg_rb_mevp* p, int32_t parity, int32_t* parity_out)
{
    NSDG_CHECK_ARG(ctx && p && p->ctx == ctx && parity_out, "plan does not belong to this context");
    NSDG_CHECK_ARG(parity == 0 || parity == 1, "parity must be 0 or 1");
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(ctx->nx == p->g.nx && ctx->ny == p->g.ny, "nsdg_grid_set does not match the plan's local array");
    const Geometry& g = p->g;
    const nsdg_rb_mevp_desc& d = p->d;
    int par = parity, it = 0, rc = nsdg_p2p_check(ctx, __func__);
    if (rc != NSDG_OK)
        return rc; // a pipeline wait gave up in an earlier launch: the fields this call would start from are wrong
    std::vector<Range> rng;
    if (p->per_pass >= 2) {
        // a block without neighbours never exchanges: all passes of one kernel form one group (one graph)
        const int k = g.multi() ? p->group_passes : (1 << 28);
        const bool split_ok = d.overlap && g.multi() && (g.j1 - g.j0) >= g.depth_below + g.depth_above + 5;
        // passes of per_pass sub-iterations, then what is left of nsub through the kernels with fewer sub-iterations per pass
        for (int v = p->per_pass; v >= 2; --v) {
            while (d.nsub - it >= v) {
                const int m = std::min(k, (d.nsub - it) / v);
                // everything of the group before its exchange is posted: passes 1 .. m-1 and the travelling rows of
                // pass m (or the whole pass m without the split) -- one replayable sequence
                const int par0 = par;
                rc = run_or_replay(p, std::make_tuple(v, m, par0, (int)split_ok), [&]() {
                    int q = par0;
                    std::vector<Range> r;
                    for (int i = 1; i <= m; ++i) {
                        const bool last = i == m;
                        pass_ranges(g, v, split_ok && last, m - i, r);
                        const size_t n = (split_ok && last) ? r.size() - 1 : r.size();
                        if (split_ok && last && n == 2 && v >= 3) {
                            // the rows that travel up and the rows that travel down: ONE launch (bit-identical to two)
                            const nsdg_rb_mevp_desc& d = p->d;
                            const int e = (v == 4 ? nsdg_mevp_iterate4_pair : nsdg_mevp_iterate3_pair)(p->ctx, r[0].j0, r[0].j1, r[1].j0, r[1].j1, d.s11[q],
                                d.s12[q], d.s22[q], d.s11[1 - q], d.s12[1 - q], d.s22[1 - q], d.u[q], d.v[q], d.u[1 - q], d.v[1 - q], d.packed, d.pg);
                            if (e != NSDG_OK)
                                return e;
                        } else
                            for (size_t a = 0; a < n; ++a) {
                                const int e = pass_launch(p, v, r[a], q);
                                if (e != NSDG_OK)
                                    return e;
                            }
                        if (!last)
                            q = 1 - q;
                    }
                    return (int)NSDG_OK;
                });
                if (rc != NSDG_OK)
                    return rc;
                par = (m % 2 == 0) ? 1 - par0 : par0; // parity the LAST pass of the group reads
                nsdg_halo* plan = p->rows_plan[1 - par];
                if (plan && (rc = nsdg_halo_start(ctx, plan)) != NSDG_OK)
                    return rc;
                if (split_ok) { // the interior of the last pass overlaps with the exchange
                    pass_ranges(g, v, true, 0, rng);
                    if ((rc = pass_launch(p, v, rng.back(), par)) != NSDG_OK)
                        return rc;
                }
                if (plan && (rc = nsdg_halo_finish(ctx, plan)) != NSDG_OK)
                    return rc;
                par = 1 - par;
                it += v * m;
            }
        }
    }
    const bool split = d.overlap && g.multi() && (g.j1 - g.j0) >= 4 && p->per_pass < 2;
    for (; it < d.nsub; ++it) {
        nsdg_halo* plan = p->per_pass >= 2 ? p->rows_plan[1 - par] : p->node_plan[1 - par];
        if (!split) {
            const int k0 = std::max(g.j0 - 1, 0);
            if ((rc = single_launch(p, k0, g.j0, g.j1, par)) != NSDG_OK)
                return rc;
            if (plan && ((rc = nsdg_halo_start(ctx, plan)) != NSDG_OK || (rc = nsdg_halo_finish(ctx, plan)) != NSDG_OK))
                return rc;
        } else {
            int lo = g.j0, hi = g.j1;
            if (g.has_above()) { // top owned element row -> the two node rows sent upwards
                if ((rc = single_launch(p, g.j1 - 2, g.j1 - 1, g.j1, par)) != NSDG_OK)
                    return rc;
                hi = g.j1 - 1;
            }
            if (g.has_below()) { // bottom owned element row (+ redundant ghost-row stress) -> the node row sent downwards
                if ((rc = single_launch(p, g.j0 - 1, g.j0, g.j0 + 1, par)) != NSDG_OK)
                    return rc;
                lo = g.j0 + 1;
            }
            if (plan && (rc = nsdg_halo_start(ctx, plan)) != NSDG_OK)
                return rc;
            if ((rc = single_launch(p, lo > 0 ? lo - 1 : 0, lo, hi, par)) != NSDG_OK)
                return rc;
            if (plan && (rc = nsdg_halo_finish(ctx, plan)) != NSDG_OK)
                return rc;
        }
        par = 1 - par;
    }
    *parity_out = par;
    return NSDG_OK;
}

} // extern "C"

// =================================================================================================== transport
struct nsdg_rb_transport {
    nsdg_ctx* ctx;
    Geometry g;
    nsdg_rb_transport_desc d;
    int nc;
    bool deep;
    // parity 0: the state is in phi[], t1[] receives the new state; parity 1: the other way round
    nsdg_halo* new_plan[2] = { nullptr, nullptr }; // ghost rows of the array set that receives the new state
    nsdg_halo* t2_plan = nullptr; // shallow ghost zones only: after stage 2
};

extern "C" {

int nsdg_rb_transport_create(nsdg_ctx* ctx, const nsdg_rb_transport_desc* desc, nsdg_rb_transport** out)
{
    NSDG_CHECK_ARG(ctx && desc && out, "null argument");
    *out = nullptr;
    const nsdg_rb_transport_desc& d = *desc;
    Geometry g = { d.nx, d.ny, d.j0, d.j1, d.depth_below, d.depth_above, d.rank_below, d.rank_above };
    int rc = check_geometry(g);
    if (rc != NSDG_OK)
        return rc;
    NSDG_CHECK_ARG(d.order == 2, "the row-block transport driver runs the DG2 / SSP-RK3 scheme");
    NSDG_CHECK_ARG(d.nfields >= 1 && d.nfields <= NSDG_RB_MAX_FIELDS, "nfields out of range");
    for (int f = 0; f < d.nfields; ++f)
        NSDG_CHECK_ARG(d.phi[f] && d.t1[f] && d.t2[f], "null field pointer");
    NSDG_CHECK_ARG(d.vx_dg && d.vy_dg && d.un_x && d.un_y, "null velocity pointer");
    if (d.own_bounds) {
        NSDG_CHECK_ARG(d.nbounds == 0 || d.nbounds == d.nfields, "own bounds: none, or one per advected field");
        for (int f = 0; f < d.nbounds; ++f)
            NSDG_CHECK_ARG(d.bounds[f].lo <= d.bounds[f].hi, "bounds need lo <= hi (hi = +infinity: no upper bound)");
    }
    NSDG_CHECK_ARG(!g.multi() || g.depth_above >= 1, "transport needs a ghost row on both interior sides");
    if (g.multi() && !ctx->comm) {
        nsdg_set_error("nsdg_rb_transport_create: a block with neighbours needs nsdg_comm_init* first");
        return NSDG_ERR_STATE;
    }
    nsdg_rb_transport* p = new nsdg_rb_transport();
    p->ctx = ctx;
    p->g = g;
    p->d = d;
    p->nc = 6;
    p->deep = g.multi() && std::min(g.depth_below, g.depth_above) >= 3;
    if (g.multi()) {
        for (int par = 0; par < 2 && rc == NSDG_OK; ++par) {
            SegList L;
            for (int f = 0; f < d.nfields; ++f)
                add_plane_rows(g, par == 0 ? d.t1[f] : d.phi[f], p->nc, L);
            rc = make_plan(ctx, g, L, &p->new_plan[par]);
        }
        if (rc == NSDG_OK && !p->deep) {
            SegList L;
            for (int f = 0; f < d.nfields; ++f)
                add_plane_rows(g, d.t2[f], p->nc, L);
            rc = make_plan(ctx, g, L, &p->t2_plan);
        }
    }
    if (rc != NSDG_OK) {
        nsdg_rb_transport_destroy(p);
        return rc;
    }
    *out = p;
    return NSDG_OK;
}

int nsdg_rb_transport_destroy(nsdg_rb_transport* p)
{
    if (!p)
        return NSDG_OK;
    nsdg_halo_plan_destroy(p->new_plan[0]);
    nsdg_halo_plan_destroy(p->new_plan[1]);
    nsdg_halo_plan_destroy(p->t2_plan);
    delete p;
    return NSDG_OK;
}

int nsdg_rb_transport_stats(nsdg_ctx* ctx, nsdg_rb_transport* p, nsdg_halo_stats* out, int32_t reset)
{
    NSDG_CHECK_ARG(ctx && p && p->ctx == ctx && out, "plan does not belong to this context");
    return sum_stats(ctx, { p->new_plan[0], p->new_plan[1], p->t2_plan }, out, reset);
}

int nsdg_rb_transport_run(nsdg_ctx* ctx, nsdg_rb_transport* p, double dt, int32_t parity, int32_t* parity_out)
{
    NSDG_CHECK_ARG(ctx && p && p->ctx == ctx && parity_out, "plan does not belong to this context");
    NSDG_CHECK_ARG(parity == 0 || parity == 1, "parity must be 0 or 1");
    NSDG_NEED_GRID(ctx);
    NSDG_CHECK_ARG(ctx->nx == p->g.nx && ctx->ny == p->g.ny, "nsdg_grid_set does not match the plan's local array");
    const Geometry& g = p->g;
    const nsdg_rb_transport_desc& d = p->d;
    // the closure the step runs with: the plan's own bounds, or the context's of this moment (own_bounds = 0).  The kernels read them from
    // the context, so a plan with its own swaps them in for the duration of the call
    struct BoundsSwap {
        nsdg_ctx* ctx;
        int n;
        nsdg_field_bounds b[4];
        bool on;
        ~BoundsSwap()
        {
            if (on) {
                ctx->nbounds = n;
                std::memcpy(ctx->bounds, b, sizeof b);
            }
        }
    } swap { ctx, ctx->nbounds, {}, d.own_bounds != 0 };
    if (swap.on) {
        std::memcpy(swap.b, ctx->bounds, sizeof swap.b);
        ctx->nbounds = d.nbounds;
        for (int f = 0; f < d.nbounds; ++f)
            ctx->bounds[f] = d.bounds[f];
    }
    NSDG_CHECK_ARG(ctx->nbounds == 0 || ctx->nbounds == p->d.nfields,
        "the bounds (the plan's own, or nsdg_transport_bounds_set's) were given for a different number of fields than this plan advances"); // before anything is advanced
    double* const* cur = parity == 0 ? d.phi : d.t1; // the state
    double* const* nxt = parity == 0 ? d.t1 : d.phi; // receives the new state (used as stage buffer 1 on the way)
    auto exchange = [&](nsdg_halo* plan) {
        if (!plan)
            return (int)NSDG_OK;
        const int rc = nsdg_halo_start(ctx, plan);
        return rc != NSDG_OK ? rc : nsdg_halo_finish(ctx, plan);
    };
    int rc;
    if (!g.multi() || p->deep) {
        // ghost zones at least three rows deep (or no neighbour): the three stages as ONE launch on the block's own rows -- the
        // march recomputes the stage values of the rows around them from the ghost rows of the state (bit-identical to the
        // stage launches below) -- and one exchange
        if ((rc = nsdg_transport_step_oop_rows(ctx, d.order, g.j0, g.j1, dt, d.nfields, cur, nxt, d.vx_dg, d.vy_dg, d.un_x, d.un_y)) != NSDG_OK)
            return rc;
    } else {
        // Shu-Osher SSP-RK3: out = a*phi0 + b*(phis + dt L(phis)); a stage reads one element row on each side of its rows
        if ((rc = nsdg_transport_stage(ctx, d.order, g.j0, g.j1, dt, 0.0, 1.0, d.nfields, cur, cur, nxt, d.vx_dg, d.vy_dg, d.un_x, d.un_y)) != NSDG_OK)
            return rc;
        if ((rc = exchange(p->new_plan[parity])) != NSDG_OK)
            return rc;
        if ((rc = nsdg_transport_stage(ctx, d.order, g.j0, g.j1, dt, 0.75, 0.25, d.nfields, cur, nxt, d.t2, d.vx_dg, d.vy_dg, d.un_x, d.un_y)) != NSDG_OK)
            return rc;
        if ((rc = exchange(p->t2_plan)) != NSDG_OK)
            return rc;
        if ((rc = nsdg_transport_stage(ctx, d.order, g.j0, g.j1, dt, 1.0 / 3.0, 2.0 / 3.0, d.nfields, cur, d.t2, nxt, d.vx_dg, d.vy_dg, d.un_x, d.un_y))
            != NSDG_OK)
            return rc;
        // the closure of the step on the block's own rows (the march applies it in its epilogue); the ghost rows receive limited values
        if (ctx->nbounds > 0 && (rc = nsdg_transport_limit(ctx, d.order, g.j0, g.j1, d.nfields, nxt)) != NSDG_OK)
            return rc;
    }
    if ((rc = exchange(p->new_plan[parity])) != NSDG_OK)
        return rc;
    *parity_out = 1 - parity;
    return NSDG_OK;
}

} // extern "C"
