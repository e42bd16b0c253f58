// engine.cpp — host side of the C ABI declared in include/heat_cf.h.
//
// Owns the device-resident training state (packed interaction list, W/G tables), drives the gfx950 kernels of
// ccl_train.hip on one HIP stream and implements the reference's epoch protocol
// (train/engine.cpp:156-160 LR schedule, :294-342 pass over the interactions, :345-347 zero_grad, :378-385 mean loss).
// There is no CPU compute path in this file: if HIP is unusable every entry point reports HEAT_CF_EHIP.
#include "../../include/heat_cf.h"
#include "ccl_train.hpp"
#include "eval_kernels.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace heatcf;

namespace
{
thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                           \
    do                                                                                                          \
    {                                                                                                           \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return fail(_e == hipErrorOutOfMemory ? HEAT_CF_ENOMEM : HEAT_CF_EHIP,                              \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                                     \
    } while (0)

struct EventPair
{
    hipEvent_t a, b;
};
} // namespace

struct heat_cf_engine
{
    heat_cf_config        cfg{};
    std::vector<uint64_t> milestones;
    int                   device = 0;
    hipStream_t           stream = nullptr;
    bool                  own_stream = false;
    bool                  host_mode = false;
    // borrowed host buffers (host mode)
    float* h_user_w = nullptr;
    float* h_item_w = nullptr;
    float* h_w0 = nullptr;
    // device state
    uint2*   d_clicks = nullptr;
    float*   d_user_w = nullptr;
    float*   d_item_w = nullptr;
    float*   d_user_g = nullptr;
    float*   d_item_g = nullptr;
    float*   d_w0 = nullptr;
    uint32_t* d_his = nullptr;
    uint32_t* d_masks = nullptr;
    uint64_t  max_his = 0;
    bool     own_tables = false;
    uint64_t data_rows = 0;
    // scratch
    double*   d_loss_part = nullptr;
    uint32_t  loss_cap = 0;
    double*   d_sums = nullptr; // [0] epoch sum, [1] last range sum
    uint32_t* d_ext_negs = nullptr;
    size_t    ext_cap = 0;
    uint32_t* d_stats = nullptr;
    float*    d_agg_state = nullptr; // per-stream aggregator state carried across the launches of an epoch
    size_t    agg_state_cap = 0;     // streams it holds
    // scalar state
    float    lr = 0.f;
    uint64_t epoch = 0;
    // kernel choice
    int      lpr = 0, ng = 0, nw = 1, aux = 0, upd = 0;
    uint32_t cu_count = 256;
    uint32_t auto_streams = 1;
    bool     tile_resident = false; // random-tile sampler with the tile's weight deltas held in LDS (ccl_train.hip, TS > 1)
    char     kbase[96] = {0};       // variant / policy / streams
    char     kname[112] = {0};      // kbase + what the LAST launch actually ran (tile-in-lds or not)
    // timing
    std::vector<EventPair> ev_free, ev_pending;
    double   kernel_ms = 0.0;
    uint64_t launches = 0;
};

namespace
{
size_t user_bytes(const heat_cf_engine* e) { return (size_t)e->cfg.num_users * e->cfg.emb_dim * sizeof(float); }
size_t item_bytes(const heat_cf_engine* e) { return (size_t)e->cfg.num_items * e->cfg.emb_dim * sizeof(float); }

int validate_cfg(const heat_cf_config* cfg, uint64_t data_rows, int* lpr, int* ng, int* nw)
{
    if (!cfg) return fail(HEAT_CF_EINVAL, "cfg is NULL");
    if (cfg->num_users == 0 || cfg->num_items == 0) return fail(HEAT_CF_EINVAL, "num_users and num_items must be > 0");
    if (cfg->emb_dim == 0 || cfg->emb_dim % 4 != 0 || cfg->emb_dim > 256)
        return fail(HEAT_CF_EUNSUP, "emb_dim must be a multiple of 4 in [4,256] (16-byte row segments per lane)");
    if (cfg->num_negs == 0) return fail(HEAT_CF_EINVAL, "num_negs must be > 0");
    if (cfg->num_users >= 0xFFFFFFFFull || cfg->num_items >= 0xFFFFFFFFull)
        return fail(HEAT_CF_EUNSUP, "ids must fit 32 bits on the device");
    if ((uint64_t)cfg->num_items * cfg->emb_dim * 4ull >= (1ull << 32))
        return fail(HEAT_CF_EUNSUP, "item table must be < 4 GiB (32-bit buffer offsets)");
    if (cfg->n_milestones == 0 || cfg->milestones == nullptr)
        return fail(HEAT_CF_EINVAL, "milestones must hold at least one epoch (engine.cpp:159 reads milestones[0])");
    if (cfg->n_milestones == 1 && cfg->milestones[0] == 0)
        return fail(HEAT_CF_EINVAL, "milestones[0] must be > 0 (optimizer.cpp:26 computes epoch % step_size)");
    if (cfg->neg_sampler != 0 && cfg->neg_sampler != 1) return fail(HEAT_CF_EINVAL, "neg_sampler must be 0 or 1");
    if (cfg->neg_sampler == 1 && (cfg->tile_size == 0 || cfg->tile_size > 0xFFFFFFFFull || cfg->refresh_interval == 0))
        return fail(HEAT_CF_EINVAL, "random-tile sampler needs tile_size > 0 and refresh_interval > 0");
    // behaviour aggregation sizes itself by the single-wave table wherever one holds num_negs (the multi-wave table takes
    // over beyond that) and then spreads that capacity over up to 4 waves: history gather and d x d product split over them
    if (cfg->use_aggregator && pick_variant((uint32_t)cfg->emb_dim, (uint32_t)cfg->num_negs, true, lpr, ng, nw))
        widen_for_aggregator(*lpr, ng, nw);
    else if (!pick_variant((uint32_t)cfg->emb_dim, (uint32_t)cfg->num_negs, false, lpr, ng, nw))
        return fail(HEAT_CF_EUNSUP, "no compiled kernel variant for this (emb_dim, num_negs)");
    if (data_rows >= (1ull << 40)) return fail(HEAT_CF_EINVAL, "data_rows too large");
    return HEAT_CF_OK;
}

// ---- launch plan: pure host logic (no HIP calls), shared by engine creation and heat_cf_plan() ----------------------
struct Plan
{
    int      lpr = 0, ng = 0, nw = 1;
    uint32_t coherence = 0;   // resolved HEAT_CF_COHERENCE_*
    uint32_t streams = 1;     // workgroups walking the list concurrently
    uint32_t cap_items = 0, cap_users = 0;
    uint32_t update_mode = 0; // resolved HEAT_CF_UPDATE_* (or the raw 16.. form)
    uint32_t upd_bits = 0;
    const char* binding = "";  // which bound set `streams`
    const char* regime = "";   // how the step size (l_r, clip_val) relates to what the bounds were measured at
};

// fill = workgroups the chip can keep resident for the chosen variant (from the occupancy query; 0 = unknown: caps only)
// history_rows = rows an interaction reads besides user / positive / negatives (max_his with behaviour aggregation)
// cus = compute units of the device (256 on MI355X)
int make_plan(const heat_cf_config* cfg, uint64_t data_rows, uint64_t fill, Plan* p, uint64_t history_rows = 0, uint32_t cus = 256)
{
    int rc = validate_cfg(cfg, data_rows, &p->lpr, &p->ng, &p->nw);
    if (rc) return rc;
    const uint32_t coh = cfg->coherence == HEAT_CF_COHERENCE_DEFAULT ? HEAT_CF_COHERENCE_DEVICE : cfg->coherence;
    if (coh != HEAT_CF_COHERENCE_PLAIN && coh != HEAT_CF_COHERENCE_DEVICE) return fail(HEAT_CF_EINVAL, "bad coherence");
    p->coherence = coh;
    // Streams = interactions in flight.  Every one of them computes its gradient from rows that the others are changing,
    // so their number is bounded by what was validated against the oracle's Recall@20 / NDCG@20 (+-1e-3), expressed as
    // in-flight touches per item row, streams x (num_negs + 1) / num_items:
    //   * <= 17 rows per interaction (AmazonBooks / Gowalla-PR1 configs): 0.56 (3017 streams at AmazonBooks shape) with
    //     positives written back by float atomics and negatives by plain stores;
    //   * more rows per interaction (Yelp18 config: 65): 0.45 (256 streams at Yelp18 shape, clustered graph, three seeds;
    //     400 streams = 0.68 is at the edge) — and only with the "late re-read" write-back of the negative rows; with
    //     the plain store (read-modify-write window = the whole interaction) the same shape holds the tolerance only up
    //     to 0.15 (85 streams; profiles/r02_yelp18_policy_sweep.txt);
    //   * behaviour aggregation reads up to max_his history rows, which it never writes: they count at half weight
    //     (AmazonBooks shape: 765 streams by this bound; measured: Recall@20 equal to the sequentially consistent model of
    //     the reference at 438, 512 and 768 streams, -2e-3 at 1024; profiles/r03_accl_worker_count.txt).  In this mode the
    //     stream count also moves the LOSS CURVE — in the reference's own algorithm, DESIGN.md section 3 — so the count the
    //     chip holds resident (512 four-wave streams at AmazonBooks shape) is used, not more.
    const double rows_per_interaction = (double)(cfg->num_negs + 1);
    const bool wide = rows_per_interaction > 17.0;
    const bool can_reread = wide && !cfg->use_aggregator && coh == HEAT_CF_COHERENCE_DEVICE;
    double in_flight = 0.56;
    if (cfg->use_aggregator) in_flight = 0.56 * std::min(1.0, 17.0 / (rows_per_interaction + 0.5 * (double)history_rows));
    else if (wide) in_flight = can_reread ? 0.45 : 0.15;
    p->cap_items = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (uint64_t)(in_flight * (double)cfg->num_items / rows_per_interaction));
    // A stream walks at least 256 consecutive interactions (half a `schedule(dynamic,512)` chunk of the reference,
    // train/engine.cpp:327): shorter slices cut the runs of heavy users into many pieces that train the same user row
    // concurrently from one starting value.  Measured with 8 user shards of the AmazonBooks-shaped graph
    // (profiles/r02_sim_shards.txt): 1024 streams per shard (290 interactions each) hold Recall@20 / NDCG@20 within 1e-3 of
    // single-engine training, 3017 streams per shard (98 each) lose 4e-3.
    // Wide interactions (the Yelp18 / Gowalla yaml: 64 negatives, clip_val 0.1) also bound the SHARE of an epoch that is in
    // flight: every row's persistent gradient G accumulates over the epoch, and streams / data_rows is the staleness of a
    // row relative to the updates it receives per epoch.  Clustered graphs at the yaml's 8 epochs
    // (profiles/r02_yelp18_policy_sweep.txt): Yelp18 shape (1.24 M interactions) holds +-1e-3 at 256 streams and is at
    // the edge at 400; the same tables with 0.81 M interactions and the Gowalla shape (0.81 M) hold it at 128-170 and lose
    // 1.2e-3 - 2e-3 at 200-256; with 2 M interactions 256 holds and 400 is at the edge (+1.2e-3).  Bound: a stream walks at
    // least 5600 interactions of such an epoch (Yelp18 shape: 220 streams, Gowalla shape: 144).
    // Every bound above was measured at the reference's yaml step size, l_r 0.01 (clip_val 1.0 with <= 17 rows, 0.1 with 65).
    // Round 3 (profiles/r03_lr_regime.txt): AmazonBooks shape holds +-1e-3 at 3017 streams for l_r 0.03 and 0.1 as well;
    // the Yelp18 shape at l_r 0.03 loses 2e-3 in one of two seeds at 220 streams and not at 110.  So: wide interactions walk
    // l_r / 0.01 times more interactions per stream when l_r is larger than measured (an extrapolation, reported as such);
    // anything beyond ten times the yaml step is reported as outside what was measured and planned the same way.
    const double lr_ratio = (double)cfg->l_r / 0.01;
    const bool wide_plain = wide && !cfg->use_aggregator;
    p->regime = lr_ratio <= 1.0001 ? "measured (l_r <= 0.01)"
                : (!wide_plain && lr_ratio <= 10.001) ? "measured (l_r <= 0.1, <= 17 rows per interaction)"
                : (wide_plain && lr_ratio <= 3.001)  ? "extrapolated: interactions per stream scaled by l_r / 0.01 (measured at l_r 0.03: 110 streams hold, 220 do not)"
                                                      : "outside the measured range of l_r: bounds extrapolated, validate Recall before relying on it";
    const uint64_t min_slice = wide_plain ? (uint64_t)(5600.0 * std::max(1.0, lr_ratio)) : 256;
    p->cap_users = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, std::max<uint64_t>(1, data_rows / min_slice));
    uint64_t streams = std::min<uint64_t>(p->cap_items, p->cap_users);
    p->binding = p->cap_items <= p->cap_users ? "in-flight touches per item row" : "interactions per stream";
    if (fill && fill < streams)
    {
        streams = fill;
        p->binding = "resident workgroups";
    }
    // a few workgroups more than a whole number per compute unit would make those units the tail of the launch — for the
    // multi-wave workgroups, which run at what a compute unit sustains; a single-wave stream between 1x and 4x the CU
    // count is latency-bound and simply scales with the count.  Round 3: the four-wave aggregation workgroups belong to the
    // first kind (AmazonBooks shape: 438 streams 25.8 ms per epoch, 512 = two per CU 22.4, 640 31.3, 768 26.1)
    if (cus && p->nw > 1 && streams > cus && streams < 4ull * cus) streams -= streams % cus;
    if (streams < 1) streams = 1;
    if (cfg->num_streams)
    {
        streams = cfg->num_streams;
        p->binding = "num_streams";
    }
    if (cfg->flags & HEAT_CF_FLAG_SERIAL)
    {
        streams = 1;
        p->binding = "serial";
    }
    p->streams = (uint32_t)streams;
    uint32_t um = cfg->update_mode;
    if (um == HEAT_CF_UPDATE_DEFAULT) um = HEAT_CF_UPDATE_AUTO;
    if (coh != HEAT_CF_COHERENCE_DEVICE && um == HEAT_CF_UPDATE_AUTO) um = HEAT_CF_UPDATE_OVERWRITE;
    if (um == HEAT_CF_UPDATE_AUTO)
    {
        const double touches = (double)streams * rows_per_interaction / (double)cfg->num_items;
        if (cfg->use_aggregator || !wide) um = touches <= 0.6 ? HEAT_CF_UPDATE_ATOMIC_POS : HEAT_CF_UPDATE_ATOMIC_WG;
        else if (touches <= 0.15) um = HEAT_CF_UPDATE_ATOMIC_POS;
        else um = can_reread ? HEAT_CF_UPDATE_REREAD_POS : HEAT_CF_UPDATE_ATOMIC_WG;
    }
    uint32_t bits;
    if (um == HEAT_CF_UPDATE_OVERWRITE) bits = 0u;
    else if (um == HEAT_CF_UPDATE_ATOMIC_W) bits = 0x5u;
    else if (um == HEAT_CF_UPDATE_ATOMIC_WG) bits = 0xFu;
    else if (um == HEAT_CF_UPDATE_ATOMIC_POS) bits = 0xCu;
    else if (um == HEAT_CF_UPDATE_REREAD_POS) bits = 0x1Cu;
    else if (um >= 16u && um < 48u) bits = um - 16u;
    else return fail(HEAT_CF_EINVAL, "bad update_mode");
    if (bits != 0u && coh != HEAT_CF_COHERENCE_DEVICE)
        return fail(HEAT_CF_EINVAL, "atomic / re-read update modes need HEAT_CF_COHERENCE_DEVICE");
    if ((bits & 0x10u) && (bits & 0x3u))
        return fail(HEAT_CF_EINVAL, "the late re-read (bit 4) applies to plain negative-row stores: bits 0-1 must be clear");
    if ((bits & 0x10u) && cfg->use_aggregator) return fail(HEAT_CF_EUNSUP, "the late re-read write-back is not built for behaviour aggregation");
    // The reference's literal overwrite loses updates in proportion to the number of concurrent workers (Recall@20 0.099 vs
    // 0.209 at AmazonBooks shape with ~3000 streams, profiles/r01_recall_parity_overwrite_modes.txt): without an explicit
    // num_streams it runs at a worker count the reference itself could have (64 OpenMP threads in the paper).
    if (bits == 0u && !cfg->num_streams && !(cfg->flags & HEAT_CF_FLAG_SERIAL) && p->streams > 64u)
    {
        p->streams = 64u;
        p->binding = "overwrite policy: a worker count the reference could have";
    }
    p->update_mode = um;
    p->upd_bits = bits;
    return HEAT_CF_OK;
}

// the random-tile sampler keeps its tile in LDS when the caller asks for it (HEAT_CF_FLAG_TILE_LDS) and this
// holds (ccl_train.hip, TS > 1).  Opt-in since round 3: a tile shared by the 12 streams of a workgroup, with its negative-row
// weight updates private to that workgroup until the end of the launch, is a different sampler regime from the reference's
// one tile per worker written through to the table (DESIGN.md section 3 "Tile in LDS").
bool tile_fits_lds(const heat_cf_config* cfg, const Plan& p)
{
    return cfg->neg_sampler == 1 && (cfg->flags & HEAT_CF_FLAG_SAMPLING_CALL) && (cfg->flags & HEAT_CF_FLAG_TILE_LDS) &&
           !(cfg->flags & (HEAT_CF_FLAG_SERIAL | HEAT_CF_FLAG_TILE_GLOBAL)) && p.nw == 1 && p.lpr <= 16 && p.ng <= 4 &&
           !cfg->use_aggregator && p.upd_bits == 0xCu && p.coherence == HEAT_CF_COHERENCE_DEVICE &&
           cfg->tile_size <= 0xFFFFFFFFull && cfg->tile_size * cfg->emb_dim * sizeof(float) <= 128u * 1024u;
}

int common_init(heat_cf_engine* e, const heat_cf_config* cfg, uint64_t data_rows, void* stream, uint64_t history_rows = 0)
{
    e->cfg = *cfg;
    e->milestones.assign(cfg->milestones, cfg->milestones + cfg->n_milestones);
    e->cfg.milestones = e->milestones.data();
    e->data_rows = data_rows;
    e->lr = cfg->l_r; // optimizers/optimizer.cpp:13
    e->epoch = 0;     // train/engine.cpp:19
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev <= 0)
        return fail(HEAT_CF_EHIP, "no usable HIP device (the engine has no CPU fallback)");
    if (cfg->device >= 0)
    {
        if (cfg->device >= ndev) return fail(HEAT_CF_EINVAL, "device ordinal out of range");
        HIP_TRY(hipSetDevice(cfg->device));
        e->device = cfg->device;
    }
    else
    {
        HIP_TRY(hipGetDevice(&e->device));
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, e->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HEAT_CF_EHIP, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    e->cu_count = (uint32_t)prop.multiProcessorCount;
    if (stream || (cfg->flags & HEAT_CF_FLAG_NULL_STREAM))
    {
        e->stream = (hipStream_t)stream;
        e->own_stream = false;
    }
    else
    {
        HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    }
    Plan plan;
    int prc = make_plan(cfg, data_rows, 0, &plan, history_rows);       // variant + coherence first: the occupancy query needs them
    if (prc) return prc;
    e->aux = plan.coherence == HEAT_CF_COHERENCE_DEVICE ? 16 : 0;
    // resident workgroups per CU as the runtime reports them for this variant (register / LDS limited)
    int per_cu = query_blocks_per_cu(e->lpr, e->ng, e->nw, e->aux, cfg->use_aggregator != 0, (uint32_t)cfg->emb_dim);
    if (per_cu < 1) per_cu = 1;
    uint64_t fill = (uint64_t)e->cu_count * (uint64_t)per_cu;
    // the tile-resident kernel (below) has its own shape: one workgroup of TILE_STREAMS streams per compute unit
    if (tile_fits_lds(cfg, plan)) fill = (uint64_t)e->cu_count * (uint64_t)TILE_STREAMS;
    prc = make_plan(cfg, data_rows, fill, &plan, history_rows, e->cu_count);
    if (prc) return prc;
    e->auto_streams = plan.streams;
    e->upd = (int)plan.upd_bits;
    // a stream count that rests on an asynchrony bound measured at another step size is said out loud, once per engine
    if (std::strncmp(plan.regime, "measured", 8) != 0 && !cfg->num_streams && !(cfg->flags & HEAT_CF_FLAG_SERIAL))
        std::fprintf(stderr, "heat_cf: note: l_r %g: %s (streams=%u, bound: %s)\n", (double)cfg->l_r, plan.regime,
                     (unsigned)plan.streams, plan.binding);
    // with the random-tile sampler (its sampling() call) the tile lives in LDS when it fits: 12 single-wave
    // streams per workgroup share tile_size x emb_dim fp32 of accumulated weight deltas (<= 128 KB)
    e->tile_resident = tile_fits_lds(cfg, plan);
    std::snprintf(e->kbase, sizeof(e->kbase), "ccl_train_kernel<%d,%d,%d,%d>/upd=0x%x/streams=%u", e->lpr, e->ng, e->aux, e->nw,
                  (unsigned)e->upd, (unsigned)plan.streams);
    // until a launch says otherwise the name carries the plan; heat_cf_train_range rewrites it with what it launched (a
    // tile-resident engine falls back to the table-writing kernel for caller-fed negatives and for windows longer than
    // refresh_interval calls per stream)
    std::snprintf(e->kname, sizeof(e->kname), "%s%s", e->kbase, e->tile_resident ? "/tile-in-lds" : "");
    HIP_TRY(hipMalloc(&e->d_sums, 2 * sizeof(double)));
    HIP_TRY(hipMemsetAsync(e->d_sums, 0, 2 * sizeof(double), e->stream));
    HIP_TRY(hipMalloc(&e->d_stats, 4 * sizeof(uint32_t)));
    return HEAT_CF_OK;
}

int ensure_loss_part(heat_cf_engine* e, uint32_t grid)
{
    if (grid <= e->loss_cap) return HEAT_CF_OK;
    if (e->d_loss_part)
    {
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipFree(e->d_loss_part));
        e->d_loss_part = nullptr;
    }
    HIP_TRY(hipMalloc(&e->d_loss_part, (size_t)grid * sizeof(double)));
    e->loss_cap = grid;
    return HEAT_CF_OK;
}

int get_events(heat_cf_engine* e, EventPair* p)
{
    if (e->ev_pending.size() >= 1024)
    {
        // bound the pool: fold the oldest timings into the accumulator
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (auto& q : e->ev_pending)
        {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, q.a, q.b));
            e->kernel_ms += ms;
            e->ev_free.push_back(q);
        }
        e->ev_pending.clear();
    }
    if (!e->ev_free.empty())
    {
        *p = e->ev_free.back();
        e->ev_free.pop_back();
        return HEAT_CF_OK;
    }
    HIP_TRY(hipEventCreate(&p->a));
    HIP_TRY(hipEventCreate(&p->b));
    return HEAT_CF_OK;
}

// grid geometry: `streams` sequential walkers, each over a contiguous run of interactions (multiple of 64)
void geometry(const heat_cf_engine* e, uint64_t n, uint64_t* per_block, uint32_t* grid)
{
    if (e->cfg.flags & HEAT_CF_FLAG_SERIAL)
    {
        *per_block = ((n + 63) / 64) * 64;
        *grid = 1;
        return;
    }
    uint64_t streams = (uint64_t)e->auto_streams;
    uint64_t pb = (n + streams - 1) / streams;
    pb = ((pb + 63) / 64) * 64;
    if (pb == 0) pb = 64;
    *per_block = pb;
    *grid = (uint32_t)((n + pb - 1) / pb);
}

TrainArgs make_args(const heat_cf_engine* e, uint64_t begin, uint64_t end)
{
    TrainArgs a{};
    a.clicks = e->d_clicks;
    a.user_w = e->d_user_w;
    a.user_g = e->d_user_g;
    a.item_w = e->d_item_w;
    a.item_g = e->d_item_g;
    a.begin = begin;
    a.end = end;
    a.num_items = (uint32_t)e->cfg.num_items;
    a.num_negs = (uint32_t)e->cfg.num_negs;
    a.emb_dim = (uint32_t)e->cfg.emb_dim;
    a.row_bytes = (uint32_t)e->cfg.emb_dim * 4u;
    a.item_bytes = (uint32_t)(e->cfg.num_items * e->cfg.emb_dim * 4ull);
    a.sampling_call = (e->cfg.flags & HEAT_CF_FLAG_SAMPLING_CALL) ? 1u : 0u;
    a.exact_order = (e->cfg.flags & HEAT_CF_FLAG_SERIAL) ? 1u : 0u;
    a.tile_size = e->cfg.neg_sampler == 1 ? (uint32_t)e->cfg.tile_size : 0u;
    a.refresh_interval = (uint32_t)std::max<uint64_t>(1, e->cfg.refresh_interval);
    a.upd_bits = (uint32_t)e->upd;
    a.align_cap = (e->upd & 0xF) == 0 ? 4096u : 0u; // overwrite mode keeps a user's run inside one stream; atomic modes need not
    a.lr = e->lr;
    a.clip = e->cfg.clip_val;
    a.key = epoch_key(e->cfg.seed, e->epoch);
    a.sample_base = e->cfg.sample_index_base;
    a.ext_negs = nullptr;
    a.neg_out = nullptr;
    a.loss_part = e->d_loss_part;
    a.agg = e->cfg.use_aggregator ? 1u : 0u;
    a.max_his = (uint32_t)e->max_his;
    a.his = e->d_his;
    a.masks = e->d_masks;
    a.w0 = e->d_w0;
    a.agg_lr = e->cfg.l_r; // behavior_aggregators.cpp:38: frozen at the config value, not the scheduled lr
    a.agg_w0_lds = (e->cfg.use_aggregator && agg_w0_fits_lds((uint32_t)e->cfg.emb_dim, e->lpr, e->nw)) ? 1u : 0u;
    a.agg_state = e->d_agg_state;
    return a;
}

void destroy_impl(heat_cf_engine* e)
{
    if (!e) return;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto& q : e->ev_pending) { (void)hipEventDestroy(q.a); (void)hipEventDestroy(q.b); }
    for (auto& q : e->ev_free) { (void)hipEventDestroy(q.a); (void)hipEventDestroy(q.b); }
    if (e->own_tables)
    {
        (void)hipFree(e->d_user_w);
        (void)hipFree(e->d_item_w);
        (void)hipFree(e->d_w0);
    }
    (void)hipFree(e->d_his);
    (void)hipFree(e->d_masks);
    (void)hipFree(e->d_user_g);
    (void)hipFree(e->d_item_g);
    (void)hipFree(e->d_clicks);
    (void)hipFree(e->d_loss_part);
    (void)hipFree(e->d_sums);
    (void)hipFree(e->d_ext_negs);
    (void)hipFree(e->d_stats);
    (void)hipFree(e->d_agg_state);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}
} // namespace

extern "C" {

// used by ingest.cpp (host-only translation unit) to report errors through heat_cf_last_error()
int heat_cf_set_error_(int code, const char* msg) { return fail(code, msg ? msg : ""); }

int heat_cf_abi_version(void) { return HEAT_CF_ABI_VERSION; }

const char* heat_cf_last_error(void) { return g_last_error.c_str(); }

int heat_cf_plan(const heat_cf_config* cfg, uint64_t data_rows, uint64_t resident_workgroups, char* out, uint64_t out_bytes)
{
    if (!cfg || !out || out_bytes == 0) return fail(HEAT_CF_EINVAL, "cfg / out is NULL");
    Plan p;
    int rc = make_plan(cfg, data_rows, resident_workgroups, &p);
    if (rc) return rc;
    const char* um = p.update_mode == HEAT_CF_UPDATE_OVERWRITE ? "OVERWRITE"
                   : p.update_mode == HEAT_CF_UPDATE_ATOMIC_W ? "ATOMIC_W"
                   : p.update_mode == HEAT_CF_UPDATE_ATOMIC_WG ? "ATOMIC_WG"
                   : p.update_mode == HEAT_CF_UPDATE_ATOMIC_POS ? "ATOMIC_POS"
                   : p.update_mode == HEAT_CF_UPDATE_REREAD_POS ? "REREAD_POS" : "RAW";
    const int n = std::snprintf(out, (size_t)out_bytes,
                                "{\"lanes_per_row\": %d, \"groups_per_wave\": %d, \"waves_per_workgroup\": %d, "
                                "\"negative_capacity\": %d, \"coherence\": \"%s\", \"streams\": %u, \"cap_items\": %u, "
                                "\"cap_users\": %u, \"binding\": \"%s\", \"regime\": \"%s\", \"update_mode\": \"%s\", \"update_bits\": %u, \"tile_in_lds\": %s}",
                                p.lpr, p.ng, p.nw, p.ng * (64 / p.lpr) * p.nw,
                                p.coherence == HEAT_CF_COHERENCE_DEVICE ? "device" : "plain", p.streams, p.cap_items,
                                p.cap_users, p.binding, p.regime, um, p.upd_bits, tile_fits_lds(cfg, p) ? "true" : "false");
    if (n < 0 || (uint64_t)n >= out_bytes) return fail(HEAT_CF_EINVAL, "out buffer too small");
    return HEAT_CF_OK;
}

int heat_cf_device_count(void)
{
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) return fail(HEAT_CF_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(err));
    return n;
}

static int aggregator_limits(const heat_cf_config* cfg, uint64_t max_his, int lpr, int nw)
{
    (void)cfg; (void)lpr; (void)nw;
    if (max_his == 0 || max_his > 256) return fail(HEAT_CF_EUNSUP, "behaviour aggregation supports 1 <= max_his <= 256");
    return HEAT_CF_OK;
}

int heat_cf_engine_create(const heat_cf_config* cfg, const uint64_t* clicks, uint64_t data_rows, const uint64_t* his,
                          uint64_t max_his, const uint64_t* masks, float* user_w, float* item_w, float* w0,
                          heat_cf_engine** out)
{
    if (!out) return fail(HEAT_CF_EINVAL, "out is NULL");
    *out = nullptr;
    int lpr = 0, ng = 0, nw = 1;
    int rc = validate_cfg(cfg, data_rows, &lpr, &ng, &nw);
    if (rc) return rc;
    if (!clicks && data_rows) return fail(HEAT_CF_EINVAL, "clicks is NULL");
    std::vector<uint32_t> his32, masks32;
    if (cfg->use_aggregator)
    {
        if (!his || !masks || !w0) return fail(HEAT_CF_EINVAL, "use_aggregator needs historical_items, masks and aggregator weights");
        rc = aggregator_limits(cfg, max_his, lpr, nw);
        if (rc) return rc;
        try { his32.resize(cfg->num_users * max_his); masks32.resize(cfg->num_users); }
        catch (const std::bad_alloc&) { return fail(HEAT_CF_ENOMEM, "host allocation failed"); }
        for (uint64_t u = 0; u < cfg->num_users; ++u)
        {
            const uint64_t h = masks[u];
            if (h > max_his) return fail(HEAT_CF_EINVAL, "masks[" + std::to_string(u) + "] exceeds max_his");
            masks32[u] = (uint32_t)h;
            for (uint64_t k = 0; k < max_his; ++k)
            {
                const uint64_t it = his[u * max_his + k];
                if (k < h && it >= cfg->num_items) return fail(HEAT_CF_EINVAL, "historical_items holds an id out of range");
                his32[u * max_his + k] = k < h ? (uint32_t)it : 0u;
            }
        }
        // behavior_aggregators.cpp:63 divides by masks[u]: a user with interactions but an empty history would poison
        // its row with inf/NaN (the reference's datasets never produce one, README.md:83-86)
        for (uint64_t i = 0; i < data_rows; ++i)
            if (clicks[2 * i] < cfg->num_users && masks32[clicks[2 * i]] == 0)
                return fail(HEAT_CF_EINVAL, "user " + std::to_string(clicks[2 * i]) + " has interactions but masks == 0");
    }
    if (!user_w || !item_w) return fail(HEAT_CF_EINVAL, "user_w / item_w is NULL");
    // range check + pack to u32 pairs (the reference performs no bounds checks; on a GPU an out-of-range id
    // is a memory fault, so it is rejected here)
    std::vector<uint2> packed;
    try { packed.resize(data_rows); }
    catch (const std::bad_alloc&) { return fail(HEAT_CF_ENOMEM, "host allocation failed"); }
    for (uint64_t i = 0; i < data_rows; ++i)
    {
        const uint64_t u = clicks[2 * i], it = clicks[2 * i + 1];
        if (u >= cfg->num_users || it >= cfg->num_items)
            return fail(HEAT_CF_EINVAL, "click_dataset row " + std::to_string(i) + " has an id out of range");
        packed[i] = make_uint2((uint32_t)u, (uint32_t)it);
    }
    heat_cf_engine* e = new (std::nothrow) heat_cf_engine();
    if (!e) return fail(HEAT_CF_ENOMEM, "host allocation failed");
    e->lpr = lpr;
    e->ng = ng;
    e->nw = nw;
    e->host_mode = true;
    e->h_user_w = user_w;
    e->h_item_w = item_w;
    e->h_w0 = w0;
#define CREATE_TRY(expr)            \
    do                              \
    {                               \
        int _rc = (expr);           \
        if (_rc)                    \
        {                           \
            destroy_impl(e);        \
            return _rc;             \
        }                           \
    } while (0)
    CREATE_TRY(common_init(e, cfg, data_rows, nullptr, cfg->use_aggregator ? max_his : 0));
    auto body = [&]() -> int {
        e->own_tables = true;
        HIP_TRY(hipMalloc(&e->d_user_w, std::max<size_t>(user_bytes(e), 16)));
        HIP_TRY(hipMalloc(&e->d_item_w, std::max<size_t>(item_bytes(e), 16)));
        HIP_TRY(hipMalloc(&e->d_user_g, std::max<size_t>(user_bytes(e), 16)));
        HIP_TRY(hipMalloc(&e->d_item_g, std::max<size_t>(item_bytes(e), 16)));
        HIP_TRY(hipMalloc(&e->d_clicks, std::max<size_t>(data_rows * sizeof(uint2), 16)));
        HIP_TRY(hipMemcpyAsync(e->d_user_w, user_w, user_bytes(e), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_item_w, item_w, item_bytes(e), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_clicks, packed.data(), data_rows * sizeof(uint2), hipMemcpyHostToDevice, e->stream));
        // embeddings/embedding.cpp:12-13 + memory/array.hpp:22-24: owned, zero-initialised gradient tables
        HIP_TRY(hipMemsetAsync(e->d_user_g, 0, user_bytes(e), e->stream));
        HIP_TRY(hipMemsetAsync(e->d_item_g, 0, item_bytes(e), e->stream));
        if (cfg->use_aggregator)
        {
            e->max_his = max_his;
            const size_t w0b = (size_t)cfg->emb_dim * cfg->emb_dim * sizeof(float);
            HIP_TRY(hipMalloc(&e->d_w0, w0b));
            HIP_TRY(hipMalloc(&e->d_his, his32.size() * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&e->d_masks, masks32.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpyAsync(e->d_w0, w0, w0b, hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(e->d_his, his32.data(), his32.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(e->d_masks, masks32.data(), masks32.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
        return HEAT_CF_OK;
    };
    CREATE_TRY(body());
    *out = e;
    return HEAT_CF_OK;
}

int heat_cf_engine_create_device(const heat_cf_config* cfg, const void* d_clicks, uint64_t data_rows, const void* d_his,
                                 uint64_t max_his, const void* d_masks, void* d_user_w, void* d_item_w, void* d_w0,
                                 void* stream, heat_cf_engine** out)
{
    if (!out) return fail(HEAT_CF_EINVAL, "out is NULL");
    *out = nullptr;
    int lpr = 0, ng = 0, nw = 1;
    int rc = validate_cfg(cfg, data_rows, &lpr, &ng, &nw);
    if (rc) return rc;
    if (!d_clicks && data_rows) return fail(HEAT_CF_EINVAL, "d_clicks is NULL");
    if (!d_user_w || !d_item_w) return fail(HEAT_CF_EINVAL, "d_user_w / d_item_w is NULL");
    if (cfg->use_aggregator)
    {
        if (!d_his || !d_masks || !d_w0) return fail(HEAT_CF_EINVAL, "use_aggregator needs historical_items, masks and aggregator weights");
        rc = aggregator_limits(cfg, max_his, lpr, nw);
        if (rc) return rc;
        if ((uintptr_t)d_w0 & 15u) return fail(HEAT_CF_EINVAL, "aggregator weights must be 16-byte aligned");
    }
    if (((uintptr_t)d_user_w | (uintptr_t)d_item_w) & 15u) return fail(HEAT_CF_EINVAL, "tables must be 16-byte aligned");
    heat_cf_engine* e = new (std::nothrow) heat_cf_engine();
    if (!e) return fail(HEAT_CF_ENOMEM, "host allocation failed");
    e->lpr = lpr;
    e->ng = ng;
    e->nw = nw;
    e->host_mode = false;
    CREATE_TRY(common_init(e, cfg, data_rows, stream, cfg->use_aggregator ? max_his : 0));
    auto body = [&]() -> int {
        e->own_tables = false;
        e->d_user_w = (float*)d_user_w;
        e->d_item_w = (float*)d_item_w;
        e->d_w0 = (float*)d_w0;
        HIP_TRY(hipMalloc(&e->d_user_g, std::max<size_t>(user_bytes(e), 16)));
        HIP_TRY(hipMalloc(&e->d_item_g, std::max<size_t>(item_bytes(e), 16)));
        HIP_TRY(hipMalloc(&e->d_clicks, std::max<size_t>(data_rows * sizeof(uint2), 16)));
        HIP_TRY(hipMemsetAsync(e->d_user_g, 0, user_bytes(e), e->stream));
        HIP_TRY(hipMemsetAsync(e->d_item_g, 0, item_bytes(e), e->stream));
        HIP_TRY(hipMemsetAsync(e->d_stats, 0, 4 * sizeof(uint32_t), e->stream));
        HIP_TRY(launch_pack_clicks((const uint64_t*)d_clicks, e->d_clicks, data_rows, e->d_stats, e->stream));
        if (cfg->use_aggregator)
        {
            e->max_his = max_his;
            HIP_TRY(hipMalloc(&e->d_his, std::max<size_t>(cfg->num_users * max_his * sizeof(uint32_t), 16)));
            HIP_TRY(hipMalloc(&e->d_masks, std::max<size_t>(cfg->num_users * sizeof(uint32_t), 16)));
            HIP_TRY(launch_pack_history((const uint64_t*)d_his, (const uint64_t*)d_masks, e->d_his, e->d_masks, cfg->num_users,
                                        (uint32_t)max_his, cfg->num_items, e->d_clicks, data_rows, e->d_stats + 3, e->stream));
        }
        uint32_t stats[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(stats, e->d_stats, sizeof(stats), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (data_rows && (stats[2] || stats[0] >= cfg->num_users || stats[1] >= cfg->num_items))
            return fail(HEAT_CF_EINVAL, "click_dataset holds an id out of range");
        if (stats[3] & 1u) return fail(HEAT_CF_EINVAL, "masks holds a length above max_his");
        if (stats[3] & 2u) return fail(HEAT_CF_EINVAL, "historical_items holds an id out of range");
        if (stats[3] & 4u) return fail(HEAT_CF_EINVAL, "a user has interactions but masks == 0");
        return HEAT_CF_OK;
    };
    CREATE_TRY(body());
    *out = e;
    return HEAT_CF_OK;
}

void heat_cf_engine_destroy(heat_cf_engine* e) { destroy_impl(e); }

int heat_cf_begin_epoch(heat_cf_engine* e)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    // train/engine.cpp:156-160 with optimizers/optimizer.cpp:24-38 (gamma = 0.1)
    if (e->milestones.size() > 1)
    {
        if (std::find(e->milestones.begin(), e->milestones.end(), e->epoch) != e->milestones.end()) e->lr = e->lr * 0.1f;
    }
    else
    {
        const uint64_t step = e->milestones[0];
        if (e->epoch > 0 && e->epoch % step == 0) e->lr = e->lr * 0.1f;
    }
    HIP_TRY(hipMemsetAsync(e->d_sums, 0, 2 * sizeof(double), e->stream));
    // the reference builds its per-worker aggregator inside the epoch's parallel region (train/engine.cpp:313-318):
    // counter and unflushed pairs start from zero every epoch
    if (e->d_agg_state)
        HIP_TRY(hipMemsetAsync(e->d_agg_state, 0, e->agg_state_cap * agg_state_floats(e->lpr) * sizeof(float), e->stream));
    return HEAT_CF_OK;
}

int heat_cf_train_range(heat_cf_engine* e, uint64_t begin, uint64_t end, const uint64_t* neg_ids, double* loss_sum)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    if (begin > end || end > e->data_rows) return fail(HEAT_CF_EINVAL, "interaction range out of bounds");
    HIP_TRY(hipSetDevice(e->device));
    const uint64_t n = end - begin;
    if (n == 0)
    {
        if (loss_sum) *loss_sum = 0.0;
        return HEAT_CF_OK;
    }
    uint64_t per_block = 0;
    uint32_t grid = 0;
    geometry(e, n, &per_block, &grid);
    // tile-resident kernel: 12 streams per workgroup; the tile must live for the whole launch (a stream makes per_block
    // calls) and the sampler must be the engine's own
    const bool resident = e->tile_resident && !neg_ids && per_block <= std::max<uint64_t>(1, e->cfg.refresh_interval);
    uint32_t launch_grid = grid, loss_slots = grid;
    if (resident)
    {
        launch_grid = (grid + (uint32_t)TILE_STREAMS - 1) / (uint32_t)TILE_STREAMS;
        loss_slots = launch_grid * (uint32_t)TILE_STREAMS;
    }
    if (e->tile_resident) std::snprintf(e->kname, sizeof(e->kname), "%s%s", e->kbase, resident ? "/tile-in-lds" : "");
    int rc = ensure_loss_part(e, loss_slots);
    if (rc) return rc;
    if (e->cfg.use_aggregator && !e->d_agg_state)
    {
        // per-stream aggregator state (call counter + pairs not yet applied), sized ONCE for the largest grid any range can
        // produce (geometry() never launches more than auto_streams workgroups): a later, larger window of the same epoch
        // must not find its predecessors' counters reallocated away
        const size_t cap = std::max<size_t>(e->auto_streams, 1);
        const size_t bytes = cap * agg_state_floats(e->lpr) * sizeof(float);
        HIP_TRY(hipMalloc(&e->d_agg_state, bytes));
        HIP_TRY(hipMemsetAsync(e->d_agg_state, 0, bytes, e->stream));
        e->agg_state_cap = cap;
    }
    if (e->cfg.use_aggregator && loss_slots > e->agg_state_cap) return fail(HEAT_CF_EINVAL, "internal: grid exceeds the aggregator state");
    TrainArgs a = make_args(e, begin, end);
    a.per_block = per_block;
    a.tile_streams = resident ? (uint32_t)TILE_STREAMS : 0u;
    if (neg_ids)
    {
        const size_t cnt = (size_t)n * e->cfg.num_negs;
        std::vector<uint32_t> tmp(cnt);
        for (size_t i = 0; i < cnt; ++i)
        {
            if (neg_ids[i] >= e->cfg.num_items) return fail(HEAT_CF_EINVAL, "neg_ids holds an id out of range");
            tmp[i] = (uint32_t)neg_ids[i];
        }
        if (cnt > e->ext_cap)
        {
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (e->d_ext_negs) HIP_TRY(hipFree(e->d_ext_negs));
            e->d_ext_negs = nullptr;
            HIP_TRY(hipMalloc(&e->d_ext_negs, cnt * sizeof(uint32_t)));
            e->ext_cap = cnt;
        }
        HIP_TRY(hipMemcpyAsync(e->d_ext_negs, tmp.data(), cnt * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream)); // tmp goes out of scope
        a.ext_negs = e->d_ext_negs;
        a.ext_base = begin;
    }
    EventPair ev;
    rc = get_events(e, &ev);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev.a, e->stream));
    a.loss_part = e->d_loss_part;   // (re)allocated above
    HIP_TRY(launch_train(a, e->lpr, e->ng, e->nw, launch_grid, e->aux, e->stream));
    HIP_TRY(hipEventRecord(ev.b, e->stream));
    e->ev_pending.push_back(ev);
    e->launches += 1;
    HIP_TRY(hipMemsetAsync(e->d_sums + 1, 0, sizeof(double), e->stream));
    HIP_TRY(launch_loss_reduce(e->d_loss_part, loss_slots, e->d_sums + 1, e->stream));
    HIP_TRY(launch_loss_reduce(e->d_loss_part, loss_slots, e->d_sums, e->stream));
    if (loss_sum)
    {
        HIP_TRY(hipMemcpyAsync(loss_sum, e->d_sums + 1, sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return HEAT_CF_OK;
}

int heat_cf_zero_grad(heat_cf_engine* e)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemsetAsync(e->d_user_g, 0, user_bytes(e), e->stream));
    HIP_TRY(hipMemsetAsync(e->d_item_g, 0, item_bytes(e), e->stream));
    return HEAT_CF_OK;
}

int heat_cf_end_epoch(heat_cf_engine* e)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    int rc = heat_cf_zero_grad(e); // train/engine.cpp:345-347
    if (rc) return rc;
    e->epoch += 1;                 // :378
    if (e->host_mode && !(e->cfg.flags & HEAT_CF_FLAG_LAZY_SYNC)) return heat_cf_sync_to_host(e);
    return HEAT_CF_OK;
}

int heat_cf_train_one_epoch(heat_cf_engine* e, float* mean_loss)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    int rc = heat_cf_begin_epoch(e);
    if (rc) return rc;
    rc = heat_cf_train_range(e, 0, e->data_rows, nullptr, nullptr);
    if (rc) return rc;
    double sum = 0.0;
    HIP_TRY(hipMemcpyAsync(&sum, e->d_sums, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    rc = heat_cf_end_epoch(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (mean_loss) *mean_loss = e->data_rows ? (float)(sum / (double)e->data_rows) : 0.0f; // :383-385
    return HEAT_CF_OK;
}

int heat_cf_sample_negatives(heat_cf_engine* e, uint64_t begin, uint64_t end, uint64_t* out)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    if (begin > end || end > e->data_rows || !out) return fail(HEAT_CF_EINVAL, "bad range / out");
    if (e->cfg.num_negs > 256) return fail(HEAT_CF_EUNSUP, "num_negs > 256");
    HIP_TRY(hipSetDevice(e->device));
    const uint64_t n = end - begin;
    if (n == 0) return HEAT_CF_OK;
    uint64_t per_block = 0;
    uint32_t grid = 0;
    geometry(e, n, &per_block, &grid);
    TrainArgs a = make_args(e, begin, end);
    a.per_block = per_block;
    uint64_t* d_out = nullptr;
    const size_t bytes = (size_t)n * e->cfg.num_negs * sizeof(uint64_t);
    HIP_TRY(hipMalloc(&d_out, bytes));
    hipError_t err = launch_sample_negs(a, grid, begin, d_out, e->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    (void)hipFree(d_out);
    HIP_TRY(err);
    return HEAT_CF_OK;
}

int heat_cf_evaluate0(heat_cf_engine* e, float* sim)
{
    if (!e || !sim) return fail(HEAT_CF_EINVAL, "engine / sim is NULL");
    HIP_TRY(hipSetDevice(e->device));
    const uint64_t U = e->cfg.num_users, I = e->cfg.num_items;
    // row panels keep the device scratch bounded (the full matrix is 19.3 GB at AmazonBooks shape, README.md:104)
    const uint64_t panel = std::max<uint64_t>(1, std::min<uint64_t>(U, (1ull << 30) / (I * sizeof(float)) + 1));
    float* d_panel = nullptr;
    HIP_TRY(hipMalloc(&d_panel, (size_t)panel * I * sizeof(float)));
    hipError_t err = hipSuccess;
    for (uint64_t u0 = 0; u0 < U && err == hipSuccess; u0 += panel)
    {
        const uint64_t rows = std::min(panel, U - u0);
        err = launch_sim_panel(e->d_user_w + u0 * e->cfg.emb_dim, e->d_item_w, d_panel, (uint32_t)rows, (uint32_t)I,
                               (uint32_t)e->cfg.emb_dim, e->stream);
        if (err == hipSuccess)
            err = hipMemcpyAsync(sim + u0 * I, d_panel, (size_t)rows * I * sizeof(float), hipMemcpyDeviceToHost, e->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    }
    (void)hipFree(d_panel);
    HIP_TRY(err);
    return HEAT_CF_OK;
}

// Materialised path: sim panel -> mask -> k rounds of arg-max.  Any k; also the cross-check for the fused path.
static int topk_panels(heat_cf_engine* e, uint64_t u_begin, uint64_t u_end, uint32_t k, const uint64_t* mask_indptr,
                       const uint32_t* mask_items, uint32_t* topk)
{
    HIP_TRY(hipSetDevice(e->device));
    const uint64_t nu = u_end - u_begin;
    if (nu == 0) return HEAT_CF_OK;
    const uint64_t I = e->cfg.num_items, d = e->cfg.emb_dim;
    const uint64_t panel = std::max<uint64_t>(1, std::min<uint64_t>(nu, (1ull << 30) / (I * sizeof(float)) + 1));
    float*    d_panel = nullptr;
    uint64_t* d_indptr = nullptr;
    uint32_t* d_items = nullptr;
    uint32_t* d_topk = nullptr;
    std::vector<uint64_t> rel;
    hipError_t err = hipMalloc(&d_panel, (size_t)panel * I * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&d_topk, (size_t)panel * k * sizeof(uint32_t));
    if (err == hipSuccess && mask_indptr) err = hipMalloc(&d_indptr, (panel + 1) * sizeof(uint64_t));
    int rc = HEAT_CF_OK;
    size_t items_cap = 0;
    for (uint64_t p0 = 0; p0 < nu && err == hipSuccess && rc == HEAT_CF_OK; p0 += panel)
    {
        const uint64_t rows = std::min(panel, nu - p0);
        const uint64_t ug = u_begin + p0;
        err = launch_sim_panel(e->d_user_w + ug * d, e->d_item_w, d_panel, (uint32_t)rows, (uint32_t)I, (uint32_t)d, e->stream);
        if (err == hipSuccess && mask_indptr)
        {
            const uint64_t lo = mask_indptr[ug], hi = mask_indptr[ug + rows];
            if (hi < lo) { rc = fail(HEAT_CF_EINVAL, "mask_indptr must be non-decreasing"); break; }
            for (uint64_t i = lo; i < hi; ++i)
                if (mask_items[i] >= I) { rc = fail(HEAT_CF_EINVAL, "mask_items holds an id out of range"); break; }
            if (rc) break;
            rel.resize(rows + 1);
            for (uint64_t u = 0; u <= rows; ++u)
            {
                if (mask_indptr[ug + u] < lo || mask_indptr[ug + u] > hi) { rc = fail(HEAT_CF_EINVAL, "mask_indptr must be non-decreasing"); break; }
                rel[u] = mask_indptr[ug + u] - lo;
            }
            if (rc) break;
            if ((hi - lo) > items_cap)
            {
                (void)hipStreamSynchronize(e->stream);
                (void)hipFree(d_items);
                d_items = nullptr;
                items_cap = (size_t)(hi - lo);
                err = hipMalloc(&d_items, items_cap * sizeof(uint32_t));
            }
            if (err == hipSuccess) err = hipMemcpyAsync(d_indptr, rel.data(), (rows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream);
            if (err == hipSuccess && hi > lo)
                err = hipMemcpyAsync(d_items, mask_items + lo, (hi - lo) * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream);
            if (err == hipSuccess && hi > lo) err = launch_mask_panel(d_panel, (uint32_t)rows, (uint32_t)I, d_indptr, d_items, e->stream);
        }
        if (err == hipSuccess) err = launch_topk_rows(d_panel, (uint32_t)rows, (uint32_t)I, k, d_topk, e->stream);
        if (err == hipSuccess)
            err = hipMemcpyAsync(topk + p0 * k, d_topk, (size_t)rows * k * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    }
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(d_panel);
    (void)hipFree(d_indptr);
    (void)hipFree(d_items);
    (void)hipFree(d_topk);
    if (rc) return rc;
    HIP_TRY(err);
    return HEAT_CF_OK;
}


// Fused path (topk_fused.hip): scores live in registers only.  Users go in panels of at most 2^20 so the partial lists
// stay small; mask rows are used as they are when sorted ascending and sorted into a private copy otherwise.
static int topk_fused(heat_cf_engine* e, uint64_t u_begin, uint64_t u_end, uint32_t k, const uint64_t* mask_indptr,
                      const uint32_t* mask_items, uint32_t* topk)
{
    const uint64_t nu = u_end - u_begin, I = e->cfg.num_items, d = e->cfg.emb_dim;
    const bool trace = getenv("HEAT_CF_TRACE") != nullptr; // stderr: where the host time of this call goes
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_start = now();
    uint64_t panel_cap = 1ull << 20;
    if (const char* pc = getenv("HEAT_CF_TOPK_PANEL")) // tests: walk several user panels with a small table
        panel_cap = std::max<uint64_t>(1, strtoull(pc, nullptr, 10));
    const uint64_t panel = std::min<uint64_t>(nu, panel_cap);
    const uint32_t slots = e->cu_count;     // topk_fused_splits knows how many workgroups of its kernel a compute unit holds
    const uint64_t last = nu % panel ? nu % panel : panel;
    const size_t part_elems = (size_t)k * std::max<uint64_t>(panel * topk_fused_splits((uint32_t)panel, (uint32_t)I, slots, (uint32_t)d, k),
                                                             last * topk_fused_splits((uint32_t)last, (uint32_t)I, slots, (uint32_t)d, k));
    float*    d_pv = nullptr;
    float*    d_thr = nullptr;
    uint32_t* d_pi = nullptr;
    uint32_t* d_topk = nullptr;
    uint64_t* d_indptr = nullptr;
    uint32_t* d_items = nullptr;
    uint32_t* d_raw = nullptr;
    void*     d_temp = nullptr;
    size_t items_cap = 0, temp_cap = 0;
    std::vector<uint64_t> rel;
    hipError_t err = hipMalloc(&d_pv, part_elems * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&d_thr, (size_t)panel * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&d_pi, part_elems * sizeof(uint32_t));
    if (err == hipSuccess) err = hipMalloc(&d_topk, (size_t)panel * k * sizeof(uint32_t));
    if (err == hipSuccess && mask_indptr) err = hipMalloc(&d_indptr, (panel + 1) * sizeof(uint64_t));
    int rc = HEAT_CF_OK;
    const auto t_alloc = now();
    double ms_check = 0, ms_kernel = 0;
    for (uint64_t p0 = 0; p0 < nu && err == hipSuccess && rc == HEAT_CF_OK; p0 += panel)
    {
        const uint64_t rows = std::min(panel, nu - p0);
        const uint64_t ug = u_begin + p0;
        uint64_t n_items = 0;
        const auto t0 = now();
        if (mask_indptr)
        {
            const uint64_t lo = mask_indptr[ug], hi = mask_indptr[ug + rows];
            if (hi < lo) { rc = fail(HEAT_CF_EINVAL, "mask_indptr must be non-decreasing"); break; }
            rel.resize(rows + 1);
            bool ascending = true, in_range = true, monotone = true;
            for (uint64_t u = 0; u < rows; ++u)
            {
                const uint64_t a = mask_indptr[ug + u], b = mask_indptr[ug + u + 1];
                rel[u] = a - lo;
                if (a < lo || b < a || b > hi) { monotone = false; break; }
                uint32_t prev = 0;
                for (const uint32_t *q = mask_items + a, *qe = mask_items + b; q < qe; ++q)
                {
                    const uint32_t v = *q;
                    in_range &= v < I;
                    ascending &= v >= prev;
                    prev = v;
                }
            }
            rel[rows] = hi - lo;
            if (!monotone) { rc = fail(HEAT_CF_EINVAL, "mask_indptr must be non-decreasing"); break; }
            if (!in_range) { rc = fail(HEAT_CF_EINVAL, "mask_items holds an id out of range"); break; }
            n_items = hi - lo;
            if (n_items > 0xFFFFFFFFull) { rc = fail(HEAT_CF_EUNSUP, "more than 2^32 mask items in one user panel"); break; }
            if (n_items > items_cap)
            {
                (void)hipStreamSynchronize(e->stream);
                (void)hipFree(d_items);
                (void)hipFree(d_raw);
                d_items = d_raw = nullptr;
                items_cap = (size_t)n_items;
                err = hipMalloc(&d_items, items_cap * sizeof(uint32_t));
                if (err == hipSuccess) err = hipMalloc(&d_raw, items_cap * sizeof(uint32_t));
            }
            if (err == hipSuccess) err = hipMemcpyAsync(d_indptr, rel.data(), (rows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream);
            if (err == hipSuccess && n_items)
                err = hipMemcpyAsync(ascending ? d_items : d_raw, mask_items + lo, n_items * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream);
            if (err == hipSuccess && n_items && !ascending)
            {
                // rows in file order (cf/datasets.py:31-79): order them on the device, one segmented sort
                uint32_t id_bits = 1;
                while (id_bits < 32 && (1ull << id_bits) < I) ++id_bits;
                size_t need = 0;
                err = sort_mask_rows(d_raw, d_items, (uint32_t)n_items, (uint32_t)rows, d_indptr, id_bits, nullptr, &need, e->stream);
                if (err == hipSuccess && need > temp_cap)
                {
                    (void)hipStreamSynchronize(e->stream);
                    (void)hipFree(d_temp);
                    d_temp = nullptr;
                    temp_cap = need;
                    err = hipMalloc(&d_temp, temp_cap);
                }
                if (err == hipSuccess)
                    err = sort_mask_rows(d_raw, d_items, (uint32_t)n_items, (uint32_t)rows, d_indptr, id_bits, d_temp, &need, e->stream);
            }
        }
        const uint32_t splits = topk_fused_splits((uint32_t)rows, (uint32_t)I, slots, (uint32_t)d, k);
        const auto t1 = now();
        ms_check += ms(t0, t1);
        if (err == hipSuccess)
            err = launch_topk_fused(e->d_user_w + ug * d, e->d_item_w, (uint32_t)rows, (uint32_t)I, (uint32_t)d, k,
                                    mask_indptr ? d_indptr : nullptr, d_items, splits, d_pv, d_pi, d_topk, d_thr, e->stream);
        if (err == hipSuccess)
            err = hipMemcpyAsync(topk + p0 * k, d_topk, (size_t)rows * k * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(e->stream); // rel is reused by the next panel
        ms_kernel += ms(t1, now());
    }
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(d_pv);
    (void)hipFree(d_thr);
    (void)hipFree(d_pi);
    (void)hipFree(d_topk);
    (void)hipFree(d_indptr);
    (void)hipFree(d_items);
    (void)hipFree(d_raw);
    (void)hipFree(d_temp);
    if (trace)
        fprintf(stderr, "[heat_cf] topk fused: alloc %.2f ms, mask check+upload %.2f ms, kernels+download %.2f ms, total %.2f ms\n",
                ms(t_start, t_alloc), ms_check, ms_kernel, ms(t_start, now()));
    if (rc) return rc;
    HIP_TRY(err);
    return HEAT_CF_OK;
}

int heat_cf_topk(heat_cf_engine* e, uint64_t u_begin, uint64_t u_end, uint32_t k, const uint64_t* mask_indptr,
                 const uint32_t* mask_items, uint32_t* topk)
{
    if (!e || !topk) return fail(HEAT_CF_EINVAL, "engine / topk is NULL");
    if (u_begin > u_end || u_end > e->cfg.num_users) return fail(HEAT_CF_EINVAL, "user range out of bounds");
    if (k == 0 || k > e->cfg.num_items) return fail(HEAT_CF_EINVAL, "k must be in [1, num_items]");
    if (mask_indptr && !mask_items) return fail(HEAT_CF_EINVAL, "mask_items is NULL");
    HIP_TRY(hipSetDevice(e->device));
    if (u_end == u_begin) return HEAT_CF_OK;
    // HEAT_CF_TOPK_PATH=panel forces the materialised path (tests compare the two)
    const char* path = getenv("HEAT_CF_TOPK_PATH");
    const bool want_panel = path && strcmp(path, "panel") == 0;
    const bool can_fuse = k <= TOPK_FUSED_MAX_K && (e->cfg.emb_dim % 4) == 0 && e->cfg.num_items < 0xFFFFFF00ull;
    if (can_fuse && !want_panel) return topk_fused(e, u_begin, u_end, k, mask_indptr, mask_items, topk);
    return topk_panels(e, u_begin, u_end, k, mask_indptr, mask_items, topk);
}

int heat_cf_sync_to_host(heat_cf_engine* e)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    if (!e->host_mode) return HEAT_CF_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(e->h_user_w, e->d_user_w, user_bytes(e), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_item_w, e->d_item_w, item_bytes(e), hipMemcpyDeviceToHost, e->stream));
    if (e->cfg.use_aggregator && e->h_w0 && e->d_w0)
        HIP_TRY(hipMemcpyAsync(e->h_w0, e->d_w0, (size_t)e->cfg.emb_dim * e->cfg.emb_dim * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return HEAT_CF_OK;
}

int heat_cf_sync_from_host(heat_cf_engine* e)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    if (!e->host_mode) return HEAT_CF_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(e->d_user_w, e->h_user_w, user_bytes(e), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_item_w, e->h_item_w, item_bytes(e), hipMemcpyHostToDevice, e->stream));
    if (e->cfg.use_aggregator && e->h_w0 && e->d_w0)
        HIP_TRY(hipMemcpyAsync(e->d_w0, e->h_w0, (size_t)e->cfg.emb_dim * e->cfg.emb_dim * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return HEAT_CF_OK;
}

int heat_cf_sync_delta(heat_cf_engine* e, const void* d_ref, void* d_mine, void* d_sum)
{
    if (!e || !d_ref || !d_sum) return fail(HEAT_CF_EINVAL, "engine / buffer is NULL");
    if (((uintptr_t)d_ref | (uintptr_t)d_mine | (uintptr_t)d_sum) & 15u) return fail(HEAT_CF_EINVAL, "buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(launch_item_delta(e->d_item_w, (const float*)d_ref, (float*)d_mine, (float*)d_sum,
                              (size_t)e->cfg.num_items * e->cfg.emb_dim, e->stream));
    return HEAT_CF_OK;
}

int heat_cf_sync_apply(heat_cf_engine* e, void* d_ref, const void* d_sum, const void* d_mine, float scale)
{
    if (!e || !d_ref || !d_sum) return fail(HEAT_CF_EINVAL, "engine / buffer is NULL");
    if (((uintptr_t)d_ref | (uintptr_t)d_mine | (uintptr_t)d_sum) & 15u) return fail(HEAT_CF_EINVAL, "buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(launch_item_apply(e->d_item_w, (float*)d_ref, (const float*)d_sum, (const float*)d_mine, scale,
                              (size_t)e->cfg.num_items * e->cfg.emb_dim, e->stream));
    return HEAT_CF_OK;
}

int heat_cf_sync_apply_delta(heat_cf_engine* e, void* d_ref, void* d_sum, void* d_mine, float scale)
{
    if (!e || !d_ref || !d_sum || !d_mine) return fail(HEAT_CF_EINVAL, "engine / buffer is NULL");
    if (((uintptr_t)d_ref | (uintptr_t)d_mine | (uintptr_t)d_sum) & 15u) return fail(HEAT_CF_EINVAL, "buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(launch_item_apply_delta(e->d_item_w, (float*)d_ref, (float*)d_sum, (float*)d_mine, scale,
                                    (size_t)e->cfg.num_items * e->cfg.emb_dim, e->stream));
    return HEAT_CF_OK;
}

int heat_cf_sync_apply_snap(heat_cf_engine* e, const void* d_x, void* d_snap)
{
    if (!e || !d_snap) return fail(HEAT_CF_EINVAL, "engine / buffer is NULL");
    if (((uintptr_t)d_x | (uintptr_t)d_snap) & 15u) return fail(HEAT_CF_EINVAL, "buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(launch_item_apply_snap(e->d_item_w, (const float*)d_x, (float*)d_snap, (size_t)e->cfg.num_items * e->cfg.emb_dim, e->stream));
    return HEAT_CF_OK;
}

int heat_cf_sync_delta_from(heat_cf_engine* e, const void* d_snap, const void* d_ref, void* d_mine, void* d_sum, void* stream)
{
    if (!e || !d_snap || !d_ref || !d_sum) return fail(HEAT_CF_EINVAL, "engine / buffer is NULL");
    if (((uintptr_t)d_snap | (uintptr_t)d_ref | (uintptr_t)d_mine | (uintptr_t)d_sum) & 15u) return fail(HEAT_CF_EINVAL, "buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(launch_item_delta((const float*)d_snap, (const float*)d_ref, (float*)d_mine, (float*)d_sum,
                              (size_t)e->cfg.num_items * e->cfg.emb_dim, (hipStream_t)stream));
    return HEAT_CF_OK;
}

int heat_cf_sync_finish(heat_cf_engine* e, void* d_ref, const void* d_sum, void* d_mine_x, float scale, void* stream)
{
    if (!e || !d_ref || !d_sum || !d_mine_x) return fail(HEAT_CF_EINVAL, "engine / buffer is NULL");
    if (((uintptr_t)d_ref | (uintptr_t)d_mine_x | (uintptr_t)d_sum) & 15u) return fail(HEAT_CF_EINVAL, "buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(launch_item_finish((float*)d_ref, (const float*)d_sum, (float*)d_mine_x, scale,
                               (size_t)e->cfg.num_items * e->cfg.emb_dim, (hipStream_t)stream));
    return HEAT_CF_OK;
}

int heat_cf_synchronize(heat_cf_engine* e)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return HEAT_CF_OK;
}

int heat_cf_get_device_view(heat_cf_engine* e, heat_cf_device_view* view)
{
    if (!e || !view) return fail(HEAT_CF_EINVAL, "engine / view is NULL");
    view->user_w = e->d_user_w;
    view->item_w = e->d_item_w;
    view->user_g = e->d_user_g;
    view->item_g = e->d_item_g;
    view->w0 = e->d_w0;
    view->clicks = e->d_clicks;
    view->data_rows = e->data_rows;
    view->stream = (void*)e->stream;
    return HEAT_CF_OK;
}

int heat_cf_copy_to_host(heat_cf_engine* e, const void* device_ptr, void* host_ptr, uint64_t bytes)
{
    if (!e || !device_ptr || !host_ptr) return fail(HEAT_CF_EINVAL, "engine / pointer is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(host_ptr, device_ptr, (size_t)bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return HEAT_CF_OK;
}

uint64_t heat_cf_epoch(const heat_cf_engine* e) { return e ? e->epoch : 0; }
float    heat_cf_learning_rate(const heat_cf_engine* e) { return e ? e->lr : 0.f; }

int heat_cf_set_learning_rate(heat_cf_engine* e, float l_r)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    e->lr = l_r;
    return HEAT_CF_OK;
}

int heat_cf_set_epoch(heat_cf_engine* e, uint64_t epoch)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    e->epoch = epoch;
    return HEAT_CF_OK;
}

int heat_cf_kernel_time(heat_cf_engine* e, double* total_ms, uint64_t* launches, int reset)
{
    if (!e) return fail(HEAT_CF_EINVAL, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (auto& q : e->ev_pending)
    {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, q.a, q.b));
        e->kernel_ms += ms;
        e->ev_free.push_back(q);
    }
    e->ev_pending.clear();
    if (total_ms) *total_ms = e->kernel_ms;
    if (launches) *launches = e->launches;
    if (reset)
    {
        e->kernel_ms = 0.0;
        e->launches = 0;
    }
    return HEAT_CF_OK;
}

const char* heat_cf_kernel_name(const heat_cf_engine* e) { return e ? e->kname : ""; }

} // extern "C"
