/* heat_cf.h — C ABI of the MI355X-native SimpleX/CCL collaborative-filtering training engine.
 *
 * This is the drop-in boundary for ONE hot path of visuOwO/HEAT: the per-interaction fused
 * forward+backward+clipped-SGD step and its epoch loop.  Every entry point cites the reference
 * interface it replaces (paths relative to /root/reference/cf_cpu/src).  The reference's own
 * boundary is the pybind11 module `cf_c` (pybind/init_modules.cpp:13-152); heat_amd/csrc/cf_c_module.cpp
 * re-creates that module on top of this ABI, and INTEGRATION.md shows the binding a HEAT maintainer
 * would add.
 *
 * Conventions
 *   - plain C types only; all functions return 0 on success, a negative HEAT_CF_E* code on error;
 *     heat_cf_last_error() returns a thread-local human-readable message.
 *   - ids are uint64 (`idx_t`, splatt/base.h:49-50), values are fp32 (`val_t`, CMakeLists.txt:11).
 *   - "host mode": the caller's numpy-style host buffers are BORROWED and trained in place, exactly like
 *     the reference (init_modules.cpp:45-56,74-84; memory/array.hpp:30-34): the engine uploads them once,
 *     keeps tables resident in HBM, and writes weights back into the same buffers at the end of
 *     heat_cf_train_one_epoch() (or on heat_cf_sync_to_host()).
 *   - "device mode": the caller owns device buffers (e.g. torch tensors) and passes raw device pointers;
 *     nothing is copied; used by bench.py and the multi-GPU driver (RCCL all-reduce on the caller's tensors).
 *   - The HIP device code is gfx950-only.  There is NO CPU fallback: without a usable GPU every compute
 *     entry point fails with HEAT_CF_EHIP.
 */
#ifndef HEAT_CF_H
#define HEAT_CF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEAT_CF_ABI_VERSION 1

/* error codes */
#define HEAT_CF_OK        0
#define HEAT_CF_EINVAL   -1 /* bad argument / shape / dtype-equivalent (-> Python ValueError)   */
#define HEAT_CF_EHIP     -2 /* HIP runtime error, no device, kernel fault (-> RuntimeError)     */
#define HEAT_CF_ENOMEM   -3 /* host or device allocation failed                                 */
#define HEAT_CF_EUNSUP   -4 /* configuration outside the compiled kernel family                 */

/* flags */
#define HEAT_CF_FLAG_SERIAL        0x1u /* one wave walks all interactions in stored order (parity tests)      */
#define HEAT_CF_FLAG_LAZY_SYNC     0x2u /* host mode: do not write weights back after every epoch              */
#define HEAT_CF_FLAG_NULL_STREAM   0x8u /* device mode with stream == NULL: launch on the legacy default stream instead of
                                           creating a private non-blocking one (callers whose other work is there)   */
#define HEAT_CF_FLAG_TILE_GLOBAL   0x10u /* random-tile sampler: never hold the tile in LDS (the default since round 3; the flag
                                           still wins over HEAT_CF_FLAG_TILE_LDS)                                        */
#define HEAT_CF_FLAG_TILE_LDS      0x20u /* random-tile sampler (with HEAT_CF_FLAG_SAMPLING_CALL): hold the tile in LDS when it
                                           fits — 12 streams per workgroup share tile_size x emb_dim fp32 of accumulated
                                           weight deltas, flushed to the table by float atomics at the end of the launch.
                                           Opt-in: the tile is then shared by 12 streams and its negative-row updates are
                                           private to the workgroup for the launch, unlike the reference's one tile per
                                           worker written through to the table                                          */
#define HEAT_CF_FLAG_SAMPLING_CALL 0x4u /* use sampler.sampling() (engine.cpp:333) instead of the live
                                           ignore_pos_sampling() (engine.cpp:332)                              */

/* cache policy of table-row traffic (see DESIGN.md "Hogwild across XCDs") */
#define HEAT_CF_COHERENCE_DEFAULT 0 /* engine picks (device-coherent)                                           */
#define HEAT_CF_COHERENCE_PLAIN   1 /* plain loads/stores: per-XCD L2 copies may diverge inside one launch      */
#define HEAT_CF_COHERENCE_DEVICE  2 /* sc1 loads + sc1 write-through stores: coherent across the 8 XCDs         */

/* How an item row's (W, G) update is written back (DESIGN.md "Hogwild at GPU concurrency").  The reference writes
 * rows back with memcpy (memory/array.hpp:52-55); with 8 CPU threads two workers almost never hold the same row, with
 * thousands of concurrent GPU streams they constantly do, and a literal overwrite then drops most updates of popular
 * rows.  Float atomic adds (W += -lr*G, G += G_new - G_read) lose nothing; they are bound by the memory-side atomic
 * rate (~20 G 64-byte requests/s), so they are used where the measured conflict rate makes them necessary. */
#define HEAT_CF_UPDATE_DEFAULT    0 /* = AUTO                                                                        */
#define HEAT_CF_UPDATE_OVERWRITE  1 /* literal overwrite of W and G for every row (one stream per user run)          */
#define HEAT_CF_UPDATE_ATOMIC_W   2 /* W by atomic add for every item row, G overwritten                             */
#define HEAT_CF_UPDATE_ATOMIC_WG  3 /* W and G by atomic add for every item row: no update is ever lost              */
#define HEAT_CF_UPDATE_ATOMIC_POS 4 /* positive item row: W and G by atomic add; negative rows: overwrite            */
#define HEAT_CF_UPDATE_AUTO       5 /* up to 17 rows per interaction: ATOMIC_POS (validated up to 0.56 in-flight touches per
                                       item row, streams * (num_negs + 1) / num_items); more rows per interaction:
                                       ATOMIC_POS up to 0.15 in-flight touches, REREAD_POS above                      */
#define HEAT_CF_UPDATE_REREAD_POS 6 /* positive row: W and G by atomic add; negative rows: W re-read together with G two
                                       row groups ahead of the update, plain stores ("late re-read": the
                                       read-modify-write window of a negative row is one memory round trip)           */
/* values 16..47: raw policy bits (16 + bit0 neg W atomic + bit1 neg G atomic + bit2 pos W atomic + bit3 pos G atomic
 *                + bit4 "late re-read": a negative row's W is read again next to its G row just before its update and the
 *                update is applied to that fresh value — the read-modify-write window of a negative row shrinks from
 *                the whole interaction to one memory round trip; needs HEAT_CF_COHERENCE_DEVICE and plain negative-row
 *                stores, i.e. bits 0-1 clear) */

/* Replaces cf::modules::CFConfig (modules/cf_config.hpp:12-35; bound at pybind/init_modules.cpp:13-33).
 * The first 13 fields are the reference's, in its constructor order.  The rest are extensions the
 * reference does not have (it ignores yaml `seed`, main.py:19-124). */
typedef struct heat_cf_config
{
    uint64_t emb_dim;
    uint64_t num_negs;
    uint64_t num_users;
    uint64_t num_items;
    uint64_t train_size;
    uint64_t neg_sampler;       /* 0: uniform random, 1: random tiling (cf_config.hpp:27) */
    uint64_t tile_size;
    uint64_t refresh_interval;
    uint64_t num_subepochs;     /* kwarg `num_subepoches`; only the fork's MPI scaffold reads it */
    float    l2;                /* plumbed, never applied (matrix_factorization.cpp:126,146,165,168) */
    float    clip_val;
    const uint64_t* milestones;
    uint64_t n_milestones;
    float    l_r;
    /* ---- extensions ---- */
    uint64_t seed;              /* negative-sampler seed (yaml `seed`, e.g. 2022) */
    uint64_t sample_index_base; /* added to the local interaction index in the sampler counter (multi-GPU shards) */
    uint32_t use_aggregator;    /* 0: MF-CCL (north-star scope); 1: behaviour aggregation (behavior_aggregators.cpp) */
    uint32_t flags;             /* HEAT_CF_FLAG_* */
    uint32_t coherence;         /* HEAT_CF_COHERENCE_* */
    int32_t  device;            /* HIP device ordinal; -1 = current device */
    uint32_t num_streams;       /* concurrent sequential interaction streams (workgroups); 0 = auto: what fills the GPU,
                                   capped at 0.56 * num_items / (num_negs + 1) (0.45 above 17 rows per interaction) and
                                   at data_rows / 256 (a stream walks at least 256 interactions; 5600 above 17 rows per
                                   interaction) — the asynchrony
                                   validated against the oracle's Recall/NDCG at AmazonBooks and Yelp18 shape, DESIGN.md
                                   section 3 — rounded down to whole workgroups per compute unit */
    uint32_t update_mode;       /* HEAT_CF_UPDATE_* */
} heat_cf_config;

typedef struct heat_cf_engine heat_cf_engine;

/* addresses of the device-resident state (for collectives / zero-copy wrapping by the caller) */
typedef struct heat_cf_device_view
{
    void*    user_w;  /* [num_users, emb_dim] fp32 */
    void*    item_w;  /* [num_items, emb_dim] fp32 */
    void*    user_g;  /* persistent clipped-gradient rows (embeddings/embedding.cpp:12-13) */
    void*    item_g;
    void*    w0;      /* [emb_dim, emb_dim] fp32 or NULL */
    void*    clicks;  /* [data_rows] packed {u32 user, u32 item} */
    uint64_t data_rows;
    void*    stream;  /* hipStream_t the engine launches on */
} heat_cf_device_view;

int         heat_cf_abi_version(void);
const char* heat_cf_last_error(void);
/* Launch plan the engine would use for `cfg` — kernel variant, coherence, number of streams (asynchrony cap) and update
 * policy — as a small JSON object written to `out`.  Pure host logic, no GPU needed; `resident_workgroups` is what the
 * chip keeps resident for the variant (0 = unknown: only the caps apply).  "binding" names the bound that set the stream
 * count; "regime" says how l_r relates to the step size those bounds were measured at ("measured ...", "extrapolated ...",
 * "outside the measured range ..."; an engine created outside the measured range says so once on stderr).  Give `out` at
 * least 1024 bytes.  With use_aggregator != 0 an engine also
 * counts max_his history rows per interaction in the stream bound, which this call cannot know: read the engine's own
 * choice from heat_cf_kernel_name (".../streams=N"). */
int         heat_cf_plan(const heat_cf_config* cfg, uint64_t data_rows, uint64_t resident_workgroups, char* out,
                         uint64_t out_bytes);
/* number of visible HIP devices, or a negative error code */
int         heat_cf_device_count(void);

/* Replaces the constructors bound at pybind/init_modules.cpp:45-59 (ClickDataset), :74-84
 * (MatrixFactorization), :95-99 (AggregatorWeights) and :109-116 (Engine) — one call builds the whole
 * training state.  HOST pointers, borrowed:
 *   clicks [data_rows,2] u64 (user,item) in LightGCN order (datasets.py:63-67)
 *   his    [num_users,max_his] u64, masks [num_users,1] u64   (may be NULL when use_aggregator == 0)
 *   user_w [num_users,emb_dim], item_w [num_items,emb_dim] fp32, trained IN PLACE
 *   w0     [emb_dim,emb_dim] fp32 (may be NULL when use_aggregator == 0)
 * Ids are range-checked here (the reference does not check; an out-of-range id on a GPU is a fault). */
int heat_cf_engine_create(const heat_cf_config* cfg, const uint64_t* clicks, uint64_t data_rows,
                          const uint64_t* his, uint64_t max_his, const uint64_t* masks, float* user_w,
                          float* item_w, float* w0, heat_cf_engine** out);

/* Device-mode twin: every pointer is a DEVICE pointer owned by the caller (clicks still u64 pairs, they are
 * packed to u32 pairs into engine-owned memory and range-checked on the GPU; with use_aggregator != 0 the same
 * holds for d_his [num_users,max_his] u64 and d_masks [num_users] u64, and d_w0 [emb_dim,emb_dim] fp32 is trained
 * in place — the form a multi-GPU caller uses to all-reduce W0, train/engine.cpp:355-359).  `stream` is a
 * hipStream_t (NULL = the engine creates its own non-blocking stream). */
int heat_cf_engine_create_device(const heat_cf_config* cfg, const void* d_clicks, uint64_t data_rows,
                                 const void* d_his, uint64_t max_his, const void* d_masks, void* d_user_w,
                                 void* d_item_w, void* d_w0, void* stream, heat_cf_engine** out);

void heat_cf_engine_destroy(heat_cf_engine* e);

/* Replaces Engine::train_one_epoch() (train/engine.hpp:29; semantics = upstream OpenMP body
 * train/engine.cpp:294-342 + LR schedule :156-160 + zero_grad :345-347 + mean loss :380-385):
 * LR step for the current epoch, one pass over all interactions, zero both G tables, epoch += 1,
 * host mode: weights written back into the borrowed buffers.  *mean_loss = sum(loss)/data_rows. */
int heat_cf_train_one_epoch(heat_cf_engine* e, float* mean_loss);

/* The same epoch cut into pieces (multi-GPU windows, parity tests):
 *   begin_epoch : LR schedule (engine.cpp:156-160)
 *   train_range : interactions [begin,end) in stored order.  neg_ids (HOST, [end-begin, num_negs] u64, may be
 *                 NULL) overrides the on-GPU sampler with caller-fed negatives.  Asynchronous on the engine's
 *                 stream unless loss_sum != NULL (then it synchronises and returns the fp64 loss sum).
 *   end_epoch   : zero_grad (engine.cpp:345-347), epoch += 1 (:378), host-mode write-back.  */
int heat_cf_begin_epoch(heat_cf_engine* e);
int heat_cf_train_range(heat_cf_engine* e, uint64_t begin, uint64_t end, const uint64_t* neg_ids, double* loss_sum);
int heat_cf_end_epoch(heat_cf_engine* e);

/* Negatives the on-GPU sampler (Philox4x32-10, counter = interaction index) yields for interactions
 * [begin,end) of the CURRENT epoch, written to HOST out[(end-begin), num_negs] u64.  Replaces nothing in the
 * reference API; it exposes negative_samplers/uniform_random_negative_sampler.cpp:26-36 semantics for tests. */
int heat_cf_sample_negatives(heat_cf_engine* e, uint64_t begin, uint64_t end, uint64_t* out);

/* Replaces Engine::evaluate0() + the PyMatrix copy (train/engine.cpp:388-400, pybind/init_modules.cpp:122-129):
 * sim[num_users,num_items] = U * V^T, fp32, written to HOST memory. */
int heat_cf_evaluate0(heat_cf_engine* e, float* sim);

/* Fused evaluate0 + train-mask + top-k (train/engine.cpp:388-400 + metrics.py:21-29): for users
 * [u_begin,u_end) writes the k best item ids (descending score, ties by ascending id, masked items last) to HOST
 * topk[(u_end-u_begin), k] u32.  mask_indptr [num_users+1] / mask_items is the CSR of train items per user (HOST, may
 * be NULL = no masking; rows need not be sorted).  Scores are the fp32 dots evaluate0 returns, bit for bit; for
 * k <= 64 they are never stored (no [users, items] matrix), larger k goes through score panels. */
int heat_cf_topk(heat_cf_engine* e, uint64_t u_begin, uint64_t u_end, uint32_t k, const uint64_t* mask_indptr,
                 const uint32_t* mask_items, uint32_t* topk);

/* host <-> device weight copies (host mode only; no-ops in device mode) */
int heat_cf_sync_to_host(heat_cf_engine* e);
int heat_cf_sync_from_host(heat_cf_engine* e);
/* Multi-GPU item-table exchange, the two element-wise passes around the caller's collective (replaces the
 * per-row MPI_Allreduce + "/ world_size" loop of train/engine.cpp:366-375).  All pointers are DEVICE buffers of
 * num_items * emb_dim fp32 owned by the caller, 16-byte aligned; both calls are asynchronous on the engine's stream.
 *   delta : mine = sum = W_item - ref                           (this rank's change since the common reference)
 *   [caller all-reduces `sum` over the ranks — RCCL over xGMI — and may train the next window meanwhile]
 *   apply : W_item += scale * sum - mine ;  ref += scale * sum   (scale 1: every rank's updates applied;
 *                                                                  scale 1/world_size: replicas averaged, the fork's intent)
 *           d_mine == NULL (nothing trained since `delta`): W_item = ref = ref + scale * sum, bit-identical on every rank */
int heat_cf_sync_delta(heat_cf_engine* e, const void* d_ref, void* d_mine, void* d_sum);
int heat_cf_sync_apply(heat_cf_engine* e, void* d_ref, const void* d_sum, const void* d_mine, float scale);
/* apply of one exchange and delta of the next in ONE pass (what the overlapped schedule runs at every window boundary):
 *   W_item += scale * sum - mine ; ref += scale * sum ; mine = sum = W_item - ref      (same bits as apply, then delta) */
int heat_cf_sync_apply_delta(heat_cf_engine* e, void* d_ref, void* d_sum, void* d_mine, float scale);
/* The same exchange with ONE pass on the engine's stream (pipelined form; tables bit-identical to delta / apply with d_mine):
 *   apply_snap (engine's stream)  : W_item += x (d_x == NULL: nothing to add) ; snap = W_item
 *   delta_from (caller's stream)  : mine = sum = snap - ref
 *   [caller all-reduces `sum` on that stream while the engine trains its next window]
 *   finish     (caller's stream)  : s = scale * sum ; x = s - mine, written over d_mine_x ; ref += s
 * `stream` is a hipStream_t of the caller (the exchange stream); the caller orders it after apply_snap and orders the next
 * apply_snap after finish (events).  None of the three touches the persistent gradient rows. */
int heat_cf_sync_apply_snap(heat_cf_engine* e, const void* d_x, void* d_snap);
int heat_cf_sync_delta_from(heat_cf_engine* e, const void* d_snap, const void* d_ref, void* d_mine, void* d_sum, void* stream);
int heat_cf_sync_finish(heat_cf_engine* e, void* d_ref, const void* d_sum, void* d_mine_x, float scale, void* stream);
/* blocks until everything queued on the engine's stream has finished */
int heat_cf_synchronize(heat_cf_engine* e);

int heat_cf_get_device_view(heat_cf_engine* e, heat_cf_device_view* view);
/* copies `bytes` from device memory (e.g. a table of the device view) to HOST memory on the engine's stream and waits */
int heat_cf_copy_to_host(heat_cf_engine* e, const void* device_ptr, void* host_ptr, uint64_t bytes);

/* scalar state */
uint64_t heat_cf_epoch(const heat_cf_engine* e);
float    heat_cf_learning_rate(const heat_cf_engine* e);
int      heat_cf_set_learning_rate(heat_cf_engine* e, float l_r);
int      heat_cf_set_epoch(heat_cf_engine* e, uint64_t epoch);
/* zero both persistent gradient tables (embeddings/embedding.cpp:41-45) */
int      heat_cf_zero_grad(heat_cf_engine* e);

/* LightGCN text ingest: parses "user item item ...\n" lines — the per-line Python loop of
 * cf/datasets.py:31-79 — in one pass over the mmap'ed file.  Output (malloc'ed, release with heat_cf_free_lightgcn):
 * clicks [n_interactions,2] u64 in FILE ORDER (datasets.py:74-78), line_user [num_lines] (user id of each line),
 * line_start [num_lines+1] (first interaction of each line).  Host-only. */
typedef struct heat_cf_lightgcn
{
    uint64_t  num_lines;
    uint64_t  n_interactions;
    uint64_t  max_user_id;
    uint64_t  max_item_id;
    uint64_t* clicks;
    uint64_t* line_user;
    uint64_t* line_start;
} heat_cf_lightgcn;
int  heat_cf_parse_lightgcn(const char* path, char separator, heat_cf_lightgcn* out);
void heat_cf_free_lightgcn(heat_cf_lightgcn* g);

/* Timing of the training kernel(s), measured with HIP events on the engine's stream:
 * accumulated kernel milliseconds and launch count since the last reset. */
int heat_cf_kernel_time(heat_cf_engine* e, double* total_ms, uint64_t* launches, int reset);
/* name of the kernel variant the engine dispatches for its configuration (for profiles/) */
const char* heat_cf_kernel_name(const heat_cf_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* HEAT_CF_H */
