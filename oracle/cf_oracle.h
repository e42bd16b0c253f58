/* TEST INFRASTRUCTURE — CPU oracle, NOT product code.
 *
 * Plain-C restatement of the reference's (visuOwO/HEAT) per-interaction SimpleX/CCL
 * SGD hot path, written from the reference's source text; every function cites the
 * reference file:line it follows (paths relative to /root/reference/cf_cpu/src).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (heat_amd/) never links, imports or falls back to it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - RNG / samplers / SGD / clip / LR schedule / row read-write / zero_grad: PINNED
 *     bit-exactly against the reference's own Eigen-free translation units compiled
 *     here (oracle/_ref, tests/golden/ref_kats.json).
 *   - forward_backward, behaviour aggregator, epoch loop: the reference needs Eigen 3.4
 *     (absent, no network) and has no tests or golden vectors => "parity unpinned" by
 *     reference fixtures; cross-checked against a float64 analytic model instead.
 */
#ifndef CF_ORACLE_H
#define CF_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- random/uniform.hpp:16-30 : std::mt19937_64 + std::uniform_int_distribution(0,max) ---- */
typedef struct
{
    uint64_t mt[312];
    int      idx;
} orc_mt64;

void     orc_mt64_seed(orc_mt64* g, uint64_t seed);
uint64_t orc_mt64_next(orc_mt64* g);

typedef struct
{
    orc_mt64 rng;
    uint64_t max_idx;
} orc_uniform;

void     orc_uniform_init(orc_uniform* u, uint64_t max_idx, uint64_t seed);
uint64_t orc_uniform_read(orc_uniform* u);

/* ---- cf_config.hpp:12-35 ---- */
typedef struct
{
    uint64_t emb_dim, num_negs, num_users, num_items, train_size;
    uint64_t neg_sampler; /* 0 uniform, 1 random tile */
    uint64_t tile_size, refresh_interval, num_subepochs;
    float    l2, clip_val;
    const uint64_t* milestones;
    uint64_t n_milestones;
    float    l_r;
} orc_config;

/* ---- negative_samplers/{uniform_random,random_tile}_negative_sampler.cpp ---- */
typedef struct
{
    uint64_t    num_negs;
    int         is_tile;
    orc_uniform neg_sampler;  /* Uniform(num_items-1, seed) */
    orc_uniform tile_sampler; /* Uniform(tile_size-1, seed) (tile sampler only) */
    uint64_t    tile_size, refresh_interval, iterations;
    uint64_t*   neg_tile;
} orc_sampler;

void orc_sampler_init(orc_sampler* s, const orc_config* cfg, uint64_t seed, int is_tile);
void orc_sampler_free(orc_sampler* s);
void orc_sampler_sampling(orc_sampler* s, uint64_t* neg_ids);
void orc_sampler_ignore_pos_sampling(orc_sampler* s, uint64_t user_id, uint64_t pos_id, uint64_t* neg_ids);

/* ---- optimizers/{optimizer,sgd}.cpp ---- */
float orc_clip_grad(float grad, float clip_val);
void  orc_sparse_step(float* emb, float* grad, uint64_t emb_dim, float clip_val, float l_r);
float orc_scheduler_step_lr(float l_r, uint64_t epoch, uint64_t step_size, float gamma);
float orc_scheduler_multi_step_lr(float l_r, uint64_t epoch, const uint64_t* milestones, uint64_t n, float gamma);

/* ---- engine / model state ---- */
typedef struct
{
    orc_config cfg;
    const uint64_t* clicks;   /* [data_rows,2] borrowed */
    uint64_t        data_rows;
    const uint64_t* his;      /* [num_users,max_his] borrowed (may be NULL if !use_aggregator) */
    const uint64_t* masks;    /* [num_users,1] borrowed */
    uint64_t        max_his;
    float* user_w;            /* borrowed, trained in place (model.cpp:12-13, array.hpp:30-34) */
    float* item_w;
    float* user_g;            /* owned, zero-init, persistent within an epoch (embedding.cpp:12-13) */
    float* item_g;
    float* w0;                /* [d,d] row-major, borrowed (behavior_aggregators.cpp:19) */
    int    use_aggregator;    /* reference: always 1 (matrix_factorization.cpp:38,152); 0 = MF-CCL scope */
    float  l_r;               /* optimizer's current learning rate */
    uint64_t epoch;
} orc_engine;

orc_engine* orc_engine_create(const orc_config* cfg, const uint64_t* clicks, uint64_t data_rows, const uint64_t* his,
                              uint64_t max_his, const uint64_t* masks, float* user_w, float* item_w, float* w0,
                              int use_aggregator);
void orc_engine_destroy(orc_engine* e);
void orc_engine_zero_grad(orc_engine* e);
/* engine.cpp:156-160 : applies the LR schedule for e->epoch (does not advance the epoch) */
void orc_engine_lr_step(orc_engine* e);

/* per-thread state: ThreadBuffer (thread_buffer.hpp:16-60) + BehaviorAggregator (behavior_aggregators.cpp:28-46) */
typedef struct
{
    orc_engine* e;
    float *user_emb, *user_grad, *pos_emb, *pos_grad, *neg_embs, *neg_grad;
    float *upu, *upp, *und, *nnd, *nn_, *nn3, *score, *es, *lg;
    /* aggregator */
    uint64_t iteration, mini_batch_size;
    float    gamma, agg_l_r;
    float *means, *w0_grad_accu, *f_c0;
    int      have_means;
} orc_worker;

orc_worker* orc_worker_create(orc_engine* e);
void        orc_worker_destroy(orc_worker* w);
/* models/matrix_factorization.cpp:15-181 : one interaction, returns the loss */
float orc_forward_backward(orc_worker* w, uint64_t user_id, uint64_t pos_id, const uint64_t* neg_ids);

/* Serial walk over interactions [begin,end) in stored order with caller-supplied negatives
 * (neg_ids[(i-begin)*num_negs + k]); returns the double loss sum.  No LR step / zero_grad. */
double orc_train_range(orc_engine* e, orc_worker* w, uint64_t begin, uint64_t end, const uint64_t* neg_ids);

/* train/engine.cpp:294-342 (upstream OpenMP body) + :156-160 + :345-347 + :378-385.
 * sampler_call: 0 = ignore_pos_sampling (engine.cpp:332, the live call), 1 = sampling (:333).
 * num_threads <= 0: omp_get_max_threads().  neg_out (optional, [data_rows,num_negs]) records the
 * negatives actually used.  Returns mean loss as val_t. */
float orc_train_one_epoch(orc_engine* e, int num_threads, int sampler_call, uint64_t* neg_out);

/* train/engine.cpp:388-400 : sim[num_users,num_items] = U * V^T (fp32) */
void orc_evaluate0(const orc_engine* e, float* sim);

#ifdef __cplusplus
}
#endif
#endif
