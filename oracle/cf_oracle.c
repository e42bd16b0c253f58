/* TEST INFRASTRUCTURE — CPU oracle, NOT product code.  See cf_oracle.h for scope,
 * citation conventions and pinning status.  All `file:line` citations are relative
 * to /root/reference/cf_cpu/src. */
#define _POSIX_C_SOURCE 200809L
#include "cf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* ------------------------------------------------------------------------------------------
 * std::mt19937_64 (ISO C++ [rand.predef]; Matsumoto & Nishimura's MT19937-64 parameters).
 * random/uniform.hpp:19 `rng.seed(seed)`, :28 `dist(rng)`.
 * ------------------------------------------------------------------------------------------ */
#define MT_N 312
#define MT_M 156
#define MT_A 0xB5026F5AA96619E9ull
#define MT_UPPER 0xFFFFFFFF80000000ull
#define MT_LOWER 0x000000007FFFFFFFull

void orc_mt64_seed(orc_mt64* g, uint64_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < MT_N; ++i)
        g->mt[i] = 6364136223846793005ull * (g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) + (uint64_t)i;
    g->idx = MT_N;
}

uint64_t orc_mt64_next(orc_mt64* g)
{
    if (g->idx >= MT_N)
    {
        for (int i = 0; i < MT_N; ++i)
        {
            uint64_t x = (g->mt[i] & MT_UPPER) | (g->mt[(i + 1) % MT_N] & MT_LOWER);
            uint64_t xa = x >> 1;
            if (x & 1ull) xa ^= MT_A;
            g->mt[i] = g->mt[(i + MT_M) % MT_N] ^ xa;
        }
        g->idx = 0;
    }
    uint64_t y = g->mt[g->idx++];
    y ^= (y >> 29) & 0x5555555555555555ull;
    y ^= (y << 17) & 0x71D67FFFEDA60000ull;
    y ^= (y << 37) & 0xFFF7EEE000000000ull;
    y ^= (y >> 43);
    return y;
}

/* random/uniform.hpp:16-30.  std::uniform_int_distribution<uint64_t>(0,max_idx) over a 64-bit
 * URBG is, in the libstdc++ that builds the reference here (GCC 11.4, bits/uniform_int_dist.h
 * `_S_nd`), Lemire's nearly-divisionless method (ACM TOMACS 29(1), 2019) on a 128-bit product;
 * a full-range request returns the raw draw.  Pinned by tests/golden/ref_kats.json. */
void orc_uniform_init(orc_uniform* u, uint64_t max_idx, uint64_t seed)
{
    u->max_idx = max_idx;
    orc_mt64_seed(&u->rng, seed);
}

uint64_t orc_uniform_read(orc_uniform* u)
{
    const uint64_t urange = u->max_idx; /* b - a, a = 0 */
    if (urange == UINT64_MAX) return orc_mt64_next(&u->rng);
    const uint64_t range = urange + 1;
    unsigned __int128 product = (unsigned __int128)orc_mt64_next(&u->rng) * (unsigned __int128)range;
    uint64_t low = (uint64_t)product;
    if (low < range)
    {
        uint64_t threshold = (0 - range) % range;
        while (low < threshold)
        {
            product = (unsigned __int128)orc_mt64_next(&u->rng) * (unsigned __int128)range;
            low = (uint64_t)product;
        }
    }
    return (uint64_t)(product >> 64);
}

/* ------------------------------------------------------------------------------------------
 * negative samplers
 * ------------------------------------------------------------------------------------------ */
/* uniform_random_negative_sampler.cpp:10-15, random_tile_negative_sampler.cpp:11-21 */
void orc_sampler_init(orc_sampler* s, const orc_config* cfg, uint64_t seed, int is_tile)
{
    memset(s, 0, sizeof(*s));
    s->num_negs = cfg->num_negs;
    s->is_tile = is_tile;
    orc_uniform_init(&s->neg_sampler, cfg->num_items - 1, seed);
    if (is_tile)
    {
        orc_uniform_init(&s->tile_sampler, cfg->tile_size - 1, seed);
        s->tile_size = cfg->tile_size;
        s->refresh_interval = cfg->refresh_interval;
        s->iterations = 0;
        s->neg_tile = (uint64_t*)calloc(cfg->tile_size, sizeof(uint64_t));
    }
}

void orc_sampler_free(orc_sampler* s)
{
    free(s->neg_tile);
    s->neg_tile = NULL;
}

/* uniform_random_negative_sampler.cpp:17-24 ; random_tile_negative_sampler.cpp:23-45 */
void orc_sampler_sampling(orc_sampler* s, uint64_t* neg_ids)
{
    if (!s->is_tile)
    {
        for (uint64_t i = 0; i < s->num_negs; ++i) neg_ids[i] = orc_uniform_read(&s->neg_sampler);
        return;
    }
    if (s->iterations % s->refresh_interval == 0)
        for (uint64_t i = 0; i < s->tile_size; ++i) s->neg_tile[i] = orc_uniform_read(&s->neg_sampler);
    for (uint64_t i = 0; i < s->num_negs; ++i) neg_ids[i] = s->neg_tile[orc_uniform_read(&s->tile_sampler)];
    s->iterations += 1;
}

/* uniform_random_negative_sampler.cpp:26-36 ; random_tile_negative_sampler.cpp:47-57 (identical:
 * the tile sampler's version does NOT use the tile).  A draw equal to pos_id leaves the slot
 * unchanged (it keeps the previous sample's id, or the initial 0). */
void orc_sampler_ignore_pos_sampling(orc_sampler* s, uint64_t user_id, uint64_t pos_id, uint64_t* neg_ids)
{
    (void)user_id;
    for (uint64_t i = 0; i < s->num_negs; ++i)
    {
        uint64_t neg_id = orc_uniform_read(&s->neg_sampler);
        if (neg_id != pos_id) neg_ids[i] = neg_id;
    }
}

/* ------------------------------------------------------------------------------------------
 * optimizer
 * ------------------------------------------------------------------------------------------ */
/* optimizers/optimizer.cpp:17-22 : std::min then std::max */
float orc_clip_grad(float grad, float clip_val)
{
    float c = (clip_val < grad) ? clip_val : grad; /* std::min(grad, clip_val) */
    const float neg = -clip_val;
    c = (c < neg) ? neg : c;                       /* std::max(c, -clip_val) */
    return c;
}

/* optimizers/sgd.cpp:14-26 : grad <- clip(grad); emb -= lr * grad; clamped grad stored back */
void orc_sparse_step(float* emb, float* grad, uint64_t emb_dim, float clip_val, float l_r)
{
    for (uint64_t i = 0; i < emb_dim; ++i)
    {
        grad[i] = orc_clip_grad(grad[i], clip_val);
        emb[i] -= l_r * grad[i];
    }
}

/* optimizers/optimizer.cpp:24-30 */
float orc_scheduler_step_lr(float l_r, uint64_t epoch, uint64_t step_size, float gamma)
{
    if (epoch > 0 && epoch % step_size == 0) l_r = l_r * gamma;
    return l_r;
}

/* optimizers/optimizer.cpp:32-38 */
float orc_scheduler_multi_step_lr(float l_r, uint64_t epoch, const uint64_t* milestones, uint64_t n, float gamma)
{
    for (uint64_t i = 0; i < n; ++i)
        if (milestones[i] == epoch) return l_r * gamma;
    return l_r;
}

/* ------------------------------------------------------------------------------------------
 * engine / embeddings
 * ------------------------------------------------------------------------------------------ */
static void* aligned_zero(size_t bytes)
{
    void* p = NULL;
    if (bytes == 0) bytes = 64;
    if (posix_memalign(&p, 64, bytes) != 0) return NULL; /* splatt/base.c:12-36 : 64-B aligned */
    memset(p, 0, bytes);                                 /* array.hpp:22-24 : par_memset(.,0,.) */
    return p;
}

orc_engine* orc_engine_create(const orc_config* cfg, const uint64_t* clicks, uint64_t data_rows, const uint64_t* his,
                              uint64_t max_his, const uint64_t* masks, float* user_w, float* item_w, float* w0,
                              int use_aggregator)
{
    orc_engine* e = (orc_engine*)calloc(1, sizeof(orc_engine));
    e->cfg = *cfg;
    if (cfg->n_milestones)
    {
        uint64_t* ms = (uint64_t*)malloc(cfg->n_milestones * sizeof(uint64_t));
        memcpy(ms, cfg->milestones, cfg->n_milestones * sizeof(uint64_t));
        e->cfg.milestones = ms;
    }
    e->clicks = clicks;
    e->data_rows = data_rows;
    e->his = his;
    e->max_his = max_his;
    e->masks = masks;
    e->user_w = user_w;
    e->item_w = item_w;
    /* embeddings/embedding.cpp:12-13 : grads = new ValArray(num_embs, emb_dim, nullptr) -> owned, zeroed */
    e->user_g = (float*)aligned_zero(cfg->num_users * cfg->emb_dim * sizeof(float));
    e->item_g = (float*)aligned_zero(cfg->num_items * cfg->emb_dim * sizeof(float));
    e->w0 = w0;
    e->use_aggregator = use_aggregator;
    e->l_r = cfg->l_r; /* optimizers/optimizer.cpp:13 */
    e->epoch = 0;      /* train/engine.cpp:19 */
    return e;
}

void orc_engine_destroy(orc_engine* e)
{
    if (!e) return;
    free(e->user_g);
    free(e->item_g);
    free((void*)e->cfg.milestones);
    free(e);
}

/* embeddings/embedding.cpp:41-45 via train/engine.cpp:345-347 */
void orc_engine_zero_grad(orc_engine* e)
{
    memset(e->user_g, 0, e->cfg.num_users * e->cfg.emb_dim * sizeof(float));
    memset(e->item_g, 0, e->cfg.num_items * e->cfg.emb_dim * sizeof(float));
}

/* train/engine.cpp:156-160 */
void orc_engine_lr_step(orc_engine* e)
{
    if (e->cfg.n_milestones > 1)
        e->l_r = orc_scheduler_multi_step_lr(e->l_r, e->epoch, e->cfg.milestones, e->cfg.n_milestones, 0.1f);
    else
        e->l_r = orc_scheduler_step_lr(e->l_r, e->epoch, e->cfg.milestones[0], 0.1f);
}

static float* falloc(size_t n) { return (float*)aligned_zero(n * sizeof(float)); }

orc_worker* orc_worker_create(orc_engine* e)
{
    const uint64_t d = e->cfg.emb_dim, N = e->cfg.num_negs;
    orc_worker* w = (orc_worker*)calloc(1, sizeof(orc_worker));
    w->e = e;
    /* memory/thread_buffer.hpp:23-30 */
    w->user_emb = falloc(d);
    w->user_grad = falloc(d);
    w->pos_emb = falloc(d);
    w->pos_grad = falloc(d);
    w->neg_embs = falloc(N * d);
    w->neg_grad = falloc(d);
    w->upu = falloc(d);
    w->upp = falloc(d);
    w->und = falloc(N);
    w->nnd = falloc(N);
    w->nn_ = falloc(N);
    w->nn3 = falloc(N);
    w->score = falloc(N);
    w->es = falloc(N);
    w->lg = falloc(N);
    /* behavior_aggregators/behavior_aggregators.cpp:28-46 */
    w->iteration = 0;
    w->mini_batch_size = 32;
    w->gamma = 0.4f;            /* :37 (double literal 0.4 stored into val_t) */
    w->agg_l_r = e->cfg.l_r;    /* :38 frozen at construction from the CONFIG, not the scheduled lr */
    w->means = falloc(d);
    w->f_c0 = falloc(d);
    w->w0_grad_accu = falloc(d * d);
    w->have_means = 0;
    return w;
}

void orc_worker_destroy(orc_worker* w)
{
    if (!w) return;
    free(w->user_emb); free(w->user_grad); free(w->pos_emb); free(w->pos_grad);
    free(w->neg_embs); free(w->neg_grad); free(w->upu); free(w->upp);
    free(w->und); free(w->nnd); free(w->nn_); free(w->nn3); free(w->score); free(w->es); free(w->lg);
    free(w->means); free(w->f_c0); free(w->w0_grad_accu);
    free(w);
}

static inline float dotf(const float* a, const float* b, uint64_t d)
{
    /* Eigen's vectorised dot has an unspecified summation order; the oracle sums left to right in fp32. */
    float s = 0.0f;
    for (uint64_t i = 0; i < d; ++i) s += a[i] * b[i];
    return s;
}

/* behavior_aggregators/behavior_aggregators.cpp:51-127 */
static void agg_forward(orc_worker* w, uint64_t user_id, float* user_emb)
{
    const orc_engine* e = w->e;
    const uint64_t d = e->cfg.emb_dim;
    const uint64_t* his_ids = e->his + user_id * e->max_his; /* :60 */
    const uint64_t num_his = e->masks[user_id];              /* :61-62 */
    const float r_num_his = (float)(1.0 / (double)num_his);  /* :63 `val_t r = 1.0 / num_his` (double divide) */
    /* :96-105 means = colwise sum of history rows * r_num_his */
    for (uint64_t j = 0; j < d; ++j) w->means[j] = 0.0f;
    for (uint64_t h = 0; h < num_his; ++h)
    {
        const float* row = e->item_w + his_ids[h] * d;
        for (uint64_t j = 0; j < d; ++j) w->means[j] += row[j];
    }
    for (uint64_t j = 0; j < d; ++j) w->means[j] *= r_num_his;
    /* :118 f_c0 = means(1xd) * W0(dxd, row-major) */
    for (uint64_t j = 0; j < d; ++j) w->f_c0[j] = 0.0f;
    for (uint64_t i = 0; i < d; ++i)
    {
        const float m = w->means[i];
        const float* wrow = e->w0 + i * d;
        for (uint64_t j = 0; j < d; ++j) w->f_c0[j] += m * wrow[j];
    }
    /* :122 user_emb = gamma*user_emb + (1-gamma)*f_c0, in place */
    const float g = w->gamma, omg = 1 - w->gamma;
    for (uint64_t j = 0; j < d; ++j) user_emb[j] = g * user_emb[j] + omg * w->f_c0[j];
    w->iteration += 1; /* :124 */
    w->have_means = 1;
}

/* behavior_aggregators/behavior_aggregators.cpp:129-153 */
static void agg_backward(orc_worker* w, float* outs_grad)
{
    orc_engine* e = w->e;
    const uint64_t d = e->cfg.emb_dim;
    const float omg = 1 - w->gamma;
    /* :132 f_c0_grad = outs_grad*(1-gamma) ; :134-139 accu[i,:] += means[i]*f_c0_grad */
    for (uint64_t i = 0; i < d; ++i)
    {
        const float m = w->means[i];
        float* arow = w->w0_grad_accu + i * d;
        for (uint64_t j = 0; j < d; ++j) arow[j] += m * (outs_grad[j] * omg);
    }
    /* :141-146 every mini_batch_size calls: W0 -= lr * (accu / 32); accu = 0  (lock-free on shared W0) */
    if (w->iteration > 0 && (w->iteration % w->mini_batch_size == 0))
    {
        const float mb = (float)w->mini_batch_size;
        for (uint64_t i = 0; i < d * d; ++i)
        {
            e->w0[i] -= w->agg_l_r * (w->w0_grad_accu[i] / mb);
            w->w0_grad_accu[i] = 0.0f;
        }
    }
    /* :148-152 */
    for (uint64_t j = 0; j < d; ++j) outs_grad[j] *= w->gamma;
}

/* models/matrix_factorization.cpp:15-181 */
float orc_forward_backward(orc_worker* w, uint64_t user_id, uint64_t pos_id, const uint64_t* neg_ids)
{
    orc_engine* e = w->e;
    const uint64_t d = e->cfg.emb_dim, N = e->cfg.num_negs;
    const float clip = e->cfg.clip_val, lr = e->l_r;
    float* user_emb = w->user_emb;
    float* pos_emb = w->pos_emb;

    /* :31-32 read_row = memcpy into the thread buffer (memory/array.hpp:46-50) */
    memcpy(user_emb, e->user_w + user_id * d, d * sizeof(float));
    memcpy(pos_emb, e->item_w + pos_id * d, d * sizeof(float));

    /* :38 aggregator forward is unconditional in the reference; use_aggregator=0 is the MF-CCL scope */
    if (e->use_aggregator) agg_forward(w, user_id, user_emb);

    /* :45-47 */
    const float user_user_dot = dotf(user_emb, user_emb, d);
    const float pos_pos_dot = dotf(pos_emb, pos_emb, d);
    const float user_pos_dot = dotf(user_emb, pos_emb, d);

    /* :53-63 */
    const float eps = 1e-8f;
    const float user_norm = sqrtf(user_user_dot > eps ? user_user_dot : eps); /* std::max(uu, eps) */
    const float pos_norm = sqrtf(pos_pos_dot > eps ? pos_pos_dot : eps);
    const float user_norm3 = user_norm * user_norm * user_norm;
    const float pos_norm3 = pos_norm * pos_norm * pos_norm;
    const float r_u3_p = 1 / (user_norm3 * pos_norm);
    const float r_u_p3 = 1 / (user_norm * pos_norm3);
    for (uint64_t i = 0; i < d; ++i)
    {
        w->upu[i] = (user_user_dot * pos_emb[i] - user_pos_dot * user_emb[i]) * r_u3_p;
        w->upp[i] = -(pos_pos_dot * user_emb[i] - user_pos_dot * pos_emb[i]) * r_u_p3;
    }

    /* :69-77 gather negatives (the grad row read at :75-76 is dead) */
    for (uint64_t k = 0; k < N; ++k) memcpy(w->neg_embs + k * d, e->item_w + neg_ids[k] * d, d * sizeof(float));

    /* :82-85 (pos_neg_dots is computed by the reference and never used) */
    for (uint64_t k = 0; k < N; ++k)
    {
        const float* n = w->neg_embs + k * d;
        w->und[k] = dotf(user_emb, n, d);
        w->nnd[k] = dotf(n, n, d);
    }

    /* :91-96 */
    const float user_pos_cos = user_pos_dot / (user_norm * pos_norm);
    for (uint64_t k = 0; k < N; ++k)
    {
        const float sel = (w->nnd[k] < eps) ? eps : w->nnd[k];
        w->nn_[k] = sqrtf(sel);
        w->nn3[k] = w->nn_[k] * w->nn_[k] * w->nn_[k];
        const float cosk = w->und[k] / (user_norm * w->nn_[k]);
        w->score[k] = cosk - user_pos_cos;
    }

    /* :101-109 ; score_mul is a double that Eigen converts to the array's scalar type (float) */
    const double score_mul_d = 1.0 / 0.07;
    const float score_mul = (float)score_mul_d;
    float max_score = -INFINITY;
    for (uint64_t k = 0; k < N; ++k)
    {
        w->score[k] *= score_mul;
        if (w->score[k] > max_score) max_score = w->score[k];
    }
    float exp_score_sum = 0.0f;
    for (uint64_t k = 0; k < N; ++k)
    {
        w->es[k] = expf(w->score[k] - max_score);
        exp_score_sum += w->es[k];
    }
    exp_score_sum = (float)((double)exp_score_sum + exp(-1.0 * (double)max_score)); /* :106 double exp, float += */
    const float loss = max_score + logf(exp_score_sum);                                /* :107 std::log(float) */
    for (uint64_t k = 0; k < N; ++k) w->lg[k] = (w->es[k] / exp_score_sum) * score_mul;

    /* :118-121 read the persistent grad rows of user and positive BEFORE the negative loop */
    float* user_grad = w->user_grad;
    float* pos_grad = w->pos_grad;
    memcpy(user_grad, e->user_g + user_id * d, d * sizeof(float));
    memcpy(pos_grad, e->item_g + pos_id * d, d * sizeof(float));

    /* :127-150 */
    for (uint64_t k = 0; k < N; ++k)
    {
        const uint64_t neg_id = neg_ids[k];
        float* n = w->neg_embs + k * d;
        float* neg_grad = w->neg_grad;
        memcpy(neg_grad, e->item_g + neg_id * d, d * sizeof(float)); /* :133 fresh read each slot */
        const float r_u3_n = 1 / (user_norm3 * w->nn_[k]);
        const float r_u_n3 = 1 / (user_norm * w->nn3[k]);
        const float lgk = w->lg[k], undk = w->und[k], nndk = w->nnd[k];
        for (uint64_t i = 0; i < d; ++i)
        {
            const float u_n_cos_u_grad = (user_user_dot * n[i] - undk * user_emb[i]) * r_u3_n; /* :138 */
            const float u_n_cos_n_grad = (nndk * user_emb[i] - undk * n[i]) * r_u_n3;          /* :139 raw nn */
            user_grad[i] += lgk * (u_n_cos_u_grad - w->upu[i]);                                /* :141 */
            pos_grad[i] += lgk * w->upp[i];                                                    /* :142 */
            neg_grad[i] += lgk * u_n_cos_n_grad;                                               /* :143 */
        }
        orc_sparse_step(n, neg_grad, d, clip, lr);                    /* :147 */
        memcpy(e->item_w + neg_id * d, n, d * sizeof(float));         /* :148 */
        memcpy(e->item_g + neg_id * d, neg_grad, d * sizeof(float));  /* :149 */
    }

    if (e->use_aggregator) agg_backward(w, user_grad); /* :152 */

    orc_sparse_step(user_emb, user_grad, d, clip, lr); /* :166 */
    orc_sparse_step(pos_emb, pos_grad, d, clip, lr);   /* :169 */

    memcpy(e->user_w + user_id * d, user_emb, d * sizeof(float)); /* :171 (aggregated u written back if agg on) */
    memcpy(e->user_g + user_id * d, user_grad, d * sizeof(float));
    memcpy(e->item_w + pos_id * d, pos_emb, d * sizeof(float));
    memcpy(e->item_g + pos_id * d, pos_grad, d * sizeof(float));
    return loss;
}

double orc_train_range(orc_engine* e, orc_worker* w, uint64_t begin, uint64_t end, const uint64_t* neg_ids)
{
    const uint64_t N = e->cfg.num_negs;
    double loss = 0.0;
    for (uint64_t i = begin; i < end; ++i)
    {
        /* datasets/click_dataset.cpp:17-22 */
        const uint64_t user_id = e->clicks[i * 2 + 0];
        const uint64_t pos_id = e->clicks[i * 2 + 1];
        loss += orc_forward_backward(w, user_id, pos_id, neg_ids + (i - begin) * N);
    }
    return loss;
}

/* train/engine.cpp:294-342 (upstream OpenMP body, commented out in this fork) + :156-160 + :345-347 + :378-385 */
float orc_train_one_epoch(orc_engine* e, int num_threads, int sampler_call, uint64_t* neg_out)
{
    const uint64_t N = e->cfg.num_negs;
    const uint64_t iterations = e->data_rows; /* README.md:94 `iterations 2380730` = train_data->data_rows */
    double local_loss = 0.0;
    if (num_threads <= 0) num_threads = omp_get_max_threads();

    orc_engine_lr_step(e); /* :156-160, uses this->epoch before the increment at :378 */

#pragma omp parallel num_threads(num_threads) reduction(+ : local_loss)
    {
        uint64_t* neg_ids = (uint64_t*)calloc(N, sizeof(uint64_t)); /* :298 zero-initialised vector */
        const uint64_t thread_id = (uint64_t)omp_get_thread_num();
        const uint64_t seed = (e->epoch + 1) * thread_id; /* :302 */
        orc_sampler sampler;
        orc_sampler_init(&sampler, &e->cfg, seed, e->cfg.neg_sampler == 1); /* :305-311 */
        orc_worker* w = orc_worker_create(e);                                /* :313-318 */

#pragma omp for schedule(dynamic, 512) /* :327 */
        for (uint64_t i = 0; i < iterations; ++i)
        {
            const uint64_t idx = i; /* :330 positive_sampler->read(i): identity, shuffle() is never called */
            const uint64_t user_id = e->clicks[idx * 2 + 0]; /* :331 */
            const uint64_t pos_id = e->clicks[idx * 2 + 1];
            if (sampler_call == 0)
                orc_sampler_ignore_pos_sampling(&sampler, user_id, pos_id, neg_ids); /* :332 */
            else
                orc_sampler_sampling(&sampler, neg_ids); /* :333 */
            if (neg_out) memcpy(neg_out + idx * N, neg_ids, N * sizeof(uint64_t));
            local_loss += (double)orc_forward_backward(w, user_id, pos_id, neg_ids); /* :337-339 */
        }
        orc_worker_destroy(w);
        orc_sampler_free(&sampler);
        free(neg_ids);
    }

    orc_engine_zero_grad(e); /* :345-347 */
    e->epoch += 1;           /* :378 */
    return (float)(local_loss / (double)iterations); /* :383-385 */
}

/* Dot product of evaluate0: one fused multiply-add per k, k ascending, from 0.  The reference computes sim = U * V^T with
 * an Eigen GEMM (train/engine.cpp:394-398) whose summation order is unspecified; the order fixed here is the one the
 * fp32 matrix core executes, so the GPU ranking can be checked bit for bit. */
static inline float dot_fma(const float* a, const float* b, uint64_t d)
{
    float acc = 0.0f;
    for (uint64_t k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}

/* train/engine.cpp:388-400 */
void orc_evaluate0(const orc_engine* e, float* sim)
{
    const uint64_t d = e->cfg.emb_dim, U = e->cfg.num_users, I = e->cfg.num_items;
#pragma omp parallel for schedule(static)
    for (uint64_t u = 0; u < U; ++u)
    {
        const float* ur = e->user_w + u * d;
        for (uint64_t i = 0; i < I; ++i) sim[u * I + i] = dot_fma(ur, e->item_w + i * d, d);
    }
}
