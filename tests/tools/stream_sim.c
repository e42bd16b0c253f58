/* TEST INFRASTRUCTURE (analysis aid, uses the CPU oracle) — NOT product code.
 *
 * A sequentially-consistent model of how the GPU engine schedules the interaction list: S streams, each with its own
 * per-worker state (ThreadBuffer + BehaviorAggregator of the reference, train/engine.cpp:313-318), advanced in lockstep,
 * one interaction per stream per round.  Every step sees the tables as the previous step left them, so what this isolates
 * is the ORDER in which interactions are applied and the number of per-worker aggregators — not staleness or lost updates.
 *   layout 0: stream s walks the contiguous slice [s*per, (s+1)*per)              (the engine's launch geometry)
 *   layout 1: chunks of `chunk` interactions dealt round-robin, chunk c to stream c % S: all streams sweep the list
 *             together, the idealised `#pragma omp for schedule(dynamic, chunk)` of train/engine.cpp:327
 * mb: calls a worker accumulates before it applies its W0 step (32 = behavior_aggregators.cpp:36,141-146).
 * Negatives: a counter-based generator keyed by (seed, epoch, interaction index, slot), i.e. independent of S and layout; or
 * (worker_sampler) the reference's own per-worker sampler through its sampling() call — the random-tile sampler when
 * cfg.neg_sampler == 1, one tile per worker refreshed every refresh_interval calls of that worker.
 */
#include "../../oracle/cf_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

double sim_epoch(orc_engine* e, int S, int layout, uint64_t chunk, uint64_t seed, int mb, int worker_sampler)
{
    const uint64_t n = e->data_rows, N = e->cfg.num_negs, I = e->cfg.num_items;
    orc_worker** w = (orc_worker**)calloc((size_t)S, sizeof(orc_worker*));
    uint64_t* neg = (uint64_t*)calloc(N, sizeof(uint64_t));
    orc_sampler* samp = (orc_sampler*)calloc((size_t)S, sizeof(orc_sampler));
    for (int s = 0; s < S; ++s)
    {
        w[s] = orc_worker_create(e);
        /* train/engine.cpp:302-311: one sampler per worker, seeded (epoch + 1) * worker id; the tile sampler when cfg.neg_sampler == 1 */
        if (worker_sampler)
            orc_sampler_init(&samp[s], &e->cfg, worker_sampler == 2 ? mix64(seed * 7919ull + e->epoch * 1000003ull + (uint64_t)s) /* no seed shared by two (epoch, worker) pairs */
                                                                    : (e->epoch + 1) * (uint64_t)s, e->cfg.neg_sampler == 1);
        if (mb > 0 && mb != 32)
        {
            /* the same W0 step per call, lr/32 * (means (x) f_grad), applied every `mb` calls of the worker instead of every 32 */
            w[s]->mini_batch_size = (uint64_t)mb;
            w[s]->agg_l_r = w[s]->agg_l_r * (float)mb / 32.0f;
        }
    }
    orc_engine_lr_step(e);
    double loss = 0.0;
    const uint64_t key = mix64(seed * 1000003ull + e->epoch);
#define STEP(s, i)                                                                              \
    do                                                                                          \
    {                                                                                           \
        const uint64_t u_ = e->clicks[2 * (i)], p_ = e->clicks[2 * (i) + 1];                    \
        if (worker_sampler) orc_sampler_sampling(&samp[s], neg); /* train/engine.cpp:333 */     \
        else for (uint64_t k = 0; k < N; ++k)                                                   \
        {                                                                                       \
            uint64_t id = (uint64_t)(((unsigned __int128)mix64(key ^ mix64((i) * 131ull + k)) * I) >> 64); \
            if (id == p_) id = (id + 1) % I;                                                    \
            neg[k] = id;                                                                        \
        }                                                                                       \
        loss += (double)orc_forward_backward(w[s], u_, p_, neg);                                \
    } while (0)
    if (layout == 0)
    {
        uint64_t per = (n + (uint64_t)S - 1) / (uint64_t)S;
        per = ((per + 63) / 64) * 64;
        for (uint64_t t = 0; t < per; ++t)
            for (int s = 0; s < S; ++s)
            {
                const uint64_t i = (uint64_t)s * per + t;
                if (i < n && i < ((uint64_t)s + 1) * per) STEP(s, i);
            }
    }
    else
    {
        const uint64_t nchunks = (n + chunk - 1) / chunk;
        for (uint64_t c0 = 0; c0 < nchunks; c0 += (uint64_t)S)
            for (uint64_t t = 0; t < chunk; ++t)
                for (int s = 0; s < S; ++s)
                {
                    const uint64_t i = (c0 + (uint64_t)s) * chunk + t;
                    if (c0 + (uint64_t)s < nchunks && i < n) STEP(s, i);
                }
    }
#undef STEP
    for (int s = 0; s < S; ++s) orc_worker_destroy(w[s]);
    if (worker_sampler) for (int s = 0; s < S; ++s) orc_sampler_free(&samp[s]);
    free(samp);
    free(w);
    free(neg);
    orc_engine_zero_grad(e);
    e->epoch += 1;
    return loss / (double)n;
}
