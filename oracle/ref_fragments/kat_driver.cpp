// TEST INFRASTRUCTURE — not product code.
//
// Known-answer-test (KAT) driver for the Eigen-free translation units of the
// reference (visuOwO/HEAT).  This file is OUR code; it only #includes the
// reference's public headers and is linked against the reference's own .cpp/.c
// files compiled *where they lie* under /root/reference (see oracle/Makefile,
// target `ref`).  Nothing from the reference is copied into this repository.
//
// Reference units exercised (paths relative to /root/reference/cf_cpu/src):
//   modules/random/uniform.hpp:16-30                      Uniform (mt19937_64)
//   modules/negative_samplers/uniform_random_negative_sampler.cpp:10-36
//   modules/negative_samplers/random_tile_negative_sampler.cpp:11-57
//   modules/optimizers/sgd.cpp:14-26, optimizer.cpp:17-38 clip / SGD / LR sched
//   modules/embeddings/embedding.cpp:10-45, memory/array.hpp:13-61
//   modules/datasets/click_dataset.cpp:17-22
//   splatt/base.c:12-41, splatt/util.c:6-19
//
// Output: one JSON document (argv[1]) that tests/golden/gen_ref_kats.py copies
// to tests/golden/ref_kats.json.  The 4 Eigen-dependent units (model,
// matrix_factorization, behavior_aggregators, engine) are unbuildable here
// (Eigen 3.4 submodule is empty, Eigen not installed) and are NOT stubbed.
#include <cstdio>
#include <cstdint>
#include <memory>
#include <vector>
#include <string>

#include "modules/cf_config.hpp"
#include "modules/random/uniform.hpp"
#include "modules/random/shuffle.hpp"
#include "modules/negative_samplers/uniform_random_negative_sampler.hpp"
#include "modules/negative_samplers/random_tile_negative_sampler.hpp"
#include "modules/optimizers/sgd.hpp"
#include "modules/embeddings/embedding.hpp"
#include "modules/datasets/click_dataset.hpp"

using cf::modules::CFConfig;
namespace ns = cf::modules::negative_samplers;
namespace opt = cf::modules::optimizers;

static FILE* out;
static bool first_key = true;

static void key(const char* k)
{
    fprintf(out, "%s\n  \"%s\": ", first_key ? "" : ",", k);
    first_key = false;
}
static void arr_u64(const std::vector<uint64_t>& v)
{
    fprintf(out, "[");
    for (size_t i = 0; i < v.size(); ++i) fprintf(out, "%s%llu", i ? ", " : "", (unsigned long long)v[i]);
    fprintf(out, "]");
}
// floats are emitted as their IEEE-754 bit pattern so the fixture is bit-exact
static void arr_f32_bits(const std::vector<float>& v)
{
    fprintf(out, "[");
    for (size_t i = 0; i < v.size(); ++i)
    {
        uint32_t b;
        memcpy(&b, &v[i], 4);
        fprintf(out, "%s%u", i ? ", " : "", b);
    }
    fprintf(out, "]");
}

static std::shared_ptr<CFConfig> make_cfg(idx_t emb_dim, idx_t num_negs, idx_t num_items, idx_t tile, idx_t refresh,
                                          val_t clip, val_t lr, std::vector<idx_t> milestones)
{
    return std::make_shared<CFConfig>(emb_dim, num_negs, 1000, num_items, 5000, 0, tile, refresh, 2, 1e-7f, clip,
                                      milestones, lr);
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s out.json\n", argv[0]); return 2; }
    out = fopen(argv[1], "w");
    if (!out) { perror("fopen"); return 2; }
    fprintf(out, "{");

    key("sizeof_idx_t"); fprintf(out, "%zu", sizeof(idx_t));
    key("sizeof_val_t"); fprintf(out, "%zu", sizeof(val_t));

    // ---- Uniform(max_idx, seed).read() streams ------------------------------------------
    {
        struct { idx_t max_idx, seed; int n; } cases[] = {
            {91598, 0, 64}, {91598, 1, 64}, {91598, 7, 64}, {511, 3, 64}, {0, 5, 8}, {1, 9, 32},
            {999999, 2022, 64}, {40980, 14, 64}, {38047, 16, 64}, {UINT64_MAX - 1, 4, 16}, {UINT64_MAX, 4, 16},
            {(1ull << 63), 11, 32},   // threshold/rejection path is hot for ranges near 2^63
            {(1ull << 63) + 12345, 12, 32},
        };
        key("uniform_streams");
        fprintf(out, "[");
        for (size_t c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c)
        {
            cf::modules::random::Uniform u(cases[c].max_idx, cases[c].seed);
            std::vector<uint64_t> v;
            for (int i = 0; i < cases[c].n; ++i) v.push_back(u.read());
            fprintf(out, "%s\n    {\"max_idx\": %llu, \"seed\": %llu, \"values\": ", c ? "," : "",
                    (unsigned long long)cases[c].max_idx, (unsigned long long)cases[c].seed);
            arr_u64(v);
            fprintf(out, "}");
        }
        fprintf(out, "]");
    }

    // ---- UniformRandomNegativeSampler ----------------------------------------------------
    {
        key("uniform_sampler");
        fprintf(out, "[");
        struct { idx_t num_items, num_negs, seed; } cases[] = {{91599, 16, 0}, {91599, 16, 1}, {38048, 64, 5}, {17, 4, 3}};
        for (size_t c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c)
        {
            auto cfg = make_cfg(64, cases[c].num_negs, cases[c].num_items, 512, 8192, 1.0f, 0.01f, {10});
            // (a) sampling(): 3 consecutive calls
            ns::UniformRandomNegativeSampler s(cfg, cases[c].seed);
            std::vector<idx_t> neg(cases[c].num_negs);
            fprintf(out, "%s\n    {\"num_items\": %llu, \"num_negs\": %llu, \"seed\": %llu, \"sampling\": [", c ? "," : "",
                    (unsigned long long)cases[c].num_items, (unsigned long long)cases[c].num_negs,
                    (unsigned long long)cases[c].seed);
            for (int call = 0; call < 3; ++call)
            {
                s.sampling(neg);
                fprintf(out, "%s", call ? ", " : "");
                arr_u64(std::vector<uint64_t>(neg.begin(), neg.end()));
            }
            fprintf(out, "],");
            // (b) ignore_pos_sampling(): learn the raw stream first, then pick pos ids that collide with
            //     specific draws so that the "slot left unchanged" behaviour is exercised.
            ns::UniformRandomNegativeSampler probe(cfg, cases[c].seed);
            std::vector<idx_t> raw(cases[c].num_negs);
            std::vector<uint64_t> pos_ids;
            std::vector<std::vector<uint64_t>> raws;
            for (int call = 0; call < 4; ++call)
            {
                probe.sampling(raw);
                raws.push_back(std::vector<uint64_t>(raw.begin(), raw.end()));
            }
            // call 0: collide with slot 2 (slot keeps initial 0); call 1: collide with slot 0;
            // call 2: no collision (pos = num_items, out of range); call 3: collide with last slot.
            pos_ids.push_back(raws[0][2]);
            pos_ids.push_back(raws[1][0]);
            pos_ids.push_back(cases[c].num_items);
            pos_ids.push_back(raws[3][cases[c].num_negs - 1]);
            ns::UniformRandomNegativeSampler s2(cfg, cases[c].seed);
            std::vector<idx_t> neg2(cases[c].num_negs);   // zero-initialised like engine.cpp:298
            fprintf(out, " \"pos_ids\": ");
            arr_u64(pos_ids);
            fprintf(out, ", \"ignore_pos_sampling\": [");
            for (int call = 0; call < 4; ++call)
            {
                s2.ignore_pos_sampling(123, pos_ids[call], neg2);
                fprintf(out, "%s", call ? ", " : "");
                arr_u64(std::vector<uint64_t>(neg2.begin(), neg2.end()));
            }
            fprintf(out, "]}");
        }
        fprintf(out, "]");
    }

    // ---- RandomTileNegativeSampler ---------------------------------------------------------
    {
        key("tile_sampler");
        fprintf(out, "[");
        struct { idx_t num_items, num_negs, tile, refresh, seed; int calls; } cases[] = {
            {91599, 16, 512, 8192, 3, 3}, {91599, 16, 8, 2, 4, 6}, {40981, 16, 512, 3, 9, 7}};
        for (size_t c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c)
        {
            auto cfg = make_cfg(64, cases[c].num_negs, cases[c].num_items, cases[c].tile, cases[c].refresh, 1.0f, 0.01f, {10});
            ns::RandomTileNegativeSampler s(cfg, cases[c].seed);
            std::vector<idx_t> neg(cases[c].num_negs);
            fprintf(out, "%s\n    {\"num_items\": %llu, \"num_negs\": %llu, \"tile_size\": %llu, \"refresh_interval\": %llu, \"seed\": %llu, \"sampling\": [",
                    c ? "," : "", (unsigned long long)cases[c].num_items, (unsigned long long)cases[c].num_negs,
                    (unsigned long long)cases[c].tile, (unsigned long long)cases[c].refresh, (unsigned long long)cases[c].seed);
            for (int call = 0; call < cases[c].calls; ++call)
            {
                s.sampling(neg);
                fprintf(out, "%s", call ? ", " : "");
                arr_u64(std::vector<uint64_t>(neg.begin(), neg.end()));
            }
            fprintf(out, "], \"tile_after\": ");
            arr_u64(std::vector<uint64_t>(s.neg_tile.begin(), s.neg_tile.end()));
            // ignore_pos_sampling of the tile sampler does not use the tile (random_tile_negative_sampler.cpp:47-57)
            ns::RandomTileNegativeSampler s2(cfg, cases[c].seed);
            std::vector<idx_t> neg2(cases[c].num_negs);
            s2.ignore_pos_sampling(1, cases[c].num_items, neg2);
            fprintf(out, ", \"ignore_pos_first\": ");
            arr_u64(std::vector<uint64_t>(neg2.begin(), neg2.end()));
            fprintf(out, "}");
        }
        fprintf(out, "]");
    }

    // ---- SGD::sparse_step / clip_grad / LR schedulers ------------------------------------------
    {
        key("sgd");
        fprintf(out, "[");
        struct { idx_t d; val_t clip, lr; } cases[] = {{64, 1.0f, 0.01f}, {128, 0.1f, 0.01f}, {7, 0.5f, 0.3f}};
        for (size_t c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c)
        {
            auto cfg = make_cfg(cases[c].d, 16, 91599, 512, 8192, cases[c].clip, cases[c].lr, {10});
            opt::SGD sgd(cfg);
            std::vector<float> e(cases[c].d), g(cases[c].d);
            for (idx_t i = 0; i < cases[c].d; ++i)
            {
                e[i] = 0.01f * (float)i - 0.2f;
                g[i] = 0.1f * ((float)i - (float)cases[c].d / 2.0f) * ((i % 3 == 0) ? -1.0f : 1.0f);
            }
            std::vector<float> e0 = e, g0 = g;
            sgd.sparse_step(e.data(), g.data());
            std::vector<float> e1 = e, g1 = g;
            sgd.sparse_step(e.data(), g.data());   // second touch: starts from the stored (clamped) grad
            fprintf(out, "%s\n    {\"d\": %llu, \"clip\": %u, \"lr\": %u, \"e0\": ", c ? "," : "", (unsigned long long)cases[c].d,
                    *(uint32_t*)&cases[c].clip, *(uint32_t*)&cases[c].lr);
            arr_f32_bits(e0); fprintf(out, ", \"g0\": "); arr_f32_bits(g0);
            fprintf(out, ", \"e1\": "); arr_f32_bits(e1); fprintf(out, ", \"g1\": "); arr_f32_bits(g1);
            fprintf(out, ", \"e2\": "); arr_f32_bits(e); fprintf(out, ", \"g2\": "); arr_f32_bits(g);
            fprintf(out, "}");
        }
        fprintf(out, "]");

        key("clip_grad");
        {
            auto cfg = make_cfg(64, 16, 91599, 512, 8192, 1.0f, 0.01f, {10});
            opt::SGD sgd(cfg);
            std::vector<float> in = {-3.0f, -1.0f, -0.999f, -0.0f, 0.0f, 0.5f, 1.0f, 1.0001f, 1e30f, -1e30f};
            std::vector<float> o;
            for (float x : in) o.push_back(sgd.clip_grad(x, 1.0f));
            fprintf(out, "{\"in\": "); arr_f32_bits(in); fprintf(out, ", \"clip\": 1065353216, \"out\": "); arr_f32_bits(o); fprintf(out, "}");
        }

        key("step_lr");
        {
            // engine.cpp:156-160: one milestone -> scheduler_step_lr(epoch, milestones[0], 0.1)
            auto cfg = make_cfg(64, 16, 91599, 512, 8192, 1.0f, 0.01f, {2});
            opt::SGD sgd(cfg);
            std::vector<float> lrs;
            for (idx_t epoch = 0; epoch < 7; ++epoch) { sgd.scheduler_step_lr(epoch, 2, 0.1f); lrs.push_back(sgd.l_r); }
            fprintf(out, "{\"lr0\": %u, \"step\": 2, \"gamma\": %u, \"lr_after_epoch\": ", *(uint32_t*)&cfg->l_r, 1036831949u);
            arr_f32_bits(lrs); fprintf(out, "}");
        }
        key("multi_step_lr");
        {
            std::vector<idx_t> ms = {1, 4, 5};
            auto cfg = make_cfg(64, 16, 91599, 512, 8192, 1.0f, 0.01f, ms);
            opt::SGD sgd(cfg);
            std::vector<float> lrs;
            for (idx_t epoch = 0; epoch < 7; ++epoch) { sgd.scheduler_multi_step_lr(epoch, ms, 0.1f); lrs.push_back(sgd.l_r); }
            fprintf(out, "{\"lr0\": %u, \"milestones\": [1, 4, 5], \"lr_after_epoch\": ", *(uint32_t*)&cfg->l_r);
            arr_f32_bits(lrs); fprintf(out, "}");
        }
    }

    // ---- Embedding / Array / ClickDataset / Shuffle ------------------------------------------------
    {
        key("embedding");
        std::vector<float> w(5 * 4);
        for (size_t i = 0; i < w.size(); ++i) w[i] = 0.5f * (float)i;
        cf::modules::embeddings::Embedding emb(5, 4, w.data());
        std::vector<float> buf(4), g(4, 0.0f);
        emb.read_weights(3, buf.data());
        std::vector<float> row3 = buf;
        buf[1] = -7.0f;
        emb.write_weights(1, buf.data());
        emb.read_grads(2, g.data());
        std::vector<float> g_init = g;           // owned grads are zero-initialised (array.hpp:22-24)
        g = {1, 2, 3, 4};
        emb.write_grads(2, g.data());
        std::vector<float> gall(emb.grads->data, emb.grads->data + 20);
        emb.zero_grad();
        std::vector<float> gzero(emb.grads->data, emb.grads->data + 20);
        fprintf(out, "{\"row3\": "); arr_f32_bits(row3);
        fprintf(out, ", \"weights_after_write\": "); arr_f32_bits(w);   // borrowed: caller's buffer mutated in place
        fprintf(out, ", \"grad_init_row2\": "); arr_f32_bits(g_init);
        fprintf(out, ", \"grads_after_write\": "); arr_f32_bits(gall);
        fprintf(out, ", \"grads_after_zero\": "); arr_f32_bits(gzero);
        fprintf(out, ", \"grads_aligned64\": %d}", (int)(((uintptr_t)emb.grads->data % 64) == 0));

        key("click_dataset");
        std::vector<idx_t> clicks = {0, 10, 0, 11, 1, 12, 2, 13, 2, 14};
        std::vector<idx_t> his(3 * 2, 0), masks(3, 1);
        cf::modules::datasets::ClickDataset ds(5, 2, clicks.data(), 3, 2, his.data(), 3, 1, masks.data());
        std::vector<uint64_t> ui;
        for (idx_t i = 0; i < 5; ++i) { idx_t u, it; ds.read_user_item(i, u, it); ui.push_back(u); ui.push_back(it); }
        fprintf(out, "{\"pairs\": "); arr_u64(ui); fprintf(out, ", \"data_rows\": %llu, \"max_his_default\": %d}",
                (unsigned long long)ds.data_rows, ds.max_his);

        key("shuffle_identity");
        cf::modules::random::Shuffle sh(6);
        std::vector<uint64_t> idx;
        for (idx_t i = 0; i < 6; ++i) idx.push_back(sh.read(i));
        arr_u64(idx);
    }

    fprintf(out, "\n}\n");
    fclose(out);
    return 0;
}
