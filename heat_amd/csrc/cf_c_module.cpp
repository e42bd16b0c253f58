// cf_c_module.cpp — the reference's pybind11 module surface (`cf_c.modules.*`) on top of the MI355X C ABI.
//
// Mirrors /root/reference/cf_cpu/src/pybind/init_modules.cpp:13-152 and cf_c.cpp:6-9: same module / submodule /
// class names, same constructor keyword arguments, same read-write attributes, same method names and return
// types, so cf/cpp_base.py-style `c_class(**kwargs)` construction (cf/cpp_base.py:7-11) and the epoch loop of
// cf/main.py:103-124 drive it unchanged.  Differences, all deliberate:
//   * numpy buffers are kept alive by the wrapper objects (the reference borrows raw pointers with no keep-alive,
//     init_modules.cpp:48-54) and wrong dtypes / non-contiguous arrays raise ValueError instead of silently binding
//     a temporary converted copy;
//   * the compute runs on the GPU behind include/heat_cf.h; the GIL is released for the epoch;
//   * CFConfig carries extra optional attributes (seed, use_aggregator, flags, coherence, device, num_streams,
//     sample_index_base) that the reference does not have.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/heat_cf.h"

namespace py = pybind11;

namespace
{
using idx_t = uint64_t; // splatt/base.h:49-50
using val_t = float;    // CMakeLists.txt:11

void check(int rc)
{
    if (rc == HEAT_CF_OK) return;
    const std::string msg = heat_cf_last_error();
    if (rc == HEAT_CF_EINVAL) throw std::invalid_argument(msg);
    if (rc == HEAT_CF_ENOMEM) throw std::bad_alloc();
    throw std::runtime_error(msg);
}

template <typename T>
void require(const py::array& a, int ndim, const char* name)
{
    if (!py::dtype::of<T>().is(a.dtype()) && !a.dtype().equal(py::dtype::of<T>()))
        throw std::invalid_argument(std::string(name) + ": wrong dtype (expected " +
                                    std::string(py::str(py::dtype::of<T>())) + ")");
    if (a.ndim() != ndim) throw std::invalid_argument(std::string(name) + ": expected " + std::to_string(ndim) + "-D array");
    if (!(a.flags() & py::array::c_style)) throw std::invalid_argument(std::string(name) + ": must be C-contiguous");
    if (!a.writeable() && std::is_same<T, val_t>::value) throw std::invalid_argument(std::string(name) + ": must be writeable");
}

uint64_t env_u64(const char* name, uint64_t dflt)
{
    const char* v = std::getenv(name);
    return v ? std::strtoull(v, nullptr, 0) : dflt;
}

// cf::modules::CFConfig (modules/cf_config.hpp:12-35)
struct CFConfig
{
    CFConfig(idx_t emb_dim, idx_t num_negs, idx_t num_users, idx_t num_items, idx_t train_size, idx_t neg_sampler,
             idx_t tile_size, idx_t refresh_interval, idx_t num_subepochs, val_t l2, val_t clip_val,
             std::vector<idx_t>& milestones, val_t l_r)
        : emb_dim(emb_dim), num_negs(num_negs), num_users(num_users), num_items(num_items), train_size(train_size),
          neg_sampler(neg_sampler), tile_size(tile_size), refresh_interval(refresh_interval), num_subepochs(num_subepochs),
          l2(l2), clip_val(clip_val), milestones(milestones), l_r(l_r)
    {
        std::cout << "Test in initialize cf_config" << std::endl; // cf_config.hpp:19
        seed = env_u64("HEAT_CF_SEED", 2022);
        use_aggregator = (uint32_t)env_u64("HEAT_CF_USE_AGGREGATOR", 0);
        flags = (uint32_t)env_u64("HEAT_CF_FLAGS", 0);
        coherence = (uint32_t)env_u64("HEAT_CF_COHERENCE", 0);
        num_streams = (uint32_t)env_u64("HEAT_CF_NUM_STREAMS", 0);
        update_mode = (uint32_t)env_u64("HEAT_CF_UPDATE_MODE", 0);
        device = -1;
        sample_index_base = 0;
    }
    idx_t emb_dim, num_negs, num_users, num_items, train_size, neg_sampler, tile_size, refresh_interval, num_subepochs;
    val_t l2, clip_val;
    std::vector<idx_t> milestones;
    val_t l_r;
    // extensions
    uint64_t seed, sample_index_base;
    uint32_t use_aggregator, flags, coherence, num_streams, update_mode;
    int32_t  device;
};

// cf::modules::datasets::Dataset / ClickDataset (datasets/dataset.hpp:15-33, click_dataset.hpp:13-22)
struct Dataset
{
    virtual ~Dataset() = default;
    py::array click_dataset, historical_items, masks;
    idx_t data_rows = 0;
    idx_t data_cols = 0;
    int   max_his = 0; // dataset.cpp:19
};
struct ClickDataset : Dataset
{
    ClickDataset(py::array clicks, py::array his, py::array msk)
    {
        require<idx_t>(clicks, 2, "click_dataset");
        require<idx_t>(his, 2, "historical_items");
        require<idx_t>(msk, 2, "masks");
        if (clicks.shape(1) != 2) throw std::invalid_argument("click_dataset: expected shape [train_size, 2]");
        if (msk.shape(1) != 1 || msk.shape(0) != his.shape(0))
            throw std::invalid_argument("masks: expected shape [num_users, 1] matching historical_items");
        click_dataset = clicks;
        historical_items = his;
        masks = msk;
        data_rows = (idx_t)clicks.shape(0);
        data_cols = (idx_t)clicks.shape(1);
    }
};

// cf::modules::models::Model / MatrixFactorization (models/model.cpp:10-14)
struct Model
{
    virtual ~Model() = default;
    std::shared_ptr<CFConfig> cfg;
    py::array user_weights, item_weights;
};
struct MatrixFactorization : Model
{
    MatrixFactorization(std::shared_ptr<CFConfig> c, py::array uw, py::array iw)
    {
        if (!c) throw std::invalid_argument("cf_config is None");
        require<val_t>(uw, 2, "user_weights");
        require<val_t>(iw, 2, "item_weights");
        if ((idx_t)uw.shape(0) != c->num_users || (idx_t)uw.shape(1) != c->emb_dim)
            throw std::invalid_argument("user_weights: expected shape [num_users, emb_dim]");
        if ((idx_t)iw.shape(0) != c->num_items || (idx_t)iw.shape(1) != c->emb_dim)
            throw std::invalid_argument("item_weights: expected shape [num_items, emb_dim]");
        cfg = c;
        user_weights = uw;
        item_weights = iw;
    }
};

// cf::modules::behavior_aggregators::AggregatorWeights (behavior_aggregators.cpp:19-25)
struct AggregatorWeights
{
    explicit AggregatorWeights(py::array w0)
    {
        require<val_t>(w0, 2, "aggregator_weights0");
        if (w0.shape(0) != w0.shape(1)) throw std::invalid_argument("aggregator_weights0: expected [emb_dim, emb_dim]");
        weights0 = w0;
        emb_dim = (int)w0.shape(0);
    }
    py::array weights0;
    int       emb_dim;
};

// cf::modules::train::Engine (train/engine.hpp:20-50)
struct Engine
{
    Engine(std::shared_ptr<Dataset> ds, std::shared_ptr<AggregatorWeights> aw, std::shared_ptr<Model> m,
           std::shared_ptr<CFConfig> c)
        : dataset(ds), agg(aw), model(m), cfg(c)
    {
        if (!ds || !m || !c) throw std::invalid_argument("dataset, model and cf_config are required");
        if (c->use_aggregator && !aw) throw std::invalid_argument("aggregator_weights is required when use_aggregator is set");
        if ((idx_t)ds->historical_items.shape(0) != c->num_users && c->use_aggregator)
            throw std::invalid_argument("historical_items: expected num_users rows");
        heat_cf_config hc{};
        hc.emb_dim = c->emb_dim;
        hc.num_negs = c->num_negs;
        hc.num_users = c->num_users;
        hc.num_items = c->num_items;
        hc.train_size = c->train_size;
        hc.neg_sampler = c->neg_sampler;
        hc.tile_size = c->tile_size;
        hc.refresh_interval = c->refresh_interval;
        hc.num_subepochs = c->num_subepochs;
        hc.l2 = c->l2;
        hc.clip_val = c->clip_val;
        hc.milestones = c->milestones.data();
        hc.n_milestones = c->milestones.size();
        hc.l_r = c->l_r;
        hc.seed = c->seed;
        hc.sample_index_base = c->sample_index_base;
        hc.use_aggregator = c->use_aggregator;
        hc.flags = c->flags;
        hc.coherence = c->coherence;
        hc.device = c->device;
        hc.num_streams = c->num_streams;
        hc.update_mode = c->update_mode;
        // engine.cpp:79 iterates train_data->data_rows (a Python-settable attribute, init_modules.cpp:60)
        const idx_t rows = std::min<idx_t>(ds->data_rows, (idx_t)ds->click_dataset.shape(0));
        check(heat_cf_engine_create(&hc, static_cast<const uint64_t*>(ds->click_dataset.data()), rows,
                                    static_cast<const uint64_t*>(ds->historical_items.data()),
                                    (uint64_t)ds->historical_items.shape(1),
                                    static_cast<const uint64_t*>(ds->masks.data()),
                                    static_cast<float*>(m->user_weights.mutable_data()),
                                    static_cast<float*>(m->item_weights.mutable_data()),
                                    aw ? static_cast<float*>(aw->weights0.mutable_data()) : nullptr, &handle));
    }
    ~Engine() { heat_cf_engine_destroy(handle); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    val_t train_one_epoch()
    {
        float loss = 0.f;
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = heat_cf_train_one_epoch(handle, &loss);
        }
        check(rc);
        return loss;
    }

    py::array_t<val_t> evaluate0()
    {
        // init_modules.cpp:122-129: a fresh [num_users, num_items] copy
        py::array_t<val_t> sim({(py::ssize_t)cfg->num_users, (py::ssize_t)cfg->num_items});
        int rc;
        {
            float* p = sim.mutable_data();
            py::gil_scoped_release nogil;
            rc = heat_cf_evaluate0(handle, p);
        }
        check(rc);
        return sim;
    }

    py::array_t<uint32_t> topk(uint32_t k, py::object indptr, py::object items)
    {
        py::array_t<uint32_t> out({(py::ssize_t)cfg->num_users, (py::ssize_t)k});
        const uint64_t* ip = nullptr;
        const uint32_t* it = nullptr;
        py::array a_ip, a_it;
        if (!indptr.is_none())
        {
            a_ip = py::array::ensure(indptr);
            a_it = py::array::ensure(items);
            require<uint64_t>(a_ip, 1, "mask_indptr");
            require<uint32_t>(a_it, 1, "mask_items");
            if ((idx_t)a_ip.shape(0) != cfg->num_users + 1) throw std::invalid_argument("mask_indptr: expected num_users+1 entries");
            ip = static_cast<const uint64_t*>(a_ip.data());
            it = static_cast<const uint32_t*>(a_it.data());
            if (ip[cfg->num_users] > (uint64_t)a_it.shape(0)) throw std::invalid_argument("mask_items: shorter than mask_indptr[-1]");
        }
        int rc;
        {
            uint32_t* p = out.mutable_data();
            py::gil_scoped_release nogil;
            rc = heat_cf_topk(handle, 0, cfg->num_users, k, ip, it, p);
        }
        check(rc);
        return out;
    }

    std::shared_ptr<Dataset>           dataset;
    std::shared_ptr<AggregatorWeights> agg;
    std::shared_ptr<Model>             model;
    std::shared_ptr<CFConfig>          cfg;
    heat_cf_engine*                    handle = nullptr;
};

// test::test_out (modules/test/test_out.cpp:9-11): a print helper, not a test
struct test_out
{
    void print(const std::string& s) { std::cout << s << std::endl; }
};
} // namespace

PYBIND11_MODULE(cf_c, cf_module)
{
    cf_module.doc() = "HEAT cf_c module surface on MI355X (gfx950 HIP kernels behind include/heat_cf.h)";
    py::module_ modules = cf_module.def_submodule("modules", "modules");

    py::class_<CFConfig, std::shared_ptr<CFConfig>>(modules, "CFConfig")
        .def(py::init<idx_t, idx_t, idx_t, idx_t, idx_t, idx_t, idx_t, idx_t, idx_t, val_t, val_t, std::vector<idx_t>&, val_t>(),
             py::arg("emb_dim"), py::arg("num_negs"), py::arg("num_users"), py::arg("num_items"), py::arg("train_size"),
             py::arg("neg_sampler"), py::arg("tile_size"), py::arg("refresh_interval"), py::arg("num_subepoches"),
             py::arg("l2"), py::arg("clip_val"), py::arg("milestones"), py::arg("l_r"))
        .def_readwrite("emb_dim", &CFConfig::emb_dim)
        // extensions (not in the reference)
        .def_readwrite("seed", &CFConfig::seed)
        .def_readwrite("sample_index_base", &CFConfig::sample_index_base)
        .def_readwrite("use_aggregator", &CFConfig::use_aggregator)
        .def_readwrite("flags", &CFConfig::flags)
        .def_readwrite("coherence", &CFConfig::coherence)
        .def_readwrite("num_streams", &CFConfig::num_streams)
        .def_readwrite("update_mode", &CFConfig::update_mode)
        .def_readwrite("device", &CFConfig::device);

    py::module_ datasets = modules.def_submodule("datasets", "datasets");
    py::class_<Dataset, std::shared_ptr<Dataset>>(datasets, "Dataset");
    py::class_<ClickDataset, Dataset, std::shared_ptr<ClickDataset>>(datasets, "ClickDataset")
        .def(py::init<py::array, py::array, py::array>(), py::arg("click_dataset"), py::arg("historical_items"),
             py::arg("masks"))
        .def_readwrite("data_rows", &ClickDataset::data_rows)
        .def_readwrite("max_his", &ClickDataset::max_his);

    py::module_ models = modules.def_submodule("models", "models");
    py::class_<Model, std::shared_ptr<Model>>(models, "Model");
    py::class_<MatrixFactorization, Model, std::shared_ptr<MatrixFactorization>>(models, "MatrixFactorization")
        .def(py::init<std::shared_ptr<CFConfig>, py::array, py::array>(), py::arg("cf_config"), py::arg("user_weights"),
             py::arg("item_weights"));

    py::module_ aggs = modules.def_submodule("behavior_aggregators", "Behavior aggregators");
    py::class_<AggregatorWeights, std::shared_ptr<AggregatorWeights>>(aggs, "AggregatorWeights")
        .def(py::init<py::array>(), py::arg("aggregator_weights0"))
        .def_readwrite("emb_dim", &AggregatorWeights::emb_dim);

    py::module_ train = modules.def_submodule("train", "train");
    py::class_<Engine, std::shared_ptr<Engine>>(train, "Engine")
        .def(py::init<std::shared_ptr<Dataset>, std::shared_ptr<AggregatorWeights>, std::shared_ptr<Model>,
                      std::shared_ptr<CFConfig>>(),
             py::arg("dataset"), py::arg("aggregator_weights"), py::arg("model"), py::arg("cf_config"))
        .def("train_one_epoch", &Engine::train_one_epoch)
        .def("evaluate0", &Engine::evaluate0)
        // extension: fused evaluate + mask + top-k (ids only), for shapes whose dense matrix does not fit the host
        .def("topk", &Engine::topk, py::arg("k"), py::arg("mask_indptr") = py::none(), py::arg("mask_items") = py::none());

    py::module_ test = modules.def_submodule("test", "test");
    py::class_<test_out, std::shared_ptr<test_out>>(test, "test_out").def(py::init<>()).def("test", &test_out::print);
}
