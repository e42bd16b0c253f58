"""Host-side counterparts of the reference's thin Python wrappers, in one module.

The reference spreads six ~20-line files over `cf_cpu/cf/` (`cpp_base.py:2-11`, `cf_config.py:5-40`, `models.py:8-32`,
`behavior_aggregators.py:8-17`, `train.py:4-24`, `utils.py:5-9`); each wrapper owns a `c_instance` built from a class of
the `cf_c` pybind11 module.  The same public names, constructor keywords and attributes are kept here (so a driver
written for the reference keeps working), the implementation is organised around one small base class that resolves its
`cf_c` class lazily from a dotted path.  `heat_amd.cf.{cpp_base,cf_config,models,behavior_aggregators,train,utils}` are
import-compatible aliases of this module.
"""
import yaml
import torch
from torch import nn


def _cf_c():
    from heat_amd import cf_c          # the built pybind11 module; ImportError if csrc/ has not been compiled
    return cf_c


class CPPBase:
    """Owner of a `c_instance`; `c_class` is the pybind11 class it is built from."""
    _c_path = None                      # e.g. "modules.train.Engine"

    def __init__(self):
        self.c_instance = None
        self.c_class = None
        if self._c_path:
            obj = _cf_c()
            for part in self._c_path.split("."):
                obj = getattr(obj, part)
            self.c_class = obj

    def init_c_instance(self, **init_args):
        if self.c_class is None:
            raise RuntimeError("c_class is None. ")
        self.c_instance = self.c_class(**init_args)


class CFConfig(CPPBase):
    """Training configuration; `init_c_instance()` materialises `cf_c.modules.CFConfig` once the dataset has filled in
    num_users / num_items / train_size (cf/main.py:40-44,75)."""
    _c_path = "modules.CFConfig"
    _REFERENCE_FIELDS = ("emb_dim", "num_negs", "num_users", "num_items", "train_size", "neg_sampler", "tile_size",
                         "refresh_interval", "num_subepoches", "l2", "clip_val", "milestones", "l_r")
    _EXTENSIONS = ("seed", "use_aggregator", "coherence", "flags", "num_streams", "update_mode")

    def __init__(self, emb_dim=64, num_negs=4, max_his=8, num_users=128, num_items=128, train_size=128, neg_sampler=0,
                 tile_size=1024, num_subepoches=2, refresh_interval=2048, l2=1.e-3, clip_val=0.1, milestones=(),
                 l_r=1.e-3, seed=2022, use_aggregator=False, coherence=0, flags=0, num_streams=0, update_mode=0):
        super().__init__()
        given = dict(locals())
        for name in self._REFERENCE_FIELDS + self._EXTENSIONS:
            setattr(self, name, given[name])
        self.milestones = list(milestones)
        self.en_his = True              # cf_config.py:26, never plumbed to C++
        self.max_his = max_his

    def init_c_instance(self):
        self.c_instance = self.c_class(**{name: getattr(self, name) for name in self._REFERENCE_FIELDS})
        for name in self._EXTENSIONS:
            setattr(self.c_instance, name, int(getattr(self, name)))


class Model(CPPBase, nn.Module):
    """N(0, 1e-2) user / item tables created with torch on the host (models.py:13-16)."""

    def __init__(self, config):
        CPPBase.__init__(self)
        nn.Module.__init__(self)
        self.user_embedding = nn.Embedding(config.num_users, config.emb_dim, dtype=torch.float32)
        self.item_embedding = nn.Embedding(config.num_items, config.emb_dim, dtype=torch.float32)
        for table in (self.user_embedding, self.item_embedding):
            nn.init.normal_(table.weight, std=1e-2)
        self.user_weights = self.item_weights = None


class MatrixFactorization(Model):
    _c_path = "modules.models.MatrixFactorization"

    def init_c_instance(self, config=None):
        # numpy views of the torch parameters: the engine trains them in place (models.py:30-32).  Where a GPU is present
        # the parameters are moved to page-locked host memory first: the engine's per-epoch write-back into these very
        # buffers (the reference's in-place contract, init_modules.cpp:79-81) is then one DMA at PCIe rate instead of a
        # staged pageable copy.
        if torch.cuda.is_available():
            for table in (self.user_embedding, self.item_embedding):
                table.weight.data = table.weight.data.pin_memory()
        self.user_weights = self.user_embedding.weight.detach().cpu().numpy()
        self.item_weights = self.item_embedding.weight.detach().cpu().numpy()
        CPPBase.init_c_instance(self, cf_config=config.c_instance, user_weights=self.user_weights,
                                item_weights=self.item_weights)


class AggregatorWeights(CPPBase, nn.Module):
    """The d x d aggregation matrix W0, N(0, 1e-2) (behavior_aggregators.py:14-17)."""
    _c_path = "modules.behavior_aggregators.AggregatorWeights"

    def __init__(self, config):
        CPPBase.__init__(self)
        nn.Module.__init__(self)
        self.f_c0 = nn.Linear(config.emb_dim, config.emb_dim, bias=False, dtype=torch.float32)
        nn.init.normal_(self.f_c0.weight, std=1e-2)
        self.aggregator_weights0 = self.f_c0.weight.detach().cpu().numpy()
        self.init_c_instance(aggregator_weights0=self.aggregator_weights0)


class Engine(CPPBase):
    """Epoch driver (train.py:4-24) plus the fused GPU top-k extension."""
    _c_path = "modules.train.Engine"

    def __init__(self, dataset=None, aggregator_weights=None, model=None, cf_config=None):
        super().__init__()
        self.init_c_instance(dataset=dataset.c_instance,
                             aggregator_weights=None if aggregator_weights is None else aggregator_weights.c_instance,
                             model=model.c_instance, cf_config=cf_config.c_instance)

    def train_one_epoch(self):
        return self.c_instance.train_one_epoch()

    def evaluate0(self):
        return self.c_instance.evaluate0()

    def topk(self, k, mask_indptr=None, mask_items=None):
        """ids of the k best items per user, train items masked, computed on the GPU (not in the reference)."""
        return self.c_instance.topk(k, mask_indptr, mask_items)


def load_config(config_path):
    """yaml -> dict (utils.py:5-9)."""
    with open(config_path) as handle:
        return yaml.safe_load(handle)
