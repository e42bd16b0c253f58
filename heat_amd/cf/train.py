"""Mirror of the reference's cf/train.py:4-24."""
from .cpp_base import CPPBase


class Engine(CPPBase):
    def __init__(self, dataset=None, aggregator_weights=None, model=None, cf_config=None):
        super().__init__()
        from heat_amd import cf_c
        self.c_class = cf_c.modules.train.Engine
        self.init_c_instance(dataset=dataset.c_instance,
                             aggregator_weights=aggregator_weights.c_instance if aggregator_weights else None,
                             model=model.c_instance, cf_config=cf_config.c_instance)

    def train_one_epoch(self):
        return self.c_instance.train_one_epoch()

    def evaluate0(self):
        return self.c_instance.evaluate0()

    def topk(self, k, mask_indptr=None, mask_items=None):
        """Extension: ids of the k best items per user, train items masked, computed on the GPU."""
        return self.c_instance.topk(k, mask_indptr, mask_items)
