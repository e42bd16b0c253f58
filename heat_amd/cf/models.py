"""Mirror of the reference's cf/models.py:8-32: N(0, 1e-2) embedding tables created on the host, handed to the
engine as numpy arrays that are trained in place."""
import numpy as np
import torch
import torch.nn as nn

from .cpp_base import CPPBase


class Model(CPPBase, nn.Module):
    def __init__(self, config):
        CPPBase.__init__(self)
        nn.Module.__init__(self)
        self.user_embedding = nn.Embedding(config.num_users, config.emb_dim, dtype=torch.float32)
        self.item_embedding = nn.Embedding(config.num_items, config.emb_dim, dtype=torch.float32)
        nn.init.normal_(self.user_embedding.weight, std=1e-2)   # models.py:15-16
        nn.init.normal_(self.item_embedding.weight, std=1e-2)
        self.user_weights = None
        self.item_weights = None


class MatrixFactorization(Model):
    def __init__(self, config):
        super().__init__(config)
        from heat_amd import cf_c
        self.c_class = cf_c.modules.models.MatrixFactorization

    def init_c_instance(self, config=None):
        self.user_weights = self.user_embedding.weight.detach().cpu().numpy()   # models.py:30-31: shared memory,
        self.item_weights = self.item_embedding.weight.detach().cpu().numpy()   # trained in place
        self.c_instance = self.c_class(cf_config=config.c_instance, user_weights=self.user_weights,
                                       item_weights=self.item_weights)
