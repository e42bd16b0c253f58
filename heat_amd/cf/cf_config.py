"""Mirror of the reference's cf/cf_config.py:5-40 (same constructor keywords, same `init_c_instance`), plus the
extension attributes the MI355X engine understands (seed, use_aggregator, coherence, flags, num_streams)."""
from .cpp_base import CPPBase


class CFConfig(CPPBase):
    def __init__(self, emb_dim=64, num_negs=4, max_his=8, num_users=128, num_items=128, train_size=128,
                 neg_sampler=0, tile_size=1024, num_subepoches=2, refresh_interval=2048, l2=1.e-3, clip_val=0.1,
                 milestones=(), l_r=1.e-3, seed=2022, use_aggregator=False, coherence=0, flags=0, num_streams=0, update_mode=0):
        super().__init__()
        from heat_amd import cf_c  # the built pybind11 module; raises ImportError when it has not been built
        self.c_class = cf_c.modules.CFConfig
        self.emb_dim = emb_dim
        self.num_negs = num_negs
        self.num_users = num_users
        self.num_items = num_items
        self.train_size = train_size
        self.neg_sampler = neg_sampler
        self.tile_size = tile_size
        self.refresh_interval = refresh_interval
        self.num_subepoches = num_subepoches
        self.l2 = l2
        self.clip_val = clip_val
        self.milestones = list(milestones)
        self.l_r = l_r
        # dataset (cf_config.py:25-27)
        self.en_his = True
        self.max_his = max_his
        # extensions
        self.seed = seed
        self.use_aggregator = use_aggregator
        self.coherence = coherence
        self.flags = flags
        self.num_streams = num_streams
        self.update_mode = update_mode

    def init_c_instance(self):
        self.c_instance = self.c_class(emb_dim=self.emb_dim, num_negs=self.num_negs, num_users=self.num_users,
                                       num_items=self.num_items, train_size=self.train_size,
                                       neg_sampler=self.neg_sampler, tile_size=self.tile_size,
                                       refresh_interval=self.refresh_interval, num_subepoches=self.num_subepoches,
                                       l2=self.l2, clip_val=self.clip_val, milestones=self.milestones, l_r=self.l_r)
        self.c_instance.seed = int(self.seed)
        self.c_instance.use_aggregator = int(bool(self.use_aggregator))
        self.c_instance.coherence = int(self.coherence)
        self.c_instance.flags = int(self.flags)
        self.c_instance.num_streams = int(self.num_streams)
        self.c_instance.update_mode = int(self.update_mode)
