"""Mirror of the reference's cf/datasets.py:14-107 (ClickDataset) without mpi4py: LightGCN text format
(`user item item ...` per line), history matrix + masks, interaction list in file order.

Differences from the reference, all deliberate: `random.sample` of the history is seeded (`seed` argument; the
reference is unseeded, datasets.py:48), parsing is vectorised for large files, and sharding a dataset by user
range lives in heat_amd.cf.distributed (the fork's SubClickDataset + pickling over MPI is not reproduced)."""
import random

import numpy as np

from .cpp_base import CPPBase


def _parse_cached(file_path, separator, cache):
    """Native parse of a LightGCN text file with a binary side-car cache (`<file>.heatcf.npz`, keyed by the source's size
    and mtime): the second load of a large file is one np.load instead of a text parse."""
    import os
    from heat_amd import abi
    side = file_path + ".heatcf.npz"
    st = os.stat(file_path)
    key = np.array([st.st_size, st.st_mtime_ns, ord(separator[:1] or " ")], dtype=np.int64)
    if cache and os.path.exists(side):
        try:
            with np.load(side, allow_pickle=False) as z:      # our own file; nothing executable in it
                if np.array_equal(z["key"], key):
                    return z["clicks"], z["line_user"], z["line_start"]
        except (OSError, KeyError, ValueError):
            pass
    clicks, line_user, line_start = abi.parse_lightgcn(file_path, separator)
    if cache:
        try:
            np.savez(side, key=key, clicks=clicks, line_user=line_user, line_start=line_start)
        except OSError:
            pass                                              # read-only dataset directory: just skip the cache
    return clicks, line_user, line_start


class Dataset(CPPBase):
    def __init__(self):
        super().__init__()


class ClickDataset(Dataset):
    def __init__(self, file_path=None, separator=' ', config=None, seed=2022, user_items=None, is_train=None, cache=True):
        """file_path: LightGCN txt; or user_items: {user_id: [items]} (synthetic graphs)."""
        super().__init__()
        self.file_path = file_path if file_path is not None else "<memory>"
        self.user_items_dic = {}
        self.max_his = config.max_his
        rnd = random.Random(seed)
        if user_items is None:
            # native one-pass parser (heat_cf_parse_lightgcn) instead of the reference's per-line Python loop
            clicks, line_user, line_start = _parse_cached(file_path, separator, cache)
            items_all = clicks[:, 1]
            for k, u in enumerate(line_user.tolist()):
                self.user_items_dic[u] = items_all[line_start[k]:line_start[k + 1]].tolist()   # a repeated user id: last line wins (datasets.py:56)
        else:
            self.user_items_dic = {int(u): [int(i) for i in items] for u, items in user_items.items()}
        # datasets.py:44-45: one history row per line, indexed by user id (sized by the largest id so that a file
        # that skips users, e.g. a test split, does not index out of range as the reference would)
        num_lines = max(len(self.user_items_dic), (max(self.user_items_dic) + 1) if self.user_items_dic else 0)
        self.his_items = np.zeros((num_lines, self.max_his), dtype=np.uint64)
        self.masks = np.zeros((num_lines, 1), dtype=np.uint64)
        item_ids = set()
        pairs = []
        for user_id, items in self.user_items_dic.items():
            if len(items) >= self.max_his:                       # datasets.py:58-61
                self.his_items[user_id] = rnd.sample(items, self.max_his)
                self.masks[user_id] = self.max_his
            elif len(items) > 0:                                 # :62-66 pad with the last item
                self.his_items[user_id] = items + [items[-1]] * (self.max_his - len(items))
                self.masks[user_id] = len(items)
            else:                                                # :67-72
                print(f"Warning {user_id} has 0 items !!! ")
            item_ids.update(items)
            pairs.extend((user_id, it) for it in items)          # :74-78 interactions in file order
        self.user_item_ids = pairs
        # datasets.py:96-97 count distinct ids and assume they are 0..n-1; sparse id spaces (an item that never occurs in
        # train) would then index past the tables, so the table sizes cover the largest id as well
        self.num_users = max(len(self.user_items_dic), (max(self.user_items_dic) + 1) if self.user_items_dic else 0)
        self.num_items = max(len(item_ids), (max(item_ids) + 1) if item_ids else 0)
        self._min_max = (min(self.user_items_dic) if self.user_items_dic else 0,
                         max(self.user_items_dic) if self.user_items_dic else 0,
                         min(item_ids) if item_ids else 0, max(item_ids) if item_ids else 0)
        self.gen_dataset_info()
        train = ('train' in self.file_path) if is_train is None else is_train   # datasets.py:82
        if train:
            from heat_amd import cf_c
            self.c_class = cf_c.modules.datasets.ClickDataset
            config.num_users = self.num_users
            config.num_items = self.num_items
            config.train_size = len(self.user_item_ids)
            self.click_dataset = np.array(self.user_item_ids, dtype=np.uint64).reshape(-1, 2)
            self.init_c_instance(click_dataset=self.click_dataset, historical_items=self.his_items, masks=self.masks)
            self.c_instance.max_his = self.max_his

    def gen_dataset_info(self):
        lo_u, hi_u, lo_i, hi_i = self._min_max
        print(f'gen dataset info of {self.file_path} ')
        if hi_u - lo_u + 1 != self.num_users:
            print('Warning user_id is not continuous! ')
        if hi_i - lo_i + 1 != self.num_items:
            print('Warning item_id is not continuous! ')
        print(f'number of users: {self.num_users}; min_user_id: {lo_u}; max_user_id: {hi_u}')
        print(f'number of items: {self.num_items}; min_item_id: {lo_i}; max_item_id: {hi_i}')
        print(f'total samples: {len(self.user_item_ids)} ')

    def get_user_items(self):
        return self.user_items_dic

    def train_csr(self):
        """(indptr u64 [num_users+1], items u32) of this dataset's items per user id — the mask for top-k eval."""
        n = (max(self.user_items_dic) + 1) if self.user_items_dic else 0
        lens = np.zeros(n, dtype=np.int64)
        for u, items in self.user_items_dic.items():
            lens[u] = len(items)
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        items = np.empty(int(indptr[-1]), dtype=np.uint32)
        for u, its in self.user_items_dic.items():
            items[int(indptr[u]):int(indptr[u + 1])] = its
        return indptr, items
