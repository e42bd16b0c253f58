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
    def __init__(self, file_path=None, separator=' ', config=None, seed=2022, user_items=None, is_train=None, cache=True,
                 csr=None):
        """file_path: LightGCN txt; or user_items: {user_id: [items]}; or csr = (indptr, items) over user ids 0..n-1
        (synthetic graphs).

        Everything sized by the interactions is built from arrays (one CSR of the users' item lists in the reference's
        iteration order): the history matrix is one gather, the interaction list one repeat — the reference's per-user
        Python loop (datasets.py:50-78) takes minutes at 10^8 interactions.  `user_items_dic` stays a dict of lists (the
        metrics and the reference's callers read it); `user_item_ids` (a list of 2.4 M tuples at AmazonBooks size) is
        materialised only when somebody asks for it."""
        super().__init__()
        self.file_path = file_path if file_path is not None else "<memory>"
        self.max_his = config.max_his
        rnd = random.Random(seed)
        # ---- users in the reference's dict order, their items as one CSR -------------------------------------------------
        if csr is not None:
            indptr, items_all = np.asarray(csr[0], dtype=np.int64), np.asarray(csr[1], dtype=np.uint64)
            lens = np.diff(indptr)
            # a train split lists every user (an empty one gets the reference's warning and still counts in num_users);
            # other splits list the users that have items, as a file would
            users = (np.arange(lens.size) if is_train else np.flatnonzero(lens > 0)).astype(np.int64)
            starts, lens = indptr[users], lens[users]
        elif user_items is None:
            # native one-pass parser (heat_cf_parse_lightgcn) instead of the reference's per-line Python loop
            clicks, line_user, line_start = _parse_cached(file_path, separator, cache)
            items_all = np.ascontiguousarray(clicks[:, 1]).astype(np.uint64, copy=False)
            line_user = np.asarray(line_user, dtype=np.int64)
            line_start = np.asarray(line_start, dtype=np.int64)
            # a repeated user id: the dict keeps the position of its FIRST line and the items of its LAST (datasets.py:56)
            uniq, first = np.unique(line_user, return_index=True)
            last = len(line_user) - 1 - np.unique(line_user[::-1], return_index=True)[1]
            order = np.argsort(first, kind="stable")
            users, src = uniq[order], last[order]
            starts, lens = line_start[src], line_start[src + 1] - line_start[src]
        else:
            users = np.fromiter((int(u) for u in user_items), dtype=np.int64, count=len(user_items))
            lens = np.fromiter((len(v) for v in user_items.values()), dtype=np.int64, count=len(user_items))
            items_all = np.fromiter((int(i) for v in user_items.values() for i in v), dtype=np.uint64, count=int(lens.sum()))
            starts = (np.cumsum(lens) - lens).astype(np.int64)          # exclusive prefix sum (length 0 for an empty dict)
        tolist = items_all.tolist()
        self.user_items_dic = {int(u): tolist[a:a + n] for u, a, n in zip(users.tolist(), starts.tolist(), lens.tolist())}
        # datasets.py:44-45: one history row per line, indexed by user id (sized by the largest id so that a file
        # that skips users, e.g. a test split, does not index out of range as the reference would)
        num_lines = max(len(users), int(users.max()) + 1 if users.size else 0)
        self.his_items = np.zeros((num_lines, self.max_his), dtype=np.uint64)
        self.masks = np.zeros((num_lines, 1), dtype=np.uint64)
        have = lens > 0
        for u in users[~have].tolist():                          # datasets.py:67-72
            print(f"Warning {u} has 0 items !!! ")
        # :62-66 fewer than max_his items: the list padded with its last item = gather at min(column, len - 1)
        short = have & (lens < self.max_his)
        col = np.minimum(np.arange(self.max_his, dtype=np.int64)[None, :], lens[short, None] - 1)
        self.his_items[users[short]] = items_all[starts[short, None] + col]
        self.masks[users[short], 0] = lens[short].astype(np.uint64)
        # :58-61 max_his or more: a seeded random.sample per user, in dict order (as the loop drew them)
        for u, a, n in zip(users[lens >= self.max_his].tolist(), starts[lens >= self.max_his].tolist(),
                           lens[lens >= self.max_his].tolist()):
            self.his_items[u] = rnd.sample(tolist[a:a + n], self.max_his)
            self.masks[u] = self.max_his
        # :74-78 interactions in dict order
        take = np.repeat(starts - (np.cumsum(lens) - lens), lens) + np.arange(int(lens.sum()), dtype=np.int64)
        self._clicks = np.stack([np.repeat(users.astype(np.uint64), lens), items_all[take]], axis=1) if users.size else \
            np.zeros((0, 2), dtype=np.uint64)
        self._pairs = None
        # datasets.py:96-97 count distinct ids and assume they are 0..n-1; sparse id spaces (an item that never occurs in
        # train) would then index past the tables, so the table sizes cover the largest id as well
        item_ids = np.unique(self._clicks[:, 1])
        self.num_users = num_lines
        self.num_items = max(int(item_ids.size), int(item_ids[-1]) + 1 if item_ids.size else 0)
        self._min_max = (int(users.min()) if users.size else 0, int(users.max()) if users.size else 0,
                         int(item_ids[0]) if item_ids.size else 0, int(item_ids[-1]) if item_ids.size else 0)
        self.gen_dataset_info()
        train = ('train' in self.file_path) if is_train is None else is_train   # datasets.py:82
        if train:
            from heat_amd import cf_c
            self.c_class = cf_c.modules.datasets.ClickDataset
            config.num_users = self.num_users
            config.num_items = self.num_items
            config.train_size = self._clicks.shape[0]
            self.click_dataset = np.ascontiguousarray(self._clicks)
            self.init_c_instance(click_dataset=self.click_dataset, historical_items=self.his_items, masks=self.masks)
            self.c_instance.max_his = self.max_his

    @property
    def user_item_ids(self):
        """[(user, item), ...] in file order (datasets.py:74-78), built on first use."""
        if self._pairs is None:
            self._pairs = list(map(tuple, self._clicks.tolist()))
        return self._pairs

    def gen_dataset_info(self):
        lo_u, hi_u, lo_i, hi_i = self._min_max
        print(f'gen dataset info of {self.file_path} ')
        if hi_u - lo_u + 1 != self.num_users:
            print('Warning user_id is not continuous! ')
        if hi_i - lo_i + 1 != self.num_items:
            print('Warning item_id is not continuous! ')
        print(f'number of users: {self.num_users}; min_user_id: {lo_u}; max_user_id: {hi_u}')
        print(f'number of items: {self.num_items}; min_item_id: {lo_i}; max_item_id: {hi_i}')
        print(f'total samples: {self._clicks.shape[0]} ')

    def get_user_items(self):
        return self.user_items_dic

    def train_csr(self):
        """(indptr u64 [num_users+1], items u32) of this dataset's items per user id — the mask for top-k eval."""
        n = (max(self.user_items_dic) + 1) if self.user_items_dic else 0
        users = self._clicks[:, 0].astype(np.int64)
        lens = np.bincount(users, minlength=n)
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        order = np.argsort(users, kind="stable")                 # dict order within a user = file order
        return indptr, self._clicks[order, 1].astype(np.uint32)
