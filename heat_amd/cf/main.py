"""Counterpart of the reference's cf/main.py:19-124 (same yaml keys, same flow, same epoch log line), without
mpi4py: `python -m heat_amd.cf.main --config <yaml>`.  With `--synthetic <shape>` the LightGCN files named by the
yaml are replaced by a seeded synthetic graph of that shape (the real datasets are not available offline)."""
import argparse
import os
import time

import numpy as np
import torch

from . import metrics, synthetic, utils
from .behavior_aggregators import AggregatorWeights
from .cf_config import CFConfig
from .datasets import ClickDataset
from .models import MatrixFactorization
from .train import Engine


def _graph_to_datasets(graph, cf_config, seed):
    train = ClickDataset(config=cf_config, seed=seed, csr=(graph.train_indptr, graph.train_items), is_train=True)
    test = ClickDataset(config=cf_config, seed=seed, csr=(graph.test_indptr, graph.test_items), is_train=False)
    return train, test


def _main_distributed(args, model_config, cf_config, train_data, test_data, seed):
    """torchrun path (one process per GPU): users sharded by contiguous range, item table replicated and synchronised by
    heat_amd.cf.distributed — what the fork does with mpi4py + per-row MPI collectives (cf/main.py:47-97, engine.cpp:262-375)."""
    import torch.distributed as dist
    from .distributed import ShardedTrainer
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HEAT_CF_DIST_BACKEND=gloo: several ranks share one GPU (tests; RCCL refuses two ranks on a device)
    backend = os.environ.get("HEAT_CF_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    agg = {}
    if getattr(cf_config, "use_aggregator", False):             # ACCL: history rows follow their users, W0 is replicated
        cf_config.init_c_instance()
        w0 = AggregatorWeights(cf_config).aggregator_weights0   # drawn before the tables, as in the one-GPU flow below
        agg = dict(his=train_data.his_items, masks=train_data.masks, w0=w0)
    model = MatrixFactorization(cf_config)                      # identical initial tables on every rank (same torch seed)
    user_w = model.user_embedding.weight.detach().cpu().numpy()
    item_w = model.item_embedding.weight.detach().cpu().numpy()
    trainer = ShardedTrainer(train_data.click_dataset, user_w, item_w, num_negs=cf_config.num_negs, seed=seed,
                             clip_val=cf_config.clip_val, l_r=cf_config.l_r, milestones=tuple(cf_config.milestones),
                             refresh_interval=cf_config.refresh_interval, neg_sampler=cf_config.neg_sampler,
                             tile_size=cf_config.tile_size,
                             sync_interactions=int(model_config.get('sync_interactions', 0)),   # 0: streams x refresh_interval, capped at one epoch
                             windows_per_epoch=int(model_config.get('sync_windows', 0)),        # > 0: that many equal windows per epoch instead
                             overlap=bool(model_config.get('sync_overlap', False)),              # all-reduce of a window hidden behind the next one
                             num_streams=int(cf_config.num_streams), update_mode=int(cf_config.update_mode),
                             **agg)
    lo, hi = trainer.lo, trainer.hi
    indptr, items = train_data.train_csr()
    local_indptr = (indptr[lo:hi + 1] - indptr[lo]).astype(np.uint64)
    local_items = items[int(indptr[lo]):int(indptr[hi])]
    results = {}
    for epoch in range(model_config['epochs']):
        start_time = time.time()
        epoch_loss = trainer.train_one_epoch(want_loss=True)
        if rank == 0:
            print(f'epoch: {epoch}; loss: {epoch_loss}; epoch_time: {time.time() - start_time}')
        if epoch > 0 and epoch % model_config['eval_interval'] == 0:
            top = trainer.engine.topk(20, mask_indptr=local_indptr, mask_items=local_items)     # this rank's users
            local_test = {u - lo: v for u, v in test_data.user_items_dic.items() if lo <= u < hi}
            sums = torch.zeros(2, dtype=torch.float64, device=torch.device("cuda", local_rank))
            if local_test:
                r = metrics.evaluate_topk(type("T", (), {"user_items_dic": local_test})(), top, ['Recall(k=20)'], quiet=True,
                                          by_user_id=True)
                sums[0], sums[1] = r['Recall(k=20)'] * len(local_test), len(local_test)
            dist.all_reduce(sums)
            results = {'Recall(k=20)': float(sums[0] / sums[1].clamp(min=1))}
            if rank == 0:
                print('[Metrics] Recall(k=20): {:.6f}'.format(results['Recall(k=20)']))
    dist.barrier()
    dist.destroy_process_group()
    return results


def main(argv=None):
    print('this is main ...')
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', type=str, default=os.path.join(os.path.dirname(__file__), 'benchmarks', 'AmazonBooks',
                                                                   'MF_CCL', 'configs', 'config0.yaml'))
    parser.add_argument('--synthetic', type=str, default=None, help='amazonbooks | gowalla | yelp18: seeded synthetic graph')
    parser.add_argument('--scale', type=float, default=1.0)
    parser.add_argument('--distributed', action='store_true', help='use the sharded multi-GPU trainer even with one rank')
    parser.add_argument('--dense-eval', action='store_true',
                        help="evaluate as the reference does (main.py:117-121): evaluate0() returns the dense num_users x "
                             "num_items score matrix to the host, numpy masks / partitions / sorts it.  Default: the fused GPU "
                             "top-k (same ids, tests/test_gpu_parity.py::test_evaluate0_and_topk) + the same scoring")
    parser.add_argument('--gpu-topk', action='store_true', help='accepted for older command lines: the fused top-k is the default')
    args = parser.parse_args(argv)
    config_dic = utils.load_config(args.config)
    dataset_config = config_dic['dataset_config']
    model_config = config_dic['model_config']
    print(model_config)
    seed = int(model_config.get('seed', 2022))      # the reference never reads `seed`; we honour it
    torch.manual_seed(seed)
    cf_config = CFConfig(emb_dim=model_config['embedding_dim'], num_negs=model_config['num_negs'],
                         max_his=model_config['max_his'], neg_sampler=model_config['neg_sampler'],
                         tile_size=model_config['tile_size'], refresh_interval=model_config['refresh_interval'],
                         l2=model_config['embedding_regularizer'], clip_val=model_config['clip_val'],
                         milestones=model_config['milestones'], l_r=model_config['learning_rate'], seed=seed,
                         use_aggregator=bool(model_config.get('use_aggregator', False)),
                         # optional extension keys (absent from the reference's yaml): 0 = the engine's own choice
                         num_streams=int(model_config.get('num_streams', 0)),
                         update_mode=int(model_config.get('update_mode', 0)),
                         # `sampling_call: true` = the loop's alternative sampler call (train/engine.cpp:333): the only one in
                         # which neg_sampler 1 draws from its tile (random_tile_negative_sampler.cpp:23-45 vs :47-57)
                         # `tile_in_lds: true` (with sampling_call): hold the tile's weight deltas in LDS where they fit (opt-in)
                         flags=(4 if model_config.get('sampling_call', False) else 0) |
                               (0x20 if model_config.get('tile_in_lds', False) else 0))
    print('--- Start loading data ---')
    if args.synthetic:
        graph, _, _ = synthetic.make_named(args.synthetic, seed=seed, scale=args.scale)
        train_data, test_data = _graph_to_datasets(graph, cf_config, seed)
    else:
        train_file = os.path.join(dataset_config['data_dir'], dataset_config['train_data'])
        test_file = os.path.join(dataset_config['data_dir'], dataset_config['test_data'])
        train_data = ClickDataset(train_file, separator=dataset_config['separator'], config=cf_config, seed=seed)
        test_data = ClickDataset(test_file, separator=dataset_config['separator'], config=cf_config, seed=seed)
    print('--- Finished loading data ---')
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.distributed:
        return _main_distributed(args, model_config, cf_config, train_data, test_data, seed)
    cf_config.init_c_instance()
    aggregator_weights = AggregatorWeights(cf_config)
    model = MatrixFactorization(cf_config)
    model.init_c_instance(cf_config)
    engine = Engine(train_data, aggregator_weights, model, cf_config)
    eval_interval = model_config['eval_interval']
    results = {}
    for epoch in range(model_config['epochs']):
        start_time = time.time()
        epoch_loss = engine.train_one_epoch()
        epoch_time = time.time() - start_time
        print(f'epoch: {epoch}; loss: {epoch_loss}; epoch_time: {epoch_time}')      # main.py:113
        if epoch > 0 and epoch % eval_interval == 0:                                # main.py:115
            print('--- Start evaluation ---')
            model.eval()
            with torch.no_grad():
                eva_metrics = ['Recall(k=20)']                                      # main.py:120
                if not args.dense_eval:
                    indptr, items = train_data.train_csr()
                    top = engine.topk(20, indptr, items)
                    results = metrics.evaluate_topk(test_data, top, eva_metrics)
                else:
                    sim_matrix = engine.evaluate0()
                    print(f'sim_matrix shape: {np.shape(sim_matrix)} !!! ')
                    results = metrics.evaluate_metrics(train_data, test_data, sim_matrix, eva_metrics)
    return results


if __name__ == "__main__":
    main()
