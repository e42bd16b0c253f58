# Evidence run (GPU box): Recall@20/NDCG@20 of the default engine vs the 8-thread oracle for three seeds, popularity-only
# and clustered AmazonBooks-shaped graphs.
mkdir -p gpurun_out/seeds
timeout -k 10 500 python tests/tools/recall_parity.py --shape amazonbooks --epochs 5 --update 0 --oracle-threads 8 --seeds 2022,7,99 > gpurun_out/seeds/amazonbooks_seeds.txt 2>&1
timeout -k 10 300 python tests/tools/recall_parity.py --shape amazonbooks --clusters 64 --epochs 5 --update 0,3 --oracle-threads 8,8 > gpurun_out/seeds/amazonbooks_clustered.txt 2>&1
grep -v amdgpu.ids gpurun_out/seeds/amazonbooks_seeds.txt gpurun_out/seeds/amazonbooks_clustered.txt
