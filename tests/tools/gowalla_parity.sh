# Evidence run (GPU box): BASELINE.json configs[0] shape (Gowalla d=64 negs=16), default engine vs two 8-thread oracle runs.
mkdir -p gpurun_out/gowalla
timeout -k 10 300 python tests/tools/recall_parity.py --shape gowalla --epochs 5 --update 0 --oracle-threads 8,8 --seeds 2022,7 > gpurun_out/gowalla/gowalla.txt 2>&1
timeout -k 10 300 python tests/tools/recall_parity.py --shape gowalla --clusters 64 --epochs 5 --update 0 --oracle-threads 8,8 > gpurun_out/gowalla/gowalla_clustered.txt 2>&1
grep -hv amdgpu.ids gpurun_out/gowalla/gowalla.txt gpurun_out/gowalla/gowalla_clustered.txt
