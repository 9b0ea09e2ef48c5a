cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
OBB_DIST_BACKEND=gloo OBB_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --batch 512 > gpurun_out/bench_2rank.log 2>&1; grep metric gpurun_out/bench_2rank.log | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('2 ranks on one GPU (gloo):', round(d['value']), d['n_gpus'], round(d['ms_per_step'],3))"
bash tools/profile_all.sh
