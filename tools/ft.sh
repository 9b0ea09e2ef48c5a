cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q > gpurun_out/t.log 2>&1; tail -6 gpurun_out/t.log
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hm -- python3 tools/layer_profile.py 256 gpurun_out/hm > gpurun_out/hm.log 2>&1 && python3 tools/layer_profile.py report gpurun_out/hm > gpurun_out/hm.txt; grep "total\||model" gpurun_out/hm.txt | cut -c1-130
python3 bench.py --no-cpu-baseline | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['forward_ms'],3))"
