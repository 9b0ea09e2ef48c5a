cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q > gpurun_out/t.log 2>&1; tail -4 gpurun_out/t.log
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sp -- python3 tools/layer_profile.py 256 gpurun_out/sp > gpurun_out/sp.log 2>&1 && python3 tools/layer_profile.py report gpurun_out/sp > gpurun_out/sp.txt; grep "total\|pool" gpurun_out/sp.txt | cut -c1-120
python3 bench.py --no-cpu-baseline | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['forward_ms'],3))"
