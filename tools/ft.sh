cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/xcd -- python3 tools/layer_profile.py 256 gpurun_out/xcd > gpurun_out/xcd.log 2>&1 && python3 tools/layer_profile.py report gpurun_out/xcd > gpurun_out/xcd.txt; grep "total\|model.5 \|model.7 \|model.8.cv1\|model.9.cv2\|model.13.cv1\|model.22.cv1\|model.4.cv2" gpurun_out/xcd.txt | cut -c1-100
rm -rf gpurun_out/pmc_fetch
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/layer_profile.py 256 gpurun_out/pmc_fetch > gpurun_out/pmc_fetch.log 2>&1
python3 bench.py --no-cpu-baseline | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['forward_ms'],3))"
