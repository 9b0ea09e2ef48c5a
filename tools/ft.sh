cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q > gpurun_out/t.log 2>&1; tail -4 gpurun_out/t.log
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dw3 -- python3 tools/layer_profile.py 256 gpurun_out/dw3 > gpurun_out/dw3.log 2>&1 && python3 tools/layer_profile.py report gpurun_out/dw3 > gpurun_out/dw3.txt; grep "total\|dwconv" gpurun_out/dw3.txt | cut -c1-100
