cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q > gpurun_out/t.log 2>&1; tail -4 gpurun_out/t.log
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pl -- python3 tools/layer_profile.py 256 gpurun_out/pl > gpurun_out/pl.log 2>&1 && python3 tools/layer_profile.py report gpurun_out/pl > gpurun_out/pl.txt; grep "total\|pool" gpurun_out/pl.txt | cut -c1-110
