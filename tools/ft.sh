cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bn2 -- python3 tools/layer_profile.py 256 gpurun_out/bn2 > gpurun_out/bn2.log 2>&1 && python3 tools/layer_profile.py report gpurun_out/bn2 > gpurun_out/bn2.txt; grep "total\|bneck" gpurun_out/bn2.txt | cut -c1-110
