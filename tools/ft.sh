cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_metrics.py tests/test_gpu_geometry.py -x -q > gpurun_out/t.log 2>&1; tail -8 gpurun_out/t.log
