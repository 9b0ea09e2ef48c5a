cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_metrics.py -x -q > gpurun_out/t.log 2>&1; tail -6 gpurun_out/t.log
