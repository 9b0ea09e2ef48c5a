cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
