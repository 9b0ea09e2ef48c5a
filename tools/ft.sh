cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
python3 bench.py --no-cpu-baseline | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['forward_ms'],3))"
