cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
python3 bench.py > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; python3 -c "
import json
d=json.loads(open('gpurun_out/bench_r01b.json').read()); print(round(d['value']), round(d['ms_per_step'],3), d['roofline'], d['cpu_baseline']['value'])"
