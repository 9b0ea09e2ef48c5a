# Collects every artifact under profiles/ for the current build (run on the GPU box via gpurun; outputs under gpurun_out/final).
# Then: python tools/install_profiles.py gpurun_out/final   (copies / condenses them into profiles/r02_*)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
# two halves (a gpurun call is capped at 1200 s): PART=a benches + kernel traces, PART=b counters + post-processing + stamps
if [ "${PART:-a}" = a ]; then
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_prof_b.log 2>&1
OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bs -- python3 bench.py --no-cpu-baseline --no-extras --no-pipeline > $O/bench_prof_bs.log 2>&1
OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 bench.py --no-cpu-baseline --no-extras --no-pipeline --precision f32 --steps 6 > $O/bench_prof_f32.log 2>&1
OBB_FWD_SPLIT=1 python3 bench.py --no-cpu-baseline --no-extras --no-pipeline > $O/bench_seq_plain.json
python3 bench.py --no-cpu-baseline --no-extras --no-pipeline > $O/bench_nopipe.json
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers -- python3 tools/layer_profile.py 256 $O/layers > $O/layers.log 2>&1 && python3 tools/layer_profile.py report $O/layers > $O/layers.txt
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers32 -- python3 tools/layer_profile.py 256 $O/layers32 > $O/layers32.log 2>&1 && python3 tools/layer_profile.py report $O/layers32 > $O/layers32.txt
fi
if [ "${PART:-b}" = b ]; then
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/layer_profile.py 256 $O/pmc_fetch > $O/pmc_fetch.log 2>&1
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/layer_profile.py 256 $O/pmc_write > $O/pmc_write.log 2>&1
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 tools/layer_profile.py 256 $O/pmc_sq > $O/pmc_sq.log 2>&1
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq32 -- python3 tools/layer_profile.py 256 $O/pmc_sq32 > $O/pmc_sq32.log 2>&1
python3 tools/postproc_bench.py > $O/postproc.txt 2> $O/postproc.err
python3 tools/merge_scaling.py > $O/merge_scaling.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pptrace -- python3 tools/pp_trace.py > $O/pptrace.log 2>&1
STAMP_TAIL=70 bash tools/stamp_conv.sh > $O/stamps.txt 2>&1
fi
cd $GRAFT_REPO_ROOT
grep -h metric $O/bench.json $O/bench_prof_b.log $O/bench_prof_bs.log $O/bench_seq_plain.json $O/bench_nopipe.json $O/bench_prof_f32.log | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['dtype'], round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['forward_ms'],3), d['config']['step_pipelining'])"
