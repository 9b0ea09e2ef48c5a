# Collects every artifact under profiles/ for the current build (run on the GPU box via gpurun; outputs under gpurun_out/final).
# Then: python tools/install_profiles.py gpurun_out/final   (copies / condenses them into profiles/r03_*)
# The headline arithmetic is fp32 (bench.py default); the f16 fast mode is profiled beside it.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${OUT:-gpurun_out/final}; mkdir -p $O
LP="python3 tools/layer_profile.py"
# parts (a gpurun call is capped at 1200 s): a = benches + kernel traces, b = f32 counters, c = f16 counters + post-processing
if [ "${PART:-a}" = a ]; then
python3 bench.py > $O/bench.json 2> $O/bench.err
OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 8 > $O/bench_prof_f32.log 2>&1
OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16 -- python3 bench.py --no-cpu-baseline --no-extras --no-pipeline --precision f16 > $O/bench_prof_f16.log 2>&1
python3 bench.py --no-cpu-baseline --no-extras --no-pipeline > $O/bench_nopipe.json
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers32 -- $LP 256 $O/layers32 > $O/layers32.log 2>&1 && $LP report $O/layers32 > $O/layers32.txt
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers32_512 -- $LP 512 $O/layers32_512 > $O/layers32_512.log 2>&1 && $LP report $O/layers32_512 > $O/layers32_512.txt
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers -- $LP 256 $O/layers > $O/layers.log 2>&1 && $LP report $O/layers > $O/layers.txt
OBB_SIZE=128 OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers32_128 -- $LP 8192 $O/layers32_128 > $O/layers32_128.log 2>&1 && $LP report $O/layers32_128 > $O/layers32_128.txt
OBB_SIZE=128 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers_128 -- $LP 8192 $O/layers_128 > $O/layers_128.log 2>&1 && $LP report $O/layers_128 > $O/layers_128.txt
fi
if [ "${PART:-a}" = a32 ]; then  # the fp32 subset of part a (after a change of the fp32 kernels only)
python3 bench.py > $O/bench.json 2> $O/bench.err
OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 8 > $O/bench_prof_f32.log 2>&1
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers32 -- $LP 256 $O/layers32 > $O/layers32.log 2>&1 && $LP report $O/layers32 > $O/layers32.txt
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers32_512 -- $LP 512 $O/layers32_512 > $O/layers32_512.log 2>&1 && $LP report $O/layers32_512 > $O/layers32_512.txt
OBB_SIZE=128 OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/layers32_128 -- $LP 8192 $O/layers32_128 > $O/layers32_128.log 2>&1 && $LP report $O/layers32_128 > $O/layers32_128.txt
fi
if [ "${PART:-a}" = b ]; then
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch32 -- $LP 256 $O/pmc_fetch32 > $O/pmc_fetch32.log 2>&1
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write32 -- $LP 256 $O/pmc_write32 > $O/pmc_write32.log 2>&1
OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq32 -- $LP 256 $O/pmc_sq32 > $O/pmc_sq32.log 2>&1
fi
if [ "${PART:-a}" = c ]; then
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $LP 256 $O/pmc_fetch > $O/pmc_fetch.log 2>&1
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $LP 256 $O/pmc_write > $O/pmc_write.log 2>&1
OBB_GRAPH=0 OBB_FWD_SPLIT=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- $LP 256 $O/pmc_sq > $O/pmc_sq.log 2>&1
python3 tools/postproc_bench.py > $O/postproc.txt 2> $O/postproc.err
python3 tools/merge_scaling.py > $O/merge_scaling.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pptrace -- python3 tools/pp_trace.py > $O/pptrace.log 2>&1
fi
cd $GRAFT_REPO_ROOT
ls $O | head -50
