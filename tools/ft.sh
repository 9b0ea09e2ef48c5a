cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OBB_GRAPH=0 OBB_FWD_SPLIT=1
for t in 1 8 32; do
  OBB_FUSED_TPW=$t OBB_FUSED_WPE=2 OBB_FUSED_DBG=60 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fts_$t -- python3 tools/fused_timing.py run > gpurun_out/fts_$t.log 2>&1 && python3 tools/fused_timing.py report gpurun_out/fts_$t
done
for t in 32; do
  OBB_FUSED_TPW=$t OBB_FUSED_WPE=2 OBB_FUSED_DBG=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ftf_$t -- python3 tools/fused_timing.py run > gpurun_out/ftf_$t.log 2>&1 && python3 tools/fused_timing.py report gpurun_out/ftf_$t
done
