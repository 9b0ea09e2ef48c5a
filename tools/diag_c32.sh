# Diagnostic: profiles the fp32 layers under the timing-only ablations of k_conv_f32 (OBB_C32_DBG: 1 = no global fetch, 2 = no epilogue,
# 3 = both, 4 = every activation LDS read twice, 8 = no MFMAs; DIAG_SET="0 4 8" selects the runs) through build_diag/libobbhip_diag.so = the library with f32path.hip compiled -DOBB_DIAG:
#   mkdir -p build_diag && cd build_diag && for f in ../oriented-object-detection_amd/csrc/*.hip; do X=""; [ $(basename $f) = f32path.hip ] && X=-DOBB_DIAG;
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno $X -c $f -o $(basename $f).o; done; hipcc --offload-arch=gfx950 -shared -fPIC -o libobbhip_diag.so *.o
# Run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/diag; mkdir -p $O
for V in ${DIAG_SET:-0 1 2 3}; do
OBB_C32_DBG=$V OBB_LIB=$PWD/build_diag/libobbhip_diag.so OBB_PREC=f32 OBB_GRAPH=0 OBB_FWD_SPLIT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/v$V -- python3 tools/layer_profile.py 512 $O/v$V > $O/v$V.log 2>&1 && python3 tools/layer_profile.py report $O/v$V > $O/v$V.txt
tail -1 $O/v$V.txt
done
