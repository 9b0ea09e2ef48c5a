# Diagnostic: builds libobbhip_stamps.so (conv.hip with -DOBB_STAMPS: s_memtime stamps around the phases of a tile) and runs one eager
# forward of 256 tiles through it; every k_conv_igemm launch prints its per-wave-tile phase cycles.  Run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=oriented-object-detection_amd; B=gpurun_out/stamps_build; mkdir -p $B
for f in $D/csrc/*.hip; do
  X=""; case "$(basename $f)" in conv.hip|front.hip) X="-DOBB_STAMPS";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno $X -c $f -o $B/$(basename $f).o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libobbhip_stamps.so $B/*.o
OBB_LIB=$PWD/$B/libobbhip_stamps.so OBB_GRAPH=0 OBB_FWD_SPLIT=1 ${STAMP_ENV} python3 tools/layer_profile.py 256 gpurun_out/stamps 2> gpurun_out/stamps.txt
grep STAMPS gpurun_out/stamps.txt | tail -${STAMP_TAIL:-45}
