#!/bin/bash
# per-kernel register / scratch figures of one object file of the library: tools/kstats.sh f32path [name filter]
O=/root/repo/oriented-object-detection_amd/build/$1.hip.o
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $O $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.o --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.o | python3 -c "
import sys,re
txt=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', blk) or [None,'?'])[1]
    n=g('name')
    if flt and flt not in n: continue
    print('vgpr %3s sgpr %3s scratch %4s lds %6s  %s' % (g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), n[:160]))
" "$2"
rm -rf $T
