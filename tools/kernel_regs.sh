#!/bin/bash
# Developer tool (CPU): VGPR / AGPR / SGPR / spill / LDS figures of the kernels in libdsdenoise.so whose name matches $1 (default: all hot kernels)
set -e
D=$(mktemp -d)
SO=${2:-$(dirname $0)/../diffsinger_amd/libdsdenoise.so}
cp $SO $D/lib.so && /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$D/fat.bin $D/lib.so 2>/dev/null   # on a COPY: objcopy rewrites its input (and its mtime: the build then looks up to date)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$D/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$D/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $D/dev.co | python3 -c "
import sys,re
pat=sys.argv[1]
txt=sys.stdin.read()
for b in txt.split('- .agpr_count')[1:]:
    name=re.search(r'\.name:\s+(\S+)',b).group(1)
    if re.search(pat,name):
        g=lambda k: re.search(k+r':\s+(\d+)',b).group(1)
        print(name[:70].ljust(70), 'agpr',re.match(r':\s+(\d+)',b).group(1),'vgpr',g(r'\.vgpr_count'),'sgpr',g(r'\.sgpr_count'),'spill',g(r'\.vgpr_spill_count'),'scratch',g(r'\.private_segment_fixed_size'))
" "${1:-k_loop|k_layer|k_head|k_lat}"
rm -rf $D
