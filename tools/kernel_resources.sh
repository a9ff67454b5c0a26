#!/bin/bash
# Register / LDS / spill figures of the kernels in a built object (no GPU needed):
#   tools/kernel_resources.sh [pattern] [object = planer_amd/build/conv_winograd.o; conv_direct.o holds the implicit-GEMM kernels]
# Extracts the gfx950 code object from the offload bundle and reads the kernel metadata notes.
R=$(cd "$(dirname "$0")/.." && pwd)
pat=${1:-.}
obj=${2:-$R/planer_amd/build/conv_winograd.o}
tmp=$(mktemp -d)
cp "$obj" $tmp/o.o
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null 2>&1)
co=$(ls $tmp/o.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in re.split(r'\n  - ', txt):
    m = re.search(r'\.name:\s+(\S+)', blk)
    if not m or '.vgpr_count' not in blk: continue
    name = m.group(1)
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [0, '?'])[1]
    row = '%-90s vgpr %s agpr %s sgpr %s lds %s spill v%s s%s scratch %s' % (name[:90], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'))
    if re.search(r'$pat', name): print(row)
"
rm -rf $tmp
