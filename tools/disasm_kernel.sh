#!/bin/bash
# Disassembly of one kernel out of a built object (no GPU needed):
#   tools/disasm_kernel.sh <mangled-name regex> [object = planer_amd/build/conv_winograd.o; conv_direct.o holds the implicit-GEMM kernels]  > kernel.s
R=$(cd "$(dirname "$0")/.." && pwd)
pat=$1
obj=${2:-$R/planer_amd/build/conv_winograd.o}
tmp=$(mktemp -d)
cp "$obj" $tmp/o.o
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null 2>&1)
co=$(ls $tmp/o.o.*gfx950* | head -1)
sym=$(/opt/rocm/lib/llvm/bin/llvm-readelf -s --wide "$co" | awk "{print \$8}" | grep -E "$pat" | grep -v "\.kd$" | head -1)
echo "; $sym" 
/opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols="$sym" "$co" | sed 's/\/\/.*//' 
rm -rf $tmp
