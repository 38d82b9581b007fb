#!/bin/bash
# usage: scripts/kres.sh file.o  -> per-kernel registers / LDS / scratch of the gfx950 code object; disassembly in /tmp/dis/<name>.s
mkdir -p /tmp/dis && cd /tmp/dis
b=$(basename $1 .o)
cp $1 ./$b.o
rm -f $b.o.*gfx950*
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading $b.o > /dev/null 2>&1
mv $(ls $b.o.*gfx950* | head -1) $b.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $b.co | grep -E "\.name:|\.vgpr_count|\.sgpr_count|\.agpr_count|group_segment_fixed|private_segment_fixed|vgpr_spill" | awk '/\.name:/{if(l)print l; l=$2; next}{l=l" "$1$2}END{print l}' | sed 's/\.//g'
/opt/rocm/lib/llvm/bin/llvm-objdump -d $b.co > $b.s 2>/dev/null
