#!/bin/bash
# Registers / scratch / spills of every kernel in libavsim.so (gfx950 code objects inside the fat binary); runs anywhere.
# usage: tools/kernel_resources.sh [path/to/libavsim.so]
so=$(realpath ${1:-$(dirname $0)/../av_aloha_amd/libavsim.so})
t=$(mktemp -d); cp $so $t/lib.so; cd $t
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1
for f in lib.so.*gfx950; do
  echo "== code object $f ($(stat -c %s $f) bytes)"
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|\.private_segment_fixed_size|\.vgpr_spill_count|agpr_count" | paste - - - - - - | sed 's/  */ /g; s/\t/ /g' | cut -c1-260
done
rm -rf $t
