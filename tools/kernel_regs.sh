#!/bin/bash
# register / spill counts of the kernels in a built library whose names match $2 (default: all): tools/kernel_regs.sh lib.so delta
LIB=$(readlink -f ${1:-/root/repo/wiggletools_amd/csrc/libwiggletools_amd.so})
T=$(mktemp -d)
cd $T
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading $LIB > /dev/null 2>&1
D=$(dirname $LIB)
for f in $D/$(basename $LIB).*.hipv4*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f 2>/dev/null | grep -E "\.name:|\.vgpr_count|\.sgpr_count|vgpr_spill|\.private_segment_fixed" | paste - - - - - | grep -i "${2:-.}" | sed 's/[ \t][ \t]*/ /g'
done
rm -f $D/$(basename $LIB).*.hipv4* $D/$(basename $LIB).*.host*
rm -rf $T
