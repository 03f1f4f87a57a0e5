#!/bin/bash
# kernel resource usage (VGPRs, SGPR spills, scratch) of a built device object: tools/kernel_resources.sh <obj.o> [name filter]
set -e
obj=${1:-build/obj/tdunes_device.hip.o}
filt=${2:-persist}
tmp=$(mktemp -d)
cp "$obj" $tmp/x.o
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1 || true)
co=$(ls $tmp/x.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|spill_count|\.private_segment_fixed_size" | sed 's/^ *//' | paste - - - - - - | grep -E "$filt" | sed 's/_ZN12_GLOBAL__N_1//' | awk '{printf "%-70s priv %s sgpr %s sspill %s vgpr %s vspill %s\n", substr($2,1,70), $4, $6, $8, $10, $12}'
rm -rf $tmp
