#!/bin/bash
# Build a variant of libgfv.so with extra compiler flags next to the product library (A/B experiments on one box):
#   bash profiles/tools/build_variant.sh nt "-DGFV_NT_SAVE=1"      -> profiles/tools/variants/libgfv_nt.so
# run with  GFV_LIB=profiles/tools/variants/libgfv_nt.so python bench.py ...
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; extra=$2
O=$R/profiles/tools/variants; B=/tmp/gfv_variant_$name
mkdir -p $O $B
pids=()
for f in $R/gen-fvgn-steady_amd/csrc/*.hip; do
  ff=$(head -1 $f | sed -n 's#^// gfv-build-flags:##p')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $ff $extra -c $f -o $B/$(basename $f .hip).o &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o $O/libgfv_$name.so
echo built $O/libgfv_$name.so
