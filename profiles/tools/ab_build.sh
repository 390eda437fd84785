#!/bin/bash
# Build an experiment variant of libgfv.so next to the product one:  profiles/tools/ab_build.sh <name> <extra hipcc flags...>
# -> gen-fvgn-steady_amd/gfv/libgfv_<name>.so ; select it at run time with GFV_LIB=<path> (gfv/lib.py).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
B=$R/gen-fvgn-steady_amd/csrc/build_$name
mkdir -p $B
for f in $R/gen-fvgn-steady_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c $f -o $B/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o $R/gen-fvgn-steady_amd/gfv/libgfv_$name.so
rm -rf $B
echo built libgfv_$name.so
