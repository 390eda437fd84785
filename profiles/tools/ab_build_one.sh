#!/bin/bash
# like ab_build.sh, with the extra flags applied to ONE source only:  ab_build_one.sh <name> <file.hip> <extra flags...>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; one=$2; shift; shift
B=$R/gen-fvgn-steady_amd/csrc/build_$name
mkdir -p $B
for f in $R/gen-fvgn-steady_amd/csrc/*.hip; do
  extra=""
  [ "$(basename $f)" = "$one" ] && extra="$*"
  own=$(head -1 $f | sed -n 's|^// gfv-build-flags:||p')   # per-file flags, as gfv/build.py reads them
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $own $extra -c $f -o $B/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o $R/gen-fvgn-steady_amd/gfv/libgfv_$name.so
rm -rf $B
echo built libgfv_$name.so
