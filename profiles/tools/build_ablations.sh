#!/bin/bash
# Ablation builds of tchain.hip (-DABL_NOSEG / NOSTORE / NOGELU / NOW) into scratch/libgfv_<tag>.so; run a launch with GFV_LIB=<that .so> (profiles/r01_tchain_pmc.txt)
# build ablation variants of tchain into scratch/libgfv_<tag>.so
cd /root/repo/gen-fvgn-steady_amd/csrc/build
for v in "BASE:" "NOSEG:-DABL_NOSEG" "NOSTORE:-DABL_NOSTORE" "NOGELU:-DABL_NOGELU" "NOW:-DABL_NOW" "ALL:-DABL_NOSEG -DABL_NOSTORE -DABL_NOGELU -DABL_NOW"; do
  tag=${v%%:*}; fl=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $fl -c ../tchain.hip -o /tmp/tchain_$tag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $fl -c ../tchain_fwd.hip -o /tmp/tchain_fwd_$tag.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC dw.o fvm.o misc.o prof.o rowtile.o segreduce.o slice.o wimg.o /tmp/tchain_$tag.o /tmp/tchain_fwd_$tag.o -o /root/repo/scratch/libgfv_$tag.so &
done
wait
ls -la /root/repo/scratch/*.so
