#!/bin/bash
# The whole GPU suite under each documented switch in turn: ~2 minutes per setting on the GPU box.
#   gpurun --timeout 3000 -- 'bash profiles/tools/suite_switches.sh > gpurun_out/suite_switches.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "X=0" "GFV_RECOMPUTE=1" "GFV_CMDLIST_NATIVE=0" "GFV_OVERLAP=0" "GFV_DEFER=0" "GFV_DROPIN_REPLAY=0" "GFV_NATIVE_PLAN=0" \
           "GFV_PREP_FUSE=0 GFV_FVM_FUSE=0 GFV_TRANS_REDUCE_MERGE=0" "GFV_CSR_FUSE=0" "GFV_EDGE_FACTOR=0" "GFV_FUSE_DW=0" \
           "GFV_CBWD=0 GFV_CFWD=0 GFV_CTRANS=0 GFV_LIN1S=0"; do
  echo "== $cfg"
  env $cfg timeout 900 python3 -m pytest tests -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tail -12
done
