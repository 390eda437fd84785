#!/bin/bash
# usage: scratch/pmc.sh <lib> <tag>  -> SQ counters for the edge-MLP forward launch
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
[ -n "$1" ] && export GFV_LIB=$R/$1
tag=$2
mkdir -p $R/gpurun_out/$tag
cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$tag/p$i -- python3 profiles/tools/edge_mlp_launch.py > $R/gpurun_out/$tag/p$i.log 2>&1
  f=$(find $R/gpurun_out/$tag/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda:[0,0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'tchain' in r['Kernel_Name'] or 'rowtile' in r['Kernel_Name']:
        a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k,(v,n) in acc.items(): print(f"{k:32s} {v/n:16.0f}  (mean of {n} launches)")
PY
  find $R/gpurun_out/$tag/p$i -name "*.csv" -size +1M -delete
done
