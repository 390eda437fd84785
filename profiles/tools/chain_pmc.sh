#!/bin/bash
# SQ counters of one fused-MLP launch per kernel family:  gpurun -- 'bash profiles/tools/chain_pmc.sh <tag> [rows]'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
tag=$1; rows=${2:-603992}
O=$R/gpurun_out/$tag; mkdir -p $O; : > $O/summary.txt
for fam in row col; do
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$fam$i -- python3 $R/profiles/tools/chain_pmc_run.py $fam $rows > $O/p$fam$i.log 2>&1
  f=$(find $O/p$fam$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $fam <<'PY' >> $O/summary.txt
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0]))
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    if 'chain' in n:
        a=acc[n][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for n,d in sorted(acc.items()):
    for k,(v,c) in d.items(): print(f"{sys.argv[2]:4s} {n[:44]:44s} {k:28s} {v/c:16.0f}  (mean of {c})")
PY
  rm -rf $O/p$fam$i
done
done
cat $O/summary.txt
