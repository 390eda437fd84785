#!/bin/bash
# SQ / TCP counters per kernel (mean per launch) for kernels whose name contains one of the given substrings, from short
# eager bench runs (one rocprofv3 --pmc pass per counter set):  gpurun -- 'bash profiles/tools/sq_pmc_kernels.sh <tag> slice_ deslice'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
tag=$1; shift
O=$R/gpurun_out/$tag; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --graph off --min-time 0 --cpu-budget 0 --profile-steps 1 --skip-fp32-form --skip-drop-in > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$@" <<'PY' >> $O/summary.txt
import csv, sys, collections
pats=sys.argv[2:]
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0]))
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    if any(p in n for p in pats):
        a=acc[n][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for n,d in sorted(acc.items()):
    for k,(v,c) in d.items(): print(f"{n[:40]:40s} {k:30s} {v/c:16.0f}  (mean of {c})")
PY
  rm -rf $O/p$i
done
cat $O/summary.txt
