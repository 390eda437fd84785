#!/bin/bash
# One step's device timeline from rocprofv3 --kernel-trace (command-list replay): per hardware queue busy time, gaps, and the
# longest kernels;  gpurun -- '[ABFLAGS="--workload cavity --cells 5041"] [TL_TAG=timeline_cavity] bash profiles/tools/timeline.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TL_TAG:-timeline}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --cpu-budget 0 --min-time 0.3 --graph list --skip-fp32-form --profile-steps 0 --skip-copy-rate --skip-drop-in $ABFLAGS > $O/bench.json 2> $O/err.txt
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# steps are delimited by adam_kernel; take the last 40 complete steps
idx=[i for i,r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
idx=idx[-70:-20] if len(idx)>90 else idx[2:-2]   # (long steps - 8 meshes per GPU - run fewer of them)
agg=collections.defaultdict(lambda:[0.0,0.0,0]); steps=0; span=0.0
gapsum=collections.defaultdict(float)
for a,b in zip(idx[:-1],idx[1:]):
    seg=rows[a+1:b+1]
    t0=min(int(r["Start_Timestamp"]) for r in seg); t1=max(int(r["End_Timestamp"]) for r in seg)
    span+=(t1-t0); steps+=1
    byq=collections.defaultdict(list)
    for r in seg: byq[r["Queue_Id"]].append(r)
    for q,rs in byq.items():
        busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rs)
        agg[q][0]+=busy; agg[q][2]+=len(rs)
        rs.sort(key=lambda r:int(r["Start_Timestamp"]))
        gaps=sum(max(0,int(n["Start_Timestamp"])-int(p["End_Timestamp"])) for p,n in zip(rs[:-1],rs[1:]))
        agg[q][1]+=gaps
print("steps",steps,"span/step ms",span/steps/1e6)
for q,(busy,gaps,n) in agg.items(): print("queue",q,"kernels/step %.1f busy %.3f ms gaps %.3f ms" % (n/steps,busy/steps/1e6,gaps/steps/1e6))
# one step: list main-queue gaps > 3 us with neighbours
a,b=idx[len(idx)//2],idx[len(idx)//2+1]
seg=rows[a+1:b+1]
mainq=max(agg,key=lambda q:agg[q][0])
rs=[r for r in seg if r["Queue_Id"]==mainq]; rs.sort(key=lambda r:int(r["Start_Timestamp"]))
for p,n in zip(rs[:-1],rs[1:]):
    g=int(n["Start_Timestamp"])-int(p["End_Timestamp"])
    if g>3000: print("gap %.1f us after %s before %s" % (g/1e3,p["Kernel_Name"][:50].replace("(anonymous namespace)::",""),n["Kernel_Name"][:50].replace("(anonymous namespace)::","")))
PY
# the same step, both queues in time order (offsets in us from the step's first kernel): what the side queue does and when
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
idx=idx[-70:-20] if len(idx)>90 else idx[2:-2]
a,b=idx[len(idx)//2],idx[len(idx)//2+1]
seg=rows[a+1:b+1]
t0=int(seg[0]["Start_Timestamp"])
qs=sorted({r["Queue_Id"] for r in seg})
print("queues",qs)
for r in seg:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:46]
    s=(int(r["Start_Timestamp"])-t0)/1e3; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    print("%s %8.1f %7.1f  %s%s" % (r["Queue_Id"], s, d, "" if r["Queue_Id"]==qs[0] else "            ", n))
PY
rm -rf $O/prof
