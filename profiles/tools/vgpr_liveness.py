"""Straight-line VGPR liveness over the ISA of ONE kernel: where the register pressure peaks and which registers are alive there
for how long - what showed that the LayerNorm-backward chain kept 32 zeroed accumulators alive through its prologue.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only csrc/tchain.hip -o t.s
    awk '$0 ~ "^_ZN.*<mangled instantiation>.*:" {on=1} on && /s_endpgm/ {on=0} on' t.s > k.s
    python profiles/tools/vgpr_liveness.py k.s

Approximation: control flow is ignored (one backward pass over the listing), so loop-carried values show as long live ranges
and the number is a lower bound of what the allocator needs."""
import re,sys
from collections import Counter
lines=[l for l in open(sys.argv[1]).read().split('\n')]
ins=[]
def regs(o):
    r=set()
    for a,b in re.findall(r'v\[(\d+):(\d+)\]',o): r.update(range(int(a),int(b)+1))
    o2=re.sub(r'v\[\d+:\d+\]','',o)
    for a in re.findall(r'\bv(\d+)\b',o2): r.add(int(a))
    return r
for i,l in enumerate(lines):
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    m=re.match(r'^(\S+)\s*(.*)$',t)
    op,rest=m.group(1),m.group(2).split(';')[0]
    ops=[o.strip() for o in rest.split(',')] if rest else []
    if not ops: continue
    store = op.startswith('global_store') or op.startswith('ds_write') or op.startswith('scratch_store') or op.startswith('v_cmp') or op.startswith('s_') or op.startswith('buffer_store')
    d=set() if store else regs(ops[0])
    u=set()
    for o in (ops if store else ops[1:]): u|=regs(o)
    if 'mfma' in op: u|=regs(ops[-1])
    if op.startswith('v_fmac') or 'dpp' in op: u|=regs(ops[0])
    ins.append((i,op,d,u,t))
live=set(); press=[0]*len(ins); snaps={}
for k in range(len(ins)-1,-1,-1):
    i,op,d,u,t=ins[k]
    live-=d; live|=u
    press[k]=len(live); snaps[k]=None
mx=max(press); k0=press.index(mx)
print("max pressure",mx,"at line",ins[k0][0])
live=set()
for k in range(len(ins)-1,-1,-1):
    i,op,d,u,t=ins[k]
    live-=d; live|=u
    if k==k0: lv=set(live); break
rows=[]
for r in sorted(lv):
    dline=None
    for k in range(k0,-1,-1):
        if r in ins[k][2]: dline=k; break
    nline=None
    for k in range(k0+1,len(ins)):
        if r in ins[k][3]: nline=k; break
    rows.append((r, ins[dline][0] if dline is not None else -1, ins[dline][4][:50] if dline is not None else '?', ins[nline][0] if nline is not None else -1, ins[nline][4][:60] if nline is not None else '?'))
far=[x for x in rows if x[3]-ins[k0][0]>400 or x[3]==-1]
print("live at peak:",len(rows),"; next use more than 400 lines away:",len(far))
for x in far: print(x)
