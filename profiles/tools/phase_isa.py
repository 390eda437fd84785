"""Static instruction mix per barrier-separated phase of one kernel instantiation (straight-line count over the listing: loop
bodies count once, so for the tile loop of the column-owner kernels the numbers are per tile).

    profiles/tools/kres.sh gen-fvgn-steady_amd/csrc/colchain.hip
    python profiles/tools/phase_isa.py /tmp/kres_colchain.s colchain_bwd_kernelILb0ELb1ELb0ELb0ELb0ELb0E
"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
m = re.search(r'^(_ZN\S*%s\S*):' % re.escape(sys.argv[2]), txt, re.M)
body = txt[m.end():txt.index('s_endpgm', m.end())].split('\n')
phases = [Counter()]
for l in body:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
        continue
    op = t.split()[0]
    if op == 's_barrier':
        phases.append(Counter())
        continue
    if op.startswith('v_mfma'): k = 'mfma'
    elif op.startswith('v_exp') or op.startswith('v_rcp') or op.startswith('v_rsq') or op.startswith('v_log') or op.startswith('v_sqrt'): k = 'trans'
    elif op.startswith('v_'): k = 'valu'
    elif op.startswith('ds_'): k = 'lds'
    elif op.startswith('buffer_') or op.startswith('global_') or op.startswith('scratch_') or op.startswith('flat_'): k = 'vmem'
    elif op.startswith('s_waitcnt'): k = 'wait'
    elif op.startswith('s_'): k = 'salu'
    else: k = 'other'
    phases[-1][k] += 1
print("phase   valu trans  mfma   lds  vmem  salu  wait")
for i, c in enumerate(phases):
    print("%5d %6d %5d %5d %5d %5d %5d %5d" % (i, c['valu'], c['trans'], c['mfma'], c['lds'], c['vmem'], c['salu'], c['wait']))
tot = sum(phases[3:], Counter()) if len(phases) > 4 else Counter()
print("from phase 3 on: valu %d trans %d mfma %d lds %d vmem %d" % (tot['valu'], tot['trans'], tot['mfma'], tot['lds'], tot['vmem']))
