"""Where a kernel spills: scratch_load / scratch_store instructions of one instantiation, with the count of s_barrier
instructions seen before each (= the phase of a barrier-separated kernel) and the line inside the kernel body.

    profiles/tools/kres.sh gen-fvgn-steady_amd/csrc/colchain.hip          # leaves /tmp/kres_colchain.s
    python profiles/tools/spills.py /tmp/kres_colchain.s colchain_bwd_kernelILb0ELb0ELb0ELb0ELb0ELb1E
"""
import re, sys
txt = open(sys.argv[1]).read()
m = re.search(r'^(_ZN\S*%s\S*):' % re.escape(sys.argv[2]), txt, re.M)
body = txt[m.end():txt.index('s_endpgm', m.end())].split('\n')
bar = 0
per = {}
for i, l in enumerate(body):
    t = l.strip()
    if t.startswith('s_barrier'):
        bar += 1
    if t.startswith('scratch_'):
        per.setdefault(bar, []).append((i, t.split(';')[0].strip()))
for b, items in sorted(per.items()):
    st = sum(1 for _, t in items if 'store' in t)
    print("after barrier %2d: %2d stores %2d loads   lines %d..%d" % (b, st, len(items) - st, items[0][0], items[-1][0]))
print('barriers', bar, 'lines', len(body))
