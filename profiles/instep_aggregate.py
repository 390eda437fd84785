"""rocprofv3 --kernel-trace --stats of the bench command -> average duration (us) of every kernel class INSIDE the two-stream
training step, keyed by the kernel names bench.py reports (profiles/pmc_aggregate.py's naming): kernel_stats_in_step.json.
The stats must come from a run that executes ONE form of the step only (`--graph list --skip-fp32-form --profile-steps 0
--cpu-budget 0`: no eager calibration, no fp32 / reduced-precision forms, no single-stream profile steps), so that template
instantiations folded into one class (LOWP, element ops) are not averaged across forms.  The library's source hash is stored
with the averages ("__lib_srchash__"): bench.py ignores a file that belongs to another build.
usage: instep_aggregate.py <kernel_stats.csv> <out.json> [libgfv.so.srchash]"""
import csv
import json
import re
import sys


def key_of(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"tchain_kernel<([^>]*)>", name)
    if m:
        return "tchain_kernel<" + ", ".join(a.strip() for a in m.group(1).split(",")[:6]) + ", *>"
    m = re.match(r"([A-Za-z_0-9]+)", name)
    base = m.group(1) if m else name
    if base.startswith("seg_gather_sum"):
        return "seg_gather_sum_vec"
    return base


acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = key_of(r["Name"])
    a = acc.setdefault(k, [0.0, 0])
    a[0] += float(r["TotalDurationNs"])
    a[1] += int(r["Calls"])
out = {k: v[0] / v[1] / 1e3 for k, v in acc.items() if v[1]}
if len(sys.argv) > 3:
    try:
        out["__lib_srchash__"] = open(sys.argv[3]).read().strip()
    except OSError:
        pass
json.dump(out, open(sys.argv[2], "w"), indent=0, sort_keys=True)
for k, v in sorted(((k, v) for k, v in out.items() if k in acc), key=lambda kv: -acc[kv[0]][0])[:14]:
    print(f"{k:60s} {acc[k][1]:7d} launches  {v:8.1f} us")
