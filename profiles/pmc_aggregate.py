"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/collect.sh) into per-kernel HBM-side traffic.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts the 128-B requests of wide streaming reads as 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section): the read figure is doubled.  Infinity-Cache hits are included
(the counters sit on the L2 <-> fabric interface).  Writes pmc_traffic.json (bytes per launch, keyed by the kernel
names bench.py reports, plus "__step_total__": all kernels of one training step summed) next to the text table printed on
stdout.  usage: pmc_aggregate.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <steps the profiled command ran>"""
import collections
import csv
import glob
import json
import os
import re
import sys

out_dir = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0


def key_of(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"tchain_kernel<([^>]*)>", name)
    if m:   # the classes bench.py prices: the element-op parameter (7th) is folded into its class
        return "tchain_kernel<" + ", ".join(a.strip() for a in m.group(1).split(",")[:6]) + ", *>"
    m = re.match(r"([A-Za-z_0-9]+)", name)
    base = m.group(1) if m else name
    return "seg_gather_sum_vec" if base.startswith("seg_gather_sum") else base


acc = {c: collections.defaultdict(lambda: [0.0, 0]) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    for f in glob.glob(os.path.join(out_dir, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            a = acc[c][key_of(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
print("# HBM-side traffic per launch: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of one bench.py run,")
print("# mean over launches; KiB counters, reads x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B), Infinity-Cache hits")
print(f"# included.  The profiled command ran {steps:g} training steps (all launch modes / forms it times count).")
print("kernel, launches, FETCH_SIZE KiB (raw), read MB (x2), WRITE_SIZE KiB, write MB, traffic MB per launch, MB per step")
traffic, total = {}, 0.0
rows = []
for key in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f, w = acc["FETCH_SIZE"].get(key, [0.0, 0]), acc["WRITE_SIZE"].get(key, [0.0, 0])
    n = max(f[1], w[1])
    if not n:
        continue
    fk, wk = f[0] / n, w[0] / n
    rd, wr = 2 * fk * 1024 / 1e6, wk * 1024 / 1e6
    traffic[key] = int((rd + wr) * 1e6)
    per_step = (rd + wr) * n / steps
    total += per_step
    rows.append((per_step, f"{key}, {n}, {fk:.1f}, {rd:.2f}, {wk:.1f}, {wr:.2f}, {rd + wr:.2f}, {per_step:.1f}"))
for _, line in sorted(rows, reverse=True):
    print(line)
print(f"# total: {total / 1e3:.2f} GB per step")
traffic["__step_total__"] = int(total * 1e6)
json.dump(traffic, open(os.path.join(out_dir, "pmc_traffic.json"), "w"))
