"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/collect.sh) into per-kernel HBM-side traffic.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts the 128-B requests of wide streaming reads as 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section): the read figure is doubled.  Infinity-Cache hits are included
(the counters sit on the L2 <-> fabric interface).  Writes pmc_traffic.json (bytes per launch, keyed by the kernel
names bench.py reports) next to the text table printed on stdout."""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
_CH = [f"tchain_kernel<1, {lnm}, {rag}, {h}>" for lnm, rag in ((0, "false"), (1, "false"), (2, "false"), (0, "true"))
       for h in ("true", "false")]   # last template argument: split-fp16 products (true) / fp32 MFMA (false)
KEYS = {k: k for k in (*_CH, "rowtile_chain_kernel", "dw_multi_h_kernel", "dw_multi_kernel", "seg_gather_sum_vec")}
acc = {c: collections.defaultdict(lambda: [0.0, 0]) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    for f in glob.glob(os.path.join(out_dir, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            for pat, key in KEYS.items():
                if pat in r["Kernel_Name"]:
                    a = acc[c][key]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
                    break
print("# HBM-side traffic per launch: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of")
print("# `bench.py --steps 3 --warmup 1 --no-graph --cpu-budget 0 --profile-steps 1`, mean over launches; KiB counters,")
print("# reads x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B), Infinity-Cache hits included.")
print("kernel, launches, FETCH_SIZE KiB (raw), read MB (x2), WRITE_SIZE KiB, write MB, traffic MB per launch")
traffic = {}
for key in KEYS.values():
    f, w = acc["FETCH_SIZE"].get(key), acc["WRITE_SIZE"].get(key)
    if not f or not w or not f[1] or not w[1]:
        continue
    fk, wk = f[0] / f[1], w[0] / w[1]
    rd, wr = 2 * fk * 1024 / 1e6, wk * 1024 / 1e6
    traffic[key] = int((rd + wr) * 1e6)
    print(f"{key}, {f[1]}, {fk:.1f}, {rd:.2f}, {wk:.1f}, {wr:.2f}, {rd + wr:.2f}")
json.dump(traffic, open(os.path.join(out_dir, "pmc_traffic.json"), "w"))
