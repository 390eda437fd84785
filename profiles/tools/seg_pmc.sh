#!/bin/bash
# HBM-side bytes and L2 hit rate of the segmented-reduce launches of the step, per launch shape (profiles/tools/seg_bench.py):
#   gpurun -- 'bash profiles/tools/seg_pmc.sh 8 > gpurun_out/seg_pmc_b8.txt'
# three rocprofv3 passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum), the program directly behind `--`.
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-8}
O=$R/gpurun_out/seg_pmc_$B
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SEG_PMC=1 GFV_SEG_FORM=${GFV_SEG_FORM:-0}
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pass$i -- python3 $R/profiles/tools/seg_bench.py --worker $B 3 /tmp/seg_pmc.pt > $O/pass$i.log 2>&1
done
# timings of the same launches, no profiler
unset SEG_PMC
SEG_FORMS=$GFV_SEG_FORM python3 $R/profiles/tools/seg_bench.py $B 30 > $O/times.txt 2>&1
python3 - $O <<'PY'
import csv, glob, os, re, sys
O = sys.argv[1]
shapes = [l.strip() for l in open(os.path.join(O, "pass1.log")) if l.startswith("SEGSHAPE")]
times = [float(re.search(r":\s+([0-9.]+) us", l).group(1)) for l in open(os.path.join(O, "times.txt")) if " us " in l]
vals = {}
for i in (1, 2, 3):
    rows = []
    for f in glob.glob(os.path.join(O, f"pass{i}", "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "seg_gather_sum" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        vals.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        vals[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
def per_shape(name):
    d = [v for _, v in sorted(vals.get(name, {}).items())]
    return [sum(d[4 * k + 1:4 * k + 4]) / 3.0 for k in range(len(shapes))] if len(d) >= 4 * len(shapes) else [float("nan")] * len(shapes)
fetch, write, hit, miss = (per_shape(n) for n in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"))
print("# segmented reduce per launch shape: rocprofv3 --pmc (separate passes; KiB counters; reads x2: gfx950 FETCH_SIZE counts the")
print("# 128-B requests of wide reads as 64 B, MI355X_MICROARCH.md), mean of 3 launches behind a warm-up launch; time: HIP events,")
print("# 30 back-to-back launches, no profiler")
for k, sh in enumerate(shapes):
    rd, wr = 2 * fetch[k] * 1024, write[k] * 1024
    us = times[k] if k < len(times) else float("nan")
    alg = float(re.search(r"distinct_bytes=(\d+)", sh).group(1))
    print(f"{sh.split('|')[0][9:].strip():50s} {us:7.2f} us | counter bytes {(rd + wr) / 1e6:7.1f} MB (read {rd / 1e6:6.1f} + write {wr / 1e6:6.1f}) = "
          f"{(rd + wr) / us / 1e6:6.2f} TB/s = {(rd + wr) / us / 1e6 / 8.0:.3f} of 8 TB/s | algorithmic {alg / 1e6:6.1f} MB = {alg / us / 1e6 / 8.0:.3f} | "
          f"L2 hit rate {hit[k] / (hit[k] + miss[k]):.3f}")
PY
