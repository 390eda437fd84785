"""Aggregation behind profiles/tools/latency_floor.sh.
  --reduce <kernel_trace.csv> <out.json>: the last complete steps of a rocprofv3 kernel trace (steps end with adam_kernel) ->
      per launch POSITION in the step (the step is a fixed launch sequence): kernel name, grid, mean / min duration
  --table <dir>: trace_<cells>.json of several mesh sizes -> per kernel class (name + role: node / edge / cell / fixed, from how
      its grid grows) launches per step, microseconds at every size, floor (intercept) and slope (us per 1 000 rows) of a
      least-squares line over the sizes"""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n


def reduce(trace, out):
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel") or " adam_kernel" in r["Kernel_Name"] or "::adam_kernel" in r["Kernel_Name"]]
    ends = ends[-61:-1] if len(ends) > 70 else ends[len(ends) // 2:]
    steps = [rows[a + 1:b + 1] for a, b in zip(ends[:-1], ends[1:])]
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n]
    pos = []
    for i in range(n):
        d = [int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"]) for s in steps]
        r = steps[0][i]
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        pos.append({"kernel": short(r["Kernel_Name"]), "wgs": grid // max(wg, 1), "us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3})
    span = [int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in steps]
    busy = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s) for s in steps]
    json.dump({"steps": len(steps), "launches": n, "span_us": sum(span) / len(span) / 1e3, "busy_us": sum(busy) / len(busy) / 1e3,
               "pos": pos}, open(out, "w"))


def fit(xs, ys):
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    sxx = sum((x - mx) ** 2 for x in xs)
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sxx if sxx else 0.0
    return my - b * mx, b


def table(d):
    sizes = sorted(int(f[6:-5]) for f in os.listdir(d) if f.startswith("trace_") and f.endswith(".json"))
    tr = {c: json.load(open(os.path.join(d, f"trace_{c}.json"))) for c in sizes}
    print("# floor + slope per kernel of the step: cavity meshes, single stream, command-list replay, rocprofv3 kernel trace")
    print("# cells            " + "".join(f"{c:>10d}" for c in sizes))
    print("# launches / step  " + "".join(f"{tr[c]['launches']:>10d}" for c in sizes))
    print("# busy us / step   " + "".join(f"{tr[c]['busy_us']:>10.1f}" for c in sizes))
    print("# span us / step   " + "".join(f"{tr[c]['span_us']:>10.1f}" for c in sizes))
    # classes: (kernel name, rank of its grid size among that kernel's launches of the step) - the node-level, edge-level and
    # fixed-size launches of one kernel are different classes; the ranks line up across mesh sizes
    def classes(c):
        by = collections.defaultdict(list)
        for p in tr[c]["pos"]:
            by[p["kernel"]].append(p)
        out = {}
        for k, ps in by.items():
            grids = sorted({p["wgs"] for p in ps})
            for p in ps:
                out.setdefault((k, grids.index(p["wgs"]), len(grids)), []).append(p)
        return out
    cls = {c: classes(c) for c in sizes}
    keys = []
    for c in sizes:
        for k in cls[c]:
            if k not in keys:
                keys.append(k)
    groups = collections.OrderedDict((k, None) for k in keys)
    same = False
    print(f"{'kernel':58s} {'role':6s} {'n':>3s} " + "".join(f"{c:>9d}" for c in sizes) + f" {'floor':>7s} {'us/1k cells':>11s}")
    tot = {c: 0.0 for c in sizes}
    lines = []
    for (k, rank, nr) in groups:
        us, cnt, wg = [], 0, []
        for c in sizes:
            v = cls[c].get((k, rank, nr), [])
            us.append(sum(p["us"] for p in v) / len(v) if v else float("nan"))
            wg.append(v[0]["wgs"] if v else 0)
            cnt = max(cnt, len(v))
        role = "fixed" if (wg[-1] and wg[0] and wg[-1] < 1.5 * wg[0]) else f"g{rank + 1}/{nr}"
        ok = [(c, u) for c, u in zip(sizes, us) if u == u]
        a, b = fit([c / 1e3 for c, _ in ok], [u for _, u in ok]) if len(ok) >= 2 else (float("nan"), float("nan"))
        ref = us[1] if len(us) > 1 and us[1] == us[1] else next((u for u in us if u == u), 0.0)
        lines.append((cnt * ref, f"{k[:58]:58s} {role:6s} {cnt:3d} " + "".join(f"{u:9.1f}" for u in us) + f" {a:7.1f} {b:11.3f}"
                      + "   wgs " + "/".join(str(w) for w in wg)))
    for _, ln in sorted(lines, key=lambda t: -t[0]):
        print(ln)


if __name__ == "__main__":
    if sys.argv[1] == "--reduce":
        reduce(sys.argv[2], sys.argv[3])
    else:
        table(sys.argv[2])
