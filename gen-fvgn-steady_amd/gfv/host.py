"""Host-side placement of the training loop's threads (round 6).

The drop-in iteration on a small mesh is bound by the HOST: ~2 ms of Python / autograd work on two threads (the caller's, and
PyTorch's autograd device thread, which runs the backward of the model's autograd node and its 159 AccumulateGrad nodes) against
1.5 ms of device work.  On the GPU box's 256-CPU host the scheduler places the two threads anywhere; when they sit on different
core complexes every Python object they both touch (parameters, the autograd context, reference counts) migrates between L3
caches: the same loop takes 1.8 - 2.3 ms per iteration with the threads placed freely and 1.48 - 1.50 ms with the process confined
to the CPUs of ONE L3 (profiles/r06_host_affinity.txt; torch.optim.Adam: 2.8 - 3.0 against 2.3 ms).  `pin_to_l3()` does that for
the calling thread - and every thread it creates afterwards inherits it (call it before the first backward).  A driver calls it
once at its top, or is started under `taskset`; nothing in this package calls it implicitly (a library does not move its host
process around) - bench.py does, and says so in its JSON line.
"""
from __future__ import annotations

import os


def _parse_cpu_list(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def l3_group(cpu):
    """The logical CPUs that share the last-level cache of `cpu` (empty set: not known on this host)."""
    for idx in (3, 2):
        path = f"/sys/devices/system/cpu/cpu{cpu}/cache/index{idx}/shared_cpu_list"
        try:
            with open(path) as f:
                return _parse_cpu_list(f.read())
        except OSError:
            continue
    return set()


def l3_groups(allowed=None):
    """The distinct L3 groups among the allowed CPUs, in CPU order."""
    allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
    groups, seen = [], set()
    for c in sorted(allowed):
        if c in seen:
            continue
        g = (l3_group(c) & allowed) or {c}
        seen |= g
        groups.append(g)
    return groups


def pin_to_l3(cpu=None, rank=None):
    """Confine the calling thread (and the threads it will create) to the CPUs sharing the L3 of `cpu` (default: the CPU it is
    running on).  `rank` (one process per GPU on one node): the rank-th L3 group of the host instead, so that the ranks of a job
    never share one.  Returns the previous affinity set (hand it to `restore`), or None when nothing was changed."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    try:
        prev = os.sched_getaffinity(0)
        if rank is not None:
            groups = l3_groups(prev)
            if len(groups) < 2:
                return None
            group = groups[int(rank) % len(groups)]
            os.sched_setaffinity(0, group)
            return prev
        cpu = os.sched_getcpu() if cpu is None and hasattr(os, "sched_getcpu") else (cpu if cpu is not None else min(prev))
        group = l3_group(cpu) & prev
        if len(group) < 2 or group == prev:
            return None
        os.sched_setaffinity(0, group)
        return prev
    except OSError:
        return None


def restore(prev):
    if prev:
        try:
            os.sched_setaffinity(0, prev)
        except OSError:
            pass
