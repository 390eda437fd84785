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


def numa_cpus(node):
    """The logical CPUs of NUMA node `node` (empty set: not known)."""
    try:
        with open(f"/sys/devices/system/node/node{int(node)}/cpulist") as f:
            return _parse_cpu_list(f.read())
    except (OSError, ValueError):
        return set()


def gpu_numa_nodes():
    """NUMA node of every visible GPU, by device index (-1: not known), from the PCI address the runtime reports and sysfs.
    Queries device properties only - no context is created on the other devices.  [] when it cannot be told."""
    try:
        import torch
        out = []
        for i in range(torch.cuda.device_count()):
            p = torch.cuda.get_device_properties(i)
            addr = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            try:
                with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
                    out.append(int(f.read().strip()))
            except (OSError, ValueError):
                out.append(-1)
        return out
    except Exception:
        return []


def rank_l3_group(rank, allowed=None, gpu_nodes=None, device=None):
    """The L3 group for local rank `rank` of a one-process-per-GPU job: by default the rank-th group of the host; when the NUMA
    nodes of the GPUs are known (`gpu_nodes[device]`), a group on the GPU's OWN node - the k-th one for the k-th GPU of that node -
    so that the launch path of a GPU on the second socket does not cross the socket link.  None: fewer than two groups."""
    groups = l3_groups(allowed)
    if len(groups) < 2:
        return None
    fallback = groups[int(rank) % len(groups)]
    try:
        if not gpu_nodes or device is None or not (0 <= device < len(gpu_nodes)) or gpu_nodes[device] < 0:
            return fallback
        node = gpu_nodes[device]
        cpus = numa_cpus(node)
        local = [g for g in groups if g <= cpus]
        if not local:
            return fallback
        k = sum(1 for j in range(device) if gpu_nodes[j] == node)   # this GPU's position among the GPUs of its node
        k += int(rank) // len(gpu_nodes)                             # (more ranks than GPUs - a self-test: the next group)
        return local[k % len(local)]
    except Exception:
        return fallback


def pin_to_l3(cpu=None, rank=None, gpu_nodes=None, device=None):
    """Confine the calling thread (and the threads it will create) to the CPUs sharing the L3 of `cpu` (default: the CPU it is
    running on).  `rank` (one process per GPU on one node): an L3 group of its own instead (`rank_l3_group`: the rank-th of the
    host, or - `gpu_nodes` / `device` given - one on the GPU's NUMA node), so that the ranks of a job never share one.  Returns
    the previous affinity set (hand it to `restore`), or None when nothing was changed."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    try:
        prev = os.sched_getaffinity(0)
        if rank is not None:
            group = rank_l3_group(rank, prev, gpu_nodes, device)
            if not group:
                return None
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
