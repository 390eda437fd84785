"""Build libgfv.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libgfv.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def file_flags(src):
    """Extra hipcc flags a source asks for in its first line: `// gfv-build-flags: <flags>`."""
    with open(src) as f:
        first = f.readline()
    tag = "// gfv-build-flags:"
    return first[len(tag):].split() if first.startswith(tag) else []


def source_hash():
    """sha256 over every source the library is built from (a fresh checkout has arbitrary mtimes: the hash decides)."""
    import hashlib
    h = hashlib.sha256()
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "gfv.h"))
    for f in sources() + hdrs:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _deps(src, seen=None):
    """The local headers a source includes (transitively): `#include "x.h"` relative to the including file."""
    import re
    seen = set() if seen is None else seen
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(src).read(), re.M):
        h = os.path.normpath(os.path.join(os.path.dirname(src), m.group(1)))
        if h not in seen and os.path.exists(h):
            seen.add(h)
            _deps(h, seen)
    return sorted(seen)


def _unit_hash(src):
    import hashlib
    h = hashlib.sha256()
    for f in [src] + _deps(src):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS + file_flags(src)).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> gfv/libgfv.so.  Up to date = the library exists AND was built from exactly these sources
    (content hash kept next to it); otherwise the objects whose source or headers changed (a content hash per object, kept
    next to it) are recompiled and the library is linked again."""
    stamp = LIB + ".srchash"
    want = source_hash()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return LIB
    srcs = sources()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        uh = _unit_hash(s)
        hs = o + ".hash"
        if force or not os.path.exists(o) or not os.path.exists(hs) or open(hs).read().strip() != uh:
            jobs.append(([HIPCC, *FLAGS, *file_flags(s), "-c", s, "-o", o], hs, uh))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    def compile_one(job):
        cmd, hs, uh = job
        if os.path.exists(hs):
            os.remove(hs)
        run(cmd)
        with open(hs, "w") as f:
            f.write(uh + "\n")

    # the heavy units first (the persistent backward's and the row-owner chain's instantiations take 20 - 27 s each, most others
    # 3 - 6 s: started last they would be the tail of the build), one hipcc per CPU up to 8
    weight = {"colchain.hip": 9, "fvm.hip": 8, "tchain.hip": 7, "lin1.hip": 6, "plan.hip": 5}
    jobs.sort(key=lambda j: (-weight.get(os.path.basename(j[0][-3]), 0), -os.path.getsize(j[0][-3])))
    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 4))) as ex:
        list(ex.map(compile_one, jobs))
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
