"""Build csrc/*.hip into lib/libssie_hip.so with hipcc for gfx950 (in-tree, no JIT cache)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libssie_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-w"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_dep():
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def up_to_date():
    return os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_dep()


def build(force: bool = False, verbose: bool = True) -> str:
    if up_to_date() and not force:
        return LIB
    cc = hipcc()
    if cc is None:
        raise RuntimeError("hipcc not found: cannot build libssie_hip.so")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    dep_t = _newest_dep()

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
                and os.path.getmtime(obj) >= hdr_t):
            return obj
        cmd = [cc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
