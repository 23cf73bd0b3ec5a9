"""Build libendosurf_hip.so (hipcc, gfx950 only) in-tree under endosurf_amd/lib/.

``python -m endosurf_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
Objects are rebuilt only when a source/header is newer (seconds per file after the first time).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libendosurf_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-pass-failed"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "endosurf_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, extra):
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    # an object is reused only if it was built with the SAME code-generation flags (-D...): a dev build (-DES_DEV_SWITCHES) followed by
    # a plain build must not link dev objects into the product library
    codegen = " ".join([*FLAGS, *[e for e in extra if not e.startswith("-R")]])
    stamp = obj + ".flags"
    same_flags = os.path.exists(stamp) and open(stamp).read() == codegen
    if (not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(srcp)
            and os.path.getmtime(obj) > _newest_header_mtime()):
        return obj, False, ""
    cmd = [hipcc(), *FLAGS, *extra, "-c", srcp, "-o", obj]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{p.stdout}\n{p.stderr}")
    with open(stamp, "w") as f:
        f.write(codegen)
    return obj, True, p.stderr


def build(force: bool = False, verbose: bool = True, extra=()) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, list(extra)), srcs))
    objs = [r[0] for r in res]
    rebuilt = any(r[1] for r in res)
    for (o, did, err), s in zip(res, srcs):
        if verbose and did:
            print(f"[endosurf_amd.build] compiled {s}" + (f"\n{err}" if err.strip() and "-Rpass" in " ".join(extra) else ""))
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"link failed:\n{p.stdout}\n{p.stderr}")
        if verbose:
            print(f"[endosurf_amd.build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a.startswith(("-R", "-save", "-D"))])
