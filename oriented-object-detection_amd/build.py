"""Builds libobbhip.so (hand-written HIP for gfx950) in-tree with hipcc.  No JIT cache, no torch extension:
the .so travels to the GPU box with the repo snapshot and is loaded with ctypes (see _lib.py)."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libobbhip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-math-errno", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libobbhip.so)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "obbhip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, jobs=None):
    if not force and not stale():
        return OUT
    cc = hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max([os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))] +
                [os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "obbhip.h"))])
    procs, objs = [], []
    jobs = jobs or min(8, os.cpu_count() or 1)
    pending = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        pending.append([cc, f"--offload-arch={ARCH}", *FLAGS, "-c", src, "-o", obj])
    while pending or procs:
        while pending and len(procs) < jobs:
            cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        cmd, p = procs.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            for _, q in procs:
                q.kill()
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    link = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT, *objs]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
