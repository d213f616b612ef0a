"""Builds libjutul_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so the
shared object travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libjutul_hip.so")
SOURCES = ["jh_setup.cpp", "jh_api.cpp", "jh_comm.cpp", "jh_kernels.hip", "jh_assembly.hip", "jh_ilu.hip",
           "jh_krylov.hip", "jh_halo.hip", "jh_custom.cpp", "jh_sell.hip", "jh_partition.cpp", "jh_subdomain.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-I/opt/rocm/include"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    headers = [os.path.join(CSRC, "jh_internal.hpp"), os.path.join(HERE, "..", "include", "jutul_hip.h")]
    extra = os.environ.get("JH_EXTRA_FLAGS", "").split()  # compile-time experiments, e.g. -DJH_PFW_SCALAR=12 (use with force)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(bdir, src + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc, "-x", "hip"] + FLAGS + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for w in ex.map(run, jobs):
            if verbose and w.strip():
                print(w)
    if jobs or force or _newer(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
