"""Build libphyx_amd.so (HIP, gfx950 only) in-tree with hipcc.

No CUDA shims, no multi-arch fat binary, one code path.  -ffp-contract=off is part of the numerical
contract (the kernels must round exactly like the strict-IEEE oracle), not a tuning choice.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libphyx_amd.so")
SOURCES = ["runtime.hip", "schedule.hip", "solver.hip", "c_api_solver.hip", "broadphase.hip", "world.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-shared"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "phyx_amd.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc()] + FLAGS + ["-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed building libphyx_amd.so")
    if verbose and res.stdout.strip():
        print(res.stdout)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
