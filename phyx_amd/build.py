"""Build libphyx_amd.so (HIP, gfx950 only) in-tree with hipcc.

No CUDA shims, no multi-arch fat binary, one code path.  -ffp-contract=off is part of the numerical
contract (the kernels must round exactly like the strict-IEEE oracle), not a tuning choice.

Each translation unit is compiled to its own object (in parallel, only when it or a header changed), then linked.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libphyx_amd.so")
SOURCES = ["runtime.hip", "schedule.hip", "solver.hip", "solver_build.hip", "islands.hip", "exchange.hip", "comm.hip", "c_api_solver.hip", "broadphase.hip", "world.hip", "reslab.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result"]
# per-file extras.  islands.hip: the SLP vectoriser pairs the joint update's multiplies and adds into v_pk_mul_f32 / v_pk_add_f32
# and pays for every pair with moves into adjacent registers; measured on the island kernel (bit-identical results): 83.0 us
# with it, 78.3 us without.  The HBM path's kernels (solver.hip) are a few per cent faster with it, so it stays on there.
# round 3: the machine scheduler's iterative-ilp strategy (orders each block for instruction-level parallelism instead of register
# pressure — the class step is one wave's dependent chain, and the kernel has registers to spare at 4 waves per SIMD): launch
# 76.6 -> 74.1 us together with the select-based tag update (tools/build_variants.sh A/B: max-ilp 75.5, max-memory-clause 75.0)
# round 5: with ONE straight-line form of the class step (island_kernel.h half_step) the default scheduler is the better one again: island
# launch 61.5 us under iterative-ilp, 60.5 under the default, 60.4 under max-ilp, 61.5 / 62.0 under max-memory-clause / iterative-minreg
FILE_FLAGS = {"islands.hip": ["-fno-slp-vectorize", "-Wno-unused-function"]}      # (it includes solver_kernels.h for the shared device helpers only)
# roctx ranges (phase names of the reference's MICROPROFILE scopes) are resolved at run time with dlopen: no link dependency
LINK = ["-shared", "-ldl"]
# the sweeps' arithmetic contract (include/phyx_amd.h phx_arith_mode): fused multiply-adds unless PHX_ARITH=source
if os.environ.get("PHX_ARITH", "fused") == "source":
    FLAGS = FLAGS + ["-DPHX_ARITH_FMA=0"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _header_time():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(HERE, "..", "include", "phyx_amd.h"), __file__]
    return max(os.path.getmtime(d) for d in deps)


def _flags_changed():
    """The objects on disk were compiled under other flags (PHX_ARITH): everything is stale."""
    stamp = os.path.join(OBJ, "flags.txt")
    return not os.path.exists(stamp) or open(stamp).read() != " ".join(FLAGS)


def needs_build():
    if not os.path.exists(OUT) or _flags_changed():
        return True
    t = os.path.getmtime(OUT)
    return _header_time() > t or any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in _sources())


def _compile(src, force, verbose):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _header_time()):
        return obj, ""
    cmd = [hipcc()] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", "-o", obj, path]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed compiling %s" % src)
    return obj, res.stdout


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    force = force or _flags_changed()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = list(pool.map(lambda s: _compile(s, force, verbose), _sources()))
    objs = [o for o, _ in results]
    log = "".join(t for _, t in results)
    cmd = [hipcc(), "--offload-arch=gfx950"] + LINK + ["-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed linking libphyx_amd.so")
    with open(os.path.join(OBJ, "flags.txt"), "w") as f:
        f.write(" ".join(FLAGS))
    if verbose and (log + res.stdout).strip():
        print(log + res.stdout)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
