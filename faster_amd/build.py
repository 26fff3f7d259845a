"""Builds faster_amd/libfasterhip.so (hipcc, gfx950 only) in-tree so that it travels to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libfasterhip.so")
SOURCES = [os.path.join(HERE, "csrc", "fh_capi.hip"), os.path.join(HERE, "csrc", "fh_pool.hip"),
           os.path.join(HERE, "csrc", "fh_map.hip")]
import glob  # noqa: E402

DEPS = SOURCES + sorted(glob.glob(os.path.join(HERE, "csrc", "*.hpp"))) + [os.path.join(ROOT, "include", "fasterhip.h")]
HOST_SO = os.path.join(HERE, "libsolverhip.so")
HOST_SOURCES = [os.path.join(HERE, "host", "solver_hip.cpp"), os.path.join(HERE, "host", "decomp_hip.cpp"),
                os.path.join(HERE, "host", "jps_hip.cpp"), os.path.join(HERE, "host", "corridor_frontend.cpp")]  # (JpsHip searches ONE query on the host)
HOST_DEPS = HOST_SOURCES + [os.path.join(HERE, "host", "solver_hip.hpp"), os.path.join(HERE, "host", "faster_stub.hpp"),
                            os.path.join(HERE, "host", "decomp_hip.hpp"), os.path.join(HERE, "host", "jps_hip.hpp"),
                            os.path.join(HERE, "host", "corridor_frontend.hpp"),
                            os.path.join(ROOT, "include", "fasterhip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build_device(force=False, verbose=False):
    if force or _stale(SO, DEPS):
        # -sink-insts-to-avoid-spills / -disable-machine-licm: the solve kernels are compiled for 3 wavefronts per SIMD (168 vector
        # registers); with the default hoisting of loop invariants out of the branch-and-bound loops they spill ~150 registers to
        # scratch (HBM round trips in the per-problem code), with these two options ~20 (measured: +6 % pairs/s).
        # FASTERHIP_EXTRA_FLAGS: diagnostic / experimental builds only (e.g. -DFH_PROFILE)
        extra = os.environ.get("FASTERHIP_EXTRA_FLAGS", "").split()
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-sink-insts-to-avoid-spills",
               "-mllvm", "-disable-machine-licm"] + extra + ["-o", SO] + SOURCES
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return SO


def build_host(force=False, verbose=False):
    """C++ SolverHip class (SolverGurobi surface) linked against libfasterhip.so."""
    if not all(os.path.exists(s) for s in HOST_SOURCES):
        return None
    if force or _stale(HOST_SO, HOST_DEPS + [SO]):
        cmd = ["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-o", HOST_SO] + HOST_SOURCES + [
            "-L", HERE, "-lfasterhip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_SO


FRONT_SO = os.path.join(HERE, "libfasterfront.so")
FRONT_SOURCES = [os.path.join(HERE, "host", "corridor_frontend.cpp")]
FRONT_DEPS = FRONT_SOURCES + [os.path.join(HERE, "host", "corridor_frontend.hpp")]


def build_frontend(force=False, verbose=False):
    """CPU corridor front-end (voxel path search + ellipsoid decomposition), SURVEY.md 8(f) N1."""
    if force or _stale(FRONT_SO, FRONT_DEPS):
        cmd = ["g++", "-O2", "-std=c++14", "-fPIC", "-fopenmp", "-shared", "-o", FRONT_SO] + FRONT_SOURCES
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return FRONT_SO


def build_all(force=False, verbose=False):
    build_device(force, verbose)
    build_host(force, verbose)
    build_frontend(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
