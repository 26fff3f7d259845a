"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/fasterhip.h
declares, struct layouts agree between the header (as compiled by gcc into the oracle), numpy and the HIP
library, and the product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from faster_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from faster_amd import build as fb

    fb.build_all()
    return fb


def declared_functions():
    text = open(os.path.join(ROOT, "include", "fasterhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fh_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_entry_points():
    names = declared_functions()
    for n in ("fh_create", "fh_destroy", "fh_solve_batch", "fh_solve_batch_device", "fh_sample_batch", "fh_sample_batch_device",
              "fh_pair_glue_device", "fh_sync", "fh_timing_read", "fh_last_error", "fh_version"):
        assert n in names


def test_header_compiles_on_its_own_as_c_and_cxx():
    """include/fasterhip.h is the drop-in boundary: a C99 or C++11 translation unit that includes nothing else must compile
    (a cgo / JNI / ctypes-generator consumer sees exactly this file)."""
    import subprocess

    hdr = os.path.join(ROOT, "include", "fasterhip.h")
    for cmd in (["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-pedantic", hdr],
                ["g++", "-fsyntax-only", "-x", "c++", "-std=c++11", "-Wall", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0 and not r.stderr.strip(), (cmd, r.stderr[-2000:])


def test_library_exports_every_declared_symbol(built):
    from faster_amd import capi

    L = capi.lib()
    for name in declared_functions():
        assert hasattr(L, name), "libfasterhip.so does not export %s" % name
    assert set(capi.SYMBOLS) == set(declared_functions())
    assert L.fh_version().decode().startswith("fasterhip")


def test_library_is_gfx950_only(built):
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    import subprocess

    out = subprocess.run([objdump, "--offloading", built.SO], capture_output=True, text=True).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs


def test_struct_layouts(oracle, built):
    L = oracle.lib()
    assert L.orc_sizeof_problem() == abi.problem_dtype.itemsize == 264
    assert L.orc_sizeof_result() == abi.result_dtype.itemsize == 1600
    assert abi.problem_dtype.fields["dc"][1] == 64 and abi.problem_dtype.fields["x0"][1] == 120
    assert abi.result_dtype.fields["factor"][1] == 24 and abi.result_dtype.fields["coeff"][1] == 48
    assert abi.result_dtype.fields["assign"][1] == 48 + 8 * 16 * 12
    from faster_amd import capi

    p = np.zeros((), dtype=abi.params_dtype)
    capi.lib().fh_default_params(abi.ptr(p.reshape(1)))
    d = abi.default_params()
    for f in ("feas_tol", "dep_tol", "max_nodes", "max_iters"):
        assert p[f] == d[f]


def test_no_cpu_fallback_without_gpu(built):
    """Without a HIP device the product refuses to run (it must never route through the oracle or a CPU path)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from faster_amd import capi

    with pytest.raises(capi.FasterHipError) as e:
        capi.Context(0)
    assert "no HIP device" in str(e.value)
    h = ctypes.c_void_p()
    rc = capi.lib().fh_create(ctypes.byref(h), -1)
    assert rc == -2 and h.value
    pr = abi.make_problems(1)
    res = np.zeros(1, dtype=abi.result_dtype)
    assert capi.lib().fh_solve_batch(h, abi.ptr(pr), None, 0, 1, abi.ptr(res)) == -2
    assert capi.lib().fh_sync(h) == -2
    # the entry points of the faithful replan: arguments are checked first (FH_ERR_ARG = -1), then the missing device is reported — never a CPU path
    L, vp = capi.lib(), ctypes.c_void_p
    dummy = np.zeros(64)
    d = abi.ptr(dummy)
    grid = np.zeros(1, dtype=abi.voxel_grid_dtype)
    grid["res"], grid["dims"] = 0.2, (4, 4, 4)
    bbox = np.array([2.0, 2.0, 1.0])
    assert L.fh_append_plans_device(h, d, d, d, d, 4, 2.0, 8, d, d, None) == -1          # r_frac > 1
    assert L.fh_append_plans_device(h, d, d, d, d, 4, 0.5, 8, d, d, None) == -2
    assert L.fh_corridor_problems_device(h, d, d, d, d, d, d, 4, 96, 99, d) == -1        # n_seg > FH_MAX_SEG
    assert L.fh_corridor_problems_device(h, d, d, d, d, d, d, 4, 96, 6, d) == -2
    assert L.fh_safe_corridor_batch_device(h, d, d, d, d, 100, d, d, 4, abi.ptr(grid), 4, 0.5, 3, abi.ptr(bbox), 0.05, 0.0, 96, 6, d, d, None, None) == -1  # max_points
    assert L.fh_safe_corridor_batch_device(h, d, d, d, d, 4, d, d, 4, abi.ptr(grid), 4, 0.5, 3, abi.ptr(bbox), 0.05, 0.0, 96, 6, d, d, None, None) == -2
    capi.lib().fh_destroy(h)
    # the voxel map / path search and the device pool have no CPU path either
    with pytest.raises(capi.FasterHipError):
        capi.Map(0)
    mh = ctypes.c_void_p()
    assert capi.lib().fh_map_create(ctypes.byref(mh), 0) == -2 and not mh.value


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "faster_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert "liboracle" not in text and "faster_oracle" not in text.replace("oracle/faster_oracle.c", ""), fn
    # the diagnostic scripts are not test infrastructure either
    for fn in os.listdir(os.path.join(ROOT, "scripts")):
        if fn.endswith((".py", ".sh")):
            text = open(os.path.join(ROOT, "scripts", fn), errors="ignore").read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) and "liboracle" not in text, fn
