"""The reduced-space basis tables of the solve kernels (faster_amd/csrc/fh_basis.hip.hpp: computed on the host, read by the device)
checked on the CPU by tests/cpp/test_basis.cpp: orthogonality, annihilation of the final-state functionals of
setConstraintsXf (solverGurobi.cpp:332-357) at any step, the minimum-norm particular solution, the consistency rows for N < 3, the
inverse row norms and the structurally constant rows of a whole trajectory."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_basis_tables():
    exe = os.path.join(ROOT, "tests", "cpp", "test_basis")
    src = exe + ".cpp"
    hdr = os.path.join(ROOT, "faster_amd", "csrc", "fh_basis.hip.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-2000:]
    assert int(r.stdout.split()[1]) > 10000


def test_sample_clock_short_cut_equals_the_loop():
    """faster_amd/csrc/fh_clock.hpp: the reference's sample clock `t = t + DC` (fillX, solverGurobi.cpp:131-135) evaluated a binade at a
    time — the same double and the same interval as the loop, for the reference's DC and for step sizes with every mantissa length
    (tests/cpp/test_clock.cpp; with and without fused multiply-adds: every intermediate is an exact integer, so contraction cannot matter)."""
    src = os.path.join(ROOT, "tests", "cpp", "test_clock.cpp")
    hdr = os.path.join(ROOT, "faster_amd", "csrc", "fh_clock.hpp")
    for flags in (["-ffp-contract=off"], ["-ffp-contract=fast", "-mfma"]):
        exe = os.path.join(ROOT, "tests", "cpp", "test_clock")
        subprocess.check_call(["g++", "-O2", "-std=c++14"] + flags + [src, "-o", exe])
        r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
        assert int(r.stdout.split()[0]) > 500000
    assert os.path.getsize(hdr) > 0


def test_divisions_by_multiply_high_are_exact():
    """faster_amd/csrc/fh_udiv.hpp: a cell number split by a launch-constant divisor with one multiply-high and one correction is the
    integer quotient for every n < 2^28 (tests/cpp/test_udiv.cpp: every divisor up to 200 000, powers of two and their neighbours,
    random divisors up to 2^27; multiples of the divisor and their neighbours), and the inverse computed with one double division is
    floor(2^32 / d) for d <= 2^20."""
    src = os.path.join(ROOT, "tests", "cpp", "test_udiv.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "test_udiv")
    subprocess.check_call(["g++", "-O2", "-std=c++14", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
    assert int(r.stdout.split()[0]) > 100000000
