"""The 29-bit-limb Fq of the MSM kernels (zkos-monorepo_amd/csrc/field29.hip.hpp) against the 8 x 32-bit arithmetic, on the CPU:
both compile for the host, so products, squares, lazy subtractions and chains of mixed additions (with doublings, opposite
points and the identity spliced in) are compared value for value without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_field29_matches_field32(tmp_path):
    exe = tmp_path / "field29_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "native", "field29_check.cpp")],
                   check=True)
    out = subprocess.run([str(exe), "4000"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok 4000 products"), out


def test_ntt_lazy_values_match_reduced_arithmetic(tmp_path):
    """The NTT butterflies' lazy arithmetic (zkos-monorepo_amd/csrc/ntt_lazy.hip.hpp): 32 x by shifts against the product -- random
    values, the ends of the range and every value next to a multiple of p / 32 -- and chains of butterflies on values in
    [0, 4p) against the reduced arithmetic, every intermediate inside its bound."""
    exe = tmp_path / "ntt_lazy_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "native", "ntt_lazy_check.cpp")],
                   check=True)
    out = subprocess.run([str(exe), "20000"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok ") and "240000 lazy butterflies" in out, out
