"""The product's field arithmetic header compiles for the host too (verify.cpp, the challenger): its signed-lane extension product,
inverse and scaling (round 4) are checked there against 128-bit integer arithmetic -- no GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_signed_lane_field_arithmetic_matches_wide_integers(tmp_path):
    exe = tmp_path / "field_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "lurk_amd", "csrc"), "-o", str(exe), os.path.join(ROOT, "tests", "field_check.cpp")],
                   check=True, capture_output=True, timeout=300)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("bad=0"), out.stdout[-500:]
