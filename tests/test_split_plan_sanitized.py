"""The split prover's index arithmetic (lurk_amd/csrc/split_plan.h: host only) as a property check under AddressSanitizer + UBSan:
tests/cpp/split_plan_check.cpp draws random commitments -- ragged widths, the three kinds of row sources, next-row copies, dead
column runs, matrices below the cut -- for G = 2 .. 64 ranks and checks the G plans against each other (block sizes, buffer bounds,
every cell of every slab and row block written exactly once).  GPU AddressSanitizer does not exist on this pool; this is the part of
the split that is pure host arithmetic and decides where every word goes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_random_plans_under_sanitizers(tmp_path):
    exe = tmp_path / "split_plan_check"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", str(exe),
                            os.path.join(ROOT, "tests", "cpp", "split_plan_check.cpp")], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([str(exe), "25"], capture_output=True, text=True, env=env, timeout=900)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert run.stdout.startswith("ok: 3150 plans"), run.stdout
