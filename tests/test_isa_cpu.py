"""The built gfx950 code objects carry no trace of the hipcc miscompile met in round 3 (profiles/r03/marchk_miscompile.txt):
an exec-masked region with nothing in it, i.e. the masked move of a `uniform || per_lane ? value : 0` select dropped.  Runs
tools/check_isa.sh on the objects csrc/build.sh left (no GPU needed: llvm-objdump on the embedded code objects)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_empty_exec_masked_regions():
    objs = sorted(glob.glob(os.path.join(ROOT, "gnn_applied_linear_algebra_amd", "csrc", "gla_*.o")))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no built objects (run __graft_entry__.build()) or no llvm-objdump")
    r = subprocess.run([os.path.join(ROOT, "tools", "check_isa.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(": 0 empty exec-masked regions") == len(objs), r.stdout


def test_headline_kernel_issues_its_line_step_loads_back_to_back():
    """tools/check_prefetch_isa.py: no `s_waitcnt vmcnt` / branch inside the load sequence of the march kernel's line steps (the
    register-allocation accident of round 4 that cost 18 % with every result still bit-identical: DESIGN 3.2d)."""
    obj = os.path.join(ROOT, "gnn_applied_linear_algebra_amd", "csrc", "gla_march5.o")
    if not os.path.exists(obj) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no built object (run __graft_entry__.build()) or no llvm-objdump")
    r = subprocess.run([os.path.join(ROOT, "tools", "check_prefetch_isa.py"), obj], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
