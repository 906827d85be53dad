"""bench.py's launcher: `python bench.py --gpus N` must work as typed -- the parent starts N fresh rank processes and
itself never imports torch or touches the GPU (a process that initialised HIP must not be replaced or forked)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=120)


def test_parent_spawns_one_process_per_gpu_without_touching_torch():
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--launch-dry-run"])
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout)
    assert plan["torch_imported_in_parent"] is False
    kids = plan["children"]
    assert len(kids) == 8
    ports = {k["env"]["MASTER_PORT"] for k in kids}
    assert len(ports) == 1
    for r_, k in enumerate(kids):
        e = k["env"]
        assert e["RANK"] == str(r_) and e["LOCAL_RANK"] == str(r_) and e["WORLD_SIZE"] == "8"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert k["cmd"][1].endswith("bench.py") and k["cmd"][2:] == ["--gpus", "8", "--steps", "3", "--warmup", "1",
                                                                      "--launch-dry-run"]


def test_parent_relays_child_failure_as_nonzero_exit():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine WITHOUT a GPU (the ranks are expected to fail)")
    """No GPU in this container: the ranks fail at the first device call; the parent must report that, not rc 0."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--single-device",
              "--child-timeout", "90"])
    assert r.returncode != 0
    assert "Traceback" in r.stderr or "Error" in r.stderr


def test_rank_processes_write_the_json_line_to_the_saved_stdout():
    """A rank points fd 1 at stderr for its lifetime (RCCL / gloo print banners on stdout), so the JSON line must go through
    emit() -- a plain print() in a rank would land on stderr and the driver would see no line at all."""
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def single_gpu("):]
    assert "print(json.dumps" not in body, "rank code must call emit(), not print()"
    assert len(re.findall(r"\bemit\(out\)", body)) >= 2            # the N = 1 line and the N > 1 line


def test_layout_byte_accounting_of_the_bench_line():
    """bench.py's per-pass byte accounting (the line must say what the kernels move): 5-point plans, symmetric
    value set + full stencil + A_ii = the diagonal -> the module step moves 64 (AoS first pass) + 48 + 48 bytes per row; an
    unsymmetric plan 5 diagonals and masks; an A_ii array 8 more per plain pass."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    info = types.SimpleNamespace(dia=5, dia_canon5=1, dia_sym=3, dia_full=1)
    assert [rb for _, _, rb in bench.step_layout(info, (7, 7, 6), True, True)] == [64, 48, 48]
    assert [k for k, _, _ in bench.step_layout(info, (7, 7, 6), True, True)] == ["aos", "plain", "plain"]
    assert [rb for _, _, rb in bench.step_layout(info, (7, 7, 6), False, True)] == [72, 56, 56]       # A_ii streamed, copies of A_ii and b written
    info_u = types.SimpleNamespace(dia=5, dia_canon5=1, dia_sym=0, dia_full=0)
    assert [rb for _, _, rb in bench.step_layout(info_u, (7, 7, 6), False, True)] == [90, 74, 74]
    assert bench.pass_row_bytes(5, 2, True, "plain", False) == 48 and bench.pass_row_bytes(5, 0, False, "plain", True) == 74
