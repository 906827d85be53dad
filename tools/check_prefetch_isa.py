#!/usr/bin/env python3
"""The line step of the headline kernel must issue its coefficient loads back to back: no `s_waitcnt vmcnt` and no branch
between the first and the last global load of a steady-state line step of jacobi_march5_kernel<double, 7, plain, ..., SYM>.
Round 4 met the failure mode: an innocent-looking select on the A_ii load made hipcc put a `s_waitcnt vmcnt(4)` and a branch
INSIDE the load sequence (the load's destination register was still the target of a load in flight) -- the prefetch drained every
line step, 218 -> 257 us, every result still bit-identical.  Usage: check_prefetch_isa.py [gla_march5.o]; exit 1 on a violation."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = os.environ.get("OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
KERNELS = [  # (mangled-name fragment <double, TS, MODE 0, RCP, HD, STATIC, SYM, FM, DGM>, 8-byte loads of a line step's own prefetch:
    #  x, b, the diagonals read, A_ii unless it is the centre coefficient (DGM 1; DGM 2 holds both bodies);
    #  line steps that may carry a conservative wait: one per loop header, i.e. per body)
    ("jacobi_march5_kernelIdLi7ELi0ELb1ELi2ELb0ELb1ELb1ELi2E", 6, 2), ("jacobi_march5_kernelIdLi6ELi0ELb1ELi2ELb0ELb1ELb1ELi2E", 6, 2),
    ("jacobi_march5_kernelIdLi7ELi0ELb1ELi2ELb0ELb1ELb1ELi1E", 6, 1), ("jacobi_march5_kernelIdLi7ELi0ELb1ELi2ELb0ELb1ELb1ELi0E", 7, 1),
    ("jacobi_march5_kernelIdLi7ELi0ELb1ELi2ELb0ELb1ELb0ELi0E", 7, 1), ("jacobi_march5_kernelIdLi7ELi0ELb1ELi2ELb0ELb0ELb0ELi0E", 8, 1),
]


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    b = os.path.join(tmp, os.path.basename(obj))
    subprocess.run(["cp", obj, b], check=True)
    subprocess.run([OBJDUMP, "--offloading", os.path.basename(b)], cwd=tmp, check=True, capture_output=True)
    co = [f for f in os.listdir(tmp) if "gfx950" in f]
    if not co:
        raise SystemExit(f"{obj}: no gfx950 code object")
    return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout


def check(text, frag, nloads, headers):
    m = re.search(r"^[0-9a-f]+ <([^>]*%s[^>]*)>:\n" % re.escape(frag), text, re.M)
    if not m:
        return f"{frag}: kernel not found"
    body = text[m.end():]
    nxt = re.search(r"^[0-9a-f]+ <_Z[^>]*>:\n", body, re.M)
    body = body[:nxt.start()] if nxt else body
    ops = [ln.split("//")[0].split() for ln in body.splitlines() if ln.startswith("\t")]
    ops = [o for o in ops if o]
    segs, seg = [], []
    for o in ops:
        if o[0] == "s_barrier":
            segs.append(seg)
            seg = []
        else:
            seg.append(o)
    segs.append(seg)
    if len(segs) < 8:
        return f"{frag}: only {len(segs)} barrier segments"
    bad, n_steps = {}, 0
    for k, sg in enumerate(segs[len(segs) // 2:-1], len(segs) // 2):      # the steady-state line steps (unrolled over the sets)
        loads = [i for i, o in enumerate(sg) if o[0].startswith("global_load_dwordx2")]
        if len(loads) < nloads:
            continue
        n_steps += 1
        head = loads[:nloads]                                      # the step's own prefetch (later loads belong to the rare generic path)
        for i in range(head[0], head[-1]):
            o = sg[i]
            if (o[0] == "s_waitcnt" and any(a.startswith("vmcnt") for a in o[1:])) or o[0].startswith("s_cbranch"):
                bad.setdefault(k, f"segment {k}: `{' '.join(o)}` between the loads of a line step")
    # (the unrolled copy at the loop header may carry one conservative wait from hipcc's counter merge -- it did in round 3's
    #  kernel too; a wait or a branch inside the load sequence of SEVERAL copies is the drain this tool is here to catch)
    if n_steps < 4:
        return f"{frag}: only {n_steps} steady-state line steps recognised"
    return f"{frag}: " + "; ".join(list(bad.values())[:3]) if len(bad) > headers else None


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gnn_applied_linear_algebra_amd", "csrc", "gla_march5.o")
    text = disassemble(obj)
    errs = [e for e in (check(text, k, nl, hd) for k, nl, hd in KERNELS) if e]
    for e in errs:
        print(e)
    print(f"{os.path.basename(obj)}: {len(KERNELS) - len(errs)} of {len(KERNELS)} march kernels issue their line-step loads back to back")
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
