"""Import harness for the UNMODIFIED reference modules (TEST INFRASTRUCTURE ONLY).

Only usable where /root/reference exists (the build container).  It is used by
  * oracle/gen_golden.py  -- to emit tests/golden/*.npz fixtures, and
  * tests/test_oracle_vs_reference.py -- to pin oracle/reference_port.py live.
Nothing in the product package, bench.py's timed path or the GPU tests imports it.

Recipe: third-party stand-ins from oracle/standins are put
on sys.path, bytecode writing is disabled (the reference tree is read-only), and
the reference's broken imports are repaired from the outside:
  * pytorch/JacobiGNN.py:49  imports GNNResidual from ChebyGNN (it lives in GNNResidual.py)
  * pytorch/VCycle.py:48-51, pytorch/DirectInterpGNN.py:180 import *_Meta modules that
    do not exist in the tree -> aliased to the real modules.
"""
import os
import sys
import types
import importlib

REFERENCE_DIR = "/root/reference/pytorch"
_STANDINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "standins")


def reference_available():
    return os.path.isdir(REFERENCE_DIR)


def load_reference():
    """Return a namespace with the reference's hot-path modules imported."""
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_DIR)
    sys.dont_write_bytecode = True
    for p in (_STANDINS, REFERENCE_DIR):
        if p not in sys.path:
            sys.path.insert(0, p)
    ns = types.SimpleNamespace()
    ns.UtilsGNN = importlib.import_module("UtilsGNN")
    ns.MatVecGNN = importlib.import_module("MatVecGNN")
    ns.GNNResidual = importlib.import_module("GNNResidual")
    ns.ChebyGNN = importlib.import_module("ChebyGNN")
    ns.ChebyGNN.GNNResidual = ns.GNNResidual.GNNResidual      # JacobiGNN.py:49 workaround
    ns.JacobiGNN = importlib.import_module("JacobiGNN")
    ns.PowerMethodGNN = importlib.import_module("PowerMethodGNN")
    ns.SOCClassicGNN = importlib.import_module("SOCClassicGNN")
    ns.DirectInterpGNN = importlib.import_module("DirectInterpGNN")
    ns.SOCSAGNN = importlib.import_module("SOCSAGNN")
    ns.MatrixWeightedNorm = importlib.import_module("MatrixWeightedNorm")
    for alias, mod in (("SOCClassicGNN_Meta", ns.SOCClassicGNN),
                       ("DirectInterpGNN_Meta", ns.DirectInterpGNN),
                       ("JacobiGNN_Meta", ns.JacobiGNN),
                       ("ChebyGNN_Meta", ns.ChebyGNN)):
        sys.modules.setdefault(alias, mod)
    from torch_geometric.nn import MetaLayer
    ns.MetaLayer = MetaLayer
    return ns


def run_vcycle_script(seed, torch):
    """Execute pytorch/VCycle.py (a whole-file script, VCycle.py:239-277) under a seed and
    return its module globals (A, b, x, functions, ...)."""
    import runpy
    import io
    import contextlib
    load_reference()
    torch.manual_seed(seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g = runpy.run_path(os.path.join(REFERENCE_DIR, "VCycle.py"), run_name="vcycle_ref")
    g["__stdout__"] = buf.getvalue()
    return g
