"""gnn_applied_linear_algebra_amd -- MI355X-native kernels behind the operator surface of
sandialabs/gnn-applied-linear-algebra's sparse message-passing hot path.

Module names mirror the reference's flat `pytorch/` directory (UtilsGNN, MatVecGNN, GNNResidual, JacobiGNN,
ChebyGNN, PowerMethodGNN, SOCClassicGNN, DirectInterpGNN, VCycle, SOCSAGNN, MatrixWeightedNorm).  The sparse arithmetic
of the path -- every SpMV / SpMM / aggregation, strength of connection, interpolation weights, C/F splitting, coalesce,
SpGEMM, halo exchange -- runs in libgla_hip.so (csrc/, C ABI in include/gla.h); torch ops remain for tensor glue (cat,
stack, slicing), the dense LU of a coarsest AMG level and the O(n) index vectors of the distributed setup.  There is
no CPU fallback.
"""
from . import _lib  # noqa: F401
from .meta import MetaLayer  # noqa: F401
from .plan import CsrPlan, plan_for, col_dot, col_scale_rsqrt, invalidate  # noqa: F401
from . import sparse_ops  # noqa: F401
from . import autograd  # noqa: F401
from . import (UtilsGNN, MatVecGNN, GNNResidual, JacobiGNN, ChebyGNN, PowerMethodGNN, SOCClassicGNN,  # noqa: F401
               DirectInterpGNN, SOCSAGNN, MatrixWeightedNorm, VCycle, TrainableJacobiDiag)

__all__ = ["MetaLayer", "CsrPlan", "plan_for", "invalidate", "UtilsGNN", "MatVecGNN", "GNNResidual", "JacobiGNN", "ChebyGNN",
           "PowerMethodGNN", "SOCClassicGNN", "DirectInterpGNN", "SOCSAGNN", "MatrixWeightedNorm", "VCycle", "TrainableJacobiDiag", "autograd"]
