"""Deterministic stand-in for pyamg's randomised CLJP splitting: every other vertex
is a C point, the rule the reference's MATLAB tests use (matlab/test_vcycle.m:65-67,
matlab/test_direct_interpolation.m:63-65).  Parity for true CLJP is unpinned
(SURVEY.md section 8c); the splitting is an *input* of DirectInterpGNN."""
import numpy as np


def CLJP(S, color=False):
    s = np.zeros(S.shape[0], dtype=np.int64)
    s[0::2] = 1
    return s
