"""qpth_amd -- MI355X-native drop-in for the hot path of locuslab/qpth: the batched
differentiable dense QP solver `QPFunction(...)(Q, p, G, h, A, b)` (qpth/qp.py:18-183) on
hand-written gfx950 HIP kernels (one QP per workgroup, KKT blocks in LDS).

    from qpth_amd.qp import QPFunction, QPSolvers
    zhat = QPFunction(verbose=-1)(Q, p, G, h, A, b)      # tensors on a HIP device
"""
from . import qp, solvers, util  # noqa: F401
from .qp import QPFunction, QPSolvers  # noqa: F401
from .kkt import set_stall_policy  # noqa: F401

__version__ = "0.1.0"
