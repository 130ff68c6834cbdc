"""Helpers mirroring qpth/util.py:18-59 (same names, argument meaning and error behaviour)."""
import numpy as np
import torch


def print_header(msg):
    print('===>', msg)


def to_np(t):
    if t is None:
        return None
    if t.nelement() == 0:
        return np.array([])
    return t.detach().cpu().numpy()


def bger(x, y):
    """batched outer product (B,a),(B,b) -> (B,a,b)   [util.py:18-19]"""
    return x.unsqueeze(2) * y.unsqueeze(1)


def get_sizes(G, A=None):
    """-> (nineq, nz, neq, nBatch)   [util.py:22-33]"""
    if G.dim() == 2:
        nineq, nz = G.size()
        nBatch = 1
    elif G.dim() == 3:
        nBatch, nineq, nz = G.size()
    else:
        raise RuntimeError("Unexpected number of dimensions.")
    neq = None
    if A is not None:
        neq = A.size(1) if A.nelement() > 0 else 0
    return nineq, nz, neq, nBatch


def bdiag(d):
    """(B,sz) -> (B,sz,sz) batched diagonal   [util.py:36-41]"""
    return torch.diag_embed(d)


def expandParam(X, nBatch, nDim):
    """Broadcast an un-batched parameter over the batch as a stride-0 view; returns
    (X, was_expanded).  0-dim, already batched and empty tensors pass through  [util.py:44-50]."""
    if X.ndimension() in (0, nDim) or X.nelement() == 0:
        return X, False
    if X.ndimension() == nDim - 1:
        return X.unsqueeze(0).expand(*([nBatch] + list(X.size()))), True
    raise RuntimeError("Unexpected number of dimensions.")


def extract_nBatch(Q, p, G, h, A, b):
    """The first parameter that carries a batch dimension defines nBatch, else 1  [util.py:53-59]."""
    dims = [3, 2, 3, 2, 3, 2]
    for param, dim in zip([Q, p, G, h, A, b], dims):
        if param.ndimension() == dim:
            return param.size(0)
    return 1
