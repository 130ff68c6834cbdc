"""Seeded problem generators shared by the tests, bench.py and the golden-vector script.

They restate the generators of the reference's own tests / timing scripts (numpy only, so
they produce bit-identical inputs wherever they run):

  prof_qp ......... /root/reference/prof-linear.py:64-75 (prof_instance), vectorised;
                    the workload of BASELINE.json (C1..C5)
  grads_qp ........ /root/reference/test.py:42-66 (get_grads): B=1, nz=10, npr.seed(1),
                    generation order L, G, z0, s0, A, p, truez
  kkt_problem ..... /root/reference/test.py:190-219 (get_kkt_problem); the reference leaves
                    torch.randn unseeded -- here it is seeded numpy
"""
import numpy as np
import numpy.random as npr


def prof_qp(B, n, m, q, seed=0, dtype=np.float64):
    npr.seed(seed)
    L = npr.rand(B, n, n)
    Q = np.matmul(L, L.transpose((0, 2, 1))) + 1e-3 * np.eye(n, n)
    G = npr.randn(B, m, n)
    z0 = npr.randn(B, n)
    s0 = npr.rand(B, m)
    p = npr.randn(B, n)
    h = np.matmul(G, np.expand_dims(z0, axis=(2))).squeeze(2) + s0
    A = npr.randn(B, q, n)
    b = np.matmul(A, np.expand_dims(z0, axis=(2))).squeeze(2)
    Q, p, G, h, A, b = [np.ascontiguousarray(x.astype(dtype)) for x in (Q, p, G, h, A, b)]
    if q == 0:
        A = np.zeros(0, dtype)
        b = np.zeros(0, dtype)
    return Q, p, G, h, A, b


# the five (neq, nineq, Qscale, Gscale, Ascale) settings of test.py:100-102,117-119,137-139,
# 154-156,174-176
GRADS_CASES = {
    "dl_dp": (2, 3, 100.0, 100.0, 100.0),
    "dl_dG": (0, 3, 1.0, 1.0, 1.0),
    "dl_dh": (0, 3, 1.0, 1.0, 1.0),
    "dl_dA": (3, 1, 100.0, 100.0, 100.0),
    "dl_db": (3, 1, 100.0, 100.0, 100.0),
}


def grads_qp(nz=10, neq=1, nineq=3, Qscale=1.0, Gscale=1.0, Ascale=1.0):
    npr.seed(1)
    L = np.random.randn(nz, nz)
    Q = Qscale * L.dot(L.T)
    G = Gscale * npr.randn(nineq, nz)
    z0 = npr.randn(nz)
    s0 = npr.rand(nineq)
    h = G.dot(z0) + s0
    A = Ascale * npr.randn(neq, nz)
    b = A.dot(z0)
    p = npr.randn(1, nz)
    truez = npr.randn(1, nz)
    Q, p, G, h, A, b, truez = [x.astype(np.float64) for x in [Q, p, G, h, A, b, truez]]
    return Q, p, G, h, A, b, truez


def kkt_problem(seed=0, nBatch=2, nx=5, nineq=4, neq=3):
    r = npr.RandomState(seed)
    Q = r.randn(nx, nx)
    Q = Q.dot(Q.T)
    p = r.randn(nx)
    G = r.randn(nBatch, nineq, nx)
    h = np.zeros((nBatch, nineq))
    A = r.randn(neq, nx)
    b = r.randn(neq)
    d = r.rand(nBatch, nineq)
    rx = r.rand(nBatch, nx)
    rs = r.rand(nBatch, nineq)
    rz = r.rand(nBatch, nineq)
    ry = r.rand(nBatch, neq)
    return Q, p, G, h, A, b, d, rx, rs, rz, ry


def random_dense_qp(B, n, m, q, seed, dtype=np.float64, well_conditioned=True):
    """A better-conditioned family (Q = L L^T / n + I) for edge-case and f32 tests."""
    r = npr.RandomState(seed)
    L = r.randn(B, n, n)
    Q = np.matmul(L, L.transpose((0, 2, 1))) / n + (1.0 if well_conditioned else 1e-3) * np.eye(n)
    G = r.randn(B, m, n)
    z0 = r.randn(B, n)
    s0 = r.rand(B, m) + 0.1
    p = r.randn(B, n)
    h = np.einsum("bmn,bn->bm", G, z0) + s0
    A = r.randn(B, q, n)
    b = np.einsum("bqn,bn->bq", A, z0)
    Q, p, G, h, A, b = [np.ascontiguousarray(x.astype(dtype)) for x in (Q, p, G, h, A, b)]
    if q == 0:
        A = np.zeros(0, dtype)
        b = np.zeros(0, dtype)
    return Q, p, G, h, A, b
