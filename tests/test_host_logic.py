"""Host-side logic above the C ABI (no kernels run): the helpers that mirror qpth/util.py, the batch
sharding arithmetic, the loud refusal of CPU tensors, argument checking of the factor object."""
import pytest
import torch

from qpth_amd import _lib, dist, util
from qpth_amd.kkt import KKTFactors, default_stall_policy
from qpth_amd.qp import QPFunction


def test_util_helpers_follow_the_reference():
    x, y = torch.randn(3, 4), torch.randn(3, 5)
    assert torch.allclose(util.bger(x, y), torch.einsum("bi,bj->bij", x, y))            # util.py:18-19
    G, A = torch.zeros(7, 6, 5), torch.zeros(7, 2, 5)
    assert util.get_sizes(G, A) == (6, 5, 2, 7)                                           # util.py:22-33
    assert util.get_sizes(G[0], torch.empty(0)) == (6, 5, 0, 1)
    d = torch.randn(2, 3)
    assert torch.equal(util.bdiag(d)[1], torch.diag(d[1]))                                # util.py:36-41
    p = torch.randn(5)
    pe, expanded = util.expandParam(p, 7, 2)                                              # util.py:44-50
    assert expanded and pe.shape == (7, 5) and pe.stride(0) == 0
    assert util.expandParam(torch.randn(7, 5), 7, 2)[1] is False
    assert util.expandParam(torch.empty(0), 7, 2)[1] is False
    with pytest.raises(RuntimeError, match="Unexpected number of dimensions"):
        util.expandParam(torch.randn(2, 3, 4, 5), 7, 2)
    e = torch.empty(0)
    assert util.extract_nBatch(torch.randn(5, 5), torch.randn(9, 5), G[0], torch.randn(6), e, e) == 9   # util.py:53-59
    assert util.extract_nBatch(torch.randn(5, 5), torch.randn(5), G[0], torch.randn(6), e, e) == 1


@pytest.mark.parametrize("nBatch,world", [(512, 8), (65536, 8), (5, 2), (3, 4), (1, 8)])
def test_shards_partition_the_batch(nBatch, world):
    bounds = [dist.shard_bounds(nBatch, r, world) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == nBatch
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    sizes = [hi - lo for lo, hi in bounds]
    assert max(sizes) - min(sizes) <= 1
    full = torch.arange(nBatch * 2.0).reshape(nBatch, 2)
    shared = torch.randn(2)
    parts = [dist.shard_params([shared, full, shared, full, torch.empty(0), torch.empty(0)], nBatch, r, world,
                               ndims=(2, 2, 2, 2, 3, 2)) for r in range(world)]
    assert torch.equal(torch.cat([p[1] for p in parts]), full)
    assert all(p[0] is shared for p in parts)


def test_cpu_tensors_are_refused_loudly():
    """There is no CPU path in the product: the error must say so (the emulator is installed by tests only)."""
    assert _lib._TEST_BACKEND is None
    Q = torch.eye(3, dtype=torch.float64).unsqueeze(0)
    G, h, p = torch.ones(1, 2, 3, dtype=torch.float64), torch.ones(1, 2, dtype=torch.float64), torch.ones(1, 3, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        QPFunction(verbose=-1)(Q, p, G, h, torch.empty(0, dtype=torch.float64), torch.empty(0, dtype=torch.float64))


def test_inconsistent_sizes_are_an_error_before_any_launch():
    Q, G = torch.eye(3).unsqueeze(0), torch.ones(1, 2, 4)
    with pytest.raises(RuntimeError, match="inconsistent QP sizes"):
        KKTFactors.build(Q, G, torch.empty(0))


def test_default_stall_policy_is_the_reference_counter_for_one_qp():
    assert default_stall_policy(1) == _lib.STALL_REFERENCE and default_stall_policy(2) == _lib.STALL_FLOOR


def test_sharded_solves_refuse_what_cannot_be_sharded():
    """the same error on every rank, before any collective: a batch smaller than the world, nothing batched"""
    import torch
    from qpth_amd import dist as qdist
    Q, p = torch.eye(3), torch.zeros(2, 3)
    e = torch.empty(0)
    with pytest.raises(RuntimeError, match="cannot be sharded"):
        qdist.check_shardable([Q, p, torch.ones(1, 3), torch.ones(1), e, e], 2, 4)
    with pytest.raises(RuntimeError, match="nothing to shard"):
        qdist.check_shardable([Q, torch.zeros(3), torch.ones(1, 3), torch.ones(1), e, e], 8, 2)
    qdist.check_shardable([Q, torch.zeros(8, 3), torch.ones(1, 3), torch.ones(1), e, e], 8, 2)


def test_pinned_status_buffers_are_never_handed_out_twice():
    """qpth_amd.kkt._PinnedPool (round 6; until then a process-global ring with an unsynchronised index that silently
    reused a buffer after 16 outstanding builds): a buffer belongs to one owner from take() to give(); a buffer given up
    while its copy is still in flight comes back only once its event has completed; an event that cannot be asked
    (recorded inside a stream capture) drops the buffer."""
    import torch
    from qpth_amd import kkt

    class Ev:
        def __init__(self, done):
            self.done = done

        def query(self):
            if self.done is None:
                raise RuntimeError("captured event")
            return self.done

    pool = kkt._PinnedPool()
    dev = torch.device("cuda", 0)
    made = []
    real = torch.Tensor.pin_memory
    torch.Tensor.pin_memory = lambda t: (made.append(t) or t)          # (no HIP runtime here: a pinned buffer is just a tensor)
    try:
        k1, a = pool.take(dev, 8)
        k2, b = pool.take(dev, 8)
        assert a is not b and len(made) == 2                           # two outstanding owners: two buffers
        pool.give(k1, a)                                               # read and returned
        k3, c = pool.take(dev, 8)
        assert c is a and len(made) == 2                               # reused only after it was given back
        ev = Ev(False)
        pool.give(k2, b, ev)                                           # owner died with the copy in flight
        k4, d = pool.take(dev, 8)
        assert d is not b and len(made) == 3                           # not handed out while in flight
        ev.done = True
        k5, e = pool.take(dev, 8)
        assert e is b                                                  # ... and back once the event has completed
        pool.give(k5, e, Ev(None))                                     # an event that cannot be asked: dropped
        k6, f = pool.take(dev, 8)
        assert f is not e and len(made) == 4
        assert pool.take(dev, 16)[1].numel() == 16                     # another size: another list
    finally:
        torch.Tensor.pin_memory = real


def test_unpack_kkt_and_kkt_resid_reg_mirror_the_reference():
    """batch.py:216-241: the two helpers of the reference's module surface that have no kernel of their own (the kernels
    that refine form the residual themselves).  A dense solve of the regularised KKT system has residual ~0 under
    kkt_resid_reg; a perturbed one has exactly the perturbation's image."""
    import numpy as np
    import torch
    from qpth_amd.solvers.pdipm import batch as pb
    rng = np.random.RandomState(3)
    B, nz, nineq, neq, eps = 3, 6, 4, 2, 1e-3
    L = rng.rand(B, nz, nz)
    Q = L @ L.transpose(0, 2, 1) + 0.1 * np.eye(nz)
    d = rng.rand(B, nineq) + 0.5
    D = np.stack([np.diag(x) for x in d])
    G, A = rng.randn(B, nineq, nz), rng.randn(B, neq, nz)
    rx, rs, rz, ry = rng.randn(B, nz), rng.randn(B, nineq), rng.randn(B, nineq), rng.randn(B, neq)
    sols = []
    for i in range(B):
        K = np.zeros((nz + 2 * nineq + neq,) * 2)
        K[:nz, :nz] = Q[i]; K[:nz, nz + nineq:nz + 2 * nineq] = G[i].T; K[:nz, nz + 2 * nineq:] = A[i].T
        K[nz:nz + nineq, nz:nz + nineq] = D[i]; K[nz:nz + nineq, nz + nineq:nz + 2 * nineq] = np.eye(nineq)
        K[nz + nineq:nz + 2 * nineq, :nz] = G[i]; K[nz + nineq:nz + 2 * nineq, nz:nz + nineq] = np.eye(nineq)
        K[nz + nineq:nz + 2 * nineq, nz + nineq:nz + 2 * nineq] = -eps * np.eye(nineq)
        K[nz + 2 * nineq:, :nz] = A[i]; K[nz + 2 * nineq:, nz + 2 * nineq:] = -eps * np.eye(neq)
        sols.append(np.linalg.solve(K, -np.concatenate([rx[i], rs[i], rz[i], ry[i]])))
    v = torch.tensor(np.stack(sols))
    dx, ds, dz, dy = pb.unpack_kkt(v, nz, nineq, neq)
    assert dx.shape == (B, nz) and ds.shape == (B, nineq) and dz.shape == (B, nineq) and dy.shape == (B, neq)
    t = lambda a: torch.tensor(a)
    res = pb.kkt_resid_reg(t(Q), t(D), t(G), t(A), eps, dx, ds, dz, dy, t(rx), t(rs), t(rz), t(ry))
    assert max(float(r.abs().max()) for r in res) < 1e-10
    bump = torch.zeros_like(dz); bump[:, 0] = 1.0
    res2 = pb.kkt_resid_reg(t(Q), t(D), t(G), t(A), eps, dx, ds, dz + bump, dy, t(rx), t(rs), t(rz), t(ry))
    assert torch.allclose(res2[0], t(G)[:, 0, :], atol=1e-10) and torch.allclose(res2[1], bump, atol=1e-10)
    assert torch.allclose(res2[2], -eps * bump, atol=1e-10)
    rx3, rs3, rz3, _ = pb.kkt_resid_reg(t(Q), t(D), t(G), None, eps, dx, ds, dz, None, t(rx), t(rs), t(rz), None)
    assert _ is None and rx3.shape == (B, nz)
