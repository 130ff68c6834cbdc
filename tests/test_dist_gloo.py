"""Multi-process path on CPU: world_size 2, gloo, 127.0.0.1 -- the batch shards across ranks with
no data-path collective; zhat is all-gathered and the gradient of a batch-shared parameter is
all-reduced to the reference's global `.mean(0)` (qp.py:159-177).  The kernels run in the
host-thread emulator here; on the GPU box the same code runs with backend "nccl" (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import problems
from conftest import load_golden, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu.harness import emulated
    from qpth_amd import dist as qdist
    from qpth_amd.qp import QPFunction
    g = load_golden("broadcast_b5_n12_m9_q3")       # Q, G, A shared by the batch; p, h, b batched
    nB = 5
    tq = []
    for k in ("Q", "p", "G", "h", "A", "b"):
        t = torch.tensor(g[k])
        tq.append(t)
    lo, hi = qdist.shard_bounds(nB, rank, world)
    Q = tq[0].clone().requires_grad_(True)          # shared: every rank holds the whole parameter
    p = tq[1].clone().requires_grad_(True)          # batched: each rank touches rows [lo, hi)
    with emulated(64):
        z_local, z_full = qdist.solve_sharded(QPFunction(verbose=-1), Q, p, tq[2], tq[3], tq[4], tq[5], nB)
        z_local.backward(torch.tensor(g["dl_dz"][lo:hi]))
    dQ = qdist.reduce_shared_grad(Q.grad, hi - lo, nB)
    # equal slices take the one-collective path of gather_batch (all_gather_into_tensor), unequal ones (nB = 5 above) the padded one
    even = torch.arange(12, dtype=torch.float64).reshape(3, 4) + 100.0 * rank
    full = qdist.gather_batch(even, 3 * world)
    want = torch.cat([torch.arange(12, dtype=torch.float64).reshape(3, 4) + 100.0 * r for r in range(world)], 0)
    assert torch.equal(full, want)
    # the asynchronous form (bench.py: the all_gather of zhat runs beside the backward launches): same tensor after .wait(),
    # for equal and for unequal slices
    pend = qdist.gather_batch(even, 3 * world, async_op=True)
    assert torch.equal(pend.wait(), want)
    odd = torch.full((hi - lo, 2), float(rank), dtype=torch.float64)
    got = qdist.gather_batch(odd, nB, async_op=True).wait()
    assert got.shape == (nB, 2) and torch.equal(got[lo:hi], odd)
    dp_full = p.grad.clone()                        # zero outside this rank's rows
    dist.all_reduce(dp_full, op=dist.ReduceOp.SUM)
    # the solver-level path: zhat, nu, lam, slacks of the whole batch on every rank
    with emulated(64):
        _, (x_f, nu_f, lam_f, s_f) = qdist.forward_sharded(tq[0], tq[1], tq[2], tq[3], tq[4], tq[5], nB, verbose=-1)
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), z=z_full.numpy(), dQ=dQ.numpy(), dp=dp_full.numpy(),
                 x=x_f.numpy(), nu=nu_f.numpy(), lam=lam_f.numpy(), s=s_f.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batch_sharding(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = np.load(tmp_path / "out.npz")
    g = load_golden("broadcast_b5_n12_m9_q3")
    assert rel_err(out["z"], g["zhat"]).max() < 1e-6
    assert np.abs(out["dQ"] - g["dQ"]).max() < 1e-6 * max(1.0, np.abs(g["dQ"]).max())
    assert np.abs(out["dp"] - g["dp"]).max() < 1e-6 * max(1.0, np.abs(g["dp"]).max())
    assert rel_err(out["x"], g["zhat"]).max() < 1e-6 and rel_err(out["nu"], g["nu"]).max() < 1e-6
    assert rel_err(out["lam"], g["lam"]).max() < 1e-5 and np.abs(out["s"] - g["slacks"]).max() < 1e-6


def test_shard_bounds_cover_the_batch():
    from qpth_amd import dist as qdist
    for nB in (1, 5, 512, 65536):
        for world in (1, 2, 3, 8):
            spans = [qdist.shard_bounds(nB, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nB
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
