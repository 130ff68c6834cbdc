"""-m gpu: the multi-GPU path over RCCL (backend "nccl") on however many MI355X are visible -- batch shards per
rank, zhat all_gathered, shared-parameter gradient all_reduced to the global mean.  The same
code is covered on CPU by tests/test_dist_gloo.py (gloo, world_size 2).

With one visible GPU the tests that need a second DEVICE skip (and say so).  The tests named `*_world_size_one` do not:
RCCL initialises, all_gathers and all_reduces with a single rank too, so the whole `nccl` code path of the package and
of bench.py -- communicator set-up on the device, all_gather_into_tensor on RCCL's own stream beside the backward
launches, the scaled all_reduce of a shared parameter's gradient, bench.py's distributed branch with its C5 point --
executes on the one GPU a box has (round 6: until then the first 8-GPU run would have been that code's first execution
on any GPU).  They prove the plumbing, not the scaling."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import problems
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from qpth_amd import dist as qdist
    from qpth_amd.qp import QPFunction
    nB, n, m, q = 64 * world + 3, 30, 20, 4                    # uneven shards on purpose
    Q, p, G, h, A, b = problems.prof_qp(nB, n, m, q, seed=9)
    Qs = torch.tensor(Q[0], device=dev, requires_grad=True)    # shared by the batch: every rank holds it
    tp = torch.tensor(p, device=dev, requires_grad=True)
    tG, th, tA, tb = [torch.tensor(x, device=dev) for x in (G, h, A, b)]
    lo, hi = qdist.shard_bounds(nB, rank, world)
    z_local, z_full = qdist.solve_sharded(QPFunction(verbose=-1), Qs, tp, tG, th, tA, tb, nB)
    # the collective bench.py overlaps with the backward: all_gather_into_tensor (equal slices) / all_gather (ragged ones)
    # in flight on RCCL's stream while the backward launches are enqueued, waited for afterwards
    pending = qdist.gather_batch(z_local.detach(), nB, async_op=True)
    z_local.backward(torch.ones_like(z_local))
    z_async = pending.wait()
    assert torch.equal(z_async, z_full)
    dQ = qdist.reduce_shared_grad(Qs.grad, hi - lo, nB)
    dp_full = tp.grad.clone()
    dist.all_reduce(dp_full, op=dist.ReduceOp.SUM)
    # solver level: zhat, nu, lam, slacks of the global batch gathered over RCCL (what a caller keeps for autograd)
    _, (x_f, nu_f, lam_f, s_f) = qdist.forward_sharded(Qs.detach(), tp.detach(), tG, th, tA, tb, nB, verbose=-1)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), z=z_full.cpu().numpy(), dQ=dQ.cpu().numpy(), dp=dp_full.cpu().numpy(),
                 x=x_f.cpu().numpy(), nu=nu_f.cpu().numpy(), lam=lam_f.cpu().numpy(), s=s_f.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run_and_compare(world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    out = np.load(tmp_path / "out.npz")
    # the same global batch on one device, shared Q
    from qpth_amd.qp import QPFunction
    dev = torch.device("cuda:0")
    nB, n, m, q = 64 * world + 3, 30, 20, 4
    Q, p, G, h, A, b = problems.prof_qp(nB, n, m, q, seed=9)
    Qs = torch.tensor(Q[0], device=dev, requires_grad=True)
    tp = torch.tensor(p, device=dev, requires_grad=True)
    z = QPFunction(verbose=-1)(Qs, tp, *[torch.tensor(x, device=dev) for x in (G, h, A, b)])
    z.backward(torch.ones_like(z))
    assert rel_err(out["z"], z.detach().cpu().numpy()).max() < 1e-9
    assert np.abs(out["dQ"] - Qs.grad.cpu().numpy()).max() < 1e-9 * max(1.0, Qs.grad.abs().max().item())
    assert np.abs(out["dp"] - tp.grad.cpu().numpy()).max() < 1e-9 * max(1.0, tp.grad.abs().max().item())
    from qpth_amd.solvers.pdipm import batch as pdipm_b
    tG, th, tA, tb = [torch.tensor(x, device=dev) for x in (G, h, A, b)]
    Qe = Qs.detach().unsqueeze(0).expand(nB, n, n)
    Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(Qe, tG, tA)
    x1, nu1, lam1, s1 = pdipm_b.forward(Qe, tp.detach(), tG, th, tA, tb, Q_LU, S_LU, R, verbose=-1)
    assert rel_err(out["x"], x1.cpu().numpy()).max() < 1e-9 and rel_err(out["nu"], nu1.cpu().numpy()).max() < 1e-9
    assert rel_err(out["lam"], lam1.cpu().numpy()).max() < 1e-9 and np.abs(out["s"] - s1.cpu().numpy()).max() < 1e-9


def test_rccl_batch_sharding(tmp_path):
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("RCCL PATH NOT EXERCISED ACROSS DEVICES: %d GPU visible on this box (needs >= 2)" % world)
    _run_and_compare(world, tmp_path)


def test_rccl_path_world_size_one(tmp_path):
    """init_process_group("nccl", world_size=1) in a spawned process, then everything qpth_amd.dist does over RCCL:
    solve_sharded (all_gather of zhat), gather_batch(async_op=True) beside the backward, reduce_shared_grad,
    forward_sharded (zhat, nu, lam, slacks gathered); the result must be the plain single-process one.  Never skips."""
    _run_and_compare(1, tmp_path)


def _bench_line(cmd, env, timeout):
    import json
    import subprocess
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_distributed_branch_world_size_one(tmp_path):
    """bench.py --gpus 1 with QPX_FORCE_DIST=1: the branch the driver's N > 1 runs take (RCCL process group, zhat
    all_gathered beside the backward inside the timed region, MAX over ranks, the C5 strong-scaling point under `extra`)
    on the one GPU of this box.  The record is kept under gpurun_out/ when that directory exists."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", QPX_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    d = _bench_line([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                    env, 900)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["global_batch"] == 512 and d["value"] > 0
    assert "all_gathered over RCCL" in d["config"]["workload"]
    x = d["extra"]["c5_strong_scaling"]
    assert x["global_batch"] == 65536 and x["per_gpu_batch"] == 65536 and x["n_gpus"] == 1 and x["value"] > 0
    out_dir = os.path.join(root, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "nccl_world1_bench.json"), "w") as f:
            f.write(json.dumps(d) + "\n")


def test_bench_strong_scaling_contract_under_torchrun(tmp_path):
    """bench.py --config c5 --gpus N under torch.distributed.run, as the driver launches it: one JSON line from rank 0
    with the strong-scaling contract (fixed global batch of 65 536 sharded over the N ranks, the gather inside the
    timed region).  Needs >= 2 visible GPUs."""
    import json
    import subprocess
    import sys
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("RCCL PATH NOT EXERCISED: %d GPU visible on this box (needs >= 2)" % world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--config", "c5", "--gpus", str(world),
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, check=True, env=env).stdout
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["config"]["global_batch"] == 65536
    assert d["value"] > 0 and d["steps"] == 3


def test_default_bench_line_carries_both_readings_of_the_metric_at_n_gpus(tmp_path):
    """The DEFAULT bench.py --gpus N (what the driver runs for the scaling curve): weak C2 as `value` and, under
    `extra.c5_strong_scaling`, the north star's fixed global batch of 65 536 over the same N ranks (round 4).  Needs >= 2
    visible GPUs."""
    import json
    import subprocess
    import sys
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("RCCL PATH NOT EXERCISED: %d GPU visible on this box (needs >= 2)" % world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "3",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, check=True, env=env).stdout
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["global_batch"] == 512 * world
    x = d["extra"]["c5_strong_scaling"]
    assert x["global_batch"] == 65536 and x["n_gpus"] == world and x["scaling"] == "strong" and x["value"] > 0


def test_tensors_on_a_device_that_is_not_current():
    """One process, two devices: the QP lives on cuda:1 while cuda:0 is the current device.  Every launch of the library
    -- the pre-factorisation, the loop, the backward and the batch contraction of the shared-parameter gradients -- has
    to run on the tensors' device and stream (KKTFactors._knob makes it current around each call); the answer is the one
    cuda:0 gives.  Needs >= 2 visible GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("MULTI-DEVICE PATH NOT EXERCISED: %d GPU visible on this box (needs >= 2)" % torch.cuda.device_count())
    from qpth_amd.qp import QPFunction
    Q, p, G, h, A, b = problems.prof_qp(48, 100, 50, 10, seed=4)
    outs = []
    for devi in (0, 1):
        dev = torch.device("cuda", devi)
        torch.cuda.set_device(0)                       # the current device stays cuda:0 in both passes
        Qs = torch.tensor(Q[0], device=dev, requires_grad=True)          # shared by the batch: qpx_batch_outer runs
        tp = torch.tensor(p, device=dev, requires_grad=True)
        tG, th, tA, tb = [torch.tensor(x, device=dev) for x in (G, h, A, b)]
        z = QPFunction(verbose=-1)(Qs, tp, tG[0], th, tA[0], tb)
        z.backward(torch.ones_like(z))
        torch.cuda.synchronize(dev)
        assert z.device == dev and Qs.grad.device == dev and tp.grad.device == dev
        outs.append((z.detach().cpu().numpy(), Qs.grad.cpu().numpy(), tp.grad.cpu().numpy()))
    for a_, b_ in zip(outs[0], outs[1]):
        assert rel_err(a_, b_).max() < 1e-9 or np.abs(a_ - b_).max() < 1e-12
