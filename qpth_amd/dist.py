"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed with
backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

QPs never interact, so the data path needs no collective: every rank solves
its contiguous slice of the batch.  Communication exists only at the edges:
  * all_gather of zhat (and lam/nu/slacks on request) when a caller wants full-batch outputs;
  * all_reduce(SUM) of the gradients of batch-SHARED parameters, scaled so that the result is
    the reference's `.mean(0)` over the GLOBAL batch (qpth/qp.py:159-177).
"""
import torch
import torch.distributed as dist


def shard_bounds(nBatch, rank, world):
    """contiguous slice [lo, hi) of a batch of nBatch owned by `rank`"""
    base, rem = divmod(nBatch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def check_shardable(params, nBatch, world, ndims=(3, 2, 3, 2, 3, 2)):
    """Refuse, identically on EVERY rank (the decision depends on nBatch, world and the tensors' ranks only, never on the
    rank's own slice), what cannot be sharded -- before any rank enters a collective the others would then hang in:
    a batch smaller than the world (some rank's slice would be empty) and a batch size that no parameter carries
    (all of Q, p, G, h, A, b un-batched: every rank would solve the same single QP)."""
    if nBatch < world:
        raise RuntimeError("qpth_amd.dist: a batch of %d cannot be sharded over %d ranks (empty slices)" % (nBatch, world))
    batched = [X for X, nd in zip(params, ndims) if X.nelement() > 0 and X.dim() == nd]
    if not batched:
        raise RuntimeError("qpth_amd.dist: no parameter is batched; there is nothing to shard (nBatch = %d)" % nBatch)
    for X in batched:
        if X.size(0) != nBatch:
            raise RuntimeError("qpth_amd.dist: a batched parameter has %d rows, nBatch is %d" % (X.size(0), nBatch))


def shard_params(params, nBatch, rank, world, ndims=(3, 2, 3, 2, 3, 2)):
    """slice batched parameters, pass un-batched / empty ones through (qpth/util.py:44-50)"""
    lo, hi = shard_bounds(nBatch, rank, world)
    out = []
    for X, nd in zip(params, ndims):
        out.append(X[lo:hi] if (X.nelement() > 0 and X.dim() == nd) else X)
    return out


class _Pending:
    """An all_gather in flight on the collective's own stream: `.wait()` makes the caller's stream wait for it and returns
    the full tensor (the caller runs its backward launches in between: the data path has no other collective)."""

    def __init__(self, out, work):
        self.out, self.work = out, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
        return self.out


def gather_batch(local, nBatch, group=None, async_op=False):
    """all_gather row blocks of possibly different length into the full (nBatch, ...) tensor.  async_op (equal slices
    only): returns a handle whose .wait() yields the tensor -- the collective then overlaps whatever the caller
    enqueues before waiting (QPFunction's backward needs nothing from the other ranks)."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(nBatch, r, world) for r in range(world)]
    maxlen = max(hi - lo for lo, hi in sizes)
    if nBatch % world == 0:
        # equal slices (the usual case): one collective straight into the full tensor -- no padding, no concatenation
        out = torch.empty((nBatch,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
        return _Pending(out, work) if async_op else out
    if async_op:
        return _Pending(gather_batch(local, nBatch, group), None)
    pad = torch.zeros((maxlen,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bf[:hi - lo] for bf, (lo, hi) in zip(bufs, sizes)], 0)


def reduce_shared_grad(local_mean_grad, n_local, nBatch, group=None):
    """Gradient of a batch-shared parameter: ranks hold the mean over their own slice; the
    global `.mean(0)` is sum_r (n_r / nBatch) * mean_r."""
    g = local_mean_grad * (float(n_local) / float(nBatch))
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g


def solve_sharded(qp_function, Q, p, G, h, A, b, nBatch, gather=True, group=None):
    """Solve the global batch data-parallel: this rank's slice through `qp_function`
    (a QPFunction(...) callable); returns the local zhat and, if gather, the full one."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    check_shardable([Q, p, G, h, A, b], nBatch, world)
    lQ, lp, lG, lh, lA, lb = shard_params([Q, p, G, h, A, b], nBatch, rank, world)
    z_local = qp_function(lQ, lp, lG, lh, lA, lb)
    if not gather:
        return z_local, None
    return z_local, gather_batch(z_local.detach(), nBatch, group)


def forward_sharded(Q, p, G, h, A, b, nBatch, gather=True, group=None, **kw):
    """The solver-level counterpart of solve_sharded: pre_factor_kkt + forward (qpth/qp.py:92-96) on this rank's slice
    of the global batch, returning what the reference's forward returns -- zhat, nu, lam, slacks -- for the slice and,
    if gather, all_gathered to the full batch (the z*, lambda*, nu* a caller keeps for its own backward).  `kw` goes
    to solvers.pdipm.batch.forward (eps, verbose, notImprovedLim, maxIter, solver)."""
    from .solvers.pdipm import batch as pdipm_b
    from .util import expandParam, extract_nBatch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    check_shardable([Q, p, G, h, A, b], nBatch, world)
    local_params = shard_params([Q, p, G, h, A, b], nBatch, rank, world)
    nloc = extract_nBatch(*local_params)
    # un-batched parameters become stride-0 views over the slice, as in QPFunction (qpth/qp.py:63-70): the solver
    # entry points take the batch size from Q, G, A
    lQ, lp, lG, lh, lA, lb = [expandParam(X, nloc, nd)[0] for X, nd in zip(local_params, (3, 2, 3, 2, 3, 2))]
    Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(lQ, lG, lA)
    local = pdipm_b.forward(lQ, lp, lG, lh, lA, lb, Q_LU, S_LU, R, **kw)
    if not gather:
        return local, None
    full = tuple(None if v is None else gather_batch(v, nBatch, group) for v in local)
    return local, full
