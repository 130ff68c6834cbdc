"""Host logic + kernel bodies on CPU: the HIP kernel bodies (qpth_amd/csrc/qpx_kernels.h) run by
the host-thread emulator (tests/emu) through the real Python surface, checked against the
reference's golden vectors and the oracle.  These are the GPU-less stand-ins for the -m gpu
parity tests: same code path above the launch, same kernel source below it.

Tolerance: 1e-6 relative for f64 (north star: 1e-4; measured ~1e-12).
"""
import contextlib
import io

import numpy as np
import pytest
import torch

import problems
from conftest import load_golden, rel_err
from emu.harness import emulated
from oracle import qp_oracle as orc
from qpth_amd.qp import QPFunction
from qpth_amd.solvers.pdipm import batch as pdipm_b

TOL = 1e-6


def tens(arrs, dtype=torch.float64, grad=True):
    out = []
    for x in arrs:
        t = torch.tensor(np.asarray(x), dtype=dtype) if np.asarray(x).size else torch.empty(0, dtype=dtype)
        if grad and t.nelement() > 0:
            t.requires_grad_(True)
        out.append(t)
    return out


def run_qpf(arrs, dl, dtype=torch.float64, threads=128, variant=0, **kw):
    tq = tens(arrs, dtype)
    with emulated(threads, variant):
        z = QPFunction(verbose=-1, **kw)(*tq)
        z.backward(torch.tensor(dl, dtype=dtype))
    return z.detach().numpy(), [t.grad.numpy() if t.grad is not None else None for t in tq]


@pytest.mark.parametrize("name", ["dl_dp", "dl_dG", "dl_dh", "dl_dA", "dl_db"])
def test_reference_gradient_problems(name):
    """test.py:99-187 (B=1): zhat and every gradient the reference produces."""
    g = load_golden("grads_" + name)
    z, grads = run_qpf([g[k] for k in ("Q", "p", "G", "h", "A", "b")], g["dl_dz"])
    assert rel_err(z, g["zhat"]).max() < TOL
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k in g:
            assert gr.shape == g[k].shape
            assert np.abs(gr - g[k]).max() <= TOL * max(1.0, np.abs(g[k]).max()), k


def test_kkt_solver_entry_points():
    """test.py:222-234: pre_factor_kkt + factor_kkt + solve_kkt against the reference's outputs
    (same three calls, same argument order)."""
    g = load_golden("kkt_solver")
    Q, p, G, h, A, b = tens([g[k] for k in ("Q", "p", "G", "h", "A", "b")], grad=False)
    nB = 2
    Qe, Ae = Q.unsqueeze(0).expand(nB, 5, 5), A.unsqueeze(0).expand(nB, 3, 5)
    d, rx, rs, rz, ry = tens([g[k] for k in ("d", "rx", "rs", "rz", "ry")], grad=False)
    with emulated():
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(Qe, G, Ae)
        pdipm_b.factor_kkt(S_LU, R, d)
        dx, ds, dz, dy = pdipm_b.solve_kkt(Q_LU, d, G, Ae, S_LU, rx, rs, rz, ry)
    for mine, key in ((dx, "dx"), (ds, "ds"), (dz, "dz"), (dy, "dy")):
        assert np.allclose(mine.numpy(), g[key], rtol=1e-8, atol=1e-9), key
        assert np.allclose(mine.numpy(), g["full_" + key], rtol=1e-4, atol=1e-2), key   # test.py:231-234


@pytest.mark.parametrize("name", ["c1_b8_n10_m5_f64", "c3s_b4_n20_m10_q4_f64"])
def test_baseline_configs_small(name):
    """BASELINE.json configs[0] (and a small neq>0 case) vs the reference on the whole batch and
    vs the reference run one QP at a time."""
    g = load_golden(name)
    arrs = [g[k] for k in ("Q", "p", "G", "h", "A", "b")]
    z, grads = run_qpf(arrs, g["dl_dz"])
    assert rel_err(z, g["zhat"]).max() < TOL
    # the reference itself returns a different (unconverged) answer for some QPs when they are
    # solved alone: its not-improved counter is batch-global (batch.py:127-140).  Where it is
    # self-consistent, the per-QP kernel agrees with both.
    same = rel_err(g["b1_zhat"], g["zhat"]) < TOL
    assert same.sum() >= len(same) - 1
    assert rel_err(z[same], g["b1_zhat"][same]).max() < TOL
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k in g:
            assert np.abs(gr - g[k]).max() <= 10 * TOL * max(1.0, np.abs(g[k]).max()), k


def test_duals_and_slacks_match_reference():
    """forward() return order x, y, z, s = zhat, nu, lam, slacks (batch.py:143,207)."""
    g = load_golden("c3s_b4_n20_m10_q4_f64")
    Q, p, G, h, A, b = tens([g[k] for k in ("Q", "p", "G", "h", "A", "b")], grad=False)
    with emulated():
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(Q, G, A)
        x, y, z, s = pdipm_b.forward(Q, p, G, h, A, b, Q_LU, S_LU, R, verbose=-1)
    assert rel_err(x.numpy(), g["zhat"]).max() < TOL
    assert rel_err(y.numpy(), g["nu"]).max() < TOL
    assert rel_err(z.numpy(), g["lam"]).max() < 1e-5
    assert np.abs(s.numpy() - g["slacks"]).max() < 1e-6


@pytest.mark.parametrize("name", ["broadcast_b5_n12_m9_q3", "unbatched_n12_m9_q3"])
def test_broadcast_parameters_and_mean_reduced_grads(name):
    """util.py:44-59 + qp.py:159-177: un-batched params broadcast, their grads mean-reduced."""
    g = load_golden(name)
    z, grads = run_qpf([g[k] for k in ("Q", "p", "G", "h", "A", "b")], g["dl_dz"])
    assert z.shape == g["zhat"].shape
    assert rel_err(z, g["zhat"]).max() < TOL
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        assert gr.shape == g[k].shape, (k, gr.shape, g[k].shape)
        assert np.abs(gr - g[k]).max() <= TOL * max(1.0, np.abs(g[k]).max()), k


def test_two_slots_and_four_waves():
    """nz > 64 (two register slots per lane) with a 256-thread workgroup, vs the oracle."""
    Q, p, G, h, A, b = problems.prof_qp(1, 70, 66, 3, seed=5)
    dl = np.random.RandomState(0).randn(1, 70)
    x, y, lam, s, grads, info = orc.qp_forward_backward(Q, p, G, h, A, b, dl_dz=dl, per_qp=True, stall_policy=1)
    z, mine = run_qpf([Q, p, G, h, A, b], dl, threads=256)
    assert rel_err(z, x).max() < TOL
    for a_, r_ in zip(mine, grads):
        assert np.abs(a_ - r_).max() <= 10 * TOL * max(1.0, np.abs(r_).max())


def test_float32():
    g = load_golden("c1_b8_n10_m5_f32")
    z, grads = run_qpf([g[k] for k in ("Q", "p", "G", "h", "A", "b")], g["dl_dz"], dtype=torch.float32)
    g64 = load_golden("c1_b8_n10_m5_f64")
    # f32 is reported, not gated: both f32 solvers sit ~1e-4 from the f64 answer
    assert rel_err(z, g64["zhat"]).max() < 5e-3
    assert rel_err(g["zhat"], g64["zhat"]).max() < 5e-3


def test_not_spd_raises_like_the_reference():
    Q = -torch.eye(4, dtype=torch.float64).unsqueeze(0)
    p = torch.zeros(1, 4, dtype=torch.float64)
    G = torch.ones(1, 2, 4, dtype=torch.float64)
    h = torch.ones(1, 2, dtype=torch.float64)
    e = torch.empty(0, dtype=torch.float64)
    with emulated(64):
        with pytest.raises(RuntimeError, match="Q is not SPD."):               # qp.py:85
            QPFunction(verbose=-1)(Q, p, G, h, e, e)
        with pytest.raises(RuntimeError, match="Cannot perform LU factorization on Q"):   # batch.py:382
            QPFunction(verbose=-1, check_Q_spd=False)(Q, p, G, h, e, e)
        with pytest.raises(RuntimeError, match="Unexpected number of dimensions."):      # util.py:50
            QPFunction(verbose=-1)(Q.unsqueeze(0), p, G, h, e, e)


@pytest.mark.parametrize("shape", [(2, 100, 50, 10), (2, 60, 100, 6)])
def test_sweep_by_groups_of_four_reports_breakdowns(shape):
    """Round 3: at order n + q + m >= 145 the pre-factorisation sweeps four pivots per barrier pair (qpx_grid.h:
    sweep_group4); a pivot that breaks down INSIDE a group -- a Q that is not SPD, an A without full row rank -- must
    raise what the rank-1 sweep raises (batch.py:382-386, 419-423), and healthy QPs of the same sizes must still solve."""
    B, n, m, q = shape
    Q, p, G, h, A, b = [torch.tensor(x) for x in problems.prof_qp(B, n, m, q, seed=3)]
    with emulated(256):
        z = QPFunction(verbose=-1)(Q, p, G, h, A, b)
        assert torch.isfinite(z).all()
        Qb = Q.clone()
        Qb[1] = -Qb[1]
        with pytest.raises(RuntimeError, match="Cannot perform LU factorization on Q"):
            QPFunction(verbose=-1, check_Q_spd=False)(Qb, p, G, h, A, b)
        Ab = A.clone()
        Ab[1, 5] = 0                            # a zero row of A in the middle of a group: its pivot is exactly 0
        bb = b.clone()                          # (two EQUAL rows leave a pivot of rounding noise of either sign)
        bb[1, 5] = 0
        with pytest.raises(RuntimeError, match="full row rank"):
            QPFunction(verbose=-1)(Q, p, G, h, Ab, bb)


def _grid_blob_regions(n, m, q=0):
    """sub-arrays of the family-(a) blob with the tile image only (qpx_layout.h: fac_layout(n, m, q, 4))"""
    al = lambda x: (x + 3) & ~3
    nbt = [t for t in (1, 2, 4, 7) if (m + 15) // 16 <= t][0]
    reg, o = {}, 0
    for name, size in (("Kneg", n * n), ("MT", n * m), ("NTn", q * n), ("W", m * q), ("S11i", q * q)):
        if size:
            reg[name] = (o, size)
        o += al(size)
    reg["gt1"] = (o, 1)
    reg["Rm"] = (o + 12, nbt * (nbt + 1) // 2 * 256)
    return reg


PREFAC_SWEEP = 1 << 14      # include/qpx.h, qpx_set_ipm_variant: pre_factor_kkt by the thread-grid sweep


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("shape", [(2, 100, 100), (1, 70, 50), (1, 64, 64), (1, 50, 112), (1, 112, 3), (1, 81, 17), (1, 97, 111), (1, 49, 1),
                                   (2, 100, 50, 10), (1, 60, 70, 6), (1, 40, 30, 10), (1, 96, 20, 16), (1, 90, 40, 5), (1, 79, 17, 1), (1, 100, 96, 12), (1, 64, 64, 40), (1, 60, 30, 50), (1, 30, 20, 28)])
def test_matrix_core_prefactorisation_writes_the_sweeps_blob(shape, wide):
    """Round 4: pre_factor_kkt (batch.py:375-429) at neq = 0, 49 <= nz <= 112 is a factorisation of Q + tile products on
    the matrix cores (qpx_prefac.h) instead of the symmetric sweep.  Same blob, array by array: -K, M^T, || G^T 1 || and
    the tile image of R (padding included: exact zeros in both); float32 tensors in float64 arithmetic too.  With
    equality constraints (nz + neq <= 112; the factorisation is of [[Q, A^T], [A, 0]], its last neq pivots negative):
    -N^T, G N and (A Q^-1 A^T)^-1 as well."""
    from qpth_amd import kkt as _dp
    B, n, m = shape[:3]
    q = shape[3] if len(shape) > 3 else 0
    rng = np.random.default_rng(n * 1000 + m)
    L = rng.standard_normal((B, n, n))
    dt = torch.float32 if wide else torch.float64
    Q = torch.tensor(L @ L.transpose(0, 2, 1) + 1e-2 * np.eye(n), dtype=dt)
    G = torch.tensor(rng.standard_normal((B, m, n)), dtype=dt)
    e = torch.tensor(rng.standard_normal((B, q, n)), dtype=dt) if q else torch.empty(0, dtype=dt)
    blobs = []
    for variant in (0, PREFAC_SWEEP):
        with emulated(256, variant):
            fac = _dp.KKTFactors.build(Q, G, e, wide=wide)
            fac.raise_on_failure()
            blobs.append(fac.blob.reshape(B, -1).clone())
    assert blobs[0].dtype == torch.float64 and blobs[0].shape == blobs[1].shape
    for name, (o, ln) in _grid_blob_regions(n, m, q).items():
        mine, ref = blobs[0][:, o:o + ln].numpy(), blobs[1][:, o:o + ln].numpy()
        assert np.abs(mine - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), name
        if name == "Rm":
            assert ((ref == 0) == (mine == 0)).all()          # the padding of the image


@pytest.mark.parametrize("name", ["c2s_b4_n100_m100_f64", "c3s_b4_n100_m50_q10_f64", "sudoku_b16_n64_m64_q40_f64"])
def test_reference_outputs_through_the_matrix_core_prefactorisation(name):
    """The REFERENCE's own outputs (golden vectors made by the unmodified reference) at sizes whose pre-factorisation is
    the matrix-core one: C2's and C3's shapes (neq = 0 / 10) and the sudoku QPs (nz = 64, neq = 40: three tile rows of
    negative pivots), forward and every gradient the reference produced."""
    g = load_golden(name)
    if "Q" in g:
        arrs = [g[k] for k in ("Q", "p", "G", "h", "A", "b")]
    else:
        B, n, m, q, seed = [int(v) for v in g["shape"]]
        arrs = list(problems.prof_qp(B, n, m, q, seed, np.float64))
    z, grads = run_qpf(arrs, g["dl_dz"], threads=256)
    assert rel_err(z, g["zhat"]).max() < TOL
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k in g and gr is not None:
            assert gr.shape == g[k].shape, k
            assert np.abs(gr - g[k]).max() <= 10 * TOL * max(1.0, np.abs(g[k]).max()), k


def test_matrix_core_prefactorisation_random_shapes():
    """a seeded walk over the sizes the dispatcher gives the matrix-core pre-factorisation (49 <= nz + neq <= 112,
    nz + neq + nineq <= 208, any split between nz and neq, both dtypes): every array of the blob against the sweep's"""
    from qpth_amd import kkt as _dp
    rng = np.random.default_rng(20260925)
    for _ in range(16):
        nn = int(rng.integers(49, 113))
        q = int(rng.integers(0, min(nn // 2, 40) + 1)) if rng.random() < 0.6 else 0
        n, m, B, wide = nn - q, int(rng.integers(1, min(112, 208 - nn) + 1)), int(rng.integers(1, 3)), bool(rng.random() < 0.3)
        dt = torch.float32 if wide else torch.float64
        L = rng.standard_normal((B, n, n))
        Q = torch.tensor(L @ L.transpose(0, 2, 1) + 1e-1 * np.eye(n), dtype=dt)
        G = torch.tensor(rng.standard_normal((B, m, n)), dtype=dt)
        A = torch.tensor(rng.standard_normal((B, q, n)), dtype=dt) if q else torch.empty(0, dtype=dt)
        blobs = []
        for variant in (0, PREFAC_SWEEP):
            with emulated(256, variant):
                fac = _dp.KKTFactors.build(Q, G, A, wide=wide)
                fac.raise_on_failure()
                blobs.append(fac.blob.reshape(B, -1).clone())
        for name, (o, ln) in _grid_blob_regions(n, m, q).items():
            mine, ref = blobs[0][:, o:o + ln].numpy(), blobs[1][:, o:o + ln].numpy()
            assert np.abs(mine - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (name, B, n, m, q, wide)


def test_matrix_core_prefactorisation_reports_a_q_that_is_not_spd():
    """... and a pivot of Q that is not positive raises what the sweep raises (batch.py:382-386), for the QP it belongs to
    only: the healthy QP of the batch still solves when the broken one is taken out."""
    Q, p, G, h, A, b = [torch.tensor(x) for x in problems.prof_qp(2, 100, 50, 0, seed=3)]
    with emulated(256):
        Qb = Q.clone()
        Qb[1] = -Qb[1]
        with pytest.raises(RuntimeError, match="Cannot perform LU factorization on Q"):
            QPFunction(verbose=-1, check_Q_spd=False)(Qb, p, G, h, A, b)
        Qc = Q.clone()
        Qc[0, 70, 70] = -1.0                       # breaks down in the fifth panel
        with pytest.raises(RuntimeError, match="Cannot perform LU factorization on Q"):
            QPFunction(verbose=-1, check_Q_spd=False)(Qc, p, G, h, A, b)
        z = QPFunction(verbose=-1)(Q, p, G, h, A, b)
        with emulated(256, PREFAC_SWEEP):
            zs = QPFunction(verbose=-1)(Q, p, G, h, A, b)
    assert rel_err(z.numpy(), zs.numpy()).max() < 1e-9


def test_matrix_core_prefactorisation_with_equality_constraints_reports_breakdowns():
    """neq > 0: the pivots of the Q block must be positive, those of the equality block negative -- also inside ONE block
    of sixteen that holds both (nz = 100: rows 96 .. 99 and the first equality rows) -- and which of the two failed is
    what the reference's two messages tell apart (batch.py:382-386, 419-423)."""
    Q, p, G, h, A, b = [torch.tensor(x) for x in problems.prof_qp(2, 100, 50, 10, seed=3)]
    with emulated(256, PREFAC_SWEEP):
        zs = QPFunction(verbose=-1)(Q, p, G, h, A, b)
    with emulated(256):
        z = QPFunction(verbose=-1)(Q, p, G, h, A, b)
        for row in (3, 98):
            Qb = Q.clone()
            Qb[1, row, row] = -1.0
            with pytest.raises(RuntimeError, match="Cannot perform LU factorization on Q"):
                QPFunction(verbose=-1, check_Q_spd=False)(Qb, p, G, h, A, b)
        for row in (0, 5, 9):                       # a zero row of A: its pivot is exactly 0
            Ab, bb = A.clone(), b.clone()
            Ab[1, row] = 0
            bb[1, row] = 0
            with pytest.raises(RuntimeError, match="full row rank"):
                QPFunction(verbose=-1)(Q, p, G, h, Ab, bb)
    assert rel_err(z.numpy(), zs.numpy()).max() < 1e-9


def test_prefactorisation_knob_round_trips():
    """bit 14 of the A/B knob selects the sweep where the matrix-core form would run; it is part of the value
    qpx_get_ipm_variant returns (KKTFactors re-applies it around later calls on the factors)."""
    from emu.harness import emu_lib
    lib = emu_lib()
    old = lib.dll.qpx_set_ipm_variant(PREFAC_SWEEP | 1024)
    try:
        assert lib.dll.qpx_get_ipm_variant() == (PREFAC_SWEEP | 1024)
    finally:
        lib.dll.qpx_set_ipm_variant(old)


def test_verbose_trace_and_inaccuracy_warning():
    g = load_golden("c1_b8_n10_m5_f64")
    tq = tens([g[k] for k in ("Q", "p", "G", "h", "A", "b")], grad=False)
    buf = io.StringIO()
    with emulated(64), contextlib.redirect_stdout(buf):
        QPFunction(verbose=1)(*tq)
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("iter: ")]
    assert len(lines) >= 6 and "pri_resid" in lines[0] and "mu:" in lines[0]      # batch.py:115-117
    # first iteration: batch means of the residuals of the reference's start point
    ref0 = orc.OracleQP(*[g[k] for k in ("Q", "p", "G", "h", "A", "b")]).forward(want_trace=True)[4]["trace"][0]
    vals = [float(v) for v in lines[0].replace(",", "").split()[3::2]]
    assert np.allclose(vals, ref0, rtol=1e-4)
    # an infeasible QP -> INACC_ERR is printed, nothing raised (batch.py:141-142)
    Q = torch.eye(2, dtype=torch.float64).unsqueeze(0)
    p = torch.zeros(1, 2, dtype=torch.float64)
    G = torch.tensor([[[1.0, 0.0], [-1.0, 0.0]]], dtype=torch.float64)
    h = torch.tensor([[-1.0, -1.0]], dtype=torch.float64)                         # x <= -1 and x >= 1
    e = torch.empty(0, dtype=torch.float64)
    buf = io.StringIO()
    with emulated(64), contextlib.redirect_stdout(buf):
        QPFunction(verbose=0)(Q, p, G, h, e, e)
    assert "qpth warning: Returning an inaccurate" in buf.getvalue()


def test_batch_of_one_uses_the_reference_stall_counter():
    """B == 1: notImprovedLim behaves exactly like the reference (its counter is per batch)."""
    g = load_golden("c1_b8_n10_m5_f64")
    i = 7   # the QP of this batch whose residual is non-monotone early on
    arrs = [g[k][i:i + 1] for k in ("Q", "p", "G", "h")] + [np.zeros(0), np.zeros(0)]
    tq = tens(arrs, grad=False)
    with emulated(64):
        z = QPFunction(verbose=-1)(*tq)
    assert rel_err(z.numpy(), g["b1_zhat"][i:i + 1]).max() < TOL


# every form of the loop kernel the dispatcher can pick (include/qpx.h, qpx_set_ipm_variant):
# 3 = the large-QP family forced at a small size, +256 / +512 = 16x16 / 8x8 thread grid, +1024 = matrix-core
# tiles with 1 / 2 / 4 waves per QP (+2048 / +4096 / +8192)
LOOP_FORMS = [3, 256, 512, 1024 + 2048, 1024 + 4096, 1024 + 8192]


@pytest.mark.parametrize("variant", LOOP_FORMS)
@pytest.mark.parametrize("shape", [(2, 12, 9, 3), (1, 40, 52, 0)])
def test_every_loop_kernel_form(variant, shape):
    """The forms differ only in how the per-iteration linear algebra is laid out on the machine: all of
    them must return the oracle's optimum and gradients."""
    B, n, m, q = shape
    arrs = problems.prof_qp(B, n, m, q, seed=11)
    dl = np.random.RandomState(5).randn(B, n)
    xr, _, _, _, grads_ref, _ = orc.qp_forward_backward(*arrs, dl, per_qp=True, stall_policy=2)
    z, grads = run_qpf(arrs, dl, variant=variant)
    assert rel_err(z, xr).max() < TOL
    for mine, ref in zip(grads, grads_ref):
        if ref is not None and mine is not None:
            assert np.abs(mine - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


# The chain-wave form of the four-wave tile kernels (the default at 4 and 7 tile rows) against the oracle at the sizes
# that exercise its padding paths.  (Round 3 also kept the same kernels without the chain wave, +16384, and a
# pre-factorisation on matrix-core tiles, +32768: both lost their same-box A/Bs and were deleted in round 4; the round-3
# library is archived beside the product for comparisons, scripts/ab_bench.py.)
@pytest.mark.parametrize("shape", [(2, 30, 100, 0), (2, 20, 70, 3), (1, 100, 112, 0), (1, 10, 81, 0)])
def test_chain_wave_form(shape):
    B, n, m, q = shape
    arrs = problems.prof_qp(B, n, m, q, seed=11)
    dl = np.random.RandomState(5).randn(B, n)
    xr, _, _, _, grads_ref, _ = orc.qp_forward_backward(*arrs, dl, per_qp=True, stall_policy=(1 if B == 1 else 2))
    z, grads = run_qpf(arrs, dl, threads=256)
    assert rel_err(z, xr).max() < TOL
    for mine, ref in zip(grads, grads_ref):
        if ref is not None and mine is not None:
            assert np.abs(mine - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("variant", LOOP_FORMS)
@pytest.mark.parametrize("name", ["c1_b8_n10_m5_f64", "c3s_b4_n20_m10_q4_f64"])
def test_every_loop_kernel_form_against_the_reference(variant, name):
    """... and against the REFERENCE's own outputs (golden vectors, whole-batch semantics): the oracle above runs with
    this library's per-QP stop rule, which makes that test a cross-form check rather than reference parity."""
    g = load_golden(name)
    z, grads = run_qpf([g[k] for k in ("Q", "p", "G", "h", "A", "b")], g["dl_dz"], variant=variant)
    assert rel_err(z, g["zhat"]).max() < TOL
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k in g:
            assert np.abs(gr - g[k]).max() <= 10 * TOL * max(1.0, np.abs(g[k]).max()), k


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_against_the_oracle(seed):
    """Odd sizes on purpose: every padding rule (tiles of 16, panels of 4, slots of 64), neq = 0 and > 0,
    nineq smaller and larger than nz, one-QP and few-QP batches."""
    rng = np.random.RandomState(1000 + seed)
    B = int(rng.randint(1, 4))
    n = int(rng.randint(1, 36))
    m = int(rng.randint(1, 70))
    q = int(rng.randint(0, min(n, 6) + 1)) if n > 1 else 0
    arrs = problems.prof_qp(B, n, m, q, seed=seed)
    dl = rng.randn(B, n)
    per_qp_policy = 1 if B == 1 else 2             # the dispatcher's default (qpth_amd.kkt.default_stall_policy)
    xr, _, _, _, grads_ref, _ = orc.qp_forward_backward(*arrs, dl, per_qp=True, stall_policy=per_qp_policy)
    z, grads = run_qpf(arrs, dl)
    assert rel_err(z, xr).max() < TOL, (B, n, m, q)
    for mine, ref in zip(grads, grads_ref):
        if ref is not None and mine is not None:
            assert np.abs(mine - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (B, n, m, q)


def test_max_iter_and_eps_are_honoured():
    """QPFunction(eps, maxIter) (qp.py:18-20): the loop stops on maxIter (status bit, best iterate returned)
    and on best_resid < eps (batch.py:127-140) -- neither is exercised by the reference's own tests."""
    from qpth_amd import _lib
    from qpth_amd.kkt import KKTFactors
    arrs = problems.prof_qp(3, 14, 11, 2, seed=4)
    Q, p, G, h, A, b = tens(arrs, grad=False)
    with emulated():
        fac = KKTFactors.build(Q, G, A)
        full = fac.ipm(p, h, b)
        capped = fac.ipm(p, h, b, maxIter=3)
        loose = fac.ipm(p, h, b, eps=1e-3)
    assert int(capped.iters.max()) == 3
    assert all(int(s) & _lib.ST_MAXITER for s in capped.status.tolist())
    assert torch.isfinite(capped.zhat).all()
    assert int(loose.iters.max()) < int(full.iters.min())
    assert float(loose.best_resid.max()) < 1e-3
    assert rel_err(loose.zhat.numpy(), full.zhat.numpy()).max() < 1e-2


@pytest.mark.skipif(bool(__import__("os").environ.get("QPX_SKIP_TSAN")), reason="QPX_SKIP_TSAN set")
def test_kernel_bodies_are_race_free_under_thread_sanitizer():
    """`make -C tests/emu tsan`: every kernel family on host threads under ThreadSanitizer.  It found a real
    write-write overlap in the tile panel's publish (two lanes of one wave, ordered only by the GPU's in-order
    LDS pipeline) that the plain emulator runs never tripped over."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.check_call(["make", "-C", here, "-s", "tsan"], timeout=900)     # ~95 s to build, ~35 s to run


@pytest.mark.parametrize("name", ["edge_b3_n1_m1_q0", "edge_b2_n6_m4_q5", "edge_dup_b2_n8_m10_q0"])
def test_edge_shapes_match_the_reference(name):
    """One variable / one constraint; neq = nz - 1; duplicated inequality rows (their multipliers and the
    gradients with respect to the duplicated rows are not unique: only z*, slacks and the other gradients
    are compared there)."""
    g = load_golden(name)
    arrs = [g[k] for k in ("Q", "p", "G", "h", "A", "b")]
    z, grads = run_qpf(arrs, g["dl_dz"])
    assert rel_err(z, g["zhat"]).max() < TOL
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k in g and gr is not None and not ("dup" in name and k in ("dG", "dh")):
            assert np.abs(gr - g[k]).max() <= 1e-5 * max(1.0, np.abs(g[k]).max()), k


# ---------------------------------------------------------------- round 2: the large-QP family (qpx_big.h)
@pytest.mark.parametrize("shape,dtype,knob", [((1, 66, 70, 0), torch.float64, 3), ((1, 20, 70, 0), torch.float32, 3),
                                              ((3, 20, 70, 0), torch.float64, 3 + (3 << 16)),
                                              ((1, 66, 70, 0), torch.float64, 3 + (1 << 27)), ((1, 66, 70, 0), torch.float64, 3 + (1 << 28)),
                                              ((1, 66, 70, 0), torch.float64, 3 + (1 << 26)), ((2, 130, 200, 0), torch.float64, 3),
                                              ((2, 66, 70, 5), torch.float64, 3), ((1, 130, 40, 70), torch.float64, 3),
                                              ((2, 66, 70, 5), torch.float32, 3), ((2, 66, 70, 5), torch.float64, 3 + (2 << 16)),
                                              # five and four blocks (odd / even: every branch of the loop over pairs of panels)
                                              ((1, 130, 300, 0), torch.float64, 3), ((1, 250, 200, 10), torch.float64, 3), ((1, 200, 130, 0), torch.float32, 3)])
def test_large_qp_family(shape, dtype, knob):
    """BASELINE.json configs[3] runs through a multi-kernel family (blocked Cholesky / triangular solves / MFMA trailing
    updates on 64 x 64 blocks, matrices in HBM).  Forced here (knob 3) at sizes of two and three blocks so that the
    blocked code paths run on the emulator: zhat and every gradient against the oracle.  Third case: the batch
    split into three parts (knob bits 16..19), as the host does on the GPU to overlap the parts on side streams.  Knob
    bits 27 / 28: the diagonal blocks by one wave /
    on the thread grid instead of the chain-wave form; bits 25 / 26: four-wave substitutions and R z' in front of the
    factorisation, the round-3 order (all kept for same-box A/B; nineq = 200: four blocks, both shapes of the
    update launches).  Round 4: equality constraints
    (neq = 5: one block; neq = 70: two blocks, the blocked solves with L11), all six gradients.  (The float32 KERNELS of
    this family -- QPFunction(refine=k) on float32 tensors; the default is float64 arithmetic -- lose gradient accuracy
    with equality constraints when nz < nineq: 0.18 relative at nz = 30, nineq = 70, neq = 3 on this generator.)"""
    B, n, m, q = shape
    f32 = dtype == torch.float32
    arrs = problems.prof_qp(B, n, m, q, seed=3, dtype=np.float32 if f32 else np.float64)
    arrs64 = problems.prof_qp(B, n, m, q, seed=3)
    dl = np.random.RandomState(0).randn(B, n)
    x, y, lam, s, grads, info = orc.qp_forward_backward(*arrs64, dl_dz=dl, per_qp=True, stall_policy=2)
    z, mine = run_qpf(arrs, dl.astype(arrs[0].dtype), dtype=dtype, threads=256, variant=knob, **({"refine": 2} if f32 else {}))
    tol = 5e-3 if f32 else TOL
    assert rel_err(z, x).max() < tol
    for a_, r_ in zip(mine, grads):
        if r_ is not None:
            assert np.abs(a_ - r_).max() <= 20 * tol * max(1.0, np.abs(r_).max())


@pytest.mark.parametrize("shape", [(2, 66, 70, 0), (2, 66, 70, 5)])
def test_large_qp_family_float32_tensors_in_float64_arithmetic(shape):
    """QPX_F32_WIDE in the large-QP family (round 4): float32 tensors, float64 blob and arithmetic -- the pack kernels
    widen on load, every output narrows on store.  Against the float64 run on the same (float32-rounded) data: forward
    and all gradients to float32 rounding, duals included."""
    from qpth_amd.kkt import KKTFactors
    B, n, m, q = shape
    arrs32 = problems.prof_qp(B, n, m, q, seed=4, dtype=np.float32)
    arrs64 = [np.asarray(a, np.float64) for a in arrs32]
    dl = np.random.RandomState(1).randn(B, n).astype(np.float32)
    z64, g64 = run_qpf(arrs64, dl.astype(np.float64), threads=256, variant=3)
    z32, g32 = run_qpf(arrs32, dl, dtype=torch.float32, threads=256, variant=3)
    assert z32.dtype == np.float32
    assert rel_err(z32, z64).max() < 1e-6
    for k, a, b_ in zip(("dQ", "dp", "dG", "dh", "dA", "db"), g32, g64):
        assert (a is None) == (b_ is None), k
        if a is not None:
            assert a.dtype == np.float32 and a.shape == b_.shape, k
            assert np.abs(a - b_).max() <= 1e-6 * max(1.0, np.abs(b_).max()), k
    t32 = tens(arrs32, torch.float32, grad=False)
    t64 = tens(arrs64, grad=False)
    with emulated(256, 3):
        f32 = KKTFactors.build(t32[0], t32[2], t32[4], wide=True)
        r32 = f32.ipm(t32[1], t32[3], t32[5])
        f64 = KKTFactors.build(t64[0], t64[2], t64[4])
        r64 = f64.ipm(t64[1], t64[3], t64[5])
    assert f32.blob.dtype == torch.float64 and r32.lam.dtype == torch.float32
    assert np.abs(r32.lam.numpy() - r64.lam.numpy()).max() < 1e-6 * max(1.0, np.abs(r64.lam.numpy()).max())
    if q:
        assert np.abs(r32.nu.numpy() - r64.nu.numpy()).max() < 1e-6 * max(1.0, np.abs(r64.nu.numpy()).max())


def test_large_qp_family_kkt_solve_with_equality_constraints():
    """factor_kkt + solve_kkt (batch.py:349-372, 435-470) in the large-QP family with neq > 0: random right-hand sides
    (rx, rs, rz, ry), the residual of the reference's KKT system (kkt_resid_reg, batch.py:228-241) must vanish."""
    B, n, m, q = 2, 66, 40, 9
    Q, p, G, h, A, b = problems.prof_qp(B, n, m, q, seed=6)
    r = np.random.RandomState(2)
    d = r.rand(B, m) + 0.1
    rx, rs, rz, ry = [r.randn(B, k) for k in (n, m, m, q)]
    tt = lambda x: torch.tensor(x)   # noqa: E731
    with emulated(256, 3):
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(tt(Q), tt(G), tt(A))
        outs = pdipm_b.solve_kkt(Q_LU, tt(d), tt(G), tt(A), S_LU, tt(rx), tt(rs), tt(rz), tt(ry))
    res = _kkt_residual(Q, G, A, d, rx, rs, rz, ry, *[v.numpy() for v in outs])
    scale = max(np.linalg.norm(v) for v in (rx, rs, rz, ry))
    assert (res < 1e-9 * scale).all(), res


# ---------------------------------------------------------------- round 2: host logic around the new C-ABI arguments
@pytest.mark.parametrize("name", ["c3s_b4_n20_m10_q4_f64", "broadcast_b5_n12_m9_q3"])
def test_backward_from_external_solutions(name):
    """QPSolvers.CVXPY (qp.py:97-120,142-155): forward = an external solver's (zhat, nu, lam, slacks) -- the
    reference's own, replayed from the golden file -- backward = qpx_backward on rebuilt factors (ctx.fac None)."""
    from qpth_amd.qp import QPSolvers
    from qpth_amd.solvers import external
    g = load_golden(name)
    arrs = [g[k] for k in ("Q", "p", "G", "h", "A", "b")]
    calls = []

    def replay(Q, p, G, h, A, b):
        i = len(calls)
        calls.append(i)
        return g["zhat"][i], (g["nu"][i] if g["nu"].shape[1] else None), g["lam"][i], g["slacks"][i]

    external.set_solver(replay)
    try:
        tq = tens(arrs)
        with emulated():
            z = QPFunction(verbose=-1, solver=QPSolvers.CVXPY)(*tq)
            z.backward(torch.tensor(g["dl_dz"]))
    finally:
        external.set_solver(None)
    assert len(calls) == g["zhat"].shape[0]
    assert np.array_equal(z.detach().numpy(), g["zhat"])
    for k, t in zip(("dQ", "dp", "dG", "dh", "dA", "db"), tq):
        if k in g:
            assert t.grad.shape == g[k].shape, k
            assert np.abs(t.grad.numpy() - g[k]).max() <= 1e-5 * max(1.0, np.abs(g[k]).max()), k


def test_cvxpy_adapter_against_a_stand_in_for_cvxpy(monkeypatch):
    """solvers/external.py:_cvxpy_solve is written against cvxpy's public API, and cvxpy is not in the image (VERDICT r3:
    the adapter had never executed).  A stand-in module with the handful of names it uses -- Variable, quad_form,
    psd_wrap, Minimize, Problem, `G @ z <= h`, `A @ z == b`, .value / .dual_value / .status -- whose Problem.solve() is the
    oracle and whose dual_value follows cvxpy's documented conventions (an inequality's dual is its non-negative
    multiplier; an equality's is the multiplier of A z - b in the Lagrangian, which is the reference's nu,
    qpth/solvers/cvxpy.py:24-27): QPFunction(solver=QPSolvers.CVXPY) through it must reproduce the reference's zhat and
    gradients.  What this pins is the adapter's plumbing and sign handling, not cvxpy itself."""
    import sys
    import types
    from qpth_amd.qp import QPSolvers
    from qpth_amd.solvers import external

    class Expr:
        __array_ufunc__ = None          # (as cvxpy's expressions: numpy defers `ndarray @ expr` to __rmatmul__)

        def __init__(self, kind, **kw):
            self.kind, self.__dict__ = kind, dict(kind=kind, **kw)

    class Variable:
        __array_ufunc__ = None

        def __init__(self, n):
            self.n, self.value = n, None

        def __rmatmul__(self, M):
            return Expr("affine", M=np.asarray(M), var=self)

    def _cmp(kind):
        def f(self, rhs):
            c = Expr(kind, M=self.M, var=self.var, rhs=np.asarray(rhs))
            c.dual_value = None
            return c
        return f
    Expr.__le__ = _cmp("le")
    Expr.__eq__ = _cmp("eq")
    Expr.__hash__ = object.__hash__
    Expr.__add__ = lambda self, other: Expr("sum", terms=[self, other])
    Expr.__rmul__ = lambda self, c: Expr("scaled", c=c, e=self)

    fake = types.ModuleType("cvxpy")
    fake.Variable = Variable
    fake.psd_wrap = lambda Q: np.asarray(Q)
    fake.quad_form = lambda z, Q: Expr("quad", Q=np.asarray(Q), var=z)
    fake.Minimize = lambda e: e

    class Problem:
        def __init__(self, obj, cons):
            self.obj, self.cons, self.status = obj, cons, None

        def solve(self):
            quad = [t.e for t in self.obj.terms if t.kind == "scaled"][0]
            lin = [t for t in self.obj.terms if t.kind == "affine"][0]
            Q, p, z = quad.Q, lin.M, quad.var
            ineq = [c for c in self.cons if c.kind == "le"][0]
            eqs = [c for c in self.cons if c.kind == "eq"]
            A = eqs[0].M[None] if eqs else np.zeros(0)
            b = eqs[0].rhs[None] if eqs else np.zeros(0)
            x, y, lam, s, info = orc.OracleQP(Q[None], p[None], ineq.M[None], ineq.rhs[None], A, b).forward()
            z.value = x[0]
            ineq.dual_value = lam[0]
            if eqs:
                eqs[0].dual_value = y[0]
            self.status = "optimal"
    fake.Problem = Problem
    monkeypatch.setitem(sys.modules, "cvxpy", fake)
    external.set_solver(None)
    g = load_golden("c3s_b4_n20_m10_q4_f64")
    tq = tens([g[k] for k in ("Q", "p", "G", "h", "A", "b")])
    with emulated():
        z = QPFunction(verbose=-1, solver=QPSolvers.CVXPY)(*tq)
        z.backward(torch.tensor(g["dl_dz"]))
    assert rel_err(z.detach().numpy(), g["zhat"]).max() < TOL
    for k, t in zip(("dQ", "dp", "dG", "dh", "dA", "db"), tq):
        assert np.abs(t.grad.numpy() - g[k]).max() <= 1e-5 * max(1.0, np.abs(g[k]).max()), k


def test_external_solver_without_cvxpy_fails_loudly():
    from qpth_amd.qp import QPSolvers
    arrs = problems.prof_qp(1, 4, 3, 0, seed=1)
    with emulated():
        with pytest.raises(RuntimeError, match="cvxpy"):
            QPFunction(verbose=-1, solver=QPSolvers.CVXPY)(*tens(arrs, grad=False))


def test_needs_input_grad_is_honoured():
    """Only the gradients autograd asks for are computed (NULL outputs of qpx_backward)."""
    arrs = problems.prof_qp(3, 12, 9, 3, seed=2)
    dl = np.random.RandomState(2).randn(3, 12)
    _, full = run_qpf(arrs, dl)
    for only in (1, 3, 2, 4):
        tq = tens(arrs, grad=False)
        tq[only].requires_grad_(True)
        with emulated():
            z = QPFunction(verbose=-1)(*tq)
            z.backward(torch.tensor(dl))
        for i, t in enumerate(tq):
            assert (t.grad is not None) == (i == only)
        assert np.array_equal(tq[only].grad.numpy(), full[only])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_shared_parameter_gradients_are_one_contraction(dtype):
    """Q, G, A shared by the batch: one factor blob, gradients = the batch mean (qp.py:159-177) formed by
    qpx_batch_outer (an MFMA contraction over the batch in both dtypes) instead of B outer products."""
    from qpth_amd.kkt import KKTFactors
    g = load_golden("broadcast_b5_n12_m9_q3")
    arrs = [g[k] for k in ("Q", "p", "G", "h", "A", "b")]
    z, grads = run_qpf(arrs, g["dl_dz"], dtype=dtype)
    tol = 1e-6 if dtype == torch.float64 else 2e-3
    assert np.abs(z - g["zhat"]).max() < tol * max(1.0, np.abs(g["zhat"]).max())
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        assert gr.shape == g[k].shape, k
        assert np.abs(gr - g[k]).max() <= 10 * tol * max(1.0, np.abs(g[k]).max()), k
    tq = tens(arrs, dtype, grad=False)
    with emulated():
        fac = KKTFactors.build(tq[0], tq[2], tq[4], nBatch=5)
    assert fac.shared and fac.blob.numel() == fac.elems


# ---------------------------------------------------------------- round 2: accuracy options (batch.py:216-346)
def _kkt_residual(Q, G, A, d, rx, rs, rz, ry, dx, ds, dz, dy):
    """|| K sol + rhs || of the reference's KKT system (kkt_resid_reg, batch.py:228-241), in float64"""
    f = lambda v: np.asarray(v, np.float64)
    Q, G, A, d, rx, rs, rz, ry, dx, ds, dz, dy = [f(v) for v in (Q, G, A, d, rx, rs, rz, ry, dx, ds, dz, dy)]
    e1 = np.einsum('bij,bj->bi', Q, dx) + np.einsum('bmi,bm->bi', G, dz) + np.einsum('bqi,bq->bi', A, dy) + rx
    e2 = d * ds + dz + rs
    e3 = np.einsum('bmi,bi->bm', G, dx) + ds + rz
    e4 = np.einsum('bqi,bi->bq', A, dx) + ry
    return np.sqrt((e1 ** 2).sum(1) + (e2 ** 2).sum(1) + (e3 ** 2).sum(1) + (e4 ** 2).sum(1))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("shape", [(600, 10, 37), (1100, 33, 20), (40, 5, 5)])
def test_batch_contraction_in_two_stages(shape, dtype):
    """qpx_batch_outer (qp.py:159-177 for a shared parameter): long batches are contracted in TWO stages when the caller
    brings the workspace -- partial tiles per chunk of whole 256-QP trips, then their sum in chunk order -- and by one
    workgroup per tile otherwise; both against the plain sum of outer products, and the two-stage result twice
    (fixed order: bit-identical)."""
    import ctypes
    from emu.harness import emu_lib
    from qpth_amd import _lib
    B, r, c = shape
    g = torch.Generator().manual_seed(B + r)
    u, w = torch.randn(B, r, dtype=dtype, generator=g), torch.randn(B, r, dtype=dtype, generator=g)
    v, x = torch.randn(B, c, dtype=dtype, generator=g), torch.randn(B, c, dtype=dtype, generator=g)
    ref = 0.5 / B * (u.double().t() @ v.double() + w.double().t() @ x.double())
    lib = emu_lib()
    code = _lib.QPX_F64 if dtype == torch.float64 else _lib.QPX_F32
    need = int(lib.dll.qpx_batch_outer_workspace_elems(code, B, r, c))
    assert (need > 0) == (B > 256)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    outs = []
    with emulated(256):
        for rep in range(2):
            out = torch.full((r, c), float("nan"), dtype=dtype)
            lib.batch_outer(u, v, w, x, 0.5, out)                       # two stages when need > 0
            outs.append(out)
        one = torch.full((r, c), float("nan"), dtype=dtype)
        lib.check(lib.dll.qpx_batch_outer(code, B, r, c, u.data_ptr(), v.data_ptr(), w.data_ptr(), x.data_ptr(), 0.5,
                                          one.data_ptr(), None, 0, None))                 # no workspace: one stage
        # ABI v8: v == NULL = a column of ones, w == x == NULL = no second product: the batch mean of u's columns, with
        # the sign in the scale -- the `.mean(0)` of a shared VECTOR parameter's gradient (qp.py:160-166,174-177)
        means = []
        for rep in range(2):
            mo = torch.full((r,), float("nan"), dtype=dtype)
            lib.batch_outer(u, None, None, None, -1.0, mo)
            means.append(mo)
        bad = lib.dll.qpx_batch_outer(code, B, r, c + 1, u.data_ptr(), None, None, None, 1.0, one.data_ptr(), None, 0, None)
    assert bad == -1                                                    # QPX_ERR_ARG: a column of ones means c = 1
    assert torch.equal(outs[0], outs[1]) and torch.equal(means[0], means[1])
    for o in (outs[0], one):
        assert (o.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    mref = -u.double().mean(0)
    assert (means[0].double() - mref).abs().max().item() <= tol * max(1.0, mref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("k", [1, 3, 40, 130])
def test_small_dense_solve(k, dtype):
    """qpx_dense_solve (v7): one general k x k system per workgroup by pivoted elimination -- the neq x neq correction of
    factor_solve_kkt_reg (batch.py:273-310) -- against numpy, on matrices that NEED the pivoting (zero diagonal), and a
    singular one (status bit, NaNs)."""
    from emu.harness import emu_lib
    from qpth_amd import _lib
    rng = np.random.RandomState(k)
    B = 3
    M = rng.randn(B, k, k)
    if k > 1:
        M[:, np.arange(k), np.arange(k)] = 0.0                       # every natural pivot is zero
    M[0] += 3.0 * np.eye(k)[::-1] if k > 1 else 0.0
    r = rng.randn(B, k)
    if k > 1:
        M[2, :, 0] = M[2, :, 1]                                      # QP 2: two equal columns = singular
    ref = [np.linalg.solve(M[i], r[i]) for i in range(2 if k > 1 else B)]
    tM, tr = torch.tensor(M, dtype=dtype), torch.tensor(r, dtype=dtype)
    st = torch.zeros(B, dtype=torch.int32)
    with emulated(256):
        x = emu_lib().dense_solve(tM.clone(), tr.clone(), st).numpy()
    tol = 1e-9 if dtype == torch.float64 else 5e-3
    for i, xr in enumerate(ref):
        assert np.abs(x[i] - xr).max() <= tol * max(1.0, np.abs(xr).max()), (i, np.abs(x[i] - xr).max())
        assert st[i] == 0
    if k > 1:
        # (in floating point the elimination of a singular matrix may end on a tiny pivot instead of an exact zero: either
        # the status bit, or a solution that does not satisfy the system)
        assert (st[2] & _lib.ST_KKT_BREAKDOWN) or not np.allclose(M[2] @ x[2], r[2], atol=1e-6)


def test_iterative_refinement_of_the_kkt_solve():
    """solve_kkt_ir (batch.py:244-270): refinement on the residual of the ORIGINAL system inside the kernel (residuals
    accumulated in float64).  In float32 on the benchmark generator one step cuts the KKT residual by > 50x; in
    float64 the refined solve still equals the reference's (test.py:236-247, test_ir_kkt_solver)."""
    B, n, m, q = 3, 30, 20, 4
    Q, p, G, h, A, b = problems.prof_qp(B, n, m, q, seed=2, dtype=np.float32)
    r = np.random.RandomState(1)
    d = (r.rand(B, m) + 0.1).astype(np.float32)
    rx, rs, rz, ry = [r.randn(B, k).astype(np.float32) for k in (n, m, m, q)]
    tt = lambda x: torch.tensor(x)   # noqa: E731
    with emulated():
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(tt(Q), tt(G), tt(A))
        o0 = pdipm_b.solve_kkt(Q_LU, tt(d), tt(G), tt(A), S_LU, tt(rx), tt(rs), tt(rz), tt(ry))
        o1 = pdipm_b.solve_kkt_ir(Q_LU, tt(d), tt(G), tt(A), S_LU, tt(rx), tt(rs), tt(rz), tt(ry), niter=1)
    r0 = _kkt_residual(Q, G, A, d, rx, rs, rz, ry, *[v.numpy() for v in o0])
    r1 = _kkt_residual(Q, G, A, d, rx, rs, rz, ry, *[v.numpy() for v in o1])
    assert (r1 < r0 / 50).all(), (r0, r1)
    g = load_golden("kkt_solver")
    Qg, pg, Gg, hg, Ag, bg = tens([g[k] for k in ("Q", "p", "G", "h", "A", "b")], grad=False)
    Qe, Ae = Qg.unsqueeze(0).expand(2, 5, 5), Ag.unsqueeze(0).expand(2, 3, 5)
    dd, rxx, rss, rzz, ryy = tens([g[k] for k in ("d", "rx", "rs", "rz", "ry")], grad=False)
    with emulated():
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(Qe, Gg, Ae)
        outs = pdipm_b.solve_kkt_ir(Q_LU, dd, Gg, Ae, S_LU, rxx, rss, rzz, ryy, niter=1)
    for mine, key in zip(outs, ("dx", "ds", "dz", "dy")):
        assert np.allclose(mine.numpy(), g["full_" + key], rtol=1e-4, atol=1e-2), key     # test.py:243-246
        assert np.allclose(mine.numpy(), g[key], rtol=1e-8, atol=1e-9), key


@pytest.mark.parametrize("solver", ["LU_FULL", "LU_PARTIAL", "IR_UNOPT"])
def test_forward_accepts_every_kkt_solver(solver):
    """forward(..., solver=KKTSolvers.X) (batch.py:47-48): all three names solve the same QPs."""
    g = load_golden("c3s_b4_n20_m10_q4_f64")
    Q, p, G, h, A, b = tens([g[k] for k in ("Q", "p", "G", "h", "A", "b")], grad=False)
    with emulated():
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(Q, G, A)
        x, y, z, s = pdipm_b.forward(Q, p, G, h, A, b, Q_LU, S_LU, R, verbose=-1, solver=getattr(pdipm_b.KKTSolvers, solver))
    assert rel_err(x.numpy(), g["zhat"]).max() < TOL
    assert rel_err(y.numpy(), g["nu"]).max() < TOL


def _kkt_solver_golden():
    g = load_golden("kkt_solver")
    Qg, Gg, Ag = tens([g[k] for k in ("Q", "G", "A")], grad=False)
    Qe, Ae = Qg.unsqueeze(0).expand(2, 5, 5), Ag.unsqueeze(0).expand(2, 3, 5)
    d, rx, rs, rz, ry = tens([g[k] for k in ("d", "rx", "rs", "rz", "ry")], grad=False)
    return g, Qe, Gg, Ae, d, rx, rs, rz, ry


def test_full_system_solvers_with_the_reference_signatures():
    """factor_solve_kkt(Q, D, G, A, rx, rs, rz, ry) and solve_kkt_ir(Q, D, G, A, ..., niter) (batch.py:244-270, 313-346; D =
    diag(d) as a matrix) against the reference's own factor_solve_kkt outputs (test.py:222-247, golden full_*)."""
    g, Qe, Gg, Ae, d, rx, rs, rz, ry = _kkt_solver_golden()
    D = torch.diag_embed(d)
    with emulated():
        full = pdipm_b.factor_solve_kkt(Qe, D, Gg, Ae, rx, rs, rz, ry)
        ir = pdipm_b.solve_kkt_ir(Qe, D, Gg, Ae, rx, rs, rz, ry, niter=1)
    for a, c, key in zip(full, ir, ("dx", "ds", "dz", "dy")):
        assert np.allclose(a.numpy(), g["full_" + key], rtol=1e-8, atol=1e-9), key
        assert np.allclose(c.numpy(), g["full_" + key], rtol=1e-8, atol=1e-9), key


def _dense_reg_solve(Q, D, G, rx, rs, rz, eps, A=None, ry=None):
    """the regularised KKT system of batch.py:273-310 solved densely in numpy (unknowns dx, ds, dz, dy)"""
    n, m = Q.shape[0], G.shape[0]
    q = 0 if A is None else A.shape[0]
    K = np.zeros((n + 2 * m + q, n + 2 * m + q))
    K[:n, :n] = Q; K[:n, n + m:n + 2 * m] = G.T
    K[n:n + m, n:n + m] = D; K[n:n + m, n + m:n + 2 * m] = np.eye(m)
    K[n + m:n + 2 * m, :n] = G; K[n + m:n + 2 * m, n:n + m] = np.eye(m); K[n + m:n + 2 * m, n + m:n + 2 * m] = -eps * np.eye(m)
    rhs = [rx, rs, rz]
    if q:
        K[:n, n + 2 * m:] = A.T; K[n + 2 * m:, :n] = A; K[n + 2 * m:, n + 2 * m:] = -eps * np.eye(q)
        rhs.append(ry)
    sol = np.linalg.solve(K, -np.concatenate(rhs))
    return sol[:n], sol[n:n + m], sol[n + m:n + 2 * m], sol[n + 2 * m:]


@pytest.mark.parametrize("q", [0, 4])
def test_regularised_full_solve(q):
    """factor_solve_kkt_reg (batch.py:273-310) against a dense solve of the same regularised system: without equality
    constraints the -eps I block is a change of d; with them (round 4) a rank-neq correction on top (neq + 2 launches)."""
    B, n, m = 3, 12, 9
    Q, p, G, h, A, b = problems.prof_qp(B, n, m, q, seed=4)
    r = np.random.RandomState(2)
    d = r.rand(B, m) + 0.1
    rx, rs, rz, ry = r.randn(B, n), r.randn(B, m), r.randn(B, m), r.randn(B, q)
    eps = 1e-3
    tA = torch.tensor(A) if q else torch.empty(0, dtype=torch.float64)
    with emulated():
        dx, ds, dz, dy = pdipm_b.factor_solve_kkt_reg(torch.tensor(Q), torch.diag_embed(torch.tensor(d)), torch.tensor(G), tA,
                                                      torch.tensor(rx), torch.tensor(rs), torch.tensor(rz),
                                                      torch.tensor(ry) if q else None, eps)
    assert (dy is None) == (q == 0)
    for i in range(B):
        ex, es, ez, ey = _dense_reg_solve(Q[i], np.diag(d[i]), G[i], rx[i], rs[i], rz[i], eps, A[i] if q else None, ry[i] if q else None)
        pairs = [(dx[i], ex), (ds[i], es), (dz[i], ez)] + ([(dy[i], ey)] if q else [])
        for mine, ref in pairs:
            assert np.abs(mine.numpy() - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())
    # the reference's own use of it (solve_kkt_ir: eps = 1e-7) on its KKT test problem (test.py:190-219): within 1e-6 of
    # the un-regularised solution the golden file holds
    g, Qe, Gg, Ae, dd, rxx, rss, rzz, ryy = _kkt_solver_golden()
    with emulated():
        outs = pdipm_b.factor_solve_kkt_reg(Qe + 1e-7 * torch.eye(5, dtype=torch.float64), torch.diag_embed(dd) + 1e-7 * torch.eye(4, dtype=torch.float64),
                                            Gg, Ae, rxx, rss, rzz, ryy, 1e-7)
    for mine, key in zip(outs, ("dx", "ds", "dz", "dy")):
        assert np.allclose(mine.numpy(), g["full_" + key], rtol=1e-5, atol=1e-5), key


def test_refinement_is_refused_loudly_where_no_kernel_implements_it():
    """qpx_factor_solve_kkt(refine > 0) on a kernel family without in-kernel refinement: QPX_ERR_UNSUPPORTED (ABI v5),
    never a silently un-refined answer; the finishing stage then runs with plain solves."""
    Q, p, G, h, A, b = [torch.tensor(x) for x in problems.prof_qp(1, 20, 70, 0, seed=1)]
    r = np.random.RandomState(0)
    d, rx, rs, rz = [torch.tensor(v) for v in (r.rand(1, 70) + 0.1, r.randn(1, 20), r.randn(1, 70), r.randn(1, 70))]
    with emulated(256, 3):                       # knob 3: the large-QP family
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(Q, G, A)
        assert not Q_LU.fac.refine_ok
        pdipm_b.solve_kkt(Q_LU, d, G, A, S_LU, rx, rs, rz, None)
        with pytest.raises(RuntimeError, match="not supported"):
            pdipm_b.solve_kkt_ir(Q_LU, d, G, A, S_LU, rx, rs, rz, None, niter=1)
    with emulated(256, 0):
        assert pdipm_b.pre_factor_kkt(Q, G, A)[0].fac.refine_ok


def test_finishing_steps_keep_the_best_iterate():
    """KKTFactors.polish keeps, per QP, the iterate with the smallest residual (batch.py:118-139): finishing steps on a
    QP the float32 loop could not finish must not make the answer worse (advisor finding of round 2: 1.4e-6 -> 3.6e-3
    without the guard)."""
    rng = np.random.RandomState(7)
    B, n, m = 3, 20, 18
    L = rng.rand(n, n)
    Q = (L @ L.T + 1e-3 * np.eye(n)).astype(np.float32)
    G = rng.randn(m, n).astype(np.float32)
    z0, s0 = rng.randn(B, n), rng.rand(B, m)
    p = rng.randn(B, n).astype(np.float32)
    h = (z0 @ G.T.astype(np.float64) + s0).astype(np.float32)
    e = np.zeros(0, np.float32)
    ref, _ = run_qpf([Q.astype(np.float64), p.astype(np.float64), G.astype(np.float64), h.astype(np.float64), e, e],
                     np.zeros((B, n)), variant=256)
    errs = {}
    for refine in (0, 2, 5):
        tq = tens([Q, p, G, h, e, e], torch.float32, grad=False)
        with emulated(128, 256):
            z = QPFunction(verbose=-1, refine=refine)(*tq)
        errs[refine] = rel_err(z.numpy(), ref).max()
    assert errs[2] <= 2 * errs[0] + 1e-6 and errs[5] <= 2 * errs[0] + 1e-6, errs


def test_float32_finishing_steps_reach_the_reference_accuracy():
    """QPFunction in float32: the loop kernel alone (refine=0) lands ~1e-4 from the float64 answer on the benchmark
    generator (cond(Q) ~ 1e6); with finishing steps (refine=2) it is as close as the reference's own float32 run, and
    so is the default at this size (refine=None: float32 data, float64 arithmetic).  Two QPs of the C2 golden pair on
    the emulator; the distribution over 32 QPs is asserted on the GPU."""
    g = load_golden("f32pair_c2_b32_n100_m100")
    B, n, m, q, seed = [int(v) for v in g["shape"]]
    arrs = [a[:2] if a.size else a for a in problems.prof_qp(B, n, m, q, seed, np.float32)]
    ref64, ref32 = g["zhat_f64"][:2], g["zhat_f32"][:2]
    tq = tens(arrs, torch.float32, grad=False)
    with emulated(256):
        fast = QPFunction(verbose=-1, refine=0)(*tq)
        good = QPFunction(verbose=-1, refine=2)(*tq)
        wide = QPFunction(verbose=-1)(*tq)
    assert wide.dtype == torch.float32
    e_fast, e_good, e_ref = rel_err(fast.numpy(), ref64), rel_err(good.numpy(), ref64), rel_err(ref32, ref64)
    e_wide = rel_err(wide.numpy(), ref64)
    assert (e_good < np.maximum(10 * e_ref, 2e-5)).all(), (e_fast, e_good, e_ref)
    assert (e_good < e_fast).all(), (e_fast, e_good)
    assert (e_wide < np.maximum(10 * e_ref, 2e-5)).all(), (e_wide, e_ref)


@pytest.mark.parametrize("shape,dtype,variant", [((3, 30, 20, 4), torch.float32, 0), ((2, 100, 100, 0), torch.float32, 0),
                                                 ((3, 30, 20, 4), torch.float64, 0), ((2, 20, 40, 3), torch.float64, 256),
                                                 ((2, 100, 100, 0), torch.float64, 0), ((2, 30, 50, 5), torch.float64, 0),
                                                 # knob 3: the large-QP family's finishing stage (qpx_big_polish.h, round 5), two and three blocks of 64
                                                 ((2, 70, 66, 3), torch.float32, 3), ((2, 130, 100, 0), torch.float64, 3), ((2, 66, 130, 5), torch.float64, 3)])
def test_finishing_stage_kernel_equals_the_host_version(shape, dtype, variant):
    """qpx_polish (include/qpx.h v6): the finishing stage as ONE kernel -- thread-grid form (float32; float64 with
    knob 256) and matrix-core tile forms (float64: one wave per QP at 2 tile rows, the chain-wave form at 4 and 7) --
    against the host-composed version it replaced (tests/polish_reference.py: the same iteration as float64 tensor ops), from
    the same start iterate, step by step: equal to rounding, equality constraints included.  The start iterate is the
    loop kernel's result after THREE iterations, so that the steps have something to do."""
    from polish_reference import polish_reference
    from qpth_amd.kkt import KKTFactors
    B, n, m, q = shape
    f32 = dtype == torch.float32
    arrs = problems.prof_qp(B, n, m, q, seed=9, dtype=np.float32 if f32 else np.float64)
    tQ, tp, tG, th, tA, tb = tens(arrs, dtype, grad=False)
    with emulated(256, variant):
        fac = KKTFactors.build(tQ, tG, tA)
        assert fac.lib.dll.qpx_polish_supported(0 if f32 else 1, n, m, q) == 1
        for steps in (1, 2):
            outs = []
            for fn in (fac.polish, lambda *a_, **k_: polish_reference(fac, *a_, **k_)):
                res = fac.ipm(tp, th, tb, maxIter=3)
                res = fn(tp, th, tb, res, steps=steps, refine=1)
                outs.append([res.zhat.numpy().copy(), res.lam.numpy().copy(), res.slacks.numpy().copy()] + ([res.nu.numpy().copy()] if q else []))
            tol = 2e-4 if f32 else 1e-9
            for a_, b_ in zip(*outs):
                assert np.abs(a_ - b_).max() <= tol * max(1.0, np.abs(b_).max()), (steps, np.abs(a_ - b_).max())


@pytest.mark.parametrize("name", ["c3s_b4_n20_m10_q4_f64", "broadcast_b5_n12_m9_q3", "unbatched_n12_m9_q3"])
def test_float32_data_in_float64_arithmetic(name):
    """float32 tensors at a size the float64 tile kernels serve (QPFunction(refine=None)): the parameters are widened
    on the device, the float64 kernels run, and everything the caller sees is float32 again -- equal, to float32
    rounding, to the float64 run on the same (float32-representable) data; parameters the batch shares stay one
    factor blob and get their batch-mean gradient."""
    from qpth_amd.qp import f64_arithmetic_serves
    g = load_golden(name)
    arrs32 = [np.asarray(g[k], np.float32) for k in ("Q", "p", "G", "h", "A", "b")]
    dl = np.asarray(g["dl_dz"], np.float32)
    assert f64_arithmetic_serves(arrs32[2].shape[-1], arrs32[2].shape[-2], arrs32[4].shape[-2] if arrs32[4].size else 0)
    z32, g32 = run_qpf(arrs32, dl, dtype=torch.float32)
    z64, g64 = run_qpf([a.astype(np.float64) for a in arrs32], dl.astype(np.float64))
    assert z32.dtype == np.float32 and z32.shape == z64.shape
    assert rel_err(z32, z64).max() < 1e-6
    for k, a, b_ in zip(("dQ", "dp", "dG", "dh", "dA", "db"), g32, g64):
        assert (a is None) == (b_ is None), k
        if a is not None:
            assert a.dtype == np.float32 and a.shape == b_.shape, k
            assert np.abs(a - b_).max() <= 1e-6 * max(1.0, np.abs(b_).max()), k


def test_float32_sizes_outside_the_matrix_core_kernels_keep_the_float32_kernels():
    """float32 tensors run in float64 arithmetic where the float64 matrix-core kernels serve the size: the tile kernels
    and (round 4) the large-QP family; the thread-grid-only and workgroup sizes keep the float32 kernels."""
    from qpth_amd import _lib
    from qpth_amd.qp import f64_arithmetic_serves
    assert f64_arithmetic_serves(100, 100, 0) and f64_arithmetic_serves(100, 50, 10) and f64_arithmetic_serves(64, 64, 0)
    assert not f64_arithmetic_serves(2, 200, 0) and not f64_arithmetic_serves(100, 100, 10)
    with emulated(64):
        lib = _lib.backend_for(torch.zeros(1))
        fam = lambda n, m, q, d=_lib.QPX_F64: lib.dll.qpx_kernel_family(d, n, m, q)   # noqa: E731
        assert fam(100, 100, 0) == _lib.FAMILY_TILE and fam(100, 100, 0, _lib.QPX_F32) == _lib.FAMILY_GRID
        assert fam(2, 200, 0) == _lib.FAMILY_GRID and fam(500, 500, 0) == _lib.FAMILY_BIG and fam(300, 200, 50) == _lib.FAMILY_BIG
        assert f64_arithmetic_serves(100, 100, 0, lib) and f64_arithmetic_serves(500, 500, 0, lib) and f64_arithmetic_serves(300, 200, 50, lib)
        assert not f64_arithmetic_serves(2, 200, 0, lib)


def test_f32_wide_abi_surface():
    """QPX_F32_WIDE (include/qpx.h): float32 arrays, float64 factors -- which sizes it serves, what it refuses, and
    that the error surface of QPFunction (not SPD, INACC warning, verbose trace) is the same through it."""
    from qpth_amd import _lib
    from qpth_amd.kkt import KKTFactors
    with emulated(64):
        dll = _lib.backend_for(torch.zeros(1)).dll
        assert dll.qpx_supported(_lib.QPX_F32_WIDE, 100, 100, 0) == 0
        assert dll.qpx_supported(_lib.QPX_F32_WIDE, 500, 500, 0) == 0           # the large-QP family widens on load too (round 4)
        assert dll.qpx_supported(_lib.QPX_F64, 500, 500, 0) == 0
        assert dll.qpx_factor_elems(_lib.QPX_F32_WIDE, 100, 100, 0) == dll.qpx_factor_elems(_lib.QPX_F64, 100, 100, 0)
        assert dll.qpx_supported(_lib.QPX_F32_WIDE, 600, 100, 0) == 0            # (round 6: up to 1 024 per dimension)
        assert dll.qpx_supported(_lib.QPX_F32_WIDE, 1100, 100, 0) == -2          # beyond 1 024 per dimension: nobody serves it
        assert dll.qpx_polish_supported(_lib.QPX_F64, 600, 100, 0) == 0 and dll.qpx_polish_supported(_lib.QPX_F64, 500, 500, 0) == 1
        # refinement reads Q, G, A in the kernels' own type: refused, loudly
        g = load_golden("c3s_b4_n20_m10_q4_f64")
        tq = tens([np.asarray(g[k], np.float32) for k in ("Q", "p", "G", "h", "A", "b")], torch.float32, grad=False)
        fac = KKTFactors.build(tq[0], tq[2], tq[4], wide=True)
        assert fac.blob.dtype == torch.float64
        d = torch.ones(4, 10)
        with pytest.raises(RuntimeError, match="code -2"):     # QPX_ERR_UNSUPPORTED since ABI v5
            fac.solve_kkt(d, tq[1], d, d, tq[5], refine=1)
        # the general KKT solve through the wide interface = the float64 solve of the same data, to float32 rounding
        outs32 = fac.solve_kkt(d, tq[1], d, d, tq[5])
        f64 = KKTFactors.build(tq[0].double(), tq[2].double(), tq[4].double())
        outs64 = f64.solve_kkt(d.double(), tq[1].double(), d.double(), d.double(), tq[5].double())
        for a_, b_ in zip(outs32, outs64):
            assert a_.dtype == torch.float32
            assert (a_.double() - b_).abs().max() <= 2e-6 * max(1.0, float(b_.abs().max()))
        with pytest.raises(TypeError):
            KKTFactors.build(tq[0].double(), tq[2].double(), tq[4].double(), wide=True)
    # error surface
    Q = -torch.eye(4).unsqueeze(0)
    p = torch.zeros(1, 4)
    G = torch.ones(1, 2, 4)
    h = torch.ones(1, 2)
    e = torch.empty(0)
    with emulated(64):
        with pytest.raises(RuntimeError, match="Q is not SPD."):
            QPFunction(verbose=-1)(Q, p, G, h, e, e)
    g = load_golden("c1_b8_n10_m5_f32")
    tq = tens([g[k] for k in ("Q", "p", "G", "h", "A", "b")], torch.float32, grad=False)
    buf = io.StringIO()
    with emulated(64), contextlib.redirect_stdout(buf):
        z = QPFunction(verbose=1)(*tq)
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("iter: ")]
    assert len(lines) >= 6 and "nan" not in lines[0]
    assert rel_err(z.numpy(), load_golden("c1_b8_n10_m5_f64")["zhat"]).max() < 1e-4
    Q = torch.eye(2).unsqueeze(0)
    p = torch.zeros(1, 2)
    G = torch.tensor([[[1.0, 0.0], [-1.0, 0.0]]])
    h = torch.tensor([[-1.0, -1.0]])
    buf = io.StringIO()
    with emulated(64), contextlib.redirect_stdout(buf):
        QPFunction(verbose=0)(Q, p, G, h, e, e)
    assert "qpth warning: Returning an inaccurate" in buf.getvalue()


def test_mismatched_parameters_are_refused_before_any_kernel_runs():
    """The kernels index raw pointers, so a parameter of another dtype, batch size or length must never reach them
    (the reference fails inside bmm / baddbmm; here it would be an out-of-bounds read)."""
    Q, p, G, h, A, b = [torch.tensor(x) for x in problems.prof_qp(3, 6, 4, 2, 0)]
    e = torch.empty(0, dtype=torch.float64)
    bad = {
        "p is torch.float32": (Q, p.float(), G, h, A, b),
        "G is torch.float32": (Q, p, G.float(), h, A, b),
        r"p has shape \(2, 6\)": (Q, p[:2], G, h, A, b),
        r"G has shape \(2, 4, 6\)": (Q, p, G[:2], h, A, b),
        r"h has shape \(3, 3\)": (Q, p, G, h[:, :3], A, b),
        "b is empty": (Q, p, G, h, A, e),
        "inconsistent QP sizes": (Q[:, :5, :5], p, G, h, A, b),
    }
    with emulated(64):
        for msg, args in bad.items():
            with pytest.raises(RuntimeError, match=msg):
                QPFunction(verbose=-1)(*args)
        with pytest.raises(TypeError, match="float32 and float64"):
            QPFunction(verbose=-1)(*[x.half() for x in (Q, p, G, h, A, b)])
        # what IS accepted: non-contiguous views, un-batched vectors, a batch of one given without the batch dimension
        z0 = QPFunction(verbose=-1)(Q, p, G, h, A, b)
        z1 = QPFunction(verbose=-1)(Q.transpose(1, 2), p, G, h, A, b)             # Q symmetric
        assert torch.allclose(z0, z1, rtol=1e-9, atol=1e-12)
        z2 = QPFunction(verbose=-1)(Q[0], p[0], G[0], h[0], A[0], b[0])
        assert z2.shape == (1, 6) and torch.allclose(z2[0], z0[0], rtol=1e-9, atol=1e-12)


def test_sizes_beyond_512_on_the_emulator():
    """Round 6: max(nz, nineq, neq) up to 1 024 (qpx_max_dim; the reference has no cap, batch.py:375-470).  One QP of
    nz = 580, nineq = 600, neq = 20 -- ten blocks of 64: sixteen vector slots per lane (big_phase_body<T, 16>), the
    substitution's rounds of streamed blocks (big_trsv_body<T, true>), sixteen blocks of the vector in the mat-vec's LDS,
    equality constraints by projection -- through the Python surface on the kernel bodies: zhat and all six gradients
    against the oracle (the GPU test of the same name runs 768 / 1 000 / 1 024 at batch size)."""
    from oracle import qp_oracle as orc
    from qpth_amd import _lib
    B, n, m, q = 1, 580, 600, 20
    arrs = problems.prof_qp(B, n, m, q, seed=3)
    dl = np.ones((B, n))
    x, y, lam, s, grads, info = orc.qp_forward_backward(*arrs, dl_dz=dl, per_qp=True)
    with emulated(64):
        dll = _lib.backend_for(torch.zeros(1)).dll
        assert dll.qpx_max_dim() == 1024 and dll.qpx_supported(_lib.QPX_F64, n, m, q) == 0
        assert dll.qpx_supported(_lib.QPX_F64, 1025, 8, 0) == -2 and dll.qpx_polish_supported(_lib.QPX_F64, n, m, q) == 0
    z, mine = run_qpf(arrs, dl)
    assert rel_err(z, x).max() < 1e-9
    for k, a_, r_ in zip(("dQ", "dp", "dG", "dh", "dA", "db"), mine, grads):
        assert np.abs(a_ - r_).max() <= 1e-8 * max(1.0, np.abs(r_).max()), k
    # the finishing stage stops at 512 per dimension and says so by name
    with pytest.raises(RuntimeError, match="served up to 512"):
        run_qpf([np.asarray(a_, np.float32) for a_ in arrs], dl.astype(np.float32), dtype=torch.float32, refine=2)
