"""Pins the oracle (oracle/qp_oracle.c, the CPU restatement of the reference's PDIPM path)
against outputs of the reference itself (tests/golden/*.npz, made by make_golden.py).

Tolerances: the oracle restates getrf/getrs rather than calling MKL, so sums are ordered
differently; agreement is at round-off level (measured 1e-9..1e-13), asserted at 1e-7
relative for f64 -- three orders tighter than the 1e-4 gate of BASELINE.json.
"""
import numpy as np
import pytest

import problems
from conftest import load_golden, rel_err
from oracle import qp_oracle as orc

F64_TOL = 1e-7


def _inputs(g, dtype=np.float64):
    if "Q" in g:
        return [g[k] for k in ("Q", "p", "G", "h", "A", "b")]
    B, n, m, q, seed = [int(v) for v in g["shape"]]
    return list(problems.prof_qp(B, n, m, q, seed, dtype))


@pytest.mark.parametrize("name", ["dl_dp", "dl_dG", "dl_dh", "dl_dA", "dl_db"])
def test_grads_family(name):
    """test.py:99-187 problems: z*, lam, nu, slacks and all six gradients vs the reference."""
    g = load_golden("grads_" + name)
    Q, p, G, h, A, b = _inputs(g)
    x, y, z, s, grads, info = orc.qp_forward_backward(Q, p, G, h, A, b, dl_dz=g["dl_dz"])
    assert rel_err(x, g["zhat"]).max() < F64_TOL
    assert rel_err(z, g["lam"]).max() < 1e-6
    assert np.abs(s - g["slacks"]).max() < 1e-6 * max(1.0, np.abs(g["slacks"]).max())
    if A.size:
        assert rel_err(y, g["nu"]).max() < 1e-6
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k in g:
            ref = g[k]
            assert gr.shape == ref.shape, k
            assert np.abs(gr - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k


def test_kkt_solver():
    """test.py:222-234: pre_factor_kkt + factor_kkt + solve_kkt vs the reference's outputs."""
    g = load_golden("kkt_solver")
    o = orc.OracleQP(g["Q"], g["p"], g["G"], g["h"], g["A"], g["b"])
    o.factor_kkt(g["d"])
    dx, ds, dz, dy = o.solve_kkt(g["d"], g["rx"], g["rs"], g["rz"], g["ry"])
    for mine, key in ((dx, "dx"), (ds, "ds"), (dz, "dz"), (dy, "dy")):
        assert np.allclose(mine, g[key], rtol=1e-9, atol=1e-10), key
        # and against the reference's independent full-LU solver (test.py:225)
        assert np.allclose(mine, g["full_" + key], rtol=1e-4, atol=1e-2), key


@pytest.mark.parametrize("name,tol", [
    ("c1_b8_n10_m5_f64", F64_TOL), ("c3s_b4_n20_m10_q4_f64", F64_TOL),
    ("c2s_b4_n100_m100_f64", F64_TOL), ("c3s_b4_n100_m50_q10_f64", F64_TOL),
    ("c5s_b6_n64_m64_f64", F64_TOL), ("c1_b8_n10_m5_f32", 2e-3),
])
def test_prof_configs_batch_semantics(name, tol):
    """BASELINE.json configs (slices): whole-batch reference semantics (per_qp=0)."""
    g = load_golden(name)
    dtype = np.float32 if name.endswith("f32") else np.float64
    Q, p, G, h, A, b = _inputs(g, dtype)
    assert np.allclose(orc_checksum(Q, p, G, h, A, b), g["input_checksum"], rtol=1e-6)
    x, y, z, s, grads, info = orc.qp_forward_backward(Q, p, G, h, A, b, dl_dz=g["dl_dz"])
    assert rel_err(x, g["zhat"]).max() < tol
    # duals of a QP with no active constraint only decay towards zero (1e-9 after 20 float32 iterations here, 1e-19
    # in the reference's run): the error is measured against a floor, not against the norm of that "zero"
    lam_floor = 1e-6 if dtype == np.float32 else 1e-12
    lam_err = np.linalg.norm(z - g["lam"], axis=1) / np.maximum(np.linalg.norm(g["lam"], axis=1), lam_floor)
    assert lam_err.max() < 10 * tol
    gt = 20 * tol
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k in g:
            assert np.abs(gr - g[k]).max() <= gt * max(1.0, np.abs(g[k]).max()), k


@pytest.mark.parametrize("name", ["c1_b8_n10_m5_f64", "c3s_b4_n20_m10_q4_f64", "c5s_b6_n64_m64_f64"])
def test_per_qp_mode_is_reference_at_batch_one(name):
    """per_qp=1, stall_policy=1 must equal the reference run on each QP alone (b1_*)."""
    g = load_golden(name)
    Q, p, G, h, A, b = _inputs(g)
    o = orc.OracleQP(Q, p, G, h, A, b)
    x, y, z, s, info = o.forward(per_qp=True, stall_policy=1)
    assert rel_err(x, g["b1_zhat"]).max() < F64_TOL
    assert rel_err(z, g["b1_lam"]).max() < 1e-6


@pytest.mark.parametrize("name", ["broadcast_b5_n12_m9_q3", "unbatched_n12_m9_q3", "cls_b32_n2_m200_f64",
                                  "sudoku_b16_n64_m64_q40_f64"])
def test_broadcast_params(name):
    """un-batched parameters (util.py:44-59) and mean-reduced gradients (qp.py:159-177), including the two
    shapes the reference's own callers use (example-cls-layer.ipynb cell 3, example-sudoku.ipynb cell 10)."""
    g = load_golden(name)
    Q, p, G, h, A, b = _inputs(g)
    x, y, z, s, grads, info = orc.qp_forward_backward(Q, p, G, h, A, b, dl_dz=g["dl_dz"])
    # absolute on the scale of the batch: the cls inputs contain p = 0 rows, whose solution is z* = 0 exactly
    assert np.abs(x - g["zhat"]).max() < F64_TOL * max(1.0, np.abs(g["zhat"]).max())
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k not in g:
            continue
        ref = g[k]
        assert gr.shape == ref.shape, (k, gr.shape, ref.shape)
        assert np.abs(gr - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k


def orc_checksum(*arrs):
    return np.array([float(np.sum(np.asarray(a, np.float64) *
                                  np.cos(np.arange(np.asarray(a).size).reshape(np.shape(a)) % 97)))
                     for a in arrs if np.asarray(a).size])


EDGE = ["edge_b3_n1_m1_q0", "edge_b2_n6_m4_q5", "edge_dup_b2_n8_m10_q0"]


@pytest.mark.parametrize("name", EDGE)
def test_edge_shapes(name):
    """One variable / one constraint; neq = nz - 1; duplicated inequality rows -- against the reference run on
    the same inputs (tests/golden/make_golden.py --extra)."""
    g = load_golden(name)
    Q, p, G, h, A, b = [g[k] for k in ("Q", "p", "G", "h", "A", "b")]
    x, y, z, s, grads, info = orc.qp_forward_backward(Q, p, G, h, A, b, dl_dz=g["dl_dz"])
    assert rel_err(x, g["zhat"]).max() < F64_TOL
    assert np.abs(s - g["slacks"]).max() < 1e-6
    if "dup" not in name:                  # duplicated rows: the multipliers of a duplicated pair are not unique
        assert np.abs(z - g["lam"]).max() < 1e-6
    for k, gr in zip(("dQ", "dp", "dG", "dh", "dA", "db"), grads):
        if k in g and not ("dup" in name and k in ("dG", "dh")):
            assert np.abs(gr - g[k]).max() <= 1e-5 * max(1.0, np.abs(g[k]).max()), k
