"""TEST INFRASTRUCTURE: the finishing stage (KKTFactors.polish / qpx_polish) written as plain float64 tensor operations.

This is the host-composed version rounds 2-4 shipped in qpth_amd/kkt.py for the sizes without a kernel (~30 eager torch
operations per step).  Since round 5 every kernel family has the stage behind the C ABI; what remains of this code is the
step-by-step REFERENCE the kernels are compared with (tests/test_emu_parity.py, tests/test_gpu_parity.py): the same
iteration of the reference's loop (qpth/solvers/pdipm/batch.py:92-198) in the original variables, residuals from the
caller's data in float64, every KKT solve through the factors' solve_kkt, best iterate kept (batch.py:118-139).
"""
import torch


def polish_reference(fac, p, h, b, res, steps=2, refine=1):
    B, n, m, q = fac.B, fac.n, fac.m, fac.q
    hp = torch.float64
    if not fac.refine_ok:
        refine = 0
    # shared parameters stay un-batched: `.to(float64)` on a stride-0 expanded view would densify it (0.5 GB at C4)
    w = lambda X: (X[0] if (X.dim() == 3 and X.size(0) == B and X.stride(0) == 0 and B > 1) else X).to(hp)   # noqa: E731
    Q, G = w(fac.Q), w(fac.G)
    A = w(fac.A) if q else None
    mv = lambda M, x: torch.einsum("ij,bj->bi", M, x) if M.dim() == 2 else torch.einsum("bij,bj->bi", M, x)      # noqa: E731
    mtv = lambda M, x: torch.einsum("ij,bi->bj", M, x) if M.dim() == 2 else torch.einsum("bij,bi->bj", M, x)     # noqa: E731
    ex = lambda X: (X if X.dim() == 2 else X.unsqueeze(0).expand(B, *X.shape)).to(hp)   # noqa: E731
    pp, hh = ex(p), ex(h)
    bb = ex(b) if q else None
    x, z, s = res.zhat.to(hp), res.lam.to(hp), res.slacks.to(hp)
    y = res.nu.to(hp) if q else None
    tiny = torch.finfo(fac.dtype).tiny
    dt = fac.dtype

    def step(v, dv):
        r = torch.where(dv < 0, -v / dv.clamp_max(-tiny), torch.full_like(v, float("inf")))
        return r.min(1, keepdim=True)[0]

    def solve(d, rx, rs, rz, ry):
        o = fac.solve_kkt(d.to(dt), rx.to(dt), rs.to(dt), rz.to(dt), ry.to(dt) if q else None, refine=refine)
        return [v.to(hp) if v is not None else None for v in o]

    def residuals(x, s, z, y):
        rx = mv(Q, x) + pp + mtv(G, z)
        rz = mv(G, x) + s - hh
        ry = None
        if q:
            rx = rx + mtv(A, y)
            ry = mv(A, x) - bb
        mu = (s * z).sum(1, keepdim=True).abs() / m
        tot = rx.norm(dim=1, keepdim=True) + rz.norm(dim=1, keepdim=True) + m * mu     # batch.py:103-107
        if q:
            tot = tot + ry.norm(dim=1, keepdim=True)
        return rx, rz, ry, mu, tot

    rx, rz, ry, mu, best_r = residuals(x, s, z, y)
    best_r = torch.where(torch.isfinite(best_r), best_r, torch.full_like(best_r, float("inf")))
    bx, bs, bz, by = x, s, z, y
    for _ in range(steps):
        # one iteration of the reference's loop (batch.py:92-198) in float64 vector arithmetic
        sc, zc = s.clamp_min(tiny), z.clamp_min(tiny)
        d = zc / sc
        dxa, dsa, dza, dya = solve(d, rx, z, rz, ry)                                        # affine direction
        al = torch.minimum(step(z, dza), step(s, dsa)).clamp_max(1.0)
        sig = (((s + al * dsa) * (z + al * dza)).sum(1, keepdim=True) / (s * z).sum(1, keepdim=True)) ** 3
        rsc = (-mu * sig + dsa * dza) / sc
        zero_n, zero_m = torch.zeros_like(rx), torch.zeros_like(rz)
        dxc, dsc, dzc, dyc = solve(d, zero_n, rsc, zero_m, torch.zeros_like(ry) if q else None)   # corrector
        dx, ds, dz = dxa + dxc, dsa + dsc, dza + dzc
        alpha = (0.999 * torch.minimum(step(z, dz), step(s, ds))).clamp_max(1.0)
        x, s, z = x + alpha * dx, s + alpha * ds, z + alpha * dz
        if q:
            y = y + alpha * (dya + dyc)
        rx, rz, ry, mu, tot = residuals(x, s, z, y)
        better = tot < best_r                                   # False for NaN: a non-finite iterate never wins
        best_r = torch.where(better, tot, best_r)
        bx, bs, bz = torch.where(better, x, bx), torch.where(better, s, bs), torch.where(better, z, bz)
        if q:
            by = torch.where(better, y, by)
    res.zhat, res.lam, res.slacks = bx.to(fac.dtype), bz.to(fac.dtype), bs.to(fac.dtype)
    if q:
        res.nu = by.to(fac.dtype)
    return res
