// tests/emu/tsan_driver.cpp -- runs the kernel bodies (host-thread emulation) under ThreadSanitizer.
// TEST INFRASTRUCTURE ONLY.  One pthread per GPU thread means every LDS exchange that is not ordered by
// a barrier (__syncthreads / wave-level ordering point) shows up as a data race -- including the ones
// a GPU would hide by executing a wave in lock step.  Usage: tsan_driver <B> <n> <m> <q> [variant [wide|-] [extras]]
// ("wide": after the float64 run, the same QPs through QPX_F32_WIDE -- float32 arrays of exactly the right size, so
// that an I/O site that still indexed them as doubles is a heap overflow under AddressSanitizer)
// Exit code 0 and no "WARNING: ThreadSanitizer" on stderr = clean.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/qpx.h"

static double urand(unsigned& s)
{
    s = s * 1664525u + 1013904223u;
    return (double)(s >> 8) / (double)(1u << 24);
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 1, n = argc > 2 ? atoi(argv[2]) : 20, m = argc > 3 ? atoi(argv[3]) : 24;
    const int q = argc > 4 ? atoi(argv[4]) : 3;
    if (argc > 5) qpx_set_ipm_variant(atoi(argv[5]));
    bool extras = false;                 // "extras": also the launches that do not depend on the size
    int max_iter = 20;                   // "it=N": stop the loop after N iterations (every launch of a pass has run by then;
                                         //  the feasibility check at the end is skipped): bounds the sanitizer runs of the large-QP family
    for (int i = 1; i < argc; ++i) {
        if (std::string(argv[i]) == "extras") extras = true;
        if (std::string(argv[i]).rfind("it=", 0) == 0) max_iter = atoi(argv[i] + 3);
    }
    unsigned seed = 12345;
    std::vector<double> Q((size_t)B * n * n), p((size_t)B * n), G((size_t)B * m * n), h((size_t)B * m);
    std::vector<double> A((size_t)B * q * n + 1), bb((size_t)B * q + 1);
    for (int s = 0; s < B; ++s) {
        std::vector<double> L((size_t)n * n);
        for (auto& v : L) v = urand(seed) - 0.5;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = (i == j) ? 1e-3 : 0.0;
                for (int k = 0; k < n; ++k) acc += L[(size_t)i * n + k] * L[(size_t)j * n + k];
                Q[((size_t)s * n + i) * n + j] = acc;
            }
        std::vector<double> z0(n);
        for (int i = 0; i < n; ++i) { p[(size_t)s * n + i] = urand(seed) - 0.5; z0[i] = urand(seed) - 0.5; }
        for (int i = 0; i < m; ++i) {
            double acc = 0;
            for (int j = 0; j < n; ++j) { const double g = urand(seed) - 0.5; G[((size_t)s * m + i) * n + j] = g; acc += g * z0[j]; }
            h[(size_t)s * m + i] = acc + urand(seed);                 // feasible: h = G z0 + s0, s0 > 0
        }
        for (int i = 0; i < q; ++i) {
            double acc = 0;
            for (int j = 0; j < n; ++j) { const double a = urand(seed) - 0.5; A[((size_t)s * q + i) * n + j] = a; acc += a * z0[j]; }
            bb[(size_t)s * q + i] = acc;
        }
    }
    const size_t fe = qpx_factor_elems(QPX_F64, n, m, q);
    std::vector<double> fac((size_t)B * fe), zhat((size_t)B * n), nu((size_t)B * q + 1), lam((size_t)B * m), sl((size_t)B * m), br(B);
    std::vector<int32_t> status(B), iters(B);
    int rc = qpx_pre_factor(QPX_F64, B, n, m, q, Q.data(), (int64_t)n * n, G.data(), (int64_t)m * n, q ? A.data() : nullptr,
                            (int64_t)q * n, fac.data(), status.data(), nullptr);
    if (rc) { fprintf(stderr, "pre_factor rc %d\n", rc); return 2; }
    rc = qpx_ipm(QPX_F64, B, n, m, q, p.data(), n, h.data(), m, q ? bb.data() : nullptr, q, fac.data(), (int64_t)fe, 1e-12, max_iter, 3,
                 B == 1 ? 1 : 2, zhat.data(), q ? nu.data() : nullptr, lam.data(), sl.data(), iters.data(), status.data(), br.data(),
                 nullptr, nullptr);
    if (rc) { fprintf(stderr, "ipm rc %d\n", rc); return 2; }
    std::vector<double> g((size_t)B * n, 1.0), dQ((size_t)B * n * n), dp((size_t)B * n), dG((size_t)B * m * n), dh((size_t)B * m);
    std::vector<double> dA((size_t)B * q * n + 1), db((size_t)B * q + 1), dx((size_t)B * n), dz((size_t)B * m), dy((size_t)B * q + 1);
    rc = qpx_backward(QPX_F64, B, n, m, q, fac.data(), (int64_t)fe, zhat.data(), lam.data(), sl.data(), q ? nu.data() : nullptr, g.data(),
                      dQ.data(), dp.data(), dG.data(), dh.data(), q ? dA.data() : nullptr, q ? db.data() : nullptr,
                      dx.data(), dz.data(), q ? dy.data() : nullptr, qpx_refine_supported(QPX_F64, n, m, q) ? 1 : 0 /* one refinement step where the family has it (ABI v5: refused elsewhere) */,
                      Q.data(), (int64_t)n * n, G.data(), (int64_t)m * n,
                      q ? A.data() : nullptr, (int64_t)q * n, status.data(), nullptr);
    if (rc) { fprintf(stderr, "backward rc %d\n", rc); return 2; }
    // the finishing stage (qpx_polish, v6) where the kernel family has it: one step from the loop's result, one refinement
    // step per solve; it may only improve the residual it reports
    if (qpx_polish_supported(QPX_F64, n, m, q)) {
        std::vector<double> z2 = zhat, nu2 = nu, lam2 = lam, sl2 = sl, br2(B);
        rc = qpx_polish(QPX_F64, B, n, m, q, Q.data(), (int64_t)n * n, p.data(), n, G.data(), (int64_t)m * n, h.data(), m,
                        q ? A.data() : nullptr, (int64_t)q * n, q ? bb.data() : nullptr, q, fac.data(), (int64_t)fe, 1,
                        qpx_refine_supported(QPX_F64, n, m, q) ? 1 : 0 /* the large-QP family's stage (v7) has no in-solve refinement */,
                        z2.data(), q ? nu2.data() : nullptr, lam2.data(), sl2.data(), br2.data(), status.data(), nullptr);
        if (rc) { fprintf(stderr, "polish rc %d\n", rc); return 2; }
        for (int s = 0; s < B; ++s)
            if (!(br2[s] == br2[s])) { fprintf(stderr, "polish: NaN residual for qp %d\n", s); return 6; }
    }
    // the shared-parameter reduction (batch-mean of dQ as one contraction over the batch)
    std::vector<double> dQm((size_t)n * n);
    rc = qpx_batch_outer(QPX_F64, B, n, n, dx.data(), zhat.data(), zhat.data(), dx.data(), 0.5, dQm.data(), nullptr, 0, nullptr);
    if (rc) { fprintf(stderr, "batch_outer rc %d\n", rc); return 2; }
    if (extras) {
        // ... and its two-stage form on a batch long enough to be cut (three chunks of 256 copies of the vectors), and the
        // small dense solve (qpx_dense_solve, v7) on a matrix that needs its pivoting
        const int BB = 600, r = n < 20 ? n : 20;
        std::vector<double> uu((size_t)BB * r), vv((size_t)BB * r), out2((size_t)r * r), out1((size_t)r * r);
        for (int s = 0; s < BB; ++s)
            for (int i = 0; i < r; ++i) { uu[(size_t)s * r + i] = dx[i] + 1e-3 * s; vv[(size_t)s * r + i] = zhat[i] - 1e-3 * s; }
        const size_t need = qpx_batch_outer_workspace_elems(QPX_F64, BB, r, r);
        std::vector<double> ws(need + 1);
        if (need == 0) { fprintf(stderr, "batch_outer: no workspace asked for a batch of %d\n", BB); return 7; }
        rc = qpx_batch_outer(QPX_F64, BB, r, r, uu.data(), vv.data(), vv.data(), uu.data(), 0.5, out2.data(), ws.data(), need, nullptr);
        if (!rc) rc = qpx_batch_outer(QPX_F64, BB, r, r, uu.data(), vv.data(), vv.data(), uu.data(), 0.5, out1.data(), nullptr, 0, nullptr);
        if (rc) { fprintf(stderr, "batch_outer (two stages) rc %d\n", rc); return 2; }
        for (int i = 0; i < r * r; ++i)
            if (std::fabs(out1[i] - out2[i]) > 1e-9 * (1.0 + std::fabs(out1[i]))) { fprintf(stderr, "batch_outer: stages disagree at %d\n", i); return 3; }
        const int k = 9;
        std::vector<double> Mk((size_t)k * k), rk(k), xk(k);
        for (int i = 0; i < k; ++i) {
            xk[i] = urand(seed) - 0.5;
            for (int j = 0; j < k; ++j) Mk[(size_t)i * k + j] = (i == j) ? 0.0 : urand(seed) - 0.5;       // zero diagonal
        }
        for (int i = 0; i < k; ++i) { double acc = 0; for (int j = 0; j < k; ++j) acc += Mk[(size_t)i * k + j] * xk[j]; rk[i] = acc; }
        int32_t st1 = 0;
        rc = qpx_dense_solve(QPX_F64, 1, k, Mk.data(), rk.data(), &st1, nullptr);
        if (rc || st1) { fprintf(stderr, "dense_solve rc %d status %d\n", rc, st1); return 2; }
        for (int i = 0; i < k; ++i)
            if (std::fabs(rk[i] - xk[i]) > 1e-8) { fprintf(stderr, "dense_solve mismatch at %d\n", i); return 3; }
    }
    for (int i = 0; i < n * n; ++i) {
        double acc = 0;
        for (int s = 0; s < B; ++s) acc += dQ[(size_t)s * n * n + i];
        if (std::fabs(acc / B - dQm[i]) > 1e-9 * (1.0 + std::fabs(dQm[i]))) { fprintf(stderr, "batch_outer mismatch at %d\n", i); return 3; }
    }
    double worst = 0;
    for (int s = 0; s < B; ++s) {
        for (int i = 0; i < m; ++i) {                               // primal feasibility G z <= h
            double acc = -h[(size_t)s * m + i];
            for (int j = 0; j < n; ++j) acc += G[((size_t)s * m + i) * n + j] * zhat[(size_t)s * n + j];
            worst = std::fmax(worst, acc);
        }
        printf("qp %d: iters %d status %d best_resid %.2e\n", s, iters[s], status[s], br[s]);
    }
    printf("max constraint violation %.2e\n", worst);
    if (worst >= 1e-6 && max_iter >= 20) return 1;
    if (argc > 6 && std::string(argv[6]) == "wide") {
        if (qpx_supported(QPX_F32_WIDE, n, m, q) != 0) { fprintf(stderr, "QPX_F32_WIDE not served at this size / knob\n"); return 4; }
        auto narrow = [](const std::vector<double>& v, size_t cnt) { std::vector<float> o(cnt); for (size_t i = 0; i < cnt; ++i) o[i] = (float)v[i]; return o; };
        std::vector<float> Qf = narrow(Q, Q.size()), pf = narrow(p, p.size()), Gf = narrow(G, G.size()), hf = narrow(h, h.size());
        std::vector<float> Af = narrow(A, (size_t)B * q * n), bf = narrow(bb, (size_t)B * q);
        std::vector<float> zf((size_t)B * n), nuf((size_t)B * q), lamf((size_t)B * m), slf((size_t)B * m), brf(B), gf((size_t)B * n, 1.0f);
        std::vector<float> dQf((size_t)B * n * n), dpf((size_t)B * n), dGf((size_t)B * m * n), dhf((size_t)B * m), dAf((size_t)B * q * n), dbf((size_t)B * q);
        std::vector<float> dxf((size_t)B * n), dzf((size_t)B * m), dyf((size_t)B * q);
        std::vector<double> facw((size_t)B * qpx_factor_elems(QPX_F32_WIDE, n, m, q));
        const int64_t few = (int64_t)qpx_factor_elems(QPX_F32_WIDE, n, m, q);
        rc = qpx_forward(QPX_F32_WIDE, B, n, m, q, Qf.data(), (int64_t)n * n, pf.data(), n, Gf.data(), (int64_t)m * n, hf.data(), m,
                         q ? Af.data() : nullptr, (int64_t)q * n, q ? bf.data() : nullptr, q, facw.data(), 1e-12, 20, 3, B == 1 ? 1 : 2,
                         zf.data(), q ? nuf.data() : nullptr, lamf.data(), slf.data(), iters.data(), status.data(), brf.data(), nullptr, nullptr);
        if (rc) { fprintf(stderr, "wide forward rc %d\n", rc); return 2; }
        rc = qpx_backward(QPX_F32_WIDE, B, n, m, q, facw.data(), few, zf.data(), lamf.data(), slf.data(), q ? nuf.data() : nullptr, gf.data(),
                          dQf.data(), dpf.data(), dGf.data(), dhf.data(), q ? dAf.data() : nullptr, q ? dbf.data() : nullptr,
                          dxf.data(), dzf.data(), q ? dyf.data() : nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, status.data(), nullptr);
        if (rc) { fprintf(stderr, "wide backward rc %d\n", rc); return 2; }
        double dz_max = 0, z_max = 0;
        for (size_t i = 0; i < zf.size(); ++i) { dz_max = std::fmax(dz_max, std::fabs((double)zf[i] - zhat[i])); z_max = std::fmax(z_max, std::fabs(zhat[i])); }
        printf("QPX_F32_WIDE: max |zhat - zhat_f64| %.2e (max |zhat| %.2e)\n", dz_max, z_max);
        if (!(dz_max <= 1e-3 * (1.0 + z_max))) return 5;
    }
    return 0;
}
