// TEST INFRASTRUCTURE ONLY: CPU restatement of the belief-space iLQR backward pass,
// /root/reference/trajopt/bspilqr/src/util.cpp:83-204 (called at bspilqr/bspilqr.py:90-104).
// Parity unpinned (no vectors in the reference, Armadillo absent); pinned by the NumPy mirror, long double and the
// reduction to the iLQR pass when T..Z vanish (tests/test_oracle_known_answers.py).
// Reference quirks: dS is not zero-initialised (:132; zero here) and line :158 reads tau.col(i) BEFORE it is
// assigned at :181 (uninitialised memory).  tau_mode 0 restates the evident intent, tau.col(i+1), like :152 and :155;
// tau_mode 1 takes that uninitialised column as zeros (what a fresh allocation usually holds).
#pragma once
#include "riccati_oracle.hpp"

namespace oracle {

template <class S>
int bsp_backward_pass(const double* Q, const double* q, const double* R, const double* r, const double* P,
                      const double* p, const double* F, const double* G, const double* Tm, const double* U,
                      const double* V, const double* X, const double* Y, const double* Z, double lmbda, int reg,
                      int tau_mode, int b, int m, int T, double* Sout, double* sout, double* tauout, double* dS,
                      double* K, double* kff, int* failed) {
    const int bb = b * b;
    const size_t nbb = (size_t)b * b, nbm = (size_t)b * m, nmm = (size_t)m * m;
    *failed = 0;
    M<S> Sn = load<S>(Q + nbb * T, b, b), sn = load<S>(q + (size_t)b * T, b, 1), taun = load<S>(p + (size_t)bb * T, bb, 1);
    store(Sout + nbb * T, Sn);
    store(sout + (size_t)b * T, sn);
    store(tauout + (size_t)bb * T, taun);
    S dS0 = 0, dS1 = 0;
    int diverge = 0;
    for (int i = T - 1; i >= 0; --i) {
        M<S> Fi = load<S>(F + nbb * i, b, b), Gi = load<S>(G + nbm * i, b, m), Ft = tr(Fi), Gt = tr(Gi);
        M<S> Ri = load<S>(R + nmm * i, m, m), Pi = load<S>(P + nbm * i, b, m);
        M<S> C = add(load<S>(Q + nbb * i, b, b), mul3(Ft, Sn, Fi));                        // :148
        M<S> D = add(Ri, mul3(Gt, Sn, Gi));                                                // :149
        M<S> E = tr(add(Pi, mul3(Ft, Sn, Gi)));                                            // :150
        M<S> vS(bb, 1);
        for (int k = 0; k < bb; ++k) vS.a[k] = Sn.a[k];                                    // vectorise(): column-major
        auto lin = [&](const double* base, const double* Lm, const double* Tm_, const double* Xm, int rows) {
            // base + L' s + T' tau + 0.5 X' vec(S), as :152-156
            M<S> o = load<S>(base, rows, 1);
            o = add(o, mul(tr(load<S>(Lm, b, rows)), sn));
            o = add(o, mul(tr(load<S>(Tm_, bb, rows)), taun));
            o = add(o, scale(mul(tr(load<S>(Xm, bb, rows)), vS), S(0.5)));
            return o;
        };
        M<S> c = lin(q + (size_t)b * i, F + nbb * i, Tm + (size_t)bb * b * i, X + (size_t)bb * b * i, b);
        M<S> d = lin(r + (size_t)m * i, G + nbm * i, V + (size_t)bb * m * i, Z + (size_t)bb * m * i, m);
        M<S> e = load<S>(p + (size_t)bb * i, bb, 1);                                       // :158
        if (tau_mode == 0) e = add(e, mul(tr(load<S>(U + (size_t)bb * bb * i, bb, bb)), taun));
        e = add(e, scale(mul(tr(load<S>(Y + (size_t)bb * bb * i, bb, bb)), vS), S(0.5)));
        M<S> Sreg = Sn;                                                                    // :160-162
        if (reg == 2)
            for (int k = 0; k < b; ++k) Sreg(k, k) += (S)lmbda;
        M<S> Ereg = tr(add(Pi, mul3(Ft, Sreg, Gi)));                                       // :164
        M<S> Dreg = add(Ri, mul3(Gt, Sreg, Gi));                                           // :166-168
        if (reg == 1)
            for (int k = 0; k < m; ++k) Dreg(k, k) += (S)lmbda;
        if (!is_sympd(Dreg)) {                                                             // :170-173
            diverge = i;
            *failed = 1;
            break;
        }
        M<S> Dinv;
        if (!gj_solve(Dreg, eye_m<S>(m), Dinv)) { *failed = 2; diverge = i; break; }       // :175
        M<S> Ki = scale(mul(Dinv, Ereg), S(-1)), ki = scale(mul(Dinv, d), S(-1));          // :176-177
        dS0 += mul(tr(ki), d)(0, 0);                                                       // :179
        dS1 += S(0.5) * mul3(tr(ki), D, ki)(0, 0);
        M<S> Kt = tr(Ki), Et = tr(E);
        M<S> si = add(add(add(c, mul3(Kt, D, ki)), mul(Kt, d)), mul(Et, ki));              // :183-184
        M<S> Si = add(add(add(C, mul3(Kt, D, Ki)), mul(Kt, E)), mul(Et, Ki));              // :186-187
        Si = scale(add(Si, tr(Si)), S(0.5));                                               // :188
        store(K + nbm * i, Ki);
        store(kff + (size_t)m * i, ki);
        store(Sout + nbb * i, Si);
        store(sout + (size_t)b * i, si);
        store(tauout + (size_t)bb * i, e);                                                 // :181
        Sn = Si; sn = si; taun = e;
    }
    dS[0] = (double)dS0;
    dS[1] = (double)dS1;
    return diverge;
}

}  // namespace oracle
