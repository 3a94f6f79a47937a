// ORACLE - TEST INFRASTRUCTURE ONLY (see riccati_oracle.hpp for the rules).  PARITY UNPINNED.
//
// extern "C" surface of the CPU oracle, loaded through ctypes by oracle/oracle.py.
// Every array is column-major with time slowest (the reference layout, util.cpp:10,21); batched
// entry points take `nb_problems` such problems stacked back to back ("problem-major").
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>
#include "ilqr_oracle.hpp"
#include "gps_oracle.hpp"
#include "bsp_oracle.hpp"

using namespace oracle;

extern "C" {

// util.cpp:83-182 in double.  Outputs must be zero-filled by the caller.  Returns the reference's
// `_diverge`; *failed gets the unambiguous flag.
int oracle_backward_pass(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                         const double* Cxu, const double* A, const double* B, double lmbda, int reg, int n,
                         int m, int T, double* Qxx, double* Qux, double* Quu, double* qx, double* qu, double* V,
                         double* v, double* dV, double* K, double* kff, int* failed) {
    return backward_pass<double>(Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, n, m, T, Qxx, Qux, Quu, qx, qu, V, v, dV,
                                 K, kff, failed);
}

// Same recursion carried in x87 long double (64-bit mantissa): the arbiter that shows which of two
// double implementations is closer to the exact recursion, and that 1e-9 is a meaningful bar.
int oracle_backward_pass_ld(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                            const double* Cxu, const double* A, const double* B, double lmbda, int reg, int n,
                            int m, int T, double* Qxx, double* Qux, double* Quu, double* qx, double* qu,
                            double* V, double* v, double* dV, double* K, double* kff, int* failed) {
    return backward_pass<long double>(Cxx, cx, Cuu, cu, Cxu, A, B, lmbda, reg, n, m, T, Qxx, Qux, Quu, qx, qu, V,
                                      v, dV, K, kff, failed);
}

int oracle_riccati_backward_pass(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                 const double* Cxu, const double* A, const double* B, const double* c, int n,
                                 int m, int T, int literal_elementwise, double* V, double* v, double* K,
                                 double* kff) {
    return riccati_backward_pass<double>(Cxx, cx, Cuu, cu, Cxu, A, B, c, n, m, T, literal_elementwise, V, v, K,
                                         kff);
}
int oracle_riccati_backward_pass_ld(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                    const double* Cxu, const double* A, const double* B, const double* c, int n,
                                    int m, int T, int literal_elementwise, double* V, double* v, double* K,
                                    double* kff) {
    return riccati_backward_pass<long double>(Cxx, cx, Cuu, cu, Cxu, A, B, c, n, m, T, literal_elementwise, V, v,
                                              K, kff);
}

// gps/src/util.cpp:272-374; outputs zero-filled by the caller
int oracle_gps_backward_pass(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                             const double* Cxu, const double* c0, const double* A, const double* B, const double* c,
                             const double* sigma_dyn, const double* alpha, int n, int m, int T, int long_double,
                             double* Qxx, double* Qux, double* Quu, double* qx, double* qu, double* q0, double* V,
                             double* v, double* v0, double* K, double* kff, double* sigma_ctl, int* failed) {
    if (long_double)
        return gps_backward_pass<long double>(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha, n, m, T, Qxx, Qux,
                                              Quu, qx, qu, q0, V, v, v0, K, kff, sigma_ctl, failed);
    return gps_backward_pass<double>(Cxx, cx, Cuu, cu, Cxu, c0, A, B, c, sigma_dyn, alpha, n, m, T, Qxx, Qux, Quu, qx,
                                     qu, q0, V, v, v0, K, kff, sigma_ctl, failed);
}

// bspilqr/src/util.cpp:83-204; outputs zero-filled by the caller
int oracle_bsp_backward_pass(const double* Q, const double* q, const double* R, const double* r, const double* P,
                             const double* p, const double* F, const double* G, const double* T_, const double* U,
                             const double* V, const double* X, const double* Y, const double* Z, double lmbda, int reg,
                             int tau_mode, int b, int m, int T, int long_double, double* S, double* s, double* tau,
                             double* dS, double* K, double* kff, int* failed) {
    if (long_double)
        return bsp_backward_pass<long double>(Q, q, R, r, P, p, F, G, T_, U, V, X, Y, Z, lmbda, reg, tau_mode, b, m, T,
                                              S, s, tau, dS, K, kff, failed);
    return bsp_backward_pass<double>(Q, q, R, r, P, p, F, G, T_, U, V, X, Y, Z, lmbda, reg, tau_mode, b, m, T, S, s, tau,
                                     dS, K, kff, failed);
}

// gps/src/util.cpp:83-269
int oracle_gps_kl_divergence(const double* pK, const double* pk, const double* pS, const double* qK, const double* qk,
                             const double* qS, const double* mu_x, const double* sigma_x, int n, int m, int T,
                             int long_double, double* kl) {
    if (long_double) return gps_kl_divergence<long double>(pK, pk, pS, qK, qk, qS, mu_x, sigma_x, n, m, T, kl);
    return gps_kl_divergence<double>(pK, pk, pS, qK, qk, qS, mu_x, sigma_x, n, m, T, kl);
}
double oracle_gps_quad_expectation(const double* mu, const double* sigma, const double* Q, const double* q, double q0,
                                   int n) {
    return gps_quad_expectation<double>(mu, sigma, Q, q, q0, n);
}
int oracle_gps_augment_cost(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                            const double* c0, const double* K, const double* kff, const double* sigma_ctl,
                            const double* alpha, int n, int m, int T, int long_double, double* agCxx, double* agcx,
                            double* agCuu, double* agcu, double* agCxu, double* agc0) {
    if (long_double)
        return gps_augment_cost<long double>(Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, n, m, T, agCxx, agcx,
                                             agCuu, agcu, agCxu, agc0);
    return gps_augment_cost<double>(Cxx, cx, Cuu, cu, Cxu, c0, K, kff, sigma_ctl, alpha, n, m, T, agCxx, agcx, agCuu,
                                    agcu, agCxu, agc0);
}
void oracle_gps_forward_pass(const double* mu_x0, const double* sigma_x0, const double* A, const double* B,
                             const double* c, const double* sigma_dyn, const double* K, const double* kff,
                             const double* sigma_ctl, int n, int m, int T, int long_double, double* mu_x,
                             double* sigma_x, double* mu_u, double* sigma_u, double* mu_xu, double* sigma_xu) {
    if (long_double)
        gps_forward_pass<long double>(mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, n, m, T, mu_x, sigma_x,
                                      mu_u, sigma_u, mu_xu, sigma_xu);
    else
        gps_forward_pass<double>(mu_x0, sigma_x0, A, B, c, sigma_dyn, K, kff, sigma_ctl, n, m, T, mu_x, sigma_x, mu_u,
                                 sigma_u, mu_xu, sigma_xu);
}

void oracle_env_dynamics(const trajopt_env_t* e, const double* x, const double* u, double* xn) {
    dynamics<double>(*e, x, u, xn);
}
double oracle_env_cost(const trajopt_env_t* e, const double* x, const double* u, const double* u_last, double a) {
    return cost(*e, x, u, u_last, a);
}
int oracle_env_jacobians(const trajopt_env_t* e, const double* x, const double* u, double* A, double* B) {
    return dynamics_jacobians(*e, x, u, A, B, nullptr);
}
void oracle_forward_pass(const trajopt_env_t* e, int T, const double* x0, const double* K, const double* kff,
                         double alpha, const double* xref, const double* uref, const double* weighting,
                         double* state, double* action, double* cost_out) {
    forward_pass(*e, T, x0, K, kff, alpha, xref, uref, weighting, state, action, cost_out);
}
int oracle_quadratize(const trajopt_env_t* e, int T, const double* xref, const double* uref,
                      const double* weighting, double* A, double* B, double* Cxx, double* cx, double* Cuu,
                      double* cu, double* Cxu) {
    return quadratize(*e, T, xref, uref, weighting, A, B, Cxx, cx, Cuu, cu, Cxu);
}
int oracle_quadratize_riccati(const trajopt_env_t* e, int T, const double* xref, const double* uref,
                              const double* weighting, double* A, double* B, double* c, double* Cxx, double* cx,
                              double* Cuu, double* cu, double* Cxu) {
    return quadratize_riccati(*e, T, xref, uref, weighting, A, B, c, Cxx, cx, Cuu, cu, Cxu);
}

// ilqr.py:136-237 for one problem.  opts = {lmbda, dlmbda, min_lmbda, max_lmbda, mult_lmbda, tolfun,
// tolgrad, min_imp}.  trace has room for nb_iter + 1 entries.  scal_out = {last_return, lmbda, dlmbda,
// alpha, iters, status, trace_len}.
int oracle_ilqr_run(const trajopt_env_t* e, int T, const double* x0, const double* kff_init,
                    const double* weighting, const double* alphas, int nb_alphas, const double* opts, int reg,
                    int nb_iter, double* xref, double* uref, double* K, double* kff, double* V, double* v,
                    double* trace, double* scal_out) {
    ILQROptions o;
    o.lmbda = opts[0]; o.dlmbda = opts[1]; o.min_lmbda = opts[2]; o.max_lmbda = opts[3]; o.mult_lmbda = opts[4];
    o.tolfun = opts[5]; o.tolgrad = opts[6]; o.min_imp = opts[7]; o.reg = reg;
    ILQRSolution s = ilqr_run(*e, T, x0, kff_init, weighting, alphas, nb_alphas, o, nb_iter);
    std::copy(s.xref.begin(), s.xref.end(), xref);
    std::copy(s.uref.begin(), s.uref.end(), uref);
    if (K) std::copy(s.K.begin(), s.K.end(), K);
    if (kff) std::copy(s.kff.begin(), s.kff.end(), kff);
    if (V) std::copy(s.V.begin(), s.V.end(), V);
    if (v) std::copy(s.v.begin(), s.v.end(), v);
    if (trace) std::copy(s.trace.begin(), s.trace.end(), trace);
    if (scal_out) {
        scal_out[0] = s.last_return; scal_out[1] = s.lmbda; scal_out[2] = s.dlmbda; scal_out[3] = s.alpha;
        scal_out[4] = s.iters; scal_out[5] = s.status; scal_out[6] = (double)s.trace.size();
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// CPU baseline legs (bench.py cpu_baseline): one problem per thread, mirroring the reference's
// one-process-per-problem joblib fan-out (examples/ilqr/mpc/pendulum_parallel.py:40-45).
// Inputs are problem-major stacks.  Returns elapsed seconds.
// ---------------------------------------------------------------------------------------------
double oracle_bench_backward_pass(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                                  const double* Cxu, const double* A, const double* B, double lmbda, int reg,
                                  int n, int m, int T, int64_t nb_problems, int nb_threads, double* K_out,
                                  double* kff_out) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        std::vector<double> Qxx(nn * T), Qux(nm * T), Quu(mm * T), qx((size_t)n * T), qu((size_t)m * T),
            V(nn * (T + 1)), v((size_t)n * (T + 1));
        double dV[2];
        int failed;
        for (;;) {
            const int64_t p = next.fetch_add(1);
            if (p >= nb_problems) break;
            backward_pass<double>(Cxx + p * nn * (T + 1), cx + p * (size_t)n * (T + 1), Cuu + p * mm * (T + 1),
                                  cu + p * (size_t)m * (T + 1), Cxu + p * nm * (T + 1), A + p * nn * T,
                                  B + p * nm * T, lmbda, reg, n, m, T, Qxx.data(), Qux.data(), Quu.data(),
                                  qx.data(), qu.data(), V.data(), v.data(), dV, K_out + p * nm * T,
                                  kff_out + p * (size_t)m * T, &failed);
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < nb_threads; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// One full iLQR iteration budget: runs `nb_iter` iterations of ilqr.py:151-231 per problem from the
// given initial states.  Returns elapsed seconds; iters_done gets the total iterations executed.
double oracle_bench_ilqr(const trajopt_env_t* e, int T, const double* x0s, const double* kff_inits,
                         const double* weighting, const double* alphas, int nb_alphas, const double* opts, int reg,
                         int nb_iter, int64_t nb_problems, int nb_threads, double* returns_out,
                         int64_t* iters_done) {
    ILQROptions o;
    o.lmbda = opts[0]; o.dlmbda = opts[1]; o.min_lmbda = opts[2]; o.max_lmbda = opts[3]; o.mult_lmbda = opts[4];
    o.tolfun = opts[5]; o.tolgrad = opts[6]; o.min_imp = opts[7]; o.reg = reg;
    const int n = e->dm_state, m = e->dm_act;
    std::atomic<int64_t> next(0), iters(0);
    auto work = [&]() {
        for (;;) {
            const int64_t p = next.fetch_add(1);
            if (p >= nb_problems) break;
            ILQRSolution s = ilqr_run(*e, T, x0s + p * n, kff_inits + p * (size_t)m * T, weighting, alphas,
                                      nb_alphas, o, nb_iter);
            if (returns_out) returns_out[p] = s.last_return;
            iters.fetch_add(s.iters);
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < nb_threads; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
    if (iters_done) *iters_done = iters.load();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
