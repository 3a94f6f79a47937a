// Device-side env models (row A8): dynamics and cost of the envs the configs name, written for the GPU.
//
//   pendulum         /root/reference/trajopt/envs/pendulum/pendulum.py:58-76, :111-125
//   cartpole         /root/reference/trajopt/envs/cartpole/cartpole.py:62-97, :112-127
//   double cartpole  /root/reference/trajopt/envs/double_cartpole/double_cartpole.py:61-155, :170-187
//   LQR (affine LTI) /root/reference/trajopt/envs/lqr/lqr_v0.py:50-66, :89-94
//   double pendulum  /root/reference/trajopt/envs/double_pendulum/double_pendulum.py:61-125, :140-154
//   quadrotor        build-defined 12-state rigid body (no counterpart in the reference)
//
// Everything is a template over the scalar type so the same vector field serves the rollout (double)
// and the quadratisation (Dual<K>: forward-mode AD carrying K tangent directions, which reproduces
// autograd's jacobian() of the discrete RK4+clip map to round-off; objects.py:108-117).
// phys[] per model:  pendulum {g, m, l, k}; cartpole {g, Mc, Mp, l, fr};
//                    double cartpole {g, Mc, Mp1, Mp2, L1, L2}; quadrotor {g, mass, L, Ixx, Iyy, Izz, kM};
//                    double pendulum {g, m1, m2, l1, l2, k1, k2}.
#pragma once
#include "common.h"
#include <type_traits>

namespace trajopt {

template <int K>
struct Dual {
    double v;
    double d[K];
};

// ---- scalar-generic arithmetic ------------------------------------------------------------------
__device__ __forceinline__ double dmake(double x) { return x; }
template <int K> __device__ __forceinline__ Dual<K> constant(double x) {
    Dual<K> r; r.v = x;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = 0.0;
    return r;
}
template <class X> struct Scalar;
template <> struct Scalar<double> {
    static __device__ __forceinline__ double c(double x) { return x; }
    static __device__ __forceinline__ double val(double x) { return x; }
};
template <int K> struct Scalar<Dual<K>> {
    static __device__ __forceinline__ Dual<K> c(double x) { return constant<K>(x); }
    static __device__ __forceinline__ double val(const Dual<K>& x) { return x.v; }
};

#define TJ_DUAL_BINOP(OP, VEXPR, DEXPR)                                                                      \
    template <int K> __device__ __forceinline__ Dual<K> operator OP(const Dual<K>& a, const Dual<K>& b) {    \
        Dual<K> r; r.v = VEXPR;                                                                              \
        _Pragma("unroll") for (int k = 0; k < K; ++k) r.d[k] = DEXPR;                                        \
        return r;                                                                                            \
    }
TJ_DUAL_BINOP(+, a.v + b.v, a.d[k] + b.d[k])
TJ_DUAL_BINOP(-, a.v - b.v, a.d[k] - b.d[k])
TJ_DUAL_BINOP(*, a.v * b.v, fma(a.d[k], b.v, a.v * b.d[k]))
#undef TJ_DUAL_BINOP
template <int K> __device__ __forceinline__ Dual<K> operator/(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r; const double ib = 1.0 / b.v; r.v = a.v * ib;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) * ib;
    return r;
}
template <int K> __device__ __forceinline__ Dual<K> operator-(const Dual<K>& a) {
    Dual<K> r; r.v = -a.v;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = -a.d[k];
    return r;
}
template <int K> __device__ __forceinline__ Dual<K> operator+(const Dual<K>& a, double b) { Dual<K> r = a; r.v += b; return r; }
template <int K> __device__ __forceinline__ Dual<K> operator+(double b, const Dual<K>& a) { Dual<K> r = a; r.v += b; return r; }
template <int K> __device__ __forceinline__ Dual<K> operator-(const Dual<K>& a, double b) { Dual<K> r = a; r.v -= b; return r; }
template <int K> __device__ __forceinline__ Dual<K> operator-(double b, const Dual<K>& a) { Dual<K> r = -a; r.v += b; return r; }
template <int K> __device__ __forceinline__ Dual<K> operator*(const Dual<K>& a, double b) {
    Dual<K> r; r.v = a.v * b;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = a.d[k] * b;
    return r;
}
template <int K> __device__ __forceinline__ Dual<K> operator*(double b, const Dual<K>& a) { return a * b; }
template <int K> __device__ __forceinline__ Dual<K> operator/(const Dual<K>& a, double b) { return a * (1.0 / b); }

// sin and cos together.  fp64 issue slots are what these kernels pay for (one fp64 op per ~5 cycles per SIMD; the
// library sincos is ~85 of them plus a Payne-Hanek path whose registers and branches sit in the hot loop).
// Here, 27 instructions and no branch:
//  * reduction by pi, k = rint(x / pi) taken from a magic-number add (the integer sits in the low mantissa bits, its
//    parity is bit 0 of the low word), r = x - k pi with a three-term Cody-Waite constant.  Each term is one FMA, and
//    the FMA is what lifts fdlibm's |x| < 2^20 pi/2 limit: k * pi_1 (33-bit constant) need not be representable, only
//    x - k pi_1 must be, and it is (it spans 2^-32 .. 2^21 for |k| < 2^51).  Measured against long double on 2 M random
//    arguments per decade: absolute error <= 4.3e-16 up to |x| = 1e12, 6e-16 at 1e15 (tests/aux/sincos_check.c);
//    beyond 2^51 pi the result is meaningless (finite garbage, then inf / NaN), as it is for inf / NaN arguments - such
//    a candidate has left every state bound long before and its return loses the line search either way
//    (ilqr.py:196 compares false on NaN, init_select_kernel refuses a non-finite return);
//  * on |r| <= pi/2 the fdlibm minimax kernels are evaluated at the HALF angle (their range is pi/4; the coefficients
//    below are fdlibm's divided by powers of four, i.e. exact, so that the polynomials take w = r^2 directly) and the
//    double-angle identities give sin r = 2 sin(r/2) cos(r/2), cos r = 1 - 2 sin^2(r/2): three more multiplications,
//    but both results share ONE sign, (-1)^k - no quadrant swap (four 32-bit selects), no second flip mask.
// Round 3: this form replaced reduction by pi/2 + swap + two flips + a ballot-guarded big-argument path (34
// instructions and a branch per call, absolute error 1.7e-16).
__device__ __forceinline__ void sincos_g(double x, double& s, double& c) {
    constexpr double kMagic = 6755399441055744.0;   // 1.5 * 2^52
    const double t = fma(x, 3.18309886183790691216e-01, kMagic);
    const double fk = t - kMagic;
    double r = fma(-fk, 2.0 * 1.57079632673412561417e+00, x);
    r = fma(-fk, 2.0 * 6.07710050630396597660e-11, r);
    r = fma(-fk, 2.0 * 2.02226624879595063154e-21, r);
    const double w = r * r;
    double ps = fma(w, 1.58969099521155010221e-10 / 4096.0, -2.50507602534068634195e-08 / 1024.0);
    double pc = fma(w, -1.13596475577881948265e-11 / 16384.0, 2.08757232129817482790e-09 / 4096.0);
    ps = fma(w, ps, 2.75573137070700676789e-06 / 256.0);
    pc = fma(w, pc, -2.75573143513906633035e-07 / 1024.0);
    ps = fma(w, ps, -1.98412698298579493134e-04 / 64.0);
    pc = fma(w, pc, 2.48015872894767294178e-05 / 256.0);
    ps = fma(w, ps, 8.33333333332248946124e-03 / 16.0);
    pc = fma(w, pc, -1.38888888888741095749e-03 / 64.0);
    ps = fma(w, ps, -1.66666666666666324348e-01 / 4.0);
    pc = fma(w, pc, 4.16666666666666019037e-02 / 16.0);
    const double sh2 = fma(r * w, ps, r);                       // 2 sin(r / 2)
    const double ch = fma(w * w, pc, fma(-0.125, w, 1.0));      // cos(r / 2)
    const double sa = sh2 * ch;
    const double ca = fma(sh2 * sh2, -0.5, 1.0);
    const int flip = (int)((unsigned)__double2loint(t) << 31);  // k odd: both change sign
    s = __hiloint2double(__double2hiint(sa) ^ flip, __double2loint(sa));
    c = __hiloint2double(__double2hiint(ca) ^ flip, __double2loint(ca));
}
template <int K> __device__ __forceinline__ void sincos_g(const Dual<K>& x, Dual<K>& s, Dual<K>& c) {
    double sv, cv;
    sincos_g(x.v, sv, cv);
    s.v = sv; c.v = cv;
#pragma unroll
    for (int k = 0; k < K; ++k) { s.d[k] = cv * x.d[k]; c.d[k] = -sv * x.d[k]; }
}

// 1 / x: one division that the models reuse instead of dividing several times by the same quantity
__device__ __forceinline__ double recip_g(double x) { return 1.0 / x; }
// 1 / x for a quantity known to be positive and well scaled (a pivot of a mass matrix): refined hardware reciprocal,
// ~1 ulp, 5 instructions instead of the 13 of an IEEE division - the rollout is a chain of dependent instructions
__device__ __forceinline__ double recip_pos(double x) { return rcp_nr(x); }
template <int K> __device__ __forceinline__ Dual<K> recip_g(const Dual<K>& x);
template <int K> __device__ __forceinline__ Dual<K> recip_pos(const Dual<K>& x) { return recip_g(x); }
template <int K> __device__ __forceinline__ Dual<K> recip_g(const Dual<K>& x) {
    Dual<K> r; r.v = 1.0 / x.v;
    const double m2 = -r.v * r.v;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = m2 * x.d[k];
    return r;
}

// np.clip(x, -lim, lim); derivative follows autograd: g * (ans != a_min) * (ans != a_max)
__device__ __forceinline__ double clip_g(double x, double lim) { return fmin(fmax(x, -lim), lim); }
template <int K> __device__ __forceinline__ Dual<K> clip_g(const Dual<K>& x, double lim) {
    Dual<K> r; r.v = fmin(fmax(x.v, -lim), lim);
    const double mask = (r.v != -lim && r.v != lim) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = mask * x.d[k];
    return r;
}

// ---- vector fields --------------------------------------------------------------------------------
template <int MODEL> struct Model;

template <> struct Model<TRAJOPT_ENV_PENDULUM> {
    static constexpr int N = 2, M = 1;
    template <class X> static __device__ __forceinline__ void f(const trajopt_env_t& e, const X (&x)[N], const X (&u)[M], X (&dx)[N]) {
        const double g = e.phys[0], m = e.phys[1], l = e.phys[2], k = e.phys[3];
        X s, c;
        sincos_g(x[0] + M_PI, s, c);
        dx[0] = x[1];
        dx[1] = (-3.0 * g / (2.0 * l)) * s + (3.0 / (m * l * l)) * (u[0] - k * x[1]);
    }
};

template <> struct Model<TRAJOPT_ENV_CARTPOLE> {
    static constexpr int N = 4, M = 1;
    template <class X> static __device__ __forceinline__ void f(const trajopt_env_t& e, const X (&x)[N], const X (&u)[M], X (&dx)[N]) {
        const double g = e.phys[0], Mc = e.phys[1], Mp = e.phys[2], l = e.phys[3], fr = e.phys[4];
        const double Mt = Mc + Mp, iMt = 1.0 / Mt;   // loop-invariant: hoisted out of the time loop
        // cartpole.py:62-97 with 1 / Mt folded into the constants (every instruction of this chain is paid for in full by a
        // lone wave: 12 fewer a step than the expressions as written, same values to round-off):
        //   num = g sin + cos (-(u - fr dq) - Mp l dth^2 sin) / Mt,  den = l (4/3 - Mp cos^2 / Mt) > 0
        const double friMt = fr * iMt, MpliMt = (Mp * l) * iMt, kl = l * (Mp * iMt), l43 = l * (4.0 / 3.0);
        X sth, cth;
        sincos_g(x[1], sth, cth);
        const X dq = x[2], dth = x[3];
        const X uim = u[0] * iMt;               // (the same value in all four RK4 stages: computed once)
        const X dth2s = dth * dth * sth;
        const X b = (friMt * dq - uim) - MpliMt * dth2s;
        const X num = g * sth + cth * b;
        const X den = l43 - (kl * cth) * cth;
        const X ddth = num * recip_pos(den);
        dx[0] = dq;
        dx[1] = dth;
        dx[2] = uim + MpliMt * (dth2s - ddth * cth);
        dx[3] = ddth;
    }
};

template <> struct Model<TRAJOPT_ENV_DOUBLE_CARTPOLE> {
    static constexpr int N = 6, M = 1;
    template <class X> static __device__ __forceinline__ void f(const trajopt_env_t& e, const X (&x)[N], const X (&u)[M], X (&dx)[N]) {
        const double g = e.phys[0], Mc = e.phys[1], Mp1 = e.phys[2], Mp2 = e.phys[3], L1 = e.phys[4], L2 = e.phys[5];
        const double Mt = Mc + Mp1 + Mp2, l1 = 0.5 * L1, l2 = 0.5 * L2, J1 = Mp1 * L1 / 12.0, J2 = Mp2 * L2 / 12.0;
        const double a = Mp1 * l1 + Mp2 * L2, b = Mp2 * l2, d = L1 * l2 * Mp2;
        const double m22 = l1 * l1 * Mp1 + L1 * L1 * Mp2 + J1, m33 = l2 * l2 * Mp2 + J2;
        X s1, c1, s2, c2, sd, cd;
        sincos_g(x[1], s1, c1);
        sincos_g(x[2], s2, c2);
        // sin / cos of th1 - th2 (double_cartpole.py:61-187 takes them of the difference): for plain numbers from the
        // subtraction identities - four instructions instead of a third sincos of 27, a seventh of the rollout's step, at
        // an absolute error of ~1e-15; for dual numbers four products of duals would cost more than the sincos does
        if constexpr (std::is_same<X, double>::value) {
            sd = s1 * c2 - c1 * s2;
            cd = c1 * c2 + s1 * s2;
        } else {
            sincos_g(x[1] - x[2], sd, cd);
        }
        const X dth1 = x[4], dth2 = x[5];
        // symmetric inertia matrix  [[Mt, m12, m13], [m12, m22, m23], [m13, m23, m33]]
        const X m12 = a * c1, m13 = b * c2, m23 = d * cd;
        // rhs = [u, 0, 0] - C dq - G
        const X r0 = u[0] + (a * dth1) * dth1 * s1 + (b * dth2) * dth2 * s2;
        const X r1 = (a - Mp2 * L2 + Mp2 * L1) * g * s1 - (d * dth2) * dth2 * sd;   // -(C23 dth2) - G2
        const X r2 = (b * g) * s2 + (d * dth1) * dth1 * sd;                          // -(C32 dth1) - G3
        // solve M dd = r by symmetric elimination (M is SPD: a mass matrix); no pivoting needed
        const double i11 = 1.0 / Mt;                    // Mt is a constant
        const X l21 = m12 * i11, l31 = m13 * i11;
        const X a22 = m22 - l21 * m12, a23 = m23 - l21 * m13, a33 = m33 - l31 * m13;
        const X b1 = r1 - l21 * r0, b2 = r2 - l31 * r0;
        const X i22 = recip_pos(a22);
        const X l32 = a23 * i22;
        const X c33 = a33 - l32 * a23;
        const X y2 = (b2 - l32 * b1) * recip_pos(c33);
        const X y1 = (b1 - a23 * y2) * i22;
        const X y0 = (r0 - m12 * y1 - m13 * y2) * i11;
        dx[0] = x[3]; dx[1] = dth1; dx[2] = dth2; dx[3] = y0; dx[4] = y1; dx[5] = y2;
    }
};

template <> struct Model<TRAJOPT_ENV_DOUBLE_PENDULUM> {
    static constexpr int N = 4, M = 2;
    template <class X> static __device__ __forceinline__ void f(const trajopt_env_t& e, const X (&x)[N], const X (&u)[M], X (&dx)[N]) {
        // double_pendulum.py:66-113 (two torque-driven links, "PolicySearchToolbox" model; th1 = 0 is upright)
        const double g = e.phys[0], m1 = e.phys[1], m2 = e.phys[2], l1 = e.phys[3], l2 = e.phys[4], k1 = e.phys[5],
                     k2 = e.phys[6];
        const double I1 = m1 * (l1 * l1 + 1e-4) / 3.0, I2 = m2 * (l2 * l2 + 1e-4) / 3.0;
        const double l1c = 0.5 * l1, l2c = 0.5 * l2;
        const double a = l1 * l2c * m2;                       // coupling
        const double h22 = I2 + l2c * l2c * m2;               // constant entries
        const double h11c = I1 + I2 + l1c * l1c * m1 + l1 * l1 * m2 + l2c * l2c * m2;
        const double g1 = g * (l1c * m1 + l1 * m2), g2 = g * l2c * m2;
        X s1, c1, s2, c2;
        sincos_g(x[0] + M_PI, s1, c1);
        sincos_g(x[1], s2, c2);
        const X dth1 = x[2], dth2 = x[3];
        const X h11 = h11c + (2.0 * a) * c2;
        const X h12 = h22 + a * c2;                            // = h21
        const X gsum = g2 * (c2 * s1 + c1 * s2);               // g l2c m2 (c2 s1 + c1 s2)
        const X b1 = g1 * s1 + gsum - a * s2 * (2.0 * dth1 * dth2 + dth2 * dth2);
        const X b2 = gsum + a * s2 * (dth1 * dth1);
        const X r1 = (u[0] - k1 * dth1) - b1, r2 = (u[1] - k2 * dth2) - b2;
        const X idet = recip_pos(h11 * h22 - h12 * h12);       // det of a mass matrix: positive
        dx[0] = dth1;
        dx[1] = dth2;
        dx[2] = (h22 * r1 - h12 * r2) * idet;
        dx[3] = (h11 * r2 - h12 * r1) * idet;
    }
};

template <> struct Model<TRAJOPT_ENV_QUAD_PENDULUM> {
    static constexpr int N = 8, M = 4;
    // quad_pendulum.py:62-228: four torque-driven links (the PolicySearchToolbox / Atkeson model).  The reference
    // inverts the 4 x 4 inertia matrix by cofactors (:188-222); here M dd = r is solved by symmetric elimination
    // (M is a mass matrix: positive definite), which is the same solution at a third of the arithmetic.
    // phys = {G, m1..m4, l1..l4, friction1..4}
    template <class X> static __device__ __forceinline__ void f(const trajopt_env_t& e, const X (&x)[N], const X (&u)[M], X (&dx)[N]) {
        const double G = e.phys[0], m1 = e.phys[1], m2 = e.phys[2], m3 = e.phys[3], m4 = e.phys[4];
        const double l1 = e.phys[5], l2 = e.phys[6], l3 = e.phys[7], l4 = e.phys[8];
        const double fr1 = e.phys[9], fr2 = e.phys[10], fr3 = e.phys[11], fr4 = e.phys[12];
        const double I1 = m1 * (l1 * l1 + 1e-4) / 3.0, I2 = m2 * (l2 * l2 + 1e-4) / 3.0, I3 = m3 * (l3 * l3 + 1e-4) / 3.0,
                     I4 = m4 * (l4 * l4 + 1e-4) / 3.0;
        const double h1 = 0.5 * l1, h2 = 0.5 * l2, h3 = 0.5 * l3, h4 = 0.5 * l4;   // centres of mass
        // lumped constants: w4 = l4cm m4, w3 = l3cm m3 + l3 m4, w2 = l2cm m2 + l2 (m3 + m4), w1 = l1cm m1 + l1 (m2 + m3 + m4)
        const double w4 = h4 * m4, w3 = h3 * m3 + l3 * m4, w2 = h2 * m2 + l2 * (m3 + m4), e8 = l1 * (m2 + m3 + m4);
        const X w1 = x[4], wv2 = x[5], wv3 = x[6], wv4 = x[7];   // angular velocities
        X s1, c1, s2, c2, s3, c3, s4, c4;
        sincos_g(x[0] + M_PI, s1, c1);   // downward position = pi (quad_pendulum.py:77)
        sincos_g(x[1], s2, c2);
        sincos_g(x[2], s3, c3);
        sincos_g(x[3], s4, c4);
        // sums of angles by the addition formulas, as the reference forms them (:92-103)
        const X s12 = s1 * c2 + c1 * s2, c12 = c1 * c2 - s1 * s2, s23 = s2 * c3 + c2 * s3, c23 = c2 * c3 - s2 * s3;
        const X s34 = s3 * c4 + c3 * s4, c34 = c3 * c4 - s3 * s4;
        const X s1234 = s12 * c34 + c12 * s34, s123 = s12 * c3 + c12 * s3, s234 = s2 * c34 + c2 * s34, c234 = c2 * c34 - s2 * s34;
        // inertia matrix (symmetric; the reference's a..p, :147-181)
        const double M44 = I4 + h4 * w4;
        const X M34 = M44 + (l3 * w4) * c4;
        const X M24 = M34 + (l2 * w4) * c34;
        const X M14 = M24 + (l1 * w4) * c234;
        const X M33 = I3 + M34 + h3 * (h3 * m3) + l3 * (l3 * m4) + (l3 * w4) * c4;
        const X M23 = M33 + (l2 * w3) * c3 + (l2 * w4) * c34;
        const X M13 = M23 + (l1 * w3) * c23 + (l1 * w4) * c234;
        const X M22 = M23 + I2 + h2 * (h2 * m2) + (l2 * l2) * (m3 + m4) + (l2 * w3) * c3 + (l2 * w4) * c34;
        const X M12 = M22 + M13 - M23 + (l1 * w2) * c2;
        const X M11 = 2.0 * M12 + I1 - M22 + (h1 * h1) * m1 + l1 * e8;
        // velocity products and generalised forces (:105-145, :152-186)
        const X v12 = w1 + wv2, v123 = v12 + wv3, q11 = w1 * w1, q12 = v12 * v12;
        const X e4b = 2.0 * w1 * wv3 + 2.0 * wv2 * wv3 + wv3 * wv3;
        const X e4a = 2.0 * w1 * wv4 + 2.0 * wv2 * wv4 + 2.0 * wv3 * wv4 + wv4 * wv4;
        const X e9a = 2.0 * w1 * wv2 + wv2 * wv2, e9 = e9a + e4b;
        const X e1 = G * (s123 * w3 + s1234 * w4);
        const X e2 = (2.0 * v123 + wv4) * wv4 * (l3 * w4) * s4;
        const X e3 = (G * w2) * s12;
        const X e4 = (e4b + e4a) * ((l2 * w4) * s34);
        const X e5a = q11 * l1 * s234, e5 = w4 * e5a;
        const X e6 = e4b * (l2 * w3) * s3;
        const X ls23 = (l1 * w3) * s23;
        const X r4 = u[3] - fr4 * wv4 - w4 * (v123 * v123 * l3 * s4 + q12 * l2 * s34 + e5a + G * s1234);
        const X r3 = u[2] - fr3 * wv3 - ((q12 * (l2 * w3) * s3 + q11 * ls23) - e2 + q12 * ((l2 * w4) * s34) + e5 + e1);
        const X r2 = u[1] - fr2 * wv2 - (q11 * (l1 * w2) * s2 - e6 + q11 * ls23 - e2 - e4 + e5 + e3 + e1);
        const X r1 = u[0] - fr1 * w1 - (-e9a * (l1 * w2) * s2 - e6 - e9 * ls23 - e2 - e4 - (e9 + e4a) * (l1 * w4) * s234 + e3 +
                                        G * (h1 * m1 + e8) * s1 + e1);
        // L D L' elimination of [[M11 M12 M13 M14], [., M22 M23 M24], [., ., M33 M34], [., ., ., M44]]
        const X i1 = recip_pos(M11);
        const X a21 = M12 * i1, a31 = M13 * i1, a41 = M14 * i1;
        const X B22 = M22 - a21 * M12, B23 = M23 - a21 * M13, B24 = M24 - a21 * M14;
        const X B33 = M33 - a31 * M13, B34 = M34 - a31 * M14, B44 = M44 - a41 * M14;
        const X y2 = r2 - a21 * r1, y3 = r3 - a31 * r1, y4 = r4 - a41 * r1;
        const X i2 = recip_pos(B22);
        const X a32 = B23 * i2, a42 = B24 * i2;
        const X C33 = B33 - a32 * B23, C34 = B34 - a32 * B24, C44 = B44 - a42 * B24;
        const X z3 = y3 - a32 * y2, z4 = y4 - a42 * y2;
        const X i3 = recip_pos(C33);
        const X a43 = C34 * i3;
        const X D44 = C44 - a43 * C34;
        const X d4 = (z4 - a43 * z3) * recip_pos(D44);
        const X d3 = (z3 - C34 * d4) * i3;
        const X d2 = (y2 - B23 * d3 - B24 * d4) * i2;
        const X d1 = (r1 - M12 * d2 - M13 * d3 - M14 * d4) * i1;
        dx[0] = w1; dx[1] = wv2; dx[2] = wv3; dx[3] = wv4;
        dx[4] = d1; dx[5] = d2; dx[6] = d3; dx[7] = d4;
    }
};

template <> struct Model<TRAJOPT_ENV_ROBOT> {
    static constexpr int N = 4, M = 2;
    template <class X> static __device__ __forceinline__ void f(const trajopt_env_t& e, const X (&x)[N], const X (&u)[M], X (&dx)[N]) {
        // robot.py:53-61: x = [x, y, th, v], u = [a, phi];  phys = {l}
        const double il = 1.0 / e.phys[0];
        X s, c, sp, cp;
        sincos_g(x[2], s, c);
        sincos_g(u[1], sp, cp);
        dx[0] = x[3] * c;
        dx[1] = x[3] * s;
        dx[2] = x[3] * (sp * recip_g(cp)) * il;   // v tan(phi) / l
        dx[3] = u[0];
    }
};

// Models whose vector field does not read its first C states (the cart's position: cartpole.py:62-97,
// double_cartpole.py:61-187 - their dx[0] is a velocity): columns
// 0 .. C - 1 of the Jacobian of clip(RK4(x, clip(u))) are then e_c times the clip mask of xn[c] exactly - x[c] enters xn[c]
// by one addition, every other tangent with respect to it is a sum of products with zero - and the quadratisation pushes
// C tangent directions fewer through the dual-number arithmetic (sys.h: jacobian_cols).
template <int MODEL> struct CyclicLeadingStates { static constexpr int value = 0; };
template <> struct CyclicLeadingStates<TRAJOPT_ENV_CARTPOLE> { static constexpr int value = 1; };
template <> struct CyclicLeadingStates<TRAJOPT_ENV_DOUBLE_CARTPOLE> { static constexpr int value = 1; };
// (the quadrotor's three position states are cyclic too, but its 16 columns go in four chunks of four: the chunk that would
// shrink to one tangent direction then carries two forms of the dual arithmetic behind a branch, and the kernel as a whole
// got slower - 4096 quadrotors 1.91 -> 2.06 ms - so it keeps all of its columns)

// robot.py:62-67 integrates with x - dt/2 k, x - dt k and xn = x - dt/6 (...): RK4 with the step -dt.  Kept as is.
template <int MODEL> struct ReverseTime { static constexpr bool value = false; };
template <> struct ReverseTime<TRAJOPT_ENV_ROBOT> { static constexpr bool value = true; };

template <> struct Model<TRAJOPT_ENV_QUADROTOR> {
    static constexpr int N = 12, M = 4;
    template <class X> static __device__ __forceinline__ void f(const trajopt_env_t& e, const X (&x)[N], const X (&u)[M], X (&dx)[N]) {
        const double g = e.phys[0], im = 1.0 / e.phys[1], L = e.phys[2], Ixx = e.phys[3], Iyy = e.phys[4], Izz = e.phys[5],
                     kM = e.phys[6];
        X sph, cph, sth, cth, sps, cps;
        sincos_g(x[3], sph, cph);
        sincos_g(x[4], sth, cth);
        sincos_g(x[5], sps, cps);
        const X p = x[9], q = x[10], r = x[11];
        const X F = (u[0] + u[1] + u[2] + u[3]) * im;
        const X icth = recip_g(cth);
        const X tth = sth * icth;
        const X sq_cr = sph * q + cph * r;
        dx[0] = x[6]; dx[1] = x[7]; dx[2] = x[8];
        dx[3] = p + tth * sq_cr;
        dx[4] = cph * q - sph * r;
        dx[5] = sq_cr * icth;
        dx[6] = (cph * sth * cps + sph * sps) * F;
        dx[7] = (cph * sth * sps - sph * cps) * F;
        dx[8] = cph * cth * F - g;
        dx[9] = (L * (u[1] - u[3]) - (Izz - Iyy) * q * r) * (1.0 / Ixx);
        dx[10] = (L * (u[2] - u[0]) - (Ixx - Izz) * p * r) * (1.0 / Iyy);
        dx[11] = (kM * (u[0] - u[1] + u[2] - u[3]) - (Iyy - Ixx) * p * q) * (1.0 / Izz);
    }
};

// env.dynamics(x, u): clip(u) -> RK4(dt) -> clip(xn)       (cartpole.py:62-97)
// UCLIP = false: the caller has already clipped u to +-umax (the rollout's control law does, ilqr.py:97; clipping
// twice is the identity) - three instructions less in a chain of dependent ones.
template <int MODEL, class X, bool UCLIP = true>
__device__ __forceinline__ void env_dynamics(const trajopt_env_t& e, const X (&x)[Model<MODEL>::N],
                                             const X (&u)[Model<MODEL>::M], X (&xn)[Model<MODEL>::N]) {
    constexpr int N = Model<MODEL>::N, M = Model<MODEL>::M;
    X uc[M];
#pragma unroll
    for (int j = 0; j < M; ++j) uc[j] = UCLIP ? clip_g(u[j], e.umax[j]) : u[j];
    const double dt = ReverseTime<MODEL>::value ? -e.dt : e.dt;
    X k1[N], k2[N], k3[N], k4[N], y[N];
    Model<MODEL>::f(e, x, uc, k1);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] + (0.5 * dt) * k1[i];
    Model<MODEL>::f(e, y, uc, k2);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] + (0.5 * dt) * k2[i];
    Model<MODEL>::f(e, y, uc, k3);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] + dt * k3[i];
    Model<MODEL>::f(e, y, uc, k4);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const X s = x[i] + (dt / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        xn[i] = clip_g(s, e.xmax[i]);
    }
}

// The clip of a double as exactly two instructions.  fmin(fmax(x, -lim), lim) makes hipcc quiet its operands first (a
// v_max_f64 x, x per limit - and it prefers recomputing those in the time loop to holding them: ten instructions a
// cartpole step); the hardware instructions already implement minnum / maxnum on any input (IEEE mode quiets a
// signalling NaN by itself), so the results are the same bit for bit.  The lower limit is the upper one under the
// instruction's negate modifier: no register (or, as hipcc kept it, high word + a move per use) for -lim.
__device__ __forceinline__ double clip_hw(double x, double lim) {
    double r;
    asm("v_max_f64 %0, %1, -%2\n\tv_min_f64 %0, %0, %2" : "=&v"(r) : "v"(x), "v"(lim));
    return r;
}
__device__ __forceinline__ double max_hw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// env_dynamics<MODEL, double, false> with the state clip as clip_hw (the rollout's time loop).
template <int MODEL>
__device__ __forceinline__ void env_dynamics_hw(const trajopt_env_t& e, const double (&x)[Model<MODEL>::N],
                                                const double (&u)[Model<MODEL>::M], double (&xn)[Model<MODEL>::N]) {
    constexpr int N = Model<MODEL>::N;
    const double dt = ReverseTime<MODEL>::value ? -e.dt : e.dt;
    double k1[N], k2[N], k3[N], k4[N], y[N];
    Model<MODEL>::f(e, x, u, k1);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] + (0.5 * dt) * k1[i];
    Model<MODEL>::f(e, y, u, k2);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] + (0.5 * dt) * k2[i];
    Model<MODEL>::f(e, y, u, k3);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] + dt * k3[i];
    Model<MODEL>::f(e, y, u, k4);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = x[i] + (dt / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        xn[i] = clip_hw(s, e.xmax[i]);
    }
}

// LQR env: xn = A x + B clip(u) + c, no state clip (lqr_v0.py:50-66).  Run-time dims (<= MAX): loops
// run to the compile-time maximum with guards so the arrays keep static indices (stay in registers).
template <class X>
__device__ __forceinline__ void lqr_dynamics(const trajopt_env_t& e, const X (&x)[TRAJOPT_ENV_MAX_STATE],
                                             const X (&u)[TRAJOPT_ENV_MAX_ACT], X (&xn)[TRAJOPT_ENV_MAX_STATE]) {
    const int n = e.dm_state, m = e.dm_act;
    X uc[TRAJOPT_ENV_MAX_ACT];
#pragma unroll
    for (int j = 0; j < TRAJOPT_ENV_MAX_ACT; ++j) uc[j] = (j < m) ? clip_g(u[j], e.umax[j]) : Scalar<X>::c(0.0);
#pragma unroll
    for (int i = 0; i < TRAJOPT_ENV_MAX_STATE; ++i) {
        if (i < n) {
            X s = Scalar<X>::c(0.0), s2 = Scalar<X>::c(0.0);
#pragma unroll
            for (int j = 0; j < TRAJOPT_ENV_MAX_STATE; ++j)
                if (j < n) s = s + e.lin_A[i + n * j] * x[j];
#pragma unroll
            for (int j = 0; j < TRAJOPT_ENV_MAX_ACT; ++j)
                if (j < m) s2 = s2 + e.lin_B[i + n * j] * uc[j];
            xn[i] = s + s2 + e.lin_c[i];
        }
    }
}

// ((x + pi) % 2pi) - pi with Python's % (result in [-pi, pi)).  floor-based instead of fmod: the library fmod
// is a loop of frexp/ldexp steps; this form differs from it by ~|x| * 1e-16.
__device__ __forceinline__ double wrap_angle(double x) {
    const double two_pi = 6.28318530717958623200e+00;
    const double y = x + M_PI;
    const double k = floor(y * 1.59154943091895345554e-01);
    double r = fma(-k, 2.44929359829470641435e-16, fma(-k, two_pi, y));
    r = (r < 0.0) ? r + two_pi : r;
    r = (r >= two_pi) ? r - two_pi : r;
    return r - M_PI;
}

// env.cost(x, u, u, a) for the plain case - no slew-rate cost, no angle wrapping, no dt scaling (the three flags are
// wave-uniform, but testing them every step costs a lone wave ~20 instructions of a 470-instruction step).
template <int N, int M>
__device__ __forceinline__ double env_cost_plain(const trajopt_env_t& e, const double (&x)[N], const double (&u)[M], double a) {
    double c = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) c = fma(u[j] * e.uw[j], u[j], c);
    if (a != 0.0) {  // Python `if a:` drops the state cost for an exactly-zero weight
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double r = x[i] - e.goal[i];
            s = fma(r * e.gw[i], r, s);
        }
        c = fma(a, s, c);
    }
    return c;
}

// State cost (z - g)'G(z - g) with the features z = phi(y) of trajopt_env_t::features (cartpole.py:119-125,161-166)
// and, when DERIV, its frozen-Jacobian derivatives per state: gx[i] = sum_f gw_f (z_f - g_f) dz_f/dy_i and
// hxx[i] = sum_f gw_f (dz_f/dy_i)^2 (every feature depends on ONE state, so J'GJ is diagonal); the caller scales
// them by 2 a.  State arrays are only indexed statically: the feature list is scanned per state.
template <int N, bool DERIV>
__device__ __forceinline__ double feature_cost(const trajopt_env_t& e, const double (&x)[N], int n, double (&gx)[N],
                                               double (&hxx)[N]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        gx[i] = 0.0;
        hxx[i] = 0.0;
        if (i < n) {
            const double y = ((e.periodic >> i) & 1) ? wrap_angle(x[i]) : x[i];
            double sn, cs;
            sincos_g(y, sn, cs);
#pragma unroll
            for (int f = 0; f < TRAJOPT_ENV_MAX_STATE; ++f) {
                const int code = e.features[f];
                if (f < e.nb_features && (code & 0xff) == i) {
                    const int kind = code >> 8;
                    const double z = kind == TRAJOPT_FEAT_ID ? y : (kind == TRAJOPT_FEAT_COS ? cs : sn);
                    const double dz = kind == TRAJOPT_FEAT_ID ? 1.0 : (kind == TRAJOPT_FEAT_COS ? -sn : cs);
                    const double r = z - e.goal[f];
                    s = fma(r * e.gw[f], r, s);
                    if (DERIV) {
                        gx[i] = fma(e.gw[f] * r, dz, gx[i]);
                        hxx[i] = fma(e.gw[f] * dz, dz, hxx[i]);
                    }
                }
            }
        }
    }
    return s;
}

// env.cost(x, u, u_last, a) (cartpole.py:112-127; LQR: lqr_v0.py:89-94).
// N, M are array capacities; n, m the live dimensions.
template <int N, int M>
__device__ __forceinline__ double env_cost(const trajopt_env_t& e, const double (&x)[N], const double (&u)[M],
                                           const double (&u_last)[M], double a, int n, int m) {
    double c = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j)
        if (j < m) {
            const double d = e.slew_rate ? (u[j] - u_last[j]) : u[j];
            c = fma(d * e.uw[j], d, c);
        }
    if (a != 0.0) {  // Python `if a:` drops the state cost for an exactly-zero weight
        double s = 0.0;
        if (e.nb_features == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i < n) {
                    const double y = ((e.periodic >> i) & 1) ? wrap_angle(x[i]) : x[i];
                    const double r = y - e.goal[i];
                    s = fma(r * e.gw[i], r, s);
                }
        } else {
            double cx_unused[N], cxx_unused[N];
            s = feature_cost<N, false>(e, x, n, cx_unused, cxx_unused);
        }
        c = fma(a, s, c);
    }
    return e.cost_times_dt ? e.dt * c : c;
}

}  // namespace trajopt
