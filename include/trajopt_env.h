/*
 * trajopt_env.h - plain-old-data description of an env model (dynamics + cost).
 *
 * The reference's env layer is Python (the `trajopt/envs/` package): each env object carries
 * `dynamics(x, u)`, `cost(x, u, u_last, a)` and a handful of constants
 * (`dt`, `umax`, `xmax`, `g`, `gw`, `uw`, `slew_rate`, `periodic`), e.g.
 * /root/reference/trajopt/envs/cartpole/cartpole.py:17-52.  On the device those
 * become a model id plus this parameter struct, passed by value to the kernels.
 *
 * Shared by the HIP library (trajopt_amd/csrc) and by the CPU oracle (oracle/),
 * which is why it has no HIP or C++ types in it.
 */
#ifndef TRAJOPT_ENV_H
#define TRAJOPT_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRAJOPT_ENV_MAX_STATE 12
#define TRAJOPT_ENV_MAX_ACT 4

#define TRAJOPT_FEAT_ID 0
#define TRAJOPT_FEAT_COS 1
#define TRAJOPT_FEAT_SIN 2
#define TRAJOPT_FEATURE(kind, state) (((kind) << 8) | (state))

/* model ids */
enum {
    TRAJOPT_ENV_PENDULUM = 1,        /* envs/pendulum/pendulum.py:15-149           dx=2 du=1 */
    TRAJOPT_ENV_CARTPOLE = 2,        /* envs/cartpole/cartpole.py:15-147           dx=4 du=1 */
    TRAJOPT_ENV_DOUBLE_CARTPOLE = 3, /* envs/double_cartpole/double_cartpole.py    dx=6 du=1 */
    TRAJOPT_ENV_LQR = 4,             /* envs/lqr/lqr_v0.py:8-114 (affine LTI)      dx<=12 du<=4 */
    TRAJOPT_ENV_QUADROTOR = 5,       /* build-defined (no counterpart in the reference) dx=12 du=4 */
    TRAJOPT_ENV_DOUBLE_PENDULUM = 6, /* envs/double_pendulum/double_pendulum.py:15-172    dx=4 du=2 */
    TRAJOPT_ENV_QUAD_PENDULUM = 8,   /* envs/quad_pendulum/quad_pendulum.py:14-303 (four links)        dx=8 du=4 */
    TRAJOPT_ENV_ROBOT = 7            /* envs/robot/robot.py:8-102 (car-like robot; its RK4 steps by -dt, :62-67) dx=4 du=2 */
};

typedef struct trajopt_env {
    int32_t model;        /* one of TRAJOPT_ENV_*                                            */
    int32_t dm_state;     /* n                                                               */
    int32_t dm_act;       /* m                                                               */
    int32_t slew_rate;    /* cost on (u - u_last) instead of u   (cartpole.py:115-118)       */
    int32_t periodic;     /* wrap angle states before the cost   (cartpole.py:121-122)       */
    int32_t cost_times_dt;/* LQR envs scale the whole cost by dt (lqr_v0.py:94)              */
    /* State-cost features z = phi(y) of the "...WithCartesianCost" envs (e.g. cartpole.py:150-166): 0 = identity
     * (z = y, goal / gw hold dm_state entries); else goal / gw hold nb_features entries and feature f is
     * TRAJOPT_FEAT_{ID,COS,SIN} of state (features[f] & 0xff), kind in bits 8-9.  The Jacobian of phi is frozen in
     * the derivatives (getval, cartpole.py:122-124): cx = 2a J'G(z - g), Cxx = 2a J'GJ (Gauss-Newton).              */
    int32_t nb_features;
    int32_t features[TRAJOPT_ENV_MAX_STATE];
    int32_t reserved0;    /* keeps the doubles 8-byte aligned                                */
    double dt;            /* RK4 step                                                        */
    double umax[TRAJOPT_ENV_MAX_ACT];    /* action clip  (+inf allowed)                      */
    double xmax[TRAJOPT_ENV_MAX_STATE];  /* state clip   (+inf allowed)                      */
    double goal[TRAJOPT_ENV_MAX_STATE];  /* g                                                */
    double gw[TRAJOPT_ENV_MAX_STATE];    /* diagonal state-cost weights                      */
    double uw[TRAJOPT_ENV_MAX_ACT];      /* diagonal action-cost weights                     */
    /* TRAJOPT_ENV_LQR only: xn = A x + B clip(u) + c, column-major A(n,n), B(n,m), c(n)     */
    double lin_A[TRAJOPT_ENV_MAX_STATE * TRAJOPT_ENV_MAX_STATE];
    double lin_B[TRAJOPT_ENV_MAX_STATE * TRAJOPT_ENV_MAX_ACT];
    double lin_c[TRAJOPT_ENV_MAX_STATE];
    /* physical constants; meaning depends on the model (see trajopt_amd/csrc/env_models.h)  */
    double phys[16];
} trajopt_env_t;

#ifdef __cplusplus
}
#endif
#endif /* TRAJOPT_ENV_H */
