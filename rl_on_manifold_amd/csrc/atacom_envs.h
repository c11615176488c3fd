// Constraint callables (value, Jacobian, bias) of the three environments, one environment per lane.
//
// Replaces, per environment,
//   circle : /root/reference/atacom/environments/circular_motion/circle_atacom.py:47-70
//   planar : /root/reference/atacom/environments/planar_air_hockey/atacom_air_hockey.py:78-107
//            (Pinocchio framesForwardKinematics / computeFrameJacobian / getFrameClassicalAcceleration
//             on MushroomRL's 3R planar arm -- parameters: DESIGN.md "Planar robot")
//   iiwa   : /root/reference/atacom/environments/iiwa_air_hockey/iiwa_hit_atacom.py:70-139 over the
//            chain of urdf/iiwa_1.urdf:69-301 with the tip frame of env_base.py:147-151.
// The reference re-runs Pinocchio forward kinematics ~10 times per physics sub-step on the same
// (q, dq); here one pass produces every frame, Jacobian row and bias term in registers.
#pragma once
#include "atacom_linalg.h"

namespace atacom {

// MODE: 0 = ATACOM wrapper (atacom/atacom.py); 1 = ErrorCorrection baseline "E" (atacom/error_correction_wrapper.py:
// the action is the joint acceleration itself, only -Jc^+ Kc c is added, q/dq refreshed every sub-step);
// 2 = Terminated baseline "T" (circle_terminated.py: unconstrained, episode ends with reward -100 when c > tol).
struct Circle {
    static constexpr int ID = 0, NQ = 2, NF = 1, NG = 1, NC = 2, NN = 3, NK = 1, OBS = 4, MODE = 0;
    static constexpr bool PUCK = false;
    // jac_zero(r, i): entry (r, i) of the constraint Jacobian is STRUCTURALLY zero (constraint_terms writes a literal 0
    // there), so the assembly of K J and J dq skips it at compile time -- the optimiser may not fold fma(K, 0, 0)
    static constexpr bool jac_zero(int, int) { return false; }
};
struct CircleEC : Circle {      // CircleEnvErrorCorrection, circle_error_correction.py:7-21
    static constexpr int NK = 2, MODE = 1;
};
struct CircleT : Circle {       // CircleEnvTerminated, circle_terminated.py:8-29
    static constexpr int NK = 2, MODE = 2;
};
struct Planar {
    static constexpr int ID = 1, NQ = 3, NF = 0, NG = 6, NC = 6, NN = 9, NK = 3, OBS = 12, MODE = 0;
    static constexpr bool PUCK = true;
    static constexpr bool jac_zero(int r, int i) { return r >= 3 && i != r - 3; }     // joint-limit rows: diagonal
};
struct Iiwa {
    static constexpr int ID = 2, NQ = 6, NF = 1, NG = 11, NC = 12, NN = 17, NK = 5, OBS = 18, MODE = 0;
    static constexpr bool PUCK = true;
    // joint-limit rows 6..11: diagonal; row 4 (height of link_4): joints 3..6 do not move the frame
    static constexpr bool jac_zero(int r, int i) { return (r >= 6 && i != r - 6) || (r == 4 && i >= 2); }
};

// Everything a kernel needs besides per-env state; passed by value (kernarg segment -> SGPRs).
template <typename T>
struct Params {
    int batch, substeps, horizon, hold_q, bias_mode, auto_reset, random_init, dynamics_mode;
    int task;                  // planar: 0 = hit, 1 = defend (atacom_air_hockey.py:22-27)
    unsigned int seed;
    T dt, dt_base, rref_tol, action_penalty, alpha_max;      // dt_base: the base env's integrator step (circle quirk Q4)
    T K[12], Kc[12], vel_max[6], acc_max[6], Kq[6], pos_limit[6];
    T base_x, base_y;
    T link[3];
    T table_bx, table_by;      // table half extents minus mallet radius (0.93, 0.46)
    T table_hx, table_hy;      // table half extents (0.98, 0.51)
    T goal_x, goal_y, goal_w;  // goal position / half width
    T ee_height;               // universal_height 0.1505
    T z4_min, z7_min;          // 0.36, 0.25  (iiwa_hit_atacom.py:104-105)
    T puck_r, mallet_r;        // 0.03165, 0.05 (env_base.py:157-158)
    T e_mallet, e_rim;         // restitution of the contact model of this build
    T term_tol;                // MODE 2: termination tolerance (circle_terminated.py:13)
    // domain randomisation of the air-hockey base envs (constructor kwargs of iiwa_hit_atacom.py:11-13 /
    // atacom_air_hockey.py:12-14; all off by default): launch-uniform bits, see NOISE_* below
    int noise;
    T obs_noise_std;           // 0.001 (env_single.py:105-107)
    T env_noise_dv;            // 0.0005 N * dt / puck mass: velocity kick per unit normal draw and sub-step (env_base.py:176-180)
};
// Params::noise bits
constexpr int NOISE_OBS = 1;     // obs_noise: puck pose (x, y, yaw) of every observation += N(0, 0.001^2)   env_single.py:105-107
constexpr int NOISE_DELAY = 2;   // obs_delay: puck / joint velocities of every observation low-passed, alpha = 0.5   :114-117
constexpr int NOISE_ENV = 4;     // env_noise: random planar force on the puck in every physics sub-step   env_base.py:176-180

// ---------------------------------------------------------------------------------------- circle
template <typename T>
__device__ __forceinline__ void constraint_terms(Circle, const Params<T>&, const T (&q)[2], const T (&dq)[2],
                                                 T (&fun)[2], T (&J)[2][2], T (&bst)[2]) {
    fun[0] = num<T>::fma(q[0], q[0], q[1] * q[1]) - T(1);      // circle_atacom.py:47-49
    fun[1] = -q[1] - T(0.5);                                   // :59-61
    J[0][0] = T(2) * q[0]; J[0][1] = T(2) * q[1];              // :51-53
    J[1][0] = T(0);        J[1][1] = T(-1);                    // :63-65
    bst[0] = T(2) * dq[0] * dq[0] + T(2) * dq[1] * dq[1];      // :55-57
    bst[1] = T(0);                                             // :67-69
}
template <typename T>
__device__ __forceinline__ void constraint_fun(Circle, const Params<T>&, const T (&q)[2], T (&fun)[2], T (&mxy)[2]) {
    fun[0] = num<T>::fma(q[0], q[0], q[1] * q[1]) - T(1);
    fun[1] = -q[1] - T(0.5);
    mxy[0] = q[0]; mxy[1] = q[1];
}

// ---------------------------------------------------------------------------------------- planar
template <typename T>
__device__ __forceinline__ void planar_fk(const Params<T>& P, const T (&q)[3], T (&cx)[3], T (&sy)[3]) {
    T th = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        th += q[i];
        T s, c;
        num<T>::sincos(th, &s, &c);
        cx[i] = P.link[i] * c;
        sy[i] = P.link[i] * s;
    }
}
template <typename T>
__device__ __forceinline__ void constraint_terms(Planar, const Params<T>& P, const T (&q)[3], const T (&dq)[3],
                                                 T (&fun)[6], T (&J)[6][3], T (&bst)[6]) {
    T cx[3], sy[3];
    planar_fk(P, q, cx, sy);
    const T xw = (cx[0] + cx[1] + cx[2]) + P.base_x;
    const T yw = (sy[0] + sy[1] + sy[2]) + P.base_y;
    fun[0] = -xw - P.table_bx;                                 // atacom_air_hockey.py:78-85
    fun[1] = -yw - P.table_by;
    fun[2] = yw - P.table_by;
    T Jx[3], Jy[3];
    Jx[2] = -sy[2];               Jy[2] = cx[2];
    Jx[1] = -(sy[1] + sy[2]);     Jy[1] = cx[1] + cx[2];
    Jx[0] = -(sy[0] + sy[1] + sy[2]); Jy[0] = cx[0] + cx[1] + cx[2];
    T vx = T(0), vy = T(0), w = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        J[0][i] = -Jx[i]; J[1][i] = -Jy[i]; J[2][i] = Jy[i];   // :87-91
        vx = num<T>::fma(Jx[i], dq[i], vx);
        vy = num<T>::fma(Jy[i], dq[i], vy);
        w += dq[i];
    }
    T ax, ay;
    if (P.bias_mode == 0) {           // reference: (0,0,w) x (vx,vy,0)   (quirk Q2)
        ax = -w * vy; ay = w * vx;
    } else {                          // exact dJ/dt dq
        T thd = T(0);
        ax = T(0); ay = T(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            thd += dq[i];
            ax = num<T>::fma(-cx[i], thd * thd, ax);
            ay = num<T>::fma(-sy[i], thd * thd, ay);
        }
    }
    bst[0] = -ax; bst[1] = -ay; bst[2] = ay;                   // :93-98
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fun[3 + i] = num<T>::fma(q[i], q[i], -P.pos_limit[i] * P.pos_limit[i]);   // :100-101
#pragma unroll
        for (int j = 0; j < 3; ++j) J[3 + i][j] = (i == j) ? T(2) * q[i] : T(0);  // :103-104
        bst[3 + i] = T(2) * dq[i] * dq[i];                                         // :106-107
    }
}
template <typename T>
__device__ __forceinline__ void constraint_fun(Planar, const Params<T>& P, const T (&q)[3], T (&fun)[6], T (&mxy)[2]) {
    T cx[3], sy[3];
    planar_fk(P, q, cx, sy);
    const T xw = (cx[0] + cx[1] + cx[2]) + P.base_x;
    const T yw = (sy[0] + sy[1] + sy[2]) + P.base_y;
    fun[0] = -xw - P.table_bx;
    fun[1] = -yw - P.table_by;
    fun[2] = yw - P.table_by;
#pragma unroll
    for (int i = 0; i < 3; ++i) fun[3 + i] = num<T>::fma(q[i], q[i], -P.pos_limit[i] * P.pos_limit[i]);
    mxy[0] = xw; mxy[1] = yw;
}

// ---------------------------------------------------------------------------------------- iiwa
// Joint origins (iiwa_1.urdf:72,110,147,184,221,258,295): translation along ONE axis of the parent
// frame, then a fixed signed-permutation rotation, then Rz(q_i):
//   kind A: rpy (pi/2, 0, pi) or (-pi/2, pi, 0) -> columns (-x, z, y)   (joints 2, 3, 5, 7)
//   kind B: rpy (pi/2, 0, 0)                    -> columns ( x, z,-y)   (joints 4, 6)
// Offsets: z .1575 | z .2025 | y .2045 | z .2155 | y .1845 | z .2155 | y .081 ; tip +z .585 of link_7.
// Joint 7 is held at 0 (iiwa_hit_atacom.py:65-68) and its axis passes through the tip, so the six
// controlled joints define everything: link_7 origin = o6 + y6 * .081, tip = o6 + y6 * .666.
template <typename T>
struct IiwaKin {
    T z[6][3];     // joint axes (world = robot base axes)
    T o[6][3];     // joint origins
    T p4[3], p7[3], pe[3];
};

// the chain from the joints' sines / cosines (split from the trigonometry so that a lane group can share ONE sincos per
// lane instead of six per lane: atacom_kernels.h, group_sincos6)
template <typename T>
__device__ __forceinline__ void iiwa_chain(const T (&sn)[6], const T (&cs)[6], IiwaKin<T>& k) {
    // frame columns X, Y, Z of the current link
    T X[3] = {T(1), T(0), T(0)}, Y[3] = {T(0), T(1), T(0)}, Z[3] = {T(0), T(0), T(1)};
    T o[3] = {T(0), T(0), T(0)};
    constexpr T off[6] = {T(0.1575), T(0.2025), T(0.2045), T(0.2155), T(0.1845), T(0.2155)};
    constexpr int off_axis[6] = {2, 2, 1, 2, 1, 2};           // 2 = along Z, 1 = along Y of the parent
    constexpr int kind[6] = {0, 1, 1, 2, 1, 2};               // 0 identity, 1 kind A, 2 kind B
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a] = num<T>::fma(off_axis[i] == 2 ? Z[a] : Y[a], off[i], o[a]);
        T nx[3], ny[3], nz[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (kind[i] == 0) { nx[a] = X[a]; ny[a] = Y[a]; nz[a] = Z[a]; }
            else if (kind[i] == 1) { nx[a] = -X[a]; ny[a] = Z[a]; nz[a] = Y[a]; }
            else { nx[a] = X[a]; ny[a] = Z[a]; nz[a] = -Y[a]; }
        }
        const T s = sn[i], c = cs[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            X[a] = num<T>::fma(c, nx[a], s * ny[a]);
            Y[a] = num<T>::fma(c, ny[a], -(s * nx[a]));
            Z[a] = nz[a];
            k.z[i][a] = Z[a];
            k.o[i][a] = o[a];
        }
        if (i == 3) {
#pragma unroll
            for (int a = 0; a < 3; ++a) k.p4[a] = o[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        k.p7[a] = num<T>::fma(Y[a], T(0.081), o[a]);
        k.pe[a] = num<T>::fma(Y[a], T(0.081) + T(0.585), o[a]);
    }
}
template <typename T>
__device__ __forceinline__ void iiwa_fk(const T (&q)[6], IiwaKin<T>& k) {
    T sn[6], cs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) num<T>::sincos(q[i], &sn[i], &cs[i]);
    iiwa_chain(sn, cs, k);
}

// linear Jacobian column i of point p:  z_i x (p - o_i)
template <typename T>
__device__ __forceinline__ void jac_col(const IiwaKin<T>& k, int i, const T (&p)[3], T (&col)[3]) {
    const T rx = p[0] - k.o[i][0], ry = p[1] - k.o[i][1], rz = p[2] - k.o[i][2];
    col[0] = num<T>::fma(k.z[i][1], rz, -(k.z[i][2] * ry));
    col[1] = num<T>::fma(k.z[i][2], rx, -(k.z[i][0] * rz));
    col[2] = num<T>::fma(k.z[i][0], ry, -(k.z[i][1] * rx));
}

// frame bias for zero joint acceleration.  mode 0 = reference (w x v, quirk Q2), 1 = exact dJ/dt dq.
template <typename T, int NJ>
__device__ __forceinline__ void frame_bias(const IiwaKin<T>& k, const T (&p)[3], const T (&Jl)[3][6],
                                           const T (&dq)[6], int mode, T (&acc)[3]) {
    T w[3] = {T(0), T(0), T(0)}, v[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int i = 0; i < NJ; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            w[a] = num<T>::fma(k.z[i][a], dq[i], w[a]);
            v[a] = num<T>::fma(Jl[a][i], dq[i], v[a]);
        }
    if (mode == 0) {
        acc[0] = num<T>::fma(w[1], v[2], -(w[2] * v[1]));
        acc[1] = num<T>::fma(w[2], v[0], -(w[0] * v[2]));
        acc[2] = num<T>::fma(w[0], v[1], -(w[1] * v[0]));
        return;
    }
    // exact: sum_i [ (w_i x z_i) x (p - o_i) + z_i x (v_p - v_{o_i}) ] dq_i
    acc[0] = acc[1] = acc[2] = T(0);
    T wi[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        T vo[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int j = 0; j < i; ++j) {
            T col[3];
            const T oi[3] = {k.o[i][0], k.o[i][1], k.o[i][2]};
            jac_col(k, j, oi, col);
#pragma unroll
            for (int a = 0; a < 3; ++a) vo[a] = num<T>::fma(col[a], dq[j], vo[a]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) wi[a] = num<T>::fma(k.z[i][a], dq[i], wi[a]);
        const T zd[3] = {wi[1] * k.z[i][2] - wi[2] * k.z[i][1], wi[2] * k.z[i][0] - wi[0] * k.z[i][2],
                         wi[0] * k.z[i][1] - wi[1] * k.z[i][0]};
        const T r[3] = {p[0] - k.o[i][0], p[1] - k.o[i][1], p[2] - k.o[i][2]};
        const T dv[3] = {v[0] - vo[0], v[1] - vo[1], v[2] - vo[2]};
        const T t0 = (zd[1] * r[2] - zd[2] * r[1]) + (k.z[i][1] * dv[2] - k.z[i][2] * dv[1]);
        const T t1 = (zd[2] * r[0] - zd[0] * r[2]) + (k.z[i][2] * dv[0] - k.z[i][0] * dv[2]);
        const T t2 = (zd[0] * r[1] - zd[1] * r[0]) + (k.z[i][0] * dv[1] - k.z[i][1] * dv[0]);
        acc[0] = num<T>::fma(t0, dq[i], acc[0]);
        acc[1] = num<T>::fma(t1, dq[i], acc[1]);
        acc[2] = num<T>::fma(t2, dq[i], acc[2]);
    }
}

template <typename T>
__device__ __forceinline__ void constraint_terms(Iiwa, const Params<T>& P, const T (&q)[6], const T (&dq)[6],
                                                 T (&fun)[12], T (&J)[12][6], T (&bst)[12]) {
    IiwaKin<T> k;
    iiwa_fk(q, k);
    ATACOM_MARK("PRE_jac");
    T Je[3][6], J4[3][6], J7[3][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        T col[3];
        jac_col(k, i, k.pe, col);
        Je[0][i] = col[0]; Je[1][i] = col[1]; Je[2][i] = col[2];
        jac_col(k, i, k.p7, col);
        J7[0][i] = col[0]; J7[1][i] = col[1]; J7[2][i] = col[2];
        if (i < 2) {
            jac_col(k, i, k.p4, col);
            J4[0][i] = col[0]; J4[1][i] = col[1]; J4[2][i] = col[2];
        } else {
            // link_4's origin lies ON the axes of joints 3 and 4 (offset 0.2155 along z of link_3,
            // iiwa_1.urdf:184): those columns are structurally zero; joints 5, 6 are past the frame
            J4[0][i] = J4[1][i] = J4[2][i] = T(0);
        }
    }
    ATACOM_MARK("PRE_bias");
    T ae[3], a4[3], a7[3];
    frame_bias<T, 6>(k, k.pe, Je, dq, P.bias_mode, ae);
    frame_bias<T, 4>(k, k.p4, J4, dq, P.bias_mode, a4);
    frame_bias<T, 6>(k, k.p7, J7, dq, P.bias_mode, a7);
    const T xw = k.pe[0] + P.base_x, yw = k.pe[1] + P.base_y;
    fun[0] = k.pe[2] - P.ee_height;                            // iiwa_hit_atacom.py:70-74
    fun[1] = -xw - P.table_bx;                                 // :93-100
    fun[2] = -yw - P.table_by;
    fun[3] = yw - P.table_by;
    fun[4] = -k.p4[2] + P.z4_min;                              // :102-106
    fun[5] = -k.p7[2] + P.z7_min;
    bst[0] = ae[2];                                            // :84-91
    bst[1] = -ae[0]; bst[2] = -ae[1]; bst[3] = ae[1]; bst[4] = -a4[2]; bst[5] = -a7[2];   // :119-130
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        J[0][i] = Je[2][i];                                    // :76-82
        J[1][i] = -Je[0][i]; J[2][i] = -Je[1][i]; J[3][i] = Je[1][i];          // :108-117
        J[4][i] = -J4[2][i]; J[5][i] = -J7[2][i];
        fun[6 + i] = num<T>::fma(q[i], q[i], -P.pos_limit[i] * P.pos_limit[i]);   // :132-133
#pragma unroll
        for (int j = 0; j < 6; ++j) J[6 + i][j] = (i == j) ? T(2) * q[i] : T(0);  // :135-136
        bst[6 + i] = T(2) * dq[i] * dq[i];                                         // :138-139
    }
}
template <typename T>
__device__ __forceinline__ void iiwa_fun_from_kin(const Params<T>& P, const IiwaKin<T>& k, const T (&q)[6], T (&fun)[12],
                                                  T (&mxy)[2]) {
    const T xw = k.pe[0] + P.base_x, yw = k.pe[1] + P.base_y;
    fun[0] = k.pe[2] - P.ee_height;
    fun[1] = -xw - P.table_bx;
    fun[2] = -yw - P.table_by;
    fun[3] = yw - P.table_by;
    fun[4] = -k.p4[2] + P.z4_min;
    fun[5] = -k.p7[2] + P.z7_min;
#pragma unroll
    for (int i = 0; i < 6; ++i) fun[6 + i] = num<T>::fma(q[i], q[i], -P.pos_limit[i] * P.pos_limit[i]);
    mxy[0] = xw; mxy[1] = yw;
}
template <typename T>
__device__ __forceinline__ void constraint_fun(Iiwa, const Params<T>& P, const T (&q)[6], T (&fun)[12], T (&mxy)[2]) {
    IiwaKin<T> k;
    iiwa_fk(q, k);
    iiwa_fun_from_kin(P, k, q, fun, mxy);
}

}  // namespace atacom
