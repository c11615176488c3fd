// Row N4, round 5: the rigid-body recursions of the iiwa + striker chain IN LINK COORDINATES ("the
// candidate" of round 4's list).
//
// atacom_dynamics.h evaluates recursive Newton-Euler and the composite-rigid-body mass matrix in WORLD coordinates: it first
// builds, per physics sub-step, the world frame of all nine bodies -- joint axes, origins, centres of mass and R I R^T of
// every inertia tensor: 135 live values, 9 x 45 operations for the tensors alone -- and the kernels that call it run at the
// edge of the register file (LDS parking of the solver state, scratch in the lane mapping).  Here every quantity of body i
// lives in body i's own joint frame:
//   * inertia tensors, centres of mass and the bodies' own composite inertias are INSTRUCTION LITERALS (atacom_iiwa_inertia.h);
//   * a joint is "translate along one axis of the parent, a signed permutation of the axes, a plane rotation by q_i"
//     (urdf/iiwa_1.urdf:72,110,147,184,221,258,295; the striker's universal joint :380-399): moving a vector across a joint
//     costs four multiply-adds, a symmetric tensor twelve; cross products with the joint offset touch two components;
//   * nothing of the chain is kept but the nine sines / cosines.
// Same equations as oracle/dynamics.py (rnea, mass_matrix), which golden set G11 pins to the reference's URDF: the tests
// compare the float64 build at 1e-10 (tests/test_gpu_dynamics.py).
// What the reference does with them: PyBullet calculateInverseDynamics (iiwa_hit_atacom.py:58-63) and stepSimulation.
#pragma once
#include "atacom_linalg.h"
#include "atacom_iiwa_inertia.h"

namespace atacom {
namespace lk {

// -DATACOM_LK_FENCE=1: a scheduling barrier after every body of the recursions (tuning: keeps the scheduler from interleaving
// bodies, i.e. from stretching live ranges, in kernels at the edge of the register file)
#ifndef ATACOM_LK_FENCE
#define ATACOM_LK_FENCE 0
#endif
#if ATACOM_LK_FENCE
#define ATACOM_LK_BODY() __builtin_amdgcn_sched_barrier(0)
#else
#define ATACOM_LK_BODY() ((void)0)
#endif

// ---- joint descriptors.  Joint i maps parent coordinates v to child coordinates v' by
//   u_k = SG[k] v[PM[k]]   (signed permutation),   v'_A = c u_A + s u_B,   v'_B = c u_B - s u_A,   v'_C = u_C
// with (c, s) = cos / sin q_i; C is the joint axis in the child's (and, through the permutation, the parent's) frame.
// Arm joints 0..6: kinds of atacom_envs.h (0: identity, 1: u = (-x, z, y), 2: u = (x, z, -y)), rotation about the local z.
// Joint 7 (striker_joint_1): about the local y; joint 8 (striker_joint_2): about the local x (atacom_dynamics.h).
constexpr int PM[9][3] = {{0, 1, 2}, {0, 2, 1}, {0, 2, 1}, {0, 2, 1}, {0, 2, 1}, {0, 2, 1}, {0, 2, 1}, {0, 1, 2}, {0, 1, 2}};
constexpr int SG[9][3] = {{1, 1, 1}, {-1, 1, 1}, {-1, 1, 1}, {1, 1, -1}, {-1, 1, 1}, {1, 1, -1}, {-1, 1, 1}, {1, 1, 1}, {1, 1, 1}};
constexpr int PA[9] = {0, 0, 0, 0, 0, 0, 0, 2, 1};
constexpr int PB[9] = {1, 1, 1, 1, 1, 1, 1, 0, 2};
constexpr int PC[9] = {2, 2, 2, 2, 2, 2, 2, 1, 0};
// origin of joint i in the parent's frame: TOFF[i] along axis TAX[i] (-1: none)
constexpr int TAX[9] = {2, 2, 1, 2, 1, 2, 1, 2, -1};
constexpr double TOFF[9] = {0.1575, 0.2025, 0.2045, 0.2155, 0.1845, 0.2155, 0.081, iiwa_body::STRIKER_OFFSET_Z, 0.0};

constexpr int sidx(int r, int c) {           // symmetric 3 x 3 stored as xx, xy, xz, yy, yz, zz
    return (r == 0 || c == 0) ? r + c : ((r == 1 || c == 1) ? r + c + 1 : 5);
}
// a body's own composite about its joint origin: first moment m c, inertia I_com + m (|c|^2 1 - c c^T)
constexpr double own_h(int i, int a) { return iiwa_body::MASS[i] * iiwa_body::COM[i][a]; }
constexpr double own_I(int i, int r, int c) {
    const double m = iiwa_body::MASS[i], x = iiwa_body::COM[i][0], y = iiwa_body::COM[i][1], z = iiwa_body::COM[i][2];
    const double cc = x * x + y * y + z * z;
    const double ci = iiwa_body::COM[i][r], cj = iiwa_body::COM[i][c];
    return iiwa_body::INERTIA[i][sidx(r, c)] + m * ((r == c ? cc : 0.0) - ci * cj);
}

template <typename T>
struct Trig9 { T s[9], c[9]; };

template <int I, typename T>
__device__ __forceinline__ void to_child(const Trig9<T>& g, const T (&v)[3], T (&o)[3]) {
    T u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] = (SG[I][k] > 0) ? v[PM[I][k]] : -v[PM[I][k]];
    constexpr int A = PA[I], B = PB[I], C = PC[I];
    o[A] = num<T>::fma(g.c[I], u[A], g.s[I] * u[B]);
    o[B] = num<T>::fma(g.c[I], u[B], -(g.s[I] * u[A]));
    o[C] = u[C];
}
template <int I, typename T>
__device__ __forceinline__ void to_parent(const Trig9<T>& g, const T (&v)[3], T (&o)[3]) {
    constexpr int A = PA[I], B = PB[I], C = PC[I];
    T u[3];
    u[A] = num<T>::fma(g.c[I], v[A], -(g.s[I] * v[B]));
    u[B] = num<T>::fma(g.s[I], v[A], g.c[I] * v[B]);
    u[C] = v[C];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[PM[I][k]] = (SG[I][k] > 0) ? u[k] : -u[k];
}
// symmetric tensor, child frame -> parent frame (same point)
template <int I, typename T>
__device__ __forceinline__ void sym_to_parent(const Trig9<T>& g, const T (&S)[6], T (&o)[6]) {
    constexpr int A = PA[I], B = PB[I], C = PC[I];
    const T c = g.c[I], s = g.s[I];
    const T c2 = num<T>::fma(c, c, -(s * s)), s2 = T(2) * c * s;             // cos / sin of 2 q
    const T Saa = S[sidx(A, A)], Sbb = S[sidx(B, B)], Sab = S[sidx(A, B)], Sac = S[sidx(A, C)], Sbc = S[sidx(B, C)];
    const T avg = T(0.5) * (Saa + Sbb), dif = T(0.5) * (Saa - Sbb);
    T U[6];                                                                   // in the u frame: u_A = c v'_A - s v'_B, u_B = s v'_A + c v'_B
    const T rot = num<T>::fma(dif, c2, -(Sab * s2));
    U[sidx(A, A)] = avg + rot;
    U[sidx(B, B)] = avg - rot;
    U[sidx(A, B)] = num<T>::fma(dif, s2, Sab * c2);
    U[sidx(A, C)] = num<T>::fma(c, Sac, -(s * Sbc));
    U[sidx(B, C)] = num<T>::fma(s, Sac, c * Sbc);
    U[sidx(C, C)] = S[sidx(C, C)];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = k; l < 3; ++l) {
            const T val = U[sidx(k, l)];
            o[sidx(PM[I][k], PM[I][l])] = (SG[I][k] * SG[I][l] > 0) ? val : -val;
        }
}
// (d e_AX) x v
template <int AX, typename T>
__device__ __forceinline__ void axis_cross(T d, const T (&v)[3], T (&o)[3]) {
    if constexpr (AX == 2) { o[0] = -(d * v[1]); o[1] = d * v[0]; o[2] = T(0); }
    else if constexpr (AX == 1) { o[0] = d * v[2]; o[1] = T(0); o[2] = -(d * v[0]); }
    else { o[0] = T(0); o[1] = -(d * v[2]); o[2] = d * v[1]; }
}
template <typename T>
__device__ __forceinline__ void crossv(const T (&u)[3], const T (&v)[3], T (&w)[3]) {
    w[0] = num<T>::fma(u[1], v[2], -(u[2] * v[1]));
    w[1] = num<T>::fma(u[2], v[0], -(u[0] * v[2]));
    w[2] = num<T>::fma(u[0], v[1], -(u[1] * v[0]));
}
template <typename T>
__device__ __forceinline__ void symv(const T (&S)[6], const T (&v)[3], T (&w)[3]) {
    w[0] = num<T>::fma(S[0], v[0], num<T>::fma(S[1], v[1], S[2] * v[2]));
    w[1] = num<T>::fma(S[1], v[0], num<T>::fma(S[3], v[1], S[4] * v[2]));
    w[2] = num<T>::fma(S[2], v[0], num<T>::fma(S[4], v[1], S[5] * v[2]));
}

template <typename T>
__device__ __forceinline__ void trig9(const T (&q)[9], Trig9<T>& g) {
#pragma unroll
    for (int i = 0; i < 9; ++i) num<T>::sincos(q[i], &g.s[i], &g.c[i]);
}

// tau[9] = M(q) ddq + C(q, dq) dq + g(q), gravity (0, 0, -9.81) as an upward acceleration of the base.
// ZERO_ACC: all joint accelerations zero (ddq not read): the bias h(q, dq).
template <typename T, bool ZERO_ACC = false>
__device__ __forceinline__ void rnea9(const Trig9<T>& g, const T (&dq)[9], const T (&ddq)[9], T (&tau)[9]) {
    T w[3] = {T(0), T(0), T(0)}, al[3] = {T(0), T(0), T(0)}, a[3] = {T(0), T(0), T(9.81)};
    T F[9][3], Nm[9][3];
    static_for<0, 9>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int C = PC[i];
        // acceleration of the joint's origin, in the parent's frame: a + al x r + w x (w x r), r = d e_AX
        T ao[3] = {a[0], a[1], a[2]};
        if constexpr (TAX[i] >= 0) {
            constexpr int AX = TAX[i];
            const T d = (T)TOFF[i];
            T t[3];
            axis_cross<AX>(d, al, t);                                  // r x al = -(al x r)
            const T wr = d * w[AX];                                    // w . r
            T ww = T(0);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k != AX) ww = num<T>::fma(w[k], w[k], ww);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // w x (w x r) = w (w . r) - r |w|^2: component AX is -d (|w|^2 - w_AX^2)
                const T cen = (k == AX) ? -(d * ww) : wr * w[k];
                ao[k] = (a[k] - t[k]) + cen;
            }
        }
        T wc[3], alc[3];
        to_child<i>(g, w, wc);
        to_child<i>(g, al, alc);
        to_child<i>(g, ao, a);
        // the joint: axis e_C of the child frame
        const T qd = dq[i];
        {   // al += e_C ddq + w x (e_C qd);  w += e_C qd
            T t[3];
            axis_cross<C>(qd, wc, t);                                  // (qd e_C) x w = -(w x e_C qd)
#pragma unroll
            for (int k = 0; k < 3; ++k) al[k] = alc[k] - t[k];
            if constexpr (!ZERO_ACC) al[C] += ddq[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) w[k] = wc[k];
            w[C] += qd;
        }
        // the body: F = m a_com, N = I al + w x (I w)   (constants of atacom_iiwa_inertia.h as literals)
        const T com[3] = {(T)iiwa_body::COM[i][0], (T)iiwa_body::COM[i][1], (T)iiwa_body::COM[i][2]};
        const T In[6] = {(T)iiwa_body::INERTIA[i][0], (T)iiwa_body::INERTIA[i][1], (T)iiwa_body::INERTIA[i][2],
                         (T)iiwa_body::INERTIA[i][3], (T)iiwa_body::INERTIA[i][4], (T)iiwa_body::INERTIA[i][5]};
        T t1[3], t2[3], t3[3];
        crossv(w, com, t1); crossv(w, t1, t2); crossv(al, com, t3);
#pragma unroll
        for (int k = 0; k < 3; ++k) F[i][k] = (T)iiwa_body::MASS[i] * (a[k] + t3[k] + t2[k]);
        T Iw[3], Ial[3];
        symv(In, w, Iw); symv(In, al, Ial);
        crossv(w, Iw, t1);
#pragma unroll
        for (int k = 0; k < 3; ++k) Nm[i][k] = Ial[k] + t1[k];
        ATACOM_LK_BODY();
    });
    // tip to base: (f, n) of the subtree beyond joint i, in body i's frame, n about joint i's origin
    T f[3] = {T(0), T(0), T(0)}, n[3] = {T(0), T(0), T(0)};
    static_for<0, 9>([&](auto kc) {
        constexpr int i = 8 - decltype(kc)::value;
        const T com[3] = {(T)iiwa_body::COM[i][0], (T)iiwa_body::COM[i][1], (T)iiwa_body::COM[i][2]};
        T t1[3];
        crossv(com, F[i], t1);
#pragma unroll
        for (int k = 0; k < 3; ++k) { f[k] += F[i][k]; n[k] += Nm[i][k] + t1[k]; }
        tau[i] = n[PC[i]];
        if constexpr (i > 0) {
            // into the parent's frame, moment about the parent joint's origin: n_p = R n + r x (R f)
            T fp[3], np[3];
            to_parent<i>(g, f, fp);
            to_parent<i>(g, n, np);
            if constexpr (TAX[i] >= 0) {
                T t[3];
                axis_cross<TAX[i]>((T)TOFF[i], fp, t);
#pragma unroll
                for (int k = 0; k < 3; ++k) np[k] += t[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { f[k] = fp[k]; n[k] = np[k]; }
        }
        ATACOM_LK_BODY();
    });
}

// Mass-matrix entries Ml[i][j] (j <= i, j < NB) for the rows i < NR, and optionally the diagonal entries of the rows >= NB
// (dg[i - NB]) -- the interface of atacom_dynamics.h's crba.  Composite inertias (mass, first moment, inertia about the
// joint origin) are carried from the tip towards the base in link coordinates; row i is the momentum (p, L) of composite i
// under joint i's unit velocity, walked down the chain: M_ij = e_C(j) . L in frame j.
template <typename T, int NB, int NR = NB>
__device__ __forceinline__ void crba(const Trig9<T>& g, T (&Ml)[NR][NB], T* dg = nullptr) {
    static_assert(NR >= NB && NR <= 9, "");
    T m = T(0), h[3] = {T(0), T(0), T(0)}, Io[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    static_for<0, 9>([&](auto kc) {
        constexpr int i = 8 - decltype(kc)::value;
        // composite i = body i + (composite i + 1 moved into frame i, about joint i's origin)
        if constexpr (i < 8) {
            constexpr int ch = i + 1;
            T hp[3], Ip[6];
            to_parent<ch>(g, h, hp);
            sym_to_parent<ch>(g, Io, Ip);
            if constexpr (TAX[ch] >= 0) {
                // reference point o_ch -> o_i with o_ch = o_i + r, r = d e_AX:
                //   I += m (|r|^2 1 - r r^T) + 2 (r . h) 1 - r h^T - h r^T,   h += m r
                constexpr int AX = TAX[ch];
                const T d = (T)TOFF[ch];
                const T md = m * d;
                const T dh = d * hp[AX];
                const T iso = num<T>::fma(md, d, T(2) * dh);                       // m d^2 + 2 d h_AX on the two other diagonals
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (k != AX) {
                        Ip[sidx(k, k)] += iso;
                        Ip[sidx(k, AX)] -= d * hp[k];                                  // - r h^T - h r^T, off-diagonal (k, AX)
                    }
                }
                // diagonal (AX, AX): m (d^2 - d^2) + 2 d h_AX - 2 d h_AX = 0
                hp[AX] += md;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) h[k] = hp[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) Io[k] = Ip[k];
        }
        m += (T)iiwa_body::MASS[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) h[k] += (T)own_h(i, k);
        Io[0] += (T)own_I(i, 0, 0); Io[1] += (T)own_I(i, 0, 1); Io[2] += (T)own_I(i, 0, 2);
        Io[3] += (T)own_I(i, 1, 1); Io[4] += (T)own_I(i, 1, 2); Io[5] += (T)own_I(i, 2, 2);
        if constexpr (i < NR) {
            constexpr int C = PC[i];
            // unit velocity of joint i: omega = e_C, the origin at rest: p = e_C x h, L = Io e_C (about the origin)
            T p[3], L[3];
            axis_cross<C>(T(1), h, p);
            L[0] = Io[sidx(0, C)]; L[1] = Io[sidx(1, C)]; L[2] = Io[sidx(2, C)];
            if constexpr (i < NB) Ml[i][i] = L[C];
            else if (dg) dg[i - NB] = L[C];
            // down the chain: frame k -> frame k - 1
            static_for<0, i>([&](auto jc) {
                constexpr int k = i - decltype(jc)::value;          // current frame; moving to k - 1
                T pp[3], Lp[3];
                to_parent<k>(g, p, pp);
                to_parent<k>(g, L, Lp);
                if constexpr (TAX[k] >= 0) {
                    T t[3];
                    axis_cross<TAX[k]>((T)TOFF[k], pp, t);
#pragma unroll
                    for (int d2 = 0; d2 < 3; ++d2) Lp[d2] += t[d2];
                }
#pragma unroll
                for (int d2 = 0; d2 < 3; ++d2) { p[d2] = pp[d2]; L[d2] = Lp[d2]; }
                if constexpr (k - 1 < NB) Ml[i][k - 1] = L[PC[k - 1]];
            });
        }
        ATACOM_LK_BODY();
    });
}

// world axes the servo set-points need (env_single.py:137-185; joint7_target / universal_target of atacom_dynamics.h):
// z6 = axis of joint 6 (= link_7's y axis with joint 7 at zero), z7 = axis of joint 7 (link_7's z), y7 = link_7's y axis.
// Only the orientation chain -- no origins, centres of mass or tensors.
template <typename T>
__device__ __forceinline__ void servo_axes(const Trig9<T>& g, T (&z6)[3], T (&z7)[3], T (&y7)[3]) {
    T X[3] = {T(1), T(0), T(0)}, Y[3] = {T(0), T(1), T(0)}, Z[3] = {T(0), T(0), T(1)};
    constexpr int kind[7] = {0, 1, 1, 2, 1, 2, 1};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        T nx[3], ny[3], nz[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (kind[i] == 0) { nx[d] = X[d]; ny[d] = Y[d]; nz[d] = Z[d]; }
            else if (kind[i] == 1) { nx[d] = -X[d]; ny[d] = Z[d]; nz[d] = Y[d]; }
            else { nx[d] = X[d]; ny[d] = Z[d]; nz[d] = -Y[d]; }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            X[d] = num<T>::fma(g.c[i], nx[d], g.s[i] * ny[d]);
            Y[d] = num<T>::fma(g.c[i], ny[d], -(g.s[i] * nx[d]));
            Z[d] = nz[d];
        }
        if (i == 5) {
#pragma unroll
            for (int d = 0; d < 3; ++d) z6[d] = Z[d];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { z7[d] = Z[d]; y7[d] = Y[d]; }
}

}  // namespace lk
}  // namespace atacom
