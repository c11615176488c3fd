// Row N4: rigid-body dynamics of the iiwa + striker chain (nine movable joints of the reference's urdf/iiwa_1.urdf),
// one evaluation per lane, everything in registers.
//
// What it replaces in the reference (iiwa "7H", torque control):
//   * acc_to_ctrl_action            /root/reference/atacom/environments/iiwa_air_hockey/iiwa_hit_atacom.py:58-63
//                                   PyBullet calculateInverseDynamics(q, dq, ddq padded with zeros)[:6]
//   * the physics sub-step          PyBullet stepSimulation under those torques, URDF joint damping
//                                   (urdf/iiwa_1.urdf:77,115,152,189,226,263,300), joint 7 and the universal joint held by
//                                   position servos (env_base.py:62-70, env_single.py:137-185)
// Bullet is a third-party dependency that is neither vendored nor installed, so this is the model of this build
// (DESIGN.md section 4a), identical in oracle/dynamics.py, whose inverse dynamics and mass matrix are pinned to the
// reference's URDF file (golden set G11).  The constants come from atacom_iiwa_inertia.h (generated from that URDF).
//
// Algorithms (Featherstone 2008), written in WORLD coordinates so that the joint axes / origins of the forward
// kinematics are reused as they are:
//   rnea9   recursive Newton-Euler: tau = M(q) ddq + C(q, dq) dq + g(q)
//   crba6   composite rigid bodies: spatial inertias about the world origin (m, m c, I_O) simply add up along the
//           chain; M_ij = a_j . L_i + (o_j x a_j) . p_i with (L_i, p_i) the momentum of composite i under joint i's unit
//           motion.  Only the 6 x 6 block of the controlled joints is formed.
//   chol6   Cholesky solve of that block.
#pragma once
#include "atacom_envs.h"
#include "atacom_iiwa_inertia.h"
#include "atacom_dynamics_link.h"
// round 5: the kernels evaluate the recursions in LINK coordinates (atacom_dynamics_link.h); the world-coordinate forms below
// stay as the A/B build (-DATACOM_DYN_LINK=0) and as the readable statement of the same equations
#ifndef ATACOM_DYN_LINK
#define ATACOM_DYN_LINK 1
#endif

namespace atacom {

template <typename T>
struct Chain9 {
    T a[9][3];      // joint axes (world)
    T o[9][3];      // joint origins
    T c[9][3];      // centres of mass of the nine bodies
    T Iw[9][6];     // inertia about the centre of mass, world axes: xx, xy, xz, yy, yz, zz
};

template <typename T>
__device__ __forceinline__ void cross3(const T (&u)[3], const T (&v)[3], T (&w)[3]) {
    w[0] = num<T>::fma(u[1], v[2], -(u[2] * v[1]));
    w[1] = num<T>::fma(u[2], v[0], -(u[0] * v[2]));
    w[2] = num<T>::fma(u[0], v[1], -(u[1] * v[0]));
}
template <typename T>
__device__ __forceinline__ void symmul(const T (&S)[6], const T (&v)[3], T (&w)[3]) {      // w = S v, S symmetric
    w[0] = num<T>::fma(S[0], v[0], num<T>::fma(S[1], v[1], S[2] * v[2]));
    w[1] = num<T>::fma(S[1], v[0], num<T>::fma(S[3], v[1], S[4] * v[2]));
    w[2] = num<T>::fma(S[2], v[0], num<T>::fma(S[4], v[1], S[5] * v[2]));
}

// forward kinematics of all nine bodies + world inertials.  Joint origins of the arm as in iiwa_fk (atacom_envs.h);
// joint 7: offset 0.081 along y of link_6 (urdf:295); the striker's universal joint sits 0.585 up the z axis of link_7
// (urdf:328-343,380-384): striker_joint_1 turns about the local y axis, striker_joint_2 about the local x axis (:383,396).
template <typename T>
__device__ __forceinline__ void iiwa_chain9(const T (&q)[9], Chain9<T>& k) {
    T X[3] = {T(1), T(0), T(0)}, Y[3] = {T(0), T(1), T(0)}, Z[3] = {T(0), T(0), T(1)};
    T o[3] = {T(0), T(0), T(0)};
    constexpr T off[7] = {T(0.1575), T(0.2025), T(0.2045), T(0.2155), T(0.1845), T(0.2155), T(0.081)};
    constexpr int off_axis[7] = {2, 2, 1, 2, 1, 2, 1};           // 2 = along Z, 1 = along Y of the parent
    constexpr int kind[7] = {0, 1, 1, 2, 1, 2, 1};               // 0 identity, 1: (-x, z, y), 2: (x, z, -y)
    auto body = [&](int i) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            k.o[i][d] = o[d];
            k.c[i][d] = num<T>::fma(X[d], (T)iiwa_body::COM[i][0],
                                    num<T>::fma(Y[d], (T)iiwa_body::COM[i][1], num<T>::fma(Z[d], (T)iiwa_body::COM[i][2], o[d])));
        }
        // Iw = R I R^T with R = [X Y Z]
        const T I0 = (T)iiwa_body::INERTIA[i][0], I1 = (T)iiwa_body::INERTIA[i][1], I2 = (T)iiwa_body::INERTIA[i][2],
                I3 = (T)iiwa_body::INERTIA[i][3], I4 = (T)iiwa_body::INERTIA[i][4], I5 = (T)iiwa_body::INERTIA[i][5];
        T t[3][3];                                               // t = R I: column j = X I_0j + Y I_1j + Z I_2j
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            t[d][0] = num<T>::fma(X[d], I0, num<T>::fma(Y[d], I1, Z[d] * I2));
            t[d][1] = num<T>::fma(X[d], I1, num<T>::fma(Y[d], I3, Z[d] * I4));
            t[d][2] = num<T>::fma(X[d], I2, num<T>::fma(Y[d], I4, Z[d] * I5));
        }
        auto e = [&](int r, int cc) { return num<T>::fma(t[r][0], X[cc], num<T>::fma(t[r][1], Y[cc], t[r][2] * Z[cc])); };
        k.Iw[i][0] = e(0, 0); k.Iw[i][1] = e(0, 1); k.Iw[i][2] = e(0, 2);
        k.Iw[i][3] = e(1, 1); k.Iw[i][4] = e(1, 2); k.Iw[i][5] = e(2, 2);
    };
#pragma unroll
    for (int i = 0; i < 7; ++i) {
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = num<T>::fma(off_axis[i] == 2 ? Z[d] : Y[d], off[i], o[d]);
        T nx[3], ny[3], nz[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (kind[i] == 0) { nx[d] = X[d]; ny[d] = Y[d]; nz[d] = Z[d]; }
            else if (kind[i] == 1) { nx[d] = -X[d]; ny[d] = Z[d]; nz[d] = Y[d]; }
            else { nx[d] = X[d]; ny[d] = Z[d]; nz[d] = -Y[d]; }
        }
        T s, c;
        num<T>::sincos(q[i], &s, &c);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            X[d] = num<T>::fma(c, nx[d], s * ny[d]);
            Y[d] = num<T>::fma(c, ny[d], -(s * nx[d]));
            Z[d] = nz[d];
            k.a[i][d] = Z[d];
        }
        body(i);
    }
    // universal joint at the tip
#pragma unroll
    for (int d = 0; d < 3; ++d) o[d] = num<T>::fma(Z[d], (T)iiwa_body::STRIKER_OFFSET_Z, o[d]);
    {
        T s, c;
        num<T>::sincos(q[7], &s, &c);                             // about the local y axis
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            k.a[7][d] = Y[d];
            const T x = X[d], z = Z[d];
            X[d] = num<T>::fma(c, x, -(s * z));
            Z[d] = num<T>::fma(s, x, c * z);
        }
        body(7);
        num<T>::sincos(q[8], &s, &c);                             // about the (new) local x axis
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            k.a[8][d] = X[d];
            const T y = Y[d], z = Z[d];
            Y[d] = num<T>::fma(c, y, s * z);
            Z[d] = num<T>::fma(c, z, -(s * y));
        }
        body(8);
    }
}

// tau[9] = M(q) ddq + C(q, dq) dq + g(q); gravity (0, 0, -9.81) enters as an upward acceleration of the base
// ZERO_ACC: all joint accelerations are zero (ddq is not read): tau = C(q, dq) dq + g(q), the bias of the equation of motion
template <typename T, bool ZERO_ACC = false>
__device__ __forceinline__ void rnea9(const Chain9<T>& k, const T (&dq)[9], const T (&ddq)[9], T (&tau)[9]) {
    T w[3] = {T(0), T(0), T(0)}, al[3] = {T(0), T(0), T(0)}, ao[3] = {T(0), T(0), T(9.81)};
    T F[9][3], Nm[9][3];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        T r[3], t1[3], t2[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) r[d] = (i == 0) ? k.o[0][d] : k.o[i][d] - k.o[i > 0 ? i - 1 : 0][d];
        cross3(w, r, t1); cross3(w, t1, t2); cross3(al, r, t1);
#pragma unroll
        for (int d = 0; d < 3; ++d) ao[d] += t1[d] + t2[d];
        T adq[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) adq[d] = k.a[i][d] * dq[i];
        cross3(w, adq, t1);
#pragma unroll
        for (int d = 0; d < 3; ++d) { al[d] += ZERO_ACC ? t1[d] : num<T>::fma(k.a[i][d], ddq[i], t1[d]); w[d] += adq[d]; }
        T rc[3], ac[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) rc[d] = k.c[i][d] - k.o[i][d];
        cross3(w, rc, t1); cross3(w, t1, t2); cross3(al, rc, t1);
#pragma unroll
        for (int d = 0; d < 3; ++d) { ac[d] = ao[d] + t1[d] + t2[d]; F[i][d] = (T)iiwa_body::MASS[i] * ac[d]; }
        T Iw_w[3], Ial[3];
        symmul(k.Iw[i], w, Iw_w); symmul(k.Iw[i], al, Ial);
        cross3(w, Iw_w, t1);
#pragma unroll
        for (int d = 0; d < 3; ++d) Nm[i][d] = Ial[d] + t1[d];
    }
    T f[3] = {T(0), T(0), T(0)}, n[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int i = 8; i >= 0; --i) {
        T t1[3];
        if (i < 8) {
            T r[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) r[d] = k.o[i < 8 ? i + 1 : 8][d] - k.o[i][d];
            cross3(r, f, t1);
#pragma unroll
            for (int d = 0; d < 3; ++d) n[d] += t1[d];
        }
        T rc[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { f[d] += F[i][d]; rc[d] = k.c[i][d] - k.o[i][d]; }
        cross3(rc, F[i], t1);
#pragma unroll
        for (int d = 0; d < 3; ++d) n[d] += Nm[i][d] + t1[d];
        tau[i] = num<T>::fma(n[0], k.a[i][0], num<T>::fma(n[1], k.a[i][1], n[2] * k.a[i][2]));
    }
}

// Mass matrix entries (lower triangle, row-major Ml[i][j], j <= i, j < NB) for the rows i < NR: NR = NB gives the NB x NB
// leading block of the controlled joints, NR = 9 adds the coupling rows of the three servo joints (M[6..8][0..5], used
// to put their accelerations on the right-hand side of the controlled joints' equation).  Composites are accumulated
// from the tip (all nine bodies contribute).
// dg (optional): the diagonal entries M_ii of the rows i >= NB (the servo joints' own inertias), dg[i - NB].
template <typename T, int NB, int NR = NB>
__device__ __forceinline__ void crba(const Chain9<T>& k, T (&Ml)[NR][NB], T* dg = nullptr) {
    static_assert(NR >= NB && NR <= 9, "");
    T vj[NB][3];                                            // velocity of joint j's body-fixed point at the world origin
#pragma unroll
    for (int j = 0; j < NB; ++j) cross3(k.o[j], k.a[j], vj[j]);
    T m = T(0), h[3] = {T(0), T(0), T(0)}, Io[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int i = 8; i >= 0; --i) {
        const T mi = (T)iiwa_body::MASS[i];
        const T cx = k.c[i][0], cy = k.c[i][1], cz = k.c[i][2];
        const T cc = num<T>::fma(cx, cx, num<T>::fma(cy, cy, cz * cz));
        m += mi;
        h[0] = num<T>::fma(mi, cx, h[0]); h[1] = num<T>::fma(mi, cy, h[1]); h[2] = num<T>::fma(mi, cz, h[2]);
        Io[0] += k.Iw[i][0] + mi * (cc - cx * cx);
        Io[1] += k.Iw[i][1] - mi * cx * cy;
        Io[2] += k.Iw[i][2] - mi * cx * cz;
        Io[3] += k.Iw[i][3] + mi * (cc - cy * cy);
        Io[4] += k.Iw[i][4] - mi * cy * cz;
        Io[5] += k.Iw[i][5] + mi * (cc - cz * cz);
        if (i < NR) {
            T vi[3], p[3], L[3], t1[3];
            if (i < NB) { vi[0] = vj[i < NB ? i : 0][0]; vi[1] = vj[i < NB ? i : 0][1]; vi[2] = vj[i < NB ? i : 0][2]; }
            else cross3(k.o[i], k.a[i], vi);
            cross3(k.a[i], h, t1);
#pragma unroll
            for (int d = 0; d < 3; ++d) p[d] = num<T>::fma(m, vi[d], t1[d]);
            symmul(Io, k.a[i], L);
            cross3(h, vi, t1);
#pragma unroll
            for (int d = 0; d < 3; ++d) L[d] += t1[d];
            if (dg && i >= NB) {
                T v = num<T>::fma(k.a[i][0], L[0], num<T>::fma(k.a[i][1], L[1], k.a[i][2] * L[2]));
                dg[i >= NB ? i - NB : 0] = num<T>::fma(vi[0], p[0], num<T>::fma(vi[1], p[1], num<T>::fma(vi[2], p[2], v)));
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j > i) continue;
                T v = num<T>::fma(k.a[j][0], L[0], num<T>::fma(k.a[j][1], L[1], k.a[j][2] * L[2]));
                v = num<T>::fma(vj[j][0], p[0], num<T>::fma(vj[j][1], p[1], num<T>::fma(vj[j][2], p[2], v)));
                Ml[i < NR ? i : 0][j] = v;
            }
        }
    }
}

// x = A^-1 b for the symmetric positive definite A given by its lower triangle (overwritten by its Cholesky factor)
// (A may carry NR - NB further rows below the block: they are not touched)
template <typename T, int NB, int NR = NB>
__device__ __forceinline__ void chol_solve(T (&A)[NR][NB], T (&b)[NB]) {
    T inv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        T d = A[j][j];
#pragma unroll
        for (int k2 = 0; k2 < j; ++k2) d = num<T>::fma(-A[j][k2], A[j][k2], d);
        const T l = num<T>::sqrt(d);
        inv[j] = num<T>::rcp(l);
        A[j][j] = l;
#pragma unroll
        for (int i = j + 1; i < NB; ++i) {
            T v = A[i][j];
#pragma unroll
            for (int k2 = 0; k2 < j; ++k2) v = num<T>::fma(-A[i][k2], A[j][k2], v);
            A[i][j] = v * inv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {                          // L y = b
        T v = b[i];
#pragma unroll
        for (int k2 = 0; k2 < i; ++k2) v = num<T>::fma(-A[i][k2], b[k2], v);
        b[i] = v * inv[i];
    }
#pragma unroll
    for (int i = NB - 1; i >= 0; --i) {                     // L^T x = y
        T v = b[i];
#pragma unroll
        for (int k2 = i + 1; k2 < NB; ++k2) v = num<T>::fma(-A[k2][i], b[k2], v);
        b[i] = v * inv[i];
    }
}

// ---- servo set-points (env_single.py:137-185) from the arm's forward kinematics
// joint 7: the angle that keeps the striker's y axis horizontal, evaluated with joint 7 at zero (env_single.py:139-142)
template <typename T>
__device__ __forceinline__ T joint7_target(const T (&Y)[3], const T (&Z)[3], T q7_cur) {
    // Y, Z: the y and z axes of link_7 with joint 7 AT ZERO.  They come from the chain the dynamics have just evaluated:
    // joint 7's origin frame maps link_6's (x, y, z) to (-x, z, y) (urdf:295, kind 1 in iiwa_chain9), so at q7 = 0 link_7's
    // y axis is link_6's z axis = the axis of joint 6 (Chain9::a[5]) and its z axis is joint 7's own axis (a[6]) -- no second
    // pass through the seven joint rotations (it had cost seven sincos and ~400 instructions per physics sub-step).
    const T down[3] = {T(0), T(0), T(-1)};
    T yd[3];
    cross3(down, Z, yd);                                                          // :144
    const T nrm = num<T>::sqrt(num<T>::fma(yd[0], yd[0], num<T>::fma(yd[1], yd[1], yd[2] * yd[2])));
    const bool big = nrm > T(1e-2);
    const T inrm = big ? num<T>::rcp(nrm) : T(1);
#pragma unroll
    for (int d = 0; d < 3; ++d) yd[d] = big ? yd[d] * inrm : Z[d];                // :146-150
    T dot = num<T>::fma(Y[0], yd[0], num<T>::fma(Y[1], yd[1], Y[2] * yd[2]));
    dot = num<T>::min(num<T>::max(dot, T(-1)), T(1));
    T target = num<T>::acos(dot);                                                 // :152
    T ax[3];
    cross3(Y, yd, ax);
    const T an = num<T>::sqrt(num<T>::fma(ax[0], ax[0], num<T>::fma(ax[1], ax[1], ax[2] * ax[2])));
    const bool abig = an > T(1e-2);
    const T ian = abig ? num<T>::rcp(an) : T(1);
    const T sgn = abig ? (ax[0] * Z[0] + ax[1] * Z[1] + ax[2] * Z[2]) * ian : Z[2];   // axis . z, axis = (0,0,1) fallback
    target *= sgn;                                                                // :161
    const T pi = T(3.14159265358979323846);
    if (target - q7_cur > pi / 2) target -= pi;                                   // :163-166
    else if (target - q7_cur < -pi / 2) target += pi;
    return target;
}

// universal joint: tilt of link_7's z axis against the table normal, signed by its y axis (env_single.py:169-185)
template <typename T>
__device__ __forceinline__ T universal_target(const T (&z7)[3], const T (&y7)[3]) {
    const T dot = num<T>::min(num<T>::max(-z7[2], T(-1)), T(1));
    const T q1 = num<T>::acos(dot);
    const T down[3] = {T(0), T(0), T(-1)};
    T ax[3];
    cross3(z7, down, ax);
    const T an = num<T>::sqrt(num<T>::fma(ax[0], ax[0], num<T>::fma(ax[1], ax[1], ax[2] * ax[2])));
    const bool abig = an > T(1e-2);
    const T ian = abig ? num<T>::rcp(an) : T(1);
    const T sgn = abig ? (ax[0] * y7[0] + ax[1] * y7[1] + ax[2] * y7[2]) * ian : y7[2];
    return q1 * sgn;
}

}  // namespace atacom
