// Kernels of the batched ATACOM step, templated on the scalar type and the environment.
//
// Mapping (DESIGN.md "Kernel design"): ONE ENVIRONMENT PER LANE, 64 consecutive environments per
// wavefront, one wavefront per workgroup (or 2 / 4 / 8 lanes per environment: atacom_quad.h).  Persistent per-env
// state lives in HBM as one 16-byte-aligned record per environment, read and written with dwordx4 accesses off one
// address (see Planes below for why not field planes);
// all per-env matrices (J_c 12x17, null basis 17x5, ...) live in VGPRs -- the kernels are compiled for
// one wave per SIMD (__launch_bounds__(64)) to get the full 512-register budget.
// The four physics sub-steps of an env step, the observation / reward / termination logic and the
// constraint statistics are fused in one kernel; k_rollout additionally keeps the state in registers
// across many env steps (state is read from HBM once and written once per launch).
#pragma once
#include <stdint.h>
#include "atacom_envs.h"
#include "atacom_quad.h"
#include "atacom_policy.h"
#include "atacom_dynamics.h"
#include "atacom_chart.h"

namespace atacom {

constexpr int WAVE = 64;
// rigid-body kernels: park the held solver state in LDS across the dynamics (env_step); -DATACOM_DYN_PARK=0: the A/B build
#ifndef ATACOM_DYN_PARK
#define ATACOM_DYN_PARK 0           // round 5: off -- with the dynamics in link coordinates (atacom_dynamics_link.h) the quad kernels
#endif                              // fit without it (no scratch, 130 AGPRs) and run 42.7 instead of 45.6 us per step
// threads per workgroup of the step / rollout kernels: the quad mapping runs 2.7 % faster with four waves per
// workgroup (one per SIMD of a CU, sharing the instruction cache), the lane mapping with one (measured, profiles/)
#ifndef ATACOM_BLOCK_GROUP
#define ATACOM_BLOCK_GROUP 256
#endif
template <int LANES> constexpr int BLOCK = (LANES > 1) ? ATACOM_BLOCK_GROUP : 64;

// ------------------------------------------------------------------ layout of the per-handle state buffer
// Fields are kept in GROUPS OF FOUR: [group][env][4] -- a lane reads / writes 16 consecutive bytes (global_load_dwordx4 /
// global_store_dwordx4), the environments of a wave are 16 bytes apart, so every access instruction of a wave is one
// contiguous, fully used segment (1 KB with one environment per lane) however the environments are mapped to lanes.
// Two regions of the one allocation:
//   hot   the fields a step reads and writes back: q, dq, s, puck, the hit bookkeeping, the statistics accumulators --
//         34 values for the iiwa task = 9 groups (2 pad values);
//   cold  the initial state a reset restores (and the servo joints of the rigid-body mode).
// History (profiles/r02_lanes_vs_batch.md): single-field planes [field][env] (round 1) cost one 64-bit per-lane address
// per field, computed up front for the loads, kept alive over the whole step for the store tail and therefore parked in
// AGPRs -- ~290 vector instructions and 68 memory instructions per step of the iiwa kernels; one record per
// environment [env][36] (one address, 9 dwordx4) was 4.5 % faster for the 8-lane mapping but 7 % slower for one
// environment per lane at 65536 environments, where a wave's 64 records are 144 bytes apart and no single access is
// contiguous.  Groups of four keep the wide accesses and the coalescing.
// Field numbering is unchanged from the plane era ("plane" = field index); pl() maps it to the buffer.
template <typename E>
struct Planes {
    // (round 5: environments without a puck -- the circle family -- no longer carry the eight puck / hit fields and the six
    // stored puck values: their step moved 186 bytes per environment against 60 algorithmic, profiles/r05_pmc_summary.md)
    static constexpr int NP = E::PUCK ? 6 : 0, NH = E::PUCK ? 1 : 0;
    static constexpr int Q = 0, DQ = Q + E::NQ, S = DQ + E::NQ, PUCK = S + E::NG, RHIT = PUCK + NP, VHX = RHIT + NH,
                         SSUM = VHX + NH, SCMAX = SSUM + 1, SDQMAX = SCMAX + 1, HOT = SDQMAX + 1,
                         IQ = HOT, IDQ = IQ + E::NQ, IS = IDQ + E::NQ, IPUCK = IS + E::NG,
                         // row N4 (iiwa): the three servo joints of the rigid-body mode, positions then velocities
                         QX = IPUCK + NP, DQX = QX + 3, AUX_END = (E::ID == 2) ? DQX + 3 : QX,
                         // obs_delay: the low-pass state of the observation's velocities, puck (3) then joints (NQ)
                         FV = AUX_END, COUNT = E::PUCK ? FV + 3 + E::NQ : FV;
    static constexpr int HOT_LD = (HOT + 3) / 4 * 4, COLD_LD = (COUNT - HOT + 3) / 4 * 4;
    static constexpr int VALUES_PER_ENV = HOT_LD + COLD_LD;          // allocation: VALUES_PER_ENV * batch elements
    static constexpr int I_HIT = 0, I_T = 1, I_CNT = 2, I_EP = 3, ICOUNT = 4;   // I_EP: episodes started (RNG counter)
    static constexpr int STATE_DIM = 2 * E::NQ + E::NG + 6 + 4;
    static constexpr int INIT_DIM = 2 * E::NQ + (E::PUCK ? 6 : 0);
};
// field `plane` of environment b (buffers come from hipMalloc: 256-byte aligned; every group element is 16-byte aligned)
template <typename E, typename V>
__device__ __forceinline__ V& pl(V* f, int plane, int B, int b) {
    using L = Planes<E>;
    V* a = static_cast<V*>(__builtin_assume_aligned(f, 16));
    const int p = (plane < L::HOT) ? plane : plane - L::HOT + L::HOT_LD;      // cold groups follow the hot ones
    return a[((size_t)(p / 4) * B + b) * 4 + (p % 4)];
}
// the integer fields: one int4 per environment
template <typename V>
__device__ __forceinline__ V& pli(V* ip, int plane, int b) {
    V* a = static_cast<V*>(__builtin_assume_aligned(ip, 16));
    return a[(size_t)b * 4 + plane];
}

// Packed rollout record of one (step, env): the (s, a, r, s', absorbing, last) tuple mushroom_rl.Core collects, as ONE
// run of F floats -- the layout the sharded collector all-gathers without a repacking pass (rollout.py).
template <typename E>
struct Record {
    static constexpr int OBS = 0, ACT = E::OBS, REW = ACT + E::NK, NOBS = REW + 1, ABS = NOBS + E::OBS, LAST = ABS + 1,
                         F = LAST + 1;
};

template <typename T, typename E>
struct EnvState {
    T q[E::NQ], dq[E::NQ], s[E::NG], puck[6];
    T r_hit, vel_hit_x;
    int has_hit, t;
    T qx[3], dqx[3];            // servo joints (rigid-body mode only; untouched otherwise)
};

// Where an environment's persistent state lives: handed to the pieces of a step that only the domain-randomisation
// options (Params::noise, off by default) touch.  NOISE is a COMPILE-TIME switch of the stepping kernels: the handles that
// ask for obs_noise / obs_delay / env_noise run their own instantiations (atacom_noise_*.hip), everybody else runs kernels
// that do not contain the options at all.  (First built as launch-uniform branches of the one kernel set, with the
// options' state kept out of the registers: the untaken branches still cost 2-4 % per step -- iiwa quad 28.1 -> 29.1 us,
// 8 lanes 26.1 -> 27.0, planar 11.0 -> 11.4 on one box, profiles/r04_ab_noise_options.log -- through the register
// allocation of kernels that run at the edge of the register file.)  Inside the NOISE kernels the options' state -- the
// low-pass of obs_delay (planes FV), the id of the running episode that keys the draws (I_EP - 1) -- is read from /
// written to the state buffer where it is needed instead of being carried across the solver; a thread only ever re-reads
// what it wrote itself (every lane of a group stores the same bits), which needs no fence.
template <typename T, bool NOISE>
struct EnvRef {
    static constexpr bool noise = NOISE;
    T* f;
    int* ip;
    int B, b;
    bool commit;                // false: a lane shadowing another lane's environment (k_rollout_mlp) -- no stores
};

// the servo-joint planes are loaded / stored only by the rigid-body kernels (DYN)
template <typename T, typename E>
__device__ __forceinline__ void load_aux(const T* __restrict__ f, int B, int b, EnvState<T, E>& st) {
    using L = Planes<E>;
#pragma unroll
    for (int i = 0; i < 3; ++i) { st.qx[i] = pl<E>(f, L::QX + i, B, b); st.dqx[i] = pl<E>(f, L::DQX + i, B, b); }
}
template <typename T, typename E>
__device__ __forceinline__ void store_aux(T* __restrict__ f, int B, int b, const EnvState<T, E>& st) {
    using L = Planes<E>;
#pragma unroll
    for (int i = 0; i < 3; ++i) { pl<E>(f, L::QX + i, B, b) = st.qx[i]; pl<E>(f, L::DQX + i, B, b) = st.dqx[i]; }
}

template <typename T>
struct StepOut {
    T reward;
    bool absorbing, last;
    T log_avg, log_max, log_dq;
#ifdef ATACOM_TIMESTAMPS
    int dbg[3] = {0, 0, 0};        // tuning build: wave-level path counters of the canonical chart (atacom_chart.h)
#endif
};

template <typename T, typename E>
__device__ __forceinline__ void load_state(const T* __restrict__ f, const int* __restrict__ ip, int B, int b,
                                           EnvState<T, E>& st) {
    using L = Planes<E>;
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) { st.q[i] = pl<E>(f, L::Q + i, B, b); st.dq[i] = pl<E>(f, L::DQ + i, B, b); }
#pragma unroll
    for (int i = 0; i < E::NG; ++i) st.s[i] = pl<E>(f, L::S + i, B, b);
    if (E::PUCK) {
#pragma unroll
        for (int i = 0; i < 6; ++i) st.puck[i] = pl<E>(f, L::PUCK + i, B, b);
        st.r_hit = pl<E>(f, L::RHIT, B, b);
        st.vel_hit_x = pl<E>(f, L::VHX, B, b);
        st.has_hit = pli(ip, L::I_HIT, b);
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) st.puck[i] = T(0);
        st.r_hit = st.vel_hit_x = T(0);
        st.has_hit = 0;
    }
    st.t = pli(ip, L::I_T, b);
}

template <typename T, typename E>
__device__ __forceinline__ void store_state(T* __restrict__ f, int* __restrict__ ip, int B, int b,
                                            const EnvState<T, E>& st) {
    using L = Planes<E>;
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) { pl<E>(f, L::Q + i, B, b) = st.q[i]; pl<E>(f, L::DQ + i, B, b) = st.dq[i]; }
#pragma unroll
    for (int i = 0; i < E::NG; ++i) pl<E>(f, L::S + i, B, b) = st.s[i];
    if (E::PUCK) {
#pragma unroll
        for (int i = 0; i < 6; ++i) pl<E>(f, L::PUCK + i, B, b) = st.puck[i];
        pl<E>(f, L::RHIT, B, b) = st.r_hit;
        pl<E>(f, L::VHX, B, b) = st.vel_hit_x;
        pli(ip, L::I_HIT, b) = st.has_hit;
    }
    pli(ip, L::I_T, b) = st.t;
}

template <typename T, typename E>
__device__ __forceinline__ void load_init(const T* __restrict__ f, int B, int b, EnvState<T, E>& st) {
    using L = Planes<E>;
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) { st.q[i] = pl<E>(f, L::IQ + i, B, b); st.dq[i] = pl<E>(f, L::IDQ + i, B, b); }
#pragma unroll
    for (int i = 0; i < E::NG; ++i) st.s[i] = pl<E>(f, L::IS + i, B, b);
#pragma unroll
    for (int i = 0; i < 6; ++i) st.puck[i] = E::PUCK ? pl<E>(f, L::IPUCK + i, B, b) : T(0);
    st.r_hit = st.vel_hit_x = T(0);
    st.has_hit = 0;
    st.t = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) st.qx[i] = st.dqx[i] = T(0);      // the servo set-points vanish at the reset pose
}

// ---- random initialisation on the device (A16, the random_init branches of circle_base.py:36-42 and
// env_hitting.py:24-25).  The reference draws from numpy's global, unseeded generator, so only the distribution can
// be matched; here the draws are a counter-based hash of (seed, env index, episode index, draw index), i.e. stateless,
// reproducible, and identical in the oracle (oracle/atacom_batched.py: device_uniform).
__device__ __forceinline__ unsigned int hash_u32(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <typename T>
__device__ __forceinline__ T device_uniform(unsigned int seed, int env, int episode, int draw) {
    const unsigned int key = seed + (unsigned int)env * 0x9E3779B9u + (unsigned int)episode * 0x85EBCA6Bu +
                             (unsigned int)draw * 0xC2B2AE35u;
    return (T)(hash_u32(hash_u32(key)) >> 8) * (T)(1.0 / 16777216.0);        // [0, 1) with 24 bits
}

// Standard normal draw number `idx` (of n_idx per env step) of step t of an episode: Box-Muller on two draws of the
// counter-based generator above (oracle/atacom_batched.py: device_normal).  The reference draws np.random.randn from the
// global, unseeded generator, so only the distribution can match.
template <typename T>
__device__ __forceinline__ T device_normal(unsigned int seed, int env, int episode, int t, int idx, int n_idx) {
    const int d = 8 + 2 * (t * n_idx + idx);
    const T u1 = device_uniform<T>(seed, env, episode, d), u2 = device_uniform<T>(seed, env, episode, d + 1);
    T sn, cs;
    num<T>::sincos(T(6.283185307179586) * u2, &sn, &cs);
    return num<T>::sqrt(T(-2) * num<T>::log(T(1) - u1)) * cs;
}

template <typename T, typename E, typename Ref>
__device__ __forceinline__ void write_obs(const Params<T>& P, const EnvState<T, E>& st, T* __restrict__ o,
                                          const Ref& ref) {
    if (E::ID == 0) {                                           // circle_base.py:83-84
        o[0] = st.q[0]; o[1] = st.q[1]; o[2] = st.dq[0]; o[3] = st.dq[1];
    } else {                                                    // env_single.py:82-120
        o[0] = st.puck[0] - P.base_x; o[1] = st.puck[1] - P.base_y; o[2] = st.puck[2];
        o[3] = st.puck[3]; o[4] = st.puck[4]; o[5] = st.puck[5];
#pragma unroll
        for (int i = 0; i < E::NQ; ++i) { o[6 + i] = st.q[i]; o[6 + E::NQ + i] = st.dq[i]; }
        if constexpr (E::PUCK && Ref::noise) {
            using L = Planes<E>;
            if (P.noise & NOISE_OBS) {
                // env_single.py:105-107; a pure function of (environment, episode, steps taken): the observation of a state is
                // the same whenever it is produced (step, masked step, reset with an empty mask, the T-step kernels)
                const int n_idx = 3 + 2 * P.substeps;
                const int ep = pli(ref.ip, L::I_EP, ref.b) - 1;      // every reset starts an episode: the running one's id
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    o[c] = num<T>::fma(P.obs_noise_std, device_normal<T>(P.seed, ref.b, ep, st.t, c, n_idx), o[c]);
            }
            if (P.noise & NOISE_DELAY) {                         // :114-117: the filtered values (advanced in env_step)
#pragma unroll
                for (int i = 0; i < 3; ++i) o[3 + i] = pl<E>(ref.f, L::FV + i, ref.B, ref.b);
#pragma unroll
                for (int i = 0; i < E::NQ; ++i) o[6 + E::NQ + i] = pl<E>(ref.f, L::FV + 3 + i, ref.B, ref.b);
            }
        }
    }
}

// slack initialisation, atacom.py:145-149
template <typename T, typename E>
__device__ __forceinline__ void slack_init(const Params<T>& P, EnvState<T, E>& st) {
    T fun[E::NC], J[E::NC][E::NQ], bst[E::NC];
    constraint_terms(E{}, P, st.q, st.dq, fun, J, bst);
#pragma unroll
    for (int g = 0; g < E::NG; ++g) {
        const int r = E::NF + g;
        T jdq = T(0);
#pragma unroll
        for (int i = 0; i < E::NQ; ++i) jdq = num<T>::fma(J[r][i], st.dq[i], jdq);
        const T gv = num<T>::fma(P.K[r], jdq, fun[r]);
        st.s[g] = num<T>::sqrt(num<T>::max(T(-2) * gv, T(0)));
    }
}

// state <- stored initial state; with P.random_init (and `draw`) the random part of the reference's reset is re-drawn.
// The episode counter advances whenever something is keyed by it (random_init, the noise options).  Returns with st.s
// valid (stored slack, or recomputed when q / dq were randomised).
template <typename T, typename E, typename Ref>
__device__ __forceinline__ void reset_env(const Params<T>& P, const Ref& ref, EnvState<T, E>& st, bool draw = true) {
    using L = Planes<E>;
    const int B = ref.B, b = ref.b;
    load_init<T, E>(ref.f, B, b, st);
    const bool noisy = E::PUCK && Ref::noise && (P.noise != 0);
    if (!P.random_init && !noisy) return;
    const int ep = pli(ref.ip, L::I_EP, b);
    if (ref.commit) pli(ref.ip, L::I_EP, b) = ep + 1;
    if (P.random_init && draw) {
    if (E::ID == 0) {
        // circle_base.py:36-42
        const T y = T(-0.5) + T(1.5) * device_uniform<T>(P.seed, b, ep, 0);
        const T sg = (device_uniform<T>(P.seed, b, ep, 1) < T(0.5)) ? T(-1) : T(1);
        const T x = num<T>::sqrt(num<T>::max(T(1) - y * y, T(0))) * sg;
        const T dx = T(-1) + T(2) * device_uniform<T>(P.seed, b, ep, 2);
        const T dy = -x * dx / y;
        const T sp = device_uniform<T>(P.seed, b, ep, 3) / num<T>::sqrt(num<T>::fma(dx, dx, dy * dy));
        st.q[0] = x; st.q[1] = y; st.dq[0] = dx * sp; st.dq[1] = dy * sp;
        slack_init<T, E>(P, st);
    } else {
        // env_hitting.py:24-25: puck uniform in hit_range = [-0.6, -0.2] x [-0.4, 0.4] (env_hitting.py:11)
        if (E::ID == 1 && P.task == 1) {
            // task 'D', AirHockeyDefend.setup [upstream]: puck uniform in start_range = [0.25, 0.65] x [-0.4, 0.4], speed
            // uniform in init_velocity_range = (1, 2.2) towards the agent within +-0.5 rad, yaw rate uniform in (-1, 1)
            st.puck[0] = T(0.25) + T(0.4) * device_uniform<T>(P.seed, b, ep, 0);
            st.puck[1] = T(-0.4) + T(0.8) * device_uniform<T>(P.seed, b, ep, 1);
            const T v = T(1) + T(1.2) * device_uniform<T>(P.seed, b, ep, 2);
            const T ang = T(-0.5) + device_uniform<T>(P.seed, b, ep, 3);
            T sa, ca;
            num<T>::sincos(ang, &sa, &ca);
            st.puck[3] = -ca * v;
            st.puck[4] = sa * v;
            st.puck[5] = T(-1) + T(2) * device_uniform<T>(P.seed, b, ep, 4);
        } else {
        st.puck[0] = T(-0.6) + T(0.4) * device_uniform<T>(P.seed, b, ep, 0);
        st.puck[1] = T(-0.4) + T(0.8) * device_uniform<T>(P.seed, b, ep, 1);
        }
    }
    }
    if constexpr (E::PUCK && Ref::noise) {
        if ((P.noise & NOISE_DELAY) && ref.commit) {
            // the first observation of an episode is unfiltered (the reference's obs_prev is None there)
#pragma unroll
            for (int i = 0; i < 3; ++i) pl<E>(ref.f, L::FV + i, B, b) = st.puck[3 + i];
#pragma unroll
            for (int i = 0; i < E::NQ; ++i) pl<E>(ref.f, L::FV + 3 + i, B, b) = st.dq[i];
        }
    }
}

// ------------------------------------------------------------------ row N4: one physics sub-step of the rigid-body mode
// ddq (in: the truncated acceleration ATACOM asks for; out: what the arm does) -- DESIGN.md section 4a:
//   servo joints (POSITION_CONTROL, env_base.py:64-70): velocity set-point v* = clip(0.1 (target - q) / dt, 1.5 v_max),
//           targets env_single.py:137-185; the motor realises dds = (v* - dq) / dt as far as its torque allows:
//           |M_ss,ii dds_i + h_s,i| <= URDF effort limit (iiwa_1.urdf:297,384,400; diagonal estimate of the motor torque);
//   tau   = inverse dynamics of the nine-joint chain for [ddq, 0, 0, 0] at the SIMULATED state (acc_to_ctrl_action,
//           iiwa_hit_atacom.py:58-63), saturated at the URDF effort limits (iiwa_1.urdf:74,112,149,186,223,260);
//           dynamics_mode 2: for [ddq, dds] -- the controller knows what the servo joints are about to do (feed-forward
//           of their reaction on the arm; NOT what the reference computes);
//   ddq   = M_aa^-1 (tau - rnea_a(q, dq, [0; dds]) - D_a dq_a)       (hybrid forward dynamics, URDF joint damping).
// LINK: the recursions in link coordinates (atacom_dynamics_link.h, round 5) or in world coordinates (atacom_dynamics.h).
template <typename T, typename E, bool LINK = true>
__device__ __forceinline__ void rigid_body_substep(const Params<T>& P, EnvState<T, E>& st, T (&ddq)[E::NQ]) {
    static_assert(E::NQ == 6, "iiwa only");
    // phase boundaries (scheduling barriers) belong to the world-coordinate form, whose three passes together overflow the
    // register file when interleaved; around the link-coordinate form they only pin the allocator: with them the lane-mapped
    // single-step kernel takes 76 scratch accesses INSIDE THE SOLVER block (73 -> 83 us per step), without them none
#ifndef ATACOM_LK_PHASE
#define ATACOM_LK_PHASE 0
#endif
    auto phase = [] { if constexpr (!(LINK && ATACOM_DYN_LINK) || ATACOM_LK_PHASE) ATACOM_PHASE(); };
    T q9[9], dq9[9];
#pragma unroll
    for (int i = 0; i < 6; ++i) { q9[i] = st.q[i]; dq9[i] = st.dq[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { q9[6 + i] = st.qx[i]; dq9[6 + i] = st.dqx[i]; }
    ATACOM_MARK("DYN_chain"); phase();
    // The equation of motion is linear in the accelerations, tau = M(q) ddq + h(q, dq), so ONE recursive Newton-Euler
    // pass (h: all accelerations zero) and the mass-matrix rows the step needs anyway replace the two passes of the literal
    // formulation (inverse dynamics of the planned acceleration; bias with the servo joints' accelerations):
    //   tau_c  = clip(M_cc ddq_plan + h_c)                       acc_to_ctrl_action + the URDF effort limits
    //   M_cc ddq = tau_c - h_c - M_cs ddq_servo - D dq_c          forward dynamics of the six controlled joints
    T zero9[9], h9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) zero9[i] = T(0);
    T Ml[9][6], Mss[3];                             // rows 0..5: M_cc (lower triangle); rows 6..8: M_sc = M_cs^T
    T z6[3], z7[3], y7[3];                          // world axes behind the servo set-points
    if constexpr (LINK && ATACOM_DYN_LINK) {
        lk::Trig9<T> tg;
        lk::trig9(q9, tg);
        ATACOM_MARK("DYN_rnea"); phase();
        lk::rnea9<T, true>(tg, dq9, zero9, h9);
        ATACOM_MARK("DYN_crba"); phase();
        lk::crba<T, 6, 9>(tg, Ml, Mss);
        ATACOM_MARK("DYN_servo"); phase();
        lk::servo_axes(tg, z6, z7, y7);
    } else {
        Chain9<T> ch;
        iiwa_chain9(q9, ch);
        ATACOM_MARK("DYN_rnea"); phase();
        rnea9<T, true>(ch, dq9, zero9, h9);
        ATACOM_MARK("DYN_crba"); phase();
        crba<T, 6, 9>(ch, Ml, Mss);
        ATACOM_MARK("DYN_servo"); phase();
#pragma unroll
        for (int d = 0; d < 3; ++d) { z6[d] = ch.a[5][d]; z7[d] = ch.a[6][d]; y7[d] = ch.a[7][d]; }
    }
    const T tgt[3] = {joint7_target(z6, z7, st.qx[0]), universal_target(z7, y7), T(0)};
    constexpr T vmax[3] = {T(1.5 * 2.356194490192345), T(1.5 * 3.1415926), T(1.5 * 3.1415926)};     // urdf:297,384,397
    constexpr T effort_s[3] = {T(40), T(10), T(10)};                                                // urdf:297,384,400
    T dds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T vstar = num<T>::min(num<T>::max(T(0.1) * (tgt[i] - st.qx[i]) / P.dt, -vmax[i]), vmax[i]);
        const T lim = num<T>::div(num<T>::max(effort_s[i] - num<T>::abs(h9[6 + i]), T(0)), Mss[i]);
        dds[i] = num<T>::min(num<T>::max((vstar - st.dqx[i]) / P.dt, -lim), lim);
    }
    const T ff = (P.dynamics_mode == 2) ? T(1) : T(0);
    constexpr T effort[6] = {T(320), T(320), T(176), T(176), T(110), T(40)};
    T tau[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        T t = h9[i];
#pragma unroll
        for (int j = 0; j < 6; ++j) t = num<T>::fma((j <= i) ? Ml[i][j <= i ? j : 0] : Ml[j][i], ddq[j], t);
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) t = num<T>::fma(ff * Ml[6 + s2][i], dds[s2], t);
        tau[i] = num<T>::min(num<T>::max(t, -effort[i]), effort[i]);
    }
    T rhs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        T r = tau[i] - h9[i] - (T)iiwa_body::DAMPING[i] * st.dq[i];
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) r = num<T>::fma(-Ml[6 + s2][i], dds[s2], r);
        rhs[i] = r;
    }
    ATACOM_MARK("DYN_solve"); phase();
    chol_solve<T, 6, 9>(Ml, rhs);
    ATACOM_MARK("DYN_end"); phase();
#pragma unroll
    for (int i = 0; i < 6; ++i) ddq[i] = rhs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        st.dqx[i] = num<T>::fma(dds[i], P.dt, st.dqx[i]);
        st.qx[i] = num<T>::fma(st.dqx[i], P.dt, st.qx[i]);
    }
}

// ------------------------------------------------------------------ the step prologue DISTRIBUTED over a lane group (iiwa)
// Round 5.  With one environment per 8 lanes the prologue of a step -- kinematics, the three frame
// Jacobians, bias terms, assembly of K J, the hoisted first reflector G(0), the pick of "my columns" -- was computed by
// every lane of the group, bitwise identically: ~1.4 k of the ~10.4 k vector instructions a wave executes per step, all of
// them 8-fold redundant.  Here the lanes share it the way the solver wants the result anyway (atacom_quad.h: column 0 of
// K J replicated, column c >= 1 in lane c - 1):
//   * trigonometry: lane l evaluates sincos(q_l) ONCE (instead of six per lane) and the group exchanges the twelve values
//     by DPP broadcasts; the serial frame chain from them stays replicated (it does not split);
//   * lane c - 1 computes Jacobian COLUMN c of the three frames from "its" joint frame (z_c, o_c: a one-hot blend of the
//     replicated chain), i.e. its own column of K J for all twelve rows; column 0 (joint 1: the base's vertical axis) is cheap
//     and stays replicated;
//   * frame velocities J dq are group sums of the lanes' column terms (one DPP butterfly of six values) instead of 3 x 6
//     replicated multiply-adds per frame;
//   * G(0): row 0 of K J is gathered once (five broadcasts), the reflector generated replicated FROM THE SAME VALUES IN THE
//     SAME ORDER as the replicated form (identical beta, tau, v); the five dense rows below take one butterfly, the six
//     joint-limit rows (one entry each) need none; every lane then updates its own column.
// Same arithmetic as constraint_terms + the assembly + G0PRE of env_step's replicated prologue, except for the summation
// order of the frame velocities and of the reflector's row sums (float64 build: 1e-8 against the oracle like every
// mapping).  bias_mode 0 (the reference's w x v) only: the exact-bias option keeps the replicated prologue.
template <typename T, int LN>
__device__ __forceinline__ void group_sincos6(const T (&q)[6], const T (&oh)[LN], T (&sn)[6], T (&cs)[6]) {
#pragma clang fp contract(off)       // every multiply-add of the group prologue is written as an explicit fma: k_step, k_rollout and
                                     // k_rollout_mlp must contract identically (their results are compared bit for bit)
    static_assert(LN >= 6, "one joint per lane");
    T qm = oh[0] * q[0];                                  // exact: the mask is 0 / 1 (lanes 6, 7: angle 0)
#pragma unroll
    for (int i = 1; i < 6; ++i) qm = num<T>::fma(oh[i], q[i], qm);
    T sm, cm;
    num<T>::sincos(qm, &sm, &cm);
    static_for<0, 6>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        sn[i] = qbcast<i, LN>(sm);
        cs[i] = qbcast<i, LN>(cm);
    });
}
// linear Jacobian column of point p for the joint frame (z, o): z x (p - o)   (the arithmetic of jac_col, atacom_envs.h)
template <typename T>
__device__ __forceinline__ void cross_col(const T (&z)[3], const T (&o)[3], const T (&p)[3], T (&col)[3]) {
#pragma clang fp contract(off)
    const T rx = p[0] - o[0], ry = p[1] - o[1], rz = p[2] - o[2];
    col[0] = num<T>::fma(z[1], rz, -(z[2] * ry));
    col[1] = num<T>::fma(z[2], rx, -(z[0] * rz));
    col[2] = num<T>::fma(z[0], ry, -(z[1] * rx));
}
// group sums of six values (LN = 8)
template <typename T>
__device__ __forceinline__ void group8_sum6(T& a, T& b, T& c, T& d, T& e, T& f) { osum_n(a, b, c, d, e, f); }

// Outputs, in the layout the lane-group solver reads (env_step): A0[r] = column 0 of K J (replicated), Amy[r] = this lane's
// column lq + 1 (zeros in lanes >= 5: their slot-0 columns are slack columns), yb = psi + Kc c0, the mallet position;
// G0: rows as G(0) leaves them, row 0 of Amy = the reflector's entry, (g0_d, g0_tau) = its beta / tau.
template <typename T, int LN, bool G0>
__device__ __forceinline__ void iiwa_prepare_group(const Params<T>& P, const T (&qc)[6], const T (&dqc)[6], const int lq,
                                                   T (&A0)[12], T (&Amy)[12], T (&yb)[12], T& g0_d, T& g0_tau, T& mx, T& my) {
#pragma clang fp contract(off)       // see group_sincos6
    static_assert(LN == 8, "written for one environment per 8 lanes");
    constexpr int NQ = 6, NC = 12;
    T oh[LN];
#pragma unroll
    for (int l = 0; l < LN; ++l) oh[l] = (lq == l) ? T(1) : T(0);
    T sn[6], cs[6];
    group_sincos6<T, LN>(qc, oh, sn, cs);
    IiwaKin<T> k;
    iiwa_chain(sn, cs, k);
    // "my" joint c = lq + 1 (lanes 0..4): frame, velocity and limit term, one-hot over joints 1..5
    T zc[3], oc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        T vz = oh[0] * k.z[1][a], vo = oh[0] * k.o[1][a];
#pragma unroll
        for (int j = 2; j < NQ; ++j) { vz = num<T>::fma(oh[j - 1], k.z[j][a], vz); vo = num<T>::fma(oh[j - 1], k.o[j][a], vo); }
        zc[a] = vz; oc[a] = vo;
    }
    T dqm = oh[0] * dqc[1];
#pragma unroll
    for (int j = 2; j < NQ; ++j) dqm = num<T>::fma(oh[j - 1], dqc[j], dqm);
    T Je[3], J7[3], Je0[3], J70[3], J40[3], J41[3];
    cross_col(zc, oc, k.pe, Je);
    cross_col(zc, oc, k.p7, J7);
    jac_col(k, 0, k.pe, Je0);
    jac_col(k, 0, k.p7, J70);
    jac_col(k, 0, k.p4, J40);
    jac_col(k, 1, k.p4, J41);          // link_4 moves with joints 1, 2 only (constraint_terms); both columns replicated
    // frame velocities v = J dq: the lanes' column terms summed over the group, + column 0
    T ve[3], v7[3], v4[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { ve[a] = Je[a] * dqm; v7[a] = J7[a] * dqm; }
    group8_sum6(ve[0], ve[1], ve[2], v7[0], v7[1], v7[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ve[a] = num<T>::fma(Je0[a], dqc[0], ve[a]);
        v7[a] = num<T>::fma(J70[a], dqc[0], v7[a]);
        v4[a] = num<T>::fma(J41[a], dqc[1], J40[a] * dqc[0]);
    }
    // angular velocities (frame_bias: fma chain over the joints in order); w4 is the prefix of w6
    T w4[3] = {T(0), T(0), T(0)}, w6[3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) w4[a] = num<T>::fma(k.z[i][a], dqc[i], w4[a]);
#pragma unroll
    for (int a = 0; a < 3; ++a) w6[a] = num<T>::fma(k.z[5][a], dqc[5], num<T>::fma(k.z[4][a], dqc[4], w4[a]));
    // bias_mode 0: "classical acceleration" w x v (quirk Q2)
    T ae[3];
    ae[0] = num<T>::fma(w6[1], ve[2], -(w6[2] * ve[1]));
    ae[1] = num<T>::fma(w6[2], ve[0], -(w6[0] * ve[2]));
    ae[2] = num<T>::fma(w6[0], ve[1], -(w6[1] * ve[0]));
    const T a4z = num<T>::fma(w4[0], v4[1], -(w4[1] * v4[0]));
    const T a7z = num<T>::fma(w6[0], v7[1], -(w6[1] * v7[0]));
    T fun[NC], bst[NC], jdq[NC], mxy[2];
    iiwa_fun_from_kin(P, k, qc, fun, mxy);
    mx = -(fun[1] + P.table_bx);            // mallet (= tip) xy, as env_step recovers it from the table rows
    my = fun[3] + P.table_by;
    bst[0] = ae[2];                                                        // iiwa_hit_atacom.py:84-91
    bst[1] = -ae[0]; bst[2] = -ae[1]; bst[3] = ae[1]; bst[4] = -a4z; bst[5] = -a7z;          // :119-130
    jdq[0] = ve[2]; jdq[1] = -ve[0]; jdq[2] = -ve[1]; jdq[3] = ve[1]; jdq[4] = -v4[2]; jdq[5] = -v7[2];
    T d2q[NQ];                              // the joint-limit rows' one Jacobian entry, 2 q_i (:135-136)
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        d2q[i] = T(2) * qc[i];
        bst[6 + i] = T(2) * dqc[i] * dqc[i];                               // :138-139
        jdq[6 + i] = num<T>::fma(d2q[i], dqc[i], T(0));
    }
#pragma unroll
    for (int r = 0; r < NC; ++r) {
        const T psi = num<T>::fma(P.K[r], bst[r], jdq[r]);                 // constraints.py:42-43
        const T c0 = num<T>::fma(P.K[r], jdq[r], fun[r]);                  // constraints.py:33-37
        yb[r] = num<T>::fma(P.Kc[r], c0, psi);
    }
    // K J: column 0 replicated, my column (the "+ 0" of env_step's assembly: no -0 entries)
    A0[0] = num<T>::fma(P.K[0], Je0[2], T(0));
    A0[1] = num<T>::fma(P.K[1], -Je0[0], T(0));
    A0[2] = num<T>::fma(P.K[2], -Je0[1], T(0));
    A0[3] = num<T>::fma(P.K[3], Je0[1], T(0));
    A0[4] = num<T>::fma(P.K[4], -J40[2], T(0));
    A0[5] = num<T>::fma(P.K[5], -J70[2], T(0));
    A0[6] = num<T>::fma(P.K[6], d2q[0], T(0));
#pragma unroll
    for (int r = 7; r < NC; ++r) A0[r] = T(0);
    Amy[0] = num<T>::fma(P.K[0], Je[2], T(0));
    Amy[1] = num<T>::fma(P.K[1], -Je[0], T(0));
    Amy[2] = num<T>::fma(P.K[2], -Je[1], T(0));
    Amy[3] = num<T>::fma(P.K[3], Je[1], T(0));
    Amy[4] = (lq == 0) ? num<T>::fma(P.K[4], -J41[2], T(0)) : T(0);        // row 4: joints 3..6 do not move link_4
    Amy[5] = num<T>::fma(P.K[5], -J7[2], T(0));
    Amy[6] = T(0);
    T dg[NQ];                               // diagonal entries K (2 q_i) of the joint-limit rows, replicated
#pragma unroll
    for (int i = 1; i < NQ; ++i) {
        dg[i] = num<T>::fma(P.K[6 + i], d2q[i], T(0));
        Amy[6 + i] = (lq == i - 1) ? dg[i] : T(0);
    }
    if constexpr (G0) {
        // G(0): row 0 = [A0[0] | A[0][1..5]]; gathered so that every lane generates the reflector from the same values in the
        // order of the replicated prologue
        T r0[NQ], v[NQ];
        static_for<1, NQ>([&](auto cc) { constexpr int c = decltype(cc)::value; r0[c] = qbcast<c - 1, LN>(Amy[0]); });
        T ss = T(0);
#pragma unroll
        for (int c = 1; c < NQ; ++c) ss = num<T>::fma(r0[c], r0[c], ss);
        T beta;
        const T sc = larfg_scale(A0[0], ss, beta, g0_tau);
        g0_d = beta;
#pragma unroll
        for (int c = 1; c < NQ; ++c) v[c] = r0[c] * sc;
        const T vmy = (lq < NQ - 1) ? Amy[0] * sc : T(0);
        // dense rows 1..5: w_r = tau (A[r][0] + sum_c A[r][c] v_c) -- the column terms summed over the group
        T wp[6];
#pragma unroll
        for (int r = 1; r < 6; ++r) wp[r] = Amy[r] * vmy;
        wp[0] = T(0);
        group8_sum6(wp[0], wp[1], wp[2], wp[3], wp[4], wp[5]);
        T w[NC];
#pragma unroll
        for (int r = 1; r < 6; ++r) w[r] = (A0[r] + wp[r]) * g0_tau;
        w[6] = A0[6] * g0_tau;                                               // joint-limit rows: one entry each
#pragma unroll
        for (int i = 1; i < NQ; ++i) w[6 + i] = num<T>::fma(dg[i], v[i], T(0)) * g0_tau;
#pragma unroll
        for (int r = 1; r < NC; ++r) {
            A0[r] -= w[r];
            Amy[r] = num<T>::fma(-w[r], vmy, Amy[r]);
        }
        Amy[0] = vmy;
    }
}

// ------------------------------------------------------------------ one env step (A1, A2, A13-A15)
// LANES = 1: one environment per lane (atacom_linalg.h).  LANES = 4: one environment per DPP quad -- the
// null-space solve is column-split over the quad (atacom_quad.h), everything else is computed redundantly
// (and bitwise identically) by the four lanes; `lq` is the lane's index in its quad.
// CHART = 1: the opt-in canonical chart (atacom_chart.h) instead of the reference's LAPACK-basis + rref(tol) chart; with
// LANES > 1 its square-root recursion is distributed over the lanes of the group (one vector per lane with 8 lanes).
// THREADS: threads per workgroup of the calling kernel (sizes the LDS slice the rigid-body mode parks its solver state in)
// PARKDYN (rigid-body mode, lane groups): park the held solver state in LDS across the dynamics.  Since the dynamics run in link
// coordinates (round 5) the step and rollout kernels fit without it and are 6 % faster (42.7 against 45.6 us per step, quad,
// 8192 environments).  The policy kernel keeps it: built without, hipcc 7.2 places nine live-range copies (v_accvgpr_write)
// of wave-wide values at the top of the JOIN block of the lane-0-only store region of the step loop, IN FRONT OF the
// `s_or_b64 exec` that reopens the mask -- three of four lanes then reload stale registers and the in-kernel network sees a
// wrong bias from the second step on (isolated in round 6: profiles/r06_exec_mask_copies.md; found in round 5 by
// test_policy_rollout_in_rigid_body_mode).  A compiler defect that depends on register pressure, not on this source: every
// kernel of the built library is audited for the pattern (tests/test_kernel_resources.py, profiles/tools/exec_restore_audit.py).
template <typename T, typename E, int LANES, bool HOLD, bool DYN = false, bool HOIST_G0 = true, int CHART = 0,
          int THREADS = BLOCK<LANES>, bool PARKDYN = false, typename Ref>
__device__ __forceinline__ void env_step(const Params<T>& P, EnvState<T, E>& st, const T (&act)[E::NK],
                                         StepOut<T>& out, const int lq, const Ref& ref) {
    using L = Planes<E>;
    constexpr int NQ = E::NQ, NF = E::NF, NG = E::NG, NC = E::NC, NN = E::NN, NK = E::NK;
    T alpha[NK];
    T anorm2 = T(0);
#pragma unroll
    for (int k = 0; k < NK; ++k) {                              // atacom.py:107-108
        // ATACOM scales every null coordinate by max(acc_max); the E baseline scales joint k by acc_max[k]
        // (error_correction_wrapper.py:106-107); the T baseline hands the raw action to the base env
        const T sc = (E::MODE == 0) ? P.alpha_max : ((E::MODE == 1) ? P.acc_max[k < NQ ? k : 0] : T(1));
        alpha[k] = num<T>::min(num<T>::max(act[k], T(-1)), T(1)) * sc;
        anorm2 = num<T>::fma(alpha[k], alpha[k], anorm2);
    }
    if (E::ID == 0) {                                           // circle_base.py:54,86-107 (logged BEFORE the step)
        const T c1 = num<T>::abs(num<T>::fma(st.q[0], st.q[0], st.q[1] * st.q[1]) - T(1));
        const T c2 = -st.q[1] - T(0.5);
        out.log_avg = out.log_max = num<T>::max(c1, c2);
        out.log_dq = num<T>::max(num<T>::abs(st.dq[0]), num<T>::abs(st.dq[1])) - T(1);
    }
    if (E::MODE == 2) {
        // CircleEnvTerminated.step (circle_terminated.py:17-29): plain CircularMotion.step, then absorbing (reward -100)
        // if any PRE-step constraint value exceeds tol (self.c is set by check_constraint at the top of step)
        const bool term = num<T>::max(out.log_max, out.log_dq) > P.term_tol;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const T acc = alpha[i < NK ? i : 0] * T(10);                                   // circle_base.py:59-60
            st.q[i] += num<T>::fma(st.dq[i], P.dt_base, acc * (P.dt_base * P.dt_base) / T(2));
            st.dq[i] = num<T>::fma(acc, P.dt_base, st.dq[i]);
        }
        const T dxr = T(1) - st.q[0];
        const T rr = num<T>::exp(-num<T>::sqrt(num<T>::fma(dxr, dxr, st.q[1] * st.q[1])));
        out.reward = term ? T(-100) : rr;
        out.absorbing = term;
        st.t += 1;
        out.last = out.absorbing || (st.t >= P.horizon);
        return;
    }
    T qc[NQ], dqc[NQ];          // what the controller sees (held over the sub-steps when hold_q)
    T m0x = T(0), m0y = T(0);   // mallet position at the start of the env step
    T A[NC][NQ], yb[NC];       // yb = psi + Kc c0: the slack-independent part of the right-hand side (one value
                               // per row carried over the sub-steps instead of two)
    // hoist of the sub-step-invariant first reflector (see prepare): every mapping, ATACOM mode, held q / dq, and an
    // equality row on top of J_c (iiwa; the planar and circle J_c start with a slack-carrying row)
    constexpr bool CANON = CHART >= 1 && E::MODE == 0;        // CHART 2 (from k_step): canonical, slack stage A in static row order
    constexpr bool G0PRE = HOIST_G0 && HOLD && E::MODE == 0 && NF > 0 && NQ > 1 && !CANON;
    T arow[NG];                 // CANON: max |K J| of every inequality row (the scale its slack is compared with)
    T g0_d = T(0), g0_tau = T(0);
    constexpr int LGC = LANES > 1 ? LANES : 4;                  // lanes per environment of the group solver
    constexpr int SQ = split_slots(NN, LGC);
    T Aq[NC][SQ];              // LANES > 1: this lane's columns of [K J | 0] (column c >= 1 -> lane (c-1) % LANES,
                               // slot (c-1) / LANES; column 0 is replicated and read from A directly, atacom_quad.h)
    T tlo[NQ], tup[NQ];         // acc_truncation bounds (atacom.py:117-121): functions of the controller's dq only
    // CANON, LANES > 1 (third form, atacom_chart_group.h): the lane's own columns / rows of A, built with A
    constexpr bool CANON3 = CANON && LANES > 1 && (ATACOM_CHART_FORM == 3);
    [[maybe_unused]] ChartPre<T, E, LGC> cpre;
    // the prologue shared by the lanes of a group instead of replicated in each (iiwa_prepare_group above; round 5)
#ifndef ATACOM_GROUP_PRE
#define ATACOM_GROUP_PRE 1          // -DATACOM_GROUP_PRE=0: the A/B build with the replicated prologue
#endif
    constexpr bool GROUP_PRE = ATACOM_GROUP_PRE && E::ID == 2 && LANES == 8 && !CANON && E::MODE == 0 && !DYN;
    auto prepare = [&](int sub) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) { qc[i] = st.q[i]; dqc[i] = st.dq[i]; }
            if constexpr (E::PUCK && Ref::noise) {
                if (P.noise & NOISE_DELAY) {
                    // obs_delay: the wrapper's dq is read off the observation (atacom.py:95-96,111-112) -- the FILTERED joint
                    // velocities (as the last observation showed them; advanced per sub-step below)
#pragma unroll
                    for (int i = 0; i < NQ; ++i) dqc[i] = pl<E>(ref.f, L::FV + 3 + i, ref.B, ref.b);
                }
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                // lo <= up always (K_q, vel_max > 0 and both are clamped into [-acc_max, acc_max]).  max(min(a, x), -a) and
                // min(max(-a, x), a) are the same clamp of x into [-a, a] (a = acc_max > 0): one v_med3_f32 each instead
                // of a min / max pair with its canonicalising v_max x, x in front
                tup[i] = num<T>::clamp(-P.Kq[i] * (dqc[i] - P.vel_max[i]), -P.acc_max[i], P.acc_max[i]);
                tlo[i] = num<T>::clamp(-P.Kq[i] * (dqc[i] + P.vel_max[i]), -P.acc_max[i], P.acc_max[i]);
            }
            if constexpr (GROUP_PRE) {
                if (P.bias_mode == 0) {         // launch-uniform; the exact-bias option keeps the replicated prologue below
                    ATACOM_MARK("PRE_group");
                    T A0[NC], Amy[NC], mx, my;
                    iiwa_prepare_group<T, LANES, G0PRE>(P, qc, dqc, lq, A0, Amy, yb, g0_d, g0_tau, mx, my);
                    if (sub == 0) { m0x = mx; m0y = my; }
#pragma unroll
                    for (int r = 0; r < NC; ++r) {
                        A[r][0] = A0[r];
                        Aq[r][0] = Amy[r];
#pragma unroll
                        for (int sl = 1; sl < SQ; ++sl) Aq[r][sl] = T(0);       // columns 9.. are slack columns
                    }
                    return;
                }
            }
            T fun[NC], J[NC][NQ], bst[NC];
            ATACOM_MARK("PRE_terms");
            constraint_terms(E{}, P, qc, dqc, fun, J, bst);
            ATACOM_MARK("PRE_assemble");
            if (E::PUCK && sub == 0) {
                // mallet (= tip) xy at the start of the step, recovered from the table constraints
                // g1 = -x - bx, g3 = y - by  (rows NF, NF+2)
                m0x = -(fun[NF < NC ? NF : 0] + P.table_bx);
                m0y = fun[NF + 2 < NC ? NF + 2 : 0] + P.table_by;
            }
#pragma unroll
            for (int r = 0; r < NC; ++r) {
                T jdq = T(0);
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    if (E::jac_zero(r, i)) { A[r][i] = T(0); continue; }     // compile time (r, i are unrolled)
                    jdq = num<T>::fma(J[r][i], dqc[i], jdq);
                    // constraints.py:39-40; the "+ 0" inside the FMA turns a -0 product into +0 exactly as the
                    // reference's diag(K) @ J matmul does (the sign of a zero steers dlarfg's sign choice)
                    A[r][i] = num<T>::fma(P.K[r], J[r][i], T(0));
                }
                const T psi = num<T>::fma(P.K[r], bst[r], jdq);      // constraints.py:42-43
                const T c0 = num<T>::fma(P.K[r], jdq, fun[r]);       // constraints.py:33-37
                yb[r] = (E::MODE == 1) ? P.Kc[r] * c0 : num<T>::fma(P.Kc[r], c0, psi);   // E: no drift term (:127)
            }
            if constexpr (G0PRE) {
                // G(0), the first right reflector of the bidiagonalisation, only sees row 0 = [K_f J_f | 0] (an equality row
                // carries no slack) and only mixes the dim_q joint columns; q, dq -- hence K J -- are held over the
                // sub-steps (HOLD), so it is generated and applied ONCE per step here instead of once per sub-step in the
                // solver (which receives row 0 = the reflector vector and the updated rows below, PRE0).
                T ss = T(0);
#pragma unroll
                for (int c = 1; c < NQ; ++c) ss = num<T>::fma(A[0][c], A[0][c], ss);
                T beta;
                const T sc = larfg_scale(A[0][0], ss, beta, g0_tau);
                g0_d = beta;
#pragma unroll
                for (int c = 1; c < NQ; ++c) A[0][c] *= sc;
#pragma unroll
                for (int r = 1; r < NC; ++r) {
                    T w = A[r][0];
#pragma unroll
                    for (int c = 1; c < NQ; ++c)
                        if (!E::jac_zero(r, c)) w = num<T>::fma(A[r][c], A[0][c], w);    // structural zeros add nothing
                    w *= g0_tau;
                    A[r][0] -= w;
#pragma unroll
                    for (int c = 1; c < NQ; ++c) A[r][c] = num<T>::fma(-w, A[0][c], A[r][c]);
                }
            }
            if constexpr (CANON) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    T m = T(0);
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        if (!E::jac_zero(NF + g, i)) m = num<T>::max(m, num<T>::abs(A[NF + g][i]));
                    arow[g] = m;
                }
            }
            if constexpr (CANON3) chart_prepare<T, E, LGC>(A, arow, yb, P.Kc, lq, cpre);
            ATACOM_MARK("PRE_blend");
            if (LANES > 1 && !CANON) {
                // this lane's columns of K J: a one-hot blend over the lane group (exact: the mask is 0 / 1 and the
                // entries carry no -0 after the fma above).  Written as arithmetic on purpose -- a `lq == l ? ... : ...`
                // select chain over all rows is turned into a divergent switch by the optimiser (measured: 350
                // instructions for this block).
                T oh[LANES];
#pragma unroll
                for (int l = 0; l < LANES; ++l) oh[l] = (lq == l) ? T(1) : T(0);
#pragma unroll
                for (int r = 0; r < NC; ++r)
#pragma unroll
                    for (int sl = 0; sl < SQ; ++sl) {
                        T v = T(0);
#pragma unroll
                        for (int l = 0; l < LANES; ++l) {
                            const int c = LANES * sl + l + 1;
                            if (c < NQ) v = num<T>::fma(oh[l], A[r][c < NQ ? c : 0], v);
                        }
                        Aq[r][sl] = v;
                    }
            }
    };
    prepare(0);
#pragma unroll 1
    for (int sub = 0; sub < P.substeps; ++sub) {
        // HOLD (the reference's zero-order hold of q, dq -- quirk Q1) is a template parameter so that the ~1.4 k
        // instructions of the kinematics stay out of the sub-step loop in the default configuration (measured -3.5 %)
        if constexpr (E::PUCK && Ref::noise) {
            if (P.noise & NOISE_DELAY) {
                // step_action_function calls env._create_observation(sim_state) in every sub-step (atacom.py:124): the
                // low-pass advances on the velocities at the START of the sub-step (env_single.py:114-119).  Through the
                // state buffer, not a register (EnvRef)
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const T fo = pl<E>(ref.f, L::FV + 3 + i, ref.B, ref.b);
                    const T fn = num<T>::fma(T(0.5), st.dq[i], T(0.5) * fo);
                    if (ref.commit) pl<E>(ref.f, L::FV + 3 + i, ref.B, ref.b) = fn;
                }
            }
        }
        if constexpr (!HOLD || E::MODE == 1) {
            if (sub > 0) prepare(sub);
        }
        // J_c = [[K_f J_f, 0], [K_g J_g, diag(s)]],  rhs = psi + K_c c     (atacom.py:151-165,183-196)
        ATACOM_MARK("SUB_begin");
        T mu[NN], y[NC];
#pragma unroll
        for (int r = 0; r < NC; ++r) {
            const T sv = (r >= NF) ? st.s[r >= NF ? r - NF : 0] : T(0);
            // rhs = psi + Kc c,  c = fun + K J dq (+ s^2 / 2 on the g rows)
            y[r] = (r >= NF) ? num<T>::fma(T(0.5) * P.Kc[r] * sv, sv, yb[r]) : yb[r];
        }
        if constexpr (CANON) {
            if constexpr (LANES == 1) canonical_mu<T, E>(A, arow, st.s, y, alpha, P.rref_tol, mu ATACOM_DBG_ARG(out.dbg));
            else if constexpr (CANON3) canonical_mu_group3<T, E, LGC, CHART == 2>(A, cpre, arow, st.s, y, alpha, P.rref_tol, mu, lq ATACOM_DBG_ARG(out.dbg));
            else canonical_mu_group<T, E, LANES, CHART == 2>(A, arow, st.s, y, alpha, P.rref_tol, mu, lq ATACOM_DBG_ARG(out.dbg));
        } else if (LANES == 1 || E::MODE != 0) {
            T x[NN], nb[NN][NN - NC], nmu[NN];
            auto aget = [&](auto rc, auto cc) -> T {
                constexpr int r = decltype(rc)::value, c = decltype(cc)::value;
                if constexpr (c < NQ) return A[r][c];
                else if constexpr (r == NF + (c - NQ)) return st.s[c - NQ];
                else return T(0);
            };
            auto yget = [&](auto rc) -> T { return y[decltype(rc)::value]; };
            bidiag_solve_null<T, NC, NN, G0PRE, NQ>(aget, yget, x, nb, g0_d, g0_tau);        // atacom.py:127 (pinv_null)
            if (E::MODE == 1) {
                // error_correction_wrapper.py:127-130: [alpha; 0] - Jc^+ (Kc c), the null basis is not used
#pragma unroll
                for (int n = 0; n < NN; ++n) mu[n] = ((n < NQ) ? alpha[n < NK ? n : 0] : T(0)) - x[n];
            } else {
                T alpha0[NN - NC];
#pragma unroll
                for (int k = 0; k < NN - NC; ++k) alpha0[k] = alpha[k < NK ? k : 0];
                rref_apply<T, NN, NN - NC>(nb, alpha0, P.rref_tol, nmu);      // atacom.py:128,131
#pragma unroll
                for (int n = 0; n < NN; ++n) mu[n] = nmu[n] - x[n];           // atacom.py:130-133
            }
        } else {
            constexpr int LG = LGC;                          // lanes per environment (this branch: 2, 4 or 8)
            constexpr int S = SQ;
            constexpr int ND = NN - NC;
            T x0, x[S], nb0[ND], nb[S][ND], nmu0, nmu[S];
            T alphaq[ND];
#pragma unroll
            for (int k = 0; k < ND; ++k) alphaq[k] = alpha[k < NK ? k : 0];
            // this lane's columns: the K J block was split once per step (Aq), the slack diagonal entry of
            // row r sits in column c = NQ + r - NF, i.e. slot (c-1)/LG of lane (c-1)%LG
            auto aget = [&](auto rc, auto sc) -> T {
                constexpr int r = decltype(rc)::value, sl = decltype(sc)::value;
                const T base = Aq[r][sl];
                if constexpr (r >= NF) {
                    constexpr int c = NQ + r - NF;
                    if constexpr ((c - 1) / LG == sl) return (lq == (c - 1) % LG) ? st.s[r - NF] : base;
                }
                return base;
            };
            auto a0get = [&](auto rc) -> T { return A[decltype(rc)::value][0]; };      // NQ >= 1: never a slack column
            auto yget = [&](auto rc) -> T { return y[decltype(rc)::value]; };
            bidiag_solve_null_quad<T, NC, NN, LG, G0PRE>(aget, a0get, yget, x0, x, nb0, nb, lq, g0_d, g0_tau);
            rref_apply_quad<T, NN, ND, LG>(nb0, nb, alphaq, P.rref_tol, nmu0, nmu, lq);
            ATACOM_MARK("MU_gather");
            mu[0] = nmu0 - x0;
            static_for<1, NN>([&](auto nc) {                        // gather mu back to every lane of the group
                constexpr int n = decltype(nc)::value;
                mu[n] = qbcast<(n - 1) % LG, LG>(nmu[(n - 1) / LG] - x[(n - 1) / LG]);
            });
        }
        ATACOM_MARK("SUB_integrate");
#pragma unroll
        for (int g = 0; g < NG; ++g) st.s[g] = num<T>::fma(mu[NQ + g], P.dt, st.s[g]);   // :135
        T ddq[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) ddq[i] = num<T>::clamp(mu[i], tlo[i], tup[i]);   // acc_truncation, atacom.py:117-121
        if (E::ID == 0) {
            // circle_atacom.py:26-27 + circle_base.py:59-63
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const T acc = num<T>::min(num<T>::max(num<T>::div(ddq[i], P.acc_max[i]), T(-1)), T(1)) * T(10);
                // the BASE env's step (circle_base.py:62-63): dt_base, not the wrapper's dt (quirk Q4)
                st.q[i] += num<T>::fma(st.dq[i], P.dt_base, acc * (P.dt_base * P.dt_base) / T(2));
                st.dq[i] = num<T>::fma(acc, P.dt_base, st.dq[i]);
            }
        } else {
            if constexpr (DYN && E::ID == 2) {
                // row N4: ddq <- forward dynamics.  What the solver holds over the sub-steps -- the lane's columns of K J, its
                // replicated column 0, the slack-independent right-hand side, the truncation bounds: 84 values -- is PARKED
                // IN LDS across the rigid-body sub-step (float, reference chart): the nine-body chain, the Newton-Euler pass
                // and the mass-matrix rows need ~300 registers of their own, and next to the held state the kernels spilled
                // to scratch (32 - 200 bytes per lane through global memory, per sub-step).  21 ds_write_b128 + 21
                // ds_read_b128 per sub-step instead; every lane owns its slice, no barrier.
#ifndef ATACOM_DYN_PARK_LANE
#define ATACOM_DYN_PARK_LANE 0      // tuning: park in the one-environment-per-lane single-step kernel as well
#endif
                constexpr bool PARK = std::is_same<T, float>::value && !CANON && E::MODE == 0 && (ATACOM_DYN_PARK || PARKDYN) &&
                                      (LANES > 1 || (ATACOM_DYN_PARK_LANE && THREADS == 64));       // (one environment per lane: measured SLOWER with the parking, 73.9 ->
                                                       // 90.7 us per step at 8192 environments -- those kernels still spill and pay
                                                       // the LDS traffic on top; and the lane-mapped policy kernel's own 100 KB of
                                                       // LDS would leave no room)
                if constexpr (PARK) {
                    constexpr int NV = (LANES > 1 ? NC * SQ + NC : NC * NQ) + NC + 2 * NQ, NG4 = (NV + 3) / 4;
                    __shared__ float4 parked[NG4 * THREADS];
                    T pk[NG4 * 4];
                    int k = 0;
                    auto walk = [&](auto&& f) {                 // the same order on the way in and out
                        k = 0;
#pragma unroll
                        for (int r = 0; r < NC; ++r) {
                            if constexpr (LANES > 1) {
#pragma unroll
                                for (int sl = 0; sl < SQ; ++sl) f(Aq[r][sl]);
                                f(A[r][0]);
                            } else {
#pragma unroll
                                for (int c = 0; c < NQ; ++c) f(A[r][c]);
                            }
                        }
#pragma unroll
                        for (int r = 0; r < NC; ++r) f(yb[r]);
#pragma unroll
                        for (int i = 0; i < NQ; ++i) { f(tlo[i]); f(tup[i]); }
                    };
#pragma unroll
                    for (int i = 0; i < NG4 * 4; ++i) pk[i] = T(0);
                    walk([&](T& v) { pk[k++] = v; });
#pragma unroll
                    for (int g4 = 0; g4 < NG4; ++g4)
                        parked[g4 * THREADS + threadIdx.x] = make_float4(pk[4 * g4], pk[4 * g4 + 1], pk[4 * g4 + 2], pk[4 * g4 + 3]);
                    asm volatile("" ::: "memory");
                    rigid_body_substep<T, E>(P, st, ddq);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int g4 = 0; g4 < NG4; ++g4) {
                        const float4 v = parked[g4 * THREADS + threadIdx.x];
                        pk[4 * g4] = v.x; pk[4 * g4 + 1] = v.y; pk[4 * g4 + 2] = v.z; pk[4 * g4 + 3] = v.w;
                    }
                    walk([&](T& v) { v = pk[k++]; });
                } else {
                    rigid_body_substep<T, E>(P, st, ddq);
                }
            }
            // dynamics model of this build: ID o FD = identity (or the rigid-body mode above), semi-implicit
            // Euler, velocity clamp at 1.5 x limit (iiwa_hit_atacom.py:48-50)
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const T vlim = T(1.5) * P.vel_max[i];
                st.dq[i] = num<T>::clamp(num<T>::fma(ddq[i], P.dt, st.dq[i]), -vlim, vlim);
                st.q[i] = num<T>::fma(st.dq[i], P.dt, st.q[i]);
            }
        }
    }
    ATACOM_MARK("POST_fk");
    if (E::ID == 0) {
        const T dx = T(1) - st.q[0];
        out.reward = num<T>::exp(-num<T>::sqrt(num<T>::fma(dx, dx, st.q[1] * st.q[1])));   // circle_base.py:65
        out.absorbing = false;                                                              // :67
    } else {
        T fun[NC], mxy[2];
        if constexpr (GROUP_PRE) {              // post-step kinematics: one sincos per lane, shared by the group
            T oh[LANES], sn[NQ], cs[NQ];
#pragma unroll
            for (int l = 0; l < LANES; ++l) oh[l] = (lq == l) ? T(1) : T(0);
            group_sincos6<T, LANES>(st.q, oh, sn, cs);
            IiwaKin<T> kin;
            iiwa_chain(sn, cs, kin);
            iiwa_fun_from_kin(P, kin, st.q, fun, mxy);
        } else {
            constraint_fun(E{}, P, st.q, fun, mxy);
        }
        // ---- puck (row N1): the arm is kinematic w.r.t. the puck, so the puck's sub-steps run after the arm's,
        // against a mallet moving uniformly from (m0x, m0y) to mxy over the env step.  Frictionless disc,
        // impulse + push-out at the mallet, elastic rims, open goal mouths (env_hitting.py:44-45).
        ATACOM_MARK("POST_puck");
        {
            const T inv_n = num<T>::rcp((T)P.substeps);
            const T ux = (mxy[0] - m0x) * inv_n / P.dt, uy = (mxy[1] - m0y) * inv_n / P.dt;
            const T R = P.puck_r + P.mallet_r;
            const T ylim = P.table_hy - P.puck_r, xlim = P.table_hx - P.puck_r;
            // domain randomisation (NOISE kernels only)
            const bool delay = E::PUCK && Ref::noise && (P.noise & NOISE_DELAY) != 0;
            const bool kick = E::PUCK && Ref::noise && (P.noise & NOISE_ENV) != 0;
            T fvp[3] = {T(0), T(0), T(0)};
            int ep_run = 0;
            if constexpr (E::PUCK && Ref::noise) {
                if (delay) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) fvp[i] = pl<E>(ref.f, L::FV + i, ref.B, ref.b);
                }
                if (kick) ep_run = pli(ref.ip, L::I_EP, ref.b) - 1;
            }
#pragma unroll 1
            for (int k = 0; k < P.substeps; ++k) {
                const T fr = (T)(k + 1) * inv_n;
                const T mx = num<T>::fma(mxy[0] - m0x, fr, m0x), my = num<T>::fma(mxy[1] - m0y, fr, m0y);
                if constexpr (E::PUCK && Ref::noise) {
                    if (delay) {                 // the sub-step's _create_observation (env_single.py:114-116)
#pragma unroll
                        for (int i = 0; i < 3; ++i) fvp[i] = num<T>::fma(T(0.5), st.puck[3 + i], T(0.5) * fvp[i]);
                    }
                    if (kick) {                  // _simulation_pre_step, env_base.py:176-180: force 0.0005 [randn, randn, 0]
                        const int n_idx = 3 + 2 * P.substeps;
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            st.puck[3 + c] = num<T>::fma(P.env_noise_dv,
                                                         device_normal<T>(P.seed, ref.b, ep_run, st.t, 3 + 2 * k + c, n_idx),
                                                         st.puck[3 + c]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) st.puck[i] = num<T>::fma(st.puck[3 + i], P.dt, st.puck[i]);
                const T dxm = st.puck[0] - mx, dym = st.puck[1] - my;
                const T dist = num<T>::sqrt(num<T>::fma(dxm, dxm, dym * dym));
                const bool hit = dist < R;
                const bool pos = dist > T(0);
                const T idist = pos ? num<T>::rcp(dist) : T(0);
                const T nx = pos ? dxm * idist : T(1), ny = pos ? dym * idist : T(0);
                const T vrel = (st.puck[3] - ux) * nx + (st.puck[4] - uy) * ny;
                const bool imp = hit && (vrel < T(0));
                const T jimp = imp ? (T(1) + P.e_mallet) * vrel : T(0);
                st.puck[3] = num<T>::fma(-jimp, nx, st.puck[3]);
                st.puck[4] = num<T>::fma(-jimp, ny, st.puck[4]);
                st.puck[0] = hit ? num<T>::fma(nx, R, mx) : st.puck[0];
                st.puck[1] = hit ? num<T>::fma(ny, R, my) : st.puck[1];
                {   // side rims
                    const T ay = num<T>::abs(st.puck[1]);
                    const bool oy = ay > ylim;
                    const T sg = st.puck[1] > T(0) ? T(1) : (st.puck[1] < T(0) ? T(-1) : T(0));
                    st.puck[1] = oy ? sg * (T(2) * ylim - ay) : st.puck[1];
                    st.puck[4] = (oy && st.puck[4] * sg > T(0)) ? -P.e_rim * st.puck[4] : st.puck[4];
                }
                bool own_rim;   // the puck met the end rim on the agent's side (x < 0) in this sub-step
                {   // end rims, open in the goal mouth
                    const T ax = num<T>::abs(st.puck[0]);
                    const bool ox = (ax > xlim) && (num<T>::abs(st.puck[1]) >= P.goal_w);
                    const T sg = st.puck[0] > T(0) ? T(1) : (st.puck[0] < T(0) ? T(-1) : T(0));
                    own_rim = ox && (sg < T(0));
                    st.puck[0] = ox ? sg * (T(2) * xlim - ax) : st.puck[0];
                    st.puck[3] = (ox && st.puck[3] * sg > T(0)) ? -P.e_rim * st.puck[3] : st.puck[3];
                }
                if (E::ID == 1 && P.task == 1) {
                    // task 'D' (AirHockeyDefend._simulation_post_step [upstream]): has_hit (bit 0) latches on puck / mallet
                    // contact, has_bounce (bit 1) on contact with the end rims of the agent's side
                    st.has_hit |= (hit ? 1 : 0) | (own_rim ? 2 : 0);
                } else {
                    const T pv2k = num<T>::fma(st.puck[3], st.puck[3], st.puck[4] * st.puck[4]);
                    const bool new_hit = (st.has_hit == 0) && (pv2k > T(0.01));      // env_hitting.py:80-85
                    st.vel_hit_x = new_hit ? st.puck[3] : st.vel_hit_x;
                    st.has_hit = new_hit ? 1 : st.has_hit;
                }
            }
            if constexpr (E::PUCK && Ref::noise) {
                if (delay) {                     // the puck part of the observation the step returns (see the joints' below)
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const T fn = num<T>::fma(T(0.5), st.puck[3 + i], T(0.5) * fvp[i]);
                        if (ref.commit) pl<E>(ref.f, L::FV + i, ref.B, ref.b) = fn;
                    }
                }
            }
        }
        ATACOM_MARK("POST_reward");
        // absorbing: env_base.py:182-194 + env_hitting.py:71-78
        const T pv2 = num<T>::fma(st.puck[3], st.puck[3], st.puck[4] * st.puck[4]);
        bool ab = (num<T>::abs(st.puck[0]) > P.table_hx) || (num<T>::abs(st.puck[1]) > P.table_hy);
        ab = ab || (num<T>::abs(mxy[0]) - P.table_hx > T(0.02)) || (num<T>::abs(mxy[1]) - P.table_hy > T(0.02));
        T r;
        if (E::ID == 1 && P.task == 1) {
            // task 'D': AirHockeyDefend.is_absorbing / .reward [upstream, restated from memory; DESIGN.md section 4]
            ab = ab || ((st.has_hit != 0) && (st.puck[0] > T(0)));             // hit or bounced, and back in the other half
            const bool conceded = (st.puck[0] + P.table_hx < T(0)) && (num<T>::abs(st.puck[1]) - P.goal_w < T(0));
            const T apy = num<T>::abs(st.puck[1]);
            // after a hit: the puck resting near the line x = -0.6 on the agent's side
            const T r_y = T(3) * num<T>::exp(T(-3) * apy);
            const T r_x = num<T>::exp(T(-5) * num<T>::abs(st.puck[0] + T(0.6)));
            const T r_vel = T(5) * num<T>::exp(T(-25) * pv2);
            const bool zone = (st.puck[0] > T(-0.8)) && (st.puck[0] < T(-0.4));
            const T r_hit = zone ? ((r_x + r_y) + r_vel) + T(1) : T(0);
            // before: the mallet on the line x = -0.6 at the puck's y (a Gaussian bump at 0.08 offset)
            const T ex = num<T>::abs(T(-0.6) - mxy[0]), ey = num<T>::abs(st.puck[1] - mxy[1]);
            const T u = (ey - T(0.08)) * T(5);                                   // sigma = 0.2
            const T r_app = num<T>::fma(T(0.3), num<T>::exp(T(-3) * ex),
                                        T(0.7 * 0.5 * 1.9947114020071635) * num<T>::exp(T(-0.5) * u * u));
            r = ab ? (conceded ? T(-50) : T(0))
                   : (((st.has_hit & 2) != 0) ? T(-1) : (((st.has_hit & 1) != 0) ? r_hit : r_app));
        } else {
        ab = ab || ((st.has_hit != 0) && (pv2 < T(0.0001)));
        // reward: env_hitting.py:39-69
        const bool goal = (st.puck[0] - P.table_hx > T(0)) && (num<T>::abs(st.puck[1]) - P.goal_w < T(0));
        const T dx = st.puck[0] - mxy[0], dy = st.puck[1] - mxy[1];
        const T dist = num<T>::sqrt(num<T>::fma(dx, dx, dy * dy));
        const T gx = P.goal_x - st.puck[0], gy = P.goal_y - st.puck[1];
        const T gn = num<T>::sqrt(num<T>::fma(gx, gx, gy * gy));
        const T ign = num<T>::rcp(gn), idist = num<T>::rcp(dist);
        T cosang = num<T>::fma(gx * ign, dx * idist, (gy * ign) * (dy * idist));      // (explicit contraction, as below)
        cosang = num<T>::min(num<T>::max(cosang, T(0)), T(1));
        const T r_app = num<T>::exp(T(-8) * (dist - T(0.08))) * cosang;
        const bool upd = !ab && (st.has_hit == 0);
        st.r_hit = upd ? r_app : st.r_hit;
        r = ab ? (goal ? T(80) : T(0)) : ((st.has_hit != 0) ? num<T>::fma(st.vel_hit_x, T(0.1), T(1) + st.r_hit) : r_app);
        }
        out.reward = num<T>::fma(-P.action_penalty, num<T>::sqrt(anorm2), r);     // explicit: the same contraction in every kernel
        out.absorbing = ab;
        // constraint statistics, atacom.py:201-205
        T cm = num<T>::abs(fun[0]);
        if (NF == 0) cm = fun[0];
#pragma unroll
        for (int r2 = 1; r2 < NC; ++r2) cm = num<T>::max(cm, (r2 < NF) ? num<T>::abs(fun[r2]) : fun[r2]);
        T dm = num<T>::abs(st.dq[0]) - P.vel_max[0];
#pragma unroll
        for (int i = 1; i < NQ; ++i) dm = num<T>::max(dm, num<T>::abs(st.dq[i]) - P.vel_max[i]);
        if constexpr (E::PUCK && Ref::noise) {
            if (P.noise & NOISE_DELAY) {
                // the observation the step returns advances the low-pass once more (env_single.py:114-119; the puck's part
                // was stored right after its sub-steps), and _update_constraint_stats gets the wrapper's dq = the observation's (atacom.py:111-114)
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const T fo = pl<E>(ref.f, L::FV + 3 + i, ref.B, ref.b);
                    const T fn = num<T>::fma(T(0.5), st.dq[i], T(0.5) * fo);
                    if (ref.commit) pl<E>(ref.f, L::FV + 3 + i, ref.B, ref.b) = fn;
                    const T d = num<T>::abs(fn) - P.vel_max[i];
                    dm = (i == 0) ? d : num<T>::max(dm, d);
                }
            }
        }
        out.log_avg = out.log_max = cm;
        out.log_dq = dm;
    }
    st.t += 1;
    out.last = out.absorbing || (st.t >= P.horizon);
}

// ------------------------------------------------------------------ kernels
// mask (nullable): environments whose byte is 0 sit the call out -- state, step counter and statistics untouched; they
// report their current observation, reward 0, absorbing 0, last 0 (a vectorised Core's finished environments).
template <typename T, typename E, int LANES, bool HOLD, bool DYN = false, int CHART = 0, bool NOISE = false>
__global__ void __launch_bounds__(BLOCK<LANES>) k_step(const Params<T> P, T* __restrict__ f, int* __restrict__ ip,
                                               const T* __restrict__ action, T* __restrict__ obs,
                                               T* __restrict__ reward, uint8_t* __restrict__ absorbing,
                                               uint8_t* __restrict__ last, const uint8_t* __restrict__ mask) {
    // (kernarg preloading -- pointers and batch size first, -amdgpu-kernarg-preload-count=16 -- was tried: the waves no
    // longer wait for a scalar load before their first state loads, but the step time did not move (27.5 us both ways)
    // and the launch-bound circle kernel got slower, 7.4 -> 10.7 us: profiles/r02_lanes_vs_batch.md)
    using L = Planes<E>;
    const int B = P.batch;
    const int gt = blockIdx.x * BLOCK<LANES> + threadIdx.x;
    const int b = gt / LANES;                  // whole quads leave together (BLOCK % LANES == 0)
    const int lq = gt % LANES;
    if (b >= B) return;
    const EnvRef<T, NOISE> ref{f, ip, B, b, true};
#ifdef ATACOM_TIMESTAMPS        // tuning build only (profiles/tools/gpu_phase_probe.py): 100 MHz wall-clock stamps per phase
    const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
#endif
    EnvState<T, E> st;
    load_state<T, E>(f, ip, B, b, st);
    if (mask && mask[b] == 0) {                // whole lane groups leave together (the mask is per environment)
        if (lq == 0) {
            write_obs<T, E>(P, st, obs + (size_t)b * E::OBS, ref);
            reward[b] = T(0);
            absorbing[b] = 0;
            if (last) last[b] = 0;
        }
        return;
    }
    // the statistics accumulators are read up front with the rest of the state: a read-modify-write at the end
    // would make the store tail wait on loads queued behind ~60 stores (vmcnt counts both on gfx9-class hardware)
    const T ssum0 = pl<E>(f, L::SSUM, B, b), scmax0 = pl<E>(f, L::SCMAX, B, b);
    const T sdq0 = pl<E>(f, L::SDQMAX, B, b);
    const int cnt0 = pli(ip, L::I_CNT, b);
    T act[E::NK];
#pragma unroll
    for (int k = 0; k < E::NK; ++k) act[k] = action[(size_t)b * E::NK + k];
    if constexpr (DYN) load_aux<T, E>(f, B, b, st);
    StepOut<T> out;
#ifdef ATACOM_TIMESTAMPS
    // the stamp must not float above the loads' completion: make it depend on a loaded value
    unsigned long long ts1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts1) : "v"(st.q[0]) : "memory");
#endif
    // single-step launches last as long as their slowest wavefront: the canonical chart's slack stage A runs in static row
    // order there (atacom_chart.h); the T-step kernels, which average over their steps, keep the per-lane scan
    env_step<T, E, LANES, HOLD, DYN, true, (CHART == 1 ? 2 : CHART)>(P, st, act, out, lq, ref);
    ATACOM_MARK("STORE");
#ifdef ATACOM_TIMESTAMPS
    unsigned long long ts2;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts2) : "v"(out.reward) : "memory");
#endif
    if (lq != 0) return;                       // the four lanes hold identical results; lane 0 writes
    write_obs<T, E>(P, st, obs + (size_t)b * E::OBS, ref);
    reward[b] = out.reward;
    absorbing[b] = out.absorbing ? 1 : 0;
    if (last) last[b] = out.last ? 1 : 0;
    pl<E>(f, L::SSUM, B, b) = ssum0 + out.log_avg;
    pl<E>(f, L::SCMAX, B, b) = num<T>::max(scmax0, out.log_max);
    pl<E>(f, L::SDQMAX, B, b) = num<T>::max(sdq0, out.log_dq);
    pli(ip, L::I_CNT, b) = cnt0 + 1;
    if (P.auto_reset && out.last) reset_env<T, E>(P, ref, st);
    store_state<T, E>(f, ip, B, b, st);
    if constexpr (DYN) store_aux<T, E>(f, B, b, st);
#ifdef ATACOM_TIMESTAMPS
    {   // overwrite the first observation entries with the stamps (low 32 bits, 10 ns ticks); ts3 after the stores landed
        unsigned long long ts3;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts3) :: "memory");
        unsigned int* o = reinterpret_cast<unsigned int*>(obs + (size_t)b * E::OBS);
        o[0] = (unsigned int)ts0; o[1] = (unsigned int)ts1; o[2] = (unsigned int)ts2; o[3] = (unsigned int)ts3;
        if constexpr (CHART == 1) { o[4] = out.dbg[0]; o[5] = out.dbg[1]; o[6] = out.dbg[2]; }
    }
#endif
}

template <typename T, typename E, int LANES, bool HOLD, bool DYN = false, int CHART = 0, bool NOISE = false>
__global__ void __launch_bounds__(BLOCK<LANES>) k_rollout(const Params<T> P, int n_steps, T* __restrict__ f,
                                                  int* __restrict__ ip, const T* __restrict__ actions,
                                                  T* __restrict__ obs, T* __restrict__ next_obs,
                                                  T* __restrict__ reward, uint8_t* __restrict__ absorbing,
                                                  uint8_t* __restrict__ last, T* __restrict__ rec, int rec_ld) {
    using L = Planes<E>;
    using R = Record<E>;
    const int B = P.batch;
    const int gt = blockIdx.x * BLOCK<LANES> + threadIdx.x;
    const int b = gt / LANES;
    const int lq = gt % LANES;
    if (b >= B) return;
    const EnvRef<T, NOISE> ref{f, ip, B, b, true};
    EnvState<T, E> st;
    load_state<T, E>(f, ip, B, b, st);
    if constexpr (DYN) load_aux<T, E>(f, B, b, st);
    T ssum = T(0), scmax = pl<E>(f, L::SCMAX, B, b), sdq = pl<E>(f, L::SDQMAX, B, b);
    // the actions of step t + 1 are fetched while step t computes: a load at the top of the step it feeds would expose
    // one HBM round trip (~1 us) per step to a wave that has nothing else to run
    T act_next[E::NK];
#pragma unroll
    for (int k = 0; k < E::NK; ++k) act_next[k] = (n_steps > 0) ? actions[(size_t)b * E::NK + k] : T(0);
#pragma unroll 1
    for (int t = 0; t < n_steps; ++t) {
        const size_t row = (size_t)t * B + b;
        T* const rrow = rec ? rec + ((size_t)t * rec_ld + b) * R::F : nullptr;     // packed record of (t, b)
        T act[E::NK];
#pragma unroll
        for (int k = 0; k < E::NK; ++k) act[k] = act_next[k];
        {
            const size_t nrow = (size_t)((t + 1 < n_steps) ? t + 1 : t) * B + b;    // last step: a harmless re-read
#pragma unroll
            for (int k = 0; k < E::NK; ++k) act_next[k] = actions[nrow * E::NK + k];
        }
        if (lq == 0) {
            if (rec) {
                write_obs<T, E>(P, st, rrow + R::OBS, ref);
#pragma unroll
                for (int k = 0; k < E::NK; ++k) rrow[R::ACT + k] = act[k];
            } else {
                write_obs<T, E>(P, st, obs + row * E::OBS, ref);
            }
        }
        StepOut<T> out;
        env_step<T, E, LANES, HOLD, DYN, true, CHART>(P, st, act, out, lq, ref);
        if (lq == 0) {
            if (rec) {
                write_obs<T, E>(P, st, rrow + R::NOBS, ref);
                rrow[R::REW] = out.reward;
                rrow[R::ABS] = out.absorbing ? T(1) : T(0);
                rrow[R::LAST] = out.last ? T(1) : T(0);
            } else {
                if (next_obs) write_obs<T, E>(P, st, next_obs + row * E::OBS, ref);
                reward[row] = out.reward;
                absorbing[row] = out.absorbing ? 1 : 0;
                last[row] = out.last ? 1 : 0;
            }
        }
        ssum += out.log_avg;
        scmax = num<T>::max(scmax, out.log_max);
        sdq = num<T>::max(sdq, out.log_dq);
        if (P.auto_reset && out.last) reset_env<T, E>(P, ref, st);
    }
    if (lq != 0) return;
    pl<E>(f, L::SSUM, B, b) += ssum;
    pl<E>(f, L::SCMAX, B, b) = scmax;
    pl<E>(f, L::SDQMAX, B, b) = sdq;
    pli(ip, L::I_CNT, b) += n_steps;
    store_state<T, E>(f, ip, B, b, st);
    if constexpr (DYN) store_aux<T, E>(f, B, b, st);
}

// Row N2: rollout with the policy MLP evaluated in the kernel (atacom_policy.h).  d_actions_out receives the action
// the policy drew (mean + std * noise, before the env's clip to [-1, 1]).
// float: the network runs on the matrix cores (mlp_forward_mfma), one wavefront = 1 (quad mapping) or 4 (lane mapping)
// GEMM column blocks of 16 environments.  Every lane of a live wave must then stay in the kernel (it supplies operand
// slices for ALL environments of its blocks), so lanes past the end of the batch shadow the last environment with
// their stores masked off.
template <typename T, typename E, int LANES, int H>
struct MlpPath {
    // 8 lanes per environment: a wave holds 8 environments -- half of the one GEMM block it can fill is padding.  The VALU form
    // spread over the 8 lanes (8 hidden units per lane, -DATACOM_MLP8_VALU=1) was built and measured against it in round 6:
    // SLOWER -- iiwa 23.7 against 23.2 us per step, planar 11.7 against 10.9, at 4096 and 8192 environments, three interleaved
    // runs (profiles/r06_ab_mlp8_valu.log): 560 vector issue slots cost a lone wave more than 100 matrix instructions and two
    // LDS round trips.  The matrix cores pay even half empty.
#ifndef ATACOM_MLP8_VALU
#define ATACOM_MLP8_VALU 0
#endif
    static constexpr bool MFMA = std::is_same<T, float>::value && H == 64 && E::OBS <= 32 && E::NK <= 8 &&
                                 !(LANES == 8 && ATACOM_MLP8_VALU);
    // blocks of 16 environments per wavefront; with 8 lanes per environment a wave holds 8 environments: ONE block whose
    // columns 8..15 are padding (zero observations in, outputs never read -- GEMM columns do not mix)
    static constexpr int NB = (16 * LANES >= WAVE) ? 1 : WAVE / (16 * LANES);
    using LM = MlpLdsM<(E::OBS <= 32 ? E::OBS : 32), 64, (E::NK <= 8 ? E::NK : 8)>;
    // 256-thread workgroups in both mappings: the staged weights (30 KB per network) are shared by 4 wavefronts; with
    // one wave per workgroup the LDS footprint capped the lane mapping at 2 waves per CU (measured: 2x the time)
    static constexpr int THREADS = 256;
    static constexpr int STAGE = (THREADS / WAVE) * LM::wave_stage(NB);   // floats of per-wave staging per workgroup
};

template <typename T, typename E, int LANES, bool HOLD, int H, bool DYN = false, int CHART = 0, bool NOISE = false>
__global__ void __launch_bounds__(256) k_rollout_mlp(const Params<T> P, const MlpArgs<T> net, int n_steps,
                                                      T* __restrict__ f, int* __restrict__ ip,
                                                      const T* __restrict__ noise, T* __restrict__ obs,
                                                      T* __restrict__ next_obs, T* __restrict__ actions_out,
                                                      T* __restrict__ reward, uint8_t* __restrict__ absorbing,
                                                      uint8_t* __restrict__ last, T* __restrict__ rec, int rec_ld) {
    using L = Planes<E>;
    using R = Record<E>;
    constexpr bool MFMA = MlpPath<T, E, LANES, H>::MFMA;
    static_assert(LANES <= 4 || MFMA || (ATACOM_MLP8_VALU && std::is_same<T, float>::value), "8 lanes per environment: float32 only");
    constexpr int THREADS = MlpPath<T, E, LANES, H>::THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lds = reinterpret_cast<T*>(smem);
    if constexpr (MFMA) mlp_stage_mfma<E::OBS, H, E::NK>(net, lds, threadIdx.x, THREADS, MlpPath<T, E, LANES, H>::STAGE);
    else mlp_stage<T, E::OBS, H, E::NK>(net, lds, threadIdx.x, THREADS);
    const int B = P.batch;
    const int gt = blockIdx.x * THREADS + threadIdx.x;
    const bool valid = gt / LANES < B;
    if constexpr (MFMA) {
        if ((gt & ~(WAVE - 1)) / LANES >= B) return;          // whole wavefront past the batch
    } else {
        if (!valid) return;
    }
    const int b = valid ? gt / LANES : B - 1;
    const int lq = gt % LANES;
    const int lane = threadIdx.x & (WAVE - 1);
    using LM = typename MlpPath<T, E, LANES, H>::LM;
    constexpr int NB = MlpPath<T, E, LANES, H>::NB;
    T* stage = lds + 2 * LM::NET + (threadIdx.x / WAVE) * LM::wave_stage(NB);  // MFMA path only
    const int erow = lane / LANES;                                             // own environment within the wavefront
    const EnvRef<T, NOISE> ref{f, ip, B, b, valid};
    EnvState<T, E> st;
    load_state<T, E>(f, ip, B, b, st);
    if constexpr (DYN) load_aux<T, E>(f, B, b, st);
    T ssum = T(0), scmax = pl<E>(f, L::SCMAX, B, b), sdq = pl<E>(f, L::SDQMAX, B, b);
#pragma unroll 1
    for (int t = 0; t < n_steps; ++t) {
        const size_t row = (size_t)t * B + b;
        // the step's exploration noise is requested before the network runs (the wave fences of the LDS staging would
        // otherwise pin the loads behind it, one exposed HBM round trip per step)
        T eps[E::NK];
#pragma unroll
        for (int k = 0; k < E::NK; ++k) eps[k] = noise ? noise[row * E::NK + k] : T(0);
        T o[E::OBS];
        write_obs<T, E>(P, st, o, ref);
        T act[E::NK], sig[E::NK];
        if constexpr (MFMA) {
            float xin[NB][LM::CH];
            mlp_obs_to_operand<E::OBS, H, E::NK, NB>(lds, stage, o, lane, erow, xin);
            mlp_forward_mfma<E::OBS, H, E::NK, NB>(lds, stage, xin, net.activation, lane, erow, act);
            if (net.sW1)
                mlp_forward_mfma<E::OBS, H, E::NK, NB>(lds + LM::NET, stage, xin, net.activation, lane, erow, sig);
        } else {
            mlp_forward<T, E::OBS, H, E::NK, LANES>(lds, lds, o, net.activation, lq, act);
            if (net.sW1)
                mlp_forward<T, E::OBS, H, E::NK, LANES>(lds + MlpLds<E::OBS, H, E::NK>::TOTAL, lds, o, net.activation, lq, sig);
        }
        if (net.sW1) {
            // state-dependent sigma = exp(clamp(log_sigma_net(obs)))   (SAC)
#pragma unroll
            for (int k = 0; k < E::NK; ++k)
                sig[k] = num<T>::exp(num<T>::min(num<T>::max(sig[k], net.log_std_min), net.log_std_max));
        } else {
#pragma unroll
            for (int k = 0; k < E::NK; ++k) sig[k] = lds[(MFMA ? (int)LM::STD : (int)MlpLds<E::OBS, H, E::NK>::STD) + k];
        }
#pragma unroll
        for (int k = 0; k < E::NK; ++k) {
            act[k] = num<T>::fma(sig[k], eps[k], act[k]);
            if (net.squash) act[k] = num<T>::tanh(act[k]);
        }
        T* const rrow = rec ? rec + ((size_t)t * rec_ld + b) * R::F : nullptr;
        if (lq == 0 && valid) {
            T* const od = rec ? rrow + R::OBS : obs + row * E::OBS;
            T* const ad = rec ? rrow + R::ACT : actions_out + row * E::NK;
#pragma unroll
            for (int i = 0; i < E::OBS; ++i) od[i] = o[i];
#pragma unroll
            for (int k = 0; k < E::NK; ++k) ad[k] = act[k];
        }
        StepOut<T> out;
        // (one environment per lane with the network's four GEMM blocks live is the one kernel at the edge of the register
        // file: with the G(0) hoist its spills move INTO the sub-step loop -- 59.6 -> 79.5 us per step, measured -- so it
        // keeps the un-hoisted solver)
#ifndef ATACOM_MLP_PARK
#define ATACOM_MLP_PARK 1           // -DATACOM_MLP_PARK=0: the build that shows the defect described at env_step (PARKDYN)
#endif
        env_step<T, E, LANES, HOLD, DYN, (LANES > 1), CHART, THREADS, /*PARKDYN*/ DYN && ATACOM_MLP_PARK>(P, st, act, out, lq, ref);
        if (lq == 0 && valid) {
            if (rec) {
                write_obs<T, E>(P, st, rrow + R::NOBS, ref);
                rrow[R::REW] = out.reward;
                rrow[R::ABS] = out.absorbing ? T(1) : T(0);
                rrow[R::LAST] = out.last ? T(1) : T(0);
            } else {
                if (next_obs) write_obs<T, E>(P, st, next_obs + row * E::OBS, ref);
                reward[row] = out.reward;
                absorbing[row] = out.absorbing ? 1 : 0;
                last[row] = out.last ? 1 : 0;
            }
        }
        ssum += out.log_avg;
        scmax = num<T>::max(scmax, out.log_max);
        sdq = num<T>::max(sdq, out.log_dq);
        if (P.auto_reset && out.last) reset_env<T, E>(P, ref, st);
    }
    if (lq != 0 || !valid) return;
    pl<E>(f, L::SSUM, B, b) += ssum;
    pl<E>(f, L::SCMAX, B, b) = scmax;
    pl<E>(f, L::SDQMAX, B, b) = sdq;
    pli(ip, L::I_CNT, b) += n_steps;
    store_state<T, E>(f, ip, B, b, st);
    if constexpr (DYN) store_aux<T, E>(f, B, b, st);
}

template <typename T, typename E>
__global__ void __launch_bounds__(WAVE) k_reset(const Params<T> P, T* __restrict__ f, int* __restrict__ ip,
                                                const uint8_t* __restrict__ mask, const T* __restrict__ init,
                                                T* __restrict__ obs) {
    using L = Planes<E>;
    const int B = P.batch;
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if (b >= B) return;
    const bool m = mask ? (mask[b] != 0) : true;
    const EnvRef<T, E::PUCK> ref{f, ip, B, b, true};      // the reset kernel carries the options as run-time branches
    EnvState<T, E> st;
    if (m) {
        if (init) {
            const T* row = init + (size_t)b * L::INIT_DIM;
#pragma unroll
            for (int i = 0; i < E::NQ; ++i) {
                pl<E>(f, L::IQ + i, B, b) = row[i];
                pl<E>(f, L::IDQ + i, B, b) = row[E::NQ + i];
            }
            if (E::PUCK) {
#pragma unroll
                for (int i = 0; i < 6; ++i) pl<E>(f, L::IPUCK + i, B, b) = row[2 * E::NQ + i];
            }
        }
        load_init<T, E>(f, B, b, st);
        slack_init<T, E>(P, st);                 // slack of the STORED initial state (reused by every auto-reset)
#pragma unroll
        for (int g = 0; g < E::NG; ++g) pl<E>(f, L::IS + g, B, b) = st.s[g];
        // an explicit state wins over the draw; every reset starts a new episode of the noise streams
        if ((P.random_init && !init) || (E::PUCK && P.noise != 0)) reset_env<T, E>(P, ref, st, !init);
        store_state<T, E>(f, ip, B, b, st);
        if constexpr (E::ID == 2) store_aux<T, E>(f, B, b, st);              // servo joints back to rest
    } else {
        load_state<T, E>(f, ip, B, b, st);
    }
    if (obs) write_obs<T, E>(P, st, obs + (size_t)b * E::OBS, ref);
}

// set the stored initial state of every env to one row (used by create), clear statistics
template <typename T, typename E>
__global__ void k_fill_init(int B, T* __restrict__ f, int* __restrict__ ip, const T* __restrict__ row) {
    using L = Planes<E>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) { pl<E>(f, L::IQ + i, B, b) = row[i]; pl<E>(f, L::IDQ + i, B, b) = row[E::NQ + i]; }
    if (E::PUCK) {
#pragma unroll
        for (int i = 0; i < 6; ++i) pl<E>(f, L::IPUCK + i, B, b) = row[2 * E::NQ + i];
    }
}

template <typename T, typename E>
__global__ void k_clear_stats(int B, T* __restrict__ f, int* __restrict__ ip) {
    using L = Planes<E>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    pl<E>(f, L::SSUM, B, b) = T(0);
    pl<E>(f, L::SCMAX, B, b) = -INFINITY;
    pl<E>(f, L::SDQMAX, B, b) = -INFINITY;
    pli(ip, L::I_CNT, b) = 0;
}

// per-block partial reduction of the statistics: partial[block] = {sum, count, cmax, dqmax} (doubles)
template <typename T, typename E>
__global__ void __launch_bounds__(256) k_stats(int B, const T* __restrict__ f, const int* __restrict__ ip,
                                               double* __restrict__ partial) {
    using L = Planes<E>;
    __shared__ double sh[4][256];
    double s = 0.0, c = 0.0, m1 = -INFINITY, m2 = -INFINITY;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        s += (double)pl<E>(f, L::SSUM, B, b);
        c += (double)pli(ip, L::I_CNT, b);
        m1 = fmax(m1, (double)pl<E>(f, L::SCMAX, B, b));
        m2 = fmax(m2, (double)pl<E>(f, L::SDQMAX, B, b));
    }
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = c; sh[2][threadIdx.x] = m1; sh[3][threadIdx.x] = m2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + w];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + w];
            sh[2][threadIdx.x] = fmax(sh[2][threadIdx.x], sh[2][threadIdx.x + w]);
            sh[3][threadIdx.x] = fmax(sh[3][threadIdx.x], sh[3][threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 4 + 0] = sh[0][0];
        partial[blockIdx.x * 4 + 1] = sh[1][0];
        partial[blockIdx.x * 4 + 2] = sh[2][0];
        partial[blockIdx.x * 4 + 3] = sh[3][0];
    }
}

// state <-> user-facing array-of-structures [B, STATE_DIM] = [q, dq, s, puck(6), has_hit, r_hit, vel_hit_x, t]
template <typename T, typename E>
__global__ void k_get_state(int B, const T* __restrict__ f, const int* __restrict__ ip, T* __restrict__ out) {
    using L = Planes<E>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    EnvState<T, E> st;
    load_state<T, E>(f, ip, B, b, st);
    if (E::PUCK == false) {
#pragma unroll
        for (int i = 0; i < 6; ++i) st.puck[i] = T(0);
    }
    T* o = out + (size_t)b * L::STATE_DIM;
    int k = 0;
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) o[k++] = st.q[i];
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) o[k++] = st.dq[i];
#pragma unroll
    for (int i = 0; i < E::NG; ++i) o[k++] = st.s[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) o[k++] = st.puck[i];
    o[k++] = (T)st.has_hit; o[k++] = st.r_hit; o[k++] = st.vel_hit_x; o[k++] = (T)st.t;
}

template <typename T, typename E>
__global__ void k_set_state(int B, T* __restrict__ f, int* __restrict__ ip, const T* __restrict__ in) {
    using L = Planes<E>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    EnvState<T, E> st;
    const T* o = in + (size_t)b * L::STATE_DIM;
    int k = 0;
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) st.q[i] = o[k++];
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) st.dq[i] = o[k++];
#pragma unroll
    for (int i = 0; i < E::NG; ++i) st.s[i] = o[k++];
#pragma unroll
    for (int i = 0; i < 6; ++i) st.puck[i] = o[k++];
    st.has_hit = ((int)o[k++]) & 3;                 // bit 0 has_hit, bit 1 has_bounce (task 'D')
    st.r_hit = o[k++]; st.vel_hit_x = o[k++]; st.t = (int)o[k++];
    store_state<T, E>(f, ip, B, b, st);
    if constexpr (E::PUCK) {
        // obs_delay: the low-pass behind the observation's velocities restarts on the injected state, like a reset does
        // (reset_env) -- a stale filter of the previous trajectory would have the controller's dq disagree with the state just
        // set.  A caller that wants a particular filter state calls atacom_set_filter_state AFTER this.
#pragma unroll
        for (int i = 0; i < 3; ++i) pl<E>(f, L::FV + i, B, b) = st.puck[3 + i];
#pragma unroll
        for (int i = 0; i < E::NQ; ++i) pl<E>(f, L::FV + 3 + i, B, b) = st.dq[i];
    }
}

// obs_delay: the low-pass state of the observation's velocities <-> [B, 3 + NQ] = [puck vx, vy, yaw rate, dq]
template <typename T, typename E>
__global__ void k_filter_io(int B, T* __restrict__ f, T* __restrict__ buf, int set) {
    using L = Planes<E>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if constexpr (E::PUCK) {
#pragma unroll
        for (int i = 0; i < 3 + E::NQ; ++i) {
            if (set) pl<E>(f, L::FV + i, B, b) = buf[(size_t)b * (3 + E::NQ) + i];
            else buf[(size_t)b * (3 + E::NQ) + i] = pl<E>(f, L::FV + i, B, b);
        }
    }
}

// servo-joint state <-> [B, 6] = [q7, qu1, qu2, dq7, dqu1, dqu2]  (iiwa)
template <typename T, typename E>
__global__ void k_get_aux(int B, const T* __restrict__ f, T* __restrict__ out) {
    using L = Planes<E>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int i = 0; i < 6; ++i) out[(size_t)b * 6 + i] = pl<E>(f, L::QX + i, B, b);
}
template <typename T, typename E>
__global__ void k_set_aux(int B, T* __restrict__ f, const T* __restrict__ in) {
    using L = Planes<E>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int i = 0; i < 6; ++i) pl<E>(f, L::QX + i, B, b) = in[(size_t)b * 6 + i];
}

// row N4 primitives (atacom_inverse_dynamics / atacom_forward_dynamics)
template <typename T>
__global__ void __launch_bounds__(WAVE) k_inverse_dynamics(int n, const T* __restrict__ q, const T* __restrict__ dq,
                                                           const T* __restrict__ ddq, T* __restrict__ tau,
                                                           T* __restrict__ M) {
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if (b >= n) return;
    T q9[9], dq9[9], dd9[9], t9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { q9[i] = q[(size_t)b * 9 + i]; dq9[i] = dq[(size_t)b * 9 + i]; dd9[i] = ddq[(size_t)b * 9 + i]; }
    T Ml[9][9];
#if ATACOM_DYN_LINK
    lk::Trig9<T> tg;
    lk::trig9(q9, tg);
    lk::rnea9(tg, dq9, dd9, t9);
    if (M) lk::crba<T, 9>(tg, Ml);
#else
    Chain9<T> ch;
    iiwa_chain9(q9, ch);
    rnea9(ch, dq9, dd9, t9);
    if (M) crba<T, 9>(ch, Ml);
#endif
#pragma unroll
    for (int i = 0; i < 9; ++i) tau[(size_t)b * 9 + i] = t9[i];
    if (M) {
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) M[((size_t)b * 9 + i) * 9 + j] = M[((size_t)b * 9 + j) * 9 + i] = Ml[i][j];
    }
}
template <typename T>
__global__ void __launch_bounds__(WAVE) k_forward_dynamics(int n, const T* __restrict__ q, const T* __restrict__ dq,
                                                           const T* __restrict__ tau6, const T* __restrict__ ddq_aux,
                                                           int use_damping, T* __restrict__ ddq6) {
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if (b >= n) return;
    T q9[9], dq9[9], dd9[9], bias[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        q9[i] = q[(size_t)b * 9 + i]; dq9[i] = dq[(size_t)b * 9 + i];
        dd9[i] = (i >= 6 && ddq_aux) ? ddq_aux[(size_t)b * 3 + (i >= 6 ? i - 6 : 0)] : T(0);
    }
    T rhs[6], Ml[6][6];
#if ATACOM_DYN_LINK
    lk::Trig9<T> tg;
    lk::trig9(q9, tg);
    lk::rnea9(tg, dq9, dd9, bias);
    lk::crba<T, 6>(tg, Ml);
#else
    Chain9<T> ch;
    iiwa_chain9(q9, ch);
    rnea9(ch, dq9, dd9, bias);
    crba<T, 6>(ch, Ml);
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i)
        rhs[i] = tau6[(size_t)b * 6 + i] - bias[i] - (use_damping ? (T)iiwa_body::DAMPING[i] * dq9[i] : T(0));
    chol_solve<T, 6>(Ml, rhs);
#pragma unroll
    for (int i = 0; i < 6; ++i) ddq6[(size_t)b * 6 + i] = rhs[i];
}

// ------------------------------------------------------------------ stand-alone primitives (parity tests)
template <typename T, typename E>
__global__ void __launch_bounds__(WAVE) k_nullspace(int n, const T* __restrict__ Jc, const T* __restrict__ rhs,
                                                    T tol, T* __restrict__ xo, T* __restrict__ nullo,
                                                    T* __restrict__ rrefo) {
    constexpr int NC = E::NC, NN = E::NN, NK = E::NN - E::NC;     // NK here = null-space dimension
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if (b >= n) return;
    T x[NN], nb[NN][NK];
    auto aget = [&](auto rc, auto cc) -> T {
        return Jc[((size_t)b * NC + decltype(rc)::value) * NN + decltype(cc)::value];
    };
    auto yget = [&](auto rc) -> T { return rhs ? rhs[(size_t)b * NC + decltype(rc)::value] : T(0); };
    bidiag_solve_null<T, NC, NN>(aget, yget, x, nb);
    if (xo) {
#pragma unroll
        for (int c = 0; c < NN; ++c) xo[(size_t)b * NN + c] = x[c];
    }
    if (nullo) {
#pragma unroll
        for (int c = 0; c < NN; ++c)
#pragma unroll
            for (int k = 0; k < NK; ++k) nullo[((size_t)b * NN + c) * NK + k] = nb[c][k];
    }
    if (rrefo) {
        // column k of rref(null) = rref_apply with alpha = e_k; the chart itself is computed once per k
#pragma unroll 1
        for (int k = 0; k < NK; ++k) {
            T nb2[NN][NK], alpha[NK], col[NN];
#pragma unroll
            for (int c = 0; c < NN; ++c)
#pragma unroll
                for (int j = 0; j < NK; ++j) nb2[c][j] = nb[c][j];
#pragma unroll
            for (int j = 0; j < NK; ++j) alpha[j] = (j == k) ? T(1) : T(0);
            rref_apply<T, NN, NK>(nb2, alpha, tol, col);
#pragma unroll
            for (int c = 0; c < NN; ++c) rrefo[((size_t)b * NN + c) * NK + k] = col[c];
        }
    }
}

// the same primitive through the lane-group solver (LN = 4 or 2 lanes per matrix)
template <typename T, typename E, int LN>
__global__ void __launch_bounds__(WAVE) k_nullspace_quad(int n, const T* __restrict__ Jc, const T* __restrict__ rhs,
                                                         T tol, T* __restrict__ xo, T* __restrict__ nullo,
                                                         T* __restrict__ rrefo) {
    constexpr int NC = E::NC, NN = E::NN, NK = E::NN - E::NC, S = split_slots(NN, LN);
    const int gt = blockIdx.x * WAVE + threadIdx.x;
    const int b = gt / LN, lq = gt % LN;
    if (b >= n) return;
    T x0, x[S], nb0[NK], nb[S][NK];
    auto aget = [&](auto rc, auto sc) -> T {
        constexpr int r = decltype(rc)::value, sl = decltype(sc)::value;
        const int c = LN * sl + lq + 1;
        return (c < NN) ? Jc[((size_t)b * NC + r) * NN + (c < NN ? c : 0)] : T(0);
    };
    auto a0get = [&](auto rc) -> T { return Jc[((size_t)b * NC + decltype(rc)::value) * NN]; };
    auto yget = [&](auto rc) -> T { return rhs ? rhs[(size_t)b * NC + decltype(rc)::value] : T(0); };
    bidiag_solve_null_quad<T, NC, NN, LN>(aget, a0get, yget, x0, x, nb0, nb, lq);
    if (lq == 0) {
        if (xo) xo[(size_t)b * NN] = x0;
        if (nullo) {
#pragma unroll
            for (int k = 0; k < NK; ++k) nullo[(size_t)b * NN * NK + k] = nb0[k];
        }
    }
#pragma unroll
    for (int sl = 0; sl < S; ++sl) {
        const int c = LN * sl + lq + 1;
        if (c < NN) {
            if (xo) xo[(size_t)b * NN + c] = x[sl];
            if (nullo) {
#pragma unroll
                for (int k = 0; k < NK; ++k) nullo[((size_t)b * NN + c) * NK + k] = nb[sl][k];
            }
        }
    }
    if (rrefo) {
#pragma unroll 1
        for (int k = 0; k < NK; ++k) {
            T nb2[S][NK], nb02[NK], alpha[NK], col[S], col0;
#pragma unroll
            for (int j = 0; j < NK; ++j) nb02[j] = nb0[j];
#pragma unroll
            for (int sl = 0; sl < S; ++sl)
#pragma unroll
                for (int j = 0; j < NK; ++j) nb2[sl][j] = nb[sl][j];
#pragma unroll
            for (int j = 0; j < NK; ++j) alpha[j] = (j == k) ? T(1) : T(0);
            rref_apply_quad<T, NN, NK, LN>(nb02, nb2, alpha, tol, col0, col, lq);
            if (lq == 0) rrefo[(size_t)b * NN * NK + k] = col0;
#pragma unroll
            for (int sl = 0; sl < S; ++sl) {
                const int c = LN * sl + lq + 1;
                if (c < NN) rrefo[((size_t)b * NN + c) * NK + k] = col[sl];
            }
        }
    }
}

// the canonical chart as a stand-alone primitive (atacom_canonical_mu): A [n, NC, NQ], s [n, NG], y [n, NC], alpha [n, NK]
template <typename T, typename E>
__global__ void __launch_bounds__(WAVE) k_chart(int n, const T* __restrict__ Ain, const T* __restrict__ sin_,
                                                const T* __restrict__ yin, const T* __restrict__ ain, T tol,
                                                T* __restrict__ mu_o) {
    constexpr int NQ = E::NQ, NF = E::NF, NG = E::NG, NC = E::NC, NN = E::NN, NK = NQ - NF;
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if (b >= n) return;
    T A[NC][NQ], arow[NG], s[NG], y[NC], alpha[NK], mu[NN];
#pragma unroll
    for (int r = 0; r < NC; ++r) {
        y[r] = yin[(size_t)b * NC + r];
#pragma unroll
        for (int i = 0; i < NQ; ++i) A[r][i] = E::jac_zero(r, i) ? T(0) : Ain[((size_t)b * NC + r) * NQ + i];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        s[g] = sin_[(size_t)b * NG + g];
        T m = T(0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) m = num<T>::max(m, num<T>::abs(A[NF + g][i]));
        arow[g] = m;
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) alpha[k] = ain[(size_t)b * NK + k];
#ifdef ATACOM_TIMESTAMPS
    int dbg[3] = {0, 0, 0};
#endif
    canonical_mu<T, E>(A, arow, s, y, alpha, tol, mu ATACOM_DBG_ARG(dbg));
#pragma unroll
    for (int c = 0; c < NN; ++c) mu_o[(size_t)b * NN + c] = mu[c];
}

template <typename T, typename E>
__global__ void __launch_bounds__(WAVE) k_terms(const Params<T> P, int n, const T* __restrict__ q,
                                                const T* __restrict__ dq, T* __restrict__ fun_o,
                                                T* __restrict__ J_o, T* __restrict__ b_o) {
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if (b >= n) return;
    T qq[E::NQ], dd[E::NQ], fun[E::NC], J[E::NC][E::NQ], bst[E::NC];
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) { qq[i] = q[(size_t)b * E::NQ + i]; dd[i] = dq[(size_t)b * E::NQ + i]; }
    constraint_terms(E{}, P, qq, dd, fun, J, bst);
#pragma unroll
    for (int r = 0; r < E::NC; ++r) {
        fun_o[(size_t)b * E::NC + r] = fun[r];
        b_o[(size_t)b * E::NC + r] = bst[r];
#pragma unroll
        for (int i = 0; i < E::NQ; ++i) J_o[((size_t)b * E::NC + r) * E::NQ + i] = J[r][i];
    }
}

}  // namespace atacom
