/* atacom_hip.h -- C ABI of libatacom_hip.so, the MI355X (gfx950) batched ATACOM environment-step engine.
 *
 * The reference (PuzeLiu/rl_on_manifold) is pure Python and has no FFI; this header is the boundary a
 * maintainer would bind with ctypes (INTEGRATION.md shows the stub).  Each entry point names the
 * reference interface it replaces (paths relative to /root/reference/).  Plain C types only; every
 * d_* pointer is DEVICE memory owned by the caller (e.g. a torch ROCm tensor's data_ptr()); `stream`
 * is a hipStream_t (0 / NULL = the null stream).  All work is enqueued asynchronously on `stream`
 * except atacom_get_stats, which synchronises that stream to return three numbers to the host.
 *
 * Floating-point type: every float buffer of a handle has the element type chosen at creation
 * (cfg.dtype: 0 = float32 -- the production path; 1 = float64 -- same kernels instantiated in double,
 * used to show algorithmic identity with the float64 reference to ~1e-10).
 *
 * Return value: 0 on success, negative on error (ATACOM_E_*); atacom_last_error() gives the message of
 * the last failing call on the calling thread.  No C++ exception crosses this boundary.
 * Thread-safety: a handle must not be used from two threads at once (the reference is single-threaded
 * too); distinct handles are independent.
 */
#ifndef ATACOM_HIP_H
#define ATACOM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATACOM_ENV_CIRCLE 0 /* atacom/environments/circular_motion/circle_atacom.py:6  CircleEnvAtacom      */
#define ATACOM_ENV_PLANAR 1 /* atacom/environments/planar_air_hockey/atacom_air_hockey.py:11 (task 'H')     */
#define ATACOM_ENV_IIWA 2   /* atacom/environments/iiwa_air_hockey/iiwa_hit_atacom.py:10 (env '7H')         */
/* baseline comparators of the circle experiment (examples/circle_exp.py:85-94), row N3: */
#define ATACOM_ENV_CIRCLE_EC 3 /* circular_motion/circle_error_correction.py:7 (env 'E'): action = q'' directly + error correction */
#define ATACOM_ENV_CIRCLE_T 4  /* circular_motion/circle_terminated.py:8 (env 'T'): unconstrained, terminate with -100 at c > tol */

#define ATACOM_F32 0
#define ATACOM_F64 1

#define ATACOM_OK 0
#define ATACOM_E_INVALID (-1)     /* bad argument / inconsistent config */
#define ATACOM_E_HIP (-2)         /* a HIP runtime call failed */
#define ATACOM_E_UNSUPPORTED (-3)

#define ATACOM_MAX_C 12 /* constraint rows (iiwa: 1 equality + 11 inequalities) */
#define ATACOM_MAX_Q 6  /* controlled joints */

/* POD configuration.  atacom_default_config() fills the reference's constants for an environment:
 *   circle  circle_atacom.py:9-18   (K_f .1, K_g 2, Kc 100, acc_max 10, vel_max 1, Kq 20, dt .01, horizon 500)
 *   planar  atacom_air_hockey.py:12-43 + examples/planar_air_hockey_exp.py:102-105
 *   iiwa    iiwa_hit_atacom.py:11-40   + examples/iiwa_air_hockey_exp.py:104-107
 * Row order of K / Kc: equality rows (f) first, then inequality rows (g), as atacom.py:151-165 stacks them. */
typedef struct atacom_config {
    int32_t struct_size; /* = sizeof(atacom_config); checked by atacom_create */
    int32_t env_id;      /* ATACOM_ENV_* */
    int32_t batch;       /* number of independent environments held by the handle */
    int32_t dtype;       /* ATACOM_F32 / ATACOM_F64 */
    int32_t substeps;    /* n_intermediate_steps: physics sub-steps per env step (circle 1, others 4) */
    int32_t horizon;     /* MDPInfo.horizon */
    int32_t hold_q;      /* 1 = reference behaviour: q, dq frozen across the sub-steps of one step
                            (atacom.py:124-126 never refreshes self.q / self.dq) */
    int32_t bias_mode;   /* 0 = reference "classical acceleration" w x v (quirk Q2); 1 = exact dJ/dt dq */
    int32_t auto_reset;  /* 1 = an env whose step returned last=1 is re-initialised inside the same call
                            (what mushroom_rl.Core does between steps); the returned obs is still the
                            terminal observation, the next call starts from the reset state */
    int32_t lanes_per_env; /* kernel mapping: 1 = one env per lane, 2 = one env per lane pair, 4 = one env per DPP
                              quad, 8 = one env per 8 lanes (null-space solve split by column over the 2 / 4 / 8
                              lanes), 0 = let the library choose per env / batch / kernel (iiwa: 8 up to 8192
                              envs, 4 up to 16384, 2 up to 32768, 1 beyond -- a STATIC policy: atacom_create launches
                              nothing hidden and two handles of the same config run the same mappings in every
                              process, on every box and on every rank of a sharded collection; atacom_get_lanes
                              reports them.  ATACOM_CALIBRATE=1 (or =verbose) in the environment opts in to timing
                              8 lanes against 4 for atacom_step of iiwa at 4096 < batch <= 8192 once per process,
                              about 20 ms -- the bits then depend on the box).
                              Results are the same algorithm in every mapping (summation order differs).
                              Not every mapping exists for every handle; a request runs on the widest instantiated mapping
                              that is not wider, and atacom_get_lanes reports it: the circle family runs one environment per
                              lane; float64 handles 1, 4 and (iiwa) 8 lanes -- policy: iiwa 8 up to 8192 envs, 4
                              beyond (canonical chart: 4 up to 16384, else 1); planar 4 up to 16384, else 1; the rigid-body kernels 1 and 4; the planar T-step
                              kernels take 8 lanes up to 8192 envs where single steps stay on 4; the policy kernel
                              (atacom_rollout_mlp) follows the T-step mapping where it has that form (float64: 4 or 1) --
                              atacom_get_policy_lanes. */
    double dt;           /* time_step */
    double rref_tol;     /* 0.05, atacom.py:128 */
    double action_penalty; /* env_hitting.py:10,68 */
    double gamma;
    double K[ATACOM_MAX_C];       /* ViabilityConstraint.K per row */
    double Kc[ATACOM_MAX_C];      /* AtacomEnvWrapper.K_c per row */
    double vel_max[ATACOM_MAX_Q];
    double acc_max[ATACOM_MAX_Q];
    double Kq[ATACOM_MAX_Q];
    double pos_limit[ATACOM_MAX_Q]; /* joint position limits (upper; lower = -upper) */
    double base_xy[2];            /* robot base in the table frame (env_base.py:50) */
    double link[3];               /* planar arm link lengths */
    double term_tol;              /* ATACOM_ENV_CIRCLE_T: termination tolerance (circle_terminated.py:13, 0.1) */
    int32_t random_init;          /* 1 = every reset (explicit without a state, or auto) re-draws the random part of the
                                     reference's reset on the device: circle point + tangential velocity
                                     (circle_base.py:36-42), puck position in hit_range (env_hitting.py:24-25) */
    int32_t seed;                 /* seed of the counter-based generator hash(seed, env, episode, draw) */
    int32_t dynamics_mode;        /* 0 = kinematic: inverse dynamics o forward dynamics taken as the identity, q'' integrates
                                     directly (default).  1 = rigid body (ATACOM_ENV_IIWA only, row N4):
                                     per sub-step the torque is the inverse dynamics of the nine-joint chain
                                     (iiwa_hit_atacom.py:58-63), saturated at the URDF effort limits, and the controlled
                                     joints follow the forward dynamics under it with the URDF joint damping, joint 7 and
                                     the striker's universal joint riding position servos (env_single.py:137-185), their
                                     motors limited by the URDF efforts (40 / 10 / 10 Nm).  2 = the same with the servo joints'
                                     accelerations fed forward into the inverse dynamics -- NOT what the reference computes
                                     (it passes zeros, iiwa_hit_atacom.py:58-63), but what keeps ATACOM's velocity guarantee
                                     when the joint-7 set-point of env_single.py:137-170 chatters near q6 = 0.  Runs one
                                     environment per lane or per quad (a request for 8 / 2 lanes maps to 4 / 1;
                                     atacom_get_lanes reports the mapping that runs) */
    int32_t chart_mode;           /* 0 = the reference's chart: LAPACK's null basis + rref with the 0.05 tolerance, reproduced
                                     decision by decision (atacom.py:127-133, null_space_coordinate.py:8-79) -- the default.
                                     1 = canonical chart (opt-in): the same mu wherever the reference's rref
                                     takes no tolerance branch, but the free coordinates are chosen by the basis-independent
                                     test ||P_S e_j|| > rref_tol, nothing is ever zeroed (N is an exact null basis on every
                                     state) and no factorisation of Jc is needed (rl_on_manifold_amd/csrc/atacom_chart.h;
                                     specification oracle/canonical_chart.py).  ATACOM environments only. */
    int32_t task;                 /* ATACOM_ENV_PLANAR only: 0 = hitting (task 'H', AirHockeyHit -- the default and the BASELINE
                                     configuration), 1 = defending (task 'D', atacom_air_hockey.py:22-27 -> mushroom_rl's
                                     AirHockeyDefend [upstream, restated from memory -- not in the reference tree]: the puck
                                     starts in the opponent's half moving towards the agent's goal; has_hit latches on
                                     mallet contact, has_bounce on the agent-side end rims; absorbing when the puck is back
                                     in the opponent's half after either, -50 for conceding a goal).  The reference's iiwa
                                     wrapper raises NotImplementedError for 'D' (iiwa_hit_atacom.py:20-21), so does this */
    int32_t reserved0;            /* keep 0 */
    double dt_base;               /* time step of the BASE environment's integrator when it differs from `dt` (0 = same).
                                     The reference's CircleEnvAtacom / CircleEnvErrorCorrection hand time_step to the wrapper
                                     only -- slack integration, atacom.py:135 -- while the base CircularMotion keeps its
                                     default 0.01 (circle_atacom.py:7-18, circle_base.py:18-19,62-63): defaults reproduce that */
    /* Domain randomisation of the air-hockey base environments -- the constructor kwargs of iiwa_hit_atacom.py:11-13 /
     * atacom_air_hockey.py:12-14, all 0 by default (ATACOM_ENV_PLANAR / _IIWA only).  The reference draws from numpy's global
     * unseeded generator; here every draw is a counter-based hash of (seed, env, episode, step, draw index), so a run is
     * reproducible and identical to the oracle's draw for draw. */
    int32_t obs_noise;            /* 1 = puck pose (x, y, yaw) of every observation += N(0, 0.001^2)   (env_single.py:105-107) */
    int32_t obs_delay;            /* 1 = puck and joint velocities of every observation are low-passed, alpha = 0.5
                                     (env_single.py:114-117), advanced at every physics sub-step (step_action_function builds an
                                     observation per sub-step, atacom.py:124) and once more for the returned observation; the
                                     wrapper then controls on -- and logs c_dq_max of -- the FILTERED joint velocities
                                     (atacom.py:95-96,111-114).  As written, the reference's filter raises on an iiwa episode's
                                     first observation (obs_prev is None) and slices obs_prev[9:12] for six joint velocities:
                                     what is built is the evident intent -- first observation unfiltered, each velocity
                                     low-passed with its own previous value */
    int32_t env_noise;            /* 1 = a random planar force 0.0005 N(0, 1) N on the puck in every physics sub-step
                                     (env_base.py:176-180), i.e. a velocity kick 0.0005 dt / puck_mass per unit draw */
    int32_t reserved1;            /* keep 0 */
    double puck_mass;             /* kg; scales env_noise.  0.01 by default [MushroomRL's puck.urdf, upstream, from memory --
                                     not in the reference tree] */
} atacom_config;

typedef struct atacom_handle atacom_handle;

/* Static shape information of an environment (what AtacomEnvWrapper.dims / MDPInfo expose). */
typedef struct atacom_dims {
    int32_t dim_q, n_f, n_g, n_null /* = action dim, atacom.py:39,51 */, obs_dim;
    int32_t state_dim;      /* floats per env in atacom_get_state / atacom_set_state:
                               [q, dq, s, puck(6), has_hit, r_hit, vel_hit_x, t]  (task 'D': the has_hit slot holds
                               has_hit + 2 has_bounce) */
    int32_t init_state_dim; /* floats per env in atacom_reset's d_init_state: [q, dq] (+ puck(6) for planar/iiwa) */
    int32_t record_dim;     /* floats per (step, env) record of atacom_rollout_packed: 2 obs_dim + n_null + 3 */
} atacom_dims;

int atacom_default_config(int32_t env_id, atacom_config* out);
int atacom_get_dims(int32_t env_id, atacom_dims* out);

/* Replaces constructing the wrapper + base env (atacom.py:10-79; circle_base.py:18-31; env_hitting.py:8-21).
 * Allocates the persistent per-env state (fields in groups of four, [group][env][4]) on `device`. */
int atacom_create(const atacom_config* cfg, int device, atacom_handle** out);
int atacom_destroy(atacom_handle* h);

/* AtacomEnvWrapper.reset (atacom.py:93-98) for every env whose mask byte is non-zero (all if d_mask is
 * NULL): state <- stored initial state, slack s <- sqrt(max(-2 g(q, dq), 0)) (atacom.py:145-149), flags
 * and step counter cleared.  If d_init_state is non-NULL its rows ([batch, init_state_dim]) first replace
 * the stored initial state of the masked envs (base_env.reset(state), circle_base.py:33-51,
 * env_hitting.py:23-37).  d_obs ([batch, obs_dim], may be NULL) receives the observation of ALL envs. */
int atacom_reset(atacom_handle* h, const uint8_t* d_mask, const void* d_init_state, void* d_obs, void* stream);

/* AtacomEnvWrapper.step (atacom.py:106-115) for the whole batch: action clip + scale, `substeps` x
 * [step_action_function (atacom.py:123-139) -> dynamics], absorbing / reward / observation, constraint
 * statistics.  d_action [batch, n_null]; d_obs [batch, obs_dim]; d_reward [batch]; d_absorbing [batch]
 * (uint8); d_last [batch] (uint8, may be NULL) = absorbing or step counter reached the horizon.
 * Non-finite actions do not reach the state: the clip to [-1, 1] (atacom.py:107) is IEEE min / max, so +-Inf clip like any
 * large value and a NaN component acts as -1 (numpy's clip would propagate the NaN into q and poison that environment for
 * the rest of the episode); other environments of the batch are never affected either way. */
int atacom_step(atacom_handle* h, const void* d_action, void* d_obs, void* d_reward, uint8_t* d_absorbing,
                uint8_t* d_last, void* stream);

/* The same step for a subset: environments whose d_mask byte is 0 sit the call out -- state, step counter and constraint
 * statistics untouched; their d_obs row is their current observation, d_reward 0, d_absorbing 0, d_last 0 (what a
 * vectorised Core does with its finished environments, without a host round trip).  d_mask NULL = all. */
int atacom_step_masked(atacom_handle* h, const uint8_t* d_mask, const void* d_action, void* d_obs, void* d_reward,
                       uint8_t* d_absorbing, uint8_t* d_last, void* stream);

/* n_steps consecutive steps in ONE kernel launch (per-env state stays in registers between steps).
 * d_actions [n_steps, batch, n_null]; outputs are time-major: d_obs [n_steps, batch, obs_dim] holds the
 * observation BEFORE each step, d_next_obs (may be NULL) the observation after it, d_reward / d_absorbing /
 * d_last [n_steps, batch] -- the (s, a, r, s', absorbing, last) tuples mushroom_rl.Core collects. */
int atacom_rollout(atacom_handle* h, int32_t n_steps, const void* d_actions, void* d_obs, void* d_next_obs,
                   void* d_reward, uint8_t* d_absorbing, uint8_t* d_last, void* stream);

/* Row N2 -- rollout with the actor network evaluated inside the kernel.  The network is the 2-hidden-layer MLP every
 * reference training script builds (examples/network.py:8-36,39-68,266-293: Linear(n_in,64)-ReLU-Linear(64,64)-ReLU-
 * Linear(64,n_out)); weights in torch.nn.Linear layout W[out][in], element type = the handle's dtype, device memory.
 * action = MLP((obs - obs_shift) * obs_scale) + std * noise   (MinMaxPreprocessor + GaussianTorchPolicy of
 * examples/iiwa_air_hockey_exp.py:32-34,138-146); d_noise [n_steps, batch, n_out] is supplied by the caller (NULL = 0). */
typedef struct atacom_mlp {
    int32_t struct_size;  /* = sizeof(atacom_mlp) */
    int32_t n_in;         /* must equal obs_dim */
    int32_t hidden;       /* units of both hidden layers; 64 supported */
    int32_t n_out;        /* must equal n_null */
    int32_t activation;   /* 0 = ReLU, 1 = tanh */
    int32_t reserved;
    const void *W1, *b1, *W2, *b2, *W3, *b3;
    const void *obs_shift, *obs_scale; /* [n_in], may be NULL (identity) */
    const void *std;                   /* [n_out], may be NULL (deterministic) */
    /* optional SAC-style policy (examples/iiwa_air_hockey_exp.py:301-339: actor_mu + actor_sigma networks of the same
     * architecture): a second network giving log(sigma) per action dim, clamped to [log_std_min, log_std_max], replaces
     * `std`; squash = 1 applies tanh to mean + sigma * eps (the squashed Gaussian SAC samples from). */
    const void *sW1, *sb1, *sW2, *sb2, *sW3, *sb3; /* all NULL = no sigma network */
    double log_std_min, log_std_max;               /* MushroomRL's SACPolicy uses -20, 2 */
    int32_t squash;
    int32_t reserved1;
} atacom_mlp;

/* Like atacom_rollout, with d_actions [n_steps, batch, n_out] an OUTPUT (the actions the policy drew). */
int atacom_rollout_mlp(atacom_handle* h, int32_t n_steps, const atacom_mlp* net, const void* d_noise, void* d_obs,
                       void* d_next_obs, void* d_actions, void* d_reward, uint8_t* d_absorbing, uint8_t* d_last,
                       void* stream);

/* The same rollouts writing ONE packed float record per (step, env) instead of six arrays:
 *   d_records [n_steps, record_batch_stride, record_dim],
 *   record = [obs(obs_dim) | action(n_null) | reward | next_obs(obs_dim) | absorbing (0/1) | last (0/1)]
 * -- the (s, a, r, s', absorbing, last) tuple of mushroom_rl.Core's dataset (what examples/circle_exp.py:72
 * `core.learn(...)` hands to agent.fit), laid out so that a sharded collector can all-gather the buffer as it is
 * (rl_on_manifold_amd/rollout.py: one collective, no repacking pass).  Exactly one of d_actions
 * ([n_steps, batch, n_null], pre-generated actions as in atacom_rollout) and net (policy evaluated in the kernel as in
 * atacom_rollout_mlp, with d_noise) must be given.  record_batch_stride >= batch lets ragged shards share one padded
 * buffer shape; rows batch..stride-1 are not written. */
int atacom_rollout_packed(atacom_handle* h, int32_t n_steps, const void* d_actions, const atacom_mlp* net,
                          const void* d_noise, void* d_records, int32_t record_batch_stride, void* stream);

/* get_constraints_logs (atacom.py:207-216; circle_base.py:109-115): out = {c_avg, c_max, c_dq_max} over every
 * (env, step) logged since the last clear.  Synchronises `stream`. */
int atacom_get_stats(atacom_handle* h, double out[3], int32_t clear, void* stream);

/* env.seed(seed) (atacom.py:90-91 -> the base env's seed): re-keys the counter-based generator behind the device-side draws
 * (random_init, obs_noise, env_noise: hash(seed, env, episode, step, draw)) from the next launch on.  The key is stored as
 * seed & 0x7fffffff (atacom_create stores cfg.seed the same way).  No kernel launch; one synchronous 64-byte host-to-device
 * copy keeps the snapshot header current (the key travels with atacom_snapshot_save).  Launches already captured in a HIP
 * graph keep the seed they were captured with. */
int atacom_set_seed(atacom_handle* h, int32_t seed);

/* The kernel mappings this handle REALLY runs: lanes per environment (1, 2, 4 or 8) of atacom_step and of the T-step kernels
 * (atacom_rollout / _packed with actions) -- cfg.lanes_per_env, or what the library chose for lanes_per_env = 0, narrowed to
 * the mappings instantiated for the handle's kernel variant (see lanes_per_env above; the two may differ: the persistent
 * state does not depend on the mapping).  Either output pointer may be NULL.
 * atacom_get_policy_lanes: the same for the policy kernel (atacom_rollout_mlp, atacom_rollout_packed with a network), which
 * has fewer forms (float64: quad and lane; float32 8 lanes: the matrix-core form). */
int atacom_get_lanes(const atacom_handle* h, int32_t* out_step_lanes, int32_t* out_rollout_lanes);
int atacom_get_policy_lanes(const atacom_handle* h, int32_t* out_policy_lanes);

/* Parity injection: d_state [batch, state_dim] = [q, dq, s, puck(6), has_hit, r_hit, vel_hit_x, t].  NOT a checkpoint (use
 * atacom_snapshot_*): the stored initial states, statistics, episode counters and servo joints are not part of it, and on a
 * handle with cfg.obs_delay atacom_set_state RESTARTS the low-pass behind the observation's velocities on the injected state
 * (as a reset does: a stale filter would have the controller's dq disagree with the state just set) -- so get_state followed
 * by set_state does not preserve the filter.  To inject a particular filter state call atacom_set_filter_state AFTER
 * atacom_set_state (that order only). */
int atacom_get_state(atacom_handle* h, void* d_state, void* stream);
int atacom_set_state(atacom_handle* h, const void* d_state, void* stream);

/* Checkpoint / resume of the WHOLE persistent state of a handle -- everything atacom_get_state leaves out as well: the stored
 * initial states, the constraint-statistics accumulators (atacom.py:201-205), the episode counters of the device-side random
 * reset and the servo joints of the rigid-body mode.  An opaque byte image (device memory, caller-owned, at least
 * atacom_snapshot_bytes(h) bytes) valid for a handle created from the same atacom_config.  It starts with a 64-byte header
 * (magic, environment, task, dtype, batch, fields per environment): atacom_snapshot_restore reads it back first -- one
 * 64-byte device-to-host copy, which synchronises `stream` -- and returns ATACOM_E_INVALID for an image of another handle
 * shape instead of mis-reading it; save is three device-to-device copies on `stream`, no synchronisation.
 * restore(save(x)) followed by the same calls reproduces the run bit for bit (the reference has no counterpart: its envs
 * are Python objects one would pickle).  The header also records the kernel mappings of the writing handle
 * (atacom_get_lanes) and its generator key (cfg.seed / atacom_set_seed): a restoring handle adopts the key, and -- if created
 * with lanes_per_env = 0 -- the mappings, so the replay is bit for bit in another process as well; a handle with a named
 * mapping keeps its own (the state does not depend on the mapping).  Launches already captured in a HIP graph keep the
 * mapping and key they were captured with; a RolloutCollector whose handle's mappings changed refuses to collect (rollout.py).
 * The header carries a format number: an image written by another library version is refused with that message
 * (ATACOM_E_INVALID), not mis-read. */
int64_t atacom_snapshot_bytes(const atacom_handle* h);
int atacom_snapshot_save(atacom_handle* h, void* d_image, void* stream);
int atacom_snapshot_restore(atacom_handle* h, const void* d_image, void* stream);

/* Row N4: the three servo joints of the rigid-body mode (joint 7, striker_joint_1, striker_joint_2):
 * d_aux [batch, 6] = [q7, qu1, qu2, dq7, dqu1, dqu2].  ATACOM_ENV_IIWA handles only. */
int atacom_get_aux_state(atacom_handle* h, void* d_aux, void* stream);
int atacom_set_aux_state(atacom_handle* h, const void* d_aux, void* stream);

/* obs_delay (cfg.obs_delay = 1; planar / iiwa handles): the low-pass state behind the observation's velocities,
 * d_filter [batch, 3 + dim_q] = [puck vx, vy, yaw rate, joint velocities] as the last observation showed them
 * (obs_prev[3:6] and the robot-velocity slice of env_single.py:114-119).  Every reset re-initialises it to the unfiltered
 * velocities, and so does atacom_set_state: call atacom_set_filter_state AFTER it (the other order is overwritten).  get / set
 * exist for parity injection; atacom_snapshot_* carries the filter. */
int atacom_get_filter_state(atacom_handle* h, void* d_filter, void* stream);
int atacom_set_filter_state(atacom_handle* h, const void* d_filter, void* stream);

/* Row N4 primitives on n states of the nine movable joints of urdf/iiwa_1.urdf (q, dq, ddq: [n, 9] = joint_1..7,
 * striker_joint_1, striker_joint_2):
 *   atacom_inverse_dynamics: d_tau [n, 9] = M(q) ddq + C(q, dq) dq + g(q)   -- PyBullet calculateInverseDynamics as the
 *     reference calls it (iiwa_hit_atacom.py:58-63; gravity (0, 0, -9.81), no joint damping); d_M [n, 9, 9] (may be
 *     NULL) receives the joint-space inertia matrix;
 *   atacom_forward_dynamics: d_ddq6 [n, 6] = accelerations of the six controlled joints under the torques d_tau6
 *     [n, 6], the servo joints following the prescribed accelerations d_ddq_aux [n, 3] (NULL = 0), with
 *     (use_damping != 0) or without the URDF joint damping. */
int atacom_inverse_dynamics(int32_t dtype, int32_t n, const void* d_q, const void* d_dq, const void* d_ddq, void* d_tau,
                            void* d_M, void* stream);
int atacom_forward_dynamics(int32_t dtype, int32_t n, const void* d_q, const void* d_dq, const void* d_tau6,
                            const void* d_ddq_aux, int32_t use_damping, void* d_ddq6, void* stream);

/* Stand-alone batched primitives (no handle), for parity tests of individual reference functions.
 *   atacom_nullspace: for n matrices Jc [n, c, c+k] (row-major) and right-hand sides [n, c]:
 *     d_x [n, c+k]        = Jc^+ rhs                         (pinv_null, null_space_coordinate.py:8-26)
 *     d_null [n, c+k, k]  = orthonormal null basis           (same function, second output)
 *     d_rref [n, c+k, k]  = rref(null, row_vectors=False, tol) (null_space_coordinate.py:40-79)
 *   shape is given by env_id (circle 2x3, planar 6x9, iiwa 12x17).
 *   atacom_constraint_terms: q, dq [n, dim_q] -> fun [n, c], J [n, c, dim_q], b [n, c]: the fun / J / b
 *     callables handed to ViabilityConstraint (circle_atacom.py:47-70, atacom_air_hockey.py:78-107,
 *     iiwa_hit_atacom.py:70-139).  cfg supplies geometry and bias_mode. */
/*   atacom_canonical_mu: the canonical chart (chart_mode 1) as a primitive, for n systems
 *     d_A [n, c, dim_q] = K J (equality row first), d_s [n, n_g], d_y [n, c] = psi + Kc c, d_alpha [n, k]
 *     -> d_mu [n, dim_q + n_g] = -Jc^+ y + N alpha.  With y = 0 and alpha = e_i it returns column i of the chart's null
 *     basis N (the invariant tests Jc N = 0, Jc mu + y = 0 are built on that). */
/*   Structural zeros: for the planar / iiwa shapes the entries of d_A that the environment's constraint Jacobian leaves
 *     structurally zero (joint-limit rows: everything off their diagonal; iiwa row 4, the height of link_4: joints 3..6) are
 *     NOT READ -- the kernels assemble K J without them -- so a general or perturbed matrix passed here is treated as
 *     having zeros there. */
int atacom_canonical_mu(int32_t env_id, int32_t dtype, int32_t n, const void* d_A, const void* d_s, const void* d_y,
                        const void* d_alpha, double tol, void* d_mu, void* stream);
int atacom_nullspace(int32_t env_id, int32_t dtype, int32_t lanes_per_env /* 1, 2, 4 or 8; narrowed like a handle's request */, int32_t n, const void* d_Jc,
                     const void* d_rhs, double tol, void* d_x, void* d_null, void* d_rref, void* stream);
int atacom_constraint_terms(const atacom_config* cfg, int32_t n, const void* d_q, const void* d_dq, void* d_fun,
                            void* d_J, void* d_b, void* stream);

const char* atacom_last_error(void);
const char* atacom_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ATACOM_HIP_H */
