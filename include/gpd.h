/*
 * gpd.h — C-ABI of libgpd.so, the MI355X (gfx950) native hot path of the vectorised
 * quadrotor simulator: BaseAviary's per-drone physics step (explicit integrator + drag,
 * ground effect, downwash), DSLPIDControl, the RL action mapping, the 12-float kinematic
 * observation and the Hover / MultiHover reward, termination and truncation tests, fused
 * into one HIP kernel over a structure-of-arrays drone state.
 *
 * The reference (utiasDSL/gym-pybullet-drones, pure Python) has no FFI layer; its boundary is
 * the Python class surface.  Each entry point below names the reference code it replaces
 * (paths relative to the reference checkout, gym_pybullet_drones/...).  INTEGRATION.md shows
 * the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (PyTorch-ROCm tensors on the
 *     Python side); the library never allocates, frees or retains tensor memory;
 *   - every call is asynchronous on the hipStream_t passed as `void* stream` (NULL = default
 *     stream), takes no locks and keeps no mutable global state except the thread-local
 *     last-error string, so it is re-entrant per stream and capturable in a hipGraph;
 *   - return value: 0 on success, a negative GPD_E* code for argument errors, a positive
 *     hipError_t for runtime errors; gpd_last_error() describes the last failure of the
 *     calling thread.  No C++ exception crosses the boundary;
 *   - drone n = env * drones_per_env + d (the D drones of one aviary are adjacent);
 *   - quaternions are (x, y, z, w), as in the reference's state vector
 *     (envs/BaseAviary.py:559);
 *   - all arithmetic is fp32, no -ffast-math; FP contraction is off and every fused multiply-add is explicit in the
 *     source, so a drone's trajectory is bit-identical in every kernel variant (gpd_step vs gpd_rollout, aviary
 *     size, batch size, lane); reciprocals/square roots are the 1-ulp hardware instructions and
 *     atan2/asin/sin/cos are <= 2-ulp polynomials (csrc/gpd_common.inc).
 */
#ifndef GPD_H
#define GPD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPD_ABI_VERSION 9

/* DroneModel (utils/enums.py:3-8) */
enum { GPD_MODEL_CF2X = 0, GPD_MODEL_CF2P = 1, GPD_MODEL_RACE = 2 };

/* ActionType (utils/enums.py:35-41) + the raw, clipped RPM action of CtrlAviary
 * (envs/CtrlAviary.py:121-140) */
enum {
    GPD_ACT_RPM = 0,       /* rpm = HOVER_RPM*(1+0.05*a[0:4])      envs/BaseRLAviary.py:191-192 */
    GPD_ACT_PID = 1,       /* a[0:3] = waypoint, DSLPID            envs/BaseRLAviary.py:193-207 */
    GPD_ACT_VEL = 2,       /* a[0:4] = direction+speed, DSLPID     envs/BaseRLAviary.py:208-223 */
    GPD_ACT_ONE_D_RPM = 3, /* rpm = HOVER_RPM*(1+0.05*a[0]) x4     envs/BaseRLAviary.py:224-225 */
    GPD_ACT_ONE_D_PID = 4, /* target = pos + 0.1*[0,0,a[0]], DSLPID envs/BaseRLAviary.py:226-235 */
    GPD_ACT_RAW_RPM = 5,   /* rpm = clip(a[0:4], 0, MAX_RPM)       envs/CtrlAviary.py:140 */
    GPD_ACT_DIRECT_RPM = 6 /* rpm = a[0:4] as is: the output of a user subclass's own
                              _preprocessAction (envs/BaseAviary.py:341,1051-1064) */
};

/* Aerodynamic add-on terms evaluated inside the explicit integrator
 * (envs/BaseAviary.py:354-367 dispatch; formulas :715-811), and the ground plane.
 *
 * GPD_PHYS_GROUND is NOT part of the reference's Physics.DYN (there the drone's pose is overwritten every step,
 * envs/BaseAviary.py:865-875, so the plane loaded at :479 never acts and a drone falls through z = 0).  It stands in
 * for what Bullet's solver does for Physics.PYB* with that plane: after every physics sub-step a drone whose collision
 * cylinder (URDF: COLLISION_H, COLLISION_Z_OFFSET) would sink below z = 0 is put back ON the plane
 * (z = params.ground_z), its downward velocity is removed (restitution 0) and it sticks laterally (vx = vy = 0); body
 * rates and attitude are left to the rigid-body equations.  "Would sink": z < ground_z, or z == ground_z with vz < 0 (a
 * resting drone whose downward velocity is too small to change its fp32 height is still in contact).  The Python classes
 * enable it for Physics.PYB* only.
 *
 * GPD_PHYS_DAMP is not part of the reference's Physics.DYN either.  It is the damping every Physics.PYB* run of the reference
 * carries without a line of the reference saying so: `p.loadURDF` (envs/BaseAviary.py:488-494) creates the drone as a Bullet
 * btMultiBody, whose default linear and angular damping is 0.04 (PyBullet's `changeDynamics` defaults; the reference never
 * changes them) and acts on the base as the forces  -M v d (1 + |v|)  and  -J w d (1 + |w|)  (Bullet 3.2.x,
 * src/BulletDynamics/Featherstone/btMultiBody.cpp, computeAccelerationsArticulatedBodyAlgorithmMultiDof, "adding damping
 * terms (only)", DAMPING_K1 = DAMPING_K2 = the coefficient; third-party source, not under the reference checkout: parity
 * unpinned).  Evaluated on the velocities at the start of the sub-step, like drag.  OPT-IN on the Python side (round 5): Physics.PYB*
 * enables GPD_PHYS_GROUND by default and GPD_PHYS_DAMP only with `pyb_like="damped"` / GPD_PYB_LIKE=damped, until
 * tests/test_pybullet_optional.py has pinned it on a box with PyBullet (`pyb_like=False` opts out of the plane too: the reference's
 * explicit integrator + the add-on models). */
enum { GPD_PHYS_GND = 1, GPD_PHYS_DRAG = 2, GPD_PHYS_DW = 4, GPD_PHYS_GROUND = 8, GPD_PHYS_DAMP = 16 };
#define GPD_BULLET_DAMPING 0.04f

/* Which task's reward / termination / truncation is evaluated in the step kernel */
enum {
    GPD_TASK_NONE = 0,       /* reward -1, never terminated/truncated (envs/CtrlAviary.py:144-190) */
    GPD_TASK_HOVER = 1,      /* envs/HoverAviary.py:68-117 */
    GPD_TASK_MULTIHOVER = 2  /* envs/MultiHoverAviary.py:75-130 */
};

enum {
    GPD_EINVAL = -1,   /* bad argument (NULL pointer, non-positive size, unknown enum) */
    GPD_ERANGE = -2,   /* size outside what the kernel supports (e.g. drones_per_env > 256) */
    GPD_ENOTSUP = -3   /* combination not supported (e.g. PID action on the RACE model) */
};

/*
 * Per-airframe constants.  Replaces the attributes BaseAviary.__init__ derives from the URDF
 * (envs/BaseAviary.py:97-128, _parseURDFParameters :985-1017) and the DSLPIDControl gains
 * (control/DSLPIDControl.py:37-60, control/BaseControl.py:35-39).  Passed BY VALUE in the
 * kernel argument segment, i.e. it lives in scalar registers: every field is wave-uniform.
 */
typedef struct GpdParams {
    int32_t drone_model;       /* GPD_MODEL_* */
    float M, inv_M;            /* mass [kg] and its reciprocal */
    float L;                   /* arm length [m] */
    float KF, KM;              /* thrust / torque coefficients */
    float GRAVITY;             /* G*M with G = 9.8            envs/BaseAviary.py:117 */
    float J[3], J_INV[3];      /* diagonal inertia and its inverse */
    float prop_x[4], prop_y[4];/* rotor offsets in the body frame (URDF prop links) */
    float gnd_eff_coeff, prop_radius, gnd_eff_h_clip;   /* envs/BaseAviary.py:128 */
    float drag_coeff[3];
    float dw_coeff[3];
    float hover_rpm, max_rpm;  /* envs/BaseAviary.py:118-119 */
    /* Rotor thrusts are carried as deviations from the hover thrust F_h = GRAVITY/4 (= KF*HOVER_RPM^2
     * exactly, envs/BaseAviary.py:118): g_i = KF*rpm_i^2 - F_h evaluated without cancellation, so that
     * "thrust minus weight" and the thrust differences that make the torques keep full fp32 relative
     * accuracy near hover (host-computed in float64): */
    float hover_thrust;        /* F_h = GRAVITY/4 */
    float hover_resid;         /* KF*float(HOVER_RPM)^2 - F_h: what rounding HOVER_RPM to fp32 leaves */
    float km_over_kf;          /* KM/KF */
    /* DSLPID (the RL aviaries always build CF2X controllers, envs/BaseRLAviary.py:75-76) */
    float pid_gravity, pid_kf; /* control/BaseControl.py:35-37 */
    float pid_inv_4kf;         /* 1/(4*pid_kf), host-computed in float64 */
    float p_for[3], i_for[3], d_for[3];
    float p_tor[3], i_tor[3], d_tor[3];
    float mixer[12];           /* row-major 4x3, control/DSLPIDControl.py:47-60 */
    float pwm2rpm_scale, inv_pwm2rpm_scale, pwm2rpm_const, min_pwm, max_pwm;
    float speed_limit;         /* ActionType.VEL, envs/BaseRLAviary.py:94-95 */
    float ground_z;            /* GPD_PHYS_GROUND: height of the base link when the collision cylinder rests on the
                                  plane, COLLISION_H/2 - COLLISION_Z_OFFSET (the 0.1 m the default INIT_XYZS adds on
                                  top of it, envs/BaseAviary.py:194-197) */
} GpdParams;

/*
 * Structure-of-arrays drone state; every row is a contiguous float[N] inside a [rows][ld]
 * block (ld >= N).  Replaces BaseAviary's pos/quat/vel/rpy_rates/last_clipped_action arrays and
 * the PyBullet state store (envs/BaseAviary.py:468-477, 509-519, 865-877) and the per-drone
 * DSLPIDControl members (control/DSLPIDControl.py:65-78).
 */
typedef struct GpdState {
    float* kin;            /* 13 x ld floats in FOUR PLANES (ABI 9; ABI <= 8: thirteen rows of ld floats):
                                P  float4[ld] at kin          (pos x, pos y, pos z, body rate x)
                                Q  float4[ld] at kin + 4 ld   (quat x, y, z, w)
                                V  float4[ld] at kin + 8 ld   (vel x, vel y, vel z, body rate y)
                                W  float [ld] at kin + 12 ld  body rate z
                              (body rates = the reference's rpy_rates, envs/BaseAviary.py:474).  A lane moves its drone with three
                              16-byte accesses and one 4-byte access per direction instead of thirteen 4-byte ones.
                              ALIGNMENT: kin must be 16-byte aligned (hipMalloc / torch allocations are; a sub-allocation at a
                              4-byte offset is not) -- every entry that moves the planes returns GPD_EINVAL otherwise */
    float* last_rpm;       /* [4][ld] last applied RPMs (last_clipped_action); NULL = not tracked
                              (required with GPD_PHYS_DRAG) */
    float* pid;            /* [9][ld]: integral_pos_e | last_rpy | integral_rpy_e; NULL unless a
                              PID action type is used */
    int32_t* step_counter; /* [num_envs] physics steps since reset (envs/BaseAviary.py:460,382) */
    int64_t ld;            /* row pitch in floats, 1 .. 2^32 - 1 (the kernels take it as a 32-bit argument; GPD_EINVAL beyond) */
    float* dw_force;       /* [ld] or NULL: body-z downwash force per drone computed OUTSIDE the step kernel
                              (gpd_downwash_global, for one aviary of more than 256 drones); used with
                              GPD_PHYS_DW when drones_per_env == 1, added in every sub-step of the call */
    /* Action history (the reference's action_buffer deque, envs/BaseRLAviary.py:65-67, 153-154, 187): a DOUBLE ring
     * of the raw actions, [2*hist_len][N][A] floats, slot-major -- every action is written to slots q and q + hist_len,
     * so that the hist_len most recent actions are always hist_len CONSECUTIVE slots, oldest first, starting at
     * ring_pos: the history tail of the observation row (:317-318) is a strided VIEW of the ring ([N][hist_len][A]
     * with strides A, N*A, 1), no copy.  A slot is one contiguous [N][A] block, the shape of an action input: the
     * push is a coalesced 16-byte store per lane.  Zero-filled at start (:153-154), never cleared by a reset
     * (SURVEY.md App. B.2).  gpd_step pushes its action itself when act_ring != NULL; after a gpd_rollout call
     * gpd_full_obs pushes the K actions of the call. */
    float* act_ring;       /* [2*hist_len][N][A] or NULL */
    int32_t* ring_pos;     /* [num_envs]: slot (0 .. hist_len-1) the NEXT action of the aviary goes to = where its oldest
                              one sits.  Kept on the device so that a captured hipGraph of steps replays correctly */
    int32_t hist_len;      /* H = ctrl_freq // 2 (envs/BaseRLAviary.py:65); 0 with act_ring == NULL */
    int32_t pad_;
    /* (ABI 8) Non-finite guard, optional: every call that writes the state back (gpd_step, gpd_rollout*, gpd_rollout_policy,
     * gpd_swarm_step) also stores one byte per drone -- 1 if any of the drone's 13 kinematic floats is NaN or +-inf in the
     * state it leaves behind, else 0.  The reference has no such check (a NaN action silently poisons `pos` / `quat` for the rest
     * of the episode, envs/BaseAviary.py:831-877); a NaN position also never trips a truncation bound
     * (every comparison is false), so without the flag a poisoned aviary runs on until the episode clock ends it.  (Deviation for
     * garbage input, DSLPID action types only: a NaN set-point is absorbed by the controller's clips -- the hardware min/max return
     * the bound where numpy's clip propagates the NaN -- so the drone sees one step of saturated commands and stays finite.)  Costs
     * thirteen adds and a compare per drone and launch, outside the step loop.  The byte describes the state the call LEAVES BEHIND: an
     * aviary that went non-finite inside a K-step launch and was reset by its episode clock in the same launch (auto_reset) ends finite
     * and reads 0 -- a caller that must see every event checks between shorter launches. */
    uint8_t* bad;          /* [N] or NULL */
} GpdState;

/* Per-call configuration of gpd_step */
typedef struct GpdStepCfg {
    int32_t num_envs;        /* E */
    int32_t drones_per_env;  /* D, 1..256 */
    int32_t act_type;        /* GPD_ACT_* */
    int32_t substeps;        /* PYB_STEPS_PER_CTRL = pyb_freq/ctrl_freq  envs/BaseAviary.py:81 */
    uint32_t physics_flags;  /* GPD_PHYS_* mask */
    float pyb_dt;            /* PYB_TIMESTEP   envs/BaseAviary.py:83 */
    float ctrl_dt;           /* CTRL_TIMESTEP  envs/BaseAviary.py:82 */
    float inv_ctrl_dt;       /* CTRL_FREQ as a float (exact), multiplies where the reference divides by dt */
    int32_t lanes_per_wave;  /* 16, 32 or 64 drones per 64-lane wavefront when drones_per_env == 1 (0 = 64):
                                fewer active lanes = more wavefronts per SIMD = shorter critical path when
                                the batch is too small to fill the chip (latency-bound regime) */
    int32_t task;            /* GPD_TASK_* */
    float xy_bound, z_bound, tilt_bound;  /* truncation box  envs/HoverAviary.py:110-111 */
    float term_dist;         /* 1e-4            envs/HoverAviary.py:93 */
    int32_t trunc_counter;   /* truncated iff step_counter > trunc_counter, i.e.
                                step_counter/PYB_FREQ > EPISODE_LEN_SEC  envs/HoverAviary.py:114 */
    int32_t target_per_env;  /* 0: target_pos is [D][3] shared by all envs; 1: [E*D][3] */
    int32_t init_per_env;    /* 0: init_pose is [D][7] shared by all envs; 1: [E*D][7] */
    int32_t auto_reset;      /* 1: envs that end (terminated|truncated) are reset in the same call,
                                the SB3 DummyVecEnv convention (examples/learn.py:54-58) */
} GpdStepCfg;

/* ABI version of the loaded library (== GPD_ABI_VERSION of the header it was built from). */
int gpd_abi_version(void);

/* Description of the last error on the calling thread ("" if none). */
const char* gpd_last_error(void);

/* sizeof(GpdParams), sizeof(GpdState), sizeof(GpdStepCfg): lets a binding verify its struct
 * mirrors before the first call. */
void gpd_struct_sizes(int32_t out[3]);

/*
 * One env.step() for E aviaries of D drones.  Replaces BaseAviary.step's hot body
 * (envs/BaseAviary.py:341-382): _preprocessAction (envs/BaseRLAviary.py:160-239, incl.
 * DSLPIDControl.computeControl, control/DSLPIDControl.py:82-259), `substeps` x { _dynamics +
 * _integrateQ (:815-892) with the _groundEffect/_drag/_downwash terms (:715-811) },
 * _updateAndStoreKinematicInformation (:509-519), the kinematic part of _computeObs
 * (envs/BaseRLAviary.py:307-315), _computeReward/_computeTerminated/_computeTruncated
 * (envs/HoverAviary.py:68-117, envs/MultiHoverAviary.py:75-130) and the step-counter update.
 *
 *   action      [E*D][A] row-major, A = 4 (RPM, VEL, RAW_RPM), 3 (PID), 1 (ONE_D_*)
 *   target_pos  [D][3] or [E*D][3]: TARGET_POS of the task (ignored for GPD_TASK_NONE, may be NULL)
 *   init_pose   [D][7] or [E*D][7]: INIT_XYZS xyz + quaternion of INIT_RPYS; used by auto_reset
 *               only (may be NULL when auto_reset == 0)
 *   obs12       [E*D][12] out: pos | rpy | vel | ang_v   (envs/BaseRLAviary.py:314); with
 *               auto_reset it holds the post-reset observation of the envs that ended
 *   reward      [E] out
 *   terminated  [E] out (0/1), truncated [E] out (0/1)
 *   term_obs12  [E*D][12] out or NULL: with auto_reset, rows of envs that ended receive the last
 *               observation of the finished episode (SB3's info["terminal_observation"]); other
 *               rows are left untouched
 */
int gpd_step(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg,
             const float* action, const float* target_pos, const float* init_pose,
             float* obs12, float* reward, uint8_t* terminated, uint8_t* truncated,
             float* term_obs12, void* stream);

/*
 * gpd_step, then a wait for it on `stream`: returns when the step's results are where the kernel wrote them (and everything queued
 * on the stream before it has completed).  How it waits is the library's business: a launch whose drones fit one wavefront (up to
 * 64 drones) ends with a release store of a sequence number into a page-locked word the calling thread spins on (7.6 us for an
 * empty kernel on MI355X where launch + hipStreamSynchronize takes 12.4); every other shape, and a word that stays silent for
 * 2 ms, waits with hipStreamSynchronize.
 * The one call behind the reference-shaped single aviaries (`BaseAviary.step`, envs/BaseAviary.py:259-383 of the reference: its
 * step() returns numpy values, so every step ends in a host read): their state, action row and outputs live in page-locked HOST
 * memory mapped into the device's address space (hipHostMalloc -- every pointer of this ABI may be such a host pointer), the
 * kernel reads and writes them over the link, and the host reads the results in place -- no copy engine, no second launch, one
 * library call per step.  Same arguments and error codes as gpd_step; a failed wait returns the positive hipError_t.
 */
int gpd_step_sync(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg,
                  const float* action, const float* target_pos, const float* init_pose,
                  float* obs12, float* reward, uint8_t* terminated, uint8_t* truncated,
                  float* term_obs12, void* stream);

/*
 * K consecutive env.step() calls in ONE launch.  Replaces the caller's stepping loop around
 * BaseAviary.step when the actions of the K steps are known up front: open-loop RPM sequences, and
 * closed-loop DSLPID tracking of a precomputed waypoint list -- the loops of examples/pid.py:132-167
 * and examples/downwash.py:93-113 (`for i in range(...): obs, ... = env.step(action)` with
 * `TARGET_POS[wp_counters[j]]` as the controller's target) -- or a fixed action
 * (action_step_stride == 0, e.g. ActionType.PID holding one waypoint, examples/learn.py:157-192 with a
 * constant policy output).  The drone state is read once, kept in registers for all K steps and
 * written once; per step only the action row is read and the observation row / reward / flags are
 * written, so the per-step HBM traffic drops from ~182 B to ~70 B per drone and the ~2 us
 * launch-to-launch latency is paid once per K steps instead of once per step.  Bitwise identical to
 * K calls of gpd_step (same instructions in the same order).
 *
 *   num_steps          K >= 1
 *   actions            step t reads [E*D][A] at actions + t*action_step_stride (floats);
 *                      action_step_stride = 0 repeats one action block for all K steps
 *   obs12              step t writes [E*D][12] at obs12 + t*obs_step_stride (floats); stride 0 keeps
 *                      only the last step's observation (each step overwrites the block)
 *   reward/terminated/truncated
 *                      step t writes [E] at base + t*env_step_stride (elements); stride 0 keeps only
 *                      the last step's values
 *   term_obs12         NULL, or laid out like obs12 (same stride): with auto_reset, the rows of step t
 *                      of the envs that ended at step t receive the last observation of the episode
 *   everything else as in gpd_step.
 */
int gpd_rollout(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg,
                int32_t num_steps, const float* actions, int64_t action_step_stride,
                const float* target_pos, const float* init_pose,
                float* obs12, int64_t obs_step_stride,
                float* reward, uint8_t* terminated, uint8_t* truncated, int64_t env_step_stride,
                float* term_obs12, void* stream);

/*
 * gpd_rollout that ALSO pushes every step's raw action into the action ring of `state` (act_ring / ring_pos / hist_len), the
 * way gpd_step does: for a consumer that keeps the history as the zero-copy view (envs/BaseRLAviary.py:65-67, 153-154, 187) and
 * needs no materialised rows -- the K actions never take the detour through a second kernel (gpd_full_obs with obs_full =
 * NULL does the same after a plain gpd_rollout, re-reading them: 1.72 vs 1.29 us per step at 65 536 drones).  No terminal
 * observations; aviaries of up to 64 drones (GPD_ENOTSUP otherwise: use gpd_rollout + gpd_full_obs).
 */
int gpd_rollout_history(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, int32_t num_steps,
                        const float* actions, int64_t action_step_stride, const float* target_pos, const float* init_pose,
                        float* obs12, int64_t obs_step_stride, float* reward, uint8_t* terminated, uint8_t* truncated,
                        int64_t env_step_stride, void* stream);

/*
 * A deterministic MLP policy evaluated INSIDE the rollout kernel: the actor of Stable-Baselines3's default `MlpPolicy`
 * (features -> Linear(in_dim, 64) -> tanh -> Linear(64, 64) -> tanh -> Linear(64, act_dim), `model.predict(obs,
 * deterministic=True)` incl. its clip to the action space [-1, 1]) -- what examples/learn.py:157-192 evaluates between two
 * env.step() calls.  Weights are float32 device arrays in torch's Linear layout (weight [out][in] row-major, bias [out]).
 *   in_dim = 12                       the policy sees the kinematic observation only
 *   in_dim = 12 + hist_len*act_dim    the policy sees the reference's full row (BaseRLAviary._computeObs): kinematics, then the
 *                                     hist_len most recent actions, oldest first (state->act_ring must be set)
 * activation: 0 = tanh (SB3's default), 1 = ReLU.
 */
typedef struct GpdPolicy {
    const float* w1; const float* b1;   /* [64][in_dim], [64] */
    const float* w2; const float* b2;   /* [64][64], [64] */
    const float* w3; const float* b3;   /* [act_dim][64], [act_dim] */
    int32_t in_dim;
    int32_t hidden;                     /* 64 */
    int32_t activation;                 /* 0 tanh, 1 relu */
    int32_t pad_;
} GpdPolicy;

/*
 * K consecutive env.step() calls in ONE launch with the policy IN the loop.  Replaces the evaluation / sampling loop of
 * examples/learn.py:157-192 (`action, _ = model.predict(obs, deterministic=True); obs, reward, terminated, truncated, info =
 * env.step(action)`), for single-drone aviaries (HoverAviary) and all five ActionTypes (for PID / VEL / ONE_D_PID the policy's
 * output is the set-point of the embedded DSLPID controller, which runs inside the same step):
 *     a_t = clip(W3 act(W2 act(W1 o_t + b1) + b2) + b3, -1, 1);   o_{t+1}, r_t, ... = step(a_t)
 * where o_t is the latest observation row -- of the reset pose when the aviary was reset in step t-1 (same-step auto-reset, the
 * history tail survives resets like the reference's action buffer).  The drone state stays in registers for all K steps like in
 * gpd_rollout; the two 64-unit layers run on the matrix cores: the 64 drones of a wavefront are the columns of
 * v_mfma_f32_32x32x16_bf16 tiles, every operand split into bf16 hi + lo parts (three products per tile: hi*hi + hi*lo + lo*hi,
 * fp32 accumulate), which keeps ~16 mantissa bits of the float32 weights and observations (actions agree with a float64
 * evaluation to ~1e-5); a layer's output tile is the next layer's B operand without leaving registers.
 *
 *   obs12_in     [E][12] the latest observation rows (the obs12 output of the previous step / reset / rollout)
 *   actions_out  [K][E][A] out or NULL: the actions the policy chose, step t at actions_out + t*E*A
 *   obs12 / reward / terminated / truncated and the strides: as in gpd_rollout
 *   noise, action_std, mean_out   training rollouts (all NULL: the deterministic policy above).  The collection loop of
 *                Stable-Baselines3's PPO (`model.learn()`, examples/learn.py:61-95: DiagGaussianDistribution with a
 *                state-independent log_std, actions clipped to the Box before env.step):
 *                    a_t = clip(mean_t + action_std[k] * noise[t][e][k], -1, 1)
 *                noise [K][E][A]: standard-normal draws of the caller (device memory); action_std [A]: exp(log_std), HOST
 *                memory; mean_out [K][E][A] out or NULL: the unclipped means.  actions_out and the action ring receive the
 *                clipped actions (what the environment saw); the log-probability of the unclipped sample is a function of
 *                noise and action_std alone.  ActionType.RPM and ONE_D_RPM.
 * With in_dim > 12 the action ring of `state` is read at the start, and EVERY step pushes its (clipped) action into both halves
 * of the ring and advances ring_pos by one (mod hist_len), exactly as gpd_step does: after the call the ring and ring_pos are what
 * num_steps gpd_step calls with the same actions would have left (history() / gpd_hist_rows / a following gpd_step continue
 * seamlessly).
 *   term_obs12   (ABI 8) NULL, or laid out like obs12 (same stride): with auto_reset, the rows of step t of the aviaries that ended
 *                at step t receive the last observation of the finished episode -- SB3's info["terminal_observation"], which its
 *                PPO bootstraps time-outs from (examples/learn.py:61-95 through `model.learn()`); other rows are left untouched.
 * GPD_ENOTSUP: drones_per_env > 1, hidden != 64, in_dim not one of the two forms, or a history longer than 17 actions of 4 or
 * 3 floats / 20 actions of 1 float (the reference's 30 Hz control: 15).
 */
int gpd_rollout_policy(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const GpdPolicy* policy,
                       int32_t num_steps, const float* obs12_in, const float* target_pos, const float* init_pose,
                       float* actions_out, float* obs12, int64_t obs_step_stride, float* reward, uint8_t* terminated,
                       uint8_t* truncated, int64_t env_step_stride, const float* noise, const float* action_std, float* mean_out,
                       float* term_obs12, void* stream);

/*
 * Full KIN observation rows with the action-history tail.  Replaces the row assembly of BaseRLAviary._computeObs
 * (envs/BaseRLAviary.py:307-320) on top of the action ring of GpdState (which replaces the deque, :65-67, 153-154, 187):
 *     row = [ pos | rpy | vel | ang_v | a(t-H+1) ... a(t) ]      H = ctrl_freq // 2, oldest action first
 * The rows are MATERIALISED only on request -- at 240 Hz control a row is 12 + 120*4 floats, ten times the bytes of the
 * whole physics step; a consumer that can take (obs12, history view) separately never needs them.
 *
 * gpd_hist_rows: the current rows, after a gpd_step call (which pushed its action into the ring itself):
 *   state       act_ring / ring_pos / hist_len are read (ring_pos of aviary n / drones_per_env)
 *   obs12       [n_drones][12] of the latest step
 *   obs_full    [n_drones][12 + hist_len*act_dim] out
 *
 * gpd_full_obs: the rows of the num_steps steps of a gpd_rollout call (call it after the rollout, same stream), built
 * from the ring as the rollout found it plus the call's own action blocks; then the K actions are pushed into the ring
 * and ring_pos advances by num_steps (mod hist_len).
 *   obs12         the rollout's observation blocks, step t at obs12 + t*obs_step_stride
 *   actions       the rollout's action blocks,      step t at actions + t*action_step_stride (0: held action)
 *   obs_full      [num_steps][n_drones][12 + hist_len*act_dim] out, step t at obs_full + t*full_step_stride;
 *                 NULL: only update the ring
 */
int gpd_hist_rows(const GpdState* state, int32_t n_drones, int32_t drones_per_env, int32_t act_dim, const float* obs12,
                  float* obs_full, void* stream);
int gpd_full_obs(const GpdState* state, int32_t num_steps, int32_t n_drones, int32_t drones_per_env, int32_t act_dim,
                 const float* obs12, int64_t obs_step_stride, const float* actions, int64_t action_step_stride,
                 float* obs_full, int64_t full_step_stride, void* stream);

/*
 * Downwash forces inside ONE aviary of n drones, any n (the step kernel itself handles aviaries of up to 256
 * drones).  Replaces the O(n^2) Python loop of BaseAviary._downwash (envs/BaseAviary.py:785-811) over all pairs
 * with  dz = z_j - z_i > 0  and  dxy < 10 m:
 *     F_i = - sum_j  DW1 * (PROP_RADIUS / (4 dz))^2 * exp(-0.5 * (dxy / (DW2*dz + DW3))^2)        (body z of drone i)
 * by binning the drones into a uniform 2-D grid of `cell` >= 10 m squares (counting sort by cell) and searching
 * the 3x3 neighbourhood of each drone's cell.  The grid is periodic: a drone outside the box the grid was laid over
 * lands in the cell its coordinates wrap to (still exact: every candidate pair is distance-tested, aliased far-apart
 * drones are rejected; the swarm may spread without piling up in border cells).  The per-drone sum is accumulated
 * in 64-bit fixed point (2^-30 N), so the result does not depend on the order the sort leaves the neighbours in; a pair
 * whose exponent 0.5 (dxy / (DW2*dz + DW3))^2 reaches 40 is dropped (it is below e^-40 = 4.3e-18 of its own amplitude).
 *
 *   kin, ld       the state block of gpd_step (plane P = positions is read: float4 n at kin + 4 n)
 *   cell, x0, y0, nx, ny   grid: cell size [m] (>= 10), lower-left corner, cells per side (nx, ny >= 3)
 *   z0, zbin, nz  every cell is split into nz height bins of zbin metres starting at z0 (bin 0 also holds everything below z0,
 *                 bin nz-1 everything above; nz = 1: no bins, z0/zbin ignored); nx*ny*nz <= 65536.  The sort key is
 *                 cell*nz + bin, so inside a cell the drones are ordered by height bin.  Ordering only: any (z0, zbin, nz)
 *                 gives the same forces bit for bit (the force kernel of ABI 4's first builds pruned by bin; the present
 *                 one does not, nz = 1 is the sensible choice).
 *   visit_order   [n] int32 permutation of 0..n-1 or NULL (= 0, 1, 2 ...): the order the sort visits the drones in.
 *                 Any permutation gives the same forces bit for bit; handing in the `order` buffer the PREVIOUS call
 *                 filled (two buffers, ping-pong: it must not alias `order`) makes neighbouring lanes share a cell, and
 *                 the sort then issues one atomic per run of equal cells instead of one per drone.
 *   cell_count    [2 (nx*ny*nz + 1)] int32 (per-key counts | per-key cursors of the sort): ZERO before the first call; every
 *                 call leaves it zeroed again (the last kernel clears it: no memset per call)
 *   cell_start    [nx*ny*nz + 1] int32 scratch
 *   order         [n] int32 out (drone index of sorted slot: a permutation of 0..n-1)
 *   sorted_xyzc   [n][4] float scratch (x, y, z, sort key as int bits), sorted by key
 *   dw_out        [n] out: the force of drone i at dw_out[i]  (pass it to gpd_step as state.dw_force)
 *   vec_state, vec_obs12, vec_out   optional (vec_out NULL: none): the sort's first pass also writes the [n][20] state vectors
 *                 of gpd_state_vectors(vec_state, vec_obs12, vec_out, n) -- a caller that computes the forces for the NEXT
 *                 sub-step right after a gpd_step needs both, and saves a launch
 * Call it once per physics sub-step, before the gpd_step launch of that sub-step (positions are the snapshot every
 * drone sees, envs/BaseAviary.py:346-347).
 */
int gpd_downwash_global(const GpdParams* params, const float* kin, int64_t ld, int32_t n, float cell, float x0,
                        float y0, int32_t nx, int32_t ny, float z0, float zbin, int32_t nz, const int32_t* visit_order,
                        int32_t* cell_count, int32_t* cell_start, int32_t* order, float* sorted_xyzc, float* dw_out,
                        const GpdState* vec_state, const float* vec_obs12, float* vec_out, void* stream);

/*
 * ONE aviary of any size, stepped by one or several ranks (GPUs): the persistent form of gpd_downwash_global.
 *
 * What changes against calling gpd_downwash_global before every gpd_step (count + scan/scatter + force + step = four
 * dependent launches per physics sub-step):
 *   - the physics sub-step kernel (gpd_swarm_step: gpd_step's arithmetic for single-drone lanes) ALSO writes the drone's new
 *     position into a packed [rows][4] array, tracks how far it has moved since the drones were last binned, and (on request)
 *     writes the drone's 20-float state vector: no separate pass over the drones after a step;
 *   - the counting sort (gpd_swarm_bin) runs every few sub-steps only.  Between two binnings the force kernel
 *     (gpd_swarm_forces) works on the STALE cell order but on CURRENT positions (slot -> row -> pos4), and widens its search
 *     from 3x3 cells to (2R+1)x(2R+1), R = ceil((10 m + 2 dmax) / cell), dmax = the largest displacement of any
 *     drone since the binning: two drones within 10 m of each other now were within 10 m + 2 dmax then, i.e. at most R cells
 *     apart.  With cell = 10 m + a skin, R stays 1 until some drone has moved half the skin; whatever the drones do, every
 *     pair the reference would sum (dz > 0, dxy < 10 m; envs/BaseAviary.py:785-811) is evaluated, on current positions, in
 *     order-independent 64-bit fixed point: the forces are the ones gpd_downwash_global returns, bit for bit.  (R beyond 3,
 *     or beyond the grid: the group sweeps every drone.)  A sub-step is then TWO launches: gpd_swarm_step, gpd_swarm_forces;
 *   - several ranks share one world: rank r owns the rows [r*slab, (r+1)*slab) -- its own_count drones, then rows without a
 *     drone (non-finite x), the LAST meta_rows rows of the slab being the rank's meta rows (NaN, sum dx, sum dy, w): workgroup b
 *     of gpd_swarm_step stores the largest squared displacement of its 256 drones in the w of meta row b (and their summed
 *     lateral displacements in y, z) -- plain stores, the force kernel takes the maximum.  After
 *     gpd_swarm_step every rank all-gathers its slab IN PLACE (gpd_allgather_obs(comm, pos4 + rank*slab*4, pos4, slab*4):
 *     16 bytes per drone, the one collective per sub-step), bins ALL rows when a binning is due (every rank the same
 *     sub-steps), and evaluates the forces of ITS drones only (workgroups of the sorted array without a drone of the rank
 *     exit at once).  Sums are integers: a world stepped by 1, 2 or 8 ranks follows the same trajectory bit for bit.
 *
 * The force of own drone i lands in dw_force[i]: pass it to gpd_swarm_step as state.dw_force (GPD_PHYS_DW).
 */
typedef struct GpdSwarm {
    int32_t n_rows;        /* rows of pos4 = world_size * slab */
    int32_t slab;          /* rows per rank (>= own_count + meta_rows: the last meta_rows are its meta rows) */
    int32_t world_size, rank;
    int32_t own_count;     /* drones of this rank: rows rank*slab .. rank*slab + own_count - 1 */
    int32_t nx, ny, nz;    /* grid: cells per side (>= 3 each), height bins per cell (1: none); nx*ny*nz <= 65536 */
    float cell, x0, y0;    /* cell size [m] (>= 10), lower-left corner of the (periodic) grid */
    float z0, zbin;        /* height bins as in gpd_downwash_global (ordering only) */
    int32_t meta_rows;     /* >= ceil(max own_count / 256), the same on every rank: one row per workgroup of gpd_swarm_step */
    float* pos4;           /* [n_rows][4] x, y, z, - of every row (meta rows: w = a partial dmax^2) */
    float* bin_pos;        /* [n_rows][4] x, y, z, - of every row at the last binning */
    int32_t* cell_count;   /* [2 (nx*ny*nz + 1)] counts | cursors: ZERO before the first gpd_swarm_bin.  gpd_swarm_bin leaves the counts
                              and cursors of its sort in it; the gpd_swarm_forces call that must follow every binning (build_lists
                              or not) clears them again -- no memset per call.  Two gpd_swarm_bin calls WITHOUT a gpd_swarm_forces
                              in between (re-pack + re-bin) need the caller to zero the array itself: the second sort would
                              otherwise start from the first one's counts and scatter out of bounds */
    int32_t* cell_start;   /* [nx*ny*nz + 1] */
    int32_t* order;        /* [n_rows] sorted slot -> row, written by gpd_swarm_bin, read by gpd_swarm_forces: ONE buffer, always that
                              of the latest binning (a captured hipGraph of sub-steps replays correctly whatever ran in between) */
    const int32_t* visit;  /* [n_rows] or NULL: the order gpd_swarm_bin visits the rows in -- any permutation gives the same forces;
                              the previous binning's order makes neighbouring lanes share a cell (one atomic per run of equal
                              cells).  Must alias neither `order` nor `visit_out` ... */
    int32_t* visit_out;    /* ... [n_rows] or NULL: gpd_swarm_bin writes a second copy of `order` here -- hand it in as `visit` next
                              time and the old `visit` buffer as `visit_out` (ping-pong; which of the two holds the newer copy
                              does not matter for correctness) */
    int32_t* slot_key;     /* [n_rows] sort key (cell*nz + bin) of every sorted slot */
    float* dw_force;       /* [>= own_count] out: body-z downwash force of own drone i (envs/BaseAviary.py:805-811) */
    /* world_size == 1 only (both NULL otherwise): the positions ALSO by sorted slot, kept current by gpd_swarm_step, so that the
     * force kernel reads its candidates as contiguous stretches instead of through order[] (a rank that holds the whole world
     * needs no exchange in row order) */
    int32_t* slot_of;      /* [n_rows] row -> sorted slot (-1: no finite position), written by gpd_swarm_bin */
    float* pos_sorted;     /* [n_rows][4] */
    /* Wake lists (optional: pair_list NULL = none).  The force launch right after a binning (build_lists != 0) keeps every pair
     * that could pass the model's tests once both drones have moved up to list_delta, and writes the pairs it evaluates to
     * pair_list; the launches until the next binning read them back instead of sweeping all candidates -- as long as no drone is
     * further than list_delta from where it was binned (otherwise they sweep as if there were no lists: still exact).
     * list_delta: just under (cell - 10 m) / 2, e.g. 0.49 (cell - 10). */
    uint32_t* pair_list;   /* [ceil(n_rows / 64)][4][list_cap * 64]: (ABI 8) 32-bit entries -- 6 bits drone of the group, 26 bits the
                              candidate's index in the array its position is read from (sorted slot / row of pos4): a replay launch
                              gathers the candidates' current positions directly, no cell table, no staged tile */
    uint16_t* pair_nb;     /* [ceil(n_rows / 64)][4][16]: [0] = batches the wavefront recorded */
    int32_t* list_ok;      /* [ceil(n_rows / 64)] */
    int32_t list_cap;      /* batches of 64 pairs per wavefront (a group of 64 drones has four), 4 .. 65535; a group that needs more sweeps */
    float list_delta;
    /* "Displacement" above is measured relative to the swarm's COMMON lateral drift since the binning (a translation all drones
     * share changes no pair: a swarm in transit keeps R = 1 and its wake lists).  gpd_swarm_forces computes the drift -- the mean
     * lateral displacement of all drones, from per-workgroup sums in the meta rows (y, z) -- for the next gpd_swarm_step. */
    float* drift;          /* [4] device floats, zero before the first call: [0..1] the drift (gpd_swarm_bin zeroes them), [2] the
                              margin of the wake lists of the current binning (gpd_swarm_bin sets it, below), [3] reserved */
    int32_t total_drones;  /* drones of the whole world (all ranks) */
    int32_t list_adapt;    /* (ABI 7) 0: the lists' margin is list_delta.  1: gpd_swarm_bin chooses it for each binning from the
                              displacement the interval that ends there has seen -- min(list_delta, max(1 cm, 3 dmax)): a swarm
                              that hovers lists hardly more pairs than the exact tests keep (at cell = 10.5 m the full margin
                              lists 40 % more); one that moves further than expected sweeps until the next binning.  Exact
                              either way. */
} GpdSwarm;

/* One physics sub-step of the rank's own_count drones (state / cfg as for gpd_step: drones_per_env = 1, num_envs = own_count,
 * substeps = 1, task NONE, no auto-reset; act_type RPM, RAW_RPM or DIRECT_RPM; state.dw_force = swarm.dw_force with
 * GPD_PHYS_DW), plus: pos4 rows of the rank, the rank's dmax^2, and -- vec_out != NULL -- the [own_count][20] state vectors
 * of gpd_state_vectors.  Replaces the body of BaseAviary.step's sub-step loop for one world (envs/BaseAviary.py:346-372).
 * When world_size * meta_rows > 1024 the call also leaves the rank's largest dmax^2 in the w of the rank's FIRST meta row (a second,
 * one-workgroup launch): gpd_swarm_forces then reads one value per rank instead of every meta row. */
/* sizeof(GpdSwarm), for a binding to verify its mirror (gpd_struct_sizes covers the three structs of ABI 1). */
int gpd_sizeof_swarm(void);

int gpd_swarm_step(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const GpdSwarm* swarm,
                   const float* action, float* obs12, float* vec_out, void* stream);
/* After a reset / any outside change of the state: pos4 rows of the rank (and its drone-less rows and meta row) from
 * state.kin, dmax^2 = 0, optionally the state vectors.  Follow it with the all-gather, gpd_swarm_bin and gpd_swarm_forces. */
int gpd_swarm_pack(const GpdState* state, const GpdSwarm* swarm, const float* obs12, float* vec_out, void* stream);
/* Counting sort of ALL rows by grid cell from pos4 (rows with a non-finite position take no part): order, slot_key,
 * cell_start, bin_pos (cell_count must be zero on entry: see GpdSwarm -- every binning is followed by gpd_swarm_forces); every rank's dmax^2 and displacement sums (in this rank's copy of pos4) and drift[0..1] back to 0;
 * drift[2] = the margin of the wake lists of this binning (list_adapt: from the largest dmax^2 found there before the reset). */
int gpd_swarm_bin(const GpdSwarm* swarm, void* stream);
/* Downwash forces of the rank's drones for the positions in pos4 -> dw_force, and drift[0..1] for the next gpd_swarm_step.
 * build_lists: non-zero on the call that follows a gpd_swarm_bin (ignored without wake lists). */
int gpd_swarm_forces(const GpdParams* params, const GpdSwarm* swarm, int32_t build_lists, void* stream);

/*
 * Masked reset.  Replaces BaseAviary.reset/_housekeeping (envs/BaseAviary.py:220-255, 451-477)
 * for the envs whose mask byte is non-zero (mask == NULL: all).  Sets pos/quat to init_pose,
 * vel, rpy_rates, last_rpm and step_counter to zero and writes the initial obs12 rows.  As in
 * the reference, the DSLPID state is NOT reset unless reset_pid != 0.
 */
int gpd_reset(const GpdState* state, const float* init_pose, int32_t init_per_env,
              const uint8_t* mask, int32_t num_envs, int32_t drones_per_env, int32_t reset_pid,
              float* obs12, void* stream);

/*
 * Batched DSLPIDControl.computeControl (control/DSLPIDControl.py:82-145) for n independent
 * controllers.  Inputs are row-major [n][3] / [n][4] arrays; `pid` is the [9][ld] controller
 * state (updated in place).  target_rpy / target_vel / target_rpy_rates may be NULL (= zeros,
 * the reference's defaults).  Outputs: rpm [n][4], pos_e [n][3] (may be NULL), yaw_e [n]
 * (may be NULL; computed_target_rpy[2] - cur_rpy[2]).
 */
int gpd_pid(const GpdParams* params, float* pid, int64_t ld, float ctrl_dt,
            const float* cur_pos, const float* cur_quat, const float* cur_vel,
            const float* target_pos, const float* target_rpy, const float* target_vel,
            const float* target_rpy_rates, float* rpm, float* pos_e, float* yaw_e,
            int32_t n, void* stream);

/*
 * gpd_pid, then a wait for it on `stream` (as gpd_step_sync waits: up to 64 controllers report through a page-locked word the
 * calling thread spins on, more take hipStreamSynchronize): the call behind the reference-shaped single-drone
 * `DSLPIDControl.computeControl` (control/DSLPIDControl.py:82-145 returns numpy values: every call ends in a host read), whose
 * operands and controller state live in page-locked host memory.  Same arguments and error codes as gpd_pid.
 */
int gpd_pid_sync(const GpdParams* params, float* pid, int64_t ld, float ctrl_dt,
                 const float* cur_pos, const float* cur_quat, const float* cur_vel,
                 const float* target_pos, const float* target_rpy, const float* target_vel,
                 const float* target_rpy_rates, float* rpm, float* pos_e, float* yaw_e,
                 int32_t n, void* stream);

/*
 * Gather the 20-float state vectors of BaseAviary._getDroneStateVector
 * (envs/BaseAviary.py:541-561): pos3 | quat4 | rpy3 | vel3 | ang_v3 | last_clipped_action4,
 * from the SoA state and the obs12 rows of the latest step.  state20 is [n][20].
 */
int gpd_state_vectors(const GpdState* state, const float* obs12, float* state20, int32_t n,
                      void* stream);

/*
 * The one optional exchange step of the multi-GPU layout: all-gather of the observation shards over RCCL (xGMI).
 * The aviaries shard across ranks with no data-path collective (one process per GPU, rank r owns a contiguous block of
 * envs); a centralised learner that wants the concatenated (E_total*D, 12) observation tensor -- what the reference's
 * single process holds in `obs` after `env.step()` (envs/BaseAviary.py:375, envs/BaseRLAviary.py:307-320; consumed by
 * examples/learn.py:157-192) -- gathers it with ONE collective per step (or per rollout: fewer, larger collectives).
 * RCCL is resolved at run time (dlopen of librccl.so.1; GPD_RCCL_LIB overrides): a single-GPU consumer needs no RCCL,
 * and GPD_ENOTSUP is returned when it cannot be found.
 *
 *   gpd_comm_unique_id   rank 0: ncclGetUniqueId; the caller distributes the GPD_COMM_ID_BYTES to the other ranks
 *                        (torch.distributed broadcast, MPI, a file ...)
 *   gpd_comm_init        every rank, after hipSetDevice: ncclCommInitRank (collective, blocks until all ranks joined)
 *   gpd_allgather_obs    ncclAllGather of `count` floats per rank: shard [count] -> full [world_size*count], rank r's
 *                        shard at full + r*count.  Asynchronous on `stream`, capturable in a hipGraph together with the
 *                        gpd_step / gpd_rollout launch that produced the shard.
 *   gpd_comm_count       ncclCommCount: the number of ranks RCCL itself says the communicator spans (a harness prints it
 *                        next to its throughput line: a run that silently fell apart into one-rank worlds shows here)
 *   gpd_comm_destroy     ncclCommDestroy (NULL is a no-op)
 * ONE communicator per process serves every count: create it once, pass any `count` to gpd_allgather_obs.
 * RCCL errors are returned as 1000 + ncclResult_t.
 */
#define GPD_COMM_ID_BYTES 128
int gpd_comm_unique_id(uint8_t id[GPD_COMM_ID_BYTES]);
int gpd_comm_init(void** comm, const uint8_t id[GPD_COMM_ID_BYTES], int32_t rank, int32_t world_size);
int gpd_comm_count(void* comm, int32_t* n_ranks);
int gpd_comm_destroy(void* comm);
int gpd_allgather_obs(void* comm, const float* shard, float* full, size_t count, void* stream);

/*
 * (ABI 8) Point-to-point exchange of float blocks between the ranks of a communicator, all operations of one call in ONE RCCL
 * group (ncclGroupStart, ncclSend / ncclRecv per block, ncclGroupEnd): the HALO exchange of a world shared by several ranks.
 * The reference's `_downwash` reads every other drone's position from the process's own arrays (envs/BaseAviary.py:798-811,
 * `self.pos[j]`); a rank of a sharded world needs the positions of the drones within the model's 10 m lateral cut-off (plus
 * a margin) of its own -- with the ranks holding stripes of the world, blocks from its two neighbours instead of the
 * all-gather of every position (gpd_allgather_obs in place; 16 B x every drone per sub-step).  xGMI is point to point: every
 * block travels over the one link between its two GPUs.
 *   sends / recvs   the operations: `count` floats at `ptr` to / from rank `peer`.  Several blocks between the same pair of
 *                   ranks are matched in the order they are listed (RCCL's rule); every send needs its receive on the peer,
 *                   in a call of its own made at the same point of the program.  Device pointers; asynchronous on `stream`,
 *                   capturable in a hipGraph.
 * GPD_ENOTSUP when RCCL (or its point-to-point entries) cannot be resolved.
 */
typedef struct GpdP2P {
    int32_t peer;          /* rank of the other side */
    int32_t pad_;
    void* ptr;             /* device pointer (send: read, receive: written) */
    int64_t count;         /* floats */
} GpdP2P;
int gpd_p2p_group(void* comm, const GpdP2P* sends, int32_t n_sends, const GpdP2P* recvs, int32_t n_recvs, void* stream);

/*
 * (ABI 8) Debug-bounds build.  A library compiled with -DGPD_DEBUG_BOUNDS (`python -c "from gym_pybullet_drones_amd import _native;
 * _native.build(debug=True)"` -> csrc/libgpd_debug.so; select it with GPD_LIB) checks every index its kernels READ FROM MEMORY before
 * they address with it, records the first violation and clamps the index, so that a corrupted ring position or wake list shows up
 * as a code instead of as an out-of-bounds access.  The reference has no counterpart (no sanitizer, no bounds
 * or race checks beyond numpy's own IndexError).  gpd_debug_status waits for `stream` and returns the record:
 *     out[0] code of the first violation (0: none), out[1] its workgroup, out[2] the offending value, out[3] violations so far;
 * reset != 0 clears the record.  A release build returns GPD_ENOTSUP (its kernels carry no checks).
 */
enum {
    GPD_DBG_RING_POS = 1,      /* GpdState.ring_pos outside [0, hist_len)          (gpd_step, gpd_rollout_history) */
    GPD_DBG_STEP_COUNTER = 2,  /* GpdState.step_counter negative */
    GPD_DBG_SLOT_ROW = 3,      /* GpdSwarm.order: sorted slot -> row outside [0, n_rows)   (gpd_swarm_forces) */
    GPD_DBG_SORT_KEY = 4,      /* GpdSwarm.slot_key outside the grid's keys */
    GPD_DBG_LIST_COUNT = 5,    /* GpdSwarm.pair_nb: more batches than list_cap */
    GPD_DBG_LIST_ENTRY = 6     /* GpdSwarm.pair_list: a candidate index outside [0, n_rows) */
};
int gpd_debug_status(uint32_t out[4], int32_t reset, void* stream);

/*
 * Diagnostics for the measurement harness (bench.py's issue roofline): runs a dependent v_fma_f32 chain at one wave per
 * SIMD and reports the shader clock it ran at [GHz] (shader-clock cycles / constant-rate wall-clock time) and, optionally,
 * the time per dependent FMA [ns].  Synchronous (it waits for `stream`).
 */
int gpd_clock_probe(double* shader_ghz, double* ns_per_fma, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPD_H */
