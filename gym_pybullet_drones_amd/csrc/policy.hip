// policy.hip -- gpd_rollout_policy
#include "gpd_common.inc"
#include "policy_kernel.inc"

#ifdef GPD_PID_POLICY_IN_POLICY_TU
void gpd_detail_launch_policy_pid(const GpdPolicyLaunch& a) {
    const Span& T = *static_cast<const Span*>(a.span);
    const dim3 grid(a.grid);
    hipStream_t st = static_cast<hipStream_t>(a.stream);
#define GPD_POL(AW_, ACT_, NK1_)                                                                                                   \
    do {                                                                                                                            \
        if (a.policy->activation == 1)                                                                                              \
            hipLaunchKernelGGL((gpd_rollout_policy_kernel<true, AW_, ACT_, NK1_, true>), grid, dim3(kBlock), 0, st, *a.params, *a.state, *a.cfg, T, \
                               *a.policy, a.obs12_in, a.target_pos, a.init_pose, a.actions_out, a.obs12, a.reward, a.terminated, a.truncated, a.term_obs12); \
        else                                                                                                                        \
            hipLaunchKernelGGL((gpd_rollout_policy_kernel<true, AW_, ACT_, NK1_, false>), grid, dim3(kBlock), 0, st, *a.params, *a.state, *a.cfg, T, \
                               *a.policy, a.obs12_in, a.target_pos, a.init_pose, a.actions_out, a.obs12, a.reward, a.terminated, a.truncated, a.term_obs12); \
    } while (0)
    switch (a.cfg->act_type) {
        case GPD_ACT_VEL: if (a.hist) GPD_POL(4, GPD_ACT_VEL, 5); else GPD_POL(4, GPD_ACT_VEL, 1); break;
        case GPD_ACT_PID: if (a.hist) GPD_POL(3, GPD_ACT_PID, 4); else GPD_POL(3, GPD_ACT_PID, 1); break;
        default: if (a.hist) GPD_POL(1, GPD_ACT_ONE_D_PID, 2); else GPD_POL(1, GPD_ACT_ONE_D_PID, 1); break;
    }
#undef GPD_POL
#undef GPD_POLN
}
#endif

GPD_DBG_READER(gpd_detail_dbg_read_policy)

extern "C" {

int gpd_rollout_policy(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const GpdPolicy* policy,
                       int32_t num_steps, const float* obs12_in, const float* target_pos, const float* init_pose,
                       float* actions_out, float* obs12, int64_t obs_step_stride, float* reward, uint8_t* terminated,
                       uint8_t* truncated, int64_t env_step_stride, const float* noise, const float* action_std, float* mean_out,
                       float* term_obs12, void* stream) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string("gpd_rollout_policy: ") + msg).c_str()); };
    if ((noise != nullptr) != (action_std != nullptr)) return bad(GPD_EINVAL, "noise and action_std come together");
    if (mean_out && !noise) return bad(GPD_EINVAL, "mean_out is written by the sampling kernels only (pass noise)");
    if (!params || !state || !cfg || !policy) return bad(GPD_EINVAL, "NULL params/state/cfg/policy");
    if (!state->kin || !state->step_counter) return bad(GPD_EINVAL, "NULL state.kin/step_counter");
    if (const char* why = state_layout_problem(state)) return bad(GPD_EINVAL, why);
    if (!obs12_in || !obs12 || !reward || !terminated || !truncated) return bad(GPD_EINVAL, "NULL obs12_in/obs12/reward/terminated/truncated");
    if (!policy->w1 || !policy->b1 || !policy->w2 || !policy->b2 || !policy->w3 || !policy->b3) return bad(GPD_EINVAL, "NULL policy weights");
    if (num_steps <= 0 || obs_step_stride < 0 || env_step_stride < 0) return bad(GPD_EINVAL, "num_steps must be positive, strides non-negative");
    if (cfg->num_envs <= 0 || cfg->substeps <= 0) return bad(GPD_EINVAL, "num_envs and substeps must be positive");
    if (cfg->drones_per_env != 1) return bad(GPD_ENOTSUP, "single-drone aviaries only (drones_per_env == 1)");
    if (cfg->act_type < GPD_ACT_RPM || cfg->act_type > GPD_ACT_ONE_D_PID) return bad(GPD_ENOTSUP, "one of the five ActionTypes (RPM, PID, VEL, ONE_D_RPM, ONE_D_PID)");
    const bool pid = cfg->act_type == GPD_ACT_PID || cfg->act_type == GPD_ACT_VEL || cfg->act_type == GPD_ACT_ONE_D_PID;
    if (pid && !state->pid) return bad(GPD_EINVAL, "PID action type needs state.pid");
    if (pid && params->pid_kf <= 0.0f) return bad(GPD_ENOTSUP, "no DSLPID controller for this airframe (CF2X/CF2P only)");
    if (cfg->task < GPD_TASK_NONE || cfg->task > GPD_TASK_MULTIHOVER) return bad(GPD_EINVAL, "unknown task");
    if (cfg->physics_flags & ~31u) return bad(GPD_EINVAL, "unknown physics flag");
    if (policy->hidden != kPolHidden) return bad(GPD_ENOTSUP, "hidden must be 64");
    if (policy->activation != 0 && policy->activation != 1) return bad(GPD_EINVAL, "activation must be 0 (tanh) or 1 (relu)");
    const int64_t N = cfg->num_envs;
    if (state->ld < N) return bad(GPD_EINVAL, "state.ld < num_envs");
    if (N > (1LL << 26)) return bad(GPD_ERANGE, "more than 2^26 drones per launch");
    if ((cfg->physics_flags & GPD_PHYS_DRAG) && !state->last_rpm) return bad(GPD_EINVAL, "GPD_PHYS_DRAG needs state.last_rpm");
    if (cfg->task != GPD_TASK_NONE && !target_pos) return bad(GPD_EINVAL, "task needs target_pos");
    if (cfg->auto_reset && !init_pose) return bad(GPD_EINVAL, "auto_reset needs init_pose");
    const int A = (cfg->act_type == GPD_ACT_RPM || cfg->act_type == GPD_ACT_VEL) ? 4 : (cfg->act_type == GPD_ACT_PID ? 3 : 1);
    const int cap = A == 4 ? 68 : (A == 3 ? 52 : 20);                 // history features the kernel's registers hold (16*NK1 - 12)
    const bool hist = policy->in_dim != 12;
    if (hist) {
        if (!state->act_ring || !state->ring_pos || state->hist_len <= 0) return bad(GPD_EINVAL, "in_dim > 12 needs the action ring of state");
        if (policy->in_dim != 12 + state->hist_len * A) return bad(GPD_ENOTSUP, "in_dim must be 12 or 12 + hist_len*act_dim");
        if (state->hist_len * A > cap) return bad(GPD_ENOTSUP, "history too long for the in-kernel policy (17 actions of 4 or 3 floats, 20 of 1)");
    } else if (state->act_ring && (!state->ring_pos || state->hist_len <= 0)) {
        return bad(GPD_EINVAL, "state.act_ring without ring_pos / hist_len");
    }
    GpdStepCfg c = *cfg;
    if (cfg->task == GPD_TASK_NONE) { target_pos = state->kin; c.target_per_env = 0; }
    const Span T{num_steps, 0, obs_step_stride, env_step_stride, 2};
    const dim3 grid(static_cast<unsigned>((N + kBlock - 1) / kBlock));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GPD_POL(PID_, AW_, ACT_, NK1_)                                                                                              \
    do {                                                                                                                            \
        if (policy->activation == 1)                                                                                                \
            hipLaunchKernelGGL((gpd_rollout_policy_kernel<PID_, AW_, ACT_, NK1_, true>), grid, dim3(kBlock), 0, st, *params, *state, c, T, \
                               *policy, obs12_in, target_pos, init_pose, actions_out, obs12, reward, terminated, truncated, term_obs12); \
        else                                                                                                                        \
            hipLaunchKernelGGL((gpd_rollout_policy_kernel<PID_, AW_, ACT_, NK1_, false>), grid, dim3(kBlock), 0, st, *params, *state, c, T, \
                               *policy, obs12_in, target_pos, init_pose, actions_out, obs12, reward, terminated, truncated, term_obs12); \
    } while (0)
#define GPD_POLN(AW_, ACT_, NK1_)                                                                                                    \
    do {                                                                                                                            \
        const float4 sd = make_float4(action_std[0], AW_ > 1 ? action_std[1] : 0.0f, AW_ > 2 ? action_std[2] : 0.0f,                 \
                                      AW_ > 3 ? action_std[3] : 0.0f);                                                               \
        if (policy->activation == 1)                                                                                                \
            hipLaunchKernelGGL((gpd_rollout_policy_noise_kernel<AW_, ACT_, NK1_, true>), grid, dim3(kBlock), 0, st, *params, *state, c, \
                               T, *policy, obs12_in, target_pos, init_pose, actions_out, obs12, reward, terminated, truncated, noise, \
                               mean_out, sd, term_obs12);                                                                           \
        else                                                                                                                        \
            hipLaunchKernelGGL((gpd_rollout_policy_noise_kernel<AW_, ACT_, NK1_, false>), grid, dim3(kBlock), 0, st, *params, *state, c, \
                               T, *policy, obs12_in, target_pos, init_pose, actions_out, obs12, reward, terminated, truncated, noise, \
                               mean_out, sd, term_obs12);                                                                           \
    } while (0)
    if (noise) {                 // sampling: the RPM action types (the ones examples/learn.py and the reference's learn.py train)
        if (pid) return bad(GPD_ENOTSUP, "sampling (noise) is implemented for ActionType.RPM and ONE_D_RPM");
        if (cfg->act_type == GPD_ACT_RPM) { if (hist) GPD_POLN(4, GPD_ACT_RPM, 5); else GPD_POLN(4, GPD_ACT_RPM, 1); }
        else { if (hist) GPD_POLN(1, GPD_ACT_ONE_D_RPM, 2); else GPD_POLN(1, GPD_ACT_ONE_D_RPM, 1); }
    } else
    if (pid) {                   // (instantiated in the main unit, see GpdPolicyLaunch)
        const GpdPolicyLaunch a{params, state, &c, &T, policy, obs12_in, target_pos, init_pose, actions_out, obs12, reward, terminated,
                                truncated, stream, grid.x, hist ? 1 : 0, term_obs12};
        gpd_detail_launch_policy_pid(a);
    } else if (cfg->act_type == GPD_ACT_RPM) {      // NK1 = K-steps of layer 1: 16*NK1 >= 12 + history features
        if (hist) GPD_POL(false, 4, GPD_ACT_RPM, 5); else GPD_POL(false, 4, GPD_ACT_RPM, 1);
    } else {
        if (hist) GPD_POL(false, 1, GPD_ACT_ONE_D_RPM, 2); else GPD_POL(false, 1, GPD_ACT_ONE_D_RPM, 1);
    }
#undef GPD_POL
#undef GPD_POLN
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_rollout_policy launch");
    return 0;
}

}  // extern "C"

