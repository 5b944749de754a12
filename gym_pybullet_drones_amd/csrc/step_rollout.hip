// step_rollout.hip -- gpd_step / gpd_rollout / gpd_rollout_history: one env step per launch, K env steps per launch (DESIGN.md sections 3.1, 3.2)
#include "gpd_common.inc"
#include "policy_kernel.inc"

namespace {

// ------------------------------------------------------------------------------------------------
// gpd_step: ONE env step per launch.  One lane per drone; the lane loads its state, steps, and stores state,
// observation row and the aviary's reward / flags itself (latency matters more than anything here: at
// N = 65 536 a launch lasts ~5 us).
// ------------------------------------------------------------------------------------------------
// ACT / S1: the action type and "one sub-step per step" as compile-time constants (no action-type ladder, no sub-step loop)
//
// The argument list STARTS with the fourteen dwords the load section needs (kernarg preload, `-mllvm -amdgpu-kernarg-preload-count=14`
// in _native.py: the command processor puts them into SGPRs before the wave starts, so the state / action / counter / target loads are
// issued without first waiting for a scalar load of the argument block -- one memory round trip off the launch's critical path; the
// by-value structs follow and are fetched while the vector loads are in flight).  Hot: copies of S.kin, S.step_counter, S.ld, the slot
// source (S.ring_pos, or the step counters when there is no ring), C.num_envs, C.lanes_per_wave, C.target_per_env.
// DC / FL: the aviary size and the physics flags as compile-time constants for BASELINE's multi-term shapes (0 / -1: from the argument
// block), as in gpd_rollout1_kernel below -- the uniform branches on them fold away, S1 then also applies to multi-drone aviaries.
// HI: with FL, whether the bits above the three add-on models (GPD_PHYS_GROUND, GPD_PHYS_DAMP: what a `Physics.PYB_*` member adds by default)
// are taken from the argument block (true) or known to be clear (false).
template <bool PID, bool EXT, bool MULTI, int AW, int ACT, bool S1, int DC = 0, int FL = -1, bool HI = false>
__global__ __launch_bounds__(kBlock) void gpd_step_kernel(
    float* __restrict__ hot_kin, const float* __restrict__ action, int32_t* __restrict__ hot_counter,
    const float* __restrict__ target_pos, const int32_t* __restrict__ hot_slot, const uint32_t hot_ld, const int32_t hot_num_envs,
    const int32_t hot_lanes_per_wave, const int32_t hot_target_per_env,
    const GpdParams P, const GpdState S_, const GpdStepCfg C_, const float* __restrict__ init_pose, float* __restrict__ obs12,
    float* __restrict__ reward, uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
    float* __restrict__ term_obs12, uint32_t* __restrict__ done_flag, const uint32_t done_seq) {
    // (built member by member: a copy of the argument struct with three members overwritten stays an 80-byte alloca in the EXT variants --
    // the `flag ? S.last_rpm : S.kin` selects become loads from a selected ADDRESS inside it -- i.e. scratch memory, and a launch that
    // needs scratch costs 1.6 us more: hover65536_ext 5.07 -> 6.65 us per step, gpurun_out/bench_r05.log of the first round-5 build)
    const GpdState S{hot_kin, S_.last_rpm, S_.pid, hot_counter, static_cast<int64_t>(hot_ld), S_.dw_force, S_.act_ring, S_.ring_pos, S_.hist_len, 0, S_.bad};
    const GpdStepCfg C{hot_num_envs, C_.drones_per_env, C_.act_type, C_.substeps, C_.physics_flags, C_.pyb_dt, C_.ctrl_dt, C_.inv_ctrl_dt,
                       hot_lanes_per_wave, C_.task, C_.xy_bound, C_.z_bound, C_.tilt_bound, C_.term_dist, C_.trunc_counter, hot_target_per_env,
                       C_.init_per_env, C_.auto_reset};
    const int D = MULTI ? (DC ? DC : C.drones_per_env) : 1;
    const int tid = threadIdx.x;
    const uint32_t N = static_cast<uint32_t>(C.num_envs) * static_cast<uint32_t>(D);
    // MULTI: whole aviaries per workgroup, one lane per drone.  Single-drone aviaries: LW = lanes_per_wave
    // (16/32/64) active lanes per 64-wide wavefront (tuning knob, see GpdStepCfg).
    const int LW = MULTI ? 64 : C.lanes_per_wave;
    const int lanes = MULTI ? (kBlock / D) * D : (kBlock / 64) * LW;
    const uint32_t n_raw = MULTI ? blockIdx.x * lanes + tid : (blockIdx.x * (kBlock / 64) + (tid >> 6)) * LW + (tid & 63);
    Lane L;
    L.tid = tid;
    L.active = (MULTI ? (tid < lanes) : ((tid & 63) < LW)) && (n_raw < N);
    L.n = L.active ? n_raw : 0u;
    L.le = MULTI ? (tid < lanes ? tid / D : 0) : tid;
    L.d = MULTI ? (L.active ? tid - L.le * D : 0) : 0;
    L.env = MULTI ? (L.active ? blockIdx.x * (lanes / D) + L.le : 0u) : L.n;
    L.shfl = MULTI && D <= 64 && (D & (D - 1)) == 0;
    L.base = MULTI ? L.le * D : tid;

    __shared__ __attribute__((aligned(16))) float sh_pos[MULTI ? 4 * kBlock : 4];   // downwash: positions of the env's drones
    __shared__ __attribute__((aligned(16))) float sh_red[MULTI ? 4 * kBlock : 4];   // reward | distance | out-of-bounds per drone
    __shared__ __attribute__((aligned(16))) float sh_rows[kBlock * 12];   // obs rows, for the coalesced store of large batches

    const uint32_t flags = EXT ? (FL >= 0 ? (static_cast<uint32_t>(FL) | (HI ? C.physics_flags & ~7u : 0u)) : C.physics_flags) : 0u;
    Carry c;
    float tgx, tgy, tgz;
    const float4 act = load_action<AW>(action, L.n);
    // action history: the slot this aviary's action goes to (read with the other loads, from a readable dummy when there is
    // no ring: the load section stays branch-free)
    int ring_q = hot_slot[L.env];
    if (S.act_ring) { GPD_DBG(ring_q >= 0 && ring_q < S.hist_len, GPD_DBG_RING_POS, ring_q); ring_q = GPD_DBG_CLAMP(ring_q, 0, S.hist_len - 1); }
    // a single step reads its reset pose only if it resets (in env_step)
    const float* ipose = reinterpret_cast<const float*>(reinterpret_cast<const char*>(init_pose) +
                                                        (C.init_per_env ? L.n * 28u : static_cast<uint32_t>(L.d) * 28u));
    load_carry<PID, EXT, false>(S, C, flags, L, target_pos, nullptr, c, tgx, tgy, tgz, nullptr);
    // An aviary that spans several waves of the workgroup (D not a power of two <= 64): its lane 0 publishes ring_pos + 1 at
    // the end of this kernel, and with no task and no downwash nothing else synchronises the waves -- every wave must have
    // READ ring_pos before any of them gets there (the barrier also waits for the loads above: vmcnt(0))
    if (MULTI && !L.shfl && S.act_ring) __syncthreads();
    c.roll = c.pitch = c.yaw = 0.0f;
    if (PID) quat_to_rpy(c.k.qx, c.k.qy, c.k.qz, c.k.qw, c.roll, c.pitch, c.yaw);

    StepOut out;
    env_step<PID, EXT, MULTI, AW, ACT, S1>(P, C, flags, D, L, act, tgx, tgy, tgz, false, ipose, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                                           0.0f, 0.0f, sh_pos, sh_red, c, out);
    // Observation rows.  A lane's row is 48 bytes, so a wave's direct stores are 48-byte-strided pieces of cache
    // lines; in the bandwidth-bound regime (large batches) the wave transposes its 64 rows through LDS and
    // stores three fully coalesced 1 KiB bursts instead (the rows of a wave are contiguous in memory); narrower waves
    // (lanes_per_wave < 64, a tuning knob) store directly.
    // (round 1 kept direct 48-byte row stores below 2^18 drones; a round-2 A/B on one box has the transposed bursts ahead at
    // every size: 4.74 -> 4.38 us per step at N = 65 536, -8..10 % with DSLPID / 8 sub-steps / 8-drone aviaries, equal at 4 096)
    const bool big = C.lanes_per_wave == 64;
    if (big) {
        float4* mine = reinterpret_cast<float4*>(sh_rows + tid * 12);
        mine[0] = make_float4(out.o[0], out.o[1], out.o[2], out.o[3]);
        mine[1] = make_float4(out.o[4], out.o[5], out.o[6], out.o[7]);
        mine[2] = make_float4(out.o[8], out.o[9], out.o[10], out.o[11]);
        const int wave0 = tid & ~63;                                  // first lane of this wave
        const uint32_t n0 = n_raw - static_cast<uint32_t>(tid & 63);  // first drone of this wave
        if (n0 < N) {
            // valid rows of this wave: its lanes that own a drone (whole aviaries per workgroup: `lanes` may be < 256)
            uint32_t rows = static_cast<uint32_t>(lanes - wave0 < 64 ? (lanes - wave0 > 0 ? lanes - wave0 : 0) : 64);
            if (N - n0 < rows) rows = N - n0;
            const char* src = reinterpret_cast<const char*>(sh_rows + wave0 * 12);
            char* dst = reinterpret_cast<char*>(obs12) + static_cast<size_t>(n0) * 48u;
            const uint32_t off = static_cast<uint32_t>(tid & 63) * 16u;
            __builtin_amdgcn_wave_barrier();                          // same wave: the LDS executes its instructions in order
            if (rows == 64u) {
                // a full wave (every wave but a ragged batch's last): the three reads in one run, one wait, three unconditional stores --
                // the masked form below reads, waits and stores three times over (three LDS round trips on the tail of the kernel)
                const float4 v0 = *reinterpret_cast<const float4*>(src + off), v1 = *reinterpret_cast<const float4*>(src + off + 1024),
                             v2 = *reinterpret_cast<const float4*>(src + off + 2048);
                __builtin_nontemporal_store(f4v{v0.x, v0.y, v0.z, v0.w}, reinterpret_cast<f4v*>(dst + off));
                __builtin_nontemporal_store(f4v{v1.x, v1.y, v1.z, v1.w}, reinterpret_cast<f4v*>(dst + off + 1024));
                __builtin_nontemporal_store(f4v{v2.x, v2.y, v2.z, v2.w}, reinterpret_cast<f4v*>(dst + off + 2048));
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float4 v = *reinterpret_cast<const float4*>(src + off + j * 1024);
                    if (off + j * 1024 < rows * 48u) {                // streamed out, not read again by this path: non-temporal
                        f4v w = {v.x, v.y, v.z, v.w};
                        __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(dst + off + j * 1024));
                    }
                }
            }
        }
        if (!L.active) return;
    } else {
        if (!L.active) return;
        store_obs12(obs12, L.n, out.o[0], out.o[1], out.o[2], out.o[3], out.o[4], out.o[5], out.o[6], out.o[7], out.o[8],
                    out.o[9], out.o[10], out.o[11]);
    }
    if (L.d == 0) {
        // (written once, read by another kernel: non-temporal like the observation bursts -- 4.38 -> 4.30 us per step, A/B)
        __builtin_nontemporal_store(out.rew, &reward[L.env]);
        __builtin_nontemporal_store(static_cast<uint8_t>(out.term ? 1 : 0), &terminated[L.env]);
        __builtin_nontemporal_store(static_cast<uint8_t>(out.trunc ? 1 : 0), &truncated[L.env]);
    }
    if (S.act_ring) {
        // push the raw action into the double ring (slots q and q + H: the H most recent actions stay H consecutive slots);
        // a slot is a contiguous [N][A] block, so this is the coalesced mirror image of the action load
        const size_t slot = static_cast<size_t>(N) * AW, at = static_cast<size_t>(ring_q) * slot + static_cast<size_t>(L.n) * AW;
        float* r0 = S.act_ring + at;
        float* r1 = r0 + static_cast<size_t>(S.hist_len) * slot;
        if (AW == 4) {
            *reinterpret_cast<float4*>(r0) = act;
            *reinterpret_cast<float4*>(r1) = act;
        } else {
            r0[0] = act.x; r1[0] = act.x;
            if (AW == 3) { r0[1] = act.y; r0[2] = act.z; r1[1] = act.y; r1[2] = act.z; }
        }
        if (L.d == 0) S.ring_pos[L.env] = ring_q + 1 == S.hist_len ? 0 : ring_q + 1;
    }
    if (out.reset && term_obs12)
        store_obs12(term_obs12, L.n, out.to[0], out.to[1], out.to[2], out.to[3], out.to[4], out.to[5], out.to[6], out.to[7],
                    out.to[8], out.to[9], out.to[10], out.to[11]);
    // the state block is streamed out non-temporally at every size: nothing of this launch reads it again, and the next launch's loads
    // miss the XCD-private L2 either way (round 5 A/B, profiles/r05_ab_step_kernel_round2.log: 3.95 -> 3.93 us per step at 65 536 drones,
    // 2.91 -> 2.87 at 4 096, equal at 4 194 304; rounds 1-4 kept ordinary stores up to 2^22 drones)
    store_carry<PID, true>(S, L, c);
    signal_done(done_flag, done_seq, L.n == 0u);         // (gpd_step_sync on a one-wave launch; NULL otherwise -- gpd_common.inc)
}

// ------------------------------------------------------------------------------------------------
// gpd_rollout: K env steps per launch.  The drone state is loaded ONCE, lives in VGPRs for all K steps and
// all sub-steps, and is stored ONCE.  A 320-thread workgroup holds two kinds of wavefronts:
//   * 4 COMPUTE waves (256 lanes, one drone each).  Per step a lane prefetches the NEXT step's action row from
//     HBM (its only global memory instruction in the loop, so the wait for it is a wait for loads only), steps,
//     and writes its observation row and the aviary's reward / flags to an LDS slot;
//   * 1 STORE wave that copies the previous step's LDS slot to HBM as fully coalesced 1 KiB dwordx4 bursts (the
//     row-major observation block of a workgroup is contiguous in memory, so the LDS hop also turns the compute
//     lanes' 48-byte-strided rows into whole cache lines).  It issues only stores and never waits for them.
// Why: gfx950 counts loads and stores on ONE in-order counter (vmcnt).  A wave that both prefetches its next
// action and stores its outputs can only wait for "the load" by also waiting for every store issued before
// it, and a store takes > 1 us to be acknowledged -- ~0.4 us per step measured at one wave per SIMD.  With the
// split, loads and stores live on different waves' counters and the only per-step synchronisation is one
// s_barrier (plus an LDS wait).
// ------------------------------------------------------------------------------------------------
constexpr int kStoreLanes = 64;
constexpr int kRollThreads = kBlock + kStoreLanes;
// LDS output ring (dynamic shared memory, sized at launch): `ring` slots of kSlotBytes each.  A small batch (one
// workgroup per CU) gets 4 slots -- the compute waves may run up to 3 steps ahead of the store wave; a large one
// gets 2 so that more workgroups fit a CU (160 KiB of LDS).  ring is a power of two.
constexpr int kSlotBytes = kBlock * 12 * 4 + kBlock * 4 + kBlock + kBlock;   // obs rows | rewards | terminated | truncated

// LDS flags of the compute-wave -> store-wave hand-off.  Plain volatile accesses are enough: one wave's LDS
// instructions execute in order, so a flag written after the data is seen after the data (the empty asm keeps the
// compiler from reordering them).
typedef __attribute__((address_space(3))) int lds_int_t;             // (explicit LDS address space: ds_read/ds_write, not flat)
typedef int i4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i4v lds_i4v_t;
__device__ __forceinline__ int lds_peek(int* p) { return *reinterpret_cast<volatile lds_int_t*>((lds_int_t*)p); }
__device__ __forceinline__ i4v lds_peek4(int* p) { return *reinterpret_cast<volatile lds_i4v_t*>((lds_i4v_t*)p); }
__device__ __forceinline__ void lds_poke(int* p, int v) {
    asm volatile("" ::: "memory");
    *reinterpret_cast<volatile lds_int_t*>((lds_int_t*)p) = v;
    asm volatile("" ::: "memory");
}



// ACT / S1: the action type and "one sub-step per step" as compile-time constants, as in the other two kernels (ACT = -1: the run-time
// ladder, which multi-drone aviaries keep -- their steps are dominated by the exchange and the barriers)
template <bool PID, bool EXT, bool MULTI, int AW, int ACT = -1, bool S1 = false>
__global__ __launch_bounds__(kRollThreads) void gpd_rollout_kernel(
    const GpdParams P, const GpdState S, const GpdStepCfg C, const Span T, const float* __restrict__ action,
    const float* __restrict__ target_pos, const float* __restrict__ init_pose, float* __restrict__ obs12,
    float* __restrict__ reward, uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
    float* __restrict__ term_obs12) {
    const int D = MULTI ? C.drones_per_env : 1;
    const int tid = threadIdx.x;
    const uint32_t N = static_cast<uint32_t>(C.num_envs) * static_cast<uint32_t>(D);
    const int lanes = MULTI ? (kBlock / D) * D : kBlock;             // drones per workgroup
    const uint32_t block_base = blockIdx.x * static_cast<uint32_t>(lanes);
    const uint32_t left = N - block_base;                            // > 0 by construction of the grid
    const int lanes_valid = left < static_cast<uint32_t>(lanes) ? static_cast<int>(left) : lanes;
    const int envs_block = lanes / D;
    const uint32_t env_base = blockIdx.x * static_cast<uint32_t>(envs_block);
    const int envs_valid = lanes_valid / D;
    const int K = T.num_steps;
    const uint32_t flags = EXT ? C.physics_flags : 0u;
    // multi-drone aviaries that fit D aligned lanes of a wave exchange wave-locally: no barrier inside a step
    const bool shfl = MULTI && D <= 64 && (D & (D - 1)) == 0;
    const bool use_flags = !MULTI || shfl;                           // hand-over protocol: LDS flags, or one barrier per step
    // workgroup barriers inside one env step (env_step): the store wave has to take part in each of them
    const int step_barriers = (MULTI && !shfl) ? (((flags & GPD_PHYS_DW) ? 2 * C.substeps : 0) + (C.task != GPD_TASK_NONE ? 2 : 0)) : 0;

    // Output ring: slot = step & (ring-1).  Single-drone aviaries hand over through flags (no barrier: a compute
    // wave never waits for its siblings, and only waits for the store wave when it is ring-1 steps ahead);
    // multi-drone aviaries already synchronise the workgroup inside every step (downwash snapshot, aviary
    // reductions) and keep the simpler two-slot, one-more-barrier-per-step hand-off.
    extern __shared__ __attribute__((aligned(16))) char sh_ring[];
    const int ring = use_flags ? T.ring : 2;
    auto slot_obs = [&](int b) { return reinterpret_cast<float*>(sh_ring + b * kSlotBytes); };
    auto slot_rew = [&](int b) { return reinterpret_cast<float*>(sh_ring + b * kSlotBytes + kBlock * 48); };
    auto slot_term = [&](int b) { return reinterpret_cast<uint8_t*>(sh_ring + b * kSlotBytes + kBlock * 52); };
    auto slot_trunc = [&](int b) { return reinterpret_cast<uint8_t*>(sh_ring + b * kSlotBytes + kBlock * 53); };
    __shared__ __attribute__((aligned(16))) int sh_prog[4];          // steps written, per compute wave
    __shared__ int sh_drained;                                       // steps copied to HBM by the store wave
    __shared__ __attribute__((aligned(16))) float sh_pos[MULTI ? 4 * kBlock : 4];
    __shared__ __attribute__((aligned(16))) float sh_red[MULTI ? 4 * kBlock : 4];
    if (use_flags) {
        if (tid < 4) sh_prog[tid] = 0;
        if (tid == 4) sh_drained = 0;
        wg_barrier();                                                // the only barrier of a flag-synchronised rollout
    }

    if (tid >= kBlock) {
        // ======================= store wave ===========================================================
        const int m = tid - kBlock;
        const bool full = lanes_valid == kBlock && envs_valid == kBlock &&
                          ((reinterpret_cast<uintptr_t>(terminated) | reinterpret_cast<uintptr_t>(truncated) |
                            static_cast<uintptr_t>(T.env_stride)) & 3) == 0;
        const uint32_t lane16 = static_cast<uint32_t>(m) * 16u;
        auto drain = [&](int step) {                                 // LDS slot of `step` -> HBM
            const int b = step & (ring - 1);
            char* og = reinterpret_cast<char*>(obs12 + step * T.obs_stride + static_cast<int64_t>(block_base) * 12);
            const char* ol = reinterpret_cast<const char*>(slot_obs(b));
            float* rg = reward + step * T.env_stride + env_base;
            uint8_t* tg = terminated + step * T.env_stride + env_base;
            uint8_t* ug = truncated + step * T.env_stride + env_base;
            if (full) {
                // a whole workgroup of single-drone aviaries: 12 + 1 unconditional 1 KiB bursts and two 256 B ones,
                // <uniform base> + <lane offset> + <immediate> addressing
                float4 v[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) v[j] = *reinterpret_cast<const float4*>(ol + lane16 + j * 1024);
                const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(slot_rew(b)) + lane16);
                const uint32_t tv = reinterpret_cast<const uint32_t*>(slot_term(b))[m];
                const uint32_t uv = reinterpret_cast<const uint32_t*>(slot_trunc(b))[m];
#pragma unroll
                for (int j = 0; j < 12; ++j) {                       // (write-once streams: non-temporal)
                    f4v w = {v[j].x, v[j].y, v[j].z, v[j].w};
                    __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(og + lane16 + j * 1024));
                }
                f4u w = {rv.x, rv.y, rv.z, rv.w};
                *reinterpret_cast<f4u*>(reinterpret_cast<char*>(rg) + lane16) = w;
                reinterpret_cast<uint32_t*>(tg)[m] = tv;
                reinterpret_cast<uint32_t*>(ug)[m] = uv;
                return;
            }
            const int chunks = lanes_valid * 3;                      // 16-byte chunks; chunk of lane m: j*64 + m
            float4 v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = *reinterpret_cast<const float4*>(ol + lane16 + j * 1024);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                if (j * kStoreLanes + m < chunks) {
                    f4v w = {v[j].x, v[j].y, v[j].z, v[j].w};
                    __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(og + lane16 + j * 1024));
                }
            }
            for (int e = m; e < envs_valid; e += kStoreLanes) {
                rg[e] = slot_rew(b)[e];
                tg[e] = slot_term(b)[e];
                ug[e] = slot_trunc(b)[e];
            }
        };
        __builtin_amdgcn_s_setprio(0);                               // fills the issue gaps of the compute wave it shares a SIMD with
        if (use_flags) {
            for (int t = 0; t < K; ++t) {
                for (;;) {                                           // until all four compute waves have written step t
                    const i4v pr = lds_peek4(sh_prog);
                    const int lo = min(min(pr.x, pr.y), min(pr.z, pr.w));
                    if (__builtin_amdgcn_readfirstlane(lo) > t) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                drain(t);
                __builtin_amdgcn_s_waitcnt(0xC07F);                  // the slot has been read (lgkmcnt(0)) ...
                lds_poke(&sh_drained, t + 1);                        // ... and may be overwritten
            }
            return;
        }
        for (int t = 0; t < K; ++t) {
            for (int i = 0; i < step_barriers; ++i) wg_barrier();    // (the compute waves' env_step barriers)
            if (t > 0) drain(t - 1);                                 // overlaps the compute waves' step t
            wg_barrier();                                            // end of step t
        }
        drain(K - 1);
        return;
    }

    // ======================= compute waves ================================================================
    __builtin_amdgcn_s_setprio(2);
    Lane L;
    L.tid = tid;
    L.active = tid < lanes_valid;
    L.n = L.active ? block_base + tid : 0u;
    L.le = MULTI ? (tid < lanes ? tid / D : 0) : tid;
    L.d = MULTI ? (L.active ? tid - L.le * D : 0) : 0;
    L.env = MULTI ? (L.active ? env_base + L.le : 0u) : L.n;
    L.shfl = shfl;
    L.base = MULTI ? L.le * D : tid;

    Carry c;
    float tgx, tgy, tgz, ip[7];
    // Action rows are prefetched TWO steps ahead into three rotating register sets (a0, a1, a2): with the store
    // wave's bursts ahead of it in the CU's memory pipeline a row takes > 1 us to arrive, longer than one step.
    // The loop is unrolled by three so that the rotation needs no register copies (a copy of a set whose load is
    // still in flight would have to wait for it).  Past the last step the loads re-read the last block.
    auto fetch = [&](int step) { return load_action<AW>(action + (step < K ? step : K - 1) * T.action_stride, L.n); };
    // the rollout keeps its reset pose in registers: no dependent global load inside the step loop
    const float* ipose = reinterpret_cast<const float*>(reinterpret_cast<const char*>(init_pose) +
                                                        (C.init_per_env ? L.n * 28u : static_cast<uint32_t>(L.d) * 28u));
    load_carry<PID, EXT>(S, C, flags, L, target_pos, C.auto_reset ? ipose : S.kin, c, tgx, tgy, tgz, ip);
    // Everything requested above has to have arrived before the step loop starts (the empty asm makes the values
    // live here; the explicit wait lets the compiler's wait-count bookkeeping start the loop with nothing pending,
    // otherwise it would re-wait, conservatively, inside every iteration).
    asm volatile("" :: "v"(c.k.px), "v"(c.k.py), "v"(c.k.pz), "v"(c.k.qx), "v"(c.k.qy), "v"(c.k.qz), "v"(c.k.qw), "v"(c.k.vx),
                       "v"(c.k.vy), "v"(c.k.vz), "v"(c.k.wx), "v"(c.k.wy), "v"(c.k.wz), "v"(tgx), "v"(tgy), "v"(tgz), "v"(c.counter), "v"(ip[0]), "v"(ip[1]), "v"(ip[2]),
                       "v"(ip[3]), "v"(ip[4]), "v"(ip[5]), "v"(ip[6]) : "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0), expcnt/lgkmcnt untouched
    c.roll = c.pitch = c.yaw = 0.0f;
    if (PID) quat_to_rpy(c.k.qx, c.k.qy, c.k.qz, c.k.qw, c.roll, c.pitch, c.yaw);

    float* const tobs_t = term_obs12;
    int drained_seen = 0;                                            // last value of sh_drained this wave has read
    auto do_step = [&](const int t, const float4 act) {
        StepOut out;
        env_step<PID, EXT, MULTI, AW, ACT, S1>(P, C, flags, D, L, act, tgx, tgy, tgz, true, ipose, ip[0], ip[1], ip[2], ip[3], ip[4],
                                               ip[5], ip[6], sh_pos, sh_red, c, out);
        const int b = t & (ring - 1);
        if (use_flags && t - ring + 1 > drained_seen) {              // slot b may still hold step t-ring: has it been drained?
            // (the flag is re-read only when the last value seen does not already clear this step: the store wave
            // normally runs one step behind, so one read clears the next ring-1 steps)
            while ((drained_seen = __builtin_amdgcn_readfirstlane(lds_peek(&sh_drained))) < t - ring + 1)
                __builtin_amdgcn_s_sleep(1);
        }
        float4* ol = reinterpret_cast<float4*>(slot_obs(b) + tid * 12);
        ol[0] = make_float4(out.o[0], out.o[1], out.o[2], out.o[3]);
        ol[1] = make_float4(out.o[4], out.o[5], out.o[6], out.o[7]);
        ol[2] = make_float4(out.o[8], out.o[9], out.o[10], out.o[11]);
        if (!MULTI || (L.active && L.d == 0)) {                      // (single-drone aviaries: every lane owns a slot)
            slot_rew(b)[L.le] = out.rew;
            slot_term(b)[L.le] = out.term ? 1 : 0;
            slot_trunc(b)[L.le] = out.trunc ? 1 : 0;
        }
        if (out.reset && tobs_t && L.active) {
            // Terminal observation of an aviary that ended (rare).  Issued through inline asm on purpose: the
            // compiler's wait-count pass does not see these stores, so they cannot make its waits for the
            // action prefetch conservative (vmcnt(0) in every iteration); stores the pass does not know about can
            // only make a counter-based wait longer, never too short (vmcnt is in-order).
            float* row = reinterpret_cast<float*>(reinterpret_cast<char*>(tobs_t + t * T.obs_stride) + L.n * 48u);
            f4v q0 = {out.to[0], out.to[1], out.to[2], out.to[3]}, q1 = {out.to[4], out.to[5], out.to[6], out.to[7]},
                q2 = {out.to[8], out.to[9], out.to[10], out.to[11]};
            asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\t"
                         "global_store_dwordx4 %0, %3, off offset:32" :: "v"(row), "v"(q0), "v"(q1), "v"(q2) : "memory");
        }
        if (!use_flags) wg_barrier();                                // end of step t
        else lds_poke(&sh_prog[tid >> 6], t + 1);                    // this wave's rows of step t are in the slot
    };
    // (a0 and a1 are requested AFTER the wait above, so that the loop is entered in the state every iteration
    // leaves behind -- two rows in flight, a0 the older -- and the compiler's wait counts stay exact)
    if (!PID) {
        float4 a0 = fetch(0), a1, a2;
        __builtin_amdgcn_sched_barrier(0);                           // (a0 must be the older of the two)
        a1 = fetch(1);
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < K; t += 3) {
            a2 = fetch(t + 2);
            do_step(t, a0);
            if (t + 1 >= K) break;
            a0 = fetch(t + 3);
            do_step(t + 1, a1);
            if (t + 2 >= K) break;
            a1 = fetch(t + 4);
            do_step(t + 2, a2);
        }
    } else {
        // The DSLPID step body is ~2x longer (the row has time to arrive within one step) and three copies of it
        // would not sit well in the instruction cache: one step of look-ahead, one copy of the body.
        float4 act = fetch(0);
        for (int t = 0; t < K; ++t) {
            const float4 act_next = fetch(t + 1);
            do_step(t, act);
            act = act_next;
        }
    }
    if (L.active) store_carry<PID>(S, L, c);
}

// ------------------------------------------------------------------------------------------------
// Per-step outputs of a single-drone rollout lane (shared by the two kernels below).
//   * observation row -> the wave's LDS patch -> three coalesced 1 KiB bursts (the 64 rows of a wave are contiguous in
//     memory), software-pipelined by one step: the bursts of step t - 1, read back from the patch a whole step ago (the
//     LDS round trip is never waited for), go out first; then this step's row is written to the patch and read back
//     into `pend`.  Same wave on both sides: the LDS executes a wave's instructions in order, so the reads see the
//     writes without any wait or barrier (the wave_barrier only pins the order for the compiler).  At step 0 the bursts
//     carry zeros to step 0's rows, which step 1 then overwrites (same lane, same addresses, program order): every
//     store is unconditional.
//   * reward and flags go out directly (already coalesced).
//   * addressing: <uniform running pointer in SGPRs> + <32-bit lane offset>, the global_store "saddr + voffset" form;
//     the empty asm keeps the zero-extension of the offsets inside the loop (hoisted, it turns every store into a
//     64-bit VALU add plus a flat-addressed store).
// ------------------------------------------------------------------------------------------------
template <bool NT_OBS = true>     // NT_OBS: the observation bursts are stored non-temporally (see launch_step for when not)
struct RollOut {
    char* og_prev; char* og; char* rg; uint8_t* tg; uint8_t* ug;     // obs block of the previous / this step, reward, flags
    int64_t obs_step, env_step;                                     // bytes / elements between consecutive steps
    uint32_t g0, g1, g2, e4, e1;                                    // lane offsets: three bursts, reward word, flag byte
    float4* mine; const char* lsrc;                                 // this lane's row in the patch; its three burst chunks
    f4v pend[3];                                                    // the previous step's three bursts
    __device__ __forceinline__ RollOut(float* obs12, float* reward, uint8_t* terminated, uint8_t* truncated, const Span& T,
                                       const uint32_t goff[3], uint32_t eoff4, uint32_t env, float* row, const char* lsrc_)
        : og_prev(reinterpret_cast<char*>(obs12)), og(reinterpret_cast<char*>(obs12)), rg(reinterpret_cast<char*>(reward)),
          tg(terminated), ug(truncated), obs_step(T.obs_stride * 4), env_step(T.env_stride), g0(goff[0]), g1(goff[1]),
          g2(goff[2]), e4(eoff4), e1(env), mine(reinterpret_cast<float4*>(row)), lsrc(lsrc_) {
        pend[0] = pend[1] = pend[2] = f4v{0, 0, 0, 0};
    }
    __device__ __forceinline__ void bursts(char* base) {
        asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2));
        if (NT_OBS) {
            __builtin_nontemporal_store(pend[0], reinterpret_cast<f4v*>(base + g0));   // written once, streamed out
            __builtin_nontemporal_store(pend[1], reinterpret_cast<f4v*>(base + g1));
            __builtin_nontemporal_store(pend[2], reinterpret_cast<f4v*>(base + g2));
        } else {
            *reinterpret_cast<f4v*>(base + g0) = pend[0];
            *reinterpret_cast<f4v*>(base + g1) = pend[1];
            *reinterpret_cast<f4v*>(base + g2) = pend[2];
        }
    }
    // `advance`: false on the first step of the launch (the pointers already address step 0)
    __device__ __forceinline__ void emit(const StepOut& out, bool advance) {
        if (advance) { og_prev = og; og += obs_step; rg += env_step * 4; tg += env_step; ug += env_step; }
        bursts(og_prev);
        mine[0] = make_float4(out.o[0], out.o[1], out.o[2], out.o[3]);
        mine[1] = make_float4(out.o[4], out.o[5], out.o[6], out.o[7]);
        mine[2] = make_float4(out.o[8], out.o[9], out.o[10], out.o[11]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(lsrc + j * 1024);
            pend[j] = f4v{v.x, v.y, v.z, v.w};
        }
        __builtin_amdgcn_wave_barrier();                             // (the next step's row writes stay behind these reads)
        asm volatile("" : "+v"(e4), "+v"(e1));
        __builtin_nontemporal_store(out.rew, reinterpret_cast<float*>(rg + e4));
        __builtin_nontemporal_store(static_cast<uint8_t>(out.term ? 1 : 0), tg + e1);
        __builtin_nontemporal_store(static_cast<uint8_t>(out.trunc ? 1 : 0), ug + e1);
    }
    __device__ __forceinline__ void flush() { bursts(og); }          // after the last step
};

// ------------------------------------------------------------------------------------------------
// gpd_rollout for single-drone aviaries: K env steps per launch with NO helper wave and NO workgroup
// synchronisation at all.  256-thread workgroups, one drone per lane; per step a lane
//   * prefetches the action row two steps ahead (three rotating register sets, loop unrolled x3),
//   * steps (env_step, everything in registers),
//   * writes its 48-byte observation row into its wave's 3 KiB LDS patch, and the wave stores the patch as three
//     fully coalesced 1 KiB dwordx4 bursts (the 64 rows of a wave are contiguous in memory); reward and flags go
//     out directly (already coalesced).
// Every global memory instruction of the loop body is UNCONDITIONAL -- this is what makes it fast: gfx950 counts
// loads and stores on one in-order counter, and only with a fixed number of operations per step can the wait for
// "the row requested two steps ago" be an exact `vmcnt(14)` (the two younger loads and the twelve stores of the two
// steps in between may still be in flight) instead of a wait for every store issued so far (a store takes > 1 us
// to be acknowledged).  Lanes without a drone (ragged last workgroup) are exact CLONES of the first drone of their
// own workgroup -- same state, same action row, same arithmetic, hence the same bits -- and store to that drone's
// addresses: a benign duplicate write instead of a branch around the stores.  (Calls that ask for terminal observations -- conditional stores -- use
// the compute-wave + store-wave kernel above.)
// ------------------------------------------------------------------------------------------------
// MULTI: aviaries of D = 2 .. 64 drones, WHOLE aviaries per wave ((64 / D) D lanes of a wave hold a drone: D consecutive lanes of one
// wave per aviary, wave-local exchange inside env_step, no workgroup barrier).  Every lane of an aviary ends a step with the
// aviary's reward and flags and stores them to the aviary's slot -- D identical writes instead of a branch; a lane without a drone
// is a clone (the lane mapping in the kernel says of whom), so clones replay their originals bit for bit.
// RING (gpd_rollout_history): every step's raw action is also pushed into the action ring, like gpd_step does (slots q and
// q + H of the double ring; two more stores per lane and step, which the explicit wait counts of the loop include because
// they are unconditional -- a compile-time variant, not a run-time test).
// (the argument list starts with the fourteen dwords the launch's first loads need -- kernarg preload, as for gpd_step_kernel: the state,
// the first action rows, the counter, the target and the reset pose are requested without first waiting for the argument block)
// DC: the aviary size as a compile-time constant (0: C.drones_per_env) -- BASELINE's two multi-drone shapes, pairs (config 5) and stacks of
// eight (config 3 ii), at one sub-step per step: the size tests, the mates loops of the downwash and of the task sums and the sub-step loop
// fold away, the LDS reads of all mates are issued together and no loop branch is taken (a taken branch costs ~60 cycles at one wave per
// SIMD).  Same operations in the same order: bit for bit the generic kernel (scratch/exp_r06/ab_unroll.py).
// FL: the physics flags as a compile-time constant too (-1: C.physics_flags) -- the reference's two multi-drone add-on sets, PYB_DW (4) and
// PYB_GND_DRAG_DW (7): the flag tests of every sub-step (uniform branches: ~11 cycles not taken, 25-60 taken) fold away.
// HI: with FL, the bits above the add-on models (the ground plane and Bullet's damping, which `Physics.PYB_*` members add by default) come
// from the argument block (true) or are known to be clear (false: exactly the reference's explicit integrator + the add-on models).
template <bool PID, bool EXT, int AW, int ACT, bool S1, bool MULTI, bool NT_OBS = true, bool RING = false, int DC = 0, int FL = -1, bool HI = false>
__global__ __launch_bounds__(kBlock) void gpd_rollout1_kernel(
    float* __restrict__ hot_kin, const float* __restrict__ action, int32_t* __restrict__ hot_counter, const float* __restrict__ target_pos,
    const float* __restrict__ init_pose, const uint32_t hot_ld, const int32_t hot_num_envs, const int32_t hot_num_steps, const uint32_t hot_bits,
    const GpdParams P, const GpdState S_, const GpdStepCfg C_, const Span T_, float* __restrict__ obs12,
    float* __restrict__ reward, uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
    float* __restrict__ term_obs12) {
    // (member by member, never a copy of the argument structs: see gpd_step_kernel)
    const GpdState S{hot_kin, S_.last_rpm, S_.pid, hot_counter, static_cast<int64_t>(hot_ld), S_.dw_force, S_.act_ring, S_.ring_pos, S_.hist_len, 0, S_.bad};
    const GpdStepCfg C{hot_num_envs, C_.drones_per_env, C_.act_type, C_.substeps, C_.physics_flags, C_.pyb_dt, C_.ctrl_dt, C_.inv_ctrl_dt,
                       C_.lanes_per_wave, C_.task, C_.xy_bound, C_.z_bound, C_.tilt_bound, C_.term_dist, C_.trunc_counter,
                       static_cast<int32_t>((hot_bits >> 2) & 1u), static_cast<int32_t>((hot_bits >> 1) & 1u), static_cast<int32_t>(hot_bits & 1u)};
    const Span T{hot_num_steps, T_.action_stride, T_.obs_stride, T_.env_stride, T_.ring, static_cast<int32_t>((hot_bits >> 3) & 1u)};
    const int tid = threadIdx.x;
    // workgroup -> drones: the identity, or (T.xcd, large batches) every XCD one contiguous eighth of the drones instead of every eighth
    // workgroup -- at 65 536 drones that changed nothing (0.816 vs 0.813-0.821 us per step, round-2 A/B)
    uint32_t bid = blockIdx.x;
    if (T.xcd) { const uint32_t per = gridDim.x >> 3; bid = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); }
    const int D = MULTI ? (DC ? DC : C.drones_per_env) : 1;
    const uint32_t N = static_cast<uint32_t>(C.num_envs) * static_cast<uint32_t>(D);
    const int K = T.num_steps;
    const uint32_t flags = EXT ? (FL >= 0 ? (static_cast<uint32_t>(FL) | (HI ? C.physics_flags & ~7u : 0u)) : C.physics_flags) : 0u;
    // lane -> drone.  A wave holds W = (64 / D) D drones: WHOLE aviaries, so that an aviary's exchange never leaves its wave (64 when D
    // divides 64 -- every lane has a drone; 63 for D = 3, 60 for D = 12 ...: the last 64 - W < D lanes of the wave are "pad" lanes).  A lane
    // without a drone -- a pad lane, or a lane past the end of the batch -- is an exact CLONE: same state, same action rows, same
    // arithmetic, hence the same bits, stored to its original's addresses (a benign duplicate write instead of a branch around the
    // stores).  Whole aviaries past the end clone the first aviary of their wave (of their workgroup, when the wave has none) and are
    // self-contained: they exchange through their own LDS slots.  A pad lane clones drone (lane - W) of its wave's first aviary and
    // reads that aviary's slots (same wave: in order).
    const int wave0 = tid & ~63, lane = tid & 63;
    const int W = MULTI ? (64 / D) * D : 64;
    const uint32_t block_base = bid * static_cast<uint32_t>(4 * W);                                   // first drone of this workgroup (< N)
    const uint32_t n0 = block_base + static_cast<uint32_t>((tid >> 6) * W);                           // first drone of this wave
    const uint32_t rows = n0 < N ? ((N - n0 < static_cast<uint32_t>(W)) ? N - n0 : static_cast<uint32_t>(W)) : 0u;   // lanes of this wave that own a drone
    const uint32_t first = rows ? n0 : block_base;                                                    // the aviary this wave's clones copy
    // the drone a row r of this wave's patch belongs to (r = lane index): its own, or its original's
    auto drone_of = [&](uint32_t r) {
        if (!MULTI) return r < rows ? n0 + r : block_base;
        const uint32_t rw = static_cast<uint32_t>(W);
        return r < rows ? n0 + r : first + (r < rw ? r - (r / static_cast<uint32_t>(D)) * static_cast<uint32_t>(D) : r - rw);
    };
    Lane L;
    L.tid = tid; L.shfl = MULTI;
    L.le = MULTI ? tid / D : tid;
    L.active = static_cast<uint32_t>(lane) < rows;
    L.n = drone_of(static_cast<uint32_t>(lane));
    L.d = MULTI ? static_cast<int>(L.n % static_cast<uint32_t>(D)) : 0;
    L.base = MULTI ? wave0 + (lane < W ? (lane / D) * D : 0) : tid;
    L.env = MULTI ? L.n / static_cast<uint32_t>(D) : L.n;

    __shared__ __attribute__((aligned(16))) float sh_rows[kBlock * 12];
    __shared__ __attribute__((aligned(16))) float sh_pos[MULTI ? 4 * kBlock : 4];   // downwash: positions of the aviary's drones
    __shared__ __attribute__((aligned(16))) float sh_red[MULTI ? 4 * kBlock : 4];   // reward | distance | out-of-bounds per drone

    // loop-invariant addressing of this lane's three 16-byte chunks of its wave's 3 KiB row patch
    uint32_t goff[3];
    const char* lsrc = reinterpret_cast<const char*>(sh_rows + wave0 * 12) + lane * 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t cidx = static_cast<uint32_t>(j * 64 + lane), r = cidx / 3u, part = cidx - 3u * r;
        goff[j] = drone_of(r) * 48u + part * 16u;                     // a clone's row goes to the row of its original
    }
    const uint32_t eoff4 = L.env * 4u;

    Carry c;
    float tgx, tgy, tgz, ip[7];
    auto fetch = [&](int step) { return load_action<AW, true>(action + (step < K ? step : K - 1) * T.action_stride, L.n); };
    const float* ipose = reinterpret_cast<const float*>(reinterpret_cast<const char*>(init_pose) +
                                                        (C.init_per_env ? L.n * 28u : static_cast<uint32_t>(L.d) * 28u));
    load_carry<PID, EXT>(S, C, flags, L, target_pos, C.auto_reset ? ipose : S.kin, c, tgx, tgy, tgz, ip);
    int ring_q = 0;                                                  // RING: the slot this aviary's next action goes to
    if constexpr (RING) {
        ring_q = S.ring_pos[L.env];
        GPD_DBG(ring_q >= 0 && ring_q < S.hist_len, GPD_DBG_RING_POS, ring_q); ring_q = GPD_DBG_CLAMP(ring_q, 0, S.hist_len - 1);
    }
    asm volatile("" :: "v"(c.k.px), "v"(c.k.py), "v"(c.k.pz), "v"(c.k.qx), "v"(c.k.qy), "v"(c.k.qz), "v"(c.k.qw), "v"(c.k.vx),
                       "v"(c.k.vy), "v"(c.k.vz), "v"(c.k.wx), "v"(c.k.wy), "v"(c.k.wz), "v"(tgx), "v"(tgy), "v"(tgz),
                       "v"(c.counter), "v"(ip[0]), "v"(ip[1]), "v"(ip[2]), "v"(ip[3]), "v"(ip[4]), "v"(ip[5]), "v"(ip[6])
                 : "memory");
    if constexpr (RING) asm volatile("" :: "v"(ring_q) : "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0): the loop starts with nothing pending
    // The ONLY workgroup barrier of the launch: every wave has its state in registers before any wave can reach its
    // store_carry, so a clone lane (above) has read its original's state of step 0, not of step K.
    __builtin_amdgcn_s_barrier();
    c.roll = c.pitch = c.yaw = 0.0f;
    if (PID) quat_to_rpy(c.k.qx, c.k.qy, c.k.qz, c.k.qw, c.roll, c.pitch, c.yaw);

    (void)term_obs12;   // (terminal observations: the host routes such calls to gpd_rollout_kernel -- a conditional
                        // store in this loop body would make the wait counts conservative again)
    RollOut<NT_OBS> ro(obs12, reward, terminated, truncated, T, goff, eoff4, L.env, sh_rows + tid * 12, lsrc);
    float irpy[3] = {0.0f, 0.0f, 0.0f};
    if (C.auto_reset) quat_to_rpy(ip[3], ip[4], ip[5], ip[6], irpy[0], irpy[1], irpy[2]);
    auto do_step = [&](const int t, const float4 act) {
        StepOut out;
        env_step<PID, EXT, MULTI, AW, ACT, S1>(P, C, flags, D, L, act, tgx, tgy, tgz, true, ipose, ip[0], ip[1], ip[2], ip[3],
                                               ip[4], ip[5], ip[6], sh_pos, sh_red, c, out, irpy);
        ro.emit(out, t > 0);                                          // (see RollOut: pipelined bursts, unconditional stores)
        if constexpr (RING) {
            const size_t slot = static_cast<size_t>(N) * AW;
            float* r0 = S.act_ring + static_cast<size_t>(ring_q) * slot + static_cast<size_t>(L.n) * AW;
            float* r1 = r0 + static_cast<size_t>(S.hist_len) * slot;
            if (AW == 4) { *reinterpret_cast<float4*>(r0) = act; *reinterpret_cast<float4*>(r1) = act; }
            else { r0[0] = act.x; r1[0] = act.x; if (AW == 3) { r0[1] = act.y; r0[2] = act.z; r1[1] = act.y; r1[2] = act.z; } }
            ring_q = ring_q + 1 == S.hist_len ? 0 : ring_q + 1;
        }
    };
    // Action rows, three steps per loop iteration: the rows of the NEXT iteration (b0..b2) are requested at the top of
    // this one and claimed at its end with an explicit vmcnt(18) -- "everything but the youngest 18 operations", i.e.
    // but the 3 x 6 stores of this iteration's steps, has completed.  The rows had three steps of arithmetic to arrive,
    // the wait never touches a store younger than three steps, and no load is in flight across the loop's back edge
    // (where the compiler's wait-count bookkeeping would otherwise fall back to a wait for nearly every store).
    if (!PID) {
        float4 a0 = fetch(0), a1 = fetch(1), a2 = fetch(2);
        asm volatile("" :: "v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w), "v"(a2.x),
                           "v"(a2.y), "v"(a2.z), "v"(a2.w) : "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int t = 0; t < K; t += 3) {
            const float4 b0 = fetch(t + 3), b1 = fetch(t + 4), b2 = fetch(t + 5);
            do_step(t, a0);
            if (t + 1 >= K) break;
            do_step(t + 1, a1);
            if (t + 2 >= K) break;
            do_step(t + 2, a2);
            asm volatile("" :: "v"(b0.x), "v"(b0.y), "v"(b0.z), "v"(b0.w), "v"(b1.x), "v"(b1.y), "v"(b1.z), "v"(b1.w),
                               "v"(b2.x), "v"(b2.y), "v"(b2.z), "v"(b2.w) : "memory");
            a0 = b0; a1 = b1; a2 = b2;
        }
    } else {
        // DSLPID action types: the step body is ~2x longer (a row arrives within one step) and three copies of it would
        // not sit well in the instruction cache -- one step per iteration, the next row claimed with an exact vmcnt(6)
        float4 a = fetch(0);
        asm volatile("" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w) : "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int t = 0; t < K; ++t) {
            const float4 b = fetch(t + 1);
            do_step(t, a);
            asm volatile("" :: "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w) : "memory");
            a = b;
        }
    }
    ro.flush();                                                      // the last step's bursts
    if (L.active) store_carry<PID>(S, L, c);
    if constexpr (RING) { if (L.active && L.d == 0) S.ring_pos[L.env] = ring_q; }
}

// the kernels compiled for one aviary size / flag set exist for BaseRLAviary's five action types (not for the raw-RPM rows of CtrlAviary
// and of subclasses with their own _preprocessAction: every variant is another pair of large kernels to compile)
template <int ACT> constexpr bool kSizedAct = ACT != GPD_ACT_RAW_RPM && ACT != GPD_ACT_DIRECT_RPM;
// GPD_ROLLOUT_SIZED=0 (diagnostics, the A/B; read once per process): the generic kernels only
inline bool sized_variants() {
    static const bool on = [] { const char* e = getenv("GPD_ROLLOUT_SIZED"); return e == nullptr || e[0] != '0'; }();
    return on;
}

template <bool PID, bool EXT, int AW, int ACT>
hipError_t launch_step(bool multi, hipStream_t st, const GpdParams& P, const GpdState& S, const GpdStepCfg& C,
                       const Span& T, const float* action, const float* target_pos, const float* init_pose,
                       float* obs12, float* reward, uint8_t* terminated, uint8_t* truncated, float* term_obs12, GpdDone* done) {
    const int64_t N = static_cast<int64_t>(C.num_envs) * C.drones_per_env;
    if (T.num_steps == 1) {      // gpd_step, or a rollout of one step: the low-latency single-step kernel
        const int lanes = multi ? (kBlock / C.drones_per_env) * C.drones_per_env : (kBlock / 64) * C.lanes_per_wave;
        const dim3 grid(static_cast<unsigned>((N + lanes - 1) / lanes));
        // the completion word is for launches whose drones all sit in wave 0 of workgroup 0 (see the end of gpd_step_kernel)
        uint32_t* const done_flag = (done != nullptr && N <= (multi ? 64 : C.lanes_per_wave)) ? done->flag : nullptr;
        const uint32_t done_seq = done != nullptr ? done->seq : 0u;
        if (done != nullptr) done->used = done_flag != nullptr;
#define GPD_STEP_HOT S.kin, action, S.step_counter, target_pos, static_cast<const int32_t*>(S.act_ring ? S.ring_pos : S.step_counter), \
                     static_cast<uint32_t>(S.ld), C.num_envs, C.lanes_per_wave, C.target_per_env
        const bool sized = sized_variants();
        bool launched = false;
        if constexpr (EXT && kSizedAct<ACT>) {      // BASELINE configs 3 (ii), 5, 3 (i) at one sub-step per step (see gpd_rollout1_kernel)
#define GPD_STEP1H(MULTI_, S1_, DC_, FL_, HI_)                                                                                                              \
    do {                                                                                                                                                 \
        hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, MULTI_, AW, ACT, S1_, DC_, FL_, HI_>), grid, dim3(kBlock), 0, st, GPD_STEP_HOT, P, S, C,               \
                           init_pose, obs12, reward, terminated, truncated, term_obs12, done_flag, done_seq);                                           \
        launched = true;                                                                                                                                 \
    } while (0)
#define GPD_STEP1(MULTI_, DC_, FL_) do { if (hi) GPD_STEP1H(MULTI_, true, DC_, FL_, true); else GPD_STEP1H(MULTI_, true, DC_, FL_, false); } while (0)
            const uint32_t low = C.physics_flags & 7u;          // the add-on models; above them: the ground plane, Bullet's damping
            const bool hi = (C.physics_flags & ~7u) != 0u, s1 = C.substeps == 1;
            if (sized && s1 && multi && C.drones_per_env == 8 && low == 7u) GPD_STEP1(true, 8, 7);
            else if (sized && s1 && multi && C.drones_per_env == 2 && low == 4u) GPD_STEP1(true, 2, 4);
            else if (sized && s1 && !multi && low == 7u) GPD_STEP1(false, 0, 7);
            // the two multi-drone add-on sets under the sub-step loop (30 Hz control of 240 Hz physics is the reference's default): one variant each,
            // the bits above the add-on models read at run time whatever they are (single drones gain 3-5 % there: no variant)
            else if (sized && multi && C.drones_per_env == 8 && low == 7u) GPD_STEP1H(true, false, 8, 7, true);
            else if (sized && multi && C.drones_per_env == 2 && low == 4u) GPD_STEP1H(true, false, 2, 4, true);
            // no add-on model, the ground plane / damping bits alone: what `Physics.PYB` -- the default of HoverAviary() and MultiHoverAviary(), whose
            // step() is this kernel -- resolves to; single drones and pairs, one sub-step or the loop
            else if (sized && low == 0u && hi && !multi) { if (s1) GPD_STEP1H(false, true, 0, 0, true); else GPD_STEP1H(false, false, 0, 0, true); }
            else if (sized && low == 0u && hi && multi && C.drones_per_env == 2) { if (s1) GPD_STEP1H(true, true, 2, 0, true); else GPD_STEP1H(true, false, 2, 0, true); }
#undef GPD_STEP1
#undef GPD_STEP1H
        }
        if (launched) {
        } else if (multi) {
            hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, true, AW, ACT, false>), grid, dim3(kBlock), 0, st, GPD_STEP_HOT, P, S, C,
                               init_pose, obs12, reward, terminated, truncated, term_obs12, done_flag, done_seq);
        } else if (C.substeps == 1) {
            hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, false, AW, ACT, true>), grid, dim3(kBlock), 0, st, GPD_STEP_HOT, P, S, C,
                               init_pose, obs12, reward, terminated, truncated, term_obs12, done_flag, done_seq);
        } else {
            hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, false, AW, ACT, false>), grid, dim3(kBlock), 0, st, GPD_STEP_HOT, P, S, C,
                               init_pose, obs12, reward, terminated, truncated, term_obs12, done_flag, done_seq);
        }
#undef GPD_STEP_HOT
    } else {
        // aviaries of up to 64 drones run gpd_rollout1_kernel with WHOLE aviaries per wave (wave-local exchange, no helper wave, no
        // workgroup barrier) -- also where the size does not divide 64 and some lanes of every wave stay without a drone: 0.50-0.65 of the
        // compute-wave + store-wave kernel's time per step at every size from 3 to 63, half-empty waves (33 drones) included
        // (scratch/exp_r06/ab_wave_local.py, profiles/r06_ab_wave_local_rollout.json).  GPD_ROLLOUT_WAVE_LOCAL=0 (diagnostics, the A/B):
        // only the powers of two, the rule of rounds 2-5
        static const char* const wl_env = getenv("GPD_ROLLOUT_WAVE_LOCAL");
        const int Dm = C.drones_per_env, per_wave = Dm <= 64 ? (64 / Dm) * Dm : 0;
        const bool pow2 = Dm <= 64 && (Dm & (Dm - 1)) == 0;
        static const bool store_wave_variant = getenv("GPD_ROLLOUT_STOREWAVE") != nullptr;   // A/B switch, diagnostics only
        // (terminal observations -- conditional stores -- and the diagnostics switch go to the compute-wave + store-wave kernel, whose
        // workgroups hold (256 / D) D drones; the two packings agree where D divides 64)
        const bool shfl = multi && Dm <= 64 && !store_wave_variant && term_obs12 == nullptr &&
                          (pow2 || S.act_ring != nullptr || !(wl_env != nullptr && wl_env[0] == '0'));   // (gpd_rollout_history: this kernel only)
        const int lanes = shfl ? 4 * per_wave : (multi ? (kBlock / Dm) * Dm : kBlock);
        const dim3 grid(static_cast<unsigned>((N + lanes - 1) / lanes));
#define GPD_ROLL1_HOT S.kin, action, S.step_counter, target_pos, init_pose, static_cast<uint32_t>(S.ld), C.num_envs, Tr.num_steps, \
                      (static_cast<uint32_t>(C.auto_reset != 0) | (static_cast<uint32_t>(C.init_per_env != 0) << 1) |      \
                       (static_cast<uint32_t>(C.target_per_env != 0) << 2) | (static_cast<uint32_t>(Tr.xcd != 0) << 3))
        Span Tr = T;
        static const char* const xcd_env = getenv("GPD_ROLLOUT_XCD");          // (diagnostics: 1 = contiguous eighths per XCD, 0 = never)
        Tr.xcd = (grid.x % 8u == 0u && xcd_env != nullptr && xcd_env[0] == '1') ? 1 : 0;
        Tr.ring = ((!multi || pow2) && grid.x <= 2u * 256u) ? 4 : 2;   // <= 2 workgroups per CU: LDS is not what limits occupancy
        const size_t lds = static_cast<size_t>(Tr.ring) * kSlotBytes;
        if (S.act_ring && !store_wave_variant && term_obs12 == nullptr && (shfl || !multi)) {   // gpd_rollout_history (it checked the shape)
            if (shfl)
                hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, false, true, true, true>), grid, dim3(kBlock), 0, st, GPD_ROLL1_HOT, P, S, C, Tr, obs12, reward, terminated, truncated, term_obs12);
            else if (C.substeps == 1)
                hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, true, false, true, true>), grid, dim3(kBlock), 0, st, GPD_ROLL1_HOT, P, S, C, Tr, obs12, reward, terminated, truncated, term_obs12);
            else
                hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, false, false, true, true>), grid, dim3(kBlock), 0, st, GPD_ROLL1_HOT, P, S, C, Tr, obs12, reward, terminated, truncated, term_obs12);
        } else
        if (shfl) {   // aviaries of 2 .. 64 drones, whole aviaries per wave: no helper wave either
            const bool sized = sized_variants();
#define GPD_ROLL1H(S1_, DC_, FL_, HI_) hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, S1_, true, true, false, DC_, FL_, HI_>), grid, dim3(kBlock), 0, st, \
                                                          GPD_ROLL1_HOT, P, S, C, Tr, obs12, reward, terminated, truncated, term_obs12)
#define GPD_ROLL1(S1_, DC_, FL_) GPD_ROLL1H(S1_, DC_, FL_, false)
            const bool s1 = C.substeps == 1;
            bool done = false;
            if constexpr (EXT && kSizedAct<ACT>) {
                // pairs with PYB_DW, stacks of eight with PYB_GND_DRAG_DW (BASELINE configs 5 and 3 ii); any size with every add-on; each with
                // the ground plane / damping bits clear (the explicit integrator + add-ons, as BASELINE words it) or from the argument block
                const uint32_t low = C.physics_flags & 7u;
                const bool hi = (C.physics_flags & ~7u) != 0u;
                if (sized && s1 && Dm == 8 && low == 7u) { if (hi) GPD_ROLL1H(true, 8, 7, true); else GPD_ROLL1H(true, 8, 7, false); done = true; }
                else if (sized && s1 && Dm == 2 && low == 4u) { if (hi) GPD_ROLL1H(true, 2, 4, true); else GPD_ROLL1H(true, 2, 4, false); done = true; }
                else if (sized && s1 && low == 7u) { if (hi) GPD_ROLL1H(true, 0, 7, true); else GPD_ROLL1H(true, 0, 7, false); done = true; }
                // the same under the sub-step loop (30 Hz control is the reference's default): one variant each, upper bits read at run time
                else if (sized && Dm == 8 && low == 7u) { GPD_ROLL1H(false, 8, 7, true); done = true; }
                else if (sized && Dm == 2 && low == 4u) { GPD_ROLL1H(false, 2, 4, true); done = true; }
                else if (sized && low == 7u) { GPD_ROLL1H(false, 0, 7, true); done = true; }
                // pairs with no add-on model and the ground plane / damping bits alone: MultiHoverAviary's defaults (Physics.PYB, 30 Hz control)
                else if (sized && Dm == 2 && low == 0u && hi) { if (s1) GPD_ROLL1H(true, 2, 0, true); else GPD_ROLL1H(false, 2, 0, true); done = true; }
            }
            if constexpr (kSizedAct<ACT>) {
                // pairs with any other flag set, and any size, at one sub-step per step (under the sub-step loop the size alone buys nothing)
                if (done) {}
                else if (sized && s1 && Dm == 2) { GPD_ROLL1(true, 2, -1); done = true; }
                else if (sized && s1) { GPD_ROLL1(true, 0, -1); done = true; }
            }
            if (!done) GPD_ROLL1(false, 0, -1);
#undef GPD_ROLL1
#undef GPD_ROLL1H
        } else if (multi) {
            hipLaunchKernelGGL((gpd_rollout_kernel<PID, EXT, true, AW>), grid, dim3(kRollThreads), lds, st, P, S, C, Tr,
                               action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
        } else if (!store_wave_variant && term_obs12 == nullptr) {
            // The headline shape -- plain DYN, RPM actions, one sub-step, a batch that leaves one wave per SIMD -- stores its
            // observation bursts as ordinary stores: measured 3-4 % faster there (0.837 -> 0.803 us per step), while every other
            // shape (larger batches, sub-step loops, multi-drone aviaries) is 1-3 % faster with non-temporal ones
            // (A/B on one box, round 2: scratch/ab.sh, scratch/ab2.sh) -- and only for long rollouts: the ordinary stores leave
            // their lines to the end-of-kernel write-back, which a 20-step launch does not amortise (1.14 vs 1.00 us per step).
            // (GPD_ROLLOUT_OBS_STORES=plain: the ordinary stores at every size -- diagnostics: the boxes of the pool differ on the
            // non-temporal streaming rate, profiles/README.md)
            static const char* const obs_stores = getenv("GPD_ROLLOUT_OBS_STORES");
            const bool plain_obs = obs_stores != nullptr && obs_stores[0] == 'p';
            if (C.substeps == 1 && !PID && !EXT && ACT == GPD_ACT_RPM && (plain_obs || (N <= (1 << 17) && T.num_steps >= 48)))
                hipLaunchKernelGGL((gpd_rollout1_kernel<false, false, 4, GPD_ACT_RPM, true, false, false>), grid, dim3(kBlock), 0, st, GPD_ROLL1_HOT, P, S, C, Tr, obs12, reward, terminated, truncated, term_obs12);
            else {
#define GPD_ROLL1S(S1_, FL_, HI_) hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, S1_, false, true, false, 0, FL_, HI_>), grid, dim3(kBlock), 0, st, \
                                                     GPD_ROLL1_HOT, P, S, C, Tr, obs12, reward, terminated, truncated, term_obs12)
                const bool s1 = C.substeps == 1, sized = sized_variants();
                bool done = false;
                if constexpr (EXT && kSizedAct<ACT>) {
                    // single drones with every add-on model (BASELINE config 3 i), or with none and the ground plane / damping bits alone (what
                    // `Physics.PYB` resolves to by default): the tests on the three add-on flags fold away
                    const uint32_t low = C.physics_flags & 7u;
                    const bool hi = (C.physics_flags & ~7u) != 0u;
                    if (sized && s1 && low == 7u) { if (hi) GPD_ROLL1S(true, 7, true); else GPD_ROLL1S(true, 7, false); done = true; }
                    else if (sized && low == 0u && hi) { if (s1) GPD_ROLL1S(true, 0, true); else GPD_ROLL1S(false, 0, true); done = true; }
                }
                if (done) {}
                else if (s1) GPD_ROLL1S(true, -1, false);
                else GPD_ROLL1S(false, -1, false);
#undef GPD_ROLL1S
            }
        } else {
            // single drones with terminal observations kept (what a VecEnv-style caller asks for): action type and sub-step count folded
            if (sized_variants() && C.substeps == 1)
                hipLaunchKernelGGL((gpd_rollout_kernel<PID, EXT, false, AW, ACT, true>), grid, dim3(kRollThreads), lds, st, P, S, C, Tr,
                                   action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
            else if (sized_variants())
                hipLaunchKernelGGL((gpd_rollout_kernel<PID, EXT, false, AW, ACT, false>), grid, dim3(kRollThreads), lds, st, P, S, C, Tr,
                                   action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
            else
                hipLaunchKernelGGL((gpd_rollout_kernel<PID, EXT, false, AW>), grid, dim3(kRollThreads), lds, st, P, S, C, Tr,
                                   action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
        }
    }
    return hipGetLastError();
}

// argument checks + launch shared by gpd_step (K = 1) and gpd_rollout
int step_impl(const char* who, const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const Span& T,
              const float* action, const float* target_pos, const float* init_pose, float* obs12, float* reward,
              uint8_t* terminated, uint8_t* truncated, float* term_obs12, void* stream, GpdDone* done = nullptr) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string(who) + ": " + msg).c_str()); };
    if (!params || !state || !cfg) return bad(GPD_EINVAL, "NULL params/state/cfg");
    if (!state->kin || !state->step_counter) return bad(GPD_EINVAL, "NULL state.kin/step_counter");
    if (const char* why = state_layout_problem(state)) return bad(GPD_EINVAL, why);
    if (!action || !obs12 || !reward || !terminated || !truncated)
        return bad(GPD_EINVAL, "NULL action/obs12/reward/terminated/truncated");
    if (cfg->num_envs <= 0 || cfg->drones_per_env <= 0 || cfg->substeps <= 0)
        return bad(GPD_EINVAL, "num_envs, drones_per_env and substeps must be positive");
    if (cfg->drones_per_env > kBlock) return bad(GPD_ERANGE, "drones_per_env > 256 is not supported");
    if (cfg->act_type < GPD_ACT_RPM || cfg->act_type > GPD_ACT_DIRECT_RPM) return bad(GPD_EINVAL, "unknown act_type");
    if (cfg->task < GPD_TASK_NONE || cfg->task > GPD_TASK_MULTIHOVER) return bad(GPD_EINVAL, "unknown task");
    if (cfg->physics_flags & ~31u) return bad(GPD_EINVAL, "unknown physics flag");
    const int64_t N = static_cast<int64_t>(cfg->num_envs) * cfg->drones_per_env;
    if (state->ld < N) return bad(GPD_EINVAL, "state.ld < num_envs*drones_per_env");
    if (N > (1LL << 26)) return bad(GPD_ERANGE, "more than 2^26 drones per launch (32-bit byte offsets)");
    const bool pid = cfg->act_type == GPD_ACT_PID || cfg->act_type == GPD_ACT_VEL || cfg->act_type == GPD_ACT_ONE_D_PID;
    if (pid && !state->pid) return bad(GPD_EINVAL, "PID action type needs state.pid");
    if (pid && params->pid_kf <= 0.0f)
        return bad(GPD_ENOTSUP, "no DSLPID controller for this airframe (CF2X/CF2P only)");
    if ((cfg->physics_flags & GPD_PHYS_DRAG) && !state->last_rpm) return bad(GPD_EINVAL, "GPD_PHYS_DRAG needs state.last_rpm");
    if (cfg->task != GPD_TASK_NONE && !target_pos) return bad(GPD_EINVAL, "task needs target_pos");
    if (cfg->auto_reset && !init_pose) return bad(GPD_EINVAL, "auto_reset needs init_pose");
    const bool multi = cfg->drones_per_env > 1;
    GpdStepCfg c = *cfg;
    if (c.lanes_per_wave == 0) c.lanes_per_wave = 64;
    if (c.lanes_per_wave != 16 && c.lanes_per_wave != 32 && c.lanes_per_wave != 64)
        return bad(GPD_EINVAL, "lanes_per_wave must be 0, 16, 32 or 64");
    const int min_lanes = multi ? (kBlock / cfg->drones_per_env) * cfg->drones_per_env : (kBlock / 64) * c.lanes_per_wave;
    if ((N + min_lanes - 1) / min_lanes > 0x7fffffffLL) return bad(GPD_ERANGE, "too many drones for one launch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ext = cfg->physics_flags != 0;
    // task NONE never uses the target: hand the kernel a readable dummy so that its load section is branch-free
    if (cfg->task == GPD_TASK_NONE) { target_pos = state->kin; c.target_per_env = 0; }
    hipError_t e;
#define GPD_LAUNCH(PID_, EXT_, AW_, ACT_)                                                                            \
    launch_step<PID_, EXT_, AW_, ACT_>(multi, st, *params, *state, c, T, action, target_pos, init_pose, obs12, reward, \
                                       terminated, truncated, term_obs12, done)
    switch (cfg->act_type) {
        case GPD_ACT_PID: e = ext ? GPD_LAUNCH(true, true, 3, GPD_ACT_PID) : GPD_LAUNCH(true, false, 3, GPD_ACT_PID); break;
        case GPD_ACT_VEL: e = ext ? GPD_LAUNCH(true, true, 4, GPD_ACT_VEL) : GPD_LAUNCH(true, false, 4, GPD_ACT_VEL); break;
        case GPD_ACT_ONE_D_PID:
            e = ext ? GPD_LAUNCH(true, true, 1, GPD_ACT_ONE_D_PID) : GPD_LAUNCH(true, false, 1, GPD_ACT_ONE_D_PID); break;
        case GPD_ACT_ONE_D_RPM:
            e = ext ? GPD_LAUNCH(false, true, 1, GPD_ACT_ONE_D_RPM) : GPD_LAUNCH(false, false, 1, GPD_ACT_ONE_D_RPM); break;
        case GPD_ACT_RAW_RPM:
            e = ext ? GPD_LAUNCH(false, true, 4, GPD_ACT_RAW_RPM) : GPD_LAUNCH(false, false, 4, GPD_ACT_RAW_RPM); break;
        case GPD_ACT_DIRECT_RPM:
            e = ext ? GPD_LAUNCH(false, true, 4, GPD_ACT_DIRECT_RPM) : GPD_LAUNCH(false, false, 4, GPD_ACT_DIRECT_RPM); break;
        default: e = ext ? GPD_LAUNCH(false, true, 4, GPD_ACT_RPM) : GPD_LAUNCH(false, false, 4, GPD_ACT_RPM); break;
    }
#undef GPD_LAUNCH
    if (e != hipSuccess) return hip_fail(e, who);
    return 0;
}

}  // namespace

// The DSLPID variants of the policy kernel are instantiated HERE, under this unit's scheduler (gpd_common.inc says why);
// GPD_PID_POLICY_IN_POLICY_TU (experiment / regression switch, tests/test_kernel_isa.py) moves them to policy.hip
#ifndef GPD_PID_POLICY_IN_POLICY_TU
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

GPD_DBG_READER(gpd_detail_dbg_read_step)

extern "C" {

int gpd_step(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const float* action,
             const float* target_pos, const float* init_pose, float* obs12, float* reward, uint8_t* terminated,
             uint8_t* truncated, float* term_obs12, void* stream) {
    const Span T{1, 0, 0, 0, 2};
    return step_impl("gpd_step", params, state, cfg, T, action, target_pos, init_pose, obs12, reward, terminated,
                     truncated, term_obs12, stream);
}

int gpd_step_sync(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const float* action,
                  const float* target_pos, const float* init_pose, float* obs12, float* reward, uint8_t* terminated,
                  uint8_t* truncated, float* term_obs12, void* stream) {
    GpdDone done = gpd_detail_done_begin();              // (how the wait works: gpd_common.inc)
    const Span T{1, 0, 0, 0, 2};
    if (int rc = step_impl("gpd_step_sync", params, state, cfg, T, action, target_pos, init_pose, obs12, reward, terminated,
                           truncated, term_obs12, stream, done.flag != nullptr ? &done : nullptr))
        return rc;
    return gpd_detail_done_wait(done, stream, "gpd_step_sync");
}

int gpd_rollout(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, int32_t num_steps,
                const float* actions, int64_t action_step_stride, const float* target_pos, const float* init_pose,
                float* obs12, int64_t obs_step_stride, float* reward, uint8_t* terminated, uint8_t* truncated,
                int64_t env_step_stride, float* term_obs12, void* stream) {
    if (num_steps <= 0) return fail(GPD_EINVAL, "gpd_rollout: num_steps must be positive");
    if (action_step_stride < 0 || obs_step_stride < 0 || env_step_stride < 0)
        return fail(GPD_EINVAL, "gpd_rollout: strides must be non-negative");
    const Span T{num_steps, action_step_stride, obs_step_stride, env_step_stride, 2};
    // a rollout never pushes into the action ring itself (gpd_full_obs does, after the call) -- also not when a one-step
    // rollout is routed to the single-step kernel
    GpdState no_ring;
    if (state) { no_ring = *state; no_ring.act_ring = nullptr; }
    return step_impl("gpd_rollout", params, state ? &no_ring : nullptr, cfg, T, actions, target_pos, init_pose, obs12, reward,
                     terminated, truncated, term_obs12, stream);
}

int gpd_rollout_history(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, int32_t num_steps,
                        const float* actions, int64_t action_step_stride, const float* target_pos, const float* init_pose,
                        float* obs12, int64_t obs_step_stride, float* reward, uint8_t* terminated, uint8_t* truncated,
                        int64_t env_step_stride, void* stream) {
    if (num_steps <= 0) return fail(GPD_EINVAL, "gpd_rollout_history: num_steps must be positive");
    if (action_step_stride < 0 || obs_step_stride < 0 || env_step_stride < 0)
        return fail(GPD_EINVAL, "gpd_rollout_history: strides must be non-negative");
    if (!state || !state->act_ring || !state->ring_pos || state->hist_len <= 0)
        return fail(GPD_EINVAL, "gpd_rollout_history: state has no action ring (act_ring / ring_pos / hist_len)");
    if (cfg && cfg->drones_per_env > 64)
        return fail(GPD_ENOTSUP, "gpd_rollout_history: aviaries of up to 64 drones (use gpd_rollout + gpd_full_obs otherwise)");
    if (getenv("GPD_ROLLOUT_STOREWAVE")) return fail(GPD_ENOTSUP, "gpd_rollout_history: not with the GPD_ROLLOUT_STOREWAVE diagnostic");
    const Span T{num_steps, action_step_stride, obs_step_stride, env_step_stride, 2};
    return step_impl("gpd_rollout_history", params, state, cfg, T, actions, target_pos, init_pose, obs12, reward, terminated,
                     truncated, nullptr, stream);
}

}  // extern "C"

