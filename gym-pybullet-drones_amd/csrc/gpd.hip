// gpd.hip — the MI355X (gfx950 / CDNA4) hot path of the vectorised quadrotor simulator.
//
// One wavefront lane per drone.  A lane loads its 13 kinematic floats from the structure-of-arrays
// state (each field a contiguous float[N] => a wave's load of one field is one coalesced 256 B
// transaction), keeps them in VGPRs across all physics sub-steps -- and, in a rollout, across all K env
// steps of the launch -- and stores them back once; the per-airframe constants arrive BY VALUE in the
// kernel-argument segment and therefore live in SGPRs (they are wave-uniform).  No MFMA: there is no
// dense contraction on this path.
//
// Kernels:
//   gpd_step_kernel      one env step per launch                                  (gpd_step)
//   gpd_rollout1_kernel  K env steps per launch, aviaries of 1, 2, 4 .. 64 drones (gpd_rollout)
//   gpd_rollout_kernel   K env steps per launch, compute waves + a store wave     (gpd_rollout: other aviary
//                        sizes, terminal observations)
//   gpd_hist_rows_kernel / gpd_hist_push_kernel  action ring + full KIN rows      (gpd_hist_rows, gpd_full_obs)
//   dwg_*_kernel         downwash inside one aviary of any size, grid binning     (gpd_downwash_global)
//   gpd_reset_kernel, gpd_pid_kernel, gpd_state20_kernel
//
// What one env step fuses (env_step; reference = utiasDSL/gym-pybullet-drones, gym_pybullet_drones/...):
//   action -> RPM                      envs/BaseRLAviary.py:187-239, envs/CtrlAviary.py:140
//   DSLPID position + attitude loops   control/DSLPIDControl.py:187-259
//   S x { forces/torques, Euler eqn, semi-implicit Euler, exact quaternion update }
//                                      envs/BaseAviary.py:831-892
//   ground effect / drag / downwash    envs/BaseAviary.py:739-743, 771-774, 798-804 (inside the
//                                      explicit integrator)
//   quat -> rpy, world body rates      envs/BaseAviary.py:517-519, 873
//   obs12 = pos|rpy|vel|ang_v          envs/BaseRLAviary.py:314
//   reward / terminated / truncated    envs/HoverAviary.py:68-117, envs/MultiHoverAviary.py:75-130
//   step counter, same-step auto-reset envs/BaseAviary.py:382, 451-477
//
// Arithmetic: fp32, NO -ffast-math, FP contraction OFF with every fused multiply-add written as fmaf() (a drone's
// trajectory is bit-identical in every kernel variant, batch size and lane).  At N = 65 536 the kernels are
// bound by the instruction issue of one wave per SIMD, so the hot functions are written for few instructions
// at <= 2 ulp instead of calling the branchy IEEE/OCML versions:
//   1/x, sqrt, 1/sqrt      v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (1 ulp)
//   quaternion exponential cos(t) and sin(t)/t as even polynomials in t^2 (no sqrt, no division, no
//                          range reduction; |t| <= 1 rad per sub-step, exact OCML path beyond)
//   atan2 / asin           one odd minimax polynomial on [0,1] (abs err 7e-8) + octant fix-up
//   rotor thrusts          carried as deviations from the hover thrust (no cancellation near hover)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and enums only: the library itself is resolved at run time (dlopen), see Rccl below
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>

#include "gpd.h"

namespace {

// Floating-point contraction is OFF for the whole file and every fused multiply-add is written out as fmaf():
// which a*b+c pairs get fused would otherwise depend on the code around each inlined copy of a function, and the
// same drone would round differently in the single-step kernel, the rollout kernel and the multi-drone variants.
// With explicit FMAs a drone's trajectory is bit-identical in every kernel variant, batch size and lane.
#pragma clang fp contract(off)

#ifndef GPD_BLOCK
#define GPD_BLOCK 256
#endif
constexpr int kBlock = GPD_BLOCK;   // 256 = 4 wavefronts; one workgroup per CU fills all 4 SIMDs

}  // namespace
// The library is built from TWO translation units of this one source: gpd.hip itself, and gpd_policy.hip, which defines
// GPD_POLICY_TU and includes this file to compile only gpd_rollout_policy -- its kernel wants another instruction scheduler
// than the step / rollout kernels (-amdgpu-sched-strategy: max-ilp fills the packed-fp32 hazards of the physics; the default
// strategy is 10 % faster on the MFMA / activation mix of the policy, A/B in round 2).  The last-error string is shared.
std::string& gpd_detail_last_error();
// The DSLPID variants of the policy kernel are instantiated in the MAIN unit (max-ilp scheduler).  Compiled in the policy unit
// (default scheduler) at commit 7333ea5, the VEL variant returned `truncated` = 1 and a task reward for every aviary whatever the
// configuration said.  ROOT CAUSE (found in the assembly; reproduced again in round 4, scratch/exp_r04/
// README.md): a register-allocation defect of hipcc (ROCm 7.2.0, AMD clang 22.0.0git roc-7.2.0), not undefined behaviour in this source -- the entry
// block loads GpdStepCfg.task .. init_per_env with `s_load_dwordx8 s[48:55], s[0:1], 0x19c`, the next block loads sixteen
// GpdParams words with `s_load_dwordx16 s[36:51], s[0:1], 0x18` OVER the first four while they are live, and only then is "the
// configuration" saved with `v_writelane_b32 v164, s48..s55, 29..36`: the step loop reloads prop_y[1..3] and gnd_eff_coeff as
// task / xy_bound / z_bound / tilt_bound.  It depends on how the allocator splits that one live range, i.e. on everything
// around it: today's source compiles clean in BOTH units under BOTH schedulers.  Fences: tests/isa_spill_check.py (backward SGPR
// liveness over every kernel of both units: a kernel-argument tuple that is spilled whole after part of it was overwritten),
// tests/test_kernel_isa.py (also builds the policy unit WITH these variants -- GPD_PID_POLICY_IN_POLICY_TU -- under both
// schedulers and checks them), the bitwise cross-kernel tests with every argument word scrambled (tests/test_gpu_policy.py), and
// tests/test_gpu_policy.py::test_vel_policy_kernel_is_right_under_both_schedulers (the variant library, built on the GPU box).
struct GpdPolicyLaunch {
    const GpdParams* params; const GpdState* state; const GpdStepCfg* cfg; const void* span; const GpdPolicy* policy;
    const float* obs12_in; const float* target_pos; const float* init_pose; float* actions_out; float* obs12; float* reward;
    uint8_t* terminated; uint8_t* truncated; void* stream; unsigned grid; int hist; float* term_obs12;
};
void gpd_detail_launch_policy_pid(const GpdPolicyLaunch& a);
#ifndef GPD_POLICY_TU
std::string& gpd_detail_last_error() {
    thread_local std::string e;
    return e;
}
#endif
// ------------------------------------------------------------------------------------------------
// Debug-bounds build (-DGPD_DEBUG_BOUNDS; `_native.build(debug=True)` -> libgpd_debug.so, used through GPD_LIB): the kernels
// check every index they READ FROM MEMORY before they address with it -- ring positions, sort keys, slot -> row entries, wake-list
// entries and counts -- record the first violation in a device word (code, workgroup, offending value, number of violations) and
// clamp the index so that the launch stays inside its buffers; `gpd_debug_status` reads the record.  The reference has nothing of
// the kind (no sanitizer, no race or bounds checks; its numpy indexing raises IndexError instead).  A
// release build compiles the checks away and `gpd_debug_status` returns GPD_ENOTSUP.  Checks live in the main unit's kernels.
// ------------------------------------------------------------------------------------------------
#if defined(GPD_DEBUG_BOUNDS) && !defined(GPD_POLICY_TU)
__device__ unsigned int gpd_dbg_word[4];
#define GPD_DBG(cond, code, val)                                                                  \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            if (atomicCAS(&gpd_dbg_word[0], 0u, static_cast<unsigned int>(code)) == 0u) {         \
                gpd_dbg_word[1] = blockIdx.x;                                                     \
                gpd_dbg_word[2] = static_cast<unsigned int>(val);                                 \
            }                                                                                     \
            atomicAdd(&gpd_dbg_word[3], 1u);                                                      \
        }                                                                                         \
    } while (0)
#define GPD_DBG_CLAMP(v, lo, hi) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
#else
#define GPD_DBG(cond, code, val) do { } while (0)
#define GPD_DBG_CLAMP(v, lo, hi) (v)
#endif
namespace {
#define g_last_error gpd_detail_last_error()

int fail(int code, const char* msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char* where) {
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
    return static_cast<int>(e);
}

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
struct Mat3 {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22;
    float m22;     // 1 - r22, computed directly (no cancellation for a near-level attitude)
};

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// exp(x) for x <= 0 as 2^(x log2 e): v_exp_f32 (1 ulp) after one rounded product -- relative error <= (1 + |x|) 6e-8,
// two instructions instead of OCML's ~20.  Used for the downwash Gaussian (evaluated for every pair of an aviary).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// Packed fp32 (v_pk_mul/add/fma_f32): TWO IEEE operations per instruction.  With one wave per SIMD the step is bound by
// the rate at which a single wave issues instructions (~5.4 cycles each, whatever they are -- scratch/issue.hip), not by
// the SIMD's arithmetic rate, so pairing two independent operations of the same kind halves their cost; each half
// rounds exactly like the scalar instruction, so results do not change by a bit.  Used only where both halves are
// naturally adjacent (no register shuffles); -fno-slp-vectorize keeps the compiler from pairing on its own (its
// shuffles cost more issue slots than the pairs save).
typedef float fp2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fp2 fma2(fp2 a, fp2 b, fp2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ fp2 splat(float s) { return fp2{s, s}; }

// Rare per-lane cases (gimbal lock, a tumbling drone, an episode end) are entered through a WAVE-UNIFORM test marked
// unlikely: the hot path pays one compare and one not-taken scalar branch (~11 cycles, scratch/issue.hip), and the
// rare block sits out of line.  The plain divergent `if` the compiler would emit instead -- s_and_saveexec +
// s_cbranch_execz around an in-line block -- costs a TAKEN branch (~60 cycles: the instruction buffer refills) on
// every step, and with one wave per SIMD nothing hides it.
__device__ __forceinline__ uint64_t lane_mask_opaque(bool p) {
    uint64_t m = __builtin_amdgcn_ballot_w64(p);
    asm volatile("" : "+s"(m));      // opaque: otherwise `if (any_lane(p)) if (p)` is folded back into the divergent `if (p)`
    return m;
}
// (a macro: __builtin_expect has to sit in the `if` itself -- returned from an inlined function it is dropped before inlining)
#define any_lane(p) __builtin_expect(lane_mask_opaque(p) != 0, 0)

// The three functions below are inlined at several places of the step kernel (rpy at the top of a step for
// DSLPID, at its tail for the observation, after an auto-reset).  FMA contraction is switched OFF inside them
// and every fused operation is written out, so that each inlined copy rounds identically: a K-step rollout
// carries the tail's rpy into the next step where a single-step launch recomputes it at the top, and the two
// must agree bit for bit.

// atan2 with the IEEE sign/quadrant conventions Bullet's Euler extraction relies on; minimax
// polynomial for atan(t)/t in t^2 on [0,1] (max abs error 7.4e-8 evaluated in fp32)
__device__ __forceinline__ float atan2_poly(float y, float x) {
#pragma clang fp contract(off)
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float t = mn * fast_rcp(mx);
    t = (mx == 0.0f) ? 0.0f : t;                 // atan2(0, 0) = 0
    const float u = t * t;
    float p = 2.766283504e-03f;
    p = fmaf(p, u, -1.573124913e-02f);
    p = fmaf(p, u, 4.213762361e-02f);
    p = fmaf(p, u, -7.456854827e-02f);
    p = fmaf(p, u, 1.061837064e-01f);
    p = fmaf(p, u, -1.419779779e-01f);
    p = fmaf(p, u, 1.999187203e-01f);
    p = fmaf(p, u, -3.333303671e-01f);
    p = fmaf(p, u, 9.999999818e-01f);
    p *= t;
    p = (ay > ax) ? (1.57079632679489661923f - p) : p;
    p = (x < 0.0f) ? (3.14159265358979323846f - p) : p;
    return copysignf(p, y);
}

// two independent atan2 at once: the same operations as atan2_poly, the multiplies and the Horner chain as packed pairs
__device__ __forceinline__ fp2 atan2_poly2(fp2 y, fp2 x) {
#pragma clang fp contract(off)
    const float ax0 = fabsf(x.x), ay0 = fabsf(y.x), ax1 = fabsf(x.y), ay1 = fabsf(y.y);
    const float mx0 = fmaxf(ax0, ay0), mn0 = fminf(ax0, ay0), mx1 = fmaxf(ax1, ay1), mn1 = fminf(ax1, ay1);
    fp2 t = fp2{mn0, mn1} * fp2{fast_rcp(mx0), fast_rcp(mx1)};
    t = fp2{(mx0 == 0.0f) ? 0.0f : t.x, (mx1 == 0.0f) ? 0.0f : t.y};   // atan2(0, 0) = 0
    const fp2 u = t * t;
    fp2 p = splat(2.766283504e-03f);
    p = fma2(p, u, splat(-1.573124913e-02f));
    p = fma2(p, u, splat(4.213762361e-02f));
    p = fma2(p, u, splat(-7.456854827e-02f));
    p = fma2(p, u, splat(1.061837064e-01f));
    p = fma2(p, u, splat(-1.419779779e-01f));
    p = fma2(p, u, splat(1.999187203e-01f));
    p = fma2(p, u, splat(-3.333303671e-01f));
    p = fma2(p, u, splat(9.999999818e-01f));
    p = p * t;
    float p0 = p.x, p1 = p.y;
    p0 = (ay0 > ax0) ? (1.57079632679489661923f - p0) : p0;
    p1 = (ay1 > ax1) ? (1.57079632679489661923f - p1) : p1;
    p0 = (x.x < 0.0f) ? (3.14159265358979323846f - p0) : p0;
    p1 = (x.y < 0.0f) ? (3.14159265358979323846f - p1) : p1;
    return fp2{copysignf(p0, y.x), copysignf(p1, y.y)};
}

// asin(s) = atan2(s, sqrt((1-s)(1+s))); the factored form keeps full relative accuracy near |s| = 1
__device__ __forceinline__ float asin_poly(float s) {
#pragma clang fp contract(off)
    return atan2_poly(s, fast_sqrt(fmaxf((1.0f - s) * (1.0f + s), 0.0f)));
}

// btMatrix3x3::setRotation (reached via p.getMatrixFromQuaternion, envs/BaseAviary.py:836);
// insensitive to |q| (DYN never renormalises q)
__device__ __forceinline__ Mat3 quat_to_mat(float x, float y, float z, float w) {
    const float d = fmaf(x, x, fmaf(y, y, fmaf(z, z, w * w)));
    const float s = 2.0f * fast_rcp(d);
    const fp2 xys = fp2{x, y} * splat(s);                     // (packed pairs, see fma2)
    const float xs = xys.x, ys = xys.y, zs = z * s;
    const fp2 wxy = splat(w) * xys;
    const float wx = wxy.x, wy = wxy.y, wz = w * zs;
    Mat3 R;
    R.r00 = 1.0f - fmaf(y, ys, z * zs); R.r01 = fmaf(x, ys, -wz);           R.r02 = fmaf(x, zs, wy);
    R.r10 = fmaf(x, ys, wz);            R.r11 = 1.0f - fmaf(x, xs, z * zs); R.r12 = fmaf(y, zs, -wx);
    R.m22 = fmaf(x, xs, y * ys);
    R.r20 = fmaf(x, zs, -wy);           R.r21 = fmaf(y, zs, wx);            R.r22 = 1.0f - R.m22;
    return R;
}

// pybullet_getEulerFromQuaternion incl. its gimbal branches (envs/BaseAviary.py:518).  The regular branch is
// evaluated straight-line for every lane; the two gimbal branches (|sarg| >= 0.99999, i.e. pitch within 0.26 deg
// of +-90 deg) are a fix-up that is skipped unless some lane of the wave needs it.
__device__ __forceinline__ void quat_to_rpy(float x, float y, float z, float w,
                                            float& roll, float& pitch, float& yaw) {
#pragma clang fp contract(off)
    const float sarg = -2.0f * fmaf(x, z, -(w * y));
    const float ww_zz = fmaf(w, w, z * z);                       // squ + sqz
    const float xx_yy = fmaf(x, x, y * y);                       // sqx + sqy
    const float ww_yy = fmaf(w, w, -(y * y));                    // squ - sqy
    const float xx_zz = fmaf(x, x, -(z * z));                    // sqx - sqz
    // roll = atan2(2(yz + wx), squ - sqx - sqy + sqz), yaw = atan2(2(xy + wz), squ + sqx - sqy - sqz): evaluated together
    const fp2 ry = atan2_poly2(splat(2.0f) * fp2{fmaf(y, z, w * x), fmaf(x, y, w * z)}, fp2{ww_zz - xx_yy, ww_yy + xx_zz});
    roll = ry.x;
    pitch = asin_poly(fminf(fmaxf(sarg, -1.0f), 1.0f));          // (clamp: only matters on the gimbal lanes, overwritten below)
    yaw = ry.y;
    const bool gimbal = fabsf(sarg) >= 0.99999f;
    if (any_lane(gimbal)) {
        asm volatile("; gimbal lock: rare");                // (keeps the wave-uniform branch from being merged with the lane test)
        if (gimbal) {
            const bool neg = sarg < 0.0f;
            roll = 0.0f;
            pitch = neg ? -1.57079632679489661923f : 1.57079632679489661923f;
            yaw = 2.0f * (neg ? atan2_poly(x, -y) : atan2_poly(-x, y));
        }
    }
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    return fminf(fmaxf(v, lo), hi);   // np.clip
}

struct Kin {   // one drone's integrator state, all in VGPRs
    float px, py, pz, qx, qy, qz, qw, vx, vy, vz, wx, wy, wz;
};

struct Pid {   // DSLPIDControl members, control/DSLPIDControl.py:73-78
    float ipx, ipy, ipz, lr, lp, ly, irx, iry, irz;
};

// DSLPIDControl.computeControl (control/DSLPIDControl.py:82-145).
// Returns the four RPMs; optionally the desired-vs-current yaw needed by the standalone entry.
__device__ __forceinline__ void dslpid(const GpdParams& P, float dt, float inv_dt, const Kin& k, float roll, float pitch,
                                       float yaw, const Mat3& R, float tx, float ty, float tz, float tyaw,
                                       float tvx, float tvy, float tvz, float trr, float trp, float try_,
                                       Pid& s, float rpm[4], float pos_e[3], float* yaw_e) {
    // ---- position loop, :187-203
    const float epx = tx - k.px, epy = ty - k.py, epz = tz - k.pz;
    const float evx = tvx - k.vx, evy = tvy - k.vy, evz = tvz - k.vz;
    s.ipx = clampf(fmaf(epx, dt, s.ipx), -2.0f, 2.0f);
    s.ipy = clampf(fmaf(epy, dt, s.ipy), -2.0f, 2.0f);
    s.ipz = clampf(clampf(fmaf(epz, dt, s.ipz), -2.0f, 2.0f), -0.15f, 0.15f);
    const float fx = fmaf(P.d_for[0], evx, fmaf(P.i_for[0], s.ipx, P.p_for[0] * epx));
    const float fy = fmaf(P.d_for[1], evy, fmaf(P.i_for[1], s.ipy, P.p_for[1] * epy));
    const float fz = fmaf(P.d_for[2], evz, fmaf(P.i_for[2], s.ipz, P.p_for[2] * epz)) + P.pid_gravity;
    const float along = fmaxf(0.0f, fmaf(fz, R.r22, fmaf(fy, R.r12, fx * R.r02)));
    const float base_pwm = (fast_sqrt(along * P.pid_inv_4kf) - P.pwm2rpm_const) * P.inv_pwm2rpm_scale;
    const float fn = fast_rsq(fmaf(fz, fz, fmaf(fy, fy, fx * fx)));
    const float zbx = fx * fn, zby = fy * fn, zbz = fz * fn;
    float sy, cy;
    sincosf(tyaw, &sy, &cy);                      // heading = [cos, sin, 0]
    float ybx = -(zbz * sy);                      // zb x heading
    float yby = zbz * cy;
    float ybz = fmaf(zbx, sy, -(zby * cy));
    const float yn = fast_rsq(fmaf(ybz, ybz, fmaf(yby, yby, ybx * ybx)));
    ybx *= yn; yby *= yn; ybz *= yn;
    const float xbx = fmaf(yby, zbz, -(ybz * zby));   // yb x zb
    const float xby = fmaf(ybz, zbx, -(ybx * zbz));
    const float xbz = fmaf(ybx, zby, -(yby * zbx));
    // ---- attitude loop, :240-259.  The reference rebuilds R* from its Euler angles through scipy;
    // that round trip returns the same orthonormal matrix (to 3e-16), so R* = [xb yb zb] is used.
    // e_R = vee(R*^T R - R^T R*): M_ij = col_i(R*) . col_j(R)
    const float m21 = fmaf(zbz, R.r21, fmaf(zby, R.r11, zbx * R.r01)), m12 = fmaf(ybz, R.r22, fmaf(yby, R.r12, ybx * R.r02));
    const float m02 = fmaf(xbz, R.r22, fmaf(xby, R.r12, xbx * R.r02)), m20 = fmaf(zbz, R.r20, fmaf(zby, R.r10, zbx * R.r00));
    const float m10 = fmaf(ybz, R.r20, fmaf(yby, R.r10, ybx * R.r00)), m01 = fmaf(xbz, R.r21, fmaf(xby, R.r11, xbx * R.r01));
    const float erx = m21 - m12, ery = m02 - m20, erz = m10 - m01;
    const float ewx = fmaf(-(roll - s.lr), inv_dt, trr);   // finite difference of Euler angles, no unwrap (:247)
    const float ewy = fmaf(-(pitch - s.lp), inv_dt, trp);
    const float ewz = fmaf(-(yaw - s.ly), inv_dt, try_);
    s.lr = roll; s.lp = pitch; s.ly = yaw;
    s.irx = clampf(clampf(fmaf(-erx, dt, s.irx), -1500.0f, 1500.0f), -1.0f, 1.0f);
    s.iry = clampf(clampf(fmaf(-ery, dt, s.iry), -1500.0f, 1500.0f), -1.0f, 1.0f);
    s.irz = clampf(fmaf(-erz, dt, s.irz), -1500.0f, 1500.0f);
    const float t0 = clampf(fmaf(P.i_tor[0], s.irx, fmaf(P.d_tor[0], ewx, -(P.p_tor[0] * erx))), -3200.0f, 3200.0f);
    const float t1 = clampf(fmaf(P.i_tor[1], s.iry, fmaf(P.d_tor[1], ewy, -(P.p_tor[1] * ery))), -3200.0f, 3200.0f);
    const float t2 = clampf(fmaf(P.i_tor[2], s.irz, fmaf(P.d_tor[2], ewz, -(P.p_tor[2] * erz))), -3200.0f, 3200.0f);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float pwm = clampf(fmaf(P.mixer[3 * m + 2], t2, fmaf(P.mixer[3 * m + 1], t1, fmaf(P.mixer[3 * m], t0, base_pwm))),
                                 P.min_pwm, P.max_pwm);
        rpm[m] = fmaf(P.pwm2rpm_scale, pwm, P.pwm2rpm_const);
    }
    if (pos_e) { pos_e[0] = epx; pos_e[1] = epy; pos_e[2] = epz; }
    if (yaw_e) {
        // yaw of the intrinsic-XYZ Euler angles of R* (scipy as_euler('XYZ'), :205): atan2(-R01, R00)
        *yaw_e = atan2_poly(-ybx, xbx) - yaw;
    }
}

// Thrust of one rotor as a deviation from the hover thrust F_h = GRAVITY/4 = KF*HOVER_RPM^2:
//   g = KF*rpm^2 - F_h = KF*(rpm - h)*(rpm + h) + (KF*h^2 - F_h),   h = float(HOVER_RPM)
// rpm - h is exact in fp32 (Sterbenz) for any rpm a quadrotor flies at, so g keeps ~1e-7 RELATIVE accuracy where
// KF*rpm^2 - F_h computed naively keeps 1e-7 of F_h -- 20..1000x coarser near hover.
__device__ __forceinline__ float thrust_dev(const GpdParams& P, float rpm) {
    return fmaf(P.KF * (rpm - P.hover_rpm), rpm + P.hover_rpm, P.hover_resid);
}
// ... and for the normalised action types, rpm = HOVER_RPM*(1 + e) (envs/BaseRLAviary.py:191-192):
//   g = F_h*((1 + e)^2 - 1) = F_h * e * (2 + e), with no reference to the rounded rpm at all
__device__ __forceinline__ float thrust_dev_norm(const GpdParams& P, float e) {
    return P.hover_thrust * (e * (2.0f + e));
}

// One physics sub-step (envs/BaseAviary.py:831-877 + :879-892), state in registers.
//   g[4]        rotor thrusts minus the hover thrust (KF*rpm_i^2 - GRAVITY/4), constant over the sub-steps
//   drag_rpm_sum  sum of the rpm the drag term sees (previous action on sub-step 0), only if DRAG
//   dw_force    body-z downwash force on this drone (already summed over the drones above), only if DW
// Forces are assembled from the deviations: total thrust T = GRAVITY + sum(g) (+ ground effect, downwash), so
//   F_z = R22*T - GRAVITY = R22*(T - GRAVITY) - GRAVITY*(1 - R22)      (1 - R22 computed directly)
// and the torques only see differences of the g_i (F_h cancels analytically: every torque row sums to zero).
//   AV          also produce the world angular velocity (only the LAST sub-step of an env step is observed)
template <bool EXT, bool AV = true>
__device__ __forceinline__ void substep(const GpdParams& P, float h, uint32_t flags, const float g[4],
                                        float drag_rpm_sum, float dw_force, Kin& k,
                                        float& avx, float& avy, float& avz) {
    const Mat3 R = quat_to_mat(k.qx, k.qy, k.qz, k.qw);
    float f0 = g[0], f1 = g[1], f2 = g[2], f3 = g[3];      // thrust deviations (ground effect adds to them)
    if (EXT && (flags & GPD_PHYS_GND)) {
        // per-rotor extra thrust (:739-743): KF*rpm_i^2 * coeff * (r/(4 h_i))^2, h_i = world z of rotor i, clipped
        // (the four rotors as two packed pairs: the same operations in the same order per rotor, half the issue slots)
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            fp2 hz = fma2(splat(R.r21), fp2{P.prop_y[i], P.prop_y[i + 1]}, fma2(splat(R.r20), fp2{P.prop_x[i], P.prop_x[i + 1]}, splat(k.pz)));
            hz = fp2{fmaxf(hz.x, P.gnd_eff_h_clip), fmaxf(hz.y, P.gnd_eff_h_clip)};
            const fp2 ratio = splat(0.25f * P.prop_radius) * fp2{fast_rcp(hz.x), fast_rcp(hz.y)};
            const fp2 ei = ((splat(P.hover_thrust) + fp2{g[i], g[i + 1]}) * splat(P.gnd_eff_coeff)) * (ratio * ratio);
            e[i] = ei.x; e[i + 1] = ei.y;
        }
        // |roll| < pi/2 and |pitch| < pi/2 on Bullet's Euler extraction (:742), without the atan2/asin:
        // pitch = asin(sarg) is inside (-pi/2, pi/2) off the gimbal branches and exactly +-pi/2 on them;
        // roll = atan2(a, b) has |roll| < pi/2 iff b > 0 (or a == b == 0).  Same expressions as quat_to_rpy.
        const float sarg = -2.0f * fmaf(k.qx, k.qz, -(k.qw * k.qy));
        const float a = 2.0f * fmaf(k.qy, k.qz, k.qw * k.qx);
        const float b = fmaf(k.qw, k.qw, k.qz * k.qz) - fmaf(k.qx, k.qx, k.qy * k.qy);
        const bool on = (sarg > -0.99999f) && (sarg < 0.99999f) && (b > 0.0f || (a == 0.0f && b == 0.0f));
        if (on) { f0 += e[0]; f1 += e[1]; f2 += e[2]; f3 += e[3]; }
    }
    float dev = ((f0 + f1) + f2) + f3;                     // total thrust minus GRAVITY
    if (EXT && (flags & GPD_PHYS_DW)) dev += dw_force;
    const float T = P.GRAVITY + dev;
    const fp2 Fxy = fp2{R.r02, R.r12} * splat(T);
    float Fx = Fxy.x, Fy = Fxy.y, Fz = fmaf(R.r22, dev, -(P.GRAVITY * R.m22));
    if (EXT && (flags & GPD_PHYS_DRAG)) {
        // world force -DRAG_COEFF * v * sum(2*pi*rpm/60) (:771-774; R R^T cancels)
        const float wsum = drag_rpm_sum * (6.28318530717958647692f / 60.0f);
        Fx = fmaf(-(P.drag_coeff[0] * k.vx), wsum, Fx);
        Fy = fmaf(-(P.drag_coeff[1] * k.vy), wsum, Fy);
        Fz = fmaf(-(P.drag_coeff[2] * k.vz), wsum, Fz);
    }
    if (EXT && (flags & GPD_PHYS_DAMP)) {
        // Bullet's default multibody damping (NOT in the reference's Physics.DYN, see gpd.h): force -M d (1 + |v|) v
        const float vn = fast_sqrt(fmaf(k.vz, k.vz, fmaf(k.vy, k.vy, k.vx * k.vx)));
        const float ml = -(P.M * GPD_BULLET_DAMPING) * (1.0f + vn);
        Fx = fmaf(ml, k.vx, Fx); Fy = fmaf(ml, k.vy, Fy); Fz = fmaf(ml, k.vz, Fz);
    }
    // z torque from KM*rpm_i^2 = (KM/KF)*(F_h + g_i): alternating signs, F_h cancels (ground effect not included, :842).
    // The airframe variants are folded into signed constants and one bit-select: no branch and no per-step select chain
    // (negating a factor negates the product exactly, so the bits are those of "compute, then flip the sign").
    const float kz = (P.drone_model == GPD_MODEL_RACE) ? -P.km_over_kf : P.km_over_kf;
    const float tz = kz * (((-g[0] + g[1]) - g[2]) + g[3]);
    const float arm = P.L * 0.70710678118654752440f;        // L / sqrt(2)
    const float arm_x = (P.drone_model == GPD_MODEL_CF2X) ? -arm : arm;
    const float tx_x = (((f0 + f1) - f2) - f3) * arm_x, ty_x = (((-f0 + f1) + f2) - f3) * arm;   // CF2X / RACE
    const float tx_p = (f1 - f3) * P.L, ty_p = (-f0 + f2) * P.L;                                 // CF2P
    const uint32_t plus = (P.drone_model == GPD_MODEL_CF2P) ? 0xffffffffu : 0u;
    float tx = __uint_as_float((__float_as_uint(tx_p) & plus) | (__float_as_uint(tx_x) & ~plus));
    float ty = __uint_as_float((__float_as_uint(ty_p) & plus) | (__float_as_uint(ty_x) & ~plus));
    // Euler's rotation equation with diagonal J
    const fp2 jwxy = fp2{P.J[0], P.J[1]} * fp2{k.wx, k.wy};
    const float jwx = jwxy.x, jwy = jwxy.y, jwz = P.J[2] * k.wz;
    fp2 txy = fp2{tx, ty} - fp2{fmaf(k.wy, jwz, -(k.wz * jwy)), fmaf(k.wz, jwx, -(k.wx * jwz))};
    float tzz = tz - fmaf(k.wx, jwy, -(k.wy * jwx));
    if (EXT && (flags & GPD_PHYS_DAMP)) {                  // ... and torque -J w d (1 + |w|)
        const float wn = fast_sqrt(fmaf(k.wz, k.wz, fmaf(k.wy, k.wy, k.wx * k.wx)));
        const float da = -GPD_BULLET_DAMPING * (1.0f + wn);
        txy = fma2(jwxy, splat(da), txy);
        tzz = fmaf(jwz, da, tzz);
    }
    // semi-implicit Euler (:860-862): position uses the NEW velocity; x and y as packed pairs
    const fp2 hh = splat(h);
    const fp2 vxy = fma2(hh, fp2{Fx, Fy} * splat(P.inv_M), fp2{k.vx, k.vy});
    k.vx = vxy.x; k.vy = vxy.y; k.vz = fmaf(h, Fz * P.inv_M, k.vz);
    const fp2 wxy = fma2(hh, fp2{P.J_INV[0], P.J_INV[1]} * txy, fp2{k.wx, k.wy});
    k.wx = wxy.x; k.wy = wxy.y; k.wz = fmaf(h, P.J_INV[2] * tzz, k.wz);
    const fp2 pxy = fma2(hh, vxy, fp2{k.px, k.py});
    k.px = pxy.x; k.py = pxy.y; k.pz = fmaf(h, k.vz, k.pz);
    if (EXT && (flags & GPD_PHYS_GROUND)) {
        // the ground plane (NOT in the reference's Physics.DYN, see gpd.h): a drone whose collision cylinder would sink
        // below z = 0 is put back on the plane, loses its downward velocity (restitution 0) and sticks laterally
        // (second clause: a drone RESTING on the plane whose downward velocity is too small to move its fp32 height, h |vz| below
        // half an ulp of ground_z, would otherwise be neither caught nor free -- it slid laterally, unstuck, until vz had grown;
        // in exact arithmetic the clause only decides the tie pz == ground_z)
        const bool hit = (k.pz < P.ground_z) | ((k.pz <= P.ground_z) & (k.vz < 0.0f));
        k.pz = hit ? P.ground_z : k.pz;
        k.vz = hit ? fmaxf(k.vz, 0.0f) : k.vz;
        k.vx = hit ? 0.0f : k.vx;
        k.vy = hit ? 0.0f : k.vy;
    }
    // exact exponential quaternion update q <- q (x) exp(w h / 2)  (:879-892)
    //   q' = cos(t) q + (sin(t)/|w|) (q (x) [w,0]),  t = |w| h / 2.  cos(t) and sin(t)/t are even functions
    //   of t, evaluated as minimax polynomials in u = t^2 (abs err 7e-8 / 5e-8 for t <= 1 rad, i.e. body
    //   rates up to 480 rad/s at 240 Hz): no sqrt, no division, no range reduction.
    const float n2 = fmaf(k.wz, k.wz, fmaf(k.wy, k.wy, k.wx * k.wx));
    const float u = n2 * (0.25f * h * h);
    const fp2 uu = splat(u);                                  // (cos t, sin t / t): the two Horner chains as one packed chain
    const fp2 csp = fma2(fma2(fma2(fma2(fp2{2.412107309e-05f, 2.693749890e-06f}, uu, fp2{-1.388295778e-03f, -1.983586443e-04f}), uu,
                                  fp2{4.166645522e-02f, 8.333314057e-03f}), uu, fp2{-4.999999736e-01f, -1.666666643e-01f}), uu,
                        fp2{9.999999995e-01f, 1.0f});
    float cs = csp.x;
    float sc = csp.y * (0.5f * h);
    const bool tumbling = !(u <= 1.0f);                    // faster than 480 rad/s (at 240 Hz): exact path
    if (any_lane(tumbling)) {
        asm volatile("; tumbling: rare");
        if (tumbling) {
            const float n = sqrtf(n2);
            float sn;
            sincosf(n * h * 0.5f, &sn, &cs);
            sc = sn / n;
        }
    }
    {
        const float lx = fmaf(k.wx, k.qw, fmaf(k.wz, k.qy, -(k.wy * k.qz)));
        const float ly = fmaf(k.wy, k.qw, fmaf(k.wx, k.qz, -(k.wz * k.qx)));
        const float lz = fmaf(k.wz, k.qw, fmaf(k.wy, k.qx, -(k.wx * k.qy)));
        const float lw = -fmaf(k.wz, k.qz, fmaf(k.wy, k.qy, k.wx * k.qx));
        const bool turn = n2 > 1e-16f;                     // !np.isclose(|w|, 0): |w| <= 1e-8 keeps q  (select, no branch)
        const fp2 nxy = fma2(splat(sc), fp2{lx, ly}, splat(cs) * fp2{k.qx, k.qy});
        const fp2 nzw = fma2(splat(sc), fp2{lz, lw}, splat(cs) * fp2{k.qz, k.qw});
        k.qx = turn ? nxy.x : k.qx; k.qy = turn ? nxy.y : k.qy;
        k.qz = turn ? nzw.x : k.qz; k.qw = turn ? nzw.y : k.qw;
    }
    // world angular velocity handed to the state store: PRE-update rotation, post-update rates (:873)
    if (AV) {
        const fp2 axy = fma2(fp2{R.r02, R.r12}, splat(k.wz), fma2(fp2{R.r01, R.r11}, splat(k.wy), fp2{R.r00, R.r10} * splat(k.wx)));
        avx = axy.x; avy = axy.y;
        avz = fmaf(R.r22, k.wz, fmaf(R.r21, k.wy, R.r20 * k.wx));
    }
}

// SoA row access as  <uniform 64-bit row base in SGPRs> + <32-bit per-lane byte offset>: this is the
// global_load/store "saddr + voffset" form, one VGPR of address for all rows instead of a 64-bit
// add per access.  (gpd_step bounds N so that every byte offset fits 32 bits.)
__device__ __forceinline__ float ld_row(const float* __restrict__ base, int64_t ld, int r, uint32_t off4) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + r * ld) + off4);
}
template <bool NT = false>   // NT: non-temporal (a working set far beyond the caches is streamed, not kept)
__device__ __forceinline__ void st_row(float* __restrict__ base, int64_t ld, int r, uint32_t off4, float v) {
    float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(base + r * ld) + off4);
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__device__ __forceinline__ void store_obs12(float* __restrict__ obs, uint32_t n, float px, float py, float pz,
                                            float roll, float pitch, float yaw, float vx, float vy, float vz,
                                            float ax, float ay, float az) {
    float4* o = reinterpret_cast<float4*>(reinterpret_cast<char*>(obs) + n * 48u);
    o[0] = make_float4(px, py, pz, roll);
    o[1] = make_float4(pitch, yaw, vx, vy);
    o[2] = make_float4(vz, ax, ay, az);
}

// ------------------------------------------------------------------------------------------------
// One env step of one drone, everything in registers.  Shared by the single-step kernel (gpd_step) and the
// rollout kernel (gpd_rollout): same statements in the same order, so K rollout steps are bitwise K single
// steps.
//   PID   : action types that run DSLPID (PID / VEL / ONE_D_PID)
//   EXT   : any of the GND/DRAG/DW terms may be enabled (flags tested at run time, uniformly)
//   MULTI : drones_per_env > 1 (env-level reductions and downwash go through LDS + workgroup barriers)
// ------------------------------------------------------------------------------------------------
// Workgroup barrier for LDS hand-offs: waits for this wave's outstanding LDS operations only (lgkmcnt), not for
// its global stores -- __syncthreads() would add a release fence, and with it a wait for every store in flight.
__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0); vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
}

struct Lane {             // which drone a lane works on
    uint32_t n;           // drone index (0 for a lane without a drone: it computes on drone 0 and stores nothing)
    uint32_t env;         // aviary index
    int tid;              // compute-lane index inside the workgroup (0 .. kBlock-1)
    int le, d;            // aviary / drone-in-aviary inside the workgroup
    bool active;
    bool shfl;            // MULTI: the D drones of an aviary are D aligned lanes of ONE wave (D a power of two <= 64):
                          // they exchange through a wave-local LDS patch, without workgroup barriers
};

struct Carry {            // what a drone carries from one env step to the next
    Kin k;
    Pid s;                // DSLPID members
    float l0, l1, l2, l3; // last_clipped_action
    int counter;          // the aviary's step counter
    float dw_in;          // downwash force computed outside the kernel (one aviary of > 256 drones), else 0
    float roll, pitch, yaw;   // rpy of the cached pose (BaseAviary.py:518), carried from the tail of one step to
                              // the top of the next (DSLPID reads it there)
};

struct StepOut {          // what one env step hands to the stores
    float o[12];          // observation row: pos | rpy | vel | ang_v (of the reset pose if the aviary was reset)
    float to[12];         // last observation of the finished episode (meaningful only if `reset`)
    float rew;
    bool term, trunc, reset;
};

// action -> RPM (computed ONCE per env step from the cached state, BaseAviary.py:341) and the rotor thrusts minus the
// hover thrust.  DSLPID action types advance the controller members in c.s and read the cached rpy in c.
template <bool PID, int AW, int ACT>
__device__ __forceinline__ void map_action(const GpdParams& P, const GpdStepCfg& C, const float4 act, Carry& c,
                                           float rpm[4], float g[4]) {
    const Kin& k = c.k;
    const int act_type = ACT >= 0 ? ACT : C.act_type;
    rpm[0] = rpm[1] = rpm[2] = rpm[3] = 0.0f;
    if (!PID) {
        if (AW == 1) {     // GPD_ACT_ONE_D_RPM
            const float e = 0.05f * act.x;
            rpm[0] = rpm[1] = rpm[2] = rpm[3] = fmaf(P.hover_rpm, e, P.hover_rpm);
            g[0] = g[1] = g[2] = g[3] = thrust_dev_norm(P, e);
        } else if (act_type == GPD_ACT_RPM) {   // NOT clipped
            // rpm = HOVER_RPM*(1 + 0.05 a), g = thrust_dev_norm(0.05 a): rotors (0,1) and (2,3) as packed pairs
            const fp2 e01 = splat(0.05f) * fp2{act.x, act.y}, e23 = splat(0.05f) * fp2{act.z, act.w};
            const fp2 r01 = fma2(splat(P.hover_rpm), e01, splat(P.hover_rpm)), r23 = fma2(splat(P.hover_rpm), e23, splat(P.hover_rpm));
            const fp2 g01 = splat(P.hover_thrust) * (e01 * (splat(2.0f) + e01)), g23 = splat(P.hover_thrust) * (e23 * (splat(2.0f) + e23));
            rpm[0] = r01.x; rpm[1] = r01.y; rpm[2] = r23.x; rpm[3] = r23.y;
            g[0] = g01.x; g[1] = g01.y; g[2] = g23.x; g[3] = g23.y;
        } else {           // GPD_ACT_RAW_RPM (clipped to [0, MAX_RPM], envs/CtrlAviary.py:140) or GPD_ACT_DIRECT_RPM (as is)
            const bool clip = act_type == GPD_ACT_RAW_RPM;
            const float lo = clip ? 0.0f : -3.0e38f, hi = clip ? P.max_rpm : 3.0e38f;
            rpm[0] = clampf(act.x, lo, hi); rpm[1] = clampf(act.y, lo, hi);
            rpm[2] = clampf(act.z, lo, hi); rpm[3] = clampf(act.w, lo, hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = thrust_dev(P, rpm[i]);
        }
    } else {
        float tx = k.px, ty = k.py, tz = k.pz, tyaw = 0.0f, tvx = 0.0f, tvy = 0.0f, tvz = 0.0f;
        if (AW == 3) {     // GPD_ACT_PID
            // waypoint limited to a 1 m approach step (_calculateNextStep, BaseAviary.py:1132-1150)
            const float dx = act.x - k.px, dy = act.y - k.py, dz = act.z - k.pz;
            const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (fast_sqrt(d2) <= 1.0f) { tx = act.x; ty = act.y; tz = act.z; }
            else { const float id = fast_rsq(d2); tx = fmaf(dx, id, k.px); ty = fmaf(dy, id, k.py); tz = fmaf(dz, id, k.pz); }
        } else if (AW == 4) {   // GPD_ACT_VEL
            const float nn2 = fmaf(act.z, act.z, fmaf(act.y, act.y, act.x * act.x));
            const float sp = P.speed_limit * fabsf(act.w);
            if (nn2 != 0.0f) { const float in = fast_rsq(nn2); tvx = sp * (act.x * in); tvy = sp * (act.y * in); tvz = sp * (act.z * in); }
            tyaw = c.yaw;                                 // keep the current yaw (:220)
        } else {   // GPD_ACT_ONE_D_PID
            tz = fmaf(0.1f, act.x, k.pz);
        }
        const Mat3 R = quat_to_mat(k.qx, k.qy, k.qz, k.qw);
        dslpid(P, C.ctrl_dt, C.inv_ctrl_dt, k, c.roll, c.pitch, c.yaw, R, tx, ty, tz, tyaw, tvx, tvy, tvz, 0.0f, 0.0f, 0.0f,
               c.s, rpm, nullptr, nullptr);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = thrust_dev(P, rpm[i]);
    }
}

// reward / terminated / truncated of a single-drone aviary (envs/HoverAviary.py:68-132); `counter` is the aviary's
// step counter BEFORE this step's increment (App. B.7)
__device__ __forceinline__ void task_single(const GpdStepCfg& C, float px, float py, float pz, float roll, float pitch,
                                            int counter, float tgx, float tgy, float tgz, float& rew, bool& term,
                                            bool& trunc) {
    const fp2 exy = fp2{tgx, tgy} - fp2{px, py};
    const float ex = exy.x, ey = exy.y, ez = tgz - pz;
    const float my_dist = fast_sqrt(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
    const float d2 = my_dist * my_dist;
    rew = fmaxf(0.0f, fmaf(-d2, d2, 2.0f));
    // (bitwise | on purpose: five compares and four scalar ORs instead of a short-circuit's exec-mask branches)
    const bool my_out = (fabsf(px) > C.xy_bound) | (fabsf(py) > C.xy_bound) | (pz > C.z_bound) |
                        (fabsf(roll) > C.tilt_bound) | (fabsf(pitch) > C.tilt_bound);
    term = my_dist < C.term_dist;
    trunc = my_out | (counter > C.trunc_counter);
}

// the value the neighbouring lane (lane ^ 1) holds: quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float pair_mate(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));
}

// ACT >= 0: the action type is a compile-time constant; S1: so is substeps == 1
template <bool PID, bool EXT, bool MULTI, int AW, int ACT = -1, bool S1 = false>
__device__ __forceinline__ void env_step(const GpdParams& P, const GpdStepCfg& C, const uint32_t flags, const int D,
                                         const Lane& L, const float4 act, const float tgx, const float tgy,
                                         const float tgz, const bool ip_regs, const float* __restrict__ ipose,
                                         const float ip0, const float ip1, const float ip2, const float ip3,
                                         const float ip4, const float ip5, const float ip6, float* sh_pos,
                                         float* sh_red, Carry& c, StepOut& out, const float* irpy = nullptr) {
    Kin& k = c.k;
    float rpm[4], g[4];                                      // RPMs; rotor thrusts minus the hover thrust
    map_action<PID, AW, ACT>(P, C, act, c, rpm, g);

    // ---- S physics sub-steps, state in registers -------------------------------------------------
    const float cur_sum = ((rpm[0] + rpm[1]) + rpm[2]) + rpm[3];
    // the first sub-step of a step sees the PREVIOUS env step's action (BaseAviary.py:359,372)
    float drag_sum = (EXT && (flags & GPD_PHYS_DRAG)) ? ((c.l0 + c.l1) + c.l2) + c.l3 : cur_sum;
    float avx = 0.0f, avy = 0.0f, avz = 0.0f;
    if (S1 && !(EXT && MULTI)) {
        // ctrl_freq == pyb_freq, known at compile time: straight-line, no loop branches (two taken branches per step)
        substep<EXT>(P, C.pyb_dt, flags, g, drag_sum, EXT ? c.dw_in : 0.0f, k, avx, avy, avz);
    } else if (!MULTI) {
        // two sub-steps per iteration: half the loop branches (HoverAviary's default 30 Hz control is 8 sub-steps).  Only
        // the LAST sub-step's world angular velocity is observed (BaseAviary.py:873 overwrites it every sub-step), so the
        // loop covers all but the last one or two sub-steps without computing it.
        const float dw = EXT ? c.dw_in : 0.0f;
        int left = C.substeps;
        for (; left > 2; left -= 2) {
            substep<EXT, false>(P, C.pyb_dt, flags, g, drag_sum, dw, k, avx, avy, avz);
            substep<EXT, false>(P, C.pyb_dt, flags, g, cur_sum, dw, k, avx, avy, avz);
            drag_sum = cur_sum;
        }
        if (left == 2) {
            substep<EXT, false>(P, C.pyb_dt, flags, g, drag_sum, dw, k, avx, avy, avz);
            drag_sum = cur_sum;
        }
        substep<EXT, true>(P, C.pyb_dt, flags, g, drag_sum, dw, k, avx, avy, avz);
    } else
    for (int ss = 0; ss < C.substeps; ++ss) {
        float dw = (EXT && !MULTI) ? c.dw_in : 0.0f;
        if (EXT && MULTI && (flags & GPD_PHYS_DW)) {
            // every drone sees the same pre-sub-step snapshot of its aviary (BaseAviary.py:346-347,798)
            auto wake_of = [&](float xj, float yj, float zj) {   // drone j above this one: its wake's push (:798-804)
                const float dz = zj - k.pz;
                const float ddx = xj - k.px, ddy = yj - k.py;
                const float dxy2 = fmaf(ddy, ddy, ddx * ddx);
                // evaluated for every mate and selected (no exec-mask branch per mate; a lane the test excludes may
                // compute inf/NaN here, which the select discards)
                const float ratio = (0.25f * P.prop_radius) * fast_rcp(dz);
                const float alpha = P.dw_coeff[0] * (ratio * ratio);
                const float beta = fmaf(P.dw_coeff[1], dz, P.dw_coeff[2]);
                const float ib = fast_rcp(beta);
                const float pushed = dw - alpha * fast_exp(-0.5f * (dxy2 * (ib * ib)));
                dw = ((dz > 0.0f) & (dxy2 < 100.0f)) ? pushed : dw;   // dz > 0 and dxy < 10 m
            };
            if (L.shfl) {
                // the aviary is D aligned lanes of THIS wave: one 16-byte LDS write per lane, one broadcast read per
                // mate, and neither a barrier nor a wait in between (the LDS executes a wave's instructions in order)
                if (D == 2) {
                    // a pair: the mate is the neighbouring lane -- three DPP moves (quad_perm [1,0,3,2]) instead of a trip through
                    // LDS, and the drone's own position (dz = 0: no contribution, whatever its turn in the reference's loop) is
                    // not evaluated at all
                    wake_of(pair_mate(k.px), pair_mate(k.py), pair_mate(k.pz));
                } else {
                float4* sp = reinterpret_cast<float4*>(sh_pos);
                __builtin_amdgcn_wave_barrier();
                sp[L.tid] = make_float4(k.px, k.py, k.pz, 0.0f);
                __builtin_amdgcn_wave_barrier();
                const int base = L.le * D;
                // D is a power of two here: groups of four, so that the LDS reads of a group are issued together and the loop
                // branch (a taken branch costs ~60 cycles at one wave per SIMD) is paid once per group
                auto mate = [&](int j) { const float4 o = sp[base + j]; wake_of(o.x, o.y, o.z); };
                for (int j = 0; j < D; j += 4) { mate(j); mate(j + 1); mate(j + 2); mate(j + 3); }
                }
            } else {
                wg_barrier();
                sh_pos[L.tid] = k.px; sh_pos[kBlock + L.tid] = k.py; sh_pos[2 * kBlock + L.tid] = k.pz;
                wg_barrier();
                const int base = L.le * D;
                for (int j = 0; j < D; ++j) wake_of(sh_pos[base + j], sh_pos[kBlock + base + j], sh_pos[2 * kBlock + base + j]);
            }
        }
        substep<EXT>(P, C.pyb_dt, flags, g, drag_sum, dw, k, avx, avy, avz);
        drag_sum = cur_sum;
    }

    // ---- cache refresh: rpy of the new quaternion (BaseAviary.py:518) ------------------------------
    quat_to_rpy(k.qx, k.qy, k.qz, k.qw, c.roll, c.pitch, c.yaw);

    // ---- task: reward / terminated / truncated -------------------------------------------------------
    float rew = -1.0f;
    bool term = false, trunc = false;
    if (!MULTI) {
        // evaluated unconditionally and masked by the (uniform) task switch: selects, no branch in the step body
        // (GPD_TASK_NONE hands the kernel a readable dummy target)
        float r;
        bool te, tr;
        task_single(C, k.px, k.py, k.pz, c.roll, c.pitch, c.counter, tgx, tgy, tgz, r, te, tr);
        const bool has_task = C.task != GPD_TASK_NONE;
        rew = has_task ? r : -1.0f;
        term = has_task & te;
        trunc = has_task & tr;
    } else if (C.task != GPD_TASK_NONE) {
        {
            const float ex = tgx - k.px, ey = tgy - k.py, ez = tgz - k.pz;
            const float my_dist = fast_sqrt(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
            const float d2 = my_dist * my_dist;
            const float my_rew = fmaxf(0.0f, fmaf(-d2, d2, 2.0f));
            const bool my_out = fabsf(k.px) > C.xy_bound || fabsf(k.py) > C.xy_bound || k.pz > C.z_bound ||
                                fabsf(c.roll) > C.tilt_bound || fabsf(c.pitch) > C.tilt_bound;
            float r = 0.0f, dsum = 0.0f, o = 0.0f;
            const float my_o = my_out ? 1.0f : 0.0f;
            if (L.shfl && D == 2) {                            // a pair: the mate's three values by DPP (a + b either way round)
                r = my_rew + pair_mate(my_rew); dsum = my_dist + pair_mate(my_dist); o = my_o + pair_mate(my_o);
            } else if (L.shfl) {                               // wave-local LDS exchange, as for the downwash
                float4* sr = reinterpret_cast<float4*>(sh_red);
                __builtin_amdgcn_wave_barrier();
                sr[L.tid] = make_float4(my_rew, my_dist, my_o, 0.0f);
                __builtin_amdgcn_wave_barrier();
                const int base = L.le * D;
                auto mate = [&](int j) { const float4 v = sr[base + j]; r += v.x; dsum += v.y; o += v.z; };   // sequential, like the
                for (int j = 0; j < D; j += 4) { mate(j); mate(j + 1); mate(j + 2); mate(j + 3); }             // reference's loops
            } else {
                wg_barrier();
                sh_red[L.tid] = my_rew; sh_red[kBlock + L.tid] = my_dist; sh_red[2 * kBlock + L.tid] = my_o;
                wg_barrier();
                const int base = L.le * D;
                for (int j = 0; j < D; ++j) {                  // sequential, like the reference's loops
                    r += sh_red[base + j]; dsum += sh_red[kBlock + base + j]; o += sh_red[2 * kBlock + base + j];
                }
            }
            rew = r;
            term = dsum < C.term_dist;
            trunc = (o > 0.0f) || (c.counter > C.trunc_counter);
        }
    }
    const bool do_reset = C.auto_reset && (term || trunc);
    c.counter = do_reset ? 0 : c.counter + C.substeps;        // BaseAviary.py:382 / :460
    out.rew = rew; out.term = term; out.trunc = trunc; out.reset = do_reset;

    // ---- observation; same-step auto-reset ----------------------------------------------------------------
    c.l0 = rpm[0]; c.l1 = rpm[1]; c.l2 = rpm[2]; c.l3 = rpm[3];
    if (any_lane(do_reset)) { asm volatile("; episode end"); if (do_reset) {
        out.to[0] = k.px; out.to[1] = k.py; out.to[2] = k.pz; out.to[3] = c.roll; out.to[4] = c.pitch; out.to[5] = c.yaw;
        out.to[6] = k.vx; out.to[7] = k.vy; out.to[8] = k.vz; out.to[9] = avx; out.to[10] = avy; out.to[11] = avz;
        if (ip_regs) k = Kin{ip0, ip1, ip2, ip3, ip4, ip5, ip6, 0, 0, 0, 0, 0, 0};
        else k = Kin{ipose[0], ipose[1], ipose[2], ipose[3], ipose[4], ipose[5], ipose[6], 0, 0, 0, 0, 0, 0};
        // (the Euler angles of the reset pose: the K-step kernels compute them once per launch -- in the headline workload an
        // aviary of some wave's 64 ends its episode in one step out of six, and this block is what that wave then runs)
        if (irpy) { c.roll = irpy[0]; c.pitch = irpy[1]; c.yaw = irpy[2]; }
        else quat_to_rpy(k.qx, k.qy, k.qz, k.qw, c.roll, c.pitch, c.yaw);
        avx = avy = avz = 0.0f;
        c.l0 = c.l1 = c.l2 = c.l3 = 0.0f;                      // last_clipped_action zeroed (BaseAviary.py:468)
    } }
    out.o[0] = k.px; out.o[1] = k.py; out.o[2] = k.pz; out.o[3] = c.roll; out.o[4] = c.pitch; out.o[5] = c.yaw;
    out.o[6] = k.vx; out.o[7] = k.vy; out.o[8] = k.vz; out.o[9] = avx; out.o[10] = avy; out.o[11] = avz;
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte vector at 4-byte alignment
typedef float f4v __attribute__((ext_vector_type(4)));

struct Span {             // internal: how the K steps of one launch are laid out in memory
    int32_t num_steps;
    int64_t action_stride;   // floats between the action blocks of consecutive steps (0: same action each step)
    int64_t obs_stride;      // floats between the obs12 (and term_obs12) blocks of consecutive steps
    int64_t env_stride;      // elements between the reward / terminated / truncated rows of consecutive steps
    int32_t ring;            // gpd_rollout: LDS output slots per workgroup (2 or 4), chosen at launch
};

// the raw action words of one drone (AW = 4, 3 or 1 floats per drone, row-major), one load instruction
template <int AW, bool NT = false>   // NT: read once, streamed in -- non-temporal
__device__ __forceinline__ float4 load_action(const float* __restrict__ action, uint32_t n) {
    const char* p = reinterpret_cast<const char*>(action) + n * static_cast<uint32_t>(AW * 4);
    if (AW == 1) {
        const float* ap = reinterpret_cast<const float*>(p);
        return make_float4(NT ? __builtin_nontemporal_load(ap) : *ap, 0.0f, 0.0f, 0.0f);
    }
    if (AW == 3) {
        const float* ap = reinterpret_cast<const float*>(p);
        if (NT) return make_float4(__builtin_nontemporal_load(ap), __builtin_nontemporal_load(ap + 1), __builtin_nontemporal_load(ap + 2), 0.0f);
        return make_float4(ap[0], ap[1], ap[2], 0.0f);
    }
    if (NT) {
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4*>(p);
}

// Loads everything a launch needs once per drone: kinematics, DSLPID members, last RPMs, step counter, task
// target, (rollout: reset pose).  Branch-free on purpose: every load is issued back to back and ONE wait
// covers them all (a load inside a branch makes the compiler wait for it -- a full memory round trip --
// before the next one is even issued).  Lanes without a drone read drone 0's rows.
template <bool PID, bool EXT, bool IP = true>     // IP: also fetch the reset pose (rollouts; a single step reads it only when it resets)
__device__ __forceinline__ void load_carry(const GpdState& S, const GpdStepCfg& C, const uint32_t flags, const Lane& L,
                                           const float* __restrict__ target_pos, const float* __restrict__ ipl,
                                           Carry& c, float& tgx, float& tgy, float& tgz, float ip[7]) {
    const int64_t ld = S.ld;
    const uint32_t off4 = L.n * 4u;
    Kin& k = c.k;
#ifdef GPD_EXP_KIN4
    {   // experiment: rows 0-3 | 4-7 | 8-11 interleaved as three float4 planes, row 12 as it is (the same 13 x ld floats)
        const uint32_t off16 = L.n * 16u;
        const f4v a = *reinterpret_cast<const f4v*>(reinterpret_cast<const char*>(S.kin) + off16);
        const f4v b = *reinterpret_cast<const f4v*>(reinterpret_cast<const char*>(S.kin + 4 * ld) + off16);
        const f4v d = *reinterpret_cast<const f4v*>(reinterpret_cast<const char*>(S.kin + 8 * ld) + off16);
        k.px = a.x; k.py = a.y; k.pz = a.z; k.qx = a.w; k.qy = b.x; k.qz = b.y; k.qw = b.z; k.vx = b.w;
        k.vy = d.x; k.vz = d.y; k.wx = d.z; k.wy = d.w; k.wz = ld_row(S.kin, ld, 12, off4);
    }
#else
    k.px = ld_row(S.kin, ld, 0, off4); k.py = ld_row(S.kin, ld, 1, off4); k.pz = ld_row(S.kin, ld, 2, off4);
    k.qx = ld_row(S.kin, ld, 3, off4); k.qy = ld_row(S.kin, ld, 4, off4); k.qz = ld_row(S.kin, ld, 5, off4);
    k.qw = ld_row(S.kin, ld, 6, off4);
    k.vx = ld_row(S.kin, ld, 7, off4); k.vy = ld_row(S.kin, ld, 8, off4); k.vz = ld_row(S.kin, ld, 9, off4);
    k.wx = ld_row(S.kin, ld, 10, off4); k.wy = ld_row(S.kin, ld, 11, off4); k.wz = ld_row(S.kin, ld, 12, off4);
#endif
    c.counter = S.step_counter[L.env];                       // every drone of an aviary reads its aviary's counter
    GPD_DBG(c.counter >= 0, GPD_DBG_STEP_COUNTER, c.counter);
    // target (task NONE: the host passes a readable dummy, the values are not used)
    const float* tp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(target_pos) +
                                                     (C.target_per_env ? L.n * 12u : static_cast<uint32_t>(L.d) * 12u));
    tgx = tp[0]; tgy = tp[1]; tgz = tp[2];
    c.s = Pid{0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (PID) {
        Pid& s = c.s;
        s.ipx = ld_row(S.pid, ld, 0, off4); s.ipy = ld_row(S.pid, ld, 1, off4); s.ipz = ld_row(S.pid, ld, 2, off4);
        s.lr = ld_row(S.pid, ld, 3, off4); s.lp = ld_row(S.pid, ld, 4, off4); s.ly = ld_row(S.pid, ld, 5, off4);
        s.irx = ld_row(S.pid, ld, 6, off4); s.iry = ld_row(S.pid, ld, 7, off4); s.irz = ld_row(S.pid, ld, 8, off4);
    }
    c.l0 = c.l1 = c.l2 = c.l3 = 0.0f;
    if (EXT) {
        // without the drag term the last RPMs are not needed: the loads then go to the (cached) first kin rows
        const float* lr = (flags & GPD_PHYS_DRAG) ? S.last_rpm : S.kin;
        c.l0 = ld_row(lr, ld, 0, off4); c.l1 = ld_row(lr, ld, 1, off4); c.l2 = ld_row(lr, ld, 2, off4); c.l3 = ld_row(lr, ld, 3, off4);
    }
    c.dw_in = 0.0f;
    if (EXT) {
        const bool ext_dw = (flags & GPD_PHYS_DW) && S.dw_force != nullptr;
        const float v = ld_row(ext_dw ? S.dw_force : S.kin, ld, 0, off4);      // (unused case: a cached kin row)
        c.dw_in = ext_dw ? v : 0.0f;
    }
    if (IP) { ip[0] = ipl[0]; ip[1] = ipl[1]; ip[2] = ipl[2]; ip[3] = ipl[3]; ip[4] = ipl[4]; ip[5] = ipl[5]; ip[6] = ipl[6]; }
}

template <bool PID, bool NT = false>
__device__ __forceinline__ void store_carry(const GpdState& S, const Lane& L, const Carry& c) {
    const int64_t ld = S.ld;
    const uint32_t off4 = L.n * 4u;
    const Kin& k = c.k;
    if (L.d == 0) S.step_counter[L.env] = c.counter;
    if (S.bad) {
        // non-finite guard (include/gpd.h): a NaN or an infinity in any of the 13 floats makes the sum non-finite (inf - inf = NaN;
        // finite values of this magnitude cannot overflow it)
        const float sum = (((k.px + k.py) + (k.pz + k.qx)) + ((k.qy + k.qz) + (k.qw + k.vx))) + (((k.vy + k.vz) + (k.wx + k.wy)) + k.wz);
        // (the DSLPID members need no check: every one of them passes a clip or is overwritten each step -- control/
        // DSLPIDControl.py:191-193, 248-251 -- and v_max / v_min return the bound for a NaN where numpy's clip would propagate it.
        // For the same reason a NaN SET-POINT never reaches the state: one step of saturated commands, then the controller
        // recovers, where the reference's drone is gone for good.  tests/test_gpu_resume.py pins both behaviours.)
        S.bad[L.n] = fabsf(sum) <= 3.4028234e38f ? 0 : 1;
    }
#ifdef GPD_EXP_KIN4
    {
        const uint32_t off16 = L.n * 16u;
        // (opaque copies: without them the twelve field reads are combined into vector loads of the Carry struct, which then stays
        // an alloca -- promoted to LDS, indexed by a flat thread id computed from the DISPATCH PACKET, a read of host memory per wave:
        // 4.0 -> 14.6 us per step at 65 536 drones, gpurun_out/ab_step_r05.log)
        float e0 = k.px, e1 = k.py, e2 = k.pz, e3 = k.qx, e4 = k.qy, e5 = k.qz, e6 = k.qw, e7 = k.vx, e8 = k.vy, e9 = k.vz, e10 = k.wx, e11 = k.wy;
        asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7), "+v"(e8), "+v"(e9), "+v"(e10), "+v"(e11));
        const f4v a = {e0, e1, e2, e3}, b = {e4, e5, e6, e7}, d = {e8, e9, e10, e11};
        f4v* pa = reinterpret_cast<f4v*>(reinterpret_cast<char*>(S.kin) + off16);
        f4v* pb = reinterpret_cast<f4v*>(reinterpret_cast<char*>(S.kin + 4 * ld) + off16);
        f4v* pd = reinterpret_cast<f4v*>(reinterpret_cast<char*>(S.kin + 8 * ld) + off16);
        if (NT) { __builtin_nontemporal_store(a, pa); __builtin_nontemporal_store(b, pb); __builtin_nontemporal_store(d, pd); }
        else { *pa = a; *pb = b; *pd = d; }
        st_row<NT>(S.kin, ld, 12, off4, k.wz);
    }
#else
    st_row<NT>(S.kin, ld, 0, off4, k.px); st_row<NT>(S.kin, ld, 1, off4, k.py); st_row<NT>(S.kin, ld, 2, off4, k.pz);
    st_row<NT>(S.kin, ld, 3, off4, k.qx); st_row<NT>(S.kin, ld, 4, off4, k.qy); st_row<NT>(S.kin, ld, 5, off4, k.qz);
    st_row<NT>(S.kin, ld, 6, off4, k.qw);
    st_row<NT>(S.kin, ld, 7, off4, k.vx); st_row<NT>(S.kin, ld, 8, off4, k.vy); st_row<NT>(S.kin, ld, 9, off4, k.vz);
    st_row<NT>(S.kin, ld, 10, off4, k.wx); st_row<NT>(S.kin, ld, 11, off4, k.wy); st_row<NT>(S.kin, ld, 12, off4, k.wz);
#endif
    if (S.last_rpm) {
        st_row<NT>(S.last_rpm, ld, 0, off4, c.l0); st_row<NT>(S.last_rpm, ld, 1, off4, c.l1);
        st_row<NT>(S.last_rpm, ld, 2, off4, c.l2); st_row<NT>(S.last_rpm, ld, 3, off4, c.l3);
    }
    if (PID) {
        const Pid& s = c.s;
        st_row<NT>(S.pid, ld, 0, off4, s.ipx); st_row<NT>(S.pid, ld, 1, off4, s.ipy); st_row<NT>(S.pid, ld, 2, off4, s.ipz);
        st_row<NT>(S.pid, ld, 3, off4, s.lr); st_row<NT>(S.pid, ld, 4, off4, s.lp); st_row<NT>(S.pid, ld, 5, off4, s.ly);
        st_row<NT>(S.pid, ld, 6, off4, s.irx); st_row<NT>(S.pid, ld, 7, off4, s.iry); st_row<NT>(S.pid, ld, 8, off4, s.irz);
    }
}

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
template <bool PID, bool EXT, bool MULTI, int AW, int ACT, bool S1>
__global__ __launch_bounds__(kBlock) void gpd_step_kernel(
    float* __restrict__ hot_kin, const float* __restrict__ action, int32_t* __restrict__ hot_counter,
    const float* __restrict__ target_pos, const int32_t* __restrict__ hot_slot, const uint32_t hot_ld, const int32_t hot_num_envs,
    const int32_t hot_lanes_per_wave, const int32_t hot_target_per_env,
    const GpdParams P, const GpdState S_, const GpdStepCfg C_, const float* __restrict__ init_pose, float* __restrict__ obs12,
    float* __restrict__ reward, uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
    float* __restrict__ term_obs12) {
    GpdState S = S_;
    S.kin = hot_kin; S.step_counter = hot_counter; S.ld = hot_ld;
    GpdStepCfg C = C_;
    C.num_envs = hot_num_envs; C.lanes_per_wave = hot_lanes_per_wave; C.target_per_env = hot_target_per_env;
    const int D = MULTI ? C.drones_per_env : 1;
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

    __shared__ __attribute__((aligned(16))) float sh_pos[MULTI ? 4 * kBlock : 4];   // downwash: positions of the env's drones
    __shared__ __attribute__((aligned(16))) float sh_red[MULTI ? 4 * kBlock : 4];   // reward | distance | out-of-bounds per drone
    __shared__ __attribute__((aligned(16))) float sh_rows[kBlock * 12];   // obs rows, for the coalesced store of large batches

    const uint32_t flags = EXT ? C.physics_flags : 0u;
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
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(src + off + j * 1024);
                if (off + j * 1024 < rows * 48u) {                    // streamed out, not read again by this path: non-temporal
                    f4v w = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(dst + off + j * 1024));
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
    // the state block: kept in the 256 MB Infinity Cache between steps while it fits (4.2M drones = 218 MB), streamed
    // through non-temporally beyond (measured: 4M 125 vs 130 us cached, 16.7M 499 vs 513..558 us streamed)
#ifdef GPD_EXP_NTSTATE
    store_carry<PID, true>(S, L, c);
#else
    if (N > (1u << 22)) store_carry<PID, true>(S, L, c);
    else store_carry<PID, false>(S, L, c);
#endif
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



template <bool PID, bool EXT, bool MULTI, int AW>
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
        env_step<PID, EXT, MULTI, AW>(P, C, flags, D, L, act, tgx, tgy, tgz, true, ipose, ip[0], ip[1], ip[2], ip[3], ip[4],
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
// MULTI: aviaries of D = 2, 4, ..., 64 drones (a power of two: D aligned lanes of one wave, wave-local exchange inside
// env_step, no workgroup barrier).  Every lane of an aviary ends a step with the aviary's reward and flags and stores
// them to the aviary's slot -- D identical writes instead of a branch; a lane without a drone is a clone of the drone with
// the same index d in its workgroup's first aviary, so whole clone aviaries replay that aviary bit for bit.
#ifdef GPD_EXP_TS
__device__ unsigned long long gpd_ts[8 * 4096 * 4];
__device__ unsigned int gpd_ts_cnt[4096];
#endif
// RING (gpd_rollout_history): every step's raw action is also pushed into the action ring, like gpd_step does (slots q and
// q + H of the double ring; two more stores per lane and step, which the explicit wait counts of the loop include because
// they are unconditional -- a compile-time variant, not a run-time test).
template <bool PID, bool EXT, int AW, int ACT, bool S1, bool MULTI, bool NT_OBS = true, bool RING = false>
__global__ __launch_bounds__(kBlock) void gpd_rollout1_kernel(
    const GpdParams P, const GpdState S, const GpdStepCfg C, const Span T, const float* __restrict__ action,
    const float* __restrict__ target_pos, const float* __restrict__ init_pose, float* __restrict__ obs12,
    float* __restrict__ reward, uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
    float* __restrict__ term_obs12) {
    const int tid = threadIdx.x;
#ifdef GPD_EXP_TS
    const unsigned long long ts0 = wall_clock64();
#endif
    // (workgroup -> drones is the identity: giving every XCD one contiguous block of drones instead of every eighth workgroup
    // changed nothing, 0.816 vs 0.813-0.821 us per step, round-2 A/B)
    const uint32_t bid = blockIdx.x;
    const int D = MULTI ? C.drones_per_env : 1;
    const uint32_t dmask = static_cast<uint32_t>(D - 1);
    const uint32_t N = static_cast<uint32_t>(C.num_envs) * static_cast<uint32_t>(D);
    const uint32_t n_raw = bid * static_cast<uint32_t>(kBlock) + tid;
    const int K = T.num_steps;
    const uint32_t flags = EXT ? C.physics_flags : 0u;
    Lane L;
    L.tid = tid; L.shfl = MULTI;
    L.le = MULTI ? tid / D : tid;
    L.d = MULTI ? static_cast<int>(tid & dmask) : 0;
    L.active = n_raw < N;
    // a lane without a drone (ragged last workgroup) clones a drone of ITS OWN workgroup -- the one with the same index d
    // in the workgroup's first aviary (which exists: the grid covers N, and N and 256 are multiples of D).  Owner and
    // clone are co-resident, and the barrier behind load_carry orders the clone's loads before the owner's store_carry.
    const uint32_t block_base = bid * static_cast<uint32_t>(kBlock);
    L.n = L.active ? n_raw : block_base + (tid & dmask);
    L.env = MULTI ? L.n / static_cast<uint32_t>(D) : L.n;

    __shared__ __attribute__((aligned(16))) float sh_rows[kBlock * 12];
    __shared__ __attribute__((aligned(16))) float sh_pos[MULTI ? 4 * kBlock : 4];   // downwash: positions of the aviary's drones
    __shared__ __attribute__((aligned(16))) float sh_red[MULTI ? 4 * kBlock : 4];   // reward | distance | out-of-bounds per drone

    // loop-invariant addressing of this lane's three 16-byte chunks of its wave's 3 KiB row patch
    const int wave0 = tid & ~63, lane = tid & 63;
    const uint32_t n0 = bid * static_cast<uint32_t>(kBlock) + wave0;      // first drone of this wave
    const uint32_t rows = n0 < N ? ((N - n0 < 64u) ? N - n0 : 64u) : 0u;         // lanes of this wave that own a drone
    uint32_t goff[3];
    const char* lsrc = reinterpret_cast<const char*>(sh_rows + wave0 * 12) + lane * 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t cidx = static_cast<uint32_t>(j * 64 + lane), r = cidx / 3u, part = cidx - 3u * r;
        goff[j] = (r < rows ? (n0 + r) : block_base + (r & dmask)) * 48u + part * 16u;   // a clone's row goes to the row of its original
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
#ifdef GPD_EXP_TS
    const unsigned long long ts1 = wall_clock64();
#endif
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
#ifdef GPD_EXP_TS
    const unsigned long long ts2 = wall_clock64();
#endif
    ro.flush();                                                      // the last step's bursts
    if (L.active) store_carry<PID>(S, L, c);
    if constexpr (RING) { if (L.active && L.d == 0) S.ring_pos[L.env] = ring_q; }
#ifdef GPD_EXP_TS
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long ts3 = wall_clock64();
    if (tid == 0 && bid < 4096) {
        const unsigned int slot = gpd_ts_cnt[bid]++ & 7u;
        unsigned long long* o = gpd_ts + (static_cast<size_t>(slot) * 4096 + bid) * 4;
        o[0] = ts0; o[2] = ts2; o[3] = ts3;
#ifdef GPD_EXP_HWID
        (void)ts1;
        o[1] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11))) << 32) |
               static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)));
#else
        o[1] = ts1;
#endif
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// gpd_rollout_policy: K env steps per launch with an MLP policy IN the loop (the actor of SB3's default MlpPolicy:
// in_dim -> 64 -> 64 -> A, tanh or ReLU; examples/learn.py:157-192).  One drone per lane, state in registers like the rollout
// kernels; per step the 64 drones of a wavefront are the COLUMNS of the matrix-core tiles:
//     H1^T [64 x 64 drones] = W1 [64 x in] . X^T [in x 64 drones]     (A operand = weights, from LDS; B = features, registers)
// with v_mfma_f32_32x32x16_bf16: 2 row tiles (hidden units) x 2 column tiles (drones) x in/16 K-steps.
//   * Precision: every operand is split into bf16 hi + lo (x = hi + lo to ~16 bits) and a tile takes three MFMAs
//     (hi.hi + hi.lo + lo.hi, fp32 accumulate): plain bf16 would resolve a position of 1.0 m to 4 mm.
//   * B operand of layer 1: lane l holds drone l's features; the tile of drones 0..31 wants K-slots 0..7 from lanes 0..31 and
//     K-slots 8..15 OF THE SAME DRONES from lanes 32..63 -- one v_permlane32_swap per register pair hands both column tiles
//     their operand (x' = [x.lo | y.lo], y' = [x.hi | y.hi]).
//   * Chaining: the C/D layout puts hidden unit (r&3) + 8(r>>2) + 4(lane>>5) of drone lane&31 in register r -- exactly "eight
//     values per lane and K-step" if the NEXT layer's K index is permuted accordingly.  The permutation is applied to the
//     weight matrices when they are staged in LDS, so activations never leave registers between layers.
//   * The action history (the tail of the reference's observation row) lives in registers as packed bf16 hi/lo pairs and
//     shifts by one action per step; its capacity is 16*NK1 - 12 features, a shorter history sits at the end of it and
//     the unused features in front meet zero weight columns (whatever shifts into them is multiplied by zero).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// two floats -> packed bf16 hi parts (round to nearest even) and packed bf16 lo parts (bf16 of the remainders)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    const f32x2 r = f32x2{a, b} - f32x2{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};      // (exact: v_pk_add_f32)
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ bf16x8 as_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(bf16x8, u32x4{a, b, c, d});
}
// tanh(x) = 2 / (1 + exp(-2x)) - 1: v_exp_f32 + v_rcp_f32 (1 ulp each) and three plain operations, no sign handling (x -> -inf:
// exp -> inf, rcp -> 0, result -1; x -> +inf: exp -> 0, result 1); absolute error ~1.5e-7
[[maybe_unused]] __device__ __forceinline__ float fast_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * -2.88539008177792681472f);            // exp(-2x) = 2^(-2 log2(e) x)
    return fmaf(2.0f, fast_rcp(1.0f + t), -1.0f);
}

// K index (unit of the previous layer) that K-step s, lane half h, element j of an operand stands for, layers 2 and 3
__device__ __forceinline__ int chained_k(int s, int h, int j) {
    const int r = 8 * (s & 1) + j;
    return 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
}

constexpr int kPolHidden = 64;
template <int NK1> struct PolLds {     // weights as A operands: [block][hi|lo][64 lanes] x 16 bytes; biases per (tile, lane half)
    uint4 w1[2 * NK1][2][64];
    uint4 w2[2 * 4][2][64];
    uint4 w3[1 * 4][2][64];
    float b1[2][2][16], b2[2][2][16], b3[2][16];
};

// PID: a DSLPID action type (the policy's output is the controller's set-point); the add-on physics terms are always compiled in
// (wave-uniform run-time tests: a handful of slots next to the policy's ~2 500)
// NOISE (training rollouts, SB3's DiagGaussianDistribution with a state-independent log_std): the action is
// mean + std[k] * eps with eps ~ N(0, 1) handed in per step and drone ([K][N][A], e.g. torch.randn), clipped to the action space
// for the environment (what SB3's collect_rollouts does before env.step) -- the log-probability follows from eps alone on the
// host; the unclipped MEAN is written out on request.  The body is shared; the deterministic kernels keep their signature.
template <bool PID, int AW, int ACT, int NK1, bool RELU, bool NOISE>
__device__ __forceinline__ void rollout_policy_body(
    const GpdParams& P, const GpdState& S, const GpdStepCfg& C, const Span& T, const GpdPolicy& Pol,
    const float* __restrict__ obs12_in, const float* __restrict__ target_pos, const float* __restrict__ init_pose,
    float* __restrict__ actions_out, float* __restrict__ obs12, float* __restrict__ reward, uint8_t* __restrict__ terminated,
    uint8_t* __restrict__ truncated, const float* __restrict__ noise, float* __restrict__ mean_out, const float sd0, const float sd1,
    const float sd2, const float sd3, float* __restrict__ term_obs12) {
    constexpr int CAP = 16 * NK1 - 12;                       // history capacity in features
    constexpr int NP = 8 * NK1;                              // packed feature registers (two bf16 features each)
    constexpr bool HIST = NK1 > 1;
    __shared__ PolLds<NK1> W;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const uint32_t N = static_cast<uint32_t>(C.num_envs);
    const uint32_t n_raw = blockIdx.x * static_cast<uint32_t>(kBlock) + tid;
    const int K = T.num_steps;
    constexpr bool EXT = true;
    const uint32_t flags = C.physics_flags;
    Lane L;
    L.tid = tid; L.shfl = false; L.le = tid; L.d = 0;
    L.active = n_raw < N;
    L.n = L.active ? n_raw : blockIdx.x * static_cast<uint32_t>(kBlock);     // (a lane without a drone replays its workgroup's first one, stores nothing)
    L.env = L.n;
    const int HA = HIST ? S.hist_len * AW : 0;               // history features in use (<= CAP, checked by the host)
    const int pad = CAP - HA;                                // unused features in front of the history

    // ---- stage the weights in LDS as A operands (bf16 hi / lo), K permuted for the chained layers -------------------------
    auto stage = [&](uint4 (*dst)[2][64], const float* __restrict__ w, int rows, int ld, int blocks_k, int tiles_m, bool first) {
        for (int e = tid; e < tiles_m * blocks_k * 64; e += kBlock) {
            const int l = e & 63, blk = e >> 6, m = blk / blocks_k, s = blk - m * blocks_k, h = l >> 5, row = 32 * m + (l & 31);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int k;
                if (first) {                                 // layer 1: feature f = 16 s + 8 h + j -> column of w1 (or none)
                    const int f = 16 * s + 8 * h + j;
                    k = f < 12 ? f : (f - 12 < pad ? -1 : f - pad);
                    if (k >= Pol.in_dim) k = -1;
                } else {
                    k = chained_k(s, h, j);
                }
                v[j] = (row < rows && k >= 0) ? w[row * ld + k] : 0.0f;
            }
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) split_pair(v[2 * i], v[2 * i + 1], hi[i], lo[i]);
            dst[blk][0][l] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            dst[blk][1][l] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    };
    stage(W.w1, Pol.w1, kPolHidden, Pol.in_dim, NK1, 2, true);
    stage(W.w2, Pol.w2, kPolHidden, kPolHidden, 4, 2, false);
    stage(W.w3, Pol.w3, AW, kPolHidden, 4, 1, false);
    if (tid < 64) {                                          // biases in C/D order: element r of (tile m, lane half h) = unit 32 m + (r&3) + 8 (r>>2) + 4 h
        const int m = tid >> 5, h = (tid >> 4) & 1, r = tid & 15, u = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
        W.b1[m][h][r] = Pol.b1[u];
        W.b2[m][h][r] = Pol.b2[u];
        if (m == 0) W.b3[h][r] = u < AW ? Pol.b3[u] : 0.0f;
    }

    // ---- the drone: state, latest observation, action history ------------------------------------------------------------
    Carry c;
    float tgx, tgy, tgz, ip[7];
    const float* ipose = reinterpret_cast<const float*>(reinterpret_cast<const char*>(init_pose) + (C.init_per_env ? L.n * 28u : 0u));
    load_carry<PID, EXT>(S, C, flags, L, target_pos, C.auto_reset ? ipose : S.kin, c, tgx, tgy, tgz, ip);
    c.roll = c.pitch = c.yaw = 0.0f;
    if (PID) quat_to_rpy(c.k.qx, c.k.qy, c.k.qz, c.k.qw, c.roll, c.pitch, c.yaw);
    float o[12];
    {
        const float4* op = reinterpret_cast<const float4*>(obs12_in + static_cast<size_t>(L.n) * 12);
        const float4 o0 = op[0], o1 = op[1], o2 = op[2];
        o[0] = o0.x; o[1] = o0.y; o[2] = o0.z; o[3] = o0.w; o[4] = o1.x; o[5] = o1.y; o[6] = o1.z; o[7] = o1.w;
        o[8] = o2.x; o[9] = o2.y; o[10] = o2.z; o[11] = o2.w;
    }
    uint32_t Fhi[NP], Flo[NP];                               // packed features: [0..5] observation, [6..] history
#pragma unroll
    for (int i = 0; i < NP; ++i) Fhi[i] = Flo[i] = 0u;
    const bool ring = S.act_ring != nullptr;                 // the action ring is kept current (one push per step, like gpd_step)
    const size_t slot = static_cast<size_t>(N) * AW;
    int ring_q = (ring ? S.ring_pos : S.step_counter)[L.env];
    if (HIST) {
        const int p = ring_q;
#pragma unroll
        for (int i = 0; i < CAP / 2; ++i) {                  // features 12 + 2i, 12 + 2i + 1 <- history elements (oldest first)
            float v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * i + q - pad;               // element of the HA-long history, < 0: unused
                v[q] = e >= 0 ? S.act_ring[static_cast<size_t>(p + e / AW) * slot + static_cast<size_t>(L.n) * AW + e % AW] : 0.0f;
            }
            split_pair(v[0], v[1], Fhi[6 + i], Flo[6 + i]);
        }
    }
    __syncthreads();                                         // weights staged

    const uint4* w1 = &W.w1[0][0][lane];                     // this lane's 16 bytes of every A block: + (blk * 2 + part) * 64
    const uint4* w2 = &W.w2[0][0][lane];
    const uint4* w3 = &W.w3[0][0][lane];
    auto a_frag = [](const uint4* base, int blk, int part) {
        const uint4 q = base[(blk * 2 + part) * 64];
        return as_frag(q.x, q.y, q.z, q.w);
    };
    auto bias16 = [&](const float* b) {
        f32x16 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 q = reinterpret_cast<const float4*>(b)[i];
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        }
        return v;
    };
    auto mfma3 = [](bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);      // (small terms first)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    };
    // activation of one output tile -> the next layer's B operands (two K-steps: registers 0..7 and 8..15), hi and lo
    auto activate = [&](const f32x16& acc, bf16x8 bh[2], bf16x8 bl[2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x0 = acc[8 * s + 2 * i], x1 = acc[8 * s + 2 * i + 1];
                float y0, y1;
                if (RELU) { y0 = fmaxf(x0, 0.0f); y1 = fmaxf(x1, 0.0f); }
                else {                                       // fast_tanh on a pair: the three plain operations as packed instructions
                    const fp2 e = fp2{x0, x1} * splat(-2.88539008177792681472f);
                    const fp2 u = fp2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} + splat(1.0f);
                    const fp2 y = fma2(splat(2.0f), fp2{fast_rcp(u.x), fast_rcp(u.y)}, splat(-1.0f));
                    y0 = y.x; y1 = y.y;
                }
                split_pair(y0, y1, hi[i], lo[i]);
            }
            bh[s] = as_frag(hi[0], hi[1], hi[2], hi[3]);
            bl[s] = as_frag(lo[0], lo[1], lo[2], lo[3]);
        }
    };

    const size_t obs_step = static_cast<size_t>(T.obs_stride), env_stride = static_cast<size_t>(T.env_stride);
    for (int t = 0; t < K; ++t) {
        float eps[4] = {0.0f, 0.0f, 0.0f, 0.0f};             // (requested here, used after the three layers)
        if constexpr (NOISE) {
            const float* er = noise + (static_cast<size_t>(t) * N + L.n) * AW;
            if (AW == 4) { const float4 q = *reinterpret_cast<const float4*>(er); eps[0] = q.x; eps[1] = q.y; eps[2] = q.z; eps[3] = q.w; }
            else { eps[0] = er[0]; if (AW == 3) { eps[1] = er[1]; eps[2] = er[2]; } }
        }
        // ---- features of this step: the observation row (the history registers are current) ------------------------------
#pragma unroll
        for (int i = 0; i < 6; ++i) split_pair(o[2 * i], o[2 * i + 1], Fhi[i], Flo[i]);
        // ---- layer 1: B operands of both column tiles by one lane-half swap per register pair ------------------------------
        bf16x8 b1h[2][NK1], b1l[2][NK1];
#pragma unroll
        for (int s = 0; s < NK1; ++s) {
            uint32_t h0[4], h1[4], l0[4], l1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto sh = __builtin_amdgcn_permlane32_swap(Fhi[8 * s + i], Fhi[8 * s + 4 + i], false, false);
                const auto sl = __builtin_amdgcn_permlane32_swap(Flo[8 * s + i], Flo[8 * s + 4 + i], false, false);
                h0[i] = sh[0]; h1[i] = sh[1]; l0[i] = sl[0]; l1[i] = sl[1];
            }
            b1h[0][s] = as_frag(h0[0], h0[1], h0[2], h0[3]); b1h[1][s] = as_frag(h1[0], h1[1], h1[2], h1[3]);
            b1l[0][s] = as_frag(l0[0], l0[1], l0[2], l0[3]); b1l[1][s] = as_frag(l1[0], l1[1], l1[2], l1[3]);
        }
        bf16x8 b2h[2][4], b2l[2][4];                         // [column tile][K-step] for layer 2
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x16 acc0 = bias16(W.b1[m][half]), acc1 = acc0;
#pragma unroll
            for (int s = 0; s < NK1; ++s) {
                const bf16x8 ah = a_frag(w1, m * NK1 + s, 0), al = a_frag(w1, m * NK1 + s, 1);
                acc0 = mfma3(ah, al, b1h[0][s], b1l[0][s], acc0);
                acc1 = mfma3(ah, al, b1h[1][s], b1l[1][s], acc1);
            }
            activate(acc0, &b2h[0][2 * m], &b2l[0][2 * m]);
            activate(acc1, &b2h[1][2 * m], &b2l[1][2 * m]);
        }
        // ---- layer 2 ----------------------------------------------------------------------------------------------------------
        bf16x8 b3h[2][4], b3l[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x16 acc0 = bias16(W.b2[m][half]), acc1 = acc0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 ah = a_frag(w2, m * 4 + s, 0), al = a_frag(w2, m * 4 + s, 1);
                acc0 = mfma3(ah, al, b2h[0][s], b2l[0][s], acc0);
                acc1 = mfma3(ah, al, b2h[1][s], b2l[1][s], acc1);
            }
            activate(acc0, &b3h[0][2 * m], &b3l[0][2 * m]);
            activate(acc1, &b3h[1][2 * m], &b3l[1][2 * m]);
        }
        // ---- layer 3: rows 0 .. AW-1 of one row tile are the action; clip to the action space (SB3 predict()) ---------------
        f32x16 y0 = bias16(W.b3[half]), y1 = y0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ah = a_frag(w3, s, 0), al = a_frag(w3, s, 1);
            y0 = mfma3(ah, al, b3h[0][s], b3l[0][s], y0);
            y1 = mfma3(ah, al, b3h[1][s], b3l[1][s], y1);
        }
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < AW; ++k) {                       // unit k sits in register k of lanes 0..31: drones 0..31 keep y0, drones 32..63 fetch y1
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(y0[k]), __float_as_uint(y1[k]), false, false);
            if constexpr (NOISE) {
                const float mu = __uint_as_float(sw[0]), sd = k == 0 ? sd0 : (k == 1 ? sd1 : (k == 2 ? sd2 : sd3));
                if (mean_out && L.active) mean_out[(static_cast<size_t>(t) * N + L.n) * AW + k] = mu;
                a[k] = clampf(fmaf(sd, eps[k], mu), -1.0f, 1.0f);
            } else {
                a[k] = clampf(__uint_as_float(sw[0]), -1.0f, 1.0f);
            }
        }
        // ---- the env step ---------------------------------------------------------------------------------------------------------
        StepOut out;
        env_step<PID, EXT, false, AW, ACT, false>(P, C, flags, 1, L, make_float4(a[0], a[1], a[2], a[3]), tgx, tgy, tgz, true, ipose,
                                                    ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], ip[6], nullptr, nullptr, c, out);
#pragma unroll
        for (int i = 0; i < 12; ++i) o[i] = out.o[i];
        if (HIST) {                                          // the history moves up by one action, the new one goes to its end
            uint32_t nh[2] = {0u, 0u}, nl[2] = {0u, 0u};
            if (AW == 4) {
                split_pair(a[0], a[1], nh[0], nl[0]);
                split_pair(a[2], a[3], nh[1], nl[1]);
#pragma unroll
                for (int i = 6; i < NP - 2; ++i) { Fhi[i] = Fhi[i + 2]; Flo[i] = Flo[i + 2]; }
                Fhi[NP - 2] = nh[0]; Flo[NP - 2] = nl[0]; Fhi[NP - 1] = nh[1]; Flo[NP - 1] = nl[1];
            } else if (AW == 3) {                            // three bf16 elements: every register takes the upper half of its
                uint32_t xh[2], xl[2];                       // successor and the lower half of the one after (v_alignbit_b32)
                split_pair(a[0], a[1], xh[0], xl[0]);
                split_pair(a[2], 0.0f, xh[1], xl[1]);
#pragma unroll
                for (int i = 6; i < NP; ++i) {
                    const uint32_t h1 = i + 1 < NP ? Fhi[i + 1] : xh[i + 1 - NP], h2 = i + 2 < NP ? Fhi[i + 2] : xh[i + 2 - NP];
                    const uint32_t l1 = i + 1 < NP ? Flo[i + 1] : xl[i + 1 - NP], l2 = i + 2 < NP ? Flo[i + 2] : xl[i + 2 - NP];
                    Fhi[i] = __builtin_amdgcn_alignbit(h2, h1, 16);
                    Flo[i] = __builtin_amdgcn_alignbit(l2, l1, 16);
                }
            } else {                                         // AW == 1: one bf16 element per step
                split_pair(0.0f, a[0], nh[0], nl[0]);         // the new action in the upper half
#pragma unroll
                for (int i = 6; i < NP - 1; ++i) {
                    Fhi[i] = __builtin_amdgcn_alignbit(Fhi[i + 1], Fhi[i], 16);
                    Flo[i] = __builtin_amdgcn_alignbit(Flo[i + 1], Flo[i], 16);
                }
                Fhi[NP - 1] = (Fhi[NP - 1] >> 16) | nh[0];
                Flo[NP - 1] = (Flo[NP - 1] >> 16) | nl[0];
            }
        }
        if (L.active) {
            f4v* orow = reinterpret_cast<f4v*>(obs12 + t * obs_step + static_cast<size_t>(L.n) * 12);
            __builtin_nontemporal_store(f4v{o[0], o[1], o[2], o[3]}, orow);          // (written once per step: streamed out)
            __builtin_nontemporal_store(f4v{o[4], o[5], o[6], o[7]}, orow + 1);
            __builtin_nontemporal_store(f4v{o[8], o[9], o[10], o[11]}, orow + 2);
            __builtin_nontemporal_store(out.rew, &reward[t * env_stride + L.env]);
            __builtin_nontemporal_store(static_cast<uint8_t>(out.term ? 1 : 0), &terminated[t * env_stride + L.env]);
            __builtin_nontemporal_store(static_cast<uint8_t>(out.trunc ? 1 : 0), &truncated[t * env_stride + L.env]);
            if (actions_out) {
                float* ar = actions_out + (static_cast<size_t>(t) * N + L.n) * AW;
                if (AW == 4) *reinterpret_cast<float4*>(ar) = make_float4(a[0], a[1], a[2], a[3]);
                else { ar[0] = a[0]; if (AW == 3) { ar[1] = a[1]; ar[2] = a[2]; } }
            }
            if (term_obs12 && out.reset) {                   // the last observation of the episode that ended in this step (SB3's
                f4v* tr = reinterpret_cast<f4v*>(term_obs12 + t * obs_step + static_cast<size_t>(L.n) * 12);   // info["terminal_observation"])
                tr[0] = f4v{out.to[0], out.to[1], out.to[2], out.to[3]};
                tr[1] = f4v{out.to[4], out.to[5], out.to[6], out.to[7]};
                tr[2] = f4v{out.to[8], out.to[9], out.to[10], out.to[11]};
            }
            if (ring) {                                      // exact fp32 action into both halves of the double ring
                float* r0 = S.act_ring + static_cast<size_t>(ring_q) * slot + static_cast<size_t>(L.n) * AW;
                float* r1 = r0 + static_cast<size_t>(S.hist_len) * slot;
                if (AW == 4) { *reinterpret_cast<float4*>(r0) = make_float4(a[0], a[1], a[2], a[3]); *reinterpret_cast<float4*>(r1) = make_float4(a[0], a[1], a[2], a[3]); }
                else { r0[0] = a[0]; r1[0] = a[0]; if (AW == 3) { r0[1] = a[1]; r0[2] = a[2]; r1[1] = a[1]; r1[2] = a[2]; } }
            }
        }
        ring_q = ring_q + 1 == S.hist_len ? 0 : ring_q + 1;
    }
    if (!L.active) return;
    store_carry<PID>(S, L, c);
    if (ring) S.ring_pos[L.env] = ring_q;
}

template <bool PID, int AW, int ACT, int NK1, bool RELU>
__global__ __launch_bounds__(kBlock) void gpd_rollout_policy_kernel(
    const GpdParams P, const GpdState S, const GpdStepCfg C, const Span T, const GpdPolicy Pol,
    const float* __restrict__ obs12_in, const float* __restrict__ target_pos, const float* __restrict__ init_pose,
    float* __restrict__ actions_out, float* __restrict__ obs12, float* __restrict__ reward, uint8_t* __restrict__ terminated,
    uint8_t* __restrict__ truncated, float* __restrict__ term_obs12) {
    rollout_policy_body<PID, AW, ACT, NK1, RELU, false>(P, S, C, T, Pol, obs12_in, target_pos, init_pose, actions_out, obs12, reward,
                                                        terminated, truncated, nullptr, nullptr, 0.0f, 0.0f, 0.0f, 0.0f, term_obs12);
}

template <int AW, int ACT, int NK1, bool RELU>
__global__ __launch_bounds__(kBlock) void gpd_rollout_policy_noise_kernel(
    const GpdParams P, const GpdState S, const GpdStepCfg C, const Span T, const GpdPolicy Pol,
    const float* __restrict__ obs12_in, const float* __restrict__ target_pos, const float* __restrict__ init_pose,
    float* __restrict__ actions_out, float* __restrict__ obs12, float* __restrict__ reward, uint8_t* __restrict__ terminated,
    uint8_t* __restrict__ truncated, const float* __restrict__ noise, float* __restrict__ mean_out, const float4 sd,
    float* __restrict__ term_obs12) {
    rollout_policy_body<false, AW, ACT, NK1, RELU, true>(P, S, C, T, Pol, obs12_in, target_pos, init_pose, actions_out, obs12, reward,
                                                         terminated, truncated, noise, mean_out, sd.x, sd.y, sd.z, sd.w, term_obs12);
}

#ifndef GPD_POLICY_TU
// ------------------------------------------------------------------------------------------------
// masked reset (envs/BaseAviary.py:451-477)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gpd_reset_kernel(const GpdState S, const float* __restrict__ init_pose,
                                                           int init_per_env, const uint8_t* __restrict__ mask,
                                                           int num_envs, int D, int reset_pid,
                                                           float* __restrict__ obs12) {
    const int64_t N = static_cast<int64_t>(num_envs) * D;
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= N) return;
    const int64_t env = n / D;
    const int d = static_cast<int>(n - env * D);
    if (mask && !mask[env]) return;
    const int64_t ld = S.ld;
    const float* ip = init_pose + (init_per_env ? n * 7 : static_cast<int64_t>(d) * 7);
    float* kin = S.kin + n;
    kin[0 * ld] = ip[0]; kin[1 * ld] = ip[1]; kin[2 * ld] = ip[2];
    kin[3 * ld] = ip[3]; kin[4 * ld] = ip[4]; kin[5 * ld] = ip[5]; kin[6 * ld] = ip[6];
#pragma unroll
    for (int r = 7; r < 13; ++r) kin[r * ld] = 0.0f;
    if (S.last_rpm) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S.last_rpm[r * ld + n] = 0.0f;
    }
    if (reset_pid && S.pid) {
#pragma unroll
        for (int r = 0; r < 9; ++r) S.pid[r * ld + n] = 0.0f;
    }
    if (d == 0) S.step_counter[env] = 0;
    if (obs12) {
        float roll, pitch, yaw;
        quat_to_rpy(ip[3], ip[4], ip[5], ip[6], roll, pitch, yaw);
        store_obs12(obs12, n, ip[0], ip[1], ip[2], roll, pitch, yaw, 0, 0, 0, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// Full KIN observation rows: [ pos | rpy | vel | ang_v | the H most recent actions, oldest first ]
// (envs/BaseRLAviary.py:307-320), materialised on request from the action ring (GpdState: a double ring [2H][N][A],
// slot-major; the H most recent actions of an aviary are the H consecutive slots starting at its ring_pos).  Pure data
// movement: one lane per output float, so that a wave writes 256 contiguous bytes.
// ------------------------------------------------------------------------------------------------
// A workgroup owns R consecutive drones (R = 64, 32, 16 ...: as many as fit WHOLE rows into 48 KiB of LDS) and (blockIdx.y) one
// step; it is a transpose through LDS, and what it writes is ONE contiguous block of global memory -- the R rows follow each
// other -- streamed out as 16-byte pieces:
//   in : a history slot -- of the ring, or an action block of the call -- is a contiguous [N][A] block -> R lanes read the R
//        drones' A floats of slot i (one 16-byte load per lane for A = 4: R x 16 contiguous bytes), 256 / R slots in flight;
//        the twelve kinematic floats of the R rows are contiguous in obs12;
//   out: R x (12 + H*A) floats, contiguous.
// (History of this kernel at N = 65 536, 240 Hz rows of 492 floats: one lane per output float, sixteen 16-byte pieces from
// sixteen 1 MB-apart slots per wave: 116 us per step; 64 drones x 16 slots per workgroup, a 256-byte segment per drone written
// by whichever workgroup held that chunk: 57-60 us -- segments that start and end inside 128-byte lines, rewritten piecemeal;
// whole rows per workgroup: see DESIGN.md section 3.3.)  Two callers:
//   gpd_hist_rows  rows of the CURRENT state (after gpd_step, which pushed its action itself): slot i of the window is ring slot
//                  ring_pos + i;
//   gpd_full_obs   rows of the K steps of a rollout: slot i of step t's window is the action of step s = t - (H-1) + i of this
//                  call (s >= 0), or -- for steps before the call -- ring slot ring_pos + H + s, the ring as the rollout found it.
__global__ __launch_bounds__(kBlock) void gpd_hist_rows_kernel(uint32_t N, int D, int A, int H, int R, const float* __restrict__ ring,
                                                               const int32_t* __restrict__ ring_pos,
                                                               const float* __restrict__ obs12, int64_t obs_stride,
                                                               const float* __restrict__ actions, int64_t act_stride,
                                                               float* __restrict__ out, int64_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) float hist_tile[];     // [R][Wp], Wp = W | 1 (odd: drone-strided accesses spread over the banks)
    const int tid = threadIdx.x;
    const uint32_t n0 = blockIdx.x * static_cast<uint32_t>(R);
    const int t = blockIdx.y;                                // step of the call (0 for the current-state rows)
    const uint32_t W = 12u + static_cast<uint32_t>(H * A), Wp = W | 1u;
    const uint32_t rows = N - n0 < static_cast<uint32_t>(R) ? N - n0 : static_cast<uint32_t>(R);
    obs12 += t * obs_stride;
    out += t * out_stride;
    // ---- in: history slots ----
    const int d = tid % R, sl0 = tid / R, spp = kBlock / R;  // this thread's drone; slots per pass
    const uint32_t n = n0 + d;
    const bool have = static_cast<uint32_t>(d) < rows;
    const int p = have ? ring_pos[n / static_cast<uint32_t>(D)] : 0;
    for (int i = sl0; i < H; i += spp) {
        const int s = t - (H - 1) + i;                       // (rollout rows) the step of this call the action belongs to
        const float* src = (actions && s >= 0) ? actions + s * act_stride + static_cast<size_t>(n) * A
                                               : ring + (static_cast<size_t>(p + (actions ? H + s : i)) * N + n) * A;
        if (have) {
            float* dst = hist_tile + d * Wp + 12 + i * A;
            if (A == 4) { const float4 q = *reinterpret_cast<const float4*>(src); dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w; }
            else for (int a = 0; a < A; ++a) dst[a] = src[a];
        }
    }
    // ---- in: the kinematic part, rows x 12 floats, contiguous in obs12 ----
    for (uint32_t j = tid; j < rows * 12u; j += kBlock) {
        const uint32_t r = j / 12u, c = j - r * 12u;
        hist_tile[r * Wp + c] = obs12[static_cast<size_t>(n0) * 12 + j];
    }
    __syncthreads();
    // ---- out: rows x W floats, contiguous ----
    float* const dst = out + static_cast<size_t>(n0) * W;
    if ((W % 4u) == 0u && (reinterpret_cast<uintptr_t>(dst) % 16u) == 0u) {
        for (uint32_t j = tid; j < rows * (W / 4u); j += kBlock) {
            const uint32_t f = 4u * j, r = f / W, c = f - r * W;
            const float* tp = hist_tile + r * Wp + c;
            __builtin_nontemporal_store(f4v{tp[0], tp[1], tp[2], tp[3]}, reinterpret_cast<f4v*>(dst + f));
        }
    } else {
        for (uint32_t f = tid; f < rows * W; f += kBlock) {
            const uint32_t r = f / W, c = f - r * W;
            __builtin_nontemporal_store(hist_tile[r * Wp + c], dst + f);
        }
    }
}

// pushes the actions of the K steps of a call into the ring (the last H of them survive): blockIdx.y = 0 is the newest step
__global__ __launch_bounds__(kBlock) void gpd_hist_push_kernel(int K, uint32_t N, int D, int A, int H,
                                                               const float* __restrict__ actions, int64_t act_stride,
                                                               float* __restrict__ ring, const int32_t* __restrict__ ring_pos) {
    // one drone per thread (A = 4: one 16-byte load, two 16-byte stores -- the float-per-thread form took 68 us for the 64 steps
    // of a rollout of 65 536 drones, most of it integer division)
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const int s = K - 1 - static_cast<int>(blockIdx.y);
    if (n >= N || s < 0) return;
    const size_t NA = static_cast<size_t>(N) * A;
    int q = ring_pos[n / static_cast<uint32_t>(D)] + s;
    q -= (q / H) * H;
    const float* src = actions + s * act_stride + static_cast<size_t>(n) * A;
    float* r0 = ring + static_cast<size_t>(q) * NA + static_cast<size_t>(n) * A;
    float* r1 = r0 + static_cast<size_t>(H) * NA;
    if (A == 4) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        *reinterpret_cast<float4*>(r0) = v;
        *reinterpret_cast<float4*>(r1) = v;
    } else {
        for (int a = 0; a < A; ++a) { const float v = src[a]; r0[a] = v; r1[a] = v; }
    }
}

// ... and then, in a launch of its own (every lane of the push has read the old value), the aviaries' ring positions advance
__global__ __launch_bounds__(kBlock) void gpd_hist_advance_kernel(int K, int E, int H, int32_t* __restrict__ ring_pos) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e < E) ring_pos[e] = (ring_pos[e] + K) % H;
}

// one row of BaseAviary._getDroneStateVector (envs/BaseAviary.py:541-561) from the SoA state and the obs12 row of the latest step
__device__ __forceinline__ void state20_row(const GpdState& S, const float* __restrict__ obs12, float* __restrict__ out, int64_t n) {
    const int64_t ld = S.ld;
    const float* kin = S.kin + n;
    const float* o = obs12 + n * 12;
    float4* w = reinterpret_cast<float4*>(out + n * 20);
    const float l0 = S.last_rpm ? S.last_rpm[0 * ld + n] : 0.0f, l1 = S.last_rpm ? S.last_rpm[1 * ld + n] : 0.0f;
    const float l2 = S.last_rpm ? S.last_rpm[2 * ld + n] : 0.0f, l3 = S.last_rpm ? S.last_rpm[3 * ld + n] : 0.0f;
    w[0] = make_float4(kin[0 * ld], kin[1 * ld], kin[2 * ld], kin[3 * ld]);
    w[1] = make_float4(kin[4 * ld], kin[5 * ld], kin[6 * ld], o[3]);
    w[2] = make_float4(o[4], o[5], kin[7 * ld], kin[8 * ld]);
    w[3] = make_float4(kin[9 * ld], o[9], o[10], o[11]);
    w[4] = make_float4(l0, l1, l2, l3);
}

// ------------------------------------------------------------------------------------------------
// Downwash inside ONE aviary of any size (envs/BaseAviary.py:785-811): uniform 2-D grid, counting sort by cell,
// 3x3 neighbourhood search.  Three kernels per physics sub-step (count, scan + scatter, force).
// ------------------------------------------------------------------------------------------------
// cell of a position: the grid is periodic (cells wrap around), so drones that leave the box the grid was laid over
// keep spreading over all cells instead of piling up at its border; far-apart drones that alias into neighbouring
// cells are rejected by the exact distance test
struct DwGrid {            // uniform periodic x-y grid of >= 10 m cells, each cell split into nz height bins (nz = 1: none)
    float inv_cell, x0, y0;
    int nx, ny;
    float z0, inv_zbin;
    int nz;
};

__device__ __forceinline__ int cell_of(float x, float y, const DwGrid& G) {
    // (clamped before the conversion: float -> int is undefined beyond the int range, and a drone flung 1e10 m away by a
    // diverging downwash term must still land in SOME cell -- which one is irrelevant, every candidate pair is distance-tested)
    const float fx = fminf(fmaxf(floorf((x - G.x0) * G.inv_cell), -1.0e9f), 1.0e9f), fy = fminf(fmaxf(floorf((y - G.y0) * G.inv_cell), -1.0e9f), 1.0e9f);
    int cx = static_cast<int>(fx) % G.nx, cy = static_cast<int>(fy) % G.ny;
    cx = cx < 0 ? cx + G.nx : cx;
    cy = cy < 0 ? cy + G.ny : cy;
    return cy * G.nx + cx;
}
// sort key: cell * nz + height bin.  Bin b holds z0 + b/inv_zbin <= z < z0 + (b+1)/inv_zbin; bin 0 also everything below,
// bin nz-1 everything above.  Inside a cell the drones are thereby ordered by height bin, and a group of drones whose lowest
// bin is b can skip every candidate in a bin below b: such a candidate is below all of them (dz < 0), the model ignores it.
__device__ __forceinline__ int key_of(float x, float y, float z, const DwGrid& G) {
    const float fz = fminf(fmaxf(floorf((z - G.z0) * G.inv_zbin), 0.0f), static_cast<float>(G.nz - 1));
    return cell_of(x, y, G) * G.nz + static_cast<int>(fz);
}

// The sort's two passes visit the drones in `visit` order (NULL: 0, 1, 2 ...).  Handing in the previous call's `order`
// makes consecutive lanes fall into the same cell (drones move centimetres per sub-step), and the lanes of a wave that
// share a cell then issue ONE atomic for their whole run instead of one each: ~64 same-address atomics per cell
// become a handful.  run_of(): this lane's run among the wave's consecutive equal cells -> (first lane, length).
__device__ __forceinline__ void run_of(int c, int lane, int& head_lane, int& len) {
    const int prev_c = __shfl_up(c, 1);
    const bool head = lane == 0 || c != prev_c;
    const uint64_t heads = __builtin_amdgcn_ballot_w64(head);
    const uint64_t upto = heads & (((lane == 63) ? 0ull : (2ull << lane)) - 1ull);      // heads at or below this lane
    head_lane = 63 - __builtin_clzll(upto);
    const uint64_t above = heads & ~(((head_lane == 63) ? 0ull : (2ull << head_lane)) - 1ull);   // heads above the run's first lane
    len = (above ? __builtin_ctzll(above) : 64) - head_lane;
}

// VEC: the pass over all drones also writes their [n][20] state vectors (what gpd_state_vectors does) -- a caller that steps
// a swarm needs both after every step, and at this size a launch costs more than the rows.
// where the positions come from: the SoA state block (gpd_downwash_global), or the packed [rows][4] array of a GpdSwarm
struct DwPos { const float* kin; int64_t ld; const float4* pos4; };
__device__ __forceinline__ void dw_pos(const DwPos& S, int d, float& x, float& y, float& z) {
    if (S.pos4) { const float4 p = S.pos4[d]; x = p.x; y = p.y; z = p.z; }
    else { x = S.kin[d]; y = S.kin[S.ld + d]; z = S.kin[2 * S.ld + d]; }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void dwg_count_kernel(const DwPos src, int n, const DwGrid G,
                                                           const int* __restrict__ visit, int* __restrict__ count,
                                                           const GpdState VS, const float* __restrict__ vec_obs12,
                                                           float* __restrict__ vec_out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int c = -1 - lane;                                      // (no drone: a run of its own, no atomic)
    if (i < n) {
        const int d = visit ? visit[i] : i;
        if constexpr (VEC) state20_row(VS, vec_obs12, vec_out, i);      // (row i, not row d: coalesced, the two jobs only share the launch)
        float x, y, z;
        dw_pos(src, d, x, y, z);
        // a drone whose position is no longer finite (the downwash model diverges when two drones pass each other
        // vertically, dz -> 0+) takes no part: it would otherwise alias into cell 0 together with every other such drone
        if (isfinite(x) && isfinite(y) && isfinite(z)) c = key_of(x, y, z, G);
    }
    int head_lane, len;
    run_of(c, lane, head_lane, len);
    if (lane == head_lane && c >= 0) atomicAdd(&count[c], len);
}

// exclusive scan of count[0..cells) into start[0..cells], one workgroup (only for more keys than the scatter kernel scans itself)
__global__ __launch_bounds__(1024) void dwg_scan_kernel(const int* __restrict__ count, int* __restrict__ start, int cells) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (cells + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, cells);
    int s = 0;
    for (int c = lo; c < hi; ++c) s += count[c];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {            // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t == 0 ? 0 : part[t - 1];
    for (int c = lo; c < hi; ++c) { const int k = count[c]; start[c] = run; run += k; }
    if (t == 1023) start[cells] = part[1023];
}

// FUSED (up to kDwScanMax keys): every workgroup turns the counts into start offsets ITSELF, in LDS -- a few thousand integers
// from L2 and eight rounds of a 256-wide scan cost less than the launch of a scan kernel between count and scatter (the
// dependent launches, not the work in them, are what a swarm sub-step is made of); workgroup 0 also
// writes them out for the force kernel.
constexpr int kDwScanMax = 4096;
// what a GpdSwarm binning writes on top of the sort (all NULL / 0 for gpd_downwash_global)
struct DwBinOut {
    int* slot_key;         // [n] sort key of every sorted slot
    int* slot_of;          // [n] or NULL: row -> sorted slot (rows without a finite position: -1)
    int* visit_out;        // [n] or NULL: a second copy of `order`, for the NEXT binning to visit the rows in
    int* list_ok;          // [ceil(n / 64)] or NULL: the groups' wake lists belong to the previous binning: all back to 0
    float4* bin_pos;       // [n] x, y, z of every row at this binning
    float* pos4_w;         // pos4 as floats: the sums and maxima in every rank's meta rows (the last meta_rows of its slab) go back to 0
    float* drift;          // [4] ... and so does the common drift [0..1]; [2]: the margin of the wake lists of THIS binning (below)
    int slab, world, meta_rows;
    int own_lo, own_cnt;   // rows whose force lands in dw_out[row - own_lo]
    float list_delta;      // the largest margin the skin allows
    int list_adapt;        // 1: the margin follows the displacement the interval that ends here has seen
};
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void dwg_scatter_kernel(const DwPos src, int n, const DwGrid G,
                                                             const int* __restrict__ visit, const int* __restrict__ count,
                                                             int* __restrict__ cursor, int* __restrict__ start_g,
                                                             int* __restrict__ order, float4* __restrict__ sorted,
                                                             float* __restrict__ dw_out, const DwBinOut B) {
    __shared__ int lstart[FUSED ? kDwScanMax + 1 : 1];
    __shared__ int part[FUSED ? kBlock : 1];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int keys = G.nx * G.ny * G.nz;
    if constexpr (FUSED) {
        const int t = threadIdx.x;
        const int per = (keys + kBlock - 1) / kBlock;
        const int lo = t * per, hi = min(lo + per, keys);
        int sum = 0;
        for (int k = lo; k < hi; ++k) sum += count[k];
        part[t] = sum;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {      // Hillis-Steele inclusive scan of the 256 partial sums
            const int v = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        int run = t == 0 ? 0 : part[t - 1];
        for (int k = lo; k < hi; ++k) { lstart[k] = run; run += count[k]; }
        if (t == kBlock - 1) lstart[keys] = part[kBlock - 1];
        __syncthreads();
        if (blockIdx.x == 0) for (int k = t; k <= keys; k += kBlock) start_g[k] = lstart[k];
    }
    auto start = [&](int k) { if constexpr (FUSED) return lstart[k]; else return start_g[k]; };
    if (B.pos4_w && blockIdx.x == 0) {                     // every rank bins on the same sub-steps: all displacements restart
        __shared__ float dmax_red[kBlock / 64];
        float d2 = 0.0f;                                   // ... the largest one of the interval that ends here first
        for (int k = threadIdx.x; k < B.world * B.meta_rows; k += kBlock) {
            float* row = B.pos4_w + (static_cast<size_t>(k / B.meta_rows + 1) * B.slab - B.meta_rows + k % B.meta_rows) * 4;
            d2 = fmaxf(d2, row[3]);
            row[1] = 0.0f; row[2] = 0.0f; row[3] = 0.0f;
        }
        if (B.drift) {
            // The wake lists' margin: a pair is listed if it could pass the model's tests after both drones have moved `delta`,
            // and a list is replayed while no drone has.  The skin allows 0.49 (cell - 10 m) -- at 10.5 m, 0.245 m and 40 % more
            // pairs than the exact tests keep; a swarm that moved 5 mm between the last two binnings (relative to its common
            // drift) needs nothing like it.  Three times what the ending interval saw, at least a centimetre, at most what the
            // skin allows; a swarm that then moves further than that sweeps until the next binning (exact either way).
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, off));
            if (lane == 0) dmax_red[threadIdx.x >> 6] = d2;
            __syncthreads();
            if (threadIdx.x == 0) {
                float m = dmax_red[0];
#pragma unroll
                for (int k = 1; k < kBlock / 64; ++k) m = fmaxf(m, dmax_red[k]);
                B.drift[0] = 0.0f; B.drift[1] = 0.0f;
                // (m == 0 exactly: nothing has moved -- the binning right after a pack / reset, where the interval that "ended" says
                // nothing about the one that starts: the full margin, not the 1 cm floor a swarm faster than 0.15 m/s outruns at once)
                B.drift[2] = (B.list_adapt && m > 0.0f) ? fminf(B.list_delta, fmaxf(0.01f, 3.0f * sqrtf(m))) : B.list_delta;
            }
        }
    }
    if (B.list_ok && i < (n + 63) / 64) B.list_ok[i] = 0;
    int c = -1 - lane, d = 0;
    float x = 0.0f, y = 0.0f, z = 0.0f;
    if (i < n) {
        d = visit ? visit[i] : i;
        dw_pos(src, d, x, y, z);
        if (B.bin_pos) B.bin_pos[d] = make_float4(x, y, z, 0.0f);
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            c = key_of(x, y, z, G);
        } else {
            if (B.slot_of) B.slot_of[d] = -1;
            // (see dwg_count_kernel) no force on it, none from it; it keeps a slot behind the sorted drones so that `order`
            // stays a permutation (the next call visits the drones in this order)
            if (d >= B.own_lo && d < B.own_lo + B.own_cnt) dw_out[d - B.own_lo] = 0.0f;
            const int slot = start(keys) + atomicAdd(&cursor[keys], 1);
            order[slot] = d;
            if (B.visit_out) B.visit_out[slot] = d;
        }
    }
    int head_lane, len;
    run_of(c, lane, head_lane, len);
    int base = 0;
    if (lane == head_lane && c >= 0) base = start(c) + atomicAdd(&cursor[c], len);   // one atomic per run of equal keys
    base = __shfl(base, head_lane);
    if (c >= 0) {
        const int slot = base + (lane - head_lane);
        order[slot] = d;
        if (sorted) sorted[slot] = make_float4(x, y, z, __int_as_float(c));
        if (B.slot_key) B.slot_key[slot] = c;
        if (B.slot_of) B.slot_of[d] = slot;
        if (B.visit_out) B.visit_out[slot] = d;
    }
}

// Every drone sweeps the candidates of the 3x3 cells around its own, staged through LDS.  A lane = a drone, and of the ~650
// candidates of a drone ~100 are above it and within 10 m, ~35 close enough for a contribution the fixed-point sum resolves:
// evaluating the model (two reciprocals, an exponential, a conversion: ~36 issue slots) for every candidate, as the first version of this kernel did, spends 95 % of the vector unit on masked-off lanes --
// and with 64 lanes SOME lane nearly always passes, so no wave-level branch ever skips it.  Hence test and evaluation are
// separated, per chunk of 32 candidates:
//   A  the tests only, two candidates per packed instruction, no compare and no branch: with d = (candidate - drone),
//      q = dxy^2 - min(100.01, 80.02 beta(dz)^2) (the 10 m cut-off and the arg < 40 cut, without a division) and the sign
//      bits of q and of -dz ANDed and shifted into a per-lane 32-bit mask (v_and + v_alignbit): 7 slots per candidate, LDS reads
//      at wave-uniform addresses;
//   Q  the set bits become (drone lane, candidate) pairs in a per-wave LDS queue: a DPP prefix sum of the lanes' popcounts
//      gives each lane its slice;
//   E  whenever the queue holds 64 pairs, every lane takes one: drone through ds_bpermute, candidate gathered from the tile, the
//      EXACT tests and the model, and a 64-bit LDS atomic onto the drone's fixed-point sum.  All 64 lanes busy.
// Phase A is a conservative filter (100.01 / 80.02 instead of 100 / 80): it may queue a pair that the exact tests of E then
// reject, never the reverse.  The per-pair arithmetic and the integer sum are those of the first version, so the result is the
// same bit for bit -- except that the first version also kept pairs beyond arg = 40 when alpha > 1e8 N (two drones within
// 28 micrometres of the same height); each such pair now drops less than 4.3e-18 alpha.
constexpr int kDwTile = 1024;     // candidates staged in LDS at a time (3 x 4 KiB, x / y / z planes)
constexpr int kDwChunk = 32;      // candidates per mask word
constexpr int kDwBatches = 4;     // queued batches of 64 pairs that trigger an evaluation (1 .. 8: no measurable difference)
constexpr int kDwQueue = 64 * kDwBatches + kDwChunk * 64;   // pending (< 64 kDwBatches) + everything one chunk can add

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v) {   // v + (v of the lane the DPP control selects; 0 where there is none / the row is masked)
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// Reduction over the 64 lanes of a wave, the result in every lane, without a trip through the LDS crossbar per level (six
// dependent ds_bpermute round trips, ~0.25 us on a path that has nothing else to issue): four DPP steps inside the rows of 16
// lanes -- quad permutes for xor 1 / xor 2; after them the four lanes of a quad hold one value, so the row_half_mirror / row_mirror
// permutes read the same VALUE lane ^ 4 / lane ^ 8 would -- then the four rows' values as scalars (v_readlane), combined
// (r0 . r1) . (r2 . r3) in every lane: bit for bit the ascending xor butterfly (1, 2, 4, .. 32) of a commutative `op`.  All 64 lanes
// must be executing.
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <class Op>
__device__ __forceinline__ float wave_allreduce(float v, Op op) {
    v = op(v, dpp_perm<0xB1>(v));                          // quad_perm [1, 0, 3, 2]
    v = op(v, dpp_perm<0x4E>(v));                          // quad_perm [2, 3, 0, 1]
    v = op(v, dpp_perm<0x141>(v));                         // row_half_mirror
    v = op(v, dpp_perm<0x140>(v));                         // row_mirror
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_max(float v) { return wave_allreduce(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ float wave_sum(float v) { return wave_allreduce(v, [](float a, float b) { return a + b; }); }

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v = dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8      -> inclusive within each row of 16
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15   -> rows 1 and 3 add the total of the row before
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31   -> rows 2 and 3 add the total of the first half
    return v;
}

// One workgroup per 64 CONSECUTIVE drones of the sorted array (lane = drone; all four waves hold the same 64 and split the
// candidates four ways -- the partial sums are integers, the split changes no bit).  Sorted by cell, row-major, 64 consecutive
// drones cover one or a few cells of one grid row (a dense swarm: one or two; a sparse one: many), and their candidates are
// the 2R+1 rows around it, each a CONTIGUOUS stretch of the sorted array from R cells left of the first to R cells right of
// the last: three runs for R = 1, six when the periodic grid wraps.  Every lane is busy whatever the cells hold -- a workgroup
// per CELL, as in the first version, swept all candidates a second time for the few drones beyond the 64th of a cell.  A group
// that straddles the end of a grid row is swept once per row segment, with the lanes of the other segment switched off.
//
// GpdSwarm (DwWorld.pos4 != NULL): the sort is STALE -- `order`, `start` and the keys are those of the last binning, the
// positions (of the group's drones and of the candidates) are the current ones, read through slot -> row -> pos4.  Two drones
// within 10 m of each other now were within 10 m + 2 dmax at the binning (dmax: the largest lateral displacement of any drone
// since, the maximum of the ranks' meta rows), i.e. at most R = ceil((10 + 2 dmax) / cell) cells apart: the search widens to
// (2R+1) x (2R+1) cells and stays exact.  R beyond kDwMaxR (or the grid): the group sweeps every sorted drone.  Only the rows
// own_lo .. own_lo + own_cnt - 1 get a force (the rank's drones); a group without one exits.
constexpr int kDwMaxR = 3, kDwMaxRuns = 2 * (2 * kDwMaxR + 1);
struct DwWorld {
    const float4* pos4;    // current positions by ROW (read through `order`), or NULL: `sorted` holds the current positions by SLOT
                           // (binned in this very call, or kept current by the step kernel: a single rank's world)
    const int* slot_key;   // sort key per slot, or NULL: the key sits in sorted[].w
    const float4* meta;    // the position array whose meta rows hold the ranks' dmax^2, or NULL: no displacement since the binning
    int own_lo, own_cnt;   // rows this launch produces forces for (dw_out[row - own_lo])
    int slab, world, meta_rows;
    float cell;
    float* drift;          // [2] out (the extra workgroup): the mean lateral displacement of all drones, for the NEXT sub-step's step kernel
    float inv_total;       // 1 / drones in the world
    int n_slots;           // entries of `order` / `sorted` / `slot_key` (what a slot index may be clamped to)
};
// Wake lists (GpdSwarm): what survives phase A changes little from one sub-step to the next -- drones move centimetres, the
// model's reach is metres.  The force launch right after a binning (MODE 1, "build") runs phase A with a margin -- a pair is
// kept if it COULD pass the exact tests once both drones have moved up to `delta` in any direction -- and writes every batch it
// evaluates to a per-wave list in HBM.  An entry is 32 bits: 6 bits lane (the drone of the group the pair belongs to) and 26 bits
// the candidate's ABSOLUTE index in the array its current position is read from (the sorted slot in a single-rank world, the row
// of pos4 in a shared one).  The launches until the next binning (MODE 2, "replay") therefore need neither the cell table nor
// the runs nor a staged tile: they read their batches back, GATHER the candidates' current positions straight from memory
// (a group's ~600 candidates are 10 KB that its four waves share: L2 hits after the first touch), and evaluate -- the same exact
// tests and integer sums.  (Round 3's entries were 16 bits, lane + slot of the LDS tile, and a replay launch re-staged the
// tiles like a sweep: of its 14 us, 3 went to dependent set-up loads and 3 to staging before the first pair was evaluated,
// profiles/r03_force_timeline.txt.)  Valid while no drone is further than delta from where it was binned (dmax, the quantity
// the search radius follows; delta is what keeps R = 1: just under half the skin cell - 10 m) and the group's list did not
// overflow; otherwise the launch sweeps as if there were no lists.  MODE 0: no lists (gpd_downwash_global, or none allocated).
constexpr int kDwMaxTiles = 16;                            // (entries of DwLists.nb per wave; [0]: the wave's batches)
struct DwLists {
    uint32_t* list;        // [groups][4 waves][cap * 64] entries, batch-major; 0xffffffff: no pair in this lane of the batch
    unsigned short* nb;    // [groups][4 waves][kDwMaxTiles]: [0] = batches the wave recorded
    int* ok;               // [groups] 1: the group's list is complete for the current binning
    int cap;               // batches per wave
    float delta;           // the displacement the lists allow for
};
template <int MODE>
__global__ __launch_bounds__(kBlock) void dwg_force_kernel(const GpdParams P, const DwGrid G, const DwWorld Wd, const DwLists Ls,
                                                           const int* __restrict__ start, const int* __restrict__ order,
                                                           const float4* __restrict__ sorted, float* __restrict__ dw_out,
                                                           int* __restrict__ cursor) {
    // one LDS block: the tile's x / y / z planes, then the four waves' queues; (build) the candidates' source indices beside them
    __shared__ __attribute__((aligned(16))) float lds_tile[3 * kDwTile + (kBlock / 64) * kDwQueue / 2];
    __shared__ int tsrc[MODE == 1 ? kDwTile : 1];
    float* const tx = lds_tile;
    float* const ty = lds_tile + kDwTile;
    float* const tz = lds_tile + 2 * kDwTile;
    unsigned short (*const queue)[kDwQueue] = reinterpret_cast<unsigned short (*)[kDwQueue]>(lds_tile + 3 * kDwTile);
    __shared__ unsigned long long sums[kBlock / 64][64];
    __shared__ int run0[kDwMaxRuns], pre[kDwMaxRuns + 1];  // first element of each candidate run; prefix sums of the run lengths
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSF)
    const unsigned long long tf0 = wall_clock64();
    unsigned long long tf1 = 0, tf2 = 0, tf3 = 0;
#endif
    const int nx = G.nx, ny = G.ny, nz = G.nz;
    const int keys = nx * ny * nz;
    // the sort's per-key counters / cursors are done with: leave them zeroed for the next binning (no memset node per call)
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < 2 * (keys + 1); k += gridDim.x * kBlock) cursor[k] = 0;   // (counts | cursors)
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (Wd.drift && blockIdx.x == gridDim.x - 1) {
        // (an EXTRA workgroup behind those of the sorted array: workgroup 0 used to do this in front of its own share, and with
        // all workgroups resident at once the launch lasts as long as its slowest one)
        // the swarm's common drift for the next sub-step (see swarm_tail): the sum of the workgroups' displacement sums, in an
        // order that depends on nothing but the layout -- every rank arrives at the same two floats
        float sx = 0.0f, sy = 0.0f;
        for (int r = 0; r < Wd.world; ++r) {               // (rank by rank: no division by a run-time meta_rows per element)
            const float4* const mr = Wd.meta + (static_cast<size_t>(r + 1) * Wd.slab - Wd.meta_rows);
            for (int k = threadIdx.x; k < Wd.meta_rows; k += kBlock) { const float4 m = mr[k]; sx += m.y; sy += m.z; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); }
        float* const red = reinterpret_cast<float*>(pre);
        if (lane == 0) { red[2 * wave] = sx; red[2 * wave + 1] = sy; }
        __syncthreads();
        if (threadIdx.x == 0) {
            Wd.drift[0] = (((red[0] + red[2]) + red[4]) + red[6]) * Wd.inv_total;
            Wd.drift[1] = (((red[1] + red[3]) + red[5]) + red[7]) * Wd.inv_total;
        }
        return;
    }
    // Everything the set-up needs from memory is requested HERE, before the first decision that depends on any of it: the
    // workgroup's early exits used to sit between the loads (drones sorted -> slot's row -> positions, keys, maxima, list
    // header: four dependent trips to memory, 3.2 of a replay launch's 15 us, profiles/r03_force_timeline.txt); slot indices
    // are clamped so that a workgroup which is about to leave reads valid memory.
    const int base = 64 * blockIdx.x;
    const int s = base + lane;
    const int sc = min(s, Wd.n_slots - 1);
    auto key_at = [&](int slot) { return Wd.slot_key ? Wd.slot_key[slot] : __float_as_int(sorted[slot].w); };
    auto pos_at = [&](int slot) { return Wd.pos4 ? Wd.pos4[order[slot]] : sorted[slot]; };
    const int sorted_n = start[keys];                      // drones with a finite position (the others: force 0, set by the sort)
    int row_l = order[sc];
    GPD_DBG(row_l >= 0 && row_l < Wd.n_slots, GPD_DBG_SLOT_ROW, row_l); row_l = GPD_DBG_CLAMP(row_l, 0, Wd.n_slots - 1);
    int key_l = key_at(sc);
    GPD_DBG(s >= start[keys] || (key_l >= 0 && key_l < keys), GPD_DBG_SORT_KEY, key_l);
    float4 me_l = Wd.pos4 ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : sorted[sc];
    // dmax^2: the largest of the step kernel's per-workgroup maxima (one meta row each, every rank's)
    // (rank by rank: an index k -> (rank, row) split would be two divisions by a run-time meta_rows per element)
    float d2 = 0.0f;
    const int tot = Wd.world * Wd.meta_rows;
    const bool few = tot <= 1024;                          // few: every wave reads them all (no LDS round, no barrier)
    auto read_maxima = [&]() {
        for (int r = 0; r < Wd.world; ++r) {
            const float4* const mr = Wd.meta + (static_cast<size_t>(r + 1) * Wd.slab - Wd.meta_rows);
            for (int k = few ? lane : static_cast<int>(threadIdx.x); k < Wd.meta_rows; k += few ? 64 : kBlock) d2 = fmaxf(d2, mr[k].w);
        }
    };
    // (many of them -- a world of 10^6 drones, or shared by many ranks -- and they are read behind the early exits instead: 7 of 8
    // groups of a rank of eight hold none of its drones, and were reading 64 rows per lane before they found out)
    if (Wd.meta && few) read_maxima();
    // (replay) whether the group's list is complete, this wave's batches on the first tile
    uint32_t* const my_list = MODE ? Ls.list + (static_cast<size_t>(blockIdx.x) * (kBlock / 64) + wave) * Ls.cap * 64 : nullptr;
    unsigned short* const my_nb = MODE ? Ls.nb + (static_cast<size_t>(blockIdx.x) * (kBlock / 64) + wave) * kDwMaxTiles : nullptr;
    int list_ok = 0, nb0 = 0;
    float delta = Ls.delta;                                // the lists' margin: what gpd_swarm_bin chose for this binning
    if (MODE && Wd.drift) delta = Wd.drift[2];
    // (replay) the wave's first four batches are requested HERE, with everything else the set-up needs: their addresses depend on
    // nothing but the workgroup, the wave and the lane (cap >= 4 is checked by the host; a wave with fewer batches reads stale
    // entries of its own list and never uses them)
    uint32_t e0 = 0xffffffffu, e1 = 0xffffffffu, e2 = 0xffffffffu, e3 = 0xffffffffu;
    if (MODE == 2) {
        list_ok = Ls.ok[blockIdx.x];
        nb0 = my_nb[0];
        e0 = my_list[lane]; e1 = my_list[64 + lane]; e2 = my_list[128 + lane]; e3 = my_list[192 + lane];
    }
    // (the loads above are all in flight; this is where they are waited for, together)
    asm volatile("" : "+v"(row_l), "+v"(key_l), "+v"(me_l.x), "+v"(me_l.y), "+v"(me_l.z), "+v"(d2), "+v"(list_ok), "+v"(nb0), "+v"(delta));
    if (MODE == 2) asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
    list_ok = __builtin_amdgcn_readfirstlane(list_ok); nb0 = __builtin_amdgcn_readfirstlane(nb0);
    if (base >= sorted_n) return;
    const bool have = s < sorted_n;
    const int my_row = have ? row_l : -1;
    const bool own = have && my_row >= Wd.own_lo && my_row < Wd.own_lo + Wd.own_cnt;
    if (__builtin_amdgcn_ballot_w64(own) == 0) return;     // (the four waves hold the same 64 slots: the whole workgroup leaves)
    unsigned short* const my_queue = queue[wave];
    unsigned long long* const my_sums = sums[wave];
    const float kr = 0.25f * P.prop_radius;
    const float cut = 8.9454f;                             // sqrt(80.02): arg < 40  <=>  dxy^2 < 80 beta^2
    const fp2 b1 = splat(-P.dw_coeff[1] * cut), b0 = splat(P.dw_coeff[2] * cut);
    float4 me = make_float4(0.0f, 0.0f, 3.0e38f, 0.0f);
    if (own) me = Wd.pos4 ? Wd.pos4[my_row] : me_l;
    // (key -> cell: nz = 1 for every grid SwarmAviary builds -- no division by a run-time value then)
    auto cell_of = [&](int key) { return nz == 1 ? key : key / nz; };
    const int my_cell = own ? cell_of(key_l) : -1;
    // search radius in cells
    int R = 1;
    if (Wd.meta) {
        // (many meta rows -- a world of 10^6 drones, or one shared by many ranks: gpd_swarm_step's own reduction left every
        // rank's maximum in the w of the rank's FIRST meta row, dwg_reduce_meta_kernel; one value per rank is read here.  Every
        // workgroup reading all 4 097 rows of a 1M-drone world was 1 GB of L2 traffic per force launch and a third of a replay
        // launch's vector instructions)
        if (!few) for (int r = lane; r < Wd.world; r += 64) d2 = fmaxf(d2, Wd.meta[static_cast<size_t>(r + 1) * Wd.slab - Wd.meta_rows].w);
        d2 = wave_max(d2);
        if (d2 > 0.0f) {                                   // (rounded up: a wider search is still exact, a narrower one is not)
            const float reach = 10.0f + 2.0002f * sqrtf(d2) + 2.0e-6f;
            const float cells = ceilf(reach / Wd.cell * 1.000001f);
            R = cells < 1.0e6f ? static_cast<int>(cells) : (1 << 20);          // (also catches inf)
        }
        R = __builtin_amdgcn_readfirstlane(R);
    }
    // replay this group's wake list?  (uniform for the workgroup)
    // (0.98: dmax <= 0.99 delta -- the build's margin test is evaluated in fp32, a pair exactly on its boundary must not matter)
    const bool replay = MODE == 2 && R == 1 && d2 <= 0.98f * (delta * delta) && list_ok != 0;
    // the group's first and last cell: the keys of its first and last sorted slot, which two of its lanes hold already
    const int c_first = cell_of(__builtin_amdgcn_readlane(key_l, 0));
    const int c_last = cell_of(__builtin_amdgcn_readlane(key_l, __builtin_amdgcn_readfirstlane(min(63, sorted_n - 1 - base))));
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSF)
    if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tf1 = wall_clock64(); }
#endif
    int lb = 0;                                            // (build) batches recorded so far
    // (build) the list is complete so far -- and only a build on the binning's own positions counts (a caller that builds later
    // gets no lists rather than lists whose margin was measured from somewhere else)
    bool rec_ok = R == 1 && d2 == 0.0f;
    const float m2 = MODE == 1 ? 2.0f * delta : 0.0f;      // (build) what two drones can have closed in on each other
    const float c2 = MODE == 1 ? cut * fabsf(P.dw_coeff[1]) * m2 : 0.0f;      // ... and what that adds to sqrt(80.02) |beta|
    const bool sweep_all = R > kDwMaxR;                    // the group's candidates: every sorted drone, one run
    const int nrows = sweep_all ? 0 : min(2 * R + 1, ny);
    my_sums[lane] = 0ull;                                  // sum of contributions in units of 2^-30 N: order-independent
    // one pair: drone `dl` of the group under the candidate at (cx, cy, cz) -- the exact tests and the model (:798-808), added to
    // the drone's sum (called by ALL lanes -- ds_bpermute reads nothing from a lane that is masked off -- with `valid` false
    // where there is no pair)
    auto pair_model = [&](int dl, bool valid, float cx, float cy, float cz) {
        const float px = __shfl(me.x, dl), py = __shfl(me.y, dl), pz = __shfl(me.z, dl);
        const float dz = cz - pz;
        const float ddx = cx - px, ddy = cy - py;
        const float dxy2 = fmaf(ddy, ddy, ddx * ddx);
        if (valid && dz > 0.0f && dxy2 < 100.0f) {     // dz > 0 and dxy < 10 m  (:800-801)
            const float ratio = kr * fast_rcp(dz);
            const float alpha = P.dw_coeff[0] * (ratio * ratio);
            const float beta = fmaf(P.dw_coeff[1], dz, P.dw_coeff[2]);
            const float ib = fast_rcp(beta);
            const float arg = 0.5f * (dxy2 * (ib * ib));
            if (arg < 40.0f) {                         // exp(-40) = 4e-18: below the 2^-31 N the sum resolves
                const float sc = (alpha * fast_exp(-arg)) * 1073741824.0f;
                // (< 4 N, i.e. unless two drones are centimetres apart: one conversion instead of the 14-instruction
                // float -> int64 sequence; the same integer either way)
                // (marked unlikely: without it the 14-instruction conversion is laid out as the fall-through and the usual case
                // pays two taken branches per batch to get around it)
                unsigned long long v;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(sc >= 0.0f && sc < 4.0e9f)) != 0, 0)) {
                    asm volatile("; a contribution of 4 N or more" ::: "memory");      // (keeps this a branch, not a select)
                    v = static_cast<unsigned long long>(__float2ll_rn(sc));
                } else {
                    v = static_cast<unsigned long long>(__float2uint_rn(sc));
                }
                atomicAdd(&my_sums[dl], v);
            }
        }
    };
    if (MODE == 2 && replay) {
        // REPLAY: the wave's recorded batches, four at a time, in a three-deep software pipeline -- the entries of batches b+8..
        // are requested while the candidates of b+4.. are gathered and b.. is evaluated (loads only in this loop: the waits the
        // compiler derives are exact counts).  An empty lane (0xffffffff) gathers index 0 and evaluates nothing.
        GPD_DBG(nb0 >= 0 && nb0 <= Ls.cap, GPD_DBG_LIST_COUNT, nb0);
        const int nbw = GPD_DBG_CLAMP(nb0, 0, Ls.cap);
        const uint32_t* const lp = my_list + lane;
        // A launch ends with its slowest workgroup, and that is the one with the most pairs (profiles/r04_swarm_force_timeline.txt:
        // 9.1 us against a median of 6.6; all workgroups are resident at once, nothing fills in behind it).  A wave with many batches
        // asks the SIMD's arbiter for priority over the three it shares the SIMD with: they have slack, it has none.
        // (thresholds 12 / 16 / 20 and 11 / 13 / 16 measure the same, profiles/r04_ab_swarm_replay_priority.txt; a scene where every
        // wave is above them gives every wave the same priority: nothing gained, nothing lost)
        if (nbw >= 12) {
            if (nbw >= 20) __builtin_amdgcn_s_setprio(3);
            else if (nbw >= 16) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        }
        auto cand = [&](uint32_t e) {
            // (clamped: the first four entries are read before the wave knows how many batches it has -- whatever a list holds,
            // the gather stays inside the array)
            GPD_DBG(e == 0xffffffffu || (e & 0x03ffffffu) < static_cast<uint32_t>(Wd.n_slots), GPD_DBG_LIST_ENTRY, e & 0x03ffffffu);
            const uint32_t idx = min(e == 0xffffffffu ? 0u : (e & 0x03ffffffu), static_cast<uint32_t>(Wd.n_slots - 1));
            return Wd.pos4 ? Wd.pos4[idx] : sorted[idx];
        };
        // (no branch around the load: beyond the wave's batches the last one is read again and discarded)
        auto entry = [&](int b) { const uint32_t e = lp[static_cast<size_t>(min(b, max(nbw - 1, 0))) * 64]; return b < nbw ? e : 0xffffffffu; };
        uint32_t f0 = entry(4), f1 = entry(5), f2 = entry(6), f3 = entry(7);
        float4 p0 = cand(e0), p1 = cand(e1), p2 = cand(e2), p3 = cand(e3);
        if (nbw < 4) { if (nbw < 1) e0 = 0xffffffffu; if (nbw < 2) e1 = 0xffffffffu; if (nbw < 3) e2 = 0xffffffffu; e3 = 0xffffffffu; }
        for (int b = 0; b < nbw; b += 4) {
            const uint32_t g0 = entry(b + 8), g1 = entry(b + 9), g2 = entry(b + 10), g3 = entry(b + 11);
            const float4 q0 = cand(f0), q1 = cand(f1), q2 = cand(f2), q3 = cand(f3);
            pair_model(static_cast<int>(e0 >> 26), e0 != 0xffffffffu, p0.x, p0.y, p0.z);
            pair_model(static_cast<int>(e1 >> 26), e1 != 0xffffffffu, p1.x, p1.y, p1.z);
            pair_model(static_cast<int>(e2 >> 26), e2 != 0xffffffffu, p2.x, p2.y, p2.z);
            pair_model(static_cast<int>(e3 >> 26), e3 != 0xffffffffu, p3.x, p3.y, p3.z);
            e0 = f0; e1 = f1; e2 = f2; e3 = f3; p0 = q0; p1 = q1; p2 = q2; p3 = q3;
            f0 = g0; f1 = g1; f2 = g2; f3 = g3;
        }
    } else
    for (int cs = c_first; cs <= c_last;) {                // row segments of the group's cells (nearly always one)
        const int cy = cs / nx;
        const int ce = sweep_all ? c_last : min(c_last, cy * nx + nx - 1);
        const int cxa = cs - cy * nx, w = ce - cs + 1 + 2 * R;     // columns cxa - R .. cxb + R, periodic
        const bool active = own && my_cell >= cs && my_cell <= ce;
        const float mez = active ? me.z : 3.0e38f;         // (switched off: nothing is above it)
        const bool fast = R == 1 && !sweep_all;            // the usual case: three rows, six runs, all in registers, no barrier
        int run0r[6], prer[7];
        const int nruns = sweep_all ? 1 : 2 * nrows;
        if (fast) {
            prer[0] = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int gy = cy + r - 1;                       // (one period at most: no modulo)
                gy = gy < 0 ? gy + ny : gy >= ny ? gy - ny : gy;
                const int row = gy * nx;
                int a0, a1, b1c;                           // run A: cells a0 .. a1; run B (the wrapped part): cells 0 .. b1c, or empty
                if (w >= nx) { a0 = 0; a1 = nx - 1; b1c = -1; }
                else {
                    const int a = cxa > 0 ? cxa - 1 : nx - 1;
                    a0 = a; a1 = min(a + w - 1, nx - 1);
                    b1c = a + w - 1 - nx;                  // (< 0: no wrap)
                }
                run0r[2 * r] = start[(row + a0) * nz];
                prer[2 * r + 1] = prer[2 * r] + (start[(row + a1 + 1) * nz] - run0r[2 * r]);
                run0r[2 * r + 1] = start[row * nz];
                prer[2 * r + 2] = prer[2 * r + 1] + (b1c >= 0 ? start[(row + b1c + 1) * nz] - run0r[2 * r + 1] : 0);
            }
        } else {
        __syncthreads();                                   // (the previous segment's tiles are done with run0 / pre)
        if (static_cast<int>(threadIdx.x) < nruns) {       // run q: row q >> 1, part A (up to the end of the row) or B (the wrapped rest)
            const int q = threadIdx.x;
            int first = 0, len = sorted_n;
            if (!sweep_all) {
                const int r = q >> 1;
                const int gy = (2 * R + 1 < ny) ? ((cy - R + r) % ny + ny) % ny : r;        // (every row once when the rows wrap)
                const int row = gy * nx;
                int a0, a1, b1c;                           // run A: cells a0 .. a1; run B: cells 0 .. b1c (< 0: none)
                if (w >= nx) { a0 = 0; a1 = nx - 1; b1c = -1; }
                else {
                    const int a = ((cxa - R) % nx + nx) % nx;
                    a0 = a; a1 = min(a + w - 1, nx - 1);
                    b1c = a + w - 1 - nx;
                }
                if ((q & 1) == 0) { first = start[(row + a0) * nz]; len = start[(row + a1 + 1) * nz] - first; }
                else { first = start[row * nz]; len = b1c >= 0 ? start[(row + b1c + 1) * nz] - first : 0; }
            }
            run0[q] = first;
            pre[q + 1] = len;                              // (lengths for now, prefix sums below)
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            pre[0] = 0;
            for (int q = 0; q < nruns; ++q) pre[q + 1] += pre[q];
        }
        __syncthreads();
        }
        const int total = fast ? prer[6] : pre[nruns];
        const fp2 mx = splat(me.x), my = splat(me.y), mz = splat(mez);
        int pending = 0;                                   // pairs in this wave's queue (wave-uniform)
        // one queued pair of the tile in LDS (queue entry: 6 bits lane, 10 bits candidate slot of the tile)
        auto evaluate = [&](unsigned e, bool valid) {
            const int dl = static_cast<int>(e >> 10), ci = static_cast<int>(e & 1023u);
            if (MODE == 1) {                               // build: the batch goes to the list as it is evaluated
                if (rec_ok && lb < Ls.cap)
                    my_list[static_cast<size_t>(lb) * 64 + lane] = valid ? ((static_cast<uint32_t>(dl) << 26) | static_cast<uint32_t>(tsrc[ci])) : 0xffffffffu;
                else rec_ok = false;
                ++lb;
            }
            pair_model(dl, valid, tx[ci], ty[ci], tz[ci]);
        };
        // (a tile holds kDwTile - 1 candidates: the entry "lane 63, slot 1023" never occurs and 0xffff can mark an empty lane)
        for (int v0 = 0; v0 < total;) {
            const int cnt = min(kDwTile - 1, total - v0);
            const int chunks = (cnt + kDwChunk - 1) / kDwChunk;
            auto source = [&](int v) {                     // position in the concatenated list -> run q, element src
                int src;
                if (fast) {
                    src = run0r[0] + v;
#pragma unroll
                    for (int q = 1; q < 6; ++q) src = (v >= prer[q]) ? run0r[q] + (v - prer[q]) : src;
                } else {
                    src = run0[0] + v;
                    for (int q = 1; q < nruns; ++q) src = (v >= pre[q]) ? run0[q] + (v - pre[q]) : src;
                }
                return src;
            };
            __syncthreads();
            for (int j = threadIdx.x; j < chunks * kDwChunk; j += kBlock) {
                float4 o = make_float4(0.0f, 0.0f, -3.0e38f, 0.0f);        // (padding of the last chunk: below everything)
                if (j < cnt) {
                    // (build) where a replay launch will find this candidate's current position: its row of pos4 in a shared
                    // world, its sorted slot in a single-rank one
                    const int sl = source(v0 + j);
                    const int src = Wd.pos4 ? order[sl] : sl;
                    o = Wd.pos4 ? Wd.pos4[src] : sorted[src];
                    if (MODE == 1) tsrc[j] = src;
                }
                tx[j] = o.x; ty[j] = o.y; tz[j] = o.z;
            }
            __syncthreads();
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSF)
            if (MODE == 2 && tf2 == 0) tf2 = wall_clock64();
#endif
            v0 += cnt;
            for (int ch = wave; ch < chunks; ch += kBlock / 64) {       // this wave's share of the candidates
                const int j0 = ch * kDwChunk;
                uint32_t mask = 0;                                     // candidate j0 + j of the chunk -> bit 31 - j
#pragma unroll 8
                for (int j = 0; j < kDwChunk; j += 2) {
                    const fp2 ox = *reinterpret_cast<const fp2*>(&tx[j0 + j]), oy = *reinterpret_cast<const fp2*>(&ty[j0 + j]),
                              oz = *reinterpret_cast<const fp2*>(&tz[j0 + j]);
                    const fp2 ddx = ox - mx, ddy = oy - my, below = mz - oz;       // below < 0: the candidate is above
                    const fp2 dd2 = fma2(ddy, ddy, ddx * ddx);
                    const fp2 bs = fma2(b1, below, b0);                            // sqrt(80.02) beta(dz)
                    if (MODE == 1) {
                        // with the margin: the candidate may come up to m2 closer in height and in the plane, |beta| may grow by
                        // DW2 m2:  dxy < min(10, sqrt(80.02) (|beta| + DW2 m2)) + m2  and  dz > -m2
                        const float tx_ = fminf(fabsf(bs.x) + c2, 10.0005f) + m2, ty_ = fminf(fabsf(bs.y) + c2, 10.0005f) + m2;
                        const fp2 q = dd2 - fp2{tx_ * tx_, ty_ * ty_};
                        const fp2 bm = below - splat(m2);
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.x) & __float_as_uint(bm.x), 31);
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.y) & __float_as_uint(bm.y), 31);
                    } else {
                        const fp2 lim = bs * bs;
                        const fp2 q = dd2 - fp2{fminf(lim.x, 100.01f), fminf(lim.y, 100.01f)};
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.x) & __float_as_uint(below.x), 31);
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.y) & __float_as_uint(below.y), 31);
                    }
                }
                const int mine = __builtin_popcount(mask);
                const int upto = wave_inclusive_scan(mine);
                int pos = pending + upto - mine;
                while (mask != 0u) {                                   // this lane's pairs -> its slice of the queue
                    const int j = __builtin_clz(mask);
                    mask &= ~(0x80000000u >> j);
                    my_queue[pos++] = static_cast<unsigned short>((lane << 10) | (j0 + j));
                }
                pending += __builtin_amdgcn_readlane(upto, 63);
                __builtin_amdgcn_wave_barrier();
                // Full batches once kDwBatches of them wait: a lane's pairs sit next to each other in the queue, so a batch of 64
                // CONSECUTIVE pairs would hit each drone's sum ~5 times in one ds_add_u64 (the LDS serialises those); batch b takes
                // every nb-th pair instead, taken from the end (the < 64 oldest stay where they are).
                if (pending >= 64 * kDwBatches) {
                    const int nb = pending >> 6;
                    pending &= 63;
                    for (int b = 0; b < nb; ++b) evaluate(my_queue[pending + b + lane * nb], true);
                }
                __builtin_amdgcn_wave_barrier();
            }
            {                                                          // (the tile is about to be replaced: everything goes)
                const int nb = (pending + 63) >> 6;
                for (int b = 0; b < nb; ++b) { const int k = b + lane * nb; evaluate(k < pending ? my_queue[k] : 0u, k < pending); }
            }
            pending = 0;
        }
        cs = ce + 1;
    }
    if (MODE == 1) {                                       // the group's list counts only if all four waves completed theirs
        if (lane == 0) my_nb[0] = static_cast<unsigned short>(rec_ok ? lb : 0);      // the wave's batches (< cap <= 65535)
        int* const okf = reinterpret_cast<int*>(pre);
        __syncthreads();
        if (lane == 0) okf[wave] = rec_ok ? 1 : 0;
        __syncthreads();
        if (threadIdx.x == 0) Ls.ok[blockIdx.x] = okf[0] & okf[1] & okf[2] & okf[3];
    }
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSF)
    if (MODE == 2) tf3 = wall_clock64();
#endif
    __syncthreads();
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSF)
    if (MODE == 2 && threadIdx.x == 0 && blockIdx.x < 4096) {
        const unsigned int slot = gpd_ts_cnt[blockIdx.x]++ & 7u;
        unsigned long long* o = gpd_ts + (static_cast<size_t>(slot) * 4096 + blockIdx.x) * 4;
        o[0] = tf0; o[1] = tf1; o[2] = tf2; o[3] = tf3;
    }
#endif
    if (wave == 0 && own) {                                // add up the four waves' shares
        unsigned long long sum = 0;
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) sum += sums[k][lane];
        dw_out[my_row - Wd.own_lo] = -static_cast<float>(static_cast<double>(static_cast<long long>(sum)) * (1.0 / 1073741824.0));
    }
}

// ------------------------------------------------------------------------------------------------
// standalone batched DSLPIDControl.computeControl
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gpd_pid_kernel(
    const GpdParams P, float* __restrict__ pid, int64_t ld, float dt, const float* __restrict__ cur_pos,
    const float* __restrict__ cur_quat, const float* __restrict__ cur_vel, const float* __restrict__ target_pos,
    const float* __restrict__ target_rpy, const float* __restrict__ target_vel,
    const float* __restrict__ target_rpy_rates, float* __restrict__ rpm_out, float* __restrict__ pos_e_out,
    float* __restrict__ yaw_e_out, int n_total) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= n_total) return;
    Kin k{};
    k.px = cur_pos[n * 3]; k.py = cur_pos[n * 3 + 1]; k.pz = cur_pos[n * 3 + 2];
    const float4 q = reinterpret_cast<const float4*>(cur_quat)[n];
    k.qx = q.x; k.qy = q.y; k.qz = q.z; k.qw = q.w;
    k.vx = cur_vel[n * 3]; k.vy = cur_vel[n * 3 + 1]; k.vz = cur_vel[n * 3 + 2];
    Pid s;
    s.ipx = pid[0 * ld + n]; s.ipy = pid[1 * ld + n]; s.ipz = pid[2 * ld + n];
    s.lr = pid[3 * ld + n]; s.lp = pid[4 * ld + n]; s.ly = pid[5 * ld + n];
    s.irx = pid[6 * ld + n]; s.iry = pid[7 * ld + n]; s.irz = pid[8 * ld + n];
    float roll, pitch, yaw;
    quat_to_rpy(k.qx, k.qy, k.qz, k.qw, roll, pitch, yaw);
    const Mat3 R = quat_to_mat(k.qx, k.qy, k.qz, k.qw);
    const float tyaw = target_rpy ? target_rpy[n * 3 + 2] : 0.0f;
    float tv[3] = {0, 0, 0}, tr[3] = {0, 0, 0};
    if (target_vel) { tv[0] = target_vel[n * 3]; tv[1] = target_vel[n * 3 + 1]; tv[2] = target_vel[n * 3 + 2]; }
    if (target_rpy_rates) {
        tr[0] = target_rpy_rates[n * 3]; tr[1] = target_rpy_rates[n * 3 + 1]; tr[2] = target_rpy_rates[n * 3 + 2];
    }
    float rpm[4], pe[3], ye;
    dslpid(P, dt, 1.0f / dt, k, roll, pitch, yaw, R, target_pos[n * 3], target_pos[n * 3 + 1], target_pos[n * 3 + 2], tyaw,
           tv[0], tv[1], tv[2], tr[0], tr[1], tr[2], s, rpm, pe, &ye);
    pid[0 * ld + n] = s.ipx; pid[1 * ld + n] = s.ipy; pid[2 * ld + n] = s.ipz;
    pid[3 * ld + n] = s.lr; pid[4 * ld + n] = s.lp; pid[5 * ld + n] = s.ly;
    pid[6 * ld + n] = s.irx; pid[7 * ld + n] = s.iry; pid[8 * ld + n] = s.irz;
    reinterpret_cast<float4*>(rpm_out)[n] = make_float4(rpm[0], rpm[1], rpm[2], rpm[3]);
    if (pos_e_out) { pos_e_out[n * 3] = pe[0]; pos_e_out[n * 3 + 1] = pe[1]; pos_e_out[n * 3 + 2] = pe[2]; }
    if (yaw_e_out) yaw_e_out[n] = ye;
}

// ------------------------------------------------------------------------------------------------
// 20-float state vectors (envs/BaseAviary.py:559-561)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gpd_state20_kernel(const GpdState S, const float* __restrict__ obs12,
                                                             float* __restrict__ out, int n_total) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= n_total) return;
    state20_row(S, obs12, out, n);
}

// ------------------------------------------------------------------------------------------------
// GpdSwarm: the physics sub-step of ONE world's drones (single-drone lanes of env_step, all add-on terms possible, the downwash
// force from state.dw_force), which also hands the next launch what it needs: the drone's new position in the packed array the
// force kernel (and, across ranks, the all-gather) reads, how far it has moved since the last binning, its state vector.
// ------------------------------------------------------------------------------------------------
struct SwarmOut {
    float4* pos4_own;          // pos4 + rank * slab: row i = own drone i
    const float4* bin_pos_own; // bin_pos + rank * slab
    float* meta_own;           // the rank's first meta row as floats (workgroup b's row: 4 b floats on)
    const int* slot_of_own;    // slot_of + rank * slab, or NULL
    float4* pos_sorted;        // [n_rows] current positions by sorted slot (with slot_of)
    const float* drift;        // [2] the swarm's common lateral drift since the binning, as of the previous sub-step
    float* vec_out;            // [n][20] or NULL
};
// The drone's new position for the next launch (by row; by sorted slot when this rank holds the whole world), and how far it is
// from where it was binned -- RELATIVE TO THE SWARM'S COMMON DRIFT: what the stale cell order and the wake lists tolerate is a
// change of the drones' positions relative to each other; a translation all drones share (a swarm in transit) changes no pair.
// `drift` is the mean lateral displacement of all drones as of the previous sub-step (the force launch in between computes it
// from the sums below and every workgroup of every rank reads the same two floats: ANY common vector keeps the bound exact, a
// good one keeps it small).  The workgroup's largest residual and its displacement sums go to the workgroup's own meta row --
// plain stores (1024 wavefronts updating ONE word with atomics cost 13 us: atomics on one address are served one after the
// other; the force kernel, which needs the maximum over all of them, reads a few hundred words instead).
struct SwarmIn { float4 b; float cx, cy; int slot; };     // what the tail needs from memory, requested with the state's loads
__device__ __forceinline__ SwarmIn swarm_head(const SwarmOut& O, bool active, uint32_t n) {
    SwarmIn I;
    I.cx = O.drift[0]; I.cy = O.drift[1];
    I.b = active ? O.bin_pos_own[n] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    I.slot = (active && O.slot_of_own) ? O.slot_of_own[n] : -1;
    return I;
}
__device__ __forceinline__ void swarm_tail(const SwarmOut& O, const SwarmIn& I, bool active, uint32_t n, float px, float py, float pz) {
    __shared__ float wg_red[kBlock / 64][3];
    const float cx = I.cx, cy = I.cy;
    float d2 = 0.0f, sx = 0.0f, sy = 0.0f;
    if (active) {
        O.pos4_own[n] = make_float4(px, py, pz, 0.0f);
        if (I.slot >= 0) O.pos_sorted[I.slot] = make_float4(px, py, pz, 0.0f);
        const float4 b = I.b;
        const float dx = px - b.x, dy = py - b.y, dz = pz - b.z;
        const float ex = dx - cx, ey = dy - cy;
        d2 = fmaf(dz, dz, fmaf(ey, ey, ex * ex));
        const bool fin = d2 == d2 && d2 < 3.0e38f;             // (a drone without a finite position takes no part in the sort)
        // (NaN AND +-inf: a non-finite position fails every pair test, it needs no search radius -- an infinite d2 would be the
        // workgroup's maximum, push R beyond kDwMaxR and turn every group of every rank into an O(N^2) sweep until the next binning)
        d2 = fin ? d2 : 0.0f;
        sx = fin ? dx : 0.0f; sy = fin ? dy : 0.0f;
    }
    d2 = wave_max(d2); sx = wave_sum(sx); sy = wave_sum(sy);
    if ((threadIdx.x & 63) == 0) { wg_red[threadIdx.x >> 6][0] = d2; wg_red[threadIdx.x >> 6][1] = sx; wg_red[threadIdx.x >> 6][2] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = wg_red[0][0], tx = wg_red[0][1], ty = wg_red[0][2];
#pragma unroll
        for (int k = 1; k < kBlock / 64; ++k) { m = fmaxf(m, wg_red[k][0]); tx += wg_red[k][1]; ty += wg_red[k][2]; }
        float* row = O.meta_own + 4 * blockIdx.x;           // (x stays non-finite: "no drone in this row")
        row[1] = tx; row[2] = ty; row[3] = m;
    }
}
template <int ACT>
__global__ __launch_bounds__(kBlock) void gpd_swarm_step_kernel(const GpdParams P, const GpdState S, const GpdStepCfg C,
                                                                const float* __restrict__ action, float* __restrict__ obs12,
                                                                const SwarmOut O) {
    const uint32_t N = static_cast<uint32_t>(C.num_envs);
    const uint32_t n_raw = blockIdx.x * kBlock + threadIdx.x;
    Lane L;
    L.tid = threadIdx.x;
    L.active = n_raw < N;
    L.n = L.active ? n_raw : 0u;
    L.le = L.tid; L.d = 0; L.env = L.n; L.shfl = false;
    const uint32_t flags = C.physics_flags;
    Carry c;
    float tgx, tgy, tgz, ip[7];
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSS)
    // (experiment build, scratch/exp_r04/step_timeline.py: where a launch of this kernel spends its time)
    const unsigned long long tss0 = wall_clock64();
    unsigned long long tss1 = 0, tss2 = 0;
#endif
    const float4 act = load_action<4>(action, L.n);
    const SwarmIn I = swarm_head(O, L.active, L.n);        // (one trip to memory with the state's loads instead of one behind the step)
    load_carry<false, true>(S, C, flags, L, S.kin, S.kin, c, tgx, tgy, tgz, ip);     // (no task, no reset: readable dummies)
    c.roll = c.pitch = c.yaw = 0.0f;
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSS)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tss1 = wall_clock64();
#endif
    StepOut out;
    env_step<false, true, false, 4, ACT, true>(P, C, flags, 1, L, act, tgx, tgy, tgz, false, S.kin, ip[0], ip[1], ip[2], ip[3], ip[4],
                                               ip[5], ip[6], nullptr, nullptr, c, out);
    swarm_tail(O, I, L.active, L.n, c.k.px, c.k.py, c.k.pz);
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSS)
    asm volatile("" :: "v"(c.k.px), "v"(c.k.qw), "v"(c.k.wz), "v"(out.o[11]) : "memory"); tss2 = wall_clock64();
#endif
    // observation rows (48 B) and state vectors (80 B, BaseAviary._getDroneStateVector, envs/BaseAviary.py:541-561): a lane's row
    // is a strided piece of cache lines, a wave's 64 rows are one contiguous block -- transposed through LDS and stored as
    // 1 KiB bursts (the wave's own LDS instructions execute in order: no barrier between its writes and its reads)
    __shared__ __attribute__((aligned(16))) float4 sh_rows[kBlock * 5];
    const int wave0 = threadIdx.x & ~63, lane = threadIdx.x & 63;
    const uint32_t n0 = n_raw - static_cast<uint32_t>(lane);               // first drone of this wave
    const uint32_t rows = __builtin_amdgcn_readfirstlane(n0 < N ? (N - n0 < 64u ? N - n0 : 64u) : 0u);
    const Kin& k = c.k;
    auto burst = [&](float* dst_rows, auto f4c) {          // F4 float4 per row; the wave's rows start at dst_rows
        constexpr int F4 = decltype(f4c)::value;
        float4* patch = sh_rows + wave0 * 5;
        __builtin_amdgcn_wave_barrier();
        if (rows == 64u) {
            // every row of the wave exists (all waves but the grid's last): F4 LDS reads in one run, then F4 stores -- with a
            // test around every store each read is waited for before its store and the next read starts behind it, F4 LDS
            // round trips in a row (profiles/r04_swarm_step_timeline.txt: the stores are issued 0.14 us sooner this way; the
            // memory pipeline paces the rest, 0.08 us of a 3.8 us workgroup is what it is worth)
            float4 v[F4];
#pragma unroll
            for (int j = 0; j < F4; ++j) v[j] = patch[j * 64 + lane];
#pragma unroll
            for (int j = 0; j < F4; ++j) {
                f4v wv = {v[j].x, v[j].y, v[j].z, v[j].w};
                __builtin_nontemporal_store(wv, reinterpret_cast<f4v*>(dst_rows) + (j * 64 + lane));
            }
        } else {
#pragma unroll
            for (int j = 0; j < F4; ++j) {
                const int idx = j * 64 + lane;
                const float4 v = patch[idx];
                if (static_cast<uint32_t>(idx) < rows * F4) {
                    f4v wv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(wv, reinterpret_cast<f4v*>(dst_rows) + idx);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    {
        float4* mine = sh_rows + wave0 * 5 + lane * 3;
        mine[0] = make_float4(out.o[0], out.o[1], out.o[2], out.o[3]);
        mine[1] = make_float4(out.o[4], out.o[5], out.o[6], out.o[7]);
        mine[2] = make_float4(out.o[8], out.o[9], out.o[10], out.o[11]);
        burst(obs12 + static_cast<size_t>(n0) * 12, std::integral_constant<int, 3>{});
    }
    if (O.vec_out) {
        float4* mine = sh_rows + wave0 * 5 + lane * 5;
        mine[0] = make_float4(k.px, k.py, k.pz, k.qx);
        mine[1] = make_float4(k.qy, k.qz, k.qw, out.o[3]);
        mine[2] = make_float4(out.o[4], out.o[5], k.vx, k.vy);
        mine[3] = make_float4(k.vz, out.o[9], out.o[10], out.o[11]);
        mine[4] = make_float4(c.l0, c.l1, c.l2, c.l3);
        burst(O.vec_out + static_cast<size_t>(n0) * 20, std::integral_constant<int, 5>{});
    }
#if defined(GPD_EXP_TS) && defined(GPD_EXP_TSS)
    if (L.active) store_carry<false>(S, L, c);
    const unsigned long long tss3 = wall_clock64();            // every store issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tss4 = wall_clock64();            // ... and acknowledged
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        const unsigned int slot = gpd_ts_cnt[blockIdx.x]++ & 7u;
        unsigned long long* o = gpd_ts + (static_cast<size_t>(slot) * 4096 + blockIdx.x) * 4;
        o[0] = tss0; o[1] = tss1; o[2] = tss2; o[3] = (tss3 & 0xffffffffull) | (tss4 << 32);
    }
    return;
#endif
    if (!L.active) return;
    store_carry<false>(S, L, c);
}

// The rank's largest squared displacement (the maximum of its workgroups' maxima, one per meta row) -> the w of its FIRST meta
// row.  Launched behind gpd_swarm_step_kernel when the world has too many meta rows for every force workgroup to read them all
// (world_size x meta_rows > 1024); the kernel boundary is the synchronisation (an in-kernel "last workgroup reduces" would need an
// agent-scope release per workgroup: microseconds each on this part, profiles/r04_doorbell_step_server.txt).
__global__ __launch_bounds__(kBlock) void dwg_reduce_meta_kernel(float* __restrict__ meta_own, int rows) {
    __shared__ float red[kBlock / 64];
    float m = 0.0f;
    for (int k = threadIdx.x; k < rows; k += kBlock) m = fmaxf(m, meta_own[4 * k + 3]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < kBlock / 64; ++k) m = fmaxf(m, red[k]);
        meta_own[3] = m;
    }
}

// after a reset / an outside change of the state: the rank's slab of pos4 from state.kin -- its drones, the rows without one
// (non-finite), the meta row (dmax^2 = 0) -- and, on request, the state vectors
__global__ __launch_bounds__(kBlock) void gpd_swarm_pack_kernel(const GpdState S, int n, int slab, float4* __restrict__ pos4_own,
                                                                const float* __restrict__ obs12, float* __restrict__ vec_out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= slab) return;
    const float nan = __int_as_float(0x7fc00000);
    if (i < n) {
        pos4_own[i] = make_float4(S.kin[i], S.kin[S.ld + i], S.kin[2 * S.ld + i], 0.0f);
        if (vec_out) state20_row(S, obs12, vec_out, i);
    } else {
        pos4_own[i] = make_float4(nan, 0.0f, 0.0f, 0.0f);     // (a non-finite x is what says "no drone"; meta rows: sums and maximum 0)
    }
}

// ------------------------------------------------------------------------------------------------
// Shader-clock probe (diagnostics for bench.py's issue roofline): 256 workgroups x 4 waves = one wave per SIMD, like the
// headline launch, each running a dependent v_fma_f32 chain; lane 0 of workgroup 0 reports the shader-clock cycles
// (s_memtime) and the constant-rate wall-clock ticks (s_memrealtime) the chain took.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gpd_clock_probe_kernel(unsigned long long* __restrict__ out, int iters) {
    float r = threadIdx.x * 1e-3f;
    const float b = 1.0001f, c = 1e-4f;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#define GPD_FMA4 "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
#define GPD_FMA16 GPD_FMA4 GPD_FMA4 GPD_FMA4 GPD_FMA4
        asm volatile(GPD_FMA16 GPD_FMA16 GPD_FMA16 GPD_FMA16 : "+v"(r) : "v"(b), "v"(c));
#undef GPD_FMA16
#undef GPD_FMA4
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
    if (r == 12345.678f) out[2] = 1;                       // (keeps the chain alive)
}

// ------------------------------------------------------------------------------------------------
// RCCL, resolved at run time: libgpd.so has no link-time dependency on it (a single-GPU consumer never loads it), and
// inside a PyTorch process dlopen() by SONAME returns the copy torch already mapped instead of a second one.
// ------------------------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    // (optional: only gpd_p2p_group needs them -- a library without them still serves the all-gather)
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl R = [] {
        Rccl r;
        const char* env = getenv("GPD_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            r.why = dlerror();
        }
        if (!r.handle) return r;
        auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p) r.why = std::string("missing symbol ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.handle, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.handle, "ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.handle, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.handle, "ncclRecv"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.AllGather || !r.GetErrorString) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
        return r;
    }();
    return R;
}

int rccl_fail(ncclResult_t e, const char* where) {
    g_last_error = std::string(where) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
    return 1000 + static_cast<int>(e);          // (positive, outside hipError_t's range)
}

int need_rccl(const char* who) {
    if (rccl().handle) return 0;
    return fail(GPD_ENOTSUP, (std::string(who) + ": RCCL is not available (" + rccl().why + ")").c_str());
}

template <bool PID, bool EXT, int AW, int ACT>
hipError_t launch_step(bool multi, hipStream_t st, const GpdParams& P, const GpdState& S, const GpdStepCfg& C,
                       const Span& T, const float* action, const float* target_pos, const float* init_pose,
                       float* obs12, float* reward, uint8_t* terminated, uint8_t* truncated, float* term_obs12) {
    const int64_t N = static_cast<int64_t>(C.num_envs) * C.drones_per_env;
    if (T.num_steps == 1) {      // gpd_step, or a rollout of one step: the low-latency single-step kernel
        const int lanes = multi ? (kBlock / C.drones_per_env) * C.drones_per_env : (kBlock / 64) * C.lanes_per_wave;
        const dim3 grid(static_cast<unsigned>((N + lanes - 1) / lanes));
#define GPD_STEP_HOT S.kin, action, S.step_counter, target_pos, static_cast<const int32_t*>(S.act_ring ? S.ring_pos : S.step_counter), \
                     static_cast<uint32_t>(S.ld), C.num_envs, C.lanes_per_wave, C.target_per_env
        if (multi) {
            hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, true, AW, ACT, false>), grid, dim3(kBlock), 0, st, GPD_STEP_HOT, P, S, C,
                               init_pose, obs12, reward, terminated, truncated, term_obs12);
        } else if (C.substeps == 1) {
            hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, false, AW, ACT, true>), grid, dim3(kBlock), 0, st, GPD_STEP_HOT, P, S, C,
                               init_pose, obs12, reward, terminated, truncated, term_obs12);
        } else {
            hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, false, AW, ACT, false>), grid, dim3(kBlock), 0, st, GPD_STEP_HOT, P, S, C,
                               init_pose, obs12, reward, terminated, truncated, term_obs12);
        }
#undef GPD_STEP_HOT
    } else {
        const int lanes = multi ? (kBlock / C.drones_per_env) * C.drones_per_env : kBlock;
        const dim3 grid(static_cast<unsigned>((N + lanes - 1) / lanes));
        Span Tr = T;
        const bool shfl = multi && C.drones_per_env <= 64 && (C.drones_per_env & (C.drones_per_env - 1)) == 0;
        Tr.ring = ((!multi || shfl) && grid.x <= 2u * 256u) ? 4 : 2;   // <= 2 workgroups per CU: LDS is not what limits occupancy
        const size_t lds = static_cast<size_t>(Tr.ring) * kSlotBytes;
        static const bool store_wave_variant = getenv("GPD_ROLLOUT_STOREWAVE") != nullptr;   // A/B switch, diagnostics only
        if (S.act_ring && !store_wave_variant && term_obs12 == nullptr && (shfl || !multi)) {   // gpd_rollout_history (it checked the shape)
            if (shfl)
                hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, false, true, true, true>), grid, dim3(kBlock), 0, st, P, S, C, Tr,
                                   action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
            else if (C.substeps == 1)
                hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, true, false, true, true>), grid, dim3(kBlock), 0, st, P, S, C, Tr,
                                   action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
            else
                hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, false, false, true, true>), grid, dim3(kBlock), 0, st, P, S, C, Tr,
                                   action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
        } else
        if (shfl && !store_wave_variant && term_obs12 == nullptr) {   // aviaries of 2..64 (power of two) drones: no helper wave either
            hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, false, true>), grid, dim3(kBlock), 0, st, P, S, C, Tr, action,
                               target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
        } else if (multi) {
            hipLaunchKernelGGL((gpd_rollout_kernel<PID, EXT, true, AW>), grid, dim3(kRollThreads), lds, st, P, S, C, Tr,
                               action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
        } else if (!store_wave_variant && term_obs12 == nullptr) {
            // The headline shape -- plain DYN, RPM actions, one sub-step, a batch that leaves one wave per SIMD -- stores its
            // observation bursts as ordinary stores: measured 3-4 % faster there (0.837 -> 0.803 us per step), while every other
            // shape (larger batches, sub-step loops, multi-drone aviaries) is 1-3 % faster with non-temporal ones
            // (A/B on one box, round 2: scratch/ab.sh, scratch/ab2.sh) -- and only for long rollouts: the ordinary stores leave
            // their lines to the end-of-kernel write-back, which a 20-step launch does not amortise (1.14 vs 1.00 us per step).
            if (C.substeps == 1 && !PID && !EXT && ACT == GPD_ACT_RPM && N <= (1 << 17) && T.num_steps >= 48)
                hipLaunchKernelGGL((gpd_rollout1_kernel<false, false, 4, GPD_ACT_RPM, true, false, false>), grid, dim3(kBlock), 0, st, P, S, C, Tr,
                                   action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
            else if (C.substeps == 1)
                hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, true, false>), grid, dim3(kBlock), 0, st, P, S, C, Tr, action,
                                   target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
            else
                hipLaunchKernelGGL((gpd_rollout1_kernel<PID, EXT, AW, ACT, false, false>), grid, dim3(kBlock), 0, st, P, S, C, Tr, action,
                                   target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
        } else {
            hipLaunchKernelGGL((gpd_rollout_kernel<PID, EXT, false, AW>), grid, dim3(kRollThreads), lds, st, P, S, C, Tr,
                               action, target_pos, init_pose, obs12, reward, terminated, truncated, term_obs12);
        }
    }
    return hipGetLastError();
}

// argument checks + launch shared by gpd_step (K = 1) and gpd_rollout
int step_impl(const char* who, const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const Span& T,
              const float* action, const float* target_pos, const float* init_pose, float* obs12, float* reward,
              uint8_t* terminated, uint8_t* truncated, float* term_obs12, void* stream) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string(who) + ": " + msg).c_str()); };
    if (!params || !state || !cfg) return bad(GPD_EINVAL, "NULL params/state/cfg");
    if (!state->kin || !state->step_counter) return bad(GPD_EINVAL, "NULL state.kin/step_counter");
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
                                       terminated, truncated, term_obs12)
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

#endif  // !GPD_POLICY_TU

}  // namespace

// ==================================================================================================
// C ABI
// ==================================================================================================
// (GPD_PID_POLICY_IN_POLICY_TU: experiment / regression switch -- instantiate the DSLPID policy kernels in the policy unit, under
// ITS scheduler, instead of the main unit; tests/test_kernel_isa.py and scratch/exp_r04/vel_probe.py)
#if defined(GPD_POLICY_TU) == defined(GPD_PID_POLICY_IN_POLICY_TU)
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

extern "C" {

#ifndef GPD_POLICY_TU

int gpd_abi_version(void) { return GPD_ABI_VERSION; }

const char* gpd_last_error(void) { return g_last_error.c_str(); }

void gpd_struct_sizes(int32_t out[3]) {
    out[0] = static_cast<int32_t>(sizeof(GpdParams));
    out[1] = static_cast<int32_t>(sizeof(GpdState));
    out[2] = static_cast<int32_t>(sizeof(GpdStepCfg));
}

int gpd_sizeof_swarm(void) { return static_cast<int>(sizeof(GpdSwarm)); }

int gpd_step(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const float* action,
             const float* target_pos, const float* init_pose, float* obs12, float* reward, uint8_t* terminated,
             uint8_t* truncated, float* term_obs12, void* stream) {
    const Span T{1, 0, 0, 0, 2};
    return step_impl("gpd_step", params, state, cfg, T, action, target_pos, init_pose, obs12, reward, terminated,
                     truncated, term_obs12, stream);
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
    if (cfg && cfg->drones_per_env > 1 && (cfg->drones_per_env > 64 || (cfg->drones_per_env & (cfg->drones_per_env - 1)) != 0))
        return fail(GPD_ENOTSUP, "gpd_rollout_history: aviaries of 1, 2, 4 .. 64 drones (use gpd_rollout + gpd_full_obs otherwise)");
    if (getenv("GPD_ROLLOUT_STOREWAVE")) return fail(GPD_ENOTSUP, "gpd_rollout_history: not with the GPD_ROLLOUT_STOREWAVE diagnostic");
    const Span T{num_steps, action_step_stride, obs_step_stride, env_step_stride, 2};
    return step_impl("gpd_rollout_history", params, state, cfg, T, actions, target_pos, init_pose, obs12, reward, terminated,
                     truncated, nullptr, stream);
}

#endif  // !GPD_POLICY_TU
#ifdef GPD_POLICY_TU
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

#endif  // GPD_POLICY_TU
#ifndef GPD_POLICY_TU
static int hist_args(const char* who, const GpdState* st, int32_t n_drones, int32_t D, int32_t A) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string(who) + ": " + msg).c_str()); };
    if (!st || !st->act_ring || !st->ring_pos || st->hist_len <= 0) return bad(GPD_EINVAL, "state has no action ring (act_ring / ring_pos / hist_len)");
    if (n_drones <= 0 || D <= 0 || n_drones % D != 0 || A <= 0 || A > 4)
        return bad(GPD_EINVAL, "n_drones must be a positive multiple of drones_per_env and act_dim in 1..4");
    const int64_t W = 12 + static_cast<int64_t>(st->hist_len) * A;
    if (static_cast<int64_t>(n_drones) * W >= (1LL << 32)) return bad(GPD_ERANGE, "n_drones*(12+hist_len*act_dim) must be < 2^32");
    if ((W | 1) * 4 > 48 * 1024) return bad(GPD_ERANGE, "a row of 12+hist_len*act_dim floats must fit 48 KiB");
    return 0;
}

// drones per workgroup of gpd_hist_rows_kernel: the largest power of two <= 64 whose whole rows fit 48 KiB of LDS
static int hist_rows_per_wg(int64_t W) {
    int R = 64;
    while (R > 1 && static_cast<int64_t>(R) * (W | 1) * 4 > 48 * 1024) R >>= 1;
    return R;
}

int gpd_hist_rows(const GpdState* state, int32_t n_drones, int32_t drones_per_env, int32_t act_dim, const float* obs12,
                  float* obs_full, void* stream) {
    if (int rc = hist_args("gpd_hist_rows", state, n_drones, drones_per_env, act_dim)) return rc;
    if (!obs12 || !obs_full) return fail(GPD_EINVAL, "gpd_hist_rows: NULL obs12/obs_full");
    const int64_t W = 12 + static_cast<int64_t>(state->hist_len) * act_dim;
    const int R = hist_rows_per_wg(W);
    const dim3 grid(static_cast<unsigned>((n_drones + R - 1) / R));
    hipLaunchKernelGGL(gpd_hist_rows_kernel, grid, dim3(kBlock), static_cast<size_t>(R) * (W | 1) * 4,
                       static_cast<hipStream_t>(stream), static_cast<uint32_t>(n_drones), drones_per_env, act_dim, state->hist_len, R,
                       state->act_ring, state->ring_pos, obs12, static_cast<int64_t>(0), static_cast<const float*>(nullptr),
                       static_cast<int64_t>(0), obs_full, static_cast<int64_t>(0));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_hist_rows launch");
    return 0;
}

int gpd_full_obs(const GpdState* state, int32_t num_steps, int32_t n_drones, int32_t drones_per_env, int32_t act_dim,
                 const float* obs12, int64_t obs_step_stride, const float* actions, int64_t action_step_stride,
                 float* obs_full, int64_t full_step_stride, void* stream) {
    if (int rc = hist_args("gpd_full_obs", state, n_drones, drones_per_env, act_dim)) return rc;
    if (!actions) return fail(GPD_EINVAL, "gpd_full_obs: NULL actions");
    if (obs_full && !obs12) return fail(GPD_EINVAL, "gpd_full_obs: obs_full needs obs12");
    if (num_steps <= 0 || num_steps > 65535) return fail(GPD_EINVAL, "gpd_full_obs: num_steps must be in 1..65535");
    if (obs_step_stride < 0 || action_step_stride < 0 || full_step_stride < 0)
        return fail(GPD_EINVAL, "gpd_full_obs: strides must be non-negative");
    const int H = state->hist_len;
    const int64_t W = 12 + static_cast<int64_t>(H) * act_dim;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (obs_full) {
        const int R = hist_rows_per_wg(W);
        const dim3 grid(static_cast<unsigned>((n_drones + R - 1) / R), static_cast<unsigned>(num_steps));
        hipLaunchKernelGGL(gpd_hist_rows_kernel, grid, dim3(kBlock), static_cast<size_t>(R) * (W | 1) * 4, st,
                           static_cast<uint32_t>(n_drones), drones_per_env, act_dim, H, R, state->act_ring, state->ring_pos, obs12,
                           obs_step_stride, actions, action_step_stride, obs_full, full_step_stride);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "gpd_full_obs launch");
    }
    // the ring is read by the kernel above and updated by the next two: same stream, in order
    const int keep = num_steps < H ? num_steps : H;
    const dim3 grid2(static_cast<unsigned>((n_drones + kBlock - 1) / kBlock), static_cast<unsigned>(keep));
    hipLaunchKernelGGL(gpd_hist_push_kernel, grid2, dim3(kBlock), 0, st, num_steps, static_cast<uint32_t>(n_drones),
                       drones_per_env, act_dim, H, actions, action_step_stride, state->act_ring, state->ring_pos);
    const int E = n_drones / drones_per_env;
    hipLaunchKernelGGL(gpd_hist_advance_kernel, dim3(static_cast<unsigned>((E + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       num_steps, E, H, state->ring_pos);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_full_obs (ring update) launch");
    return 0;
}

int gpd_downwash_global(const GpdParams* params, const float* kin, int64_t ld, int32_t n, float cell, float x0,
                        float y0, int32_t nx, int32_t ny, float z0, float zbin, int32_t nz, const int32_t* visit_order,
                        int32_t* cell_count, int32_t* cell_start, int32_t* order, float* sorted_xyzc, float* dw_out,
                        const GpdState* vec_state, const float* vec_obs12, float* vec_out, void* stream) {
    if (!params || !kin || !cell_count || !cell_start || !order || !sorted_xyzc || !dw_out)
        return fail(GPD_EINVAL, "gpd_downwash_global: NULL argument");
    if (n <= 0 || ld < n) return fail(GPD_EINVAL, "gpd_downwash_global: need 0 < n <= ld");
    if (visit_order == order) return fail(GPD_EINVAL, "gpd_downwash_global: visit_order must not alias order (ping-pong two buffers)");
    if (!(cell >= 10.0f)) return fail(GPD_EINVAL, "gpd_downwash_global: cell must be >= 10 m (the model's lateral cut-off)");
    if (nz < 1 || nz > kBlock || (nz > 1 && !(zbin > 0.0f))) return fail(GPD_EINVAL, "gpd_downwash_global: need 1 <= nz <= 256 and zbin > 0");
    if (nx < 3 || ny < 3 || static_cast<int64_t>(nx) * ny * nz > 65536)
        return fail(GPD_ERANGE, "gpd_downwash_global: need nx, ny >= 3 (periodic 3x3 search) and nx*ny*nz <= 65536");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cells = nx * ny, keys = cells * nz;
    const DwGrid G{1.0f / cell, x0, y0, nx, ny, z0, nz > 1 ? 1.0f / zbin : 0.0f, nz};
    hipError_t e;
    const dim3 grid(static_cast<unsigned>((n + kBlock - 1) / kBlock));
    const DwPos src{kin, ld, nullptr};
    if (vec_out) {
        if (!vec_state || !vec_state->kin || !vec_obs12) return fail(GPD_EINVAL, "gpd_downwash_global: vec_out needs vec_state and vec_obs12");
        if (vec_state->ld < n) return fail(GPD_EINVAL, "gpd_downwash_global: vec_state.ld < n");
        hipLaunchKernelGGL(dwg_count_kernel<true>, grid, dim3(kBlock), 0, st, src, n, G, visit_order, cell_count, *vec_state,
                           vec_obs12, vec_out);
    } else {
        hipLaunchKernelGGL(dwg_count_kernel<false>, grid, dim3(kBlock), 0, st, src, n, G, visit_order, cell_count, GpdState{},
                           nullptr, nullptr);
    }
    int32_t* const cursors = cell_count + keys + 1;       // second half of cell_count: the scatter's per-key cursors
    const DwBinOut B{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, n, 0.0f, 0};
    if (keys <= kDwScanMax) {
        hipLaunchKernelGGL(dwg_scatter_kernel<true>, grid, dim3(kBlock), 0, st, src, n, G, visit_order, cell_count, cursors,
                           cell_start, order, reinterpret_cast<float4*>(sorted_xyzc), dw_out, B);
    } else {
        hipLaunchKernelGGL(dwg_scan_kernel, dim3(1), dim3(1024), 0, st, cell_count, cell_start, keys);
        hipLaunchKernelGGL(dwg_scatter_kernel<false>, grid, dim3(kBlock), 0, st, src, n, G, visit_order, cell_count, cursors,
                           cell_start, order, reinterpret_cast<float4*>(sorted_xyzc), dw_out, B);
    }
    const DwWorld Wd{nullptr, nullptr, nullptr, 0, n, 0, 0, 0, cell, nullptr, 0.0f, n};
    hipLaunchKernelGGL(dwg_force_kernel<0>, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(kBlock), 0, st, *params, G, Wd, DwLists{},
                       cell_start, order, reinterpret_cast<const float4*>(sorted_xyzc), dw_out, cell_count);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_downwash_global launch");
    return 0;
}

static int swarm_args(const char* who, const GpdSwarm* w, bool sorted_buffers) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string(who) + ": " + msg).c_str()); };
    if (!w) return bad(GPD_EINVAL, "NULL swarm");
    if (w->world_size < 1 || w->world_size > kBlock || w->rank < 0 || w->rank >= w->world_size) return bad(GPD_EINVAL, "need 0 <= rank < world_size <= 256");
    if (w->meta_rows < 1 || w->own_count < 0 || w->own_count > w->slab - w->meta_rows)
        return bad(GPD_EINVAL, "need 0 <= own_count <= slab - meta_rows (the last meta_rows rows of a slab are its meta rows)");
    if (static_cast<int64_t>(w->meta_rows) * kBlock < w->own_count) return bad(GPD_EINVAL, "need meta_rows >= ceil(own_count / 256): one per workgroup of gpd_swarm_step");
    if ((w->slot_of == nullptr) != (w->pos_sorted == nullptr)) return bad(GPD_EINVAL, "slot_of and pos_sorted come together");
    if (w->slot_of && w->world_size != 1) return bad(GPD_EINVAL, "positions by sorted slot (slot_of / pos_sorted) need the whole world on this rank");
    if (static_cast<int64_t>(w->slab) * w->world_size != w->n_rows) return bad(GPD_EINVAL, "n_rows must be world_size * slab");
    if (w->n_rows > (1 << 26)) return bad(GPD_ERANGE, "more than 2^26 rows");
    if (!w->pos4 || !w->bin_pos || !w->drift) return bad(GPD_EINVAL, "NULL pos4 / bin_pos / drift");
    if (w->total_drones < 1 || w->total_drones > w->n_rows) return bad(GPD_EINVAL, "need 1 <= total_drones <= n_rows");
    if (sorted_buffers) {
        if (!w->cell_count || !w->cell_start || !w->order || !w->slot_key) return bad(GPD_EINVAL, "NULL cell_count / cell_start / order / slot_key");
        if (w->visit && (w->visit == w->order || w->visit == w->visit_out)) return bad(GPD_EINVAL, "visit must alias neither order nor visit_out (ping-pong visit / visit_out)");
        if (!(w->cell >= 10.0f)) return bad(GPD_EINVAL, "cell must be >= 10 m (the model's lateral cut-off)");
        if (w->nz < 1 || w->nz > kBlock || (w->nz > 1 && !(w->zbin > 0.0f))) return bad(GPD_EINVAL, "need 1 <= nz <= 256 and zbin > 0");
        if (w->nx < 3 || w->ny < 3 || static_cast<int64_t>(w->nx) * w->ny * w->nz > 65536)
            return bad(GPD_ERANGE, "need nx, ny >= 3 (periodic search) and nx*ny*nz <= 65536");
    }
    return 0;
}

int gpd_swarm_step(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const GpdSwarm* swarm,
                   const float* action, float* obs12, float* vec_out, void* stream) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string("gpd_swarm_step: ") + msg).c_str()); };
    if (!params || !state || !cfg || !action || !obs12) return bad(GPD_EINVAL, "NULL params/state/cfg/action/obs12");
    if (int rc = swarm_args("gpd_swarm_step", swarm, false)) return rc;
    if (!state->kin || !state->step_counter) return bad(GPD_EINVAL, "NULL state.kin/step_counter");
    if (cfg->drones_per_env != 1 || cfg->num_envs != swarm->own_count || cfg->num_envs <= 0) return bad(GPD_EINVAL, "need drones_per_env == 1 and num_envs == swarm.own_count > 0");
    if (cfg->substeps != 1 || cfg->task != GPD_TASK_NONE || cfg->auto_reset) return bad(GPD_ENOTSUP, "one physics sub-step per call, no task, no auto-reset");
    if (cfg->act_type != GPD_ACT_RPM && cfg->act_type != GPD_ACT_RAW_RPM && cfg->act_type != GPD_ACT_DIRECT_RPM)
        return bad(GPD_ENOTSUP, "act_type RPM, RAW_RPM or DIRECT_RPM (waypoints: gpd_pid first)");
    if (cfg->physics_flags & ~31u) return bad(GPD_EINVAL, "unknown physics flag");
    if (state->ld < cfg->num_envs) return bad(GPD_EINVAL, "state.ld < num_envs");
    if ((cfg->physics_flags & GPD_PHYS_DRAG) && !state->last_rpm) return bad(GPD_EINVAL, "GPD_PHYS_DRAG needs state.last_rpm");
    const size_t lo = static_cast<size_t>(swarm->rank) * swarm->slab;
    const SwarmOut O{reinterpret_cast<float4*>(swarm->pos4) + lo, reinterpret_cast<const float4*>(swarm->bin_pos) + lo,
                     swarm->pos4 + (lo + swarm->slab - swarm->meta_rows) * 4, swarm->slot_of ? swarm->slot_of + lo : nullptr,
                     reinterpret_cast<float4*>(swarm->pos_sorted), swarm->drift, vec_out};
    const dim3 grid(static_cast<unsigned>((cfg->num_envs + kBlock - 1) / kBlock));
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (cfg->act_type) {
        case GPD_ACT_RAW_RPM: hipLaunchKernelGGL(gpd_swarm_step_kernel<GPD_ACT_RAW_RPM>, grid, dim3(kBlock), 0, st, *params, *state, *cfg, action, obs12, O); break;
        case GPD_ACT_DIRECT_RPM: hipLaunchKernelGGL(gpd_swarm_step_kernel<GPD_ACT_DIRECT_RPM>, grid, dim3(kBlock), 0, st, *params, *state, *cfg, action, obs12, O); break;
        default: hipLaunchKernelGGL(gpd_swarm_step_kernel<GPD_ACT_RPM>, grid, dim3(kBlock), 0, st, *params, *state, *cfg, action, obs12, O); break;
    }
    // (too many meta rows for every force workgroup to read: leave the rank's maximum in its first meta row)
    if (static_cast<int64_t>(swarm->world_size) * swarm->meta_rows > 1024)
        hipLaunchKernelGGL(dwg_reduce_meta_kernel, dim3(1), dim3(kBlock), 0, st, O.meta_own, static_cast<int>(grid.x));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_swarm_step launch");
    return 0;
}

int gpd_swarm_pack(const GpdState* state, const GpdSwarm* swarm, const float* obs12, float* vec_out, void* stream) {
    if (!state || !state->kin) return fail(GPD_EINVAL, "gpd_swarm_pack: NULL state / state.kin");
    if (int rc = swarm_args("gpd_swarm_pack", swarm, false)) return rc;
    if (state->ld < swarm->own_count) return fail(GPD_EINVAL, "gpd_swarm_pack: state.ld < own_count");
    if (vec_out && !obs12) return fail(GPD_EINVAL, "gpd_swarm_pack: vec_out needs obs12");
    const size_t lo = static_cast<size_t>(swarm->rank) * swarm->slab;
    hipLaunchKernelGGL(gpd_swarm_pack_kernel, dim3(static_cast<unsigned>((swarm->slab + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), *state, swarm->own_count, swarm->slab, reinterpret_cast<float4*>(swarm->pos4) + lo,
                       obs12, vec_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_swarm_pack launch");
    return 0;
}

int gpd_swarm_bin(const GpdSwarm* w, void* stream) {
    if (int rc = swarm_args("gpd_swarm_bin", w, true)) return rc;
    if (!w->dw_force) return fail(GPD_EINVAL, "gpd_swarm_bin: NULL dw_force (a drone without a finite position gets force 0 here)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n = w->n_rows, keys = w->nx * w->ny * w->nz;
    const DwGrid G{1.0f / w->cell, w->x0, w->y0, w->nx, w->ny, w->z0, w->nz > 1 ? 1.0f / w->zbin : 0.0f, w->nz};
    const dim3 grid(static_cast<unsigned>((n + kBlock - 1) / kBlock));
    const DwPos src{nullptr, 0, reinterpret_cast<const float4*>(w->pos4)};
    hipLaunchKernelGGL(dwg_count_kernel<false>, grid, dim3(kBlock), 0, st, src, n, G, w->visit, w->cell_count, GpdState{}, nullptr, nullptr);
    int32_t* const cursors = w->cell_count + keys + 1;
    const DwBinOut B{w->slot_key, w->slot_of, w->visit_out, w->list_ok, reinterpret_cast<float4*>(w->bin_pos), w->pos4, w->drift, w->slab, w->world_size, w->meta_rows, w->rank * w->slab, w->own_count,
                     w->pair_list ? w->list_delta : 0.0f, w->list_adapt != 0};
    float4* const srt = reinterpret_cast<float4*>(w->pos_sorted);
    if (keys <= kDwScanMax) {
        hipLaunchKernelGGL(dwg_scatter_kernel<true>, grid, dim3(kBlock), 0, st, src, n, G, w->visit, w->cell_count, cursors,
                           w->cell_start, w->order, srt, w->dw_force, B);
    } else {
        hipLaunchKernelGGL(dwg_scan_kernel, dim3(1), dim3(1024), 0, st, w->cell_count, w->cell_start, keys);
        hipLaunchKernelGGL(dwg_scatter_kernel<false>, grid, dim3(kBlock), 0, st, src, n, G, w->visit, w->cell_count, cursors,
                           w->cell_start, w->order, srt, w->dw_force, B);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_swarm_bin launch");
    return 0;
}

int gpd_swarm_forces(const GpdParams* params, const GpdSwarm* w, int32_t build_lists, void* stream) {
    if (!params) return fail(GPD_EINVAL, "gpd_swarm_forces: NULL params");
    if (int rc = swarm_args("gpd_swarm_forces", w, true)) return rc;
    if (!w->dw_force) return fail(GPD_EINVAL, "gpd_swarm_forces: NULL dw_force");
    const bool lists = w->pair_list != nullptr;
    if (lists && (!w->pair_nb || !w->list_ok || w->list_cap < 4 || w->list_cap > 65535 || !(w->list_delta >= 0.0f)))
        return fail(GPD_EINVAL, "gpd_swarm_forces: pair_list needs pair_nb, list_ok, 4 <= list_cap <= 65535 and list_delta >= 0");
    // (an entry is lane << 26 | index and 0xffffffff marks an empty lane: index 2^26 - 1 of lane 63 must not exist)
    if (lists && w->n_rows >= (1 << 26)) return fail(GPD_ERANGE, "gpd_swarm_forces: wake lists address fewer than 2^26 rows");
    const DwGrid G{1.0f / w->cell, w->x0, w->y0, w->nx, w->ny, w->z0, w->nz > 1 ? 1.0f / w->zbin : 0.0f, w->nz};
    const float4* const p4 = reinterpret_cast<const float4*>(w->pos4);
    const DwWorld Wd{w->pos_sorted ? nullptr : p4, w->slot_key, p4, w->rank * w->slab, w->own_count, w->slab, w->world_size, w->meta_rows, w->cell,
                     w->drift, 1.0f / static_cast<float>(w->total_drones), w->n_rows};
    const DwLists Ls{w->pair_list, w->pair_nb, w->list_ok, w->list_cap, w->list_delta};
    const dim3 grid(static_cast<unsigned>((w->n_rows + 63) / 64) + 1u);        // (+ the workgroup that computes the drift)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float4* const srt = reinterpret_cast<const float4*>(w->pos_sorted);
    if (!lists) hipLaunchKernelGGL(dwg_force_kernel<0>, grid, dim3(kBlock), 0, st, *params, G, Wd, Ls, w->cell_start, w->order, srt, w->dw_force, w->cell_count);
    else if (build_lists) hipLaunchKernelGGL(dwg_force_kernel<1>, grid, dim3(kBlock), 0, st, *params, G, Wd, Ls, w->cell_start, w->order, srt, w->dw_force, w->cell_count);
    else hipLaunchKernelGGL(dwg_force_kernel<2>, grid, dim3(kBlock), 0, st, *params, G, Wd, Ls, w->cell_start, w->order, srt, w->dw_force, w->cell_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_swarm_forces launch");
    return 0;
}

int gpd_reset(const GpdState* state, const float* init_pose, int32_t init_per_env, const uint8_t* mask,
              int32_t num_envs, int32_t drones_per_env, int32_t reset_pid, float* obs12, void* stream) {
    if (!state || !state->kin || !state->step_counter || !init_pose)
        return fail(GPD_EINVAL, "gpd_reset: NULL state/init_pose");
    if (num_envs <= 0 || drones_per_env <= 0) return fail(GPD_EINVAL, "gpd_reset: sizes must be positive");
    const int64_t N = static_cast<int64_t>(num_envs) * drones_per_env;
    if (state->ld < N) return fail(GPD_EINVAL, "gpd_reset: state.ld < num_envs*drones_per_env");
    const int64_t blocks = (N + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(gpd_reset_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), *state, init_pose, init_per_env, mask, num_envs,
                       drones_per_env, reset_pid, obs12);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_reset launch");
    return 0;
}

int gpd_pid(const GpdParams* params, float* pid, int64_t ld, float ctrl_dt, const float* cur_pos,
            const float* cur_quat, const float* cur_vel, const float* target_pos, const float* target_rpy,
            const float* target_vel, const float* target_rpy_rates, float* rpm, float* pos_e, float* yaw_e,
            int32_t n, void* stream) {
    if (!params || !pid || !cur_pos || !cur_quat || !cur_vel || !target_pos || !rpm)
        return fail(GPD_EINVAL, "gpd_pid: NULL argument");
    if (n <= 0 || ld < n) return fail(GPD_EINVAL, "gpd_pid: need 0 < n <= ld");
    if (params->pid_kf <= 0.0f) return fail(GPD_ENOTSUP, "gpd_pid: no DSLPID controller for this airframe");
    const int blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(gpd_pid_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *params, pid,
                       ld, ctrl_dt, cur_pos, cur_quat, cur_vel, target_pos, target_rpy, target_vel, target_rpy_rates,
                       rpm, pos_e, yaw_e, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_pid launch");
    return 0;
}

int gpd_state_vectors(const GpdState* state, const float* obs12, float* state20, int32_t n, void* stream) {
    if (!state || !state->kin || !obs12 || !state20) return fail(GPD_EINVAL, "gpd_state_vectors: NULL argument");
    if (n <= 0 || state->ld < n) return fail(GPD_EINVAL, "gpd_state_vectors: need 0 < n <= state.ld");
    const int blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(gpd_state20_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *state,
                       obs12, state20, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_state_vectors launch");
    return 0;
}

int gpd_comm_unique_id(uint8_t id[GPD_COMM_ID_BYTES]) {
    if (!id) return fail(GPD_EINVAL, "gpd_comm_unique_id: NULL id");
    if (int rc = need_rccl("gpd_comm_unique_id")) return rc;
    static_assert(GPD_COMM_ID_BYTES == sizeof(ncclUniqueId), "GPD_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId u;
    ncclResult_t e = rccl().GetUniqueId(&u);
    if (e != ncclSuccess) return rccl_fail(e, "ncclGetUniqueId");
    memcpy(id, &u, sizeof(u));
    return 0;
}

int gpd_comm_init(void** comm, const uint8_t id[GPD_COMM_ID_BYTES], int32_t rank, int32_t world_size) {
    if (!comm || !id) return fail(GPD_EINVAL, "gpd_comm_init: NULL comm/id");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(GPD_EINVAL, "gpd_comm_init: need 0 <= rank < world_size");
    if (int rc = need_rccl("gpd_comm_init")) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    ncclResult_t e = rccl().CommInitRank(&c, world_size, u, rank);
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommInitRank");
    *comm = c;
    return 0;
}

int gpd_comm_count(void* comm, int32_t* n_ranks) {
    if (!comm || !n_ranks) return fail(GPD_EINVAL, "gpd_comm_count: NULL comm/n_ranks");
    if (int rc = need_rccl("gpd_comm_count")) return rc;
    int n = 0;
    ncclResult_t e = rccl().CommCount(static_cast<ncclComm_t>(comm), &n);
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommCount");
    *n_ranks = n;
    return 0;
}

int gpd_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (int rc = need_rccl("gpd_comm_destroy")) return rc;
    ncclResult_t e = rccl().CommDestroy(static_cast<ncclComm_t>(comm));
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommDestroy");
    return 0;
}

int gpd_allgather_obs(void* comm, const float* shard, float* full, size_t count, void* stream) {
    if (!comm || !shard || !full) return fail(GPD_EINVAL, "gpd_allgather_obs: NULL comm/shard/full");
    if (count == 0) return fail(GPD_EINVAL, "gpd_allgather_obs: count must be positive");
    if (int rc = need_rccl("gpd_allgather_obs")) return rc;
    ncclResult_t e = rccl().AllGather(shard, full, count, ncclFloat32, static_cast<ncclComm_t>(comm),
                                      static_cast<hipStream_t>(stream));
    if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather");
    return 0;
}

int gpd_p2p_group(void* comm, const GpdP2P* sends, int32_t n_sends, const GpdP2P* recvs, int32_t n_recvs, void* stream) {
    if (!comm || (n_sends > 0 && !sends) || (n_recvs > 0 && !recvs) || n_sends < 0 || n_recvs < 0)
        return fail(GPD_EINVAL, "gpd_p2p_group: NULL comm / operation list");
    if (int rc = need_rccl("gpd_p2p_group")) return rc;
    Rccl& R = rccl();
    if (!R.GroupStart || !R.GroupEnd || !R.Send || !R.Recv) return fail(GPD_ENOTSUP, "gpd_p2p_group: this RCCL has no ncclSend / ncclRecv / ncclGroup*");
    for (int i = 0; i < n_sends; ++i) if (!sends[i].ptr || sends[i].count <= 0 || sends[i].peer < 0) return fail(GPD_EINVAL, "gpd_p2p_group: bad send operation");
    for (int i = 0; i < n_recvs; ++i) if (!recvs[i].ptr || recvs[i].count <= 0 || recvs[i].peer < 0) return fail(GPD_EINVAL, "gpd_p2p_group: bad receive operation");
    if (n_sends + n_recvs == 0) return 0;
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ncclResult_t e = R.GroupStart();
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupStart");
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n_sends && first == ncclSuccess; ++i)
        first = R.Send(sends[i].ptr, static_cast<size_t>(sends[i].count), ncclFloat32, sends[i].peer, c, st);
    for (int i = 0; i < n_recvs && first == ncclSuccess; ++i)
        first = R.Recv(recvs[i].ptr, static_cast<size_t>(recvs[i].count), ncclFloat32, recvs[i].peer, c, st);
    e = R.GroupEnd();                                  // (always closed, also after a failed enqueue)
    if (first != ncclSuccess) return rccl_fail(first, "ncclSend / ncclRecv");
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupEnd");
    return 0;
}

#ifdef GPD_EXP_TS
extern "C" int gpd_debug_ts(unsigned long long* ts, unsigned int* cnt) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(gpd_ts), sizeof(unsigned long long) * 8 * 4096 * 4);
    hipMemcpyFromSymbol(cnt, HIP_SYMBOL(gpd_ts_cnt), sizeof(unsigned int) * 4096);
    return 0;
}
#endif
int gpd_debug_status(uint32_t out[4], int32_t reset, void* stream) {
#ifdef GPD_DEBUG_BOUNDS
    if (!out) return fail(GPD_EINVAL, "gpd_debug_status: NULL out");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(gpd_dbg_word), 4 * sizeof(uint32_t));
    if (e == hipSuccess && reset) {
        const uint32_t zero[4] = {0u, 0u, 0u, 0u};
        e = hipMemcpyToSymbol(HIP_SYMBOL(gpd_dbg_word), zero, sizeof(zero));
    }
    if (e != hipSuccess) return hip_fail(e, "gpd_debug_status");
    return 0;
#else
    (void)out; (void)reset; (void)stream;
    return fail(GPD_ENOTSUP, "gpd_debug_status: this is a release build (no -DGPD_DEBUG_BOUNDS)");
#endif
}

int gpd_clock_probe(double* shader_ghz, double* ns_per_fma, void* stream) {
    if (!shader_ghz) return fail(GPD_EINVAL, "gpd_clock_probe: NULL shader_ghz");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int dev = 0, wall_khz = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev);
    if (e != hipSuccess || wall_khz <= 0) return hip_fail(e, "gpd_clock_probe: wall clock rate");
    unsigned long long* d = nullptr;
    e = hipMalloc(&d, 3 * sizeof(unsigned long long));
    if (e != hipSuccess) return hip_fail(e, "gpd_clock_probe: hipMalloc");
    const int iters = 4000;                                  // x 64 dependent FMAs: ~0.6 ms
    unsigned long long h[2] = {0, 0};
    for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {   // (the second pass runs at the ramped-up clock)
        hipLaunchKernelGGL(gpd_clock_probe_kernel, dim3(256), dim3(kBlock), 0, st, d, iters);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "gpd_clock_probe");
    if (h[1] == 0) return fail(GPD_ENOTSUP, "gpd_clock_probe: the wall clock did not advance");
    const double secs = static_cast<double>(h[1]) / (static_cast<double>(wall_khz) * 1e3);
    *shader_ghz = static_cast<double>(h[0]) / secs * 1e-9;
    if (ns_per_fma) *ns_per_fma = secs * 1e9 / (static_cast<double>(iters) * 64.0);
    return 0;
}


#endif  // !GPD_POLICY_TU

}  // extern "C"
