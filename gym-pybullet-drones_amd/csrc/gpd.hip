// gpd.hip — the MI355X (gfx950 / CDNA4) hot path of the vectorised quadrotor simulator.
//
// One wavefront lane per drone.  A lane loads its 13 kinematic floats from the structure-of-arrays
// state (each field a contiguous float[N] => a wave's load of one field is one coalesced 256 B
// transaction), keeps them in VGPRs across all `substeps` physics sub-steps, and stores them back
// once; the per-airframe constants arrive BY VALUE in the kernel-argument segment and therefore
// live in SGPRs (they are wave-uniform).  No MFMA: there is no dense contraction on this path.
//
// What is fused (reference = utiasDSL/gym-pybullet-drones, gym_pybullet_drones/...):
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
// Arithmetic: fp32, FMA contraction on (hipcc default), NO -ffast-math.  The kernel at N = 65 536 is
// bound by the length of one wave's dependent VALU chain (one wave per SIMD), so the hot functions are
// written for few instructions at <= 2 ulp instead of calling the branchy IEEE/OCML versions:
//   1/x, sqrt, 1/sqrt      v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (1 ulp)
//   quaternion exponential cos(t) and sin(t)/t as even polynomials in t^2 (no sqrt, no division, no
//                          range reduction; |t| <= 1 rad per sub-step, exact OCML path beyond)
//   atan2 / asin           one odd minimax polynomial on [0,1] (abs err 7e-8) + octant fix-up
// Build: hipcc -O3 --offload-arch=gfx950 -fPIC -shared
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "gpd.h"

namespace {

#ifndef GPD_BLOCK
#define GPD_BLOCK 256
#endif
constexpr int kBlock = GPD_BLOCK;   // 256 = 4 wavefronts; one workgroup per CU fills all 4 SIMDs

thread_local std::string g_last_error;

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
};

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// atan2 with the IEEE sign/quadrant conventions Bullet's Euler extraction relies on; minimax
// polynomial for atan(t)/t in t^2 on [0,1] (max abs error 7.4e-8 evaluated in fp32)
__device__ __forceinline__ float atan2_poly(float y, float x) {
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

// asin(s) = atan2(s, sqrt((1-s)(1+s))); the factored form keeps full relative accuracy near |s| = 1
__device__ __forceinline__ float asin_poly(float s) {
    return atan2_poly(s, fast_sqrt(fmaxf((1.0f - s) * (1.0f + s), 0.0f)));
}

// btMatrix3x3::setRotation (reached via p.getMatrixFromQuaternion, envs/BaseAviary.py:836);
// insensitive to |q| (DYN never renormalises q)
__device__ __forceinline__ Mat3 quat_to_mat(float x, float y, float z, float w) {
    const float d = x * x + y * y + z * z + w * w;
    const float s = 2.0f * fast_rcp(d);
    const float xs = x * s, ys = y * s, zs = z * s;
    const float wx = w * xs, wy = w * ys, wz = w * zs;
    const float xx = x * xs, xy = x * ys, xz = x * zs;
    const float yy = y * ys, yz = y * zs, zz = z * zs;
    Mat3 R;
    R.r00 = 1.0f - (yy + zz); R.r01 = xy - wz;          R.r02 = xz + wy;
    R.r10 = xy + wz;          R.r11 = 1.0f - (xx + zz); R.r12 = yz - wx;
    R.r20 = xz - wy;          R.r21 = yz + wx;          R.r22 = 1.0f - (xx + yy);
    return R;
}

// pybullet_getEulerFromQuaternion incl. its gimbal branches (envs/BaseAviary.py:518)
__device__ __forceinline__ void quat_to_rpy(float x, float y, float z, float w,
                                            float& roll, float& pitch, float& yaw) {
    const float sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
    const float sarg = -2.0f * (x * z - w * y);
    if (sarg <= -0.99999f) {
        roll = 0.0f; pitch = -1.57079632679489661923f; yaw = 2.0f * atan2_poly(x, -y);
    } else if (sarg >= 0.99999f) {
        roll = 0.0f; pitch = 1.57079632679489661923f; yaw = 2.0f * atan2_poly(-x, y);
    } else {
        roll = atan2_poly(2.0f * (y * z + w * x), squ - sqx - sqy + sqz);
        pitch = asin_poly(sarg);
        yaw = atan2_poly(2.0f * (x * y + w * z), squ + sqx - sqy - sqz);
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
    s.ipx = clampf(s.ipx + epx * dt, -2.0f, 2.0f);
    s.ipy = clampf(s.ipy + epy * dt, -2.0f, 2.0f);
    s.ipz = clampf(clampf(s.ipz + epz * dt, -2.0f, 2.0f), -0.15f, 0.15f);
    const float fx = P.p_for[0] * epx + P.i_for[0] * s.ipx + P.d_for[0] * evx;
    const float fy = P.p_for[1] * epy + P.i_for[1] * s.ipy + P.d_for[1] * evy;
    const float fz = P.p_for[2] * epz + P.i_for[2] * s.ipz + P.d_for[2] * evz + P.pid_gravity;
    const float along = fmaxf(0.0f, fx * R.r02 + fy * R.r12 + fz * R.r22);
    const float base_pwm = (fast_sqrt(along * P.pid_inv_4kf) - P.pwm2rpm_const) * P.inv_pwm2rpm_scale;
    const float fn = fast_rsq(fx * fx + fy * fy + fz * fz);
    const float zbx = fx * fn, zby = fy * fn, zbz = fz * fn;
    float sy, cy;
    sincosf(tyaw, &sy, &cy);                      // heading = [cos, sin, 0]
    float ybx = zby * 0.0f - zbz * sy;            // zb x heading
    float yby = zbz * cy - zbx * 0.0f;
    float ybz = zbx * sy - zby * cy;
    const float yn = fast_rsq(ybx * ybx + yby * yby + ybz * ybz);
    ybx *= yn; yby *= yn; ybz *= yn;
    const float xbx = yby * zbz - ybz * zby;      // yb x zb
    const float xby = ybz * zbx - ybx * zbz;
    const float xbz = ybx * zby - yby * zbx;
    // ---- attitude loop, :240-259.  The reference rebuilds R* from its Euler angles through scipy;
    // that round trip returns the same orthonormal matrix (to 3e-16), so R* = [xb yb zb] is used.
    // e_R = vee(R*^T R - R^T R*): M_ij = col_i(R*) . col_j(R)
    const float m21 = zbx * R.r01 + zby * R.r11 + zbz * R.r21, m12 = ybx * R.r02 + yby * R.r12 + ybz * R.r22;
    const float m02 = xbx * R.r02 + xby * R.r12 + xbz * R.r22, m20 = zbx * R.r00 + zby * R.r10 + zbz * R.r20;
    const float m10 = ybx * R.r00 + yby * R.r10 + ybz * R.r20, m01 = xbx * R.r01 + xby * R.r11 + xbz * R.r21;
    const float erx = m21 - m12, ery = m02 - m20, erz = m10 - m01;
    const float ewx = trr - (roll - s.lr) * inv_dt;   // finite difference of Euler angles, no unwrap (:247)
    const float ewy = trp - (pitch - s.lp) * inv_dt;
    const float ewz = try_ - (yaw - s.ly) * inv_dt;
    s.lr = roll; s.lp = pitch; s.ly = yaw;
    s.irx = clampf(clampf(s.irx - erx * dt, -1500.0f, 1500.0f), -1.0f, 1.0f);
    s.iry = clampf(clampf(s.iry - ery * dt, -1500.0f, 1500.0f), -1.0f, 1.0f);
    s.irz = clampf(s.irz - erz * dt, -1500.0f, 1500.0f);
    const float t0 = clampf(-P.p_tor[0] * erx + P.d_tor[0] * ewx + P.i_tor[0] * s.irx, -3200.0f, 3200.0f);
    const float t1 = clampf(-P.p_tor[1] * ery + P.d_tor[1] * ewy + P.i_tor[1] * s.iry, -3200.0f, 3200.0f);
    const float t2 = clampf(-P.p_tor[2] * erz + P.d_tor[2] * ewz + P.i_tor[2] * s.irz, -3200.0f, 3200.0f);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float pwm = clampf(base_pwm + P.mixer[3 * m] * t0 + P.mixer[3 * m + 1] * t1 + P.mixer[3 * m + 2] * t2,
                                 P.min_pwm, P.max_pwm);
        rpm[m] = P.pwm2rpm_scale * pwm + P.pwm2rpm_const;
    }
    if (pos_e) { pos_e[0] = epx; pos_e[1] = epy; pos_e[2] = epz; }
    if (yaw_e) {
        // yaw of the intrinsic-XYZ Euler angles of R* (scipy as_euler('XYZ'), :205): atan2(-R01, R00)
        *yaw_e = atan2_poly(-ybx, xbx) - yaw;
    }
}

// One physics sub-step (envs/BaseAviary.py:831-877 + :879-892), state in registers.
//   rpm[4]      current action
//   drag_rpm_sum  sum of the rpm the drag term sees (previous action on sub-step 0), only if DRAG
//   dw_force    body-z downwash force on this drone (already summed over the drones above), only if DW
template <bool EXT>
__device__ __forceinline__ void substep(const GpdParams& P, float h, uint32_t flags, const float rpm[4],
                                        float drag_rpm_sum, float dw_force, Kin& k,
                                        float& avx, float& avy, float& avz) {
    const Mat3 R = quat_to_mat(k.qx, k.qy, k.qz, k.qw);
    const float s0 = rpm[0] * rpm[0], s1 = rpm[1] * rpm[1], s2 = rpm[2] * rpm[2], s3 = rpm[3] * rpm[3];
    float f0 = s0 * P.KF, f1 = s1 * P.KF, f2 = s2 * P.KF, f3 = s3 * P.KF;
    if (EXT && (flags & GPD_PHYS_GND)) {
        // per-rotor extra thrust (:739-743): h_i = world z of rotor i, clipped from below
        const float sq[4] = {s0, s1, s2, s3};
        float g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float hz = k.pz + R.r20 * P.prop_x[i] + R.r21 * P.prop_y[i];
            hz = fmaxf(hz, P.gnd_eff_h_clip);
            const float ratio = (0.25f * P.prop_radius) * fast_rcp(hz);
            g[i] = sq[i] * P.KF * P.gnd_eff_coeff * (ratio * ratio);
        }
        // |roll| < pi/2 and |pitch| < pi/2 on Bullet's Euler extraction (:742), without the atan2/asin:
        // pitch = asin(sarg) is inside (-pi/2, pi/2) off the gimbal branches and exactly +-pi/2 on them;
        // roll = atan2(a, b) has |roll| < pi/2 iff b > 0 (or a == b == 0).
        const float sarg = -2.0f * (k.qx * k.qz - k.qw * k.qy);
        const float a = 2.0f * (k.qy * k.qz + k.qw * k.qx);
        const float b = k.qw * k.qw - k.qx * k.qx - k.qy * k.qy + k.qz * k.qz;
        const bool on = (sarg > -0.99999f) && (sarg < 0.99999f) && (b > 0.0f || (a == 0.0f && b == 0.0f));
        if (on) { f0 += g[0]; f1 += g[1]; f2 += g[2]; f3 += g[3]; }
    }
    float fzb = ((f0 + f1) + f2) + f3;                     // np.sum order
    if (EXT && (flags & GPD_PHYS_DW)) fzb += dw_force;
    float Fx = R.r02 * fzb, Fy = R.r12 * fzb, Fz = R.r22 * fzb - P.GRAVITY;
    if (EXT && (flags & GPD_PHYS_DRAG)) {
        // world force -DRAG_COEFF * v * sum(2*pi*rpm/60) (:771-774; R R^T cancels)
        const float wsum = drag_rpm_sum * (6.28318530717958647692f / 60.0f);
        Fx -= P.drag_coeff[0] * k.vx * wsum;
        Fy -= P.drag_coeff[1] * k.vy * wsum;
        Fz -= P.drag_coeff[2] * k.vz * wsum;
    }
    float z0 = s0 * P.KM, z1 = s1 * P.KM, z2 = s2 * P.KM, z3 = s3 * P.KM;
    if (P.drone_model == GPD_MODEL_RACE) { z0 = -z0; z1 = -z1; z2 = -z2; z3 = -z3; }
    const float tz = -z0 + z1 - z2 + z3;
    float tx, ty;
    if (P.drone_model == GPD_MODEL_CF2P) {
        tx = (f1 - f3) * P.L;
        ty = (-f0 + f2) * P.L;
    } else {
        const float arm = P.L * 0.70710678118654752440f;   // L / sqrt(2)
        tx = (f0 + f1 - f2 - f3) * arm;
        ty = (-f0 + f1 + f2 - f3) * arm;
        if (P.drone_model == GPD_MODEL_CF2X) tx = -tx;
    }
    // Euler's rotation equation with diagonal J
    const float jwx = P.J[0] * k.wx, jwy = P.J[1] * k.wy, jwz = P.J[2] * k.wz;
    tx -= k.wy * jwz - k.wz * jwy;
    ty -= k.wz * jwx - k.wx * jwz;
    const float tzz = tz - (k.wx * jwy - k.wy * jwx);
    // semi-implicit Euler (:860-862): position uses the NEW velocity
    k.vx += h * (Fx * P.inv_M); k.vy += h * (Fy * P.inv_M); k.vz += h * (Fz * P.inv_M);
    k.wx += h * (P.J_INV[0] * tx); k.wy += h * (P.J_INV[1] * ty); k.wz += h * (P.J_INV[2] * tzz);
    k.px += h * k.vx; k.py += h * k.vy; k.pz += h * k.vz;
    // exact exponential quaternion update q <- q (x) exp(w h / 2)  (:879-892)
    //   q' = cos(t) q + (sin(t)/|w|) (q (x) [w,0]),  t = |w| h / 2.  cos(t) and sin(t)/t are even functions
    //   of t, evaluated as minimax polynomials in u = t^2 (abs err 7e-8 / 5e-8 for t <= 1 rad, i.e. body
    //   rates up to 480 rad/s at 240 Hz): no sqrt, no division, no range reduction.
    const float n2 = k.wx * k.wx + k.wy * k.wy + k.wz * k.wz;
    const float u = n2 * (0.25f * h * h);
    float cs, sc;
    if (__builtin_expect(u <= 1.0f, 1)) {
        cs = fmaf(fmaf(fmaf(fmaf(2.412107309e-05f, u, -1.388295778e-03f), u, 4.166645522e-02f), u, -4.999999736e-01f),
                  u, 9.999999995e-01f);
        sc = fmaf(fmaf(fmaf(fmaf(2.693749890e-06f, u, -1.983586443e-04f), u, 8.333314057e-03f), u, -1.666666643e-01f),
                  u, 1.0f) * (0.5f * h);
    } else {   // tumbling faster than 480 rad/s: exact path
        const float n = sqrtf(n2);
        float sn;
        sincosf(n * h * 0.5f, &sn, &cs);
        sc = sn / n;
    }
    if (n2 > 1e-16f) {                                     // !np.isclose(|w|, 0)  (|w| <= 1e-8 keeps q)
        const float lx = k.wz * k.qy - k.wy * k.qz + k.wx * k.qw;
        const float ly = -k.wz * k.qx + k.wx * k.qz + k.wy * k.qw;
        const float lz = k.wy * k.qx - k.wx * k.qy + k.wz * k.qw;
        const float lw = -k.wx * k.qx - k.wy * k.qy - k.wz * k.qz;
        k.qx = cs * k.qx + sc * lx; k.qy = cs * k.qy + sc * ly;
        k.qz = cs * k.qz + sc * lz; k.qw = cs * k.qw + sc * lw;
    }
    // world angular velocity handed to the state store: PRE-update rotation, post-update rates (:873)
    avx = R.r00 * k.wx + R.r01 * k.wy + R.r02 * k.wz;
    avy = R.r10 * k.wx + R.r11 * k.wy + R.r12 * k.wz;
    avz = R.r20 * k.wx + R.r21 * k.wy + R.r22 * k.wz;
}

// SoA row access as  <uniform 64-bit row base in SGPRs> + <32-bit per-lane byte offset>: this is the
// global_load/store "saddr + voffset" form, one VGPR of address for all rows instead of a 64-bit
// add per access.  (gpd_step bounds N so that every byte offset fits 32 bits.)
__device__ __forceinline__ float ld_row(const float* __restrict__ base, int64_t ld, int r, uint32_t off4) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + r * ld) + off4);
}
__device__ __forceinline__ void st_row(float* __restrict__ base, int64_t ld, int r, uint32_t off4, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base + r * ld) + off4) = v;
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
// the fused step kernel
//   PID   : action types that run DSLPID (PID / VEL / ONE_D_PID)
//   EXT   : any of the GND/DRAG/DW terms may be enabled (flags tested at run time, uniformly)
//   MULTI : drones_per_env > 1 (env-level reductions and downwash go through LDS)
// ------------------------------------------------------------------------------------------------
template <bool PID, bool EXT, bool MULTI>
__global__ __launch_bounds__(kBlock) void gpd_step_kernel(
    const GpdParams P, const GpdState S, const GpdStepCfg C, const float* __restrict__ action,
    const float* __restrict__ target_pos, const float* __restrict__ init_pose, float* __restrict__ obs12,
    float* __restrict__ reward, uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
    float* __restrict__ term_obs12) {
    const int D = MULTI ? C.drones_per_env : 1;
    const int tid = threadIdx.x;
    const uint32_t N = static_cast<uint32_t>(C.num_envs) * static_cast<uint32_t>(D);
    // MULTI: whole aviaries per workgroup, one lane per drone.  Single-drone aviaries: L = lanes_per_wave
    // (16/32/64) active lanes per 64-wide wavefront -- a batch too small to fill the chip is spread over
    // more wavefronts so that every SIMD has 2-4 waves to interleave (the kernel is bound by the latency of
    // one wave's dependent instruction chain there, not by issue slots or bandwidth).
    const int L = MULTI ? 64 : C.lanes_per_wave;
    const int lanes = MULTI ? (kBlock / D) * D : (kBlock / 64) * L;
    const uint32_t n = MULTI ? blockIdx.x * lanes + tid : (blockIdx.x * (kBlock / 64) + (tid >> 6)) * L + (tid & 63);
    const uint32_t off4 = n * 4u;
    const bool active = (MULTI ? (tid < lanes) : ((tid & 63) < L)) && (n < N);
    const int le = MULTI ? tid / D : tid;                    // env index inside the workgroup
    const int d = MULTI ? tid - le * D : 0;                  // drone index inside the env
    const uint32_t env = MULTI ? blockIdx.x * (lanes / D) + le : n;
    const int64_t ld = S.ld;

    __shared__ float sh_pos[MULTI ? 3 * kBlock : 1];         // downwash: positions of the env's drones
    __shared__ float sh_red[MULTI ? 3 * kBlock : 1];         // reward | distance | out-of-bounds per drone
    __shared__ int sh_flag[MULTI ? kBlock : 1];              // per-env done flag / counter broadcast

    // ---- load state ------------------------------------------------------------------------------
    Kin k;
    if (active) {
        k.px = ld_row(S.kin, ld, 0, off4); k.py = ld_row(S.kin, ld, 1, off4); k.pz = ld_row(S.kin, ld, 2, off4);
        k.qx = ld_row(S.kin, ld, 3, off4); k.qy = ld_row(S.kin, ld, 4, off4); k.qz = ld_row(S.kin, ld, 5, off4);
        k.qw = ld_row(S.kin, ld, 6, off4);
        k.vx = ld_row(S.kin, ld, 7, off4); k.vy = ld_row(S.kin, ld, 8, off4); k.vz = ld_row(S.kin, ld, 9, off4);
        k.wx = ld_row(S.kin, ld, 10, off4); k.wy = ld_row(S.kin, ld, 11, off4); k.wz = ld_row(S.kin, ld, 12, off4);
    } else {
        k = Kin{0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    }
    // everything the tail of the kernel needs from memory is requested NOW, together with the state, so
    // that no second round trip sits between the physics and the stores (a load issued inside the task
    // section costs a full ~1 us memory latency on the critical path of a launch that lasts ~5 us)
    int counter = 0;
    float tgx = 0.0f, tgy = 0.0f, tgz = 0.0f;
    if (active) {
        if (!MULTI || d == 0) counter = S.step_counter[env];
        if (C.task != GPD_TASK_NONE) {
            const float* tp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(target_pos) +
                                                             (C.target_per_env ? n * 12u : static_cast<uint32_t>(d) * 12u));
            tgx = tp[0]; tgy = tp[1]; tgz = tp[2];
        }
    }

    // ---- action -> RPM (computed ONCE per env step from the cached state, BaseAviary.py:341) -----
    float rpm[4] = {0, 0, 0, 0};
    if (!PID) {
        if (active) {
            if (C.act_type == GPD_ACT_ONE_D_RPM) {
                const float r = P.hover_rpm * (1.0f + 0.05f * ld_row(action, 0, 0, off4));
                rpm[0] = rpm[1] = rpm[2] = rpm[3] = r;
            } else {
                const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(action) + n * 16u);
                if (C.act_type == GPD_ACT_RAW_RPM) {
                    rpm[0] = clampf(a.x, 0.0f, P.max_rpm); rpm[1] = clampf(a.y, 0.0f, P.max_rpm);
                    rpm[2] = clampf(a.z, 0.0f, P.max_rpm); rpm[3] = clampf(a.w, 0.0f, P.max_rpm);
                } else if (C.act_type == GPD_ACT_DIRECT_RPM) {
                    rpm[0] = a.x; rpm[1] = a.y; rpm[2] = a.z; rpm[3] = a.w;
                } else {   // GPD_ACT_RPM: NOT clipped
                    rpm[0] = P.hover_rpm * (1.0f + 0.05f * a.x); rpm[1] = P.hover_rpm * (1.0f + 0.05f * a.y);
                    rpm[2] = P.hover_rpm * (1.0f + 0.05f * a.z); rpm[3] = P.hover_rpm * (1.0f + 0.05f * a.w);
                }
            }
        }
    } else {
        Pid s{0, 0, 0, 0, 0, 0, 0, 0, 0};
        float tx = k.px, ty = k.py, tz = k.pz, tyaw = 0.0f, tvx = 0.0f, tvy = 0.0f, tvz = 0.0f;
        float roll, pitch, yaw;
        quat_to_rpy(k.qx, k.qy, k.qz, k.qw, roll, pitch, yaw);
        if (active) {
            s.ipx = ld_row(S.pid, ld, 0, off4); s.ipy = ld_row(S.pid, ld, 1, off4); s.ipz = ld_row(S.pid, ld, 2, off4);
            s.lr = ld_row(S.pid, ld, 3, off4); s.lp = ld_row(S.pid, ld, 4, off4); s.ly = ld_row(S.pid, ld, 5, off4);
            s.irx = ld_row(S.pid, ld, 6, off4); s.iry = ld_row(S.pid, ld, 7, off4); s.irz = ld_row(S.pid, ld, 8, off4);
            if (C.act_type == GPD_ACT_PID) {
                // waypoint limited to a 1 m approach step (_calculateNextStep, BaseAviary.py:1132-1150)
                const float* ap = reinterpret_cast<const float*>(reinterpret_cast<const char*>(action) + n * 12u);
                const float ax = ap[0], ay = ap[1], az = ap[2];
                const float dx = ax - k.px, dy = ay - k.py, dz = az - k.pz;
                const float d2 = dx * dx + dy * dy + dz * dz;
                if (fast_sqrt(d2) <= 1.0f) { tx = ax; ty = ay; tz = az; }
                else { const float id = fast_rsq(d2); tx = k.px + dx * id; ty = k.py + dy * id; tz = k.pz + dz * id; }
            } else if (C.act_type == GPD_ACT_VEL) {
                const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(action) + n * 16u);
                const float nn2 = a.x * a.x + a.y * a.y + a.z * a.z;
                const float sp = P.speed_limit * fabsf(a.w);
                if (nn2 != 0.0f) { const float in = fast_rsq(nn2); tvx = sp * (a.x * in); tvy = sp * (a.y * in); tvz = sp * (a.z * in); }
                tyaw = yaw;                                   // keep the current yaw (:220)
            } else {   // GPD_ACT_ONE_D_PID
                tz = k.pz + 0.1f * ld_row(action, 0, 0, off4);
            }
        }
        const Mat3 R = quat_to_mat(k.qx, k.qy, k.qz, k.qw);
        dslpid(P, C.ctrl_dt, C.inv_ctrl_dt, k, roll, pitch, yaw, R, tx, ty, tz, tyaw, tvx, tvy, tvz, 0.0f, 0.0f, 0.0f, s,
               rpm, nullptr, nullptr);
        if (active) {
            st_row(S.pid, ld, 0, off4, s.ipx); st_row(S.pid, ld, 1, off4, s.ipy); st_row(S.pid, ld, 2, off4, s.ipz);
            st_row(S.pid, ld, 3, off4, s.lr); st_row(S.pid, ld, 4, off4, s.lp); st_row(S.pid, ld, 5, off4, s.ly);
            st_row(S.pid, ld, 6, off4, s.irx); st_row(S.pid, ld, 7, off4, s.iry); st_row(S.pid, ld, 8, off4, s.irz);
        }
    }

    // ---- S physics sub-steps, state in registers ---------------------------------------------------
    const uint32_t flags = EXT ? C.physics_flags : 0u;
    const float cur_sum = ((rpm[0] + rpm[1]) + rpm[2]) + rpm[3];
    float drag_sum = cur_sum;
    if (EXT && (flags & GPD_PHYS_DRAG)) {
        // the first sub-step sees the PREVIOUS env step's action (BaseAviary.py:359,372)
        if (active) {
            drag_sum = ((ld_row(S.last_rpm, ld, 0, off4) + ld_row(S.last_rpm, ld, 1, off4)) +
                        ld_row(S.last_rpm, ld, 2, off4)) + ld_row(S.last_rpm, ld, 3, off4);
        }
    }
    float avx = 0.0f, avy = 0.0f, avz = 0.0f;
    for (int s = 0; s < C.substeps; ++s) {
        float dw = 0.0f;
        if (EXT && MULTI && (flags & GPD_PHYS_DW)) {
            // every drone sees the same pre-sub-step snapshot of its aviary (BaseAviary.py:346-347,798)
            __syncthreads();
            sh_pos[tid] = k.px; sh_pos[kBlock + tid] = k.py; sh_pos[2 * kBlock + tid] = k.pz;
            __syncthreads();
            const int base = le * D;
            for (int j = 0; j < D; ++j) {
                const float dz = sh_pos[2 * kBlock + base + j] - k.pz;
                const float ddx = sh_pos[base + j] - k.px, ddy = sh_pos[kBlock + base + j] - k.py;
                const float dxy2 = ddx * ddx + ddy * ddy;
                if (dz > 0.0f && dxy2 < 100.0f) {            // dz > 0 and dxy < 10 m
                    const float ratio = (0.25f * P.prop_radius) * fast_rcp(dz);
                    const float alpha = P.dw_coeff[0] * (ratio * ratio);
                    const float beta = P.dw_coeff[1] * dz + P.dw_coeff[2];
                    const float ib = fast_rcp(beta);
                    dw += -alpha * expf(-0.5f * (dxy2 * (ib * ib)));
                }
            }
        }
        substep<EXT>(P, C.pyb_dt, flags, rpm, drag_sum, dw, k, avx, avy, avz);
        drag_sum = cur_sum;
    }

    // ---- cache refresh: rpy of the new quaternion (BaseAviary.py:518) --------------------------------
    float roll, pitch, yaw;
    quat_to_rpy(k.qx, k.qy, k.qz, k.qw, roll, pitch, yaw);

    // ---- task: reward / terminated / truncated ---------------------------------------------------------
    float rew = -1.0f;
    bool term = false, trunc = false;
    if (C.task != GPD_TASK_NONE) {
        float my_rew = 0.0f, my_dist = 0.0f;
        bool my_out = false;
        if (active) {
            const float ex = tgx - k.px, ey = tgy - k.py, ez = tgz - k.pz;
            my_dist = fast_sqrt(ex * ex + ey * ey + ez * ez);
            const float d2 = my_dist * my_dist;
            my_rew = fmaxf(0.0f, 2.0f - d2 * d2);
            my_out = fabsf(k.px) > C.xy_bound || fabsf(k.py) > C.xy_bound || k.pz > C.z_bound ||
                     fabsf(roll) > C.tilt_bound || fabsf(pitch) > C.tilt_bound;
        }
        if (!MULTI) {
            rew = my_rew;
            term = my_dist < C.term_dist;
            trunc = my_out || (counter > C.trunc_counter);   // tested BEFORE the increment (App. B.7)
        } else {
            __syncthreads();
            sh_red[tid] = my_rew; sh_red[kBlock + tid] = my_dist; sh_red[2 * kBlock + tid] = my_out ? 1.0f : 0.0f;
            if (active && d == 0) sh_flag[tid] = counter;
            __syncthreads();
            const int base = le * D;
            float r = 0.0f, dsum = 0.0f, o = 0.0f;
            for (int j = 0; j < D; ++j) {                      // sequential, like the reference's loops
                r += sh_red[base + j]; dsum += sh_red[kBlock + base + j]; o += sh_red[2 * kBlock + base + j];
            }
            counter = sh_flag[base];
            rew = r;
            term = dsum < C.term_dist;
            trunc = (o > 0.0f) || (counter > C.trunc_counter);
        }
    } else if (MULTI) {
        __syncthreads();
        if (active && d == 0) sh_flag[tid] = counter;
        __syncthreads();
        counter = sh_flag[le * D];
    }
    if (!active) return;

    const bool done = term || trunc;
    const bool do_reset = C.auto_reset && done;
    if (d == 0) {
        reward[env] = rew;
        terminated[env] = term ? 1 : 0;
        truncated[env] = trunc ? 1 : 0;
        S.step_counter[env] = do_reset ? 0 : counter + C.substeps;
    }

    // ---- store state / observation -----------------------------------------------------------------------
    float l0 = rpm[0], l1 = rpm[1], l2 = rpm[2], l3 = rpm[3];
    if (do_reset) {
        if (term_obs12) store_obs12(term_obs12, n, k.px, k.py, k.pz, roll, pitch, yaw, k.vx, k.vy, k.vz, avx, avy, avz);
        const float* ip = reinterpret_cast<const float*>(reinterpret_cast<const char*>(init_pose) +
                                                         (C.init_per_env ? n * 28u : static_cast<uint32_t>(d) * 28u));
        k = Kin{ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], ip[6], 0, 0, 0, 0, 0, 0};
        quat_to_rpy(k.qx, k.qy, k.qz, k.qw, roll, pitch, yaw);
        avx = avy = avz = 0.0f;
        l0 = l1 = l2 = l3 = 0.0f;                              // last_clipped_action zeroed (BaseAviary.py:468)
    }
    st_row(S.kin, ld, 0, off4, k.px); st_row(S.kin, ld, 1, off4, k.py); st_row(S.kin, ld, 2, off4, k.pz);
    st_row(S.kin, ld, 3, off4, k.qx); st_row(S.kin, ld, 4, off4, k.qy); st_row(S.kin, ld, 5, off4, k.qz);
    st_row(S.kin, ld, 6, off4, k.qw);
    st_row(S.kin, ld, 7, off4, k.vx); st_row(S.kin, ld, 8, off4, k.vy); st_row(S.kin, ld, 9, off4, k.vz);
    st_row(S.kin, ld, 10, off4, k.wx); st_row(S.kin, ld, 11, off4, k.wy); st_row(S.kin, ld, 12, off4, k.wz);
    if (S.last_rpm) {
        st_row(S.last_rpm, ld, 0, off4, l0); st_row(S.last_rpm, ld, 1, off4, l1);
        st_row(S.last_rpm, ld, 2, off4, l2); st_row(S.last_rpm, ld, 3, off4, l3);
    }
    store_obs12(obs12, n, k.px, k.py, k.pz, roll, pitch, yaw, k.vx, k.vy, k.vz, avx, avy, avz);
}

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

template <bool PID, bool EXT>
hipError_t launch_step(bool multi, dim3 grid, hipStream_t st, const GpdParams& P, const GpdState& S,
                       const GpdStepCfg& C, const float* action, const float* target_pos, const float* init_pose,
                       float* obs12, float* reward, uint8_t* terminated, uint8_t* truncated, float* term_obs12) {
    if (multi) {
        hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, true>), grid, dim3(kBlock), 0, st, P, S, C, action, target_pos,
                           init_pose, obs12, reward, terminated, truncated, term_obs12);
    } else {
        hipLaunchKernelGGL((gpd_step_kernel<PID, EXT, false>), grid, dim3(kBlock), 0, st, P, S, C, action, target_pos,
                           init_pose, obs12, reward, terminated, truncated, term_obs12);
    }
    return hipGetLastError();
}

}  // namespace

// ==================================================================================================
// C ABI
// ==================================================================================================
extern "C" {

int gpd_abi_version(void) { return GPD_ABI_VERSION; }

const char* gpd_last_error(void) { return g_last_error.c_str(); }

void gpd_struct_sizes(int32_t out[3]) {
    out[0] = static_cast<int32_t>(sizeof(GpdParams));
    out[1] = static_cast<int32_t>(sizeof(GpdState));
    out[2] = static_cast<int32_t>(sizeof(GpdStepCfg));
}

int gpd_step(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const float* action,
             const float* target_pos, const float* init_pose, float* obs12, float* reward, uint8_t* terminated,
             uint8_t* truncated, float* term_obs12, void* stream) {
    if (!params || !state || !cfg) return fail(GPD_EINVAL, "gpd_step: NULL params/state/cfg");
    if (!state->kin || !state->step_counter) return fail(GPD_EINVAL, "gpd_step: NULL state.kin/step_counter");
    if (!action || !obs12 || !reward || !terminated || !truncated)
        return fail(GPD_EINVAL, "gpd_step: NULL action/obs12/reward/terminated/truncated");
    if (cfg->num_envs <= 0 || cfg->drones_per_env <= 0 || cfg->substeps <= 0)
        return fail(GPD_EINVAL, "gpd_step: num_envs, drones_per_env and substeps must be positive");
    if (cfg->drones_per_env > kBlock) return fail(GPD_ERANGE, "gpd_step: drones_per_env > 256 is not supported");
    if (cfg->act_type < GPD_ACT_RPM || cfg->act_type > GPD_ACT_DIRECT_RPM)
        return fail(GPD_EINVAL, "gpd_step: unknown act_type");
    if (cfg->task < GPD_TASK_NONE || cfg->task > GPD_TASK_MULTIHOVER) return fail(GPD_EINVAL, "gpd_step: unknown task");
    if (cfg->physics_flags & ~7u) return fail(GPD_EINVAL, "gpd_step: unknown physics flag");
    const int64_t N = static_cast<int64_t>(cfg->num_envs) * cfg->drones_per_env;
    if (state->ld < N) return fail(GPD_EINVAL, "gpd_step: state.ld < num_envs*drones_per_env");
    if (N > (1LL << 26)) return fail(GPD_ERANGE, "gpd_step: more than 2^26 drones per launch (32-bit byte offsets)");
    const bool pid = cfg->act_type == GPD_ACT_PID || cfg->act_type == GPD_ACT_VEL || cfg->act_type == GPD_ACT_ONE_D_PID;
    if (pid && !state->pid) return fail(GPD_EINVAL, "gpd_step: PID action type needs state.pid");
    if (pid && params->pid_kf <= 0.0f)
        return fail(GPD_ENOTSUP, "gpd_step: no DSLPID controller for this airframe (CF2X/CF2P only)");
    if ((cfg->physics_flags & GPD_PHYS_DRAG) && !state->last_rpm)
        return fail(GPD_EINVAL, "gpd_step: GPD_PHYS_DRAG needs state.last_rpm");
    if (cfg->task != GPD_TASK_NONE && !target_pos) return fail(GPD_EINVAL, "gpd_step: task needs target_pos");
    if (cfg->auto_reset && !init_pose) return fail(GPD_EINVAL, "gpd_step: auto_reset needs init_pose");
    const bool multi = cfg->drones_per_env > 1;
    GpdStepCfg c = *cfg;
    if (c.lanes_per_wave == 0) c.lanes_per_wave = 64;
    if (c.lanes_per_wave != 16 && c.lanes_per_wave != 32 && c.lanes_per_wave != 64)
        return fail(GPD_EINVAL, "gpd_step: lanes_per_wave must be 0, 16, 32 or 64");
    const int lanes = multi ? (kBlock / cfg->drones_per_env) * cfg->drones_per_env : (kBlock / 64) * c.lanes_per_wave;
    const int64_t blocks = (N + lanes - 1) / lanes;
    if (blocks > 0x7fffffffLL) return fail(GPD_ERANGE, "gpd_step: too many drones for one launch");
    const dim3 grid(static_cast<unsigned>(blocks));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ext = cfg->physics_flags != 0;
    hipError_t e;
    if (pid) {
        e = ext ? launch_step<true, true>(multi, grid, st, *params, *state, c, action, target_pos, init_pose, obs12,
                                          reward, terminated, truncated, term_obs12)
                : launch_step<true, false>(multi, grid, st, *params, *state, c, action, target_pos, init_pose,
                                           obs12, reward, terminated, truncated, term_obs12);
    } else {
        e = ext ? launch_step<false, true>(multi, grid, st, *params, *state, c, action, target_pos, init_pose,
                                           obs12, reward, terminated, truncated, term_obs12)
                : launch_step<false, false>(multi, grid, st, *params, *state, c, action, target_pos, init_pose,
                                            obs12, reward, terminated, truncated, term_obs12);
    }
    if (e != hipSuccess) return hip_fail(e, "gpd_step launch");
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

}  // extern "C"
