/*
 * gpd_oracle.c — the oracle's arithmetic in plain C, float64.  TEST INFRASTRUCTURE (see
 * oracle/__init__.py): linked/called only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, never by the product path.
 *
 * Same formulas and ordering as oracle/aviary_oracle.py (pinned against the reference's own Python,
 * tests/test_oracle_golden.py) and oracle/batched_oracle.py; checked against the latter in
 * tests/test_oracle_c.py.  Exists so that parity runs at the full BASELINE sizes (65 536 drones x
 * 1920 physics steps) finish in seconds.  Reference lines (gym_pybullet_drones/...):
 *   step ordering            envs/BaseAviary.py:341-383
 *   _dynamics / _integrateQ  envs/BaseAviary.py:815-892
 *   ground effect/drag/downwash  envs/BaseAviary.py:715-811 inside the explicit integrator
 *   Bullet quaternion utils  SURVEY App. C (btMatrix3x3::setRotation, pybullet_getEulerFromQuaternion)
 *   DSLPID                   control/DSLPIDControl.py:82-259
 *   action mapping           envs/BaseRLAviary.py:187-239, envs/CtrlAviary.py:140
 *   tasks                    envs/HoverAviary.py:68-117, envs/MultiHoverAviary.py:75-130
 * Arrays are row-major [N][k] doubles (N = E*D, drone n = env*D + d).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct OrcParams {
    int32_t drone_model;              /* 0 cf2x, 1 cf2p, 2 racer */
    int32_t pad_;
    double M, L, KF, KM, GRAVITY;
    double J[3], J_INV[3];
    double prop_x[4], prop_y[4];
    double gnd_eff_coeff, prop_radius, gnd_eff_h_clip;
    double drag_coeff[3], dw_coeff[3];
    double hover_rpm, max_rpm;
    double pid_gravity, pid_kf;
    double p_for[3], i_for[3], d_for[3], p_tor[3], i_tor[3], d_tor[3];
    double mixer[12];
    double pwm2rpm_scale, pwm2rpm_const, min_pwm, max_pwm;
    double speed_limit;
    double ground_z;           /* PHYS_GROUND: COLLISION_H/2 - COLLISION_Z_OFFSET */
} OrcParams;

typedef struct OrcCfg {
    int32_t num_envs, drones_per_env, act_type, substeps;
    uint32_t physics_flags;
    int32_t task, pyb_freq, auto_reset;
    double pyb_dt, ctrl_dt, xy_bound, z_bound, tilt_bound, term_dist, episode_len_sec;
} OrcCfg;

enum { ACT_RPM = 0, ACT_PID = 1, ACT_VEL = 2, ACT_ONE_D_RPM = 3, ACT_ONE_D_PID = 4, ACT_RAW_RPM = 5, ACT_DIRECT_RPM = 6 };
/* PHYS_GROUND: EXTENSION, not in the reference's Physics.DYN (see oracle/aviary_oracle.py, include/gpd.h) */
/* PHYS_DAMP: EXTENSION as well -- Bullet's default multibody damping, d = 0.04 (btMultiBody.cpp, "adding damping terms (only)":
 * dv/dt -= d (1 + |v|) v, dw/dt -= d (1 + |w|) w; third-party origin and derivation in oracle/aviary_oracle.py) */
enum { PHYS_GND = 1, PHYS_DRAG = 2, PHYS_DW = 4, PHYS_GROUND = 8, PHYS_DAMP = 16 };
#define BULLET_DAMPING 0.04

static double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void quat_to_mat(const double* q, double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
    const double xs = x * s, ys = y * s, zs = z * s;
    const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs;
    const double yy = y * ys, yz = y * zs, zz = z * zs;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
    R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
}

static void quat_to_rpy(const double* q, double* rpy) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double sarg = -2.0 * (x * z - w * y);
    if (sarg <= -0.99999) { rpy[0] = 0; rpy[1] = -0.5 * M_PI; rpy[2] = 2.0 * atan2(x, -y); }
    else if (sarg >= 0.99999) { rpy[0] = 0; rpy[1] = 0.5 * M_PI; rpy[2] = 2.0 * atan2(-x, y); }
    else {
        rpy[0] = atan2(2.0 * (y * z + w * x), w * w - x * x - y * y + z * z);
        rpy[1] = asin(sarg);
        rpy[2] = atan2(2.0 * (x * y + w * z), w * w + x * x - y * y - z * z);
    }
}

/* DSLPIDControl.computeControl; pid = integral_pos_e[3] | last_rpy[3] | integral_rpy_e[3] */
static void dslpid(const OrcParams* P, double dt, const double* pos, const double* quat, const double* vel,
                   const double* tpos, double tyaw, const double* tvel, const double* trates, double* pid,
                   double* rpm, double* pos_e, double* yaw_e) {
    double R[9], rpy[3];
    quat_to_mat(quat, R);
    quat_to_rpy(quat, rpy);
    double ep[3], ev[3], f[3];
    for (int k = 0; k < 3; ++k) {
        ep[k] = tpos[k] - pos[k];
        ev[k] = tvel[k] - vel[k];
        pid[k] = clip(pid[k] + ep[k] * dt, -2.0, 2.0);
    }
    pid[2] = clip(pid[2], -0.15, 0.15);
    for (int k = 0; k < 3; ++k) f[k] = P->p_for[k] * ep[k] + P->i_for[k] * pid[k] + P->d_for[k] * ev[k];
    f[2] += P->pid_gravity;
    double along = f[0] * R[2] + f[1] * R[5] + f[2] * R[8];
    if (!(along > 0.0)) along = 0.0;
    const double base_pwm = (sqrt(along / (4.0 * P->pid_kf)) - P->pwm2rpm_const) / P->pwm2rpm_scale;
    const double fn = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    const double zb[3] = {f[0] / fn, f[1] / fn, f[2] / fn};
    const double hd[3] = {cos(tyaw), sin(tyaw), 0.0};
    double yb[3] = {zb[1] * hd[2] - zb[2] * hd[1], zb[2] * hd[0] - zb[0] * hd[2], zb[0] * hd[1] - zb[1] * hd[0]};
    const double yn = sqrt(yb[0] * yb[0] + yb[1] * yb[1] + yb[2] * yb[2]);
    yb[0] /= yn; yb[1] /= yn; yb[2] /= yn;
    const double xb[3] = {yb[1] * zb[2] - yb[2] * zb[1], yb[2] * zb[0] - yb[0] * zb[2], yb[0] * zb[1] - yb[1] * zb[0]};
    /* M = Rd^T R, Rd columns xb yb zb; e_R = vee(M - M^T) */
#define COLDOT(a, j) ((a)[0] * R[(j)] + (a)[1] * R[3 + (j)] + (a)[2] * R[6 + (j)])
    const double eR[3] = {COLDOT(zb, 1) - COLDOT(yb, 2), COLDOT(xb, 2) - COLDOT(zb, 0), COLDOT(yb, 0) - COLDOT(xb, 1)};
#undef COLDOT
    double tau[3];
    for (int k = 0; k < 3; ++k) {
        const double ew = trates[k] - (rpy[k] - pid[3 + k]) / dt;
        pid[3 + k] = rpy[k];
        pid[6 + k] = clip(pid[6 + k] - eR[k] * dt, -1500.0, 1500.0);
        if (k < 2) pid[6 + k] = clip(pid[6 + k], -1.0, 1.0);
        tau[k] = clip(-P->p_tor[k] * eR[k] + P->d_tor[k] * ew + P->i_tor[k] * pid[6 + k], -3200.0, 3200.0);
    }
    for (int m = 0; m < 4; ++m) {
        const double pwm = clip(base_pwm + P->mixer[3 * m] * tau[0] + P->mixer[3 * m + 1] * tau[1] + P->mixer[3 * m + 2] * tau[2],
                                P->min_pwm, P->max_pwm);
        rpm[m] = P->pwm2rpm_scale * pwm + P->pwm2rpm_const;
    }
    if (pos_e) { pos_e[0] = ep[0]; pos_e[1] = ep[1]; pos_e[2] = ep[2]; }
    if (yaw_e) *yaw_e = atan2(-yb[0], xb[0]) - rpy[2];
}

/* one physics sub-step of one drone; snap = pre-sub-step positions of its aviary [D][3] */
static void substep(const OrcParams* P, const OrcCfg* C, const double* rpm, const double* drag_rpm, const double* snap,
                    int d, double* pos, double* quat, double* vel, double* w, double* ang_v) {
    const double h = C->pyb_dt;
    double R[9];
    quat_to_mat(quat, R);
    double sq[4], f[4];
    for (int i = 0; i < 4; ++i) { sq[i] = rpm[i] * rpm[i]; f[i] = sq[i] * P->KF; }
    if (C->physics_flags & PHYS_GND) {
        double rpy[3];
        quat_to_rpy(quat, rpy);
        if (fabs(rpy[0]) < M_PI / 2 && fabs(rpy[1]) < M_PI / 2) {
            for (int i = 0; i < 4; ++i) {
                double hz = pos[2] + R[6] * P->prop_x[i] + R[7] * P->prop_y[i];
                if (hz < P->gnd_eff_h_clip) hz = P->gnd_eff_h_clip;
                const double ratio = P->prop_radius / (4.0 * hz);
                f[i] += sq[i] * P->KF * P->gnd_eff_coeff * ratio * ratio;
            }
        }
    }
    double fz = f[0] + f[1] + f[2] + f[3];
    if (C->physics_flags & PHYS_DW) {
        for (int j = 0; j < C->drones_per_env; ++j) {
            const double dz = snap[3 * j + 2] - snap[3 * d + 2];
            const double dx = snap[3 * j] - snap[3 * d], dy = snap[3 * j + 1] - snap[3 * d + 1];
            const double dxy = sqrt(dx * dx + dy * dy);
            if (dz > 0 && dxy < 10) {
                const double ratio = P->prop_radius / (4.0 * dz);
                const double alpha = P->dw_coeff[0] * ratio * ratio, beta = P->dw_coeff[1] * dz + P->dw_coeff[2];
                fz += -alpha * exp(-0.5 * (dxy / beta) * (dxy / beta));
            }
        }
    }
    double F[3] = {R[2] * fz, R[5] * fz, R[8] * fz - P->GRAVITY};
    if (C->physics_flags & PHYS_DRAG) {
        const double ws = 2.0 * M_PI * (drag_rpm[0] + drag_rpm[1] + drag_rpm[2] + drag_rpm[3]) / 60.0;
        for (int k = 0; k < 3; ++k) F[k] -= P->drag_coeff[k] * vel[k] * ws;
    }
    const double sgn = P->drone_model == 2 ? -1.0 : 1.0;
    const double tz = sgn * P->KM * (-sq[0] + sq[1] - sq[2] + sq[3]);
    double tx, ty;
    if (P->drone_model == 1) { tx = (f[1] - f[3]) * P->L; ty = (-f[0] + f[2]) * P->L; }
    else {
        const double arm = P->L / sqrt(2.0);
        tx = (f[0] + f[1] - f[2] - f[3]) * arm;
        ty = (-f[0] + f[1] + f[2] - f[3]) * arm;
        if (P->drone_model == 0) tx = -tx;
    }
    const double jw[3] = {P->J[0] * w[0], P->J[1] * w[1], P->J[2] * w[2]};
    const double tau[3] = {tx - (w[1] * jw[2] - w[2] * jw[1]), ty - (w[2] * jw[0] - w[0] * jw[2]), tz - (w[0] * jw[1] - w[1] * jw[0])};
    double dl = 0, da = 0;                        /* damping rates d (1 + |v|), d (1 + |w|) of the velocities BEFORE the update */
    if (C->physics_flags & PHYS_DAMP) {
        dl = BULLET_DAMPING * (1.0 + sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]));
        da = BULLET_DAMPING * (1.0 + sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
    }
    for (int k = 0; k < 3; ++k) {
        vel[k] += h * (F[k] / P->M - dl * vel[k]);
        w[k] += h * (P->J_INV[k] * tau[k] - da * w[k]);
        pos[k] += h * vel[k];
    }
    if ((C->physics_flags & PHYS_GROUND) && (pos[2] < P->ground_z || (pos[2] <= P->ground_z && vel[2] < 0))) {   /* the plane at z = 0 (second clause: the tie) */
        pos[2] = P->ground_z;
        vel[0] = 0; vel[1] = 0;
        if (vel[2] < 0) vel[2] = 0;
    }
    const double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (!(fabs(n) <= 1e-8)) {                     /* !np.isclose(n, 0) */
        const double th = n * h / 2, cs = cos(th), sc = sin(th) / n;
        const double x = quat[0], y = quat[1], z = quat[2], qw = quat[3];
        const double l[4] = {w[2] * y - w[1] * z + w[0] * qw, -w[2] * x + w[0] * z + w[1] * qw,
                             w[1] * x - w[0] * y + w[2] * qw, -w[0] * x - w[1] * y - w[2] * z};
        for (int k = 0; k < 4; ++k) quat[k] = cs * quat[k] + sc * l[k];
    }
    for (int k = 0; k < 3; ++k) ang_v[k] = R[3 * k] * w[0] + R[3 * k + 1] * w[1] + R[3 * k + 2] * w[2];
}

static int g_threads = 1;
#ifdef _OPENMP
#include <omp.h>
int orc_max_threads(void) { return omp_get_max_threads(); }
#else
int orc_max_threads(void) { return 1; }
#endif
/* threads orc_step spreads the aviaries over (the timed multi-core CPU baseline of bench.py); returns the value set */
int orc_set_threads(int n) { g_threads = n < 1 ? 1 : (n > orc_max_threads() ? orc_max_threads() : n); return g_threads; }

int orc_struct_sizes(int32_t out[2]) { out[0] = (int32_t)sizeof(OrcParams); out[1] = (int32_t)sizeof(OrcCfg); return 0; }

/*
 * One env.step() of every aviary (+ same-step auto reset).  All state arrays are updated in place.
 *   pos [N][3] quat [N][4] vel [N][3] rates [N][3] ang_v [N][3] rpy [N][3] last_rpm [N][4] pid [N][9] (may be NULL)
 *   counter [E] int64; action [N][A]; target [N][3]; init_pos [N][3], init_quat [N][4] (auto_reset only)
 *   out: obs12 [N][12], reward [E], terminated [E], truncated [E], term_obs12 [N][12] (may be NULL)
 */
int orc_step(const OrcParams* P, const OrcCfg* C, double* pos, double* quat, double* vel, double* rates, double* ang_v,
             double* rpy, double* last_rpm, double* pid, int64_t* counter, const double* action, const double* target,
             const double* init_pos, const double* init_quat, double* obs12, double* reward, uint8_t* terminated,
             uint8_t* truncated, double* term_obs12) {
    const int E = C->num_envs, D = C->drones_per_env, S = C->substeps;
    static const int ADIM[7] = {4, 3, 4, 1, 1, 4, 4};
    const int A = ADIM[C->act_type];
    if (D > 256) return -2;
    const double zero3[3] = {0, 0, 0};
    /* aviaries are independent: one per iteration, any number of threads (orc_set_threads; 1 by default) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads)
#endif
    for (int e = 0; e < E; ++e) {
        double snap[3 * 256], rpm_env[4 * 256];
        /* ---- action -> RPM from the cached state */
        for (int d = 0; d < D; ++d) {
            const int n = e * D + d;
            const double* a = action + (size_t)n * A;
            double* rpm = rpm_env + 4 * d;
            switch (C->act_type) {
                case ACT_RPM: for (int k = 0; k < 4; ++k) rpm[k] = P->hover_rpm * (1 + 0.05 * a[k]); break;
                case ACT_ONE_D_RPM: for (int k = 0; k < 4; ++k) rpm[k] = P->hover_rpm * (1 + 0.05 * a[0]); break;
                case ACT_RAW_RPM: for (int k = 0; k < 4; ++k) rpm[k] = clip(a[k], 0, P->max_rpm); break;
                case ACT_DIRECT_RPM: for (int k = 0; k < 4; ++k) rpm[k] = a[k]; break;
                default: {
                    double tpos[3] = {pos[3 * n], pos[3 * n + 1], pos[3 * n + 2]}, tvel[3] = {0, 0, 0}, tyaw = 0;
                    if (C->act_type == ACT_PID) {
                        const double dd[3] = {a[0] - pos[3 * n], a[1] - pos[3 * n + 1], a[2] - pos[3 * n + 2]};
                        const double dist = sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
                        for (int k = 0; k < 3; ++k) tpos[k] = dist <= 1 ? a[k] : pos[3 * n + k] + dd[k] / dist;
                    } else if (C->act_type == ACT_VEL) {
                        const double nn = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
                        if (nn != 0) for (int k = 0; k < 3; ++k) tvel[k] = P->speed_limit * fabs(a[3]) * (a[k] / nn);
                        tyaw = rpy[3 * n + 2];
                    } else {
                        tpos[2] += 0.1 * a[0];
                    }
                    dslpid(P, C->ctrl_dt, pos + 3 * n, quat + 4 * n, vel + 3 * n, tpos, tyaw, tvel, zero3, pid + 9 * n, rpm, 0, 0);
                }
            }
        }
        /* ---- S sub-steps; every drone sees the same pre-sub-step snapshot of the aviary */
        for (int s = 0; s < S; ++s) {
            memcpy(snap, pos + (size_t)3 * e * D, sizeof(double) * 3 * D);
            for (int d = 0; d < D; ++d) {
                const int n = e * D + d;
                substep(P, C, rpm_env + 4 * d, s == 0 ? last_rpm + 4 * n : rpm_env + 4 * d, snap, d, pos + 3 * n,
                        quat + 4 * n, vel + 3 * n, rates + 3 * n, ang_v + 3 * n);
            }
        }
        double rew = 0, dsum = 0;
        int out = 0;
        for (int d = 0; d < D; ++d) {
            const int n = e * D + d;
            memcpy(last_rpm + 4 * n, rpm_env + 4 * d, sizeof(double) * 4);
            quat_to_rpy(quat + 4 * n, rpy + 3 * n);
            double* o = obs12 + (size_t)12 * n;
            for (int k = 0; k < 3; ++k) { o[k] = pos[3 * n + k]; o[3 + k] = rpy[3 * n + k]; o[6 + k] = vel[3 * n + k]; o[9 + k] = ang_v[3 * n + k]; }
            if (C->task != 0) {
                const double ex = target[3 * n] - pos[3 * n], ey = target[3 * n + 1] - pos[3 * n + 1], ez = target[3 * n + 2] - pos[3 * n + 2];
                const double dist = sqrt(ex * ex + ey * ey + ez * ez);
                const double r = 2 - pow(dist, 4);
                rew += r > 0 ? r : 0;
                dsum += dist;
                if (fabs(pos[3 * n]) > C->xy_bound || fabs(pos[3 * n + 1]) > C->xy_bound || pos[3 * n + 2] > C->z_bound ||
                    fabs(rpy[3 * n]) > C->tilt_bound || fabs(rpy[3 * n + 1]) > C->tilt_bound) out = 1;
            }
        }
        int term = 0, trunc = 0;
        if (C->task != 0) {
            term = dsum < C->term_dist;
            trunc = out || ((double)counter[e] / C->pyb_freq > C->episode_len_sec);
            reward[e] = rew;
        } else {
            reward[e] = -1;
        }
        terminated[e] = (uint8_t)term; truncated[e] = (uint8_t)trunc;
        counter[e] += S;
        if (C->auto_reset && (term || trunc)) {
            for (int d = 0; d < D; ++d) {
                const int n = e * D + d;
                if (term_obs12) memcpy(term_obs12 + (size_t)12 * n, obs12 + (size_t)12 * n, sizeof(double) * 12);
                memcpy(pos + 3 * n, init_pos + 3 * n, sizeof(double) * 3);
                memcpy(quat + 4 * n, init_quat + 4 * n, sizeof(double) * 4);
                memset(vel + 3 * n, 0, sizeof(double) * 3); memset(rates + 3 * n, 0, sizeof(double) * 3);
                memset(ang_v + 3 * n, 0, sizeof(double) * 3); memset(last_rpm + 4 * n, 0, sizeof(double) * 4);
                quat_to_rpy(quat + 4 * n, rpy + 3 * n);
                double* o = obs12 + (size_t)12 * n;
                for (int k = 0; k < 3; ++k) { o[k] = pos[3 * n + k]; o[3 + k] = rpy[3 * n + k]; o[6 + k] = 0; o[9 + k] = 0; }
            }
            counter[e] = 0;
        }
    }
    return 0;
}

/* BaseAviary._downwash (envs/BaseAviary.py:785-811) for ONE aviary of n drones, any n: the reference's all-pairs loop, body-z
 * force of every drone from the drones above it.  pos [n][3], out [n].  (The per-aviary path above stops at 256 drones.) */
int orc_downwash_all_pairs(const OrcParams* P, int n, const double* pos, double* out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads)
#endif
    for (int i = 0; i < n; ++i) {
        double f = 0;
        for (int j = 0; j < n; ++j) {
            const double dz = pos[3 * j + 2] - pos[3 * i + 2];
            const double dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1];
            const double dxy = sqrt(dx * dx + dy * dy);
            if (dz > 0 && dxy < 10) {
                const double ratio = P->prop_radius / (4.0 * dz);
                const double alpha = P->dw_coeff[0] * ratio * ratio, beta = P->dw_coeff[1] * dz + P->dw_coeff[2];
                f += -alpha * exp(-0.5 * (dxy / beta) * (dxy / beta));
            }
        }
        out[i] = f;
    }
    return 0;
}

/* The same loop for m of the n drones only (recv [m]: their indices; out [m]): every source, a sample of the receivers --
 * what a bounded check of a world of 10^6 drones runs (the full loop is 10^12 pair tests). */
int orc_downwash_some(const OrcParams* P, int n, const double* pos, int m, const int* recv, double* out) {
    for (int k = 0; k < m; ++k) if (recv[k] < 0 || recv[k] >= n) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads)
#endif
    for (int k = 0; k < m; ++k) {
        const int i = recv[k];
        double f = 0;
        for (int j = 0; j < n; ++j) {
            const double dz = pos[3 * j + 2] - pos[3 * i + 2];
            const double dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1];
            const double dxy = sqrt(dx * dx + dy * dy);
            if (dz > 0 && dxy < 10) {
                const double ratio = P->prop_radius / (4.0 * dz);
                const double alpha = P->dw_coeff[0] * ratio * ratio, beta = P->dw_coeff[1] * dz + P->dw_coeff[2];
                f += -alpha * exp(-0.5 * (dxy / beta) * (dxy / beta));
            }
        }
        out[k] = f;
    }
    return 0;
}

/* standalone batched DSLPIDControl.computeControl: arrays [n][3]/[n][4], pid [n][9] */
int orc_pid(const OrcParams* P, double dt, int n, const double* pos, const double* quat, const double* vel,
            const double* tpos, const double* trpy, const double* tvel, const double* trates, double* pid, double* rpm,
            double* pos_e, double* yaw_e) {
    for (int i = 0; i < n; ++i)
        dslpid(P, dt, pos + 3 * i, quat + 4 * i, vel + 3 * i, tpos + 3 * i, trpy[3 * i + 2], tvel + 3 * i, trates + 3 * i,
               pid + 9 * i, rpm + 4 * i, pos_e + 3 * i, yaw_e + i);
    return 0;
}
