/*
 * A host that is NOT Python: plain C11 over include/gpd.h and the HIP runtime's C API.  It owns the device buffers (hipMalloc),
 * resets E aviaries of D drones, runs K env steps through gpd_step -- one call per step, the shape of the reference's loop
 * `obs, reward, terminated, truncated, info = env.step(action)` (examples/learn.py:157-192; envs/BaseAviary.py:341-383) -- and
 * then the same K steps again in ONE gpd_rollout call, and writes what it read back.  tests/test_gpu_c_host.py builds it,
 * feeds it the bytes of the structs the Python mirror would pass, and checks the result against that mirror (bit for bit:
 * same library) and against the float64 oracle.  Test infrastructure: it exists to show that nothing in the boundary needs
 * Python, ctypes or torch.
 *
 * input  (binary, native endianness): int32 E, D, A, K, init_rows, target_rows, has_last_rpm, has_pid | GpdParams | GpdStepCfg |
 *         init_pose [init_rows][7] f32 | target_pos [target_rows][3] f32 | actions [K][E*D][A] f32   (rows: D or E*D, as the
 *         configuration's init_per_env / target_per_env say)
 * output: kin [13 * N] f32 (GpdState.kin's four planes, ld = N) | obs12 [N][12] f32 | reward [E] f32 | terminated [E] u8 | truncated [E] u8 | step_counter [E] i32
 *         -- first after the K gpd_step calls, then the same block after the single gpd_rollout call from the same start
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "gpd.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define GPD(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, gpd_last_error()); return 4; } } while (0)

static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : 1; }

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: hover_host <in> <out>\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t h[8];
    GpdParams P;
    GpdStepCfg C;
    if (rd(f, h, sizeof h) || rd(f, &P, sizeof P) || rd(f, &C, sizeof C)) { fprintf(stderr, "short input\n"); return 2; }
    const int E = h[0], D = h[1], A = h[2], K = h[3], has_last = h[6], has_pid = h[7];
    const size_t N = (size_t)E * D, n_init = (size_t)h[4] * 7, n_tgt = (size_t)h[5] * 3, n_act = (size_t)K * N * A;
    if (h[4] != (C.init_per_env ? E * D : D) || h[5] != (C.target_per_env ? E * D : D) || C.num_envs != E || C.drones_per_env != D) {
        fprintf(stderr, "header and GpdStepCfg disagree\n");
        return 2;
    }
    float* init_h = malloc(n_init * 4);
    float* tgt_h = malloc(n_tgt * 4);
    float* act_h = malloc(n_act * 4);
    if (rd(f, init_h, n_init * 4) || rd(f, tgt_h, n_tgt * 4) || rd(f, act_h, n_act * 4)) { fprintf(stderr, "short input\n"); return 2; }
    fclose(f);
    if (gpd_abi_version() != GPD_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", gpd_abi_version(), GPD_ABI_VERSION); return 2; }

    hipStream_t st;
    HIP(hipStreamCreate(&st));
    GpdState S;
    memset(&S, 0, sizeof S);
    S.ld = (int64_t)N;
    float *init_d, *tgt_d, *act_d, *obs_d, *obsK_d, *rew_d, *rewK_d;
    uint8_t *term_d, *trunc_d, *termK_d, *truncK_d;
    HIP(hipMalloc((void**)&S.kin, 13 * N * 4));
    if (has_last) HIP(hipMalloc((void**)&S.last_rpm, 4 * N * 4));
    if (has_pid) HIP(hipMalloc((void**)&S.pid, 9 * N * 4));
    HIP(hipMalloc((void**)&S.step_counter, (size_t)E * 4));
    HIP(hipMalloc((void**)&init_d, n_init * 4));
    HIP(hipMalloc((void**)&tgt_d, n_tgt * 4));
    HIP(hipMalloc((void**)&act_d, n_act * 4));
    HIP(hipMalloc((void**)&obs_d, N * 48));
    HIP(hipMalloc((void**)&obsK_d, (size_t)K * N * 48));
    HIP(hipMalloc((void**)&rew_d, (size_t)E * 4));
    HIP(hipMalloc((void**)&rewK_d, (size_t)K * E * 4));
    HIP(hipMalloc((void**)&term_d, E));
    HIP(hipMalloc((void**)&trunc_d, E));
    HIP(hipMalloc((void**)&termK_d, (size_t)K * E));
    HIP(hipMalloc((void**)&truncK_d, (size_t)K * E));
    HIP(hipMemcpyAsync(init_d, init_h, n_init * 4, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(tgt_d, tgt_h, n_tgt * 4, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(act_d, act_h, n_act * 4, hipMemcpyHostToDevice, st));

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    float* kin_h = malloc(13 * N * 4);
    float* obs_h = malloc(N * 48);
    float* rew_h = malloc((size_t)E * 4);
    uint8_t* flag_h = malloc((size_t)2 * E);
    int32_t* cnt_h = malloc((size_t)E * 4);
    for (int pass = 0; pass < 2; ++pass) {
        GPD(gpd_reset(&S, init_d, C.init_per_env, NULL, E, D, 1, obs_d, st));
        const float *obs_last = obs_d, *rew_last = rew_d;
        const uint8_t *term_last = term_d, *trunc_last = trunc_d;
        if (pass == 0) {
            for (int k = 0; k < K; ++k)      /* the reference's loop: one step per call */
                GPD(gpd_step(&P, &S, &C, act_d + (size_t)k * N * A, tgt_d, init_d, obs_d, rew_d, term_d, trunc_d, NULL, st));
        } else {                             /* the same K steps in one launch, every step's rows kept */
            GPD(gpd_rollout(&P, &S, &C, K, act_d, (int64_t)(N * A), tgt_d, init_d, obsK_d, (int64_t)(N * 12), rewK_d, termK_d, truncK_d,
                            (int64_t)E, NULL, st));
            obs_last = obsK_d + (size_t)(K - 1) * N * 12; rew_last = rewK_d + (size_t)(K - 1) * E;
            term_last = termK_d + (size_t)(K - 1) * E; trunc_last = truncK_d + (size_t)(K - 1) * E;
        }
        HIP(hipMemcpyAsync(kin_h, S.kin, 13 * N * 4, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(obs_h, obs_last, N * 48, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(rew_h, rew_last, (size_t)E * 4, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(flag_h, term_last, E, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(flag_h + E, trunc_last, E, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(cnt_h, S.step_counter, (size_t)E * 4, hipMemcpyDeviceToHost, st));
        HIP(hipStreamSynchronize(st));
        fwrite(kin_h, 4, 13 * N, f); fwrite(obs_h, 4, N * 12, f); fwrite(rew_h, 4, E, f);
        fwrite(flag_h, 1, (size_t)2 * E, f); fwrite(cnt_h, 4, E, f);
    }
    fclose(f);
    printf("hover_host: E=%d D=%d A=%d K=%d ok\n", E, D, A, K);
    return 0;
}
