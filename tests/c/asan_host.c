/*
 * Drives the HOST side of every C-ABI entry (include/gpd.h) under AddressSanitizer + UndefinedBehaviorSanitizer, on a machine without
 * a GPU: libgpd's four units compiled host-only with the sanitizers, the HIP runtime replaced by tests/stubs/hip_stub.c (launches are
 * counted, nothing runs).  What is exercised: argument validation and error codes, the per-thread last-error string, struct plumbing,
 * and the launch arithmetic (grid / block / dynamic LDS sizes at the largest supported shapes -- where a 32-bit product overflows first).
 * Device pointers are fake non-null addresses: host code must never dereference them (ASan would say so).  SURVEY.md section 5's
 * "-fsanitize=address host build"; run by tests/test_host_sanitizers.py.  Prints one line per check; exit code = failed checks.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gpd.h"

int hipstub_launches(void);
void hipstub_last(unsigned out[7]);

static int failed;
#define CHECK(cond, what) do { if (!(cond)) { ++failed; printf("FAIL %s (line %d): %s\n", what, __LINE__, gpd_last_error()); } else printf("ok   %s\n", what); } while (0)
#define DEV(n) ((void*)(uintptr_t)(0x100000000ull + 0x1000000ull * (n)))      /* fake device addresses */

int main(void) {
    int32_t sz[3];
    gpd_struct_sizes(sz);
    CHECK(gpd_abi_version() == GPD_ABI_VERSION && sz[0] == (int)sizeof(GpdParams) && sz[1] == (int)sizeof(GpdState) && sz[2] == (int)sizeof(GpdStepCfg), "version and struct sizes");
    CHECK(gpd_sizeof_swarm() == (int)sizeof(GpdSwarm), "gpd_sizeof_swarm");

    GpdParams P;
    memset(&P, 0, sizeof P);
    P.pid_kf = 3.16e-10f;
    GpdState S;
    memset(&S, 0, sizeof S);
    GpdStepCfg C;
    memset(&C, 0, sizeof C);
    unsigned last[7];

    /* ---- error paths: codes, and a message every time ---- */
    CHECK(gpd_step(NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == GPD_EINVAL && strlen(gpd_last_error()) > 0, "gpd_step(NULL ...) -> GPD_EINVAL");
    S.kin = DEV(1); S.step_counter = DEV(2); S.ld = 65536;
    C.num_envs = 65536; C.drones_per_env = 1; C.substeps = 1; C.act_type = GPD_ACT_RPM; C.task = GPD_TASK_HOVER; C.pyb_dt = 1.0f / 240; C.ctrl_dt = 1.0f / 240;
    C.inv_ctrl_dt = 240; C.auto_reset = 1;
#define STEP(act_, tgt_) gpd_step(&P, &S, &C, act_, tgt_, DEV(5), DEV(6), DEV(7), DEV(8), DEV(9), NULL, NULL)
    CHECK(STEP(NULL, DEV(4)) == GPD_EINVAL, "NULL action");
    CHECK(STEP(DEV(3), NULL) == GPD_EINVAL, "task without target_pos");
    C.act_type = 99; CHECK(STEP(DEV(3), DEV(4)) == GPD_EINVAL, "unknown act_type"); C.act_type = GPD_ACT_RPM;
    C.physics_flags = 64; CHECK(STEP(DEV(3), DEV(4)) == GPD_EINVAL, "unknown physics flag"); C.physics_flags = 0;
    C.drones_per_env = 257; CHECK(STEP(DEV(3), DEV(4)) == GPD_ERANGE, "drones_per_env > 256"); C.drones_per_env = 1;
    S.ld = 100; CHECK(STEP(DEV(3), DEV(4)) == GPD_EINVAL, "state.ld < N"); S.ld = 65536;
    C.act_type = GPD_ACT_PID; CHECK(STEP(DEV(3), DEV(4)) == GPD_EINVAL, "PID action without state.pid"); C.act_type = GPD_ACT_RPM;
    C.physics_flags = GPD_PHYS_DRAG; CHECK(STEP(DEV(3), DEV(4)) == GPD_EINVAL, "DRAG without last_rpm"); C.physics_flags = 0;
    C.lanes_per_wave = 48; CHECK(STEP(DEV(3), DEV(4)) == GPD_EINVAL, "lanes_per_wave 48"); C.lanes_per_wave = 0;
    S.kin = (float*)((char*)DEV(1) + 4); CHECK(STEP(DEV(3), DEV(4)) == GPD_EINVAL && strstr(gpd_last_error(), "16-byte") != NULL, "state.kin at a 4-byte offset (the planes are float4)");
    CHECK(gpd_reset(&S, DEV(5), 0, NULL, 65536, 1, 0, DEV(6), NULL) == GPD_EINVAL, "gpd_reset refuses a misaligned state.kin as well");
    CHECK(gpd_state_vectors(&S, DEV(6), DEV(15), 65536, NULL) == GPD_EINVAL, "gpd_state_vectors refuses a misaligned state.kin as well"); S.kin = DEV(1);
    S.ld = 1ll << 32; CHECK(STEP(DEV(3), DEV(4)) == GPD_EINVAL && strstr(gpd_last_error(), "state.ld") != NULL, "a pitch that does not fit the kernels' 32-bit argument"); S.ld = 65536;
    C.num_envs = (1 << 26) + 1; S.ld = (1ll << 26) + 64; CHECK(STEP(DEV(3), DEV(4)) == GPD_ERANGE, "more than 2^26 drones"); C.num_envs = 65536; S.ld = 65536;

    /* ---- launch geometry ---- */
    int n0 = hipstub_launches();
    CHECK(STEP(DEV(3), DEV(4)) == 0 && hipstub_launches() == n0 + 1, "gpd_step launches once");
    CHECK(gpd_step_sync(&P, &S, &C, DEV(3), DEV(4), DEV(5), DEV(6), DEV(7), DEV(8), DEV(9), NULL, NULL) == 0 && hipstub_launches() == n0 + 2, "gpd_step_sync: one launch (and the wait for its stream)");
    CHECK(gpd_step_sync(&P, &S, &C, NULL, DEV(4), DEV(5), DEV(6), DEV(7), DEV(8), DEV(9), NULL, NULL) == GPD_EINVAL && strstr(gpd_last_error(), "gpd_step_sync") != NULL, "gpd_step_sync names itself in its errors");
    hipstub_last(last);
    CHECK(last[0] == 256 && last[3] == 256 && last[6] == 0, "65 536 single-drone aviaries: 256 workgroups of 256 lanes");
    C.num_envs = 1 << 26; S.ld = 1ll << 26;
    CHECK(STEP(DEV(3), DEV(4)) == 0, "2^26 drones in one launch");
    hipstub_last(last);
    CHECK(last[0] == (1u << 18), "2^26 drones: 2^18 workgroups");
    C.num_envs = 8192; C.drones_per_env = 8; S.ld = 65536; C.task = GPD_TASK_MULTIHOVER; C.physics_flags = 7; S.last_rpm = DEV(10);
    CHECK(STEP(DEV(3), DEV(4)) == 0, "8192 x 8 drones, every force term");
    hipstub_last(last);
    CHECK(last[0] == 256, "whole aviaries per workgroup: 32 x 8 drones each");
    C.drones_per_env = 100; C.num_envs = 1000; S.ld = 100032;
    CHECK(STEP(DEV(3), DEV(4)) == 0, "aviaries of 100 drones");
    hipstub_last(last);
    CHECK(last[0] == 500, "two 100-drone aviaries per workgroup");
    /* every action type, with and without the add-on terms */
    C.num_envs = 4096; C.drones_per_env = 1; S.ld = 4096; C.task = GPD_TASK_HOVER; S.pid = DEV(11);
    for (int act = GPD_ACT_RPM; act <= GPD_ACT_DIRECT_RPM; ++act)
        for (uint32_t fl = 0; fl <= 31; fl += 31) {
            C.act_type = act; C.physics_flags = fl;
            char what[64];
            snprintf(what, sizeof what, "gpd_step act %d flags %u", act, fl);
            CHECK(STEP(DEV(3), DEV(4)) == 0, what);
        }
    C.act_type = GPD_ACT_RPM; C.physics_flags = 0;
    /* rollouts: the store-wave kernel (terminal observations) asks for dynamic LDS, the single-drone kernel for none */
    int rc = gpd_rollout(&P, &S, &C, 64, DEV(3), 4096 * 4, DEV(4), DEV(5), DEV(6), 4096 * 12, DEV(7), DEV(8), DEV(9), 4096, NULL, NULL);
    hipstub_last(last);
    CHECK(rc == 0 && last[0] == 16 && last[3] == 256 && last[6] == 0, "gpd_rollout, 4096 aviaries: 16 workgroups, no dynamic LDS");
    rc = gpd_rollout(&P, &S, &C, 64, DEV(3), 4096 * 4, DEV(4), DEV(5), DEV(6), 4096 * 12, DEV(7), DEV(8), DEV(9), 4096, DEV(12), NULL);
    hipstub_last(last);
    CHECK(rc == 0 && last[3] == 320 && last[6] == 4 * (256 * 48 + 256 * 4 + 512), "gpd_rollout with terminal observations: 320 threads, four LDS slots");
    CHECK(gpd_rollout(&P, &S, &C, 0, DEV(3), 0, DEV(4), DEV(5), DEV(6), 0, DEV(7), DEV(8), DEV(9), 0, NULL, NULL) != 0, "gpd_rollout with K = 0 is refused");
    S.act_ring = DEV(13); S.ring_pos = DEV(14); S.hist_len = 15;
    CHECK(gpd_rollout_history(&P, &S, &C, 20, DEV(3), 4096 * 4, DEV(4), DEV(5), DEV(6), 4096 * 12, DEV(7), DEV(8), DEV(9), 4096, NULL) == 0, "gpd_rollout_history");
    CHECK(gpd_hist_rows(&S, 4096, 1, 4, DEV(6), DEV(15), NULL) == 0, "gpd_hist_rows");
    CHECK(gpd_full_obs(&S, 20, 4096, 1, 4, DEV(6), 4096 * 12, DEV(3), 4096 * 4, DEV(15), 4096 * 72, NULL) == 0, "gpd_full_obs");
    CHECK(gpd_hist_rows(&S, 4096, 1, 5, DEV(6), DEV(15), NULL) != 0, "gpd_hist_rows with act_dim 5 is refused");
    GpdPolicy pol;
    memset(&pol, 0, sizeof pol);
    pol.w1 = DEV(16); pol.b1 = DEV(17); pol.w2 = DEV(18); pol.b2 = DEV(19); pol.w3 = DEV(20); pol.b3 = DEV(21); pol.hidden = 64;
    pol.in_dim = 12 + 15 * 4;
    CHECK(gpd_rollout_policy(&P, &S, &C, &pol, 8, DEV(6), DEV(4), DEV(5), DEV(22), DEV(6), 4096 * 12, DEV(7), DEV(8), DEV(9), 4096, NULL, NULL, NULL, NULL, NULL) == 0,
          "gpd_rollout_policy, 72-float rows");
    pol.in_dim = 13;
    CHECK(gpd_rollout_policy(&P, &S, &C, &pol, 8, DEV(6), DEV(4), DEV(5), DEV(22), DEV(6), 4096 * 12, DEV(7), DEV(8), DEV(9), 4096, NULL, NULL, NULL, NULL, NULL) == GPD_ENOTSUP,
          "gpd_rollout_policy refuses in_dim 13");
    S.act_ring = NULL; S.ring_pos = NULL; S.hist_len = 0;
    CHECK(gpd_reset(&S, DEV(5), 0, NULL, 4096, 1, 1, DEV(6), NULL) == 0, "gpd_reset");
    CHECK(gpd_state_vectors(&S, DEV(6), DEV(23), 4096, NULL) == 0, "gpd_state_vectors");
    CHECK(gpd_pid(&P, DEV(11), 4096, 1.0f / 240, DEV(24), DEV(25), DEV(26), DEV(27), NULL, NULL, NULL, DEV(28), NULL, NULL, 4096, NULL) == 0, "gpd_pid");
    n0 = hipstub_launches();
    CHECK(gpd_pid_sync(&P, DEV(11), 4096, 1.0f / 240, DEV(24), DEV(25), DEV(26), DEV(27), NULL, NULL, NULL, DEV(28), NULL, NULL, 1, NULL) == 0 && hipstub_launches() == n0 + 1,
          "gpd_pid_sync: one launch (the stub never writes the completion word: the 2 ms fallback to the stream wait)");
    CHECK(gpd_pid_sync(&P, DEV(11), 4096, 1.0f / 240, DEV(24), DEV(25), DEV(26), DEV(27), NULL, NULL, NULL, DEV(28), NULL, NULL, 4096, NULL) == 0, "gpd_pid_sync, 4096 controllers (stream wait)");
    CHECK(gpd_pid_sync(&P, DEV(11), 4096, 1.0f / 240, NULL, DEV(25), DEV(26), DEV(27), NULL, NULL, NULL, DEV(28), NULL, NULL, 1, NULL) == GPD_EINVAL && strstr(gpd_last_error(), "gpd_pid_sync") != NULL, "gpd_pid_sync names itself in its errors");

    /* ---- one world ---- */
    CHECK(gpd_downwash_global(&P, DEV(1), 65536, 65536, 10.5f, -170.0f, -170.0f, 32, 32, 0.0f, 1.0f, 1, NULL, DEV(30), DEV(31), DEV(32), DEV(33), DEV(34), NULL, NULL, NULL, NULL) == 0,
          "gpd_downwash_global");
    CHECK(gpd_downwash_global(&P, DEV(1), 65536, 65536, 9.0f, -170.0f, -170.0f, 32, 32, 0.0f, 1.0f, 1, NULL, DEV(30), DEV(31), DEV(32), DEV(33), DEV(34), NULL, NULL, NULL, NULL) != 0,
          "gpd_downwash_global refuses cells under 10 m");
    GpdSwarm W;
    memset(&W, 0, sizeof W);
    W.n_rows = 65792; W.slab = 65792; W.world_size = 1; W.rank = 0; W.own_count = 65536; W.nx = W.ny = 32; W.nz = 1; W.cell = 10.5f; W.x0 = W.y0 = -170.0f; W.zbin = 1.0f;
    W.meta_rows = 256; W.pos4 = DEV(40); W.bin_pos = DEV(41); W.cell_count = DEV(42); W.cell_start = DEV(43); W.order = DEV(44); W.visit_out = DEV(45);
    W.slot_key = DEV(46); W.dw_force = DEV(47); W.slot_of = DEV(48); W.pos_sorted = DEV(49); W.pair_list = DEV(50); W.pair_nb = DEV(51); W.list_ok = DEV(52);
    W.list_cap = 48; W.list_delta = 0.245f; W.drift = DEV(53); W.total_drones = 65536; W.list_adapt = 1;
    C.num_envs = 65536; C.task = GPD_TASK_NONE; C.act_type = GPD_ACT_RAW_RPM; C.physics_flags = 31; C.auto_reset = 0; S.ld = 65536; S.dw_force = DEV(47);
    CHECK(gpd_swarm_pack(&S, &W, DEV(6), NULL, NULL) == 0, "gpd_swarm_pack");
    CHECK(gpd_swarm_bin(&W, NULL) == 0, "gpd_swarm_bin");
    n0 = hipstub_launches();
    CHECK(gpd_swarm_forces(&P, &W, 1, NULL) == 0 && hipstub_launches() == n0 + 1, "gpd_swarm_forces (build)");
    hipstub_last(last);
    CHECK(last[0] == 65792 / 64 + 1 && last[3] == 256, "one workgroup per 64 sorted drones + the drift workgroup");
    CHECK(gpd_swarm_forces(&P, &W, 0, NULL) == 0, "gpd_swarm_forces (replay)");
    CHECK(gpd_swarm_step(&P, &S, &C, &W, DEV(3), DEV(6), DEV(23), NULL) == 0, "gpd_swarm_step");
    W.list_cap = 2; CHECK(gpd_swarm_forces(&P, &W, 0, NULL) == GPD_EINVAL, "list_cap 2 is refused"); W.list_cap = 48;
    W.n_rows = 1 << 26; CHECK(gpd_swarm_forces(&P, &W, 0, NULL) != 0, "wake lists refuse 2^26 rows"); W.n_rows = 65792;
    W.cell = 5.0f; CHECK(gpd_swarm_bin(&W, NULL) != 0, "a 5 m grid is refused"); W.cell = 10.5f;
    uint32_t dbg[4];
    CHECK(gpd_debug_status(dbg, 0, NULL) == GPD_ENOTSUP, "gpd_debug_status in a release build");
    printf("%d launches recorded, %d checks failed\n", hipstub_launches(), failed);
    return failed;
}
