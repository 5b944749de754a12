// Single-wave VALU issue microbenchmark (gfx950): cycles per instruction for dependent / independent v_fma_f32
// streams, packed fp32, transcendental, SALU mixed in, at 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 issue.hip -o issue && ./issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-4f;
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a, a}, p1 = {a + 1, a}, p2 = {a + 2, a}, p3 = {a + 3, a}, pb = {b, b}, pc = {c, c};
    int s0 = iters;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 64 dependent fma
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(r0) : "v"(b), "v"(c));
        } else if (MODE == 1) { // 64 fma, 4 independent chains round-robin
            asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c));
        } else if (MODE == 2) { // 64 fma, 2 chains
            asm volatile(REP16("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")
                         : "+v"(r0), "+v"(r1) : "v"(b), "v"(c));
        } else if (MODE == 3) { // 64 dependent pk_fma
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p0) : "v"(pb), "v"(pc));
        } else if (MODE == 4) { // 64 pk_fma, 4 chains
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (MODE == 5) { // 64 dependent fmaak-style with literal (v_fmaak_f32 is 8 bytes)
            asm volatile(REP64("v_fmaak_f32 %0, %0, %1, 0x3d2c9880\n") : "+v"(r0) : "v"(b));
        } else if (MODE == 6) { // 32 fma + 32 salu interleaved (fma dependent)
            asm volatile(REP16("v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 1\n v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 1\n")
                         : "+v"(r0), "+s"(s0) : "v"(b), "v"(c) : "scc");
        } else if (MODE == 7) { // 16 x (rcp + 3 dependent fma)
            asm volatile(REP16("v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n")
                         : "+v"(r0) : "v"(b), "v"(c));
        } else if (MODE == 8) { // 16 x (rcp + 3 independent fma)
            asm volatile(REP16("v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c));
        } else if (MODE == 9) { // cmp -> cndmask pairs with nops as the compiler emits: 16 x (cmp, nop1, cndmask, fma)
            asm volatile(REP16("v_cmp_gt_f32 vcc, 0, %0\n s_nop 1\n v_cndmask_b32 %1, %1, %0, vcc\n v_fma_f32 %0, %0, %2, %3\n")
                         : "+v"(r0), "+v"(r1) : "v"(b), "v"(c) : "vcc");
        } else if (MODE == 11) { // 16 x (3 dep fma + s_branch to the next instruction: a TAKEN branch)
            asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_branch 0\n")
                         : "+v"(r0) : "v"(b), "v"(c));
        } else if (MODE == 12) { // 16 x (3 dep fma + s_cbranch_scc1 NOT taken)
            asm volatile("s_cmp_eq_u32 0, 1\n" REP16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_cbranch_scc1 0\n")
                         : "+v"(r0) : "v"(b), "v"(c) : "scc");
        } else if (MODE == 13) { // 16 x (3 dep fma + taken branch skipping 16 instructions (128 B))
            asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_branch 16\n"
                               REP16("v_fma_f32 %0, %0, %1, %2\n"))
                         : "+v"(r0) : "v"(b), "v"(c));
        } else if (MODE == 14) { // 16 x (3 dep fma + s_cbranch_execz NOT taken)
            asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_cbranch_execz 0\n")
                         : "+v"(r0) : "v"(b), "v"(c));
        } else if (MODE == 10) { // 64 fma, 8 independent chains
            asm volatile(REP4(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
                              REP4("v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"))
                         REP4(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %4, %4, %8, %9\n")) 
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.x + p2.x + p3.x + p0.y + s0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int n_instr, int waves_per_simd) {
    float* out; long long* cyc;
    int block = 256 * waves_per_simd, grid = 256;
    hipMalloc(&out, sizeof(float) * block * grid); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, block>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double per = (double)c / ((double)iters * n_instr);
    printf("%-44s waves/SIMD %d: %7.3f clk-ticks/instr (s_memtime)  %8.3f ns/instr  wall %.1f us\n", name, waves_per_simd, per,
           ms * 1e6 / ((double)iters * n_instr), ms * 1e3);
    hipFree(out); hipFree(cyc);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int w = 1; w <= 2; w *= 2) {
        run<0>("64 dependent v_fma_f32", 64, w);
        run<2>("64 v_fma_f32, 2 chains", 64, w);
        run<1>("64 v_fma_f32, 4 chains", 64, w);
        run<10>("72 v_fma_f32, 8 chains", 72, w);
        run<3>("64 dependent v_pk_fma_f32", 64, w);
        run<4>("64 v_pk_fma_f32, 4 chains", 64, w);
        run<5>("64 dependent v_fmaak_f32 (literal)", 64, w);
        run<6>("32 dep fma + 32 s_add interleaved", 64, w);
        run<7>("16 x (rcp + 3 dep fma)", 64, w);
        run<8>("16 x (rcp + 3 indep fma)", 64, w);
        run<9>("16 x (cmp, s_nop 1, cndmask, fma) [3 valu]", 48, w);
        run<11>("16 x (3 dep fma + TAKEN s_branch +0) /group", 16, w);
        run<13>("16 x (3 dep fma + TAKEN s_branch +128B) /group", 16, w);
        run<12>("16 x (3 dep fma + not-taken s_cbranch_scc1) /group", 16, w);
        run<14>("16 x (3 dep fma + not-taken s_cbranch_execz) /group", 16, w);
    }
    return 0;
}
