// Micro-benchmark, round 2: what could one dependent launch per env step cost at N = 65536 if the state block were laid out /
// stored differently?  Same method as floor.hip (hipGraph of 64 launches, HIP events): the in-place "like step" mover with
//   (a) plain stores (round-1 baseline)         (b) non-temporal stores (write-through instead of an end-of-kernel flush)
//   (c) the 13 state floats packed into 4 x float4 per drone ([4][ld] float4: 4 loads + 4 stores per lane instead of 13 + 13)
//   (d) c + non-temporal                          (e) a 600-byte by-value kernel argument (GpdParams-sized) on top of (a)
// Run twice: HIP_FORCE_DEV_KERNARG=0 and =1 (where the kernel-argument segment lives).
//   hipcc --offload-arch=gfx950 -O3 floor2.hip -o floor2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
struct Big { float v[150]; };

__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1; }

template <bool NT>
__global__ __launch_bounds__(256) void k_like(float* __restrict__ kin, const float4* __restrict__ act, f4v* __restrict__ obs, int ld, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) v[r] = kin[(size_t)r * ld + i];
    float4 a = act[i];
    float acc = a.x + a.y + a.z + a.w;
#pragma unroll
    for (int r = 0; r < 13; ++r) {
        v[r] = v[r] * 0.999f + acc * 1e-6f;
        if (NT) __builtin_nontemporal_store(v[r], &kin[(size_t)r * ld + i]); else kin[(size_t)r * ld + i] = v[r];
    }
    f4v o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]}, o2 = {v[8], v[9], v[10], v[11]};
    if (NT) { __builtin_nontemporal_store(o0, &obs[i * 3]); __builtin_nontemporal_store(o1, &obs[i * 3 + 1]); __builtin_nontemporal_store(o2, &obs[i * 3 + 2]); }
    else { obs[i * 3] = o0; obs[i * 3 + 1] = o1; obs[i * 3 + 2] = o2; }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_like4(f4v* __restrict__ kin, const float4* __restrict__ act, f4v* __restrict__ obs, int ld, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f4v s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = kin[(size_t)r * ld + i];
    float4 a = act[i];
    float acc = (a.x + a.y + a.z + a.w) * 1e-6f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s[r] = s[r] * 0.999f + acc;
        if (NT) __builtin_nontemporal_store(s[r], &kin[(size_t)r * ld + i]); else kin[(size_t)r * ld + i] = s[r];
    }
    if (NT) { __builtin_nontemporal_store(s[0], &obs[i * 3]); __builtin_nontemporal_store(s[1], &obs[i * 3 + 1]); __builtin_nontemporal_store(s[2], &obs[i * 3 + 2]); }
    else { obs[i * 3] = s[0]; obs[i * 3 + 1] = s[1]; obs[i * 3 + 2] = s[2]; }
}

__global__ __launch_bounds__(256) void k_like_big(const Big P, float* __restrict__ kin, const float4* __restrict__ act, f4v* __restrict__ obs, int ld, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) v[r] = kin[(size_t)r * ld + i];
    float4 a = act[i];
    float acc = a.x + a.y + a.z + a.w + P.v[149] + P.v[75] + P.v[3];
#pragma unroll
    for (int r = 0; r < 13; ++r) { v[r] = v[r] * P.v[r * 7] + acc * 1e-6f; kin[(size_t)r * ld + i] = v[r]; }
    f4v o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]}, o2 = {v[8], v[9], v[10], v[11]};
    obs[i * 3] = o0; obs[i * 3 + 1] = o1; obs[i * 3 + 2] = o2;
}

// ... the same 600 bytes behind a POINTER to device memory (scalar loads of the used fields) instead of by value
__global__ __launch_bounds__(256) void k_like_bigptr(const Big* __restrict__ Pp, float* __restrict__ kin, const float4* __restrict__ act, f4v* __restrict__ obs, int ld, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Big& P = *Pp;
    float v[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) v[r] = kin[(size_t)r * ld + i];
    float4 a = act[i];
    float acc = a.x + a.y + a.z + a.w + P.v[149] + P.v[75] + P.v[3];
#pragma unroll
    for (int r = 0; r < 13; ++r) { v[r] = v[r] * P.v[r * 7] + acc * 1e-6f; kin[(size_t)r * ld + i] = v[r]; }
    f4v o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]}, o2 = {v[8], v[9], v[10], v[11]};
    obs[i * 3] = o0; obs[i * 3 + 1] = o1; obs[i * 3 + 2] = o2;
}

// The rollout kernel's memory pattern without its arithmetic: per "step" a lane reads one 16-byte action and writes a 48-byte
// observation row (as three coalesced float4 planes), a reward float and two flag bytes -- 64 steps per launch, non-temporal.
__global__ __launch_bounds__(256) void k_stream(const f4v* __restrict__ act, f4v* __restrict__ obs, float* __restrict__ rew,
                                                unsigned char* __restrict__ fl, int n, int steps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    f4v acc = {0, 0, 0, 0};
    for (int t = 0; t < steps; ++t) {
        const f4v a = __builtin_nontemporal_load(&act[(size_t)t * n + i]);
        acc = acc * 0.5f + a;
        __builtin_nontemporal_store(acc, &obs[((size_t)t * 3 + 0) * n + i]);
        __builtin_nontemporal_store(acc + 1.0f, &obs[((size_t)t * 3 + 1) * n + i]);
        __builtin_nontemporal_store(acc + 2.0f, &obs[((size_t)t * 3 + 2) * n + i]);
        __builtin_nontemporal_store(acc.x, &rew[(size_t)t * n + i]);
        __builtin_nontemporal_store((unsigned char)(acc.y > 0), &fl[(size_t)t * 2 * n + i]);
        __builtin_nontemporal_store((unsigned char)(acc.z > 0), &fl[((size_t)t * 2 + 1) * n + i]);
    }
}

// pure store stream: what the memory system takes when NOTHING is read (grid-stride, 16 bytes per lane and store, non-temporal)
__global__ __launch_bounds__(256) void k_fill(f4v* __restrict__ out, size_t n16, float v) {
    const f4v x = {v, v + 1.0f, v + 2.0f, v + 3.0f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(x, &out[i]);
}
__global__ __launch_bounds__(256) void k_sum(const f4v* __restrict__ in, float* __restrict__ out, size_t n16) {
    f4v acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        acc += __builtin_nontemporal_load(&in[i]);
    if (acc.x + acc.y + acc.z + acc.w == 12345.0f) out[0] = 1.0f;
}

int main() {
    const int n = 65536, ld = 65536;
    float* a; f4v* a4; float4* act; f4v* obs;
    CK(hipMalloc(&a, (size_t)16 * ld * 4)); CK(hipMalloc(&a4, (size_t)4 * ld * 16));
    CK(hipMalloc(&act, (size_t)n * 16)); CK(hipMalloc(&obs, (size_t)n * 48));
    CK(hipMemset(a, 0, (size_t)16 * ld * 4)); CK(hipMemset(a4, 0, (size_t)4 * ld * 16)); CK(hipMemset(act, 0, (size_t)n * 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int K = 64, REP = 200;
    Big P; for (int i = 0; i < 150; ++i) P.v[i] = 0.999f;
    const char* env = getenv("HIP_FORCE_DEV_KERNARG");
    printf("# HIP_FORCE_DEV_KERNARG=%s\n", env ? env : "(unset)");
    auto bench = [&](const char* name, auto launch) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < K; ++i) launch(i);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < REP; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %.3f us per launch\n", name, ms * 1e3 / (K * REP));
        return 0;
    };
    bench("empty 256x256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, a); });
    bench("in-place like step: 13 SoA rows + act + obs (r01 baseline)", [&](int) { hipLaunchKernelGGL((k_like<false>), dim3(256), dim3(256), 0, st, a, act, obs, ld, n); });
    bench("  ... non-temporal stores", [&](int) { hipLaunchKernelGGL((k_like<true>), dim3(256), dim3(256), 0, st, a, act, obs, ld, n); });
    bench("  ... state packed as 4 x float4 per drone", [&](int) { hipLaunchKernelGGL((k_like4<false>), dim3(256), dim3(256), 0, st, a4, act, obs, ld, n); });
    bench("  ... packed + non-temporal", [&](int) { hipLaunchKernelGGL((k_like4<true>), dim3(256), dim3(256), 0, st, a4, act, obs, ld, n); });
    bench("  ... SoA + a 600-byte by-value argument", [&](int) { hipLaunchKernelGGL(k_like_big, dim3(256), dim3(256), 0, st, P, a, act, obs, ld, n); });
    Big* Pd; CK(hipMalloc(&Pd, sizeof(Big))); CK(hipMemcpy(Pd, &P, sizeof(Big), hipMemcpyHostToDevice));
    bench("  ... SoA + the 600 bytes behind a device pointer", [&](int) { hipLaunchKernelGGL(k_like_bigptr, dim3(256), dim3(256), 0, st, Pd, a, act, obs, ld, n); });
    bench("  ... packed, 1024 x 64 threads", [&](int) { hipLaunchKernelGGL((k_like4<false>), dim3(1024), dim3(64), 0, st, a4, act, obs, ld, n); });
    {   // store-only and load-only streams over 1 GiB (far beyond the 256 MB Infinity Cache), at the headline's occupancy (256
        // workgroups = one wave per SIMD) and with the chip full (4096 workgroups)
        const size_t n16 = (size_t)1 << 26;               // 2^26 x 16 B = 1 GiB
        f4v* big; float* flag; CK(hipMalloc(&big, n16 * 16)); CK(hipMalloc(&flag, 4));
        CK(hipMemset(big, 0, n16 * 16));
        for (int grid : {256, 1024, 4096}) {
            for (int kind = 0; kind < 2; ++kind) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                auto go = [&]() { if (kind == 0) hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, st, big, n16, 1.0f);
                                  else hipLaunchKernelGGL(k_sum, dim3(grid), dim3(256), 0, st, big, flag, n16); };
                go(); go(); CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 20; ++i) go();
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("%-10s 1 GiB, %4d workgroups x 256                     %.0f GB/s\n", kind == 0 ? "store-only" : "load-only", grid,
                       (double)n16 * 16 * 20 / (ms * 1e-3) / 1e9);
            }
        }
        hipFree(big); hipFree(flag);
    }
    {   // the write-heavy stream of a 64-step rollout at N = 65536: 70 B per drone and step, 77 % of it written
        const int steps = 64;
        f4v *sa, *so; float* sr; unsigned char* sf;
        CK(hipMalloc(&sa, (size_t)steps * n * 16)); CK(hipMalloc(&so, (size_t)steps * 3 * n * 16));
        CK(hipMalloc(&sr, (size_t)steps * n * 4)); CK(hipMalloc(&sf, (size_t)steps * 2 * n));
        CK(hipMemset(sa, 0, (size_t)steps * n * 16));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, st, sa, so, sr, sf, n, steps);
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 2000;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, st, sa, so, sr, sf, n, steps);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)steps * n * 70.0;
        printf("%-58s %.3f us per step, %.0f GB/s (%.3f of 8 TB/s)\n", "rollout-shaped stream, no arithmetic (64 steps/launch)",
               ms * 1e3 / (reps * steps), bytes * reps / (ms * 1e-3) / 1e9, bytes * reps / (ms * 1e-3) / 8e12);
    }
    return 0;
}
