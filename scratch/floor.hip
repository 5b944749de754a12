// Micro-benchmark: what does one dependent launch per step cost at N = 65536 (11.9 MB/step)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1; }

template <int NR, int NW>
__global__ __launch_bounds__(256) void k_soa(const float* __restrict__ in, float* __restrict__ out, int ld, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = in[(size_t)r * ld + i];
    float acc = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += v[r];
#pragma unroll
    for (int r = 0; r < NW; ++r) out[(size_t)r * ld + i] = acc + r;
}

// in place SoA: read 13 rows + 4 action, write 13 rows + obs 12 AoS (float4 x3)
__global__ __launch_bounds__(256) void k_like(float* __restrict__ kin, const float4* __restrict__ act, float4* __restrict__ obs, int ld, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) v[r] = kin[(size_t)r * ld + i];
    float4 a = act[i];
    float acc = a.x + a.y + a.z + a.w;
#pragma unroll
    for (int r = 0; r < 13; ++r) { v[r] = v[r] * 0.999f + acc * 1e-6f; kin[(size_t)r * ld + i] = v[r]; }
    obs[i * 3 + 0] = make_float4(v[0], v[1], v[2], v[3]);
    obs[i * 3 + 1] = make_float4(v[4], v[5], v[6], v[7]);
    obs[i * 3 + 2] = make_float4(v[8], v[9], v[10], v[11]);
}

int main() {
    const int n = 65536, ld = 65536;
    float *a, *b; float4 *act, *obs;
    CK(hipMalloc(&a, (size_t)32 * ld * 4)); CK(hipMalloc(&b, (size_t)32 * ld * 4));
    CK(hipMalloc(&act, (size_t)n * 16)); CK(hipMalloc(&obs, (size_t)n * 48));
    CK(hipMemset(a, 0, (size_t)32 * ld * 4)); CK(hipMemset(b, 0, (size_t)32 * ld * 4)); CK(hipMemset(act, 0, (size_t)n * 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int K = 64, REP = 50;
    auto bench = [&](const char* name, auto launch) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < K; ++i) launch(i);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < REP; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %.3f us per launch\n", name, ms * 1e3 / (K * REP));
        return 0;
    };
    bench("empty 256x256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, a); });
    bench("empty 1024x64", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, st, a); });
    bench("soa read17 write26 (a->b), 256x256", [&](int i) { hipLaunchKernelGGL((k_soa<17, 26>), dim3(256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, ld, n); });
    bench("soa read17 write26 (a->b), 1024x64", [&](int i) { hipLaunchKernelGGL((k_soa<17, 26>), dim3(1024), dim3(64), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, ld, n); });
    bench("soa read1 write1, 256x256", [&](int i) { hipLaunchKernelGGL((k_soa<1, 1>), dim3(256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, ld, n); });
    bench("soa read17 write1, 256x256", [&](int i) { hipLaunchKernelGGL((k_soa<17, 1>), dim3(256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, ld, n); });
    bench("soa read1 write26, 256x256", [&](int i) { hipLaunchKernelGGL((k_soa<1, 26>), dim3(256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, ld, n); });
    bench("in-place like step (13 rows + act + obs)", [&](int) { hipLaunchKernelGGL(k_like, dim3(256), dim3(256), 0, st, a, act, obs, ld, n); });
    bench("in-place like step, 1024x64", [&](int) { hipLaunchKernelGGL(k_like, dim3(1024), dim3(64), 0, st, a, act, obs, ld, n); });
    bench("in-place like step, 512x128", [&](int) { hipLaunchKernelGGL(k_like, dim3(512), dim3(128), 0, st, a, act, obs, ld, n); });
    return 0;
}
