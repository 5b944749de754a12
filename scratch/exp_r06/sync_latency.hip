// How long does "launch one small kernel and wait for its results on the host" take on this box, by the way the host waits?
//   A: hipLaunchKernelGGL + hipStreamSynchronize          (what gpd_step_sync does)
//   B: the kernel ends with a system-scope release store of a sequence number into host-visible memory; the host spins on it
//   C: hipExtLaunchKernelGGL with a stop event + hipEventSynchronize
// The kernel reads 16 floats from host-visible memory, runs a dependent fma chain (`iters`), writes 16 floats back.
// Build: hipcc -O3 --offload-arch=gfx950 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__global__ void work(const float* in, float* out, int iters, volatile uint32_t* flag, uint32_t seq) {
    float x = in[threadIdx.x & 15];
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.999f, 0.001f);
    out[threadIdx.x & 15] = x;
    if (flag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the row above is visible before the flag is
        if (threadIdx.x == 0) __hip_atomic_store(const_cast<uint32_t*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 3000;
    float *in, *out; uint32_t* flag;
    hipHostMalloc(&in, 64 * 4, hipHostMallocDefault); hipHostMalloc(&out, 64 * 4, hipHostMallocDefault);
    hipHostMalloc(&flag, 64, hipHostMallocDefault);
    for (int i = 0; i < 16; ++i) in[i] = 1.0f + i;
    *flag = 0;
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t stop; hipEventCreate(&stop);
    const int chains[] = {0, 400, 2160};       // ~0, ~1 us, ~5.5 us of dependent fma at ~5.4 cycles / 2.1 GHz
    for (int iters : chains) {
        for (int variant = 0; variant < 3; ++variant) {
            uint32_t seq = 0;
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipStreamSynchronize(s);
                double t0 = now();
                for (int k = 0; k < n; ++k) {
                    in[0] = float(k);
                    if (variant == 0) {
                        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, in, out, iters, (volatile uint32_t*)nullptr, 0u);
                        hipStreamSynchronize(s);
                    } else if (variant == 1) {
                        ++seq;
                        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, in, out, iters, (volatile uint32_t*)flag, seq);
                        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { }
                    } else {
                        hipExtLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, nullptr, stop, 0, in, out, iters, (volatile uint32_t*)nullptr, 0u);
                        hipEventSynchronize(stop);
                    }
                }
                double dt = (now() - t0) / n * 1e6;
                if (dt < best) best = dt;
            }
            hipStreamSynchronize(s);
            printf("chain %5d  %s  %.2f us per launch+wait   (out[0] %.3f)\n", iters,
                   variant == 0 ? "A hipStreamSynchronize      " : variant == 1 ? "B flag in host memory, spin " : "C stop event, EventSynchronize", best, out[0]);
        }
    }
    return 0;
}
