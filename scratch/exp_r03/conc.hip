// Do kernels of two HIP streams run side by side on this part?  (round 3: the --split figures of bench.py look like
// T(pair) = 2 T(half), i.e. no overlap at all.)  A kernel of G workgroups x 256 threads spins for ~T us (s_memtime);
// L launches per stream, S streams; prints the wall time per launch group.
//   hipcc --offload-arch=gfx950 -O2 conc.hip -o conc && ./conc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ void spin(long long ticks, float* out) {
    const long long t0 = wall_clock64();
    float a = threadIdx.x;
    while (static_cast<long long>(wall_clock64()) - t0 < ticks) a = a * 1.0001f + 1.0f;
    if (a == -1.0f) out[0] = a;
}
int main() {
    float* out; hipMalloc(&out, 4);
    const int L = 200;
    for (int G : {64, 128, 256, 512}) for (int S : {1, 2, 4}) {
        std::vector<hipStream_t> st(S);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const long long ticks = 3000;   // wall_clock64 runs at 100 MHz: 30 us
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int l = 0; l < L; ++l) for (auto& s : st) hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, s, ticks, out);
            hipDeviceSynchronize();
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("G=%4d workgroups, %d streams: %.2f us per launch group (one launch alone spins 30 us)\n", G, S, us / L);
        }
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
