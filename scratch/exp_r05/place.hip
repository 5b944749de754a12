// Where do the workgroups of a ONE-ROUND launch land, and does dealing the heavy ones to different CUs shorten it?
// (VERDICT r04 "next" #3: the replay launch of dwg_force_kernel ends with the CU that holds the most pairs.)
//
//   hipcc -O3 --offload-arch=gfx950 scratch/exp_r05/place.hip -o scratch/exp/place && scratch/exp/place
//
// 1 025 workgroups x 256 threads, ~34 KB of LDS each (the force kernel's footprint: four per CU, all resident at once).  Every
// workgroup records the CU it runs on (HW_ID / XCC_ID) and spins through `w[b]` units of dependent FMAs per wave (a unit ~ a batch of
// 64 pairs); the weights are drawn like the bench scene's batches per wave (mean 10.3, max 20).  Orders compared, same weights:
//   identity      workgroup b does item b (the weights in their spatial order: neighbours are alike)
//   shuffled      a random permutation
//   descending    heaviest first
//   snake         sorted, dealt to the 256 residue classes b mod 256 boustrophedon (classes = CUs IF placement is b mod 256 -> CU)
// and the co-residency itself: which b share a CU, launch after launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void work(const int* __restrict__ item_of, const int* __restrict__ weight, unsigned long long* __restrict__ hw,
                                            unsigned long long* __restrict__ t01, float* __restrict__ sink, int unit) {
    __shared__ float pad[34 * 256];
    const int b = blockIdx.x;
    const unsigned long long t0 = wall_clock64();
    const int it = item_of[b];
    const int w = weight[it];
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    pad[threadIdx.x] = x;
    for (int k = 0; k < w * unit; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x = fmaf(x, y, 1e-6f);
    }
    pad[threadIdx.x + 256] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        hw[b] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11))) << 32) |
                static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)));
        t01[2 * b] = t0;
        t01[2 * b + 1] = wall_clock64();
        sink[b] = pad[(threadIdx.x + 300) & 511];
    }
}

int main() {
    const int G = 1025, unit = 40;
    std::mt19937 rng(7);
    // weights: spatially correlated (a slow wave + noise), mean ~10.3, clipped to 4..20 -- batches per wave of the bench scene
    std::vector<int> w(G);
    std::normal_distribution<float> nz(0.0f, 1.6f);
    for (int i = 0; i < G; ++i) {
        float v = 10.3f + 3.2f * sinf(i * 0.045f) + 1.8f * sinf(i * 0.31f + 1.0f) + nz(rng);
        w[i] = std::max(4, std::min(20, static_cast<int>(lroundf(v))));
    }
    w[G - 1] = 1;      // (the extra workgroup)
    const double mean = std::accumulate(w.begin(), w.end(), 0.0) / G;
    printf("weights: mean %.2f max %d\n", mean, *std::max_element(w.begin(), w.end()));
    std::vector<int> ident(G), shuf(G), desc(G), snake(G);
    std::iota(ident.begin(), ident.end(), 0);
    shuf = ident;
    std::shuffle(shuf.begin(), shuf.end() - 1, rng);
    desc = ident;
    std::stable_sort(desc.begin(), desc.end() - 1, [&](int a, int c) { return w[a] > w[c]; });
    // snake: sorted item k goes to round r = k / 256, class c = (r even ? k % 256 : 255 - k % 256), workgroup b = r * 256 + c
    for (int k = 0; k < G - 1; ++k) {
        const int r = k / 256, c = (r & 1) ? 255 - k % 256 : k % 256;
        snake[r * 256 + c] = desc[k];
    }
    snake[G - 1] = G - 1;
    int *d_item, *d_w;
    unsigned long long *d_hw, *d_t;
    float* d_sink;
    CK(hipMalloc(&d_item, G * 4)); CK(hipMalloc(&d_w, G * 4)); CK(hipMalloc(&d_hw, G * 8)); CK(hipMalloc(&d_t, G * 16)); CK(hipMalloc(&d_sink, G * 4));
    CK(hipMemcpy(d_w, w.data(), G * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> hw(G), first_hw;
    std::vector<unsigned long long> tt(2 * G);
    struct Ord { const char* name; std::vector<int>* v; };
    Ord orders[] = {{"identity", &ident}, {"shuffled", &shuf}, {"descending", &desc}, {"snake", &snake}};
    for (int pass = 0; pass < 3; ++pass)
        for (auto& o : orders) {
            CK(hipMemcpy(d_item, o.v->data(), G * 4, hipMemcpyHostToDevice));
            for (int k = 0; k < 5; ++k) work<<<G, 256>>>(d_item, d_w, d_hw, d_t, d_sink, unit);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int reps = 200;
            for (int k = 0; k < reps; ++k) work<<<G, 256>>>(d_item, d_w, d_hw, d_t, d_sink, unit);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hw.data(), d_hw, G * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(tt.data(), d_t, G * 16, hipMemcpyDeviceToHost));
            // per-CU load of this order under the OBSERVED placement, and the launch's own timeline
            std::map<unsigned long long, std::pair<int, int>> cu;          // id -> (workgroups, weight)
            auto id = [](unsigned long long h) { return ((h >> 32) & 0xf) << 16 | (h & 0xff00); };    // xcc | se, sh, cu
            for (int b = 0; b < G; ++b) { auto& e = cu[id(hw[b])]; e.first++; e.second += w[(*o.v)[b]]; }
            int mx = 0, mxw = 0;
            for (auto& kv : cu) { mx = std::max(mx, kv.second.first); mxw = std::max(mxw, kv.second.second); }
            unsigned long long tmin = ~0ull, tmax = 0;
            std::vector<double> life(G);
            for (int b = 0; b < G; ++b) { tmin = std::min(tmin, tt[2 * b]); tmax = std::max(tmax, tt[2 * b + 1]); life[b] = (tt[2 * b + 1] - tt[2 * b]) * 0.01; }
            std::sort(life.begin(), life.end());
            int same = 0;
            if (first_hw.empty()) first_hw = hw;
            for (int b = 0; b < G; ++b) same += id(hw[b]) == id(first_hw[b]);
            int mod = 0;        // does b mod 256 decide the CU?  count b whose CU equals that of b % 256
            for (int b = 256; b < G - 1; ++b) mod += id(hw[b]) == id(hw[b % 256]);
            printf("pass %d %-10s %7.2f us per launch | CUs used %3zu, most workgroups on one CU %d, heaviest CU %3d units (mean %.1f) | last launch: span %.2f us, "
                   "workgroup lifetime median %.2f max %.2f | same CU as the very first launch: %4d/%d | CU(b) == CU(b mod 256): %d/%d\n",
                   pass, o.name, ms * 1e3 / reps, cu.size(), mx, mxw, mean * G / 256.0, (tmax - tmin) * 0.01, life[G / 2], life[G - 1], same, G, mod, G - 257);
        }
    // the placement of the last launch, first 40 workgroups
    printf("workgroup -> (xcc, se, sh, cu):");
    for (int b = 0; b < 40; ++b) printf(" %d:(%llu,%llu,%llu,%llu)", b, (hw[b] >> 32) & 0xf, (hw[b] >> 13) & 7, (hw[b] >> 12) & 1, (hw[b] >> 8) & 0xf);
    printf("\n");
    return 0;
}
