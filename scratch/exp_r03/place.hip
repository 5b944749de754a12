// Where do the workgroups of two kernels that run side by side land?  Two streams, each launches `spin` with G workgroups
// (x 256 threads, 30 us); every workgroup records (XCC_ID, HW_ID).  Plain streams, then streams created with
// hipExtStreamCreateWithCUMask in the two bit layouts one could assume (interleaved: bit i = CU i/8 of XCD i%8; blocked:
// bit i = CU i%32 of XCD i/32).
//   hipcc --offload-arch=gfx950 -O2 place.hip -o place && ./place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <map>
#include <vector>
#include <cstdint>
__global__ void spin(long long ticks, uint64_t* where) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0)
        where[blockIdx.x] = (static_cast<uint64_t>(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11))) << 32) |
                            static_cast<uint64_t>(__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)));
    while (static_cast<long long>(wall_clock64()) - t0 < ticks) {}
}
static int cu_of(uint64_t w) {   // 0 .. 255: xcc * 32 + (se, sh, cu) packed
    const uint32_t h = static_cast<uint32_t>(w), xcc = static_cast<uint32_t>(w >> 32) & 0xf;
    return static_cast<int>(xcc) << 16 | static_cast<int>((h >> 8) & 0xff);     // cu_id[11:8] sh[12] se[15:13]
}
static void report(const char* what, int G, const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {
    std::set<int> sa, sb, both;
    std::map<int, int> xa, xb;
    for (int i = 0; i < G; ++i) { sa.insert(cu_of(a[i])); sb.insert(cu_of(b[i])); xa[(a[i] >> 32) & 0xf]++; xb[(b[i] >> 32) & 0xf]++; }
    for (int c : sa) if (sb.count(c)) both.insert(c);
    printf("%-44s G=%3d: kernel A on %3zu CUs, kernel B on %3zu CUs, %3zu CUs hold both;  A per XCC:", what, G, sa.size(), sb.size(), both.size());
    for (auto& kv : xa) printf(" %d", kv.second);
    printf("  B per XCC:");
    for (auto& kv : xb) printf(" %d", kv.second);
    printf("\n");
}
int main() {
    uint64_t *wa, *wb;
    hipHostMalloc(&wa, 512 * 8); hipHostMalloc(&wb, 512 * 8);
    for (int G : {64, 128, 256}) {
        for (int mode = 0; mode < 3; ++mode) {
            hipStream_t s[2];
            for (int c = 0; c < 2; ++c) {
                if (mode == 0) { hipStreamCreateWithFlags(&s[c], hipStreamNonBlocking); continue; }
                uint32_t words[8] = {0};
                for (int i = 0; i < 256; ++i) {
                    const bool mine = mode == 1 ? ((i % 8) / 4 == c) : ((i / 32) / 4 == c);
                    if (mine) words[i / 32] |= 1u << (i % 32);
                }
                if (hipExtStreamCreateWithCUMask(&s[c], 8, words) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
            }
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, s[0], 3000, wa);
                hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, s[1], 3000, wb);
                hipDeviceSynchronize();
            }
            std::vector<uint64_t> a(wa, wa + G), b(wb, wb + G);
            report(mode == 0 ? "plain streams" : mode == 1 ? "CU mask, XCDs 0-3 / 4-7 if bits interleave" : "CU mask, XCDs 0-3 / 4-7 if bits are blocked", G, a, b);
            hipStreamDestroy(s[0]); hipStreamDestroy(s[1]);
        }
    }
    return 0;
}
