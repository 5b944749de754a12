// abi.hip -- the small kernels (masked reset, action-history rows, batched DSLPID, state vectors, clock probe), RCCL, and the library-level
// entries of the C ABI (version, last error, struct sizes, debug status)
#include <chrono>
#include "gpd_common.inc"

// the completion word of the `*_sync` entries (gpd_common.inc: GpdDone)
namespace {
struct DoneWord { uint32_t* p = nullptr; uint32_t seq = 0; uint32_t unsynced = 0; bool tried = false; };
DoneWord& done_word() {
    static thread_local DoneWord w;        // (never freed: 64 page-locked bytes per calling thread; the runtime may be gone when a thread ends)
    if (!w.tried) {
        w.tried = true;
        void* q = nullptr;
        if (hipHostMalloc(&q, 64, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && q != nullptr) {
            w.p = static_cast<uint32_t*>(q);
            *w.p = 0u;
        } else {
            (void)hipGetLastError();
        }
    }
    return w;
}
}  // namespace

GpdDone gpd_detail_done_begin() {
    static const char* const how = getenv("GPD_STEP_SYNC_WAIT");
    DoneWord& w = done_word();
    const bool word = w.p != nullptr && !(how != nullptr && how[0] == 's');
    return GpdDone{word ? w.p : nullptr, ++w.seq, false};
}

int gpd_detail_done_wait(const GpdDone& d, void* stream, const char* who) {
    DoneWord& w = done_word();
    if (d.used && d.flag != nullptr && ++w.unsynced < 256u) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 1;; ++spins) {
            if (__atomic_load_n(d.flag, __ATOMIC_ACQUIRE) == d.seq) return 0;
            __builtin_ia32_pause();
            if ((spins & 0xfffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    w.unsynced = 0;
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        gpd_detail_last_error() = std::string(who) + ": hipStreamSynchronize: " + hipGetErrorString(e);
        return static_cast<int>(e);
    }
    return 0;
}

std::string& gpd_detail_last_error() {
    thread_local std::string e;
    return e;
}

namespace {

// ------------------------------------------------------------------------------------------------
// masked reset (envs/BaseAviary.py:451-477)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gpd_reset_kernel(const GpdState S, const float* __restrict__ init_pose,
                                                           int init_per_env, const uint8_t* __restrict__ mask,
                                                           int num_envs, int D, int reset_pid,
                                                           float* __restrict__ obs12) {
    const int64_t N = static_cast<int64_t>(num_envs) * D;
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= N) return;
    const int64_t env = n / D;
    const int d = static_cast<int>(n - env * D);
    if (mask && !mask[env]) return;
    const int64_t ld = S.ld;
    const float* ip = init_pose + (init_per_env ? n * 7 : static_cast<int64_t>(d) * 7);
    kin_P(S.kin, ld)[n] = make_float4(ip[0], ip[1], ip[2], 0.0f);          // pose of the reset; velocities and body rates zero
    kin_Q(S.kin, ld)[n] = make_float4(ip[3], ip[4], ip[5], ip[6]);
    kin_V(S.kin, ld)[n] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    S.kin[12 * ld + n] = 0.0f;
    if (S.last_rpm) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S.last_rpm[r * ld + n] = 0.0f;
    }
    if (reset_pid && S.pid) {
#pragma unroll
        for (int r = 0; r < 9; ++r) S.pid[r * ld + n] = 0.0f;
    }
    if (d == 0) S.step_counter[env] = 0;
    if (obs12) {
        float roll, pitch, yaw;
        quat_to_rpy(ip[3], ip[4], ip[5], ip[6], roll, pitch, yaw);
        store_obs12(obs12, n, ip[0], ip[1], ip[2], roll, pitch, yaw, 0, 0, 0, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// Full KIN observation rows: [ pos | rpy | vel | ang_v | the H most recent actions, oldest first ]
// (envs/BaseRLAviary.py:307-320), materialised on request from the action ring (GpdState: a double ring [2H][N][A],
// slot-major; the H most recent actions of an aviary are the H consecutive slots starting at its ring_pos).  Pure data
// movement: one lane per output float, so that a wave writes 256 contiguous bytes.
// ------------------------------------------------------------------------------------------------
// A workgroup owns R consecutive drones (R = 64, 32, 16 ...: as many as fit WHOLE rows into 48 KiB of LDS) and (blockIdx.y) one
// step; it is a transpose through LDS, and what it writes is ONE contiguous block of global memory -- the R rows follow each
// other -- streamed out as 16-byte pieces:
//   in : a history slot -- of the ring, or an action block of the call -- is a contiguous [N][A] block -> R lanes read the R
//        drones' A floats of slot i (one 16-byte load per lane for A = 4: R x 16 contiguous bytes), 256 / R slots in flight;
//        the twelve kinematic floats of the R rows are contiguous in obs12;
//   out: R x (12 + H*A) floats, contiguous.
// (History of this kernel at N = 65 536, 240 Hz rows of 492 floats: one lane per output float, sixteen 16-byte pieces from
// sixteen 1 MB-apart slots per wave: 116 us per step; 64 drones x 16 slots per workgroup, a 256-byte segment per drone written
// by whichever workgroup held that chunk: 57-60 us -- segments that start and end inside 128-byte lines, rewritten piecemeal;
// whole rows per workgroup: see DESIGN.md section 3.3.)  Two callers:
//   gpd_hist_rows  rows of the CURRENT state (after gpd_step, which pushed its action itself): slot i of the window is ring slot
//                  ring_pos + i;
//   gpd_full_obs   rows of the K steps of a rollout: slot i of step t's window is the action of step s = t - (H-1) + i of this
//                  call (s >= 0), or -- for steps before the call -- ring slot ring_pos + H + s, the ring as the rollout found it.
__global__ __launch_bounds__(kBlock) void gpd_hist_rows_kernel(uint32_t N, int D, int A, int H, int R, const float* __restrict__ ring,
                                                               const int32_t* __restrict__ ring_pos,
                                                               const float* __restrict__ obs12, int64_t obs_stride,
                                                               const float* __restrict__ actions, int64_t act_stride,
                                                               float* __restrict__ out, int64_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) float hist_tile[];     // [R][Wp], Wp = W | 1 (odd: drone-strided accesses spread over the banks)
    const int tid = threadIdx.x;
    const uint32_t n0 = blockIdx.x * static_cast<uint32_t>(R);
    const int t = blockIdx.y;                                // step of the call (0 for the current-state rows)
    const uint32_t W = 12u + static_cast<uint32_t>(H * A), Wp = W | 1u;
    const uint32_t rows = N - n0 < static_cast<uint32_t>(R) ? N - n0 : static_cast<uint32_t>(R);
    obs12 += t * obs_stride;
    out += t * out_stride;
    // ---- in: history slots ----
    const int d = tid % R, sl0 = tid / R, spp = kBlock / R;  // this thread's drone; slots per pass
    const uint32_t n = n0 + d;
    const bool have = static_cast<uint32_t>(d) < rows;
    const int p = have ? ring_pos[n / static_cast<uint32_t>(D)] : 0;
    for (int i = sl0; i < H; i += spp) {
        const int s = t - (H - 1) + i;                       // (rollout rows) the step of this call the action belongs to
        const float* src = (actions && s >= 0) ? actions + s * act_stride + static_cast<size_t>(n) * A
                                               : ring + (static_cast<size_t>(p + (actions ? H + s : i)) * N + n) * A;
        if (have) {
            float* dst = hist_tile + d * Wp + 12 + i * A;
            if (A == 4) { const float4 q = *reinterpret_cast<const float4*>(src); dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w; }
            else for (int a = 0; a < A; ++a) dst[a] = src[a];
        }
    }
    // ---- in: the kinematic part, rows x 12 floats, contiguous in obs12 ----
    for (uint32_t j = tid; j < rows * 12u; j += kBlock) {
        const uint32_t r = j / 12u, c = j - r * 12u;
        hist_tile[r * Wp + c] = obs12[static_cast<size_t>(n0) * 12 + j];
    }
    __syncthreads();
    // ---- out: rows x W floats, contiguous ----
    float* const dst = out + static_cast<size_t>(n0) * W;
    if ((W % 4u) == 0u && (reinterpret_cast<uintptr_t>(dst) % 16u) == 0u) {
        for (uint32_t j = tid; j < rows * (W / 4u); j += kBlock) {
            const uint32_t f = 4u * j, r = f / W, c = f - r * W;
            const float* tp = hist_tile + r * Wp + c;
            __builtin_nontemporal_store(f4v{tp[0], tp[1], tp[2], tp[3]}, reinterpret_cast<f4v*>(dst + f));
        }
    } else {
        for (uint32_t f = tid; f < rows * W; f += kBlock) {
            const uint32_t r = f / W, c = f - r * W;
            __builtin_nontemporal_store(hist_tile[r * Wp + c], dst + f);
        }
    }
}

// pushes the actions of the K steps of a call into the ring (the last H of them survive): blockIdx.y = 0 is the newest step
__global__ __launch_bounds__(kBlock) void gpd_hist_push_kernel(int K, uint32_t N, int D, int A, int H,
                                                               const float* __restrict__ actions, int64_t act_stride,
                                                               float* __restrict__ ring, const int32_t* __restrict__ ring_pos) {
    // one drone per thread (A = 4: one 16-byte load, two 16-byte stores -- the float-per-thread form took 68 us for the 64 steps
    // of a rollout of 65 536 drones, most of it integer division)
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const int s = K - 1 - static_cast<int>(blockIdx.y);
    if (n >= N || s < 0) return;
    const size_t NA = static_cast<size_t>(N) * A;
    int q = ring_pos[n / static_cast<uint32_t>(D)] + s;
    q -= (q / H) * H;
    const float* src = actions + s * act_stride + static_cast<size_t>(n) * A;
    float* r0 = ring + static_cast<size_t>(q) * NA + static_cast<size_t>(n) * A;
    float* r1 = r0 + static_cast<size_t>(H) * NA;
    if (A == 4) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        *reinterpret_cast<float4*>(r0) = v;
        *reinterpret_cast<float4*>(r1) = v;
    } else {
        for (int a = 0; a < A; ++a) { const float v = src[a]; r0[a] = v; r1[a] = v; }
    }
}

// ... and then, in a launch of its own (every lane of the push has read the old value), the aviaries' ring positions advance
__global__ __launch_bounds__(kBlock) void gpd_hist_advance_kernel(int K, int E, int H, int32_t* __restrict__ ring_pos) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e < E) ring_pos[e] = (ring_pos[e] + K) % H;
}


// ------------------------------------------------------------------------------------------------
// standalone batched DSLPIDControl.computeControl
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gpd_pid_kernel(
    const GpdParams P, float* __restrict__ pid, int64_t ld, float dt, const float* __restrict__ cur_pos,
    const float* __restrict__ cur_quat, const float* __restrict__ cur_vel, const float* __restrict__ target_pos,
    const float* __restrict__ target_rpy, const float* __restrict__ target_vel,
    const float* __restrict__ target_rpy_rates, float* __restrict__ rpm_out, float* __restrict__ pos_e_out,
    float* __restrict__ yaw_e_out, int n_total, uint32_t* __restrict__ done_flag, const uint32_t done_seq) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= n_total) return;
    Kin k{};
    k.px = cur_pos[n * 3]; k.py = cur_pos[n * 3 + 1]; k.pz = cur_pos[n * 3 + 2];
    const float4 q = reinterpret_cast<const float4*>(cur_quat)[n];
    k.qx = q.x; k.qy = q.y; k.qz = q.z; k.qw = q.w;
    k.vx = cur_vel[n * 3]; k.vy = cur_vel[n * 3 + 1]; k.vz = cur_vel[n * 3 + 2];
    Pid s;
    s.ipx = pid[0 * ld + n]; s.ipy = pid[1 * ld + n]; s.ipz = pid[2 * ld + n];
    s.lr = pid[3 * ld + n]; s.lp = pid[4 * ld + n]; s.ly = pid[5 * ld + n];
    s.irx = pid[6 * ld + n]; s.iry = pid[7 * ld + n]; s.irz = pid[8 * ld + n];
    float roll, pitch, yaw;
    quat_to_rpy(k.qx, k.qy, k.qz, k.qw, roll, pitch, yaw);
    const Mat3 R = quat_to_mat(k.qx, k.qy, k.qz, k.qw);
    const float tyaw = target_rpy ? target_rpy[n * 3 + 2] : 0.0f;
    float tv[3] = {0, 0, 0}, tr[3] = {0, 0, 0};
    if (target_vel) { tv[0] = target_vel[n * 3]; tv[1] = target_vel[n * 3 + 1]; tv[2] = target_vel[n * 3 + 2]; }
    if (target_rpy_rates) {
        tr[0] = target_rpy_rates[n * 3]; tr[1] = target_rpy_rates[n * 3 + 1]; tr[2] = target_rpy_rates[n * 3 + 2];
    }
    float rpm[4], pe[3], ye;
    dslpid(P, dt, 1.0f / dt, k, roll, pitch, yaw, R, target_pos[n * 3], target_pos[n * 3 + 1], target_pos[n * 3 + 2], tyaw,
           tv[0], tv[1], tv[2], tr[0], tr[1], tr[2], s, rpm, pe, &ye);
    pid[0 * ld + n] = s.ipx; pid[1 * ld + n] = s.ipy; pid[2 * ld + n] = s.ipz;
    pid[3 * ld + n] = s.lr; pid[4 * ld + n] = s.lp; pid[5 * ld + n] = s.ly;
    pid[6 * ld + n] = s.irx; pid[7 * ld + n] = s.iry; pid[8 * ld + n] = s.irz;
    reinterpret_cast<float4*>(rpm_out)[n] = make_float4(rpm[0], rpm[1], rpm[2], rpm[3]);
    if (pos_e_out) { pos_e_out[n * 3] = pe[0]; pos_e_out[n * 3 + 1] = pe[1]; pos_e_out[n * 3 + 2] = pe[2]; }
    if (yaw_e_out) yaw_e_out[n] = ye;
    signal_done(done_flag, done_seq, n == 0);            // (gpd_pid_sync with n <= 64: one wave; NULL otherwise)
}

// ------------------------------------------------------------------------------------------------
// 20-float state vectors (envs/BaseAviary.py:559-561)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gpd_state20_kernel(const GpdState S, const float* __restrict__ obs12,
                                                             float* __restrict__ out, int n_total) {
    const int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= n_total) return;
    state20_row(S, obs12, out, n);
}

// ------------------------------------------------------------------------------------------------
// Shader-clock probe (diagnostics for bench.py's issue roofline): 256 workgroups x 4 waves = one wave per SIMD, like the
// headline launch, each running a dependent v_fma_f32 chain; lane 0 of workgroup 0 reports the shader-clock cycles
// (s_memtime) and the constant-rate wall-clock ticks (s_memrealtime) the chain took.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gpd_clock_probe_kernel(unsigned long long* __restrict__ out, int iters) {
    float r = threadIdx.x * 1e-3f;
    const float b = 1.0001f, c = 1e-4f;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#define GPD_FMA4 "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
#define GPD_FMA16 GPD_FMA4 GPD_FMA4 GPD_FMA4 GPD_FMA4
        asm volatile(GPD_FMA16 GPD_FMA16 GPD_FMA16 GPD_FMA16 : "+v"(r) : "v"(b), "v"(c));
#undef GPD_FMA16
#undef GPD_FMA4
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
    if (r == 12345.678f) out[2] = 1;                       // (keeps the chain alive)
}

// ------------------------------------------------------------------------------------------------
// RCCL, resolved at run time: libgpd.so has no link-time dependency on it (a single-GPU consumer never loads it), and
// inside a PyTorch process dlopen() by SONAME returns the copy torch already mapped instead of a second one.
// ------------------------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    // (optional: only gpd_p2p_group needs them -- a library without them still serves the all-gather)
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl R = [] {
        Rccl r;
        const char* env = getenv("GPD_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            r.why = dlerror();
        }
        if (!r.handle) return r;
        auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p) r.why = std::string("missing symbol ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.handle, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.handle, "ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.handle, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.handle, "ncclRecv"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.AllGather || !r.GetErrorString) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
        return r;
    }();
    return R;
}

int rccl_fail(ncclResult_t e, const char* where) {
    g_last_error = std::string(where) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
    return 1000 + static_cast<int>(e);          // (positive, outside hipError_t's range)
}

int need_rccl(const char* who) {
    if (rccl().handle) return 0;
    return fail(GPD_ENOTSUP, (std::string(who) + ": RCCL is not available (" + rccl().why + ")").c_str());
}

}  // namespace

extern "C" {

int gpd_abi_version(void) { return GPD_ABI_VERSION; }

const char* gpd_last_error(void) { return g_last_error.c_str(); }

void gpd_struct_sizes(int32_t out[3]) {
    out[0] = static_cast<int32_t>(sizeof(GpdParams));
    out[1] = static_cast<int32_t>(sizeof(GpdState));
    out[2] = static_cast<int32_t>(sizeof(GpdStepCfg));
}

static int hist_args(const char* who, const GpdState* st, int32_t n_drones, int32_t D, int32_t A) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string(who) + ": " + msg).c_str()); };
    if (!st || !st->act_ring || !st->ring_pos || st->hist_len <= 0) return bad(GPD_EINVAL, "state has no action ring (act_ring / ring_pos / hist_len)");
    if (n_drones <= 0 || D <= 0 || n_drones % D != 0 || A <= 0 || A > 4)
        return bad(GPD_EINVAL, "n_drones must be a positive multiple of drones_per_env and act_dim in 1..4");
    const int64_t W = 12 + static_cast<int64_t>(st->hist_len) * A;
    if (static_cast<int64_t>(n_drones) * W >= (1LL << 32)) return bad(GPD_ERANGE, "n_drones*(12+hist_len*act_dim) must be < 2^32");
    if ((W | 1) * 4 > 48 * 1024) return bad(GPD_ERANGE, "a row of 12+hist_len*act_dim floats must fit 48 KiB");
    return 0;
}

// drones per workgroup of gpd_hist_rows_kernel: the largest power of two <= 64 whose whole rows fit 48 KiB of LDS
static int hist_rows_per_wg(int64_t W) {
    int R = 64;
    while (R > 1 && static_cast<int64_t>(R) * (W | 1) * 4 > 48 * 1024) R >>= 1;
    return R;
}

int gpd_hist_rows(const GpdState* state, int32_t n_drones, int32_t drones_per_env, int32_t act_dim, const float* obs12,
                  float* obs_full, void* stream) {
    if (int rc = hist_args("gpd_hist_rows", state, n_drones, drones_per_env, act_dim)) return rc;
    if (!obs12 || !obs_full) return fail(GPD_EINVAL, "gpd_hist_rows: NULL obs12/obs_full");
    const int64_t W = 12 + static_cast<int64_t>(state->hist_len) * act_dim;
    const int R = hist_rows_per_wg(W);
    const dim3 grid(static_cast<unsigned>((n_drones + R - 1) / R));
    hipLaunchKernelGGL(gpd_hist_rows_kernel, grid, dim3(kBlock), static_cast<size_t>(R) * (W | 1) * 4,
                       static_cast<hipStream_t>(stream), static_cast<uint32_t>(n_drones), drones_per_env, act_dim, state->hist_len, R,
                       state->act_ring, state->ring_pos, obs12, static_cast<int64_t>(0), static_cast<const float*>(nullptr),
                       static_cast<int64_t>(0), obs_full, static_cast<int64_t>(0));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_hist_rows launch");
    return 0;
}

int gpd_full_obs(const GpdState* state, int32_t num_steps, int32_t n_drones, int32_t drones_per_env, int32_t act_dim,
                 const float* obs12, int64_t obs_step_stride, const float* actions, int64_t action_step_stride,
                 float* obs_full, int64_t full_step_stride, void* stream) {
    if (int rc = hist_args("gpd_full_obs", state, n_drones, drones_per_env, act_dim)) return rc;
    if (!actions) return fail(GPD_EINVAL, "gpd_full_obs: NULL actions");
    if (obs_full && !obs12) return fail(GPD_EINVAL, "gpd_full_obs: obs_full needs obs12");
    if (num_steps <= 0 || num_steps > 65535) return fail(GPD_EINVAL, "gpd_full_obs: num_steps must be in 1..65535");
    if (obs_step_stride < 0 || action_step_stride < 0 || full_step_stride < 0)
        return fail(GPD_EINVAL, "gpd_full_obs: strides must be non-negative");
    const int H = state->hist_len;
    const int64_t W = 12 + static_cast<int64_t>(H) * act_dim;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (obs_full) {
        const int R = hist_rows_per_wg(W);
        const dim3 grid(static_cast<unsigned>((n_drones + R - 1) / R), static_cast<unsigned>(num_steps));
        hipLaunchKernelGGL(gpd_hist_rows_kernel, grid, dim3(kBlock), static_cast<size_t>(R) * (W | 1) * 4, st,
                           static_cast<uint32_t>(n_drones), drones_per_env, act_dim, H, R, state->act_ring, state->ring_pos, obs12,
                           obs_step_stride, actions, action_step_stride, obs_full, full_step_stride);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "gpd_full_obs launch");
    }
    // the ring is read by the kernel above and updated by the next two: same stream, in order
    const int keep = num_steps < H ? num_steps : H;
    const dim3 grid2(static_cast<unsigned>((n_drones + kBlock - 1) / kBlock), static_cast<unsigned>(keep));
    hipLaunchKernelGGL(gpd_hist_push_kernel, grid2, dim3(kBlock), 0, st, num_steps, static_cast<uint32_t>(n_drones),
                       drones_per_env, act_dim, H, actions, action_step_stride, state->act_ring, state->ring_pos);
    const int E = n_drones / drones_per_env;
    hipLaunchKernelGGL(gpd_hist_advance_kernel, dim3(static_cast<unsigned>((E + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       num_steps, E, H, state->ring_pos);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_full_obs (ring update) launch");
    return 0;
}

int gpd_reset(const GpdState* state, const float* init_pose, int32_t init_per_env, const uint8_t* mask,
              int32_t num_envs, int32_t drones_per_env, int32_t reset_pid, float* obs12, void* stream) {
    if (!state || !state->kin || !state->step_counter || !init_pose)
        return fail(GPD_EINVAL, "gpd_reset: NULL state/init_pose");
    if (const char* why = state_layout_problem(state)) return fail(GPD_EINVAL, (std::string("gpd_reset: ") + why).c_str());
    if (num_envs <= 0 || drones_per_env <= 0) return fail(GPD_EINVAL, "gpd_reset: sizes must be positive");
    const int64_t N = static_cast<int64_t>(num_envs) * drones_per_env;
    if (state->ld < N) return fail(GPD_EINVAL, "gpd_reset: state.ld < num_envs*drones_per_env");
    const int64_t blocks = (N + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(gpd_reset_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), *state, init_pose, init_per_env, mask, num_envs,
                       drones_per_env, reset_pid, obs12);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_reset launch");
    return 0;
}

namespace {
int pid_impl(const char* who, const GpdParams* params, float* pid, int64_t ld, float ctrl_dt, const float* cur_pos,
             const float* cur_quat, const float* cur_vel, const float* target_pos, const float* target_rpy,
             const float* target_vel, const float* target_rpy_rates, float* rpm, float* pos_e, float* yaw_e,
             int32_t n, void* stream, GpdDone* done) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string(who) + ": " + msg).c_str()); };
    if (!params || !pid || !cur_pos || !cur_quat || !cur_vel || !target_pos || !rpm) return bad(GPD_EINVAL, "NULL argument");
    if (n <= 0 || ld < n) return bad(GPD_EINVAL, "need 0 < n <= ld");
    if (params->pid_kf <= 0.0f) return bad(GPD_ENOTSUP, "no DSLPID controller for this airframe");
    const int blocks = (n + kBlock - 1) / kBlock;
    if (done != nullptr) done->used = done->flag != nullptr && n <= 64;          // one wave: see GpdDone
    hipLaunchKernelGGL(gpd_pid_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *params, pid,
                       ld, ctrl_dt, cur_pos, cur_quat, cur_vel, target_pos, target_rpy, target_vel, target_rpy_rates,
                       rpm, pos_e, yaw_e, n, (done != nullptr && done->used) ? done->flag : nullptr, done != nullptr ? done->seq : 0u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, (std::string(who) + " launch").c_str());
    return 0;
}
}  // namespace

int gpd_pid(const GpdParams* params, float* pid, int64_t ld, float ctrl_dt, const float* cur_pos,
            const float* cur_quat, const float* cur_vel, const float* target_pos, const float* target_rpy,
            const float* target_vel, const float* target_rpy_rates, float* rpm, float* pos_e, float* yaw_e,
            int32_t n, void* stream) {
    return pid_impl("gpd_pid", params, pid, ld, ctrl_dt, cur_pos, cur_quat, cur_vel, target_pos, target_rpy, target_vel,
                    target_rpy_rates, rpm, pos_e, yaw_e, n, stream, nullptr);
}

int gpd_pid_sync(const GpdParams* params, float* pid, int64_t ld, float ctrl_dt, const float* cur_pos,
                 const float* cur_quat, const float* cur_vel, const float* target_pos, const float* target_rpy,
                 const float* target_vel, const float* target_rpy_rates, float* rpm, float* pos_e, float* yaw_e,
                 int32_t n, void* stream) {
    GpdDone done = gpd_detail_done_begin();
    if (int rc = pid_impl("gpd_pid_sync", params, pid, ld, ctrl_dt, cur_pos, cur_quat, cur_vel, target_pos, target_rpy, target_vel,
                          target_rpy_rates, rpm, pos_e, yaw_e, n, stream, &done))
        return rc;
    return gpd_detail_done_wait(done, stream, "gpd_pid_sync");
}

int gpd_state_vectors(const GpdState* state, const float* obs12, float* state20, int32_t n, void* stream) {
    if (!state || !state->kin || !obs12 || !state20) return fail(GPD_EINVAL, "gpd_state_vectors: NULL argument");
    if (const char* why = state_layout_problem(state)) return fail(GPD_EINVAL, (std::string("gpd_state_vectors: ") + why).c_str());
    if (n <= 0 || state->ld < n) return fail(GPD_EINVAL, "gpd_state_vectors: need 0 < n <= state.ld");
    const int blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(gpd_state20_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *state,
                       obs12, state20, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_state_vectors launch");
    return 0;
}

int gpd_comm_unique_id(uint8_t id[GPD_COMM_ID_BYTES]) {
    if (!id) return fail(GPD_EINVAL, "gpd_comm_unique_id: NULL id");
    if (int rc = need_rccl("gpd_comm_unique_id")) return rc;
    static_assert(GPD_COMM_ID_BYTES == sizeof(ncclUniqueId), "GPD_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId u;
    ncclResult_t e = rccl().GetUniqueId(&u);
    if (e != ncclSuccess) return rccl_fail(e, "ncclGetUniqueId");
    memcpy(id, &u, sizeof(u));
    return 0;
}

int gpd_comm_init(void** comm, const uint8_t id[GPD_COMM_ID_BYTES], int32_t rank, int32_t world_size) {
    if (!comm || !id) return fail(GPD_EINVAL, "gpd_comm_init: NULL comm/id");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(GPD_EINVAL, "gpd_comm_init: need 0 <= rank < world_size");
    if (int rc = need_rccl("gpd_comm_init")) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    ncclResult_t e = rccl().CommInitRank(&c, world_size, u, rank);
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommInitRank");
    *comm = c;
    return 0;
}

int gpd_comm_count(void* comm, int32_t* n_ranks) {
    if (!comm || !n_ranks) return fail(GPD_EINVAL, "gpd_comm_count: NULL comm/n_ranks");
    if (int rc = need_rccl("gpd_comm_count")) return rc;
    int n = 0;
    ncclResult_t e = rccl().CommCount(static_cast<ncclComm_t>(comm), &n);
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommCount");
    *n_ranks = n;
    return 0;
}

int gpd_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (int rc = need_rccl("gpd_comm_destroy")) return rc;
    ncclResult_t e = rccl().CommDestroy(static_cast<ncclComm_t>(comm));
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommDestroy");
    return 0;
}

int gpd_allgather_obs(void* comm, const float* shard, float* full, size_t count, void* stream) {
    if (!comm || !shard || !full) return fail(GPD_EINVAL, "gpd_allgather_obs: NULL comm/shard/full");
    if (count == 0) return fail(GPD_EINVAL, "gpd_allgather_obs: count must be positive");
    if (int rc = need_rccl("gpd_allgather_obs")) return rc;
    ncclResult_t e = rccl().AllGather(shard, full, count, ncclFloat32, static_cast<ncclComm_t>(comm),
                                      static_cast<hipStream_t>(stream));
    if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather");
    return 0;
}

int gpd_p2p_group(void* comm, const GpdP2P* sends, int32_t n_sends, const GpdP2P* recvs, int32_t n_recvs, void* stream) {
    if (!comm || (n_sends > 0 && !sends) || (n_recvs > 0 && !recvs) || n_sends < 0 || n_recvs < 0)
        return fail(GPD_EINVAL, "gpd_p2p_group: NULL comm / operation list");
    if (int rc = need_rccl("gpd_p2p_group")) return rc;
    Rccl& R = rccl();
    if (!R.GroupStart || !R.GroupEnd || !R.Send || !R.Recv) return fail(GPD_ENOTSUP, "gpd_p2p_group: this RCCL has no ncclSend / ncclRecv / ncclGroup*");
    for (int i = 0; i < n_sends; ++i) if (!sends[i].ptr || sends[i].count <= 0 || sends[i].peer < 0) return fail(GPD_EINVAL, "gpd_p2p_group: bad send operation");
    for (int i = 0; i < n_recvs; ++i) if (!recvs[i].ptr || recvs[i].count <= 0 || recvs[i].peer < 0) return fail(GPD_EINVAL, "gpd_p2p_group: bad receive operation");
    if (n_sends + n_recvs == 0) return 0;
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ncclResult_t e = R.GroupStart();
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupStart");
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n_sends && first == ncclSuccess; ++i)
        first = R.Send(sends[i].ptr, static_cast<size_t>(sends[i].count), ncclFloat32, sends[i].peer, c, st);
    for (int i = 0; i < n_recvs && first == ncclSuccess; ++i)
        first = R.Recv(recvs[i].ptr, static_cast<size_t>(recvs[i].count), ncclFloat32, recvs[i].peer, c, st);
    e = R.GroupEnd();                                  // (always closed, also after a failed enqueue)
    if (first != ncclSuccess) return rccl_fail(first, "ncclSend / ncclRecv");
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupEnd");
    return 0;
}

int gpd_debug_status(uint32_t out[4], int32_t reset, void* stream) {
#ifdef GPD_DEBUG_BOUNDS
    if (!out) return fail(GPD_EINVAL, "gpd_debug_status: NULL out");
    // every unit whose kernels carry checks keeps its own record: the first one that holds a violation is reported, the counts add up
    out[0] = out[1] = out[2] = out[3] = 0u;
    int (*const readers[])(unsigned int*, int, void*) = {gpd_detail_dbg_read_step, gpd_detail_dbg_read_swarm, gpd_detail_dbg_read_policy};
    for (auto rd : readers) {
        unsigned int w[4] = {0u, 0u, 0u, 0u};
        const int e = rd(w, reset, stream);
        if (e != 0) return hip_fail(static_cast<hipError_t>(e), "gpd_debug_status");
        if (out[0] == 0u && w[0] != 0u) { out[0] = w[0]; out[1] = w[1]; out[2] = w[2]; }
        out[3] += w[3];
    }
    return 0;
#else
    (void)out; (void)reset; (void)stream;
    return fail(GPD_ENOTSUP, "gpd_debug_status: this is a release build (no -DGPD_DEBUG_BOUNDS)");
#endif
}

int gpd_clock_probe(double* shader_ghz, double* ns_per_fma, void* stream) {
    if (!shader_ghz) return fail(GPD_EINVAL, "gpd_clock_probe: NULL shader_ghz");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int dev = 0, wall_khz = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev);
    if (e != hipSuccess || wall_khz <= 0) return hip_fail(e, "gpd_clock_probe: wall clock rate");
    unsigned long long* d = nullptr;
    e = hipMalloc(&d, 3 * sizeof(unsigned long long));
    if (e != hipSuccess) return hip_fail(e, "gpd_clock_probe: hipMalloc");
    const int iters = 4000;                                  // x 64 dependent FMAs: ~0.6 ms
    unsigned long long h[2] = {0, 0};
    for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {   // (the second pass runs at the ramped-up clock)
        hipLaunchKernelGGL(gpd_clock_probe_kernel, dim3(256), dim3(kBlock), 0, st, d, iters);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "gpd_clock_probe");
    if (h[1] == 0) return fail(GPD_ENOTSUP, "gpd_clock_probe: the wall clock did not advance");
    const double secs = static_cast<double>(h[1]) / (static_cast<double>(wall_khz) * 1e3);
    *shader_ghz = static_cast<double>(h[0]) / secs * 1e-9;
    if (ns_per_fma) *ns_per_fma = secs * 1e9 / (static_cast<double>(iters) * 64.0);
    return 0;
}


}  // extern "C"

