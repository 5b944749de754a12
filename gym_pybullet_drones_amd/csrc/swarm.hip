// swarm.hip -- ONE aviary of any size: binning, the downwash force kernel with its wake lists, the one-world step
#include "gpd_common.inc"

namespace {

// ------------------------------------------------------------------------------------------------
// Downwash inside ONE aviary of any size (envs/BaseAviary.py:785-811): uniform 2-D grid, counting sort by cell,
// 3x3 neighbourhood search.  Three kernels per physics sub-step (count, scan + scatter, force).
// ------------------------------------------------------------------------------------------------
// cell of a position: the grid is periodic (cells wrap around), so drones that leave the box the grid was laid over
// keep spreading over all cells instead of piling up at its border; far-apart drones that alias into neighbouring
// cells are rejected by the exact distance test
struct DwGrid {            // uniform periodic x-y grid of >= 10 m cells, each cell split into nz height bins (nz = 1: none)
    float inv_cell, x0, y0;
    int nx, ny;
    float z0, inv_zbin;
    int nz;
};

__device__ __forceinline__ int cell_of(float x, float y, const DwGrid& G) {
    // (clamped before the conversion: float -> int is undefined beyond the int range, and a drone flung 1e10 m away by a
    // diverging downwash term must still land in SOME cell -- which one is irrelevant, every candidate pair is distance-tested)
    const float fx = fminf(fmaxf(floorf((x - G.x0) * G.inv_cell), -1.0e9f), 1.0e9f), fy = fminf(fmaxf(floorf((y - G.y0) * G.inv_cell), -1.0e9f), 1.0e9f);
    int cx = static_cast<int>(fx) % G.nx, cy = static_cast<int>(fy) % G.ny;
    cx = cx < 0 ? cx + G.nx : cx;
    cy = cy < 0 ? cy + G.ny : cy;
    return cy * G.nx + cx;
}
// sort key: cell * nz + height bin.  Bin b holds z0 + b/inv_zbin <= z < z0 + (b+1)/inv_zbin; bin 0 also everything below,
// bin nz-1 everything above.  Inside a cell the drones are thereby ordered by height bin, and a group of drones whose lowest
// bin is b can skip every candidate in a bin below b: such a candidate is below all of them (dz < 0), the model ignores it.
__device__ __forceinline__ int key_of(float x, float y, float z, const DwGrid& G) {
    const float fz = fminf(fmaxf(floorf((z - G.z0) * G.inv_zbin), 0.0f), static_cast<float>(G.nz - 1));
    return cell_of(x, y, G) * G.nz + static_cast<int>(fz);
}

// The sort's two passes visit the drones in `visit` order (NULL: 0, 1, 2 ...).  Handing in the previous call's `order`
// makes consecutive lanes fall into the same cell (drones move centimetres per sub-step), and the lanes of a wave that
// share a cell then issue ONE atomic for their whole run instead of one each: ~64 same-address atomics per cell
// become a handful.  run_of(): this lane's run among the wave's consecutive equal cells -> (first lane, length).
__device__ __forceinline__ void run_of(int c, int lane, int& head_lane, int& len) {
    const int prev_c = __shfl_up(c, 1);
    const bool head = lane == 0 || c != prev_c;
    const uint64_t heads = __builtin_amdgcn_ballot_w64(head);
    const uint64_t upto = heads & (((lane == 63) ? 0ull : (2ull << lane)) - 1ull);      // heads at or below this lane
    head_lane = 63 - __builtin_clzll(upto);
    const uint64_t above = heads & ~(((head_lane == 63) ? 0ull : (2ull << head_lane)) - 1ull);   // heads above the run's first lane
    len = (above ? __builtin_ctzll(above) : 64) - head_lane;
}

// VEC: the pass over all drones also writes their [n][20] state vectors (what gpd_state_vectors does) -- a caller that steps
// a swarm needs both after every step, and at this size a launch costs more than the rows.
// where the positions come from: the SoA state block (gpd_downwash_global), or the packed [rows][4] array of a GpdSwarm
struct DwPos { const float* kin; int64_t ld; const float4* pos4; };
__device__ __forceinline__ void dw_pos(const DwPos& S, int d, float& x, float& y, float& z) {
    if (S.pos4) { const float4 p = S.pos4[d]; x = p.x; y = p.y; z = p.z; }
    else { const float4 p = kin_P(S.kin, S.ld)[d]; x = p.x; y = p.y; z = p.z; }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void dwg_count_kernel(const DwPos src, int n, const DwGrid G,
                                                           const int* __restrict__ visit, int* __restrict__ count,
                                                           const GpdState VS, const float* __restrict__ vec_obs12,
                                                           float* __restrict__ vec_out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int c = -1 - lane;                                      // (no drone: a run of its own, no atomic)
    if (i < n) {
        const int d = visit ? visit[i] : i;
        if constexpr (VEC) state20_row(VS, vec_obs12, vec_out, i);      // (row i, not row d: coalesced, the two jobs only share the launch)
        float x, y, z;
        dw_pos(src, d, x, y, z);
        // a drone whose position is no longer finite (the downwash model diverges when two drones pass each other
        // vertically, dz -> 0+) takes no part: it would otherwise alias into cell 0 together with every other such drone
        if (isfinite(x) && isfinite(y) && isfinite(z)) c = key_of(x, y, z, G);
    }
    int head_lane, len;
    run_of(c, lane, head_lane, len);
    if (lane == head_lane && c >= 0) atomicAdd(&count[c], len);
}

// exclusive scan of count[0..cells) into start[0..cells], one workgroup (only for more keys than the scatter kernel scans itself)
__global__ __launch_bounds__(1024) void dwg_scan_kernel(const int* __restrict__ count, int* __restrict__ start, int cells) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (cells + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, cells);
    int s = 0;
    for (int c = lo; c < hi; ++c) s += count[c];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {            // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t == 0 ? 0 : part[t - 1];
    for (int c = lo; c < hi; ++c) { const int k = count[c]; start[c] = run; run += k; }
    if (t == 1023) start[cells] = part[1023];
}

// FUSED (up to kDwScanMax keys): every workgroup turns the counts into start offsets ITSELF, in LDS -- a few thousand integers
// from L2 and eight rounds of a 256-wide scan cost less than the launch of a scan kernel between count and scatter (the
// dependent launches, not the work in them, are what a swarm sub-step is made of); workgroup 0 also
// writes them out for the force kernel.
constexpr int kDwScanMax = 4096;
// what a GpdSwarm binning writes on top of the sort (all NULL / 0 for gpd_downwash_global)
struct DwBinOut {
    int* slot_key;         // [n] sort key of every sorted slot
    int* slot_of;          // [n] or NULL: row -> sorted slot (rows without a finite position: -1)
    int* visit_out;        // [n] or NULL: a second copy of `order`, for the NEXT binning to visit the rows in
    int* list_ok;          // [ceil(n / 64)] or NULL: the groups' wake lists belong to the previous binning: all back to 0
    float4* bin_pos;       // [n] x, y, z of every row at this binning
    float* pos4_w;         // pos4 as floats: the sums and maxima in every rank's meta rows (the last meta_rows of its slab) go back to 0
    float* drift;          // [4] ... and so does the common drift [0..1]; [2]: the margin of the wake lists of THIS binning (below)
    int slab, world, meta_rows;
    int own_lo, own_cnt;   // rows whose force lands in dw_out[row - own_lo]
    float list_delta;      // the largest margin the skin allows
    int list_adapt;        // 1: the margin follows the displacement the interval that ends here has seen
};
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void dwg_scatter_kernel(const DwPos src, int n, const DwGrid G,
                                                             const int* __restrict__ visit, const int* __restrict__ count,
                                                             int* __restrict__ cursor, int* __restrict__ start_g,
                                                             int* __restrict__ order, float4* __restrict__ sorted,
                                                             float* __restrict__ dw_out, const DwBinOut B) {
    __shared__ int lstart[FUSED ? kDwScanMax + 1 : 1];
    __shared__ int part[FUSED ? kBlock : 1];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int keys = G.nx * G.ny * G.nz;
    if constexpr (FUSED) {
        const int t = threadIdx.x;
        const int per = (keys + kBlock - 1) / kBlock;
        const int lo = t * per, hi = min(lo + per, keys);
        int sum = 0;
        for (int k = lo; k < hi; ++k) sum += count[k];
        part[t] = sum;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {      // Hillis-Steele inclusive scan of the 256 partial sums
            const int v = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        int run = t == 0 ? 0 : part[t - 1];
        for (int k = lo; k < hi; ++k) { lstart[k] = run; run += count[k]; }
        if (t == kBlock - 1) lstart[keys] = part[kBlock - 1];
        __syncthreads();
        if (blockIdx.x == 0) for (int k = t; k <= keys; k += kBlock) start_g[k] = lstart[k];
    }
    auto start = [&](int k) { if constexpr (FUSED) return lstart[k]; else return start_g[k]; };
    if (B.pos4_w && blockIdx.x == 0) {                     // every rank bins on the same sub-steps: all displacements restart
        __shared__ float dmax_red[kBlock / 64];
        float d2 = 0.0f;                                   // ... the largest one of the interval that ends here first
        for (int k = threadIdx.x; k < B.world * B.meta_rows; k += kBlock) {
            float* row = B.pos4_w + (static_cast<size_t>(k / B.meta_rows + 1) * B.slab - B.meta_rows + k % B.meta_rows) * 4;
            d2 = fmaxf(d2, row[3]);
            row[1] = 0.0f; row[2] = 0.0f; row[3] = 0.0f;
        }
        if (B.drift) {
            // The wake lists' margin: a pair is listed if it could pass the model's tests after both drones have moved `delta`,
            // and a list is replayed while no drone has.  The skin allows 0.49 (cell - 10 m) -- at 10.5 m, 0.245 m and 40 % more
            // pairs than the exact tests keep; a swarm that moved 5 mm between the last two binnings (relative to its common
            // drift) needs nothing like it.  Three times what the ending interval saw, at least a centimetre, at most what the
            // skin allows; a swarm that then moves further than that sweeps until the next binning (exact either way).
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, off));
            if (lane == 0) dmax_red[threadIdx.x >> 6] = d2;
            __syncthreads();
            if (threadIdx.x == 0) {
                float m = dmax_red[0];
#pragma unroll
                for (int k = 1; k < kBlock / 64; ++k) m = fmaxf(m, dmax_red[k]);
                B.drift[0] = 0.0f; B.drift[1] = 0.0f;
                // (m == 0 exactly: nothing has moved -- the binning right after a pack / reset, where the interval that "ended" says
                // nothing about the one that starts: the full margin, not the 1 cm floor a swarm faster than 0.15 m/s outruns at once)
                B.drift[2] = (B.list_adapt && m > 0.0f) ? fminf(B.list_delta, fmaxf(0.01f, 3.0f * sqrtf(m))) : B.list_delta;
            }
        }
    }
    if (B.list_ok && i < (n + 63) / 64) B.list_ok[i] = 0;
    int c = -1 - lane, d = 0;
    float x = 0.0f, y = 0.0f, z = 0.0f;
    if (i < n) {
        d = visit ? visit[i] : i;
        dw_pos(src, d, x, y, z);
        if (B.bin_pos) B.bin_pos[d] = make_float4(x, y, z, 0.0f);
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            c = key_of(x, y, z, G);
        } else {
            if (B.slot_of) B.slot_of[d] = -1;
            // (see dwg_count_kernel) no force on it, none from it; it keeps a slot behind the sorted drones so that `order`
            // stays a permutation (the next call visits the drones in this order)
            if (d >= B.own_lo && d < B.own_lo + B.own_cnt) dw_out[d - B.own_lo] = 0.0f;
            const int slot = start(keys) + atomicAdd(&cursor[keys], 1);
            order[slot] = d;
            if (B.visit_out) B.visit_out[slot] = d;
        }
    }
    int head_lane, len;
    run_of(c, lane, head_lane, len);
    int base = 0;
    if (lane == head_lane && c >= 0) base = start(c) + atomicAdd(&cursor[c], len);   // one atomic per run of equal keys
    base = __shfl(base, head_lane);
    if (c >= 0) {
        const int slot = base + (lane - head_lane);
        order[slot] = d;
        if (sorted) sorted[slot] = make_float4(x, y, z, __int_as_float(c));
        if (B.slot_key) B.slot_key[slot] = c;
        if (B.slot_of) B.slot_of[d] = slot;
        if (B.visit_out) B.visit_out[slot] = d;
    }
}

// Every drone sweeps the candidates of the 3x3 cells around its own, staged through LDS.  A lane = a drone, and of the ~650
// candidates of a drone ~100 are above it and within 10 m, ~35 close enough for a contribution the fixed-point sum resolves:
// evaluating the model (two reciprocals, an exponential, a conversion: ~36 issue slots) for every candidate, as the first version of this kernel did, spends 95 % of the vector unit on masked-off lanes --
// and with 64 lanes SOME lane nearly always passes, so no wave-level branch ever skips it.  Hence test and evaluation are
// separated, per chunk of 32 candidates:
//   A  the tests only, two candidates per packed instruction, no compare and no branch: with d = (candidate - drone),
//      q = dxy^2 - min(100.01, 80.02 beta(dz)^2) (the 10 m cut-off and the arg < 40 cut, without a division) and the sign
//      bits of q and of -dz ANDed and shifted into a per-lane 32-bit mask (v_and + v_alignbit): 7 slots per candidate, LDS reads
//      at wave-uniform addresses;
//   Q  the set bits become (drone lane, candidate) pairs in a per-wave LDS queue: a DPP prefix sum of the lanes' popcounts
//      gives each lane its slice;
//   E  whenever the queue holds 64 pairs, every lane takes one: drone through ds_bpermute, candidate gathered from the tile, the
//      EXACT tests and the model, and a 64-bit LDS atomic onto the drone's fixed-point sum.  All 64 lanes busy.
// Phase A is a conservative filter (100.01 / 80.02 instead of 100 / 80): it may queue a pair that the exact tests of E then
// reject, never the reverse.  The per-pair arithmetic and the integer sum are those of the first version, so the result is the
// same bit for bit -- except that the first version also kept pairs beyond arg = 40 when alpha > 1e8 N (two drones within
// 28 micrometres of the same height); each such pair now drops less than 4.3e-18 alpha.
constexpr int kDwTile = 1024;     // candidates staged in LDS at a time (3 x 4 KiB, x / y / z planes)
constexpr int kDwChunk = 32;      // candidates per mask word
constexpr int kDwBatches = 4;     // queued batches of 64 pairs that trigger an evaluation (1 .. 8: no measurable difference)
constexpr int kDwQueue = 64 * kDwBatches + kDwChunk * 64;   // pending (< 64 kDwBatches) + everything one chunk can add

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v) {   // v + (v of the lane the DPP control selects; 0 where there is none / the row is masked)
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// Reduction over the 64 lanes of a wave, the result in every lane, without a trip through the LDS crossbar per level (six
// dependent ds_bpermute round trips, ~0.25 us on a path that has nothing else to issue): four DPP steps inside the rows of 16
// lanes -- quad permutes for xor 1 / xor 2; after them the four lanes of a quad hold one value, so the row_half_mirror / row_mirror
// permutes read the same VALUE lane ^ 4 / lane ^ 8 would -- then the four rows' values as scalars (v_readlane), combined
// (r0 . r1) . (r2 . r3) in every lane: bit for bit the ascending xor butterfly (1, 2, 4, .. 32) of a commutative `op`.  All 64 lanes
// must be executing.
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <class Op>
__device__ __forceinline__ float wave_allreduce(float v, Op op) {
    v = op(v, dpp_perm<0xB1>(v));                          // quad_perm [1, 0, 3, 2]
    v = op(v, dpp_perm<0x4E>(v));                          // quad_perm [2, 3, 0, 1]
    v = op(v, dpp_perm<0x141>(v));                         // row_half_mirror
    v = op(v, dpp_perm<0x140>(v));                         // row_mirror
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_max(float v) { return wave_allreduce(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ float wave_sum(float v) { return wave_allreduce(v, [](float a, float b) { return a + b; }); }

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v = dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8      -> inclusive within each row of 16
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15   -> rows 1 and 3 add the total of the row before
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31   -> rows 2 and 3 add the total of the first half
    return v;
}

// One workgroup per 64 CONSECUTIVE drones of the sorted array (lane = drone; all four waves hold the same 64 and split the
// candidates four ways -- the partial sums are integers, the split changes no bit).  Sorted by cell, row-major, 64 consecutive
// drones cover one or a few cells of one grid row (a dense swarm: one or two; a sparse one: many), and their candidates are
// the 2R+1 rows around it, each a CONTIGUOUS stretch of the sorted array from R cells left of the first to R cells right of
// the last: three runs for R = 1, six when the periodic grid wraps.  Every lane is busy whatever the cells hold -- a workgroup
// per CELL, as in the first version, swept all candidates a second time for the few drones beyond the 64th of a cell.  A group
// that straddles the end of a grid row is swept once per row segment, with the lanes of the other segment switched off.
//
// GpdSwarm (DwWorld.pos4 != NULL): the sort is STALE -- `order`, `start` and the keys are those of the last binning, the
// positions (of the group's drones and of the candidates) are the current ones, read through slot -> row -> pos4.  Two drones
// within 10 m of each other now were within 10 m + 2 dmax at the binning (dmax: the largest lateral displacement of any drone
// since, the maximum of the ranks' meta rows), i.e. at most R = ceil((10 + 2 dmax) / cell) cells apart: the search widens to
// (2R+1) x (2R+1) cells and stays exact.  R beyond kDwMaxR (or the grid): the group sweeps every sorted drone.  Only the rows
// own_lo .. own_lo + own_cnt - 1 get a force (the rank's drones); a group without one exits.
constexpr int kDwMaxR = 3, kDwMaxRuns = 2 * (2 * kDwMaxR + 1);
struct DwWorld {
    const float4* pos4;    // current positions by ROW (read through `order`), or NULL: `sorted` holds the current positions by SLOT
                           // (binned in this very call, or kept current by the step kernel: a single rank's world)
    const int* slot_key;   // sort key per slot, or NULL: the key sits in sorted[].w
    const float4* meta;    // the position array whose meta rows hold the ranks' dmax^2, or NULL: no displacement since the binning
    int own_lo, own_cnt;   // rows this launch produces forces for (dw_out[row - own_lo])
    int slab, world, meta_rows;
    float cell;
    float* drift;          // [2] out (the extra workgroup): the mean lateral displacement of all drones, for the NEXT sub-step's step kernel
    float inv_total;       // 1 / drones in the world
    int n_slots;           // entries of `order` / `sorted` / `slot_key` (what a slot index may be clamped to)
};
// Wake lists (GpdSwarm): what survives phase A changes little from one sub-step to the next -- drones move centimetres, the
// model's reach is metres.  The force launch right after a binning (MODE 1, "build") runs phase A with a margin -- a pair is
// kept if it COULD pass the exact tests once both drones have moved up to `delta` in any direction -- and writes every batch it
// evaluates to a per-wave list in HBM.  An entry is 32 bits: 6 bits lane (the drone of the group the pair belongs to) and 26 bits
// the candidate's ABSOLUTE index in the array its current position is read from (the sorted slot in a single-rank world, the row
// of pos4 in a shared one).  The launches until the next binning (MODE 2, "replay") therefore need neither the cell table nor
// the runs nor a staged tile: they read their batches back, GATHER the candidates' current positions straight from memory
// (a group's ~600 candidates are 10 KB that its four waves share: L2 hits after the first touch), and evaluate -- the same exact
// tests and integer sums.  (Round 3's entries were 16 bits, lane + slot of the LDS tile, and a replay launch re-staged the
// tiles like a sweep: of its 14 us, 3 went to dependent set-up loads and 3 to staging before the first pair was evaluated,
// profiles/r03_force_timeline.txt.)  Valid while no drone is further than delta from where it was binned (dmax, the quantity
// the search radius follows; delta is what keeps R = 1: just under half the skin cell - 10 m) and the group's list did not
// overflow; otherwise the launch sweeps as if there were no lists.  MODE 0: no lists (gpd_downwash_global, or none allocated).
constexpr int kDwMaxTiles = 16;                            // (entries of DwLists.nb per wave; [0]: the wave's batches)
struct DwLists {
    uint32_t* list;        // [groups][4 waves][cap * 64] entries, batch-major; 0xffffffff: no pair in this lane of the batch
    unsigned short* nb;    // [groups][4 waves][kDwMaxTiles]: [0] = batches the wave recorded
    int* ok;               // [groups] 1: the group's list is complete for the current binning
    int cap;               // batches per wave
    float delta;           // the displacement the lists allow for
};
template <int MODE>
__global__ __launch_bounds__(kBlock) void dwg_force_kernel(uint32_t* __restrict__ hot_list, unsigned short* __restrict__ hot_nb, int* __restrict__ hot_ok,
                                                           const float4* __restrict__ hot_pos4, const int* __restrict__ order,
                                                           const float4* __restrict__ sorted, const int hot_cap, const int hot_n_slots,
                                                           const GpdParams P, const DwGrid G, const DwWorld Wd_, const DwLists Ls_,
                                                           const int* __restrict__ start, float* __restrict__ dw_out,
                                                           int* __restrict__ cursor) {
    DwLists Ls = Ls_;
    Ls.list = hot_list; Ls.nb = hot_nb; Ls.ok = hot_ok; Ls.cap = hot_cap;
    DwWorld Wd = Wd_;
    Wd.pos4 = hot_pos4; Wd.n_slots = hot_n_slots;
    // one LDS block: the tile's x / y / z planes, then the four waves' queues; (build) the candidates' source indices beside them
    __shared__ __attribute__((aligned(16))) float lds_tile[3 * kDwTile + (kBlock / 64) * kDwQueue / 2];
    __shared__ int tsrc[MODE == 1 ? kDwTile : 1];
    float* const tx = lds_tile;
    float* const ty = lds_tile + kDwTile;
    float* const tz = lds_tile + 2 * kDwTile;
    unsigned short (*const queue)[kDwQueue] = reinterpret_cast<unsigned short (*)[kDwQueue]>(lds_tile + 3 * kDwTile);
    __shared__ unsigned long long sums[kBlock / 64][64];
    __shared__ int run0[kDwMaxRuns], pre[kDwMaxRuns + 1];  // first element of each candidate run; prefix sums of the run lengths
    const int nx = G.nx, ny = G.ny, nz = G.nz;
    const int keys = nx * ny * nz;
    // the sort's per-key counters / cursors are done with: leave them zeroed for the next binning (no memset node per call)
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < 2 * (keys + 1); k += gridDim.x * kBlock) cursor[k] = 0;   // (counts | cursors)
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (Wd.drift && blockIdx.x == gridDim.x - 1) {
        // (an EXTRA workgroup behind those of the sorted array: workgroup 0 used to do this in front of its own share, and with
        // all workgroups resident at once the launch lasts as long as its slowest one)
        // the swarm's common drift for the next sub-step (see swarm_tail): the sum of the workgroups' displacement sums, in an
        // order that depends on nothing but the layout -- every rank arrives at the same two floats
        float sx = 0.0f, sy = 0.0f;
        for (int r = 0; r < Wd.world; ++r) {               // (rank by rank: no division by a run-time meta_rows per element)
            const float4* const mr = Wd.meta + (static_cast<size_t>(r + 1) * Wd.slab - Wd.meta_rows);
            for (int k = threadIdx.x; k < Wd.meta_rows; k += kBlock) { const float4 m = mr[k]; sx += m.y; sy += m.z; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); }
        float* const red = reinterpret_cast<float*>(pre);
        if (lane == 0) { red[2 * wave] = sx; red[2 * wave + 1] = sy; }
        __syncthreads();
        if (threadIdx.x == 0) {
            Wd.drift[0] = (((red[0] + red[2]) + red[4]) + red[6]) * Wd.inv_total;
            Wd.drift[1] = (((red[1] + red[3]) + red[5]) + red[7]) * Wd.inv_total;
        }
        return;
    }
    // Everything the set-up needs from memory is requested HERE, before the first decision that depends on any of it: the
    // workgroup's early exits used to sit between the loads (drones sorted -> slot's row -> positions, keys, maxima, list
    // header: four dependent trips to memory, 3.2 of a replay launch's 15 us, profiles/r03_force_timeline.txt); slot indices
    // are clamped so that a workgroup which is about to leave reads valid memory.
    const int base = 64 * blockIdx.x;
    const int s = base + lane;
    const int sc = min(s, Wd.n_slots - 1);
    auto key_at = [&](int slot) { return Wd.slot_key ? Wd.slot_key[slot] : __float_as_int(sorted[slot].w); };
    auto pos_at = [&](int slot) { return Wd.pos4 ? Wd.pos4[order[slot]] : sorted[slot]; };
    const int sorted_n = start[keys];                      // drones with a finite position (the others: force 0, set by the sort)
    int row_l = order[sc];
    GPD_DBG(row_l >= 0 && row_l < Wd.n_slots, GPD_DBG_SLOT_ROW, row_l); row_l = GPD_DBG_CLAMP(row_l, 0, Wd.n_slots - 1);
    int key_l = key_at(sc);
    GPD_DBG(s >= start[keys] || (key_l >= 0 && key_l < keys), GPD_DBG_SORT_KEY, key_l);
    float4 me_l = Wd.pos4 ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : sorted[sc];
    // dmax^2: the largest of the step kernel's per-workgroup maxima (one meta row each, every rank's)
    // (rank by rank: an index k -> (rank, row) split would be two divisions by a run-time meta_rows per element)
    float d2 = 0.0f;
    const int tot = Wd.world * Wd.meta_rows;
    const bool few = tot <= 1024;                          // few: every wave reads them all (no LDS round, no barrier)
    auto read_maxima = [&]() {
        for (int r = 0; r < Wd.world; ++r) {
            const float4* const mr = Wd.meta + (static_cast<size_t>(r + 1) * Wd.slab - Wd.meta_rows);
            for (int k = few ? lane : static_cast<int>(threadIdx.x); k < Wd.meta_rows; k += few ? 64 : kBlock) d2 = fmaxf(d2, mr[k].w);
        }
    };
    // (many of them -- a world of 10^6 drones, or shared by many ranks -- and they are read behind the early exits instead: 7 of 8
    // groups of a rank of eight hold none of its drones, and were reading 64 rows per lane before they found out)
    if (Wd.meta && few) read_maxima();
    // (replay) whether the group's list is complete, this wave's batches on the first tile
    uint32_t* const my_list = MODE ? Ls.list + (static_cast<size_t>(blockIdx.x) * (kBlock / 64) + wave) * Ls.cap * 64 : nullptr;
    unsigned short* const my_nb = MODE ? Ls.nb + (static_cast<size_t>(blockIdx.x) * (kBlock / 64) + wave) * kDwMaxTiles : nullptr;
    int list_ok = 0, nb0 = 0;
    float delta = Ls.delta;                                // the lists' margin: what gpd_swarm_bin chose for this binning
    if (MODE && Wd.drift) delta = Wd.drift[2];
    // (replay) the wave's first four batches are requested HERE, with everything else the set-up needs: their addresses depend on
    // nothing but the workgroup, the wave and the lane (cap >= 4 is checked by the host; a wave with fewer batches reads stale
    // entries of its own list and never uses them)
    uint32_t e0 = 0xffffffffu, e1 = 0xffffffffu, e2 = 0xffffffffu, e3 = 0xffffffffu;
    if (MODE == 2) {
        list_ok = Ls.ok[blockIdx.x];
        nb0 = my_nb[0];
        e0 = my_list[lane]; e1 = my_list[64 + lane]; e2 = my_list[128 + lane]; e3 = my_list[192 + lane];
    }
    // (the loads above are all in flight; this is where they are waited for, together)
    asm volatile("" : "+v"(row_l), "+v"(key_l), "+v"(me_l.x), "+v"(me_l.y), "+v"(me_l.z), "+v"(d2), "+v"(list_ok), "+v"(nb0), "+v"(delta));
    if (MODE == 2) asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
    list_ok = __builtin_amdgcn_readfirstlane(list_ok); nb0 = __builtin_amdgcn_readfirstlane(nb0);
    if (base >= sorted_n) return;
    const bool have = s < sorted_n;
    const int my_row = have ? row_l : -1;
    const bool own = have && my_row >= Wd.own_lo && my_row < Wd.own_lo + Wd.own_cnt;
    if (__builtin_amdgcn_ballot_w64(own) == 0) return;     // (the four waves hold the same 64 slots: the whole workgroup leaves)
    unsigned short* const my_queue = queue[wave];
    unsigned long long* const my_sums = sums[wave];
    const float kr = 0.25f * P.prop_radius;
    const float cut = 8.9454f;                             // sqrt(80.02): arg < 40  <=>  dxy^2 < 80 beta^2
    const fp2 b1 = splat(-P.dw_coeff[1] * cut), b0 = splat(P.dw_coeff[2] * cut);
    float4 me = make_float4(0.0f, 0.0f, 3.0e38f, 0.0f);
    if (own) me = Wd.pos4 ? Wd.pos4[my_row] : me_l;
    // (key -> cell: nz = 1 for every grid SwarmAviary builds -- no division by a run-time value then)
    auto cell_of = [&](int key) { return nz == 1 ? key : key / nz; };
    const int my_cell = own ? cell_of(key_l) : -1;
    // search radius in cells
    int R = 1;
    if (Wd.meta) {
        // (many meta rows -- a world of 10^6 drones, or one shared by many ranks: gpd_swarm_step's own reduction left every
        // rank's maximum in the w of the rank's FIRST meta row, dwg_reduce_meta_kernel; one value per rank is read here.  Every
        // workgroup reading all 4 097 rows of a 1M-drone world was 1 GB of L2 traffic per force launch and a third of a replay
        // launch's vector instructions)
        if (!few) for (int r = lane; r < Wd.world; r += 64) d2 = fmaxf(d2, Wd.meta[static_cast<size_t>(r + 1) * Wd.slab - Wd.meta_rows].w);
        d2 = wave_max(d2);
        if (d2 > 0.0f) {                                   // (rounded up: a wider search is still exact, a narrower one is not)
            const float reach = 10.0f + 2.0002f * sqrtf(d2) + 2.0e-6f;
            const float cells = ceilf(reach / Wd.cell * 1.000001f);
            R = cells < 1.0e6f ? static_cast<int>(cells) : (1 << 20);          // (also catches inf)
        }
        R = __builtin_amdgcn_readfirstlane(R);
    }
    // replay this group's wake list?  (uniform for the workgroup)
    // (0.98: dmax <= 0.99 delta -- the build's margin test is evaluated in fp32, a pair exactly on its boundary must not matter)
    const bool replay = MODE == 2 && R == 1 && d2 <= 0.98f * (delta * delta) && list_ok != 0;
    // the group's first and last cell: the keys of its first and last sorted slot, which two of its lanes hold already
    const int c_first = cell_of(__builtin_amdgcn_readlane(key_l, 0));
    const int c_last = cell_of(__builtin_amdgcn_readlane(key_l, __builtin_amdgcn_readfirstlane(min(63, sorted_n - 1 - base))));
    int lb = 0;                                            // (build) batches recorded so far
    // (build) the list is complete so far -- and only a build on the binning's own positions counts (a caller that builds later
    // gets no lists rather than lists whose margin was measured from somewhere else)
    bool rec_ok = R == 1 && d2 == 0.0f;
    const float m2 = MODE == 1 ? 2.0f * delta : 0.0f;      // (build) what two drones can have closed in on each other
    const float c2 = MODE == 1 ? cut * fabsf(P.dw_coeff[1]) * m2 : 0.0f;      // ... and what that adds to sqrt(80.02) |beta|
    const bool sweep_all = R > kDwMaxR;                    // the group's candidates: every sorted drone, one run
    const int nrows = sweep_all ? 0 : min(2 * R + 1, ny);
    my_sums[lane] = 0ull;                                  // sum of contributions in units of 2^-30 N: order-independent
    // one pair: drone `dl` of the group under the candidate at (cx, cy, cz) -- the exact tests and the model (:798-808), added to
    // the drone's sum (called by ALL lanes -- ds_bpermute reads nothing from a lane that is masked off -- with `valid` false
    // where there is no pair)
    auto pair_model = [&](int dl, bool valid, float cx, float cy, float cz) {
        const float px = __shfl(me.x, dl), py = __shfl(me.y, dl), pz = __shfl(me.z, dl);
        const float dz = cz - pz;
        const float ddx = cx - px, ddy = cy - py;
        const float dxy2 = fmaf(ddy, ddy, ddx * ddx);
        if (valid && dz > 0.0f && dxy2 < 100.0f) {     // dz > 0 and dxy < 10 m  (:800-801)
            const float ratio = kr * fast_rcp(dz);
            const float alpha = P.dw_coeff[0] * (ratio * ratio);
            const float beta = fmaf(P.dw_coeff[1], dz, P.dw_coeff[2]);
            const float ib = fast_rcp(beta);
            const float arg = 0.5f * (dxy2 * (ib * ib));
            if (arg < 40.0f) {                         // exp(-40) = 4e-18: below the 2^-31 N the sum resolves
                const float sc = (alpha * fast_exp(-arg)) * 1073741824.0f;
                // (< 4 N, i.e. unless two drones are centimetres apart: one conversion instead of the 14-instruction
                // float -> int64 sequence; the same integer either way)
                // (marked unlikely: without it the 14-instruction conversion is laid out as the fall-through and the usual case
                // pays two taken branches per batch to get around it)
                unsigned long long v;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(sc >= 0.0f && sc < 4.0e9f)) != 0, 0)) {
                    asm volatile("; a contribution of 4 N or more" ::: "memory");      // (keeps this a branch, not a select)
                    v = static_cast<unsigned long long>(__float2ll_rn(sc));
                } else {
                    v = static_cast<unsigned long long>(__float2uint_rn(sc));
                }
                atomicAdd(&my_sums[dl], v);
            }
        }
    };
    if (MODE == 2 && replay) {
        // REPLAY: the wave's recorded batches, four at a time, in a three-deep software pipeline -- the entries of batches b+8..
        // are requested while the candidates of b+4.. are gathered and b.. is evaluated (loads only in this loop: the waits the
        // compiler derives are exact counts).  An empty lane (0xffffffff) gathers index 0 and evaluates nothing.
        GPD_DBG(nb0 >= 0 && nb0 <= Ls.cap, GPD_DBG_LIST_COUNT, nb0);
        const int nbw = GPD_DBG_CLAMP(nb0, 0, Ls.cap);
        const uint32_t* const lp = my_list + lane;
        // A launch ends with its slowest workgroup, and that is the one with the most pairs (profiles/r04_swarm_force_timeline.txt:
        // 9.1 us against a median of 6.6; all workgroups are resident at once, nothing fills in behind it).  A wave with many batches
        // asks the SIMD's arbiter for priority over the three it shares the SIMD with: they have slack, it has none.
        // (thresholds 12 / 16 / 20 and 11 / 13 / 16 measure the same, profiles/r04_ab_swarm_replay_priority.txt; a scene where every
        // wave is above them gives every wave the same priority: nothing gained, nothing lost)
        if (nbw >= 12) {
            if (nbw >= 20) __builtin_amdgcn_s_setprio(3);
            else if (nbw >= 16) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        }
        auto cand = [&](uint32_t e) {
            // (clamped: the first four entries are read before the wave knows how many batches it has -- whatever a list holds,
            // the gather stays inside the array)
            GPD_DBG(e == 0xffffffffu || (e & 0x03ffffffu) < static_cast<uint32_t>(Wd.n_slots), GPD_DBG_LIST_ENTRY, e & 0x03ffffffu);
            const uint32_t idx = min(e == 0xffffffffu ? 0u : (e & 0x03ffffffu), static_cast<uint32_t>(Wd.n_slots - 1));
            return Wd.pos4 ? Wd.pos4[idx] : sorted[idx];
        };
        // (no branch around the load: beyond the wave's batches the last one is read again and discarded)
        auto entry = [&](int b) { const uint32_t e = lp[static_cast<size_t>(min(b, max(nbw - 1, 0))) * 64]; return b < nbw ? e : 0xffffffffu; };
        uint32_t f0 = entry(4), f1 = entry(5), f2 = entry(6), f3 = entry(7);
        float4 p0 = cand(e0), p1 = cand(e1), p2 = cand(e2), p3 = cand(e3);
        if (nbw < 4) { if (nbw < 1) e0 = 0xffffffffu; if (nbw < 2) e1 = 0xffffffffu; if (nbw < 3) e2 = 0xffffffffu; e3 = 0xffffffffu; }
        for (int b = 0; b < nbw; b += 4) {
            const uint32_t g0 = entry(b + 8), g1 = entry(b + 9), g2 = entry(b + 10), g3 = entry(b + 11);
            const float4 q0 = cand(f0), q1 = cand(f1), q2 = cand(f2), q3 = cand(f3);
            pair_model(static_cast<int>(e0 >> 26), e0 != 0xffffffffu, p0.x, p0.y, p0.z);
            pair_model(static_cast<int>(e1 >> 26), e1 != 0xffffffffu, p1.x, p1.y, p1.z);
            pair_model(static_cast<int>(e2 >> 26), e2 != 0xffffffffu, p2.x, p2.y, p2.z);
            pair_model(static_cast<int>(e3 >> 26), e3 != 0xffffffffu, p3.x, p3.y, p3.z);
            e0 = f0; e1 = f1; e2 = f2; e3 = f3; p0 = q0; p1 = q1; p2 = q2; p3 = q3;
            f0 = g0; f1 = g1; f2 = g2; f3 = g3;
        }
    } else
    for (int cs = c_first; cs <= c_last;) {                // row segments of the group's cells (nearly always one)
        const int cy = cs / nx;
        const int ce = sweep_all ? c_last : min(c_last, cy * nx + nx - 1);
        const int cxa = cs - cy * nx, w = ce - cs + 1 + 2 * R;     // columns cxa - R .. cxb + R, periodic
        const bool active = own && my_cell >= cs && my_cell <= ce;
        const float mez = active ? me.z : 3.0e38f;         // (switched off: nothing is above it)
        const bool fast = R == 1 && !sweep_all;            // the usual case: three rows, six runs, all in registers, no barrier
        int run0r[6], prer[7];
        const int nruns = sweep_all ? 1 : 2 * nrows;
        if (fast) {
            prer[0] = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int gy = cy + r - 1;                       // (one period at most: no modulo)
                gy = gy < 0 ? gy + ny : gy >= ny ? gy - ny : gy;
                const int row = gy * nx;
                int a0, a1, b1c;                           // run A: cells a0 .. a1; run B (the wrapped part): cells 0 .. b1c, or empty
                if (w >= nx) { a0 = 0; a1 = nx - 1; b1c = -1; }
                else {
                    const int a = cxa > 0 ? cxa - 1 : nx - 1;
                    a0 = a; a1 = min(a + w - 1, nx - 1);
                    b1c = a + w - 1 - nx;                  // (< 0: no wrap)
                }
                run0r[2 * r] = start[(row + a0) * nz];
                prer[2 * r + 1] = prer[2 * r] + (start[(row + a1 + 1) * nz] - run0r[2 * r]);
                run0r[2 * r + 1] = start[row * nz];
                prer[2 * r + 2] = prer[2 * r + 1] + (b1c >= 0 ? start[(row + b1c + 1) * nz] - run0r[2 * r + 1] : 0);
            }
        } else {
        __syncthreads();                                   // (the previous segment's tiles are done with run0 / pre)
        if (static_cast<int>(threadIdx.x) < nruns) {       // run q: row q >> 1, part A (up to the end of the row) or B (the wrapped rest)
            const int q = threadIdx.x;
            int first = 0, len = sorted_n;
            if (!sweep_all) {
                const int r = q >> 1;
                const int gy = (2 * R + 1 < ny) ? ((cy - R + r) % ny + ny) % ny : r;        // (every row once when the rows wrap)
                const int row = gy * nx;
                int a0, a1, b1c;                           // run A: cells a0 .. a1; run B: cells 0 .. b1c (< 0: none)
                if (w >= nx) { a0 = 0; a1 = nx - 1; b1c = -1; }
                else {
                    const int a = ((cxa - R) % nx + nx) % nx;
                    a0 = a; a1 = min(a + w - 1, nx - 1);
                    b1c = a + w - 1 - nx;
                }
                if ((q & 1) == 0) { first = start[(row + a0) * nz]; len = start[(row + a1 + 1) * nz] - first; }
                else { first = start[row * nz]; len = b1c >= 0 ? start[(row + b1c + 1) * nz] - first : 0; }
            }
            run0[q] = first;
            pre[q + 1] = len;                              // (lengths for now, prefix sums below)
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            pre[0] = 0;
            for (int q = 0; q < nruns; ++q) pre[q + 1] += pre[q];
        }
        __syncthreads();
        }
        const int total = fast ? prer[6] : pre[nruns];
        const fp2 mx = splat(me.x), my = splat(me.y), mz = splat(mez);
        int pending = 0;                                   // pairs in this wave's queue (wave-uniform)
        // one queued pair of the tile in LDS (queue entry: 6 bits lane, 10 bits candidate slot of the tile)
        auto evaluate = [&](unsigned e, bool valid) {
            const int dl = static_cast<int>(e >> 10), ci = static_cast<int>(e & 1023u);
            if (MODE == 1) {                               // build: the batch goes to the list as it is evaluated
                if (rec_ok && lb < Ls.cap)
                    my_list[static_cast<size_t>(lb) * 64 + lane] = valid ? ((static_cast<uint32_t>(dl) << 26) | static_cast<uint32_t>(tsrc[ci])) : 0xffffffffu;
                else rec_ok = false;
                ++lb;
            }
            pair_model(dl, valid, tx[ci], ty[ci], tz[ci]);
        };
        // (a tile holds kDwTile - 1 candidates: the entry "lane 63, slot 1023" never occurs and 0xffff can mark an empty lane)
        for (int v0 = 0; v0 < total;) {
            const int cnt = min(kDwTile - 1, total - v0);
            const int chunks = (cnt + kDwChunk - 1) / kDwChunk;
            auto source = [&](int v) {                     // position in the concatenated list -> run q, element src
                int src;
                if (fast) {
                    src = run0r[0] + v;
#pragma unroll
                    for (int q = 1; q < 6; ++q) src = (v >= prer[q]) ? run0r[q] + (v - prer[q]) : src;
                } else {
                    src = run0[0] + v;
                    for (int q = 1; q < nruns; ++q) src = (v >= pre[q]) ? run0[q] + (v - pre[q]) : src;
                }
                return src;
            };
            __syncthreads();
            for (int j = threadIdx.x; j < chunks * kDwChunk; j += kBlock) {
                float4 o = make_float4(0.0f, 0.0f, -3.0e38f, 0.0f);        // (padding of the last chunk: below everything)
                if (j < cnt) {
                    // (build) where a replay launch will find this candidate's current position: its row of pos4 in a shared
                    // world, its sorted slot in a single-rank one
                    const int sl = source(v0 + j);
                    const int src = Wd.pos4 ? order[sl] : sl;
                    o = Wd.pos4 ? Wd.pos4[src] : sorted[src];
                    if (MODE == 1) tsrc[j] = src;
                }
                tx[j] = o.x; ty[j] = o.y; tz[j] = o.z;
            }
            __syncthreads();
            v0 += cnt;
            for (int ch = wave; ch < chunks; ch += kBlock / 64) {       // this wave's share of the candidates
                const int j0 = ch * kDwChunk;
                uint32_t mask = 0;                                     // candidate j0 + j of the chunk -> bit 31 - j
#pragma unroll 8
                for (int j = 0; j < kDwChunk; j += 2) {
                    const fp2 ox = *reinterpret_cast<const fp2*>(&tx[j0 + j]), oy = *reinterpret_cast<const fp2*>(&ty[j0 + j]),
                              oz = *reinterpret_cast<const fp2*>(&tz[j0 + j]);
                    const fp2 ddx = ox - mx, ddy = oy - my, below = mz - oz;       // below < 0: the candidate is above
                    const fp2 dd2 = fma2(ddy, ddy, ddx * ddx);
                    const fp2 bs = fma2(b1, below, b0);                            // sqrt(80.02) beta(dz)
                    if (MODE == 1) {
                        // with the margin: the candidate may come up to m2 closer in height and in the plane, |beta| may grow by
                        // DW2 m2:  dxy < min(10, sqrt(80.02) (|beta| + DW2 m2)) + m2  and  dz > -m2
                        const float tx_ = fminf(fabsf(bs.x) + c2, 10.0005f) + m2, ty_ = fminf(fabsf(bs.y) + c2, 10.0005f) + m2;
                        const fp2 q = dd2 - fp2{tx_ * tx_, ty_ * ty_};
                        const fp2 bm = below - splat(m2);
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.x) & __float_as_uint(bm.x), 31);
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.y) & __float_as_uint(bm.y), 31);
                    } else {
                        const fp2 lim = bs * bs;
                        const fp2 q = dd2 - fp2{fminf(lim.x, 100.01f), fminf(lim.y, 100.01f)};
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.x) & __float_as_uint(below.x), 31);
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.y) & __float_as_uint(below.y), 31);
                    }
                }
                const int mine = __builtin_popcount(mask);
                const int upto = wave_inclusive_scan(mine);
                int pos = pending + upto - mine;
                while (mask != 0u) {                                   // this lane's pairs -> its slice of the queue
                    const int j = __builtin_clz(mask);
                    mask &= ~(0x80000000u >> j);
                    my_queue[pos++] = static_cast<unsigned short>((lane << 10) | (j0 + j));
                }
                pending += __builtin_amdgcn_readlane(upto, 63);
                __builtin_amdgcn_wave_barrier();
                // Full batches once kDwBatches of them wait: a lane's pairs sit next to each other in the queue, so a batch of 64
                // CONSECUTIVE pairs would hit each drone's sum ~5 times in one ds_add_u64 (the LDS serialises those); batch b takes
                // every nb-th pair instead, taken from the end (the < 64 oldest stay where they are).
                if (pending >= 64 * kDwBatches) {
                    const int nb = pending >> 6;
                    pending &= 63;
                    for (int b = 0; b < nb; ++b) evaluate(my_queue[pending + b + lane * nb], true);
                }
                __builtin_amdgcn_wave_barrier();
            }
            {                                                          // (the tile is about to be replaced: everything goes)
                const int nb = (pending + 63) >> 6;
                for (int b = 0; b < nb; ++b) { const int k = b + lane * nb; evaluate(k < pending ? my_queue[k] : 0u, k < pending); }
            }
            pending = 0;
        }
        cs = ce + 1;
    }
    if (MODE == 1) {                                       // the group's list counts only if all four waves completed theirs
        if (lane == 0) my_nb[0] = static_cast<unsigned short>(rec_ok ? lb : 0);      // the wave's batches (< cap <= 65535)
        int* const okf = reinterpret_cast<int*>(pre);
        __syncthreads();
        if (lane == 0) okf[wave] = rec_ok ? 1 : 0;
        __syncthreads();
        if (threadIdx.x == 0) Ls.ok[blockIdx.x] = okf[0] & okf[1] & okf[2] & okf[3];
    }
    __syncthreads();
    if (wave == 0 && own) {                                // add up the four waves' shares
        unsigned long long sum = 0;
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) sum += sums[k][lane];
        dw_out[my_row - Wd.own_lo] = -static_cast<float>(static_cast<double>(static_cast<long long>(sum)) * (1.0 / 1073741824.0));
    }
}

// ------------------------------------------------------------------------------------------------
// GpdSwarm: the physics sub-step of ONE world's drones (single-drone lanes of env_step, all add-on terms possible, the downwash
// force from state.dw_force), which also hands the next launch what it needs: the drone's new position in the packed array the
// force kernel (and, across ranks, the all-gather) reads, how far it has moved since the last binning, its state vector.
// ------------------------------------------------------------------------------------------------
struct SwarmOut {
    float4* pos4_own;          // pos4 + rank * slab: row i = own drone i
    const float4* bin_pos_own; // bin_pos + rank * slab
    float* meta_own;           // the rank's first meta row as floats (workgroup b's row: 4 b floats on)
    const int* slot_of_own;    // slot_of + rank * slab, or NULL
    float4* pos_sorted;        // [n_rows] current positions by sorted slot (with slot_of)
    const float* drift;        // [2] the swarm's common lateral drift since the binning, as of the previous sub-step
    float* vec_out;            // [n][20] or NULL
};
// The drone's new position for the next launch (by row; by sorted slot when this rank holds the whole world), and how far it is
// from where it was binned -- RELATIVE TO THE SWARM'S COMMON DRIFT: what the stale cell order and the wake lists tolerate is a
// change of the drones' positions relative to each other; a translation all drones share (a swarm in transit) changes no pair.
// `drift` is the mean lateral displacement of all drones as of the previous sub-step (the force launch in between computes it
// from the sums below and every workgroup of every rank reads the same two floats: ANY common vector keeps the bound exact, a
// good one keeps it small).  The workgroup's largest residual and its displacement sums go to the workgroup's own meta row --
// plain stores (1024 wavefronts updating ONE word with atomics cost 13 us: atomics on one address are served one after the
// other; the force kernel, which needs the maximum over all of them, reads a few hundred words instead).
struct SwarmIn { float4 b; float cx, cy; int slot; };     // what the tail needs from memory, requested with the state's loads
__device__ __forceinline__ SwarmIn swarm_head(const SwarmOut& O, bool active, uint32_t n) {
    SwarmIn I;
    I.cx = O.drift[0]; I.cy = O.drift[1];
    I.b = active ? O.bin_pos_own[n] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    I.slot = (active && O.slot_of_own) ? O.slot_of_own[n] : -1;
    return I;
}
__device__ __forceinline__ void swarm_tail(const SwarmOut& O, const SwarmIn& I, bool active, uint32_t n, float px, float py, float pz) {
    __shared__ float wg_red[kBlock / 64][3];
    const float cx = I.cx, cy = I.cy;
    float d2 = 0.0f, sx = 0.0f, sy = 0.0f;
    if (active) {
        O.pos4_own[n] = make_float4(px, py, pz, 0.0f);
        if (I.slot >= 0) O.pos_sorted[I.slot] = make_float4(px, py, pz, 0.0f);
        const float4 b = I.b;
        const float dx = px - b.x, dy = py - b.y, dz = pz - b.z;
        const float ex = dx - cx, ey = dy - cy;
        d2 = fmaf(dz, dz, fmaf(ey, ey, ex * ex));
        const bool fin = d2 == d2 && d2 < 3.0e38f;             // (a drone without a finite position takes no part in the sort)
        // (NaN AND +-inf: a non-finite position fails every pair test, it needs no search radius -- an infinite d2 would be the
        // workgroup's maximum, push R beyond kDwMaxR and turn every group of every rank into an O(N^2) sweep until the next binning)
        d2 = fin ? d2 : 0.0f;
        sx = fin ? dx : 0.0f; sy = fin ? dy : 0.0f;
    }
    d2 = wave_max(d2); sx = wave_sum(sx); sy = wave_sum(sy);
    if ((threadIdx.x & 63) == 0) { wg_red[threadIdx.x >> 6][0] = d2; wg_red[threadIdx.x >> 6][1] = sx; wg_red[threadIdx.x >> 6][2] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = wg_red[0][0], tx = wg_red[0][1], ty = wg_red[0][2];
#pragma unroll
        for (int k = 1; k < kBlock / 64; ++k) { m = fmaxf(m, wg_red[k][0]); tx += wg_red[k][1]; ty += wg_red[k][2]; }
        float* row = O.meta_own + 4 * blockIdx.x;           // (x stays non-finite: "no drone in this row")
        row[1] = tx; row[2] = ty; row[3] = m;
    }
}
// (the argument list starts with what the load section needs: kernarg preload, as for gpd_step_kernel)
template <int ACT>
__global__ __launch_bounds__(kBlock) void gpd_swarm_step_kernel(float* __restrict__ hot_kin, const float* __restrict__ action,
                                                                float* __restrict__ hot_last_rpm, float* __restrict__ hot_dw_force,
                                                                const float4* __restrict__ hot_bin_pos_own, const int* __restrict__ hot_slot_of_own,
                                                                const uint32_t hot_ld, const int32_t hot_n,
                                                                const GpdParams P, const GpdState S_, const GpdStepCfg C_,
                                                                float* __restrict__ obs12, const SwarmOut O_) {
    GpdState S = S_;
    S.kin = hot_kin; S.last_rpm = hot_last_rpm; S.dw_force = hot_dw_force; S.ld = hot_ld;
    GpdStepCfg C = C_;
    C.num_envs = hot_n;
    SwarmOut O = O_;
    O.bin_pos_own = hot_bin_pos_own; O.slot_of_own = hot_slot_of_own;
    const uint32_t N = static_cast<uint32_t>(C.num_envs);
    const uint32_t n_raw = blockIdx.x * kBlock + threadIdx.x;
    Lane L;
    L.tid = threadIdx.x;
    L.active = n_raw < N;
    L.n = L.active ? n_raw : 0u;
    L.le = L.tid; L.base = L.tid; L.d = 0; L.env = L.n; L.shfl = false;
    const uint32_t flags = C.physics_flags;
    Carry c;
    float tgx, tgy, tgz, ip[7];
    const float4 act = load_action<4>(action, L.n);
    const SwarmIn I = swarm_head(O, L.active, L.n);        // (one trip to memory with the state's loads instead of one behind the step)
    load_carry<false, true>(S, C, flags, L, S.kin, S.kin, c, tgx, tgy, tgz, ip);     // (no task, no reset: readable dummies)
    c.roll = c.pitch = c.yaw = 0.0f;
    StepOut out;
    env_step<false, true, false, 4, ACT, true>(P, C, flags, 1, L, act, tgx, tgy, tgz, false, S.kin, ip[0], ip[1], ip[2], ip[3], ip[4],
                                               ip[5], ip[6], nullptr, nullptr, c, out);
    swarm_tail(O, I, L.active, L.n, c.k.px, c.k.py, c.k.pz);
    // observation rows (48 B) and state vectors (80 B, BaseAviary._getDroneStateVector, envs/BaseAviary.py:541-561): a lane's row
    // is a strided piece of cache lines, a wave's 64 rows are one contiguous block -- transposed through LDS and stored as
    // 1 KiB bursts (the wave's own LDS instructions execute in order: no barrier between its writes and its reads)
    __shared__ __attribute__((aligned(16))) float4 sh_rows[kBlock * 5];
    const int wave0 = threadIdx.x & ~63, lane = threadIdx.x & 63;
    const uint32_t n0 = n_raw - static_cast<uint32_t>(lane);               // first drone of this wave
    const uint32_t rows = __builtin_amdgcn_readfirstlane(n0 < N ? (N - n0 < 64u ? N - n0 : 64u) : 0u);
    const Kin& k = c.k;
    auto burst = [&](float* dst_rows, auto f4c) {          // F4 float4 per row; the wave's rows start at dst_rows
        constexpr int F4 = decltype(f4c)::value;
        float4* patch = sh_rows + wave0 * 5;
        __builtin_amdgcn_wave_barrier();
        if (rows == 64u) {
            // every row of the wave exists (all waves but the grid's last): F4 LDS reads in one run, then F4 stores -- with a
            // test around every store each read is waited for before its store and the next read starts behind it, F4 LDS
            // round trips in a row (profiles/r04_swarm_step_timeline.txt: the stores are issued 0.14 us sooner this way; the
            // memory pipeline paces the rest, 0.08 us of a 3.8 us workgroup is what it is worth)
            float4 v[F4];
#pragma unroll
            for (int j = 0; j < F4; ++j) v[j] = patch[j * 64 + lane];
#pragma unroll
            for (int j = 0; j < F4; ++j) {
                f4v wv = {v[j].x, v[j].y, v[j].z, v[j].w};
                __builtin_nontemporal_store(wv, reinterpret_cast<f4v*>(dst_rows) + (j * 64 + lane));
            }
        } else {
#pragma unroll
            for (int j = 0; j < F4; ++j) {
                const int idx = j * 64 + lane;
                const float4 v = patch[idx];
                if (static_cast<uint32_t>(idx) < rows * F4) {
                    f4v wv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(wv, reinterpret_cast<f4v*>(dst_rows) + idx);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    {
        float4* mine = sh_rows + wave0 * 5 + lane * 3;
        mine[0] = make_float4(out.o[0], out.o[1], out.o[2], out.o[3]);
        mine[1] = make_float4(out.o[4], out.o[5], out.o[6], out.o[7]);
        mine[2] = make_float4(out.o[8], out.o[9], out.o[10], out.o[11]);
        burst(obs12 + static_cast<size_t>(n0) * 12, std::integral_constant<int, 3>{});
    }
    if (O.vec_out) {
        float4* mine = sh_rows + wave0 * 5 + lane * 5;
        mine[0] = make_float4(k.px, k.py, k.pz, k.qx);
        mine[1] = make_float4(k.qy, k.qz, k.qw, out.o[3]);
        mine[2] = make_float4(out.o[4], out.o[5], k.vx, k.vy);
        mine[3] = make_float4(k.vz, out.o[9], out.o[10], out.o[11]);
        mine[4] = make_float4(c.l0, c.l1, c.l2, c.l3);
        burst(O.vec_out + static_cast<size_t>(n0) * 20, std::integral_constant<int, 5>{});
    }
    if (!L.active) return;
    store_carry<false>(S, L, c);
}

// The rank's largest squared displacement (the maximum of its workgroups' maxima, one per meta row) -> the w of its FIRST meta
// row.  Launched behind gpd_swarm_step_kernel when the world has too many meta rows for every force workgroup to read them all
// (world_size x meta_rows > 1024); the kernel boundary is the synchronisation (an in-kernel "last workgroup reduces" would need an
// agent-scope release per workgroup: microseconds each on this part, profiles/r04_doorbell_step_server.txt).
__global__ __launch_bounds__(kBlock) void dwg_reduce_meta_kernel(float* __restrict__ meta_own, int rows) {
    __shared__ float red[kBlock / 64];
    float m = 0.0f;
    for (int k = threadIdx.x; k < rows; k += kBlock) m = fmaxf(m, meta_own[4 * k + 3]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < kBlock / 64; ++k) m = fmaxf(m, red[k]);
        meta_own[3] = m;
    }
}

// after a reset / an outside change of the state: the rank's slab of pos4 from state.kin -- its drones, the rows without one
// (non-finite), the meta row (dmax^2 = 0) -- and, on request, the state vectors
__global__ __launch_bounds__(kBlock) void gpd_swarm_pack_kernel(const GpdState S, int n, int slab, float4* __restrict__ pos4_own,
                                                                const float* __restrict__ obs12, float* __restrict__ vec_out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= slab) return;
    const float nan = __int_as_float(0x7fc00000);
    if (i < n) {
        const float4 p = kin_P(S.kin, S.ld)[i];
        pos4_own[i] = make_float4(p.x, p.y, p.z, 0.0f);
        if (vec_out) state20_row(S, obs12, vec_out, i);
    } else {
        pos4_own[i] = make_float4(nan, 0.0f, 0.0f, 0.0f);     // (a non-finite x is what says "no drone"; meta rows: sums and maximum 0)
    }
}

}  // namespace

GPD_DBG_READER(gpd_detail_dbg_read_swarm)

extern "C" {

int gpd_sizeof_swarm(void) { return static_cast<int>(sizeof(GpdSwarm)); }

int gpd_downwash_global(const GpdParams* params, const float* kin, int64_t ld, int32_t n, float cell, float x0,
                        float y0, int32_t nx, int32_t ny, float z0, float zbin, int32_t nz, const int32_t* visit_order,
                        int32_t* cell_count, int32_t* cell_start, int32_t* order, float* sorted_xyzc, float* dw_out,
                        const GpdState* vec_state, const float* vec_obs12, float* vec_out, void* stream) {
    if (!params || !kin || !cell_count || !cell_start || !order || !sorted_xyzc || !dw_out)
        return fail(GPD_EINVAL, "gpd_downwash_global: NULL argument");
    if (n <= 0 || ld < n) return fail(GPD_EINVAL, "gpd_downwash_global: need 0 < n <= ld");
    if ((reinterpret_cast<uintptr_t>(kin) & 15u) != 0) return fail(GPD_EINVAL, "gpd_downwash_global: kin must be 16-byte aligned (plane P is read as float4)");
    if (visit_order == order) return fail(GPD_EINVAL, "gpd_downwash_global: visit_order must not alias order (ping-pong two buffers)");
    if (!(cell >= 10.0f)) return fail(GPD_EINVAL, "gpd_downwash_global: cell must be >= 10 m (the model's lateral cut-off)");
    if (nz < 1 || nz > kBlock || (nz > 1 && !(zbin > 0.0f))) return fail(GPD_EINVAL, "gpd_downwash_global: need 1 <= nz <= 256 and zbin > 0");
    if (nx < 3 || ny < 3 || static_cast<int64_t>(nx) * ny * nz > 65536)
        return fail(GPD_ERANGE, "gpd_downwash_global: need nx, ny >= 3 (periodic 3x3 search) and nx*ny*nz <= 65536");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cells = nx * ny, keys = cells * nz;
    const DwGrid G{1.0f / cell, x0, y0, nx, ny, z0, nz > 1 ? 1.0f / zbin : 0.0f, nz};
    hipError_t e;
    const dim3 grid(static_cast<unsigned>((n + kBlock - 1) / kBlock));
    const DwPos src{kin, ld, nullptr};
    if (vec_out) {
        if (!vec_state || !vec_state->kin || !vec_obs12) return fail(GPD_EINVAL, "gpd_downwash_global: vec_out needs vec_state and vec_obs12");
        if (vec_state->ld < n) return fail(GPD_EINVAL, "gpd_downwash_global: vec_state.ld < n");
        hipLaunchKernelGGL(dwg_count_kernel<true>, grid, dim3(kBlock), 0, st, src, n, G, visit_order, cell_count, *vec_state,
                           vec_obs12, vec_out);
    } else {
        hipLaunchKernelGGL(dwg_count_kernel<false>, grid, dim3(kBlock), 0, st, src, n, G, visit_order, cell_count, GpdState{},
                           nullptr, nullptr);
    }
    int32_t* const cursors = cell_count + keys + 1;       // second half of cell_count: the scatter's per-key cursors
    const DwBinOut B{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, n, 0.0f, 0};
    if (keys <= kDwScanMax) {
        hipLaunchKernelGGL(dwg_scatter_kernel<true>, grid, dim3(kBlock), 0, st, src, n, G, visit_order, cell_count, cursors,
                           cell_start, order, reinterpret_cast<float4*>(sorted_xyzc), dw_out, B);
    } else {
        hipLaunchKernelGGL(dwg_scan_kernel, dim3(1), dim3(1024), 0, st, cell_count, cell_start, keys);
        hipLaunchKernelGGL(dwg_scatter_kernel<false>, grid, dim3(kBlock), 0, st, src, n, G, visit_order, cell_count, cursors,
                           cell_start, order, reinterpret_cast<float4*>(sorted_xyzc), dw_out, B);
    }
    const DwWorld Wd{nullptr, nullptr, nullptr, 0, n, 0, 0, 0, cell, nullptr, 0.0f, n};
    hipLaunchKernelGGL(dwg_force_kernel<0>, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(kBlock), 0, st, static_cast<uint32_t*>(nullptr),
                       static_cast<unsigned short*>(nullptr), static_cast<int*>(nullptr), Wd.pos4, order,
                       reinterpret_cast<const float4*>(sorted_xyzc), 0, Wd.n_slots, *params, G, Wd, DwLists{}, cell_start, dw_out, cell_count);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_downwash_global launch");
    return 0;
}

static int swarm_args(const char* who, const GpdSwarm* w, bool sorted_buffers) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string(who) + ": " + msg).c_str()); };
    if (!w) return bad(GPD_EINVAL, "NULL swarm");
    if (w->world_size < 1 || w->world_size > kBlock || w->rank < 0 || w->rank >= w->world_size) return bad(GPD_EINVAL, "need 0 <= rank < world_size <= 256");
    if (w->meta_rows < 1 || w->own_count < 0 || w->own_count > w->slab - w->meta_rows)
        return bad(GPD_EINVAL, "need 0 <= own_count <= slab - meta_rows (the last meta_rows rows of a slab are its meta rows)");
    if (static_cast<int64_t>(w->meta_rows) * kBlock < w->own_count) return bad(GPD_EINVAL, "need meta_rows >= ceil(own_count / 256): one per workgroup of gpd_swarm_step");
    if ((w->slot_of == nullptr) != (w->pos_sorted == nullptr)) return bad(GPD_EINVAL, "slot_of and pos_sorted come together");
    if (w->slot_of && w->world_size != 1) return bad(GPD_EINVAL, "positions by sorted slot (slot_of / pos_sorted) need the whole world on this rank");
    if (static_cast<int64_t>(w->slab) * w->world_size != w->n_rows) return bad(GPD_EINVAL, "n_rows must be world_size * slab");
    if (w->n_rows > (1 << 26)) return bad(GPD_ERANGE, "more than 2^26 rows");
    if (!w->pos4 || !w->bin_pos || !w->drift) return bad(GPD_EINVAL, "NULL pos4 / bin_pos / drift");
    if (w->total_drones < 1 || w->total_drones > w->n_rows) return bad(GPD_EINVAL, "need 1 <= total_drones <= n_rows");
    if (sorted_buffers) {
        if (!w->cell_count || !w->cell_start || !w->order || !w->slot_key) return bad(GPD_EINVAL, "NULL cell_count / cell_start / order / slot_key");
        if (w->visit && (w->visit == w->order || w->visit == w->visit_out)) return bad(GPD_EINVAL, "visit must alias neither order nor visit_out (ping-pong visit / visit_out)");
        if (!(w->cell >= 10.0f)) return bad(GPD_EINVAL, "cell must be >= 10 m (the model's lateral cut-off)");
        if (w->nz < 1 || w->nz > kBlock || (w->nz > 1 && !(w->zbin > 0.0f))) return bad(GPD_EINVAL, "need 1 <= nz <= 256 and zbin > 0");
        if (w->nx < 3 || w->ny < 3 || static_cast<int64_t>(w->nx) * w->ny * w->nz > 65536)
            return bad(GPD_ERANGE, "need nx, ny >= 3 (periodic search) and nx*ny*nz <= 65536");
    }
    return 0;
}

int gpd_swarm_step(const GpdParams* params, const GpdState* state, const GpdStepCfg* cfg, const GpdSwarm* swarm,
                   const float* action, float* obs12, float* vec_out, void* stream) {
    auto bad = [&](int code, const char* msg) { return fail(code, (std::string("gpd_swarm_step: ") + msg).c_str()); };
    if (!params || !state || !cfg || !action || !obs12) return bad(GPD_EINVAL, "NULL params/state/cfg/action/obs12");
    if (int rc = swarm_args("gpd_swarm_step", swarm, false)) return rc;
    if (!state->kin || !state->step_counter) return bad(GPD_EINVAL, "NULL state.kin/step_counter");
    if (const char* why = state_layout_problem(state)) return bad(GPD_EINVAL, why);
    if (cfg->drones_per_env != 1 || cfg->num_envs != swarm->own_count || cfg->num_envs <= 0) return bad(GPD_EINVAL, "need drones_per_env == 1 and num_envs == swarm.own_count > 0");
    if (cfg->substeps != 1 || cfg->task != GPD_TASK_NONE || cfg->auto_reset) return bad(GPD_ENOTSUP, "one physics sub-step per call, no task, no auto-reset");
    if (cfg->act_type != GPD_ACT_RPM && cfg->act_type != GPD_ACT_RAW_RPM && cfg->act_type != GPD_ACT_DIRECT_RPM)
        return bad(GPD_ENOTSUP, "act_type RPM, RAW_RPM or DIRECT_RPM (waypoints: gpd_pid first)");
    if (cfg->physics_flags & ~31u) return bad(GPD_EINVAL, "unknown physics flag");
    if (state->ld < cfg->num_envs) return bad(GPD_EINVAL, "state.ld < num_envs");
    if ((cfg->physics_flags & GPD_PHYS_DRAG) && !state->last_rpm) return bad(GPD_EINVAL, "GPD_PHYS_DRAG needs state.last_rpm");
    const size_t lo = static_cast<size_t>(swarm->rank) * swarm->slab;
    const SwarmOut O{reinterpret_cast<float4*>(swarm->pos4) + lo, reinterpret_cast<const float4*>(swarm->bin_pos) + lo,
                     swarm->pos4 + (lo + swarm->slab - swarm->meta_rows) * 4, swarm->slot_of ? swarm->slot_of + lo : nullptr,
                     reinterpret_cast<float4*>(swarm->pos_sorted), swarm->drift, vec_out};
    const dim3 grid(static_cast<unsigned>((cfg->num_envs + kBlock - 1) / kBlock));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GPD_SWARM_HOT state->kin, action, state->last_rpm, state->dw_force, O.bin_pos_own, O.slot_of_own, static_cast<uint32_t>(state->ld), cfg->num_envs
    switch (cfg->act_type) {
        case GPD_ACT_RAW_RPM: hipLaunchKernelGGL(gpd_swarm_step_kernel<GPD_ACT_RAW_RPM>, grid, dim3(kBlock), 0, st, GPD_SWARM_HOT, *params, *state, *cfg, obs12, O); break;
        case GPD_ACT_DIRECT_RPM: hipLaunchKernelGGL(gpd_swarm_step_kernel<GPD_ACT_DIRECT_RPM>, grid, dim3(kBlock), 0, st, GPD_SWARM_HOT, *params, *state, *cfg, obs12, O); break;
        default: hipLaunchKernelGGL(gpd_swarm_step_kernel<GPD_ACT_RPM>, grid, dim3(kBlock), 0, st, GPD_SWARM_HOT, *params, *state, *cfg, obs12, O); break;
    }
    // (too many meta rows for every force workgroup to read: leave the rank's maximum in its first meta row)
    if (static_cast<int64_t>(swarm->world_size) * swarm->meta_rows > 1024)
        hipLaunchKernelGGL(dwg_reduce_meta_kernel, dim3(1), dim3(kBlock), 0, st, O.meta_own, static_cast<int>(grid.x));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_swarm_step launch");
    return 0;
}

int gpd_swarm_pack(const GpdState* state, const GpdSwarm* swarm, const float* obs12, float* vec_out, void* stream) {
    if (!state || !state->kin) return fail(GPD_EINVAL, "gpd_swarm_pack: NULL state / state.kin");
    if (const char* why = state_layout_problem(state)) return fail(GPD_EINVAL, (std::string("gpd_swarm_pack: ") + why).c_str());
    if (int rc = swarm_args("gpd_swarm_pack", swarm, false)) return rc;
    if (state->ld < swarm->own_count) return fail(GPD_EINVAL, "gpd_swarm_pack: state.ld < own_count");
    if (vec_out && !obs12) return fail(GPD_EINVAL, "gpd_swarm_pack: vec_out needs obs12");
    const size_t lo = static_cast<size_t>(swarm->rank) * swarm->slab;
    hipLaunchKernelGGL(gpd_swarm_pack_kernel, dim3(static_cast<unsigned>((swarm->slab + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), *state, swarm->own_count, swarm->slab, reinterpret_cast<float4*>(swarm->pos4) + lo,
                       obs12, vec_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_swarm_pack launch");
    return 0;
}

int gpd_swarm_bin(const GpdSwarm* w, void* stream) {
    if (int rc = swarm_args("gpd_swarm_bin", w, true)) return rc;
    if (!w->dw_force) return fail(GPD_EINVAL, "gpd_swarm_bin: NULL dw_force (a drone without a finite position gets force 0 here)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n = w->n_rows, keys = w->nx * w->ny * w->nz;
    const DwGrid G{1.0f / w->cell, w->x0, w->y0, w->nx, w->ny, w->z0, w->nz > 1 ? 1.0f / w->zbin : 0.0f, w->nz};
    const dim3 grid(static_cast<unsigned>((n + kBlock - 1) / kBlock));
    const DwPos src{nullptr, 0, reinterpret_cast<const float4*>(w->pos4)};
    hipLaunchKernelGGL(dwg_count_kernel<false>, grid, dim3(kBlock), 0, st, src, n, G, w->visit, w->cell_count, GpdState{}, nullptr, nullptr);
    int32_t* const cursors = w->cell_count + keys + 1;
    const DwBinOut B{w->slot_key, w->slot_of, w->visit_out, w->list_ok, reinterpret_cast<float4*>(w->bin_pos), w->pos4, w->drift, w->slab, w->world_size, w->meta_rows, w->rank * w->slab, w->own_count,
                     w->pair_list ? w->list_delta : 0.0f, w->list_adapt != 0};
    float4* const srt = reinterpret_cast<float4*>(w->pos_sorted);
    if (keys <= kDwScanMax) {
        hipLaunchKernelGGL(dwg_scatter_kernel<true>, grid, dim3(kBlock), 0, st, src, n, G, w->visit, w->cell_count, cursors,
                           w->cell_start, w->order, srt, w->dw_force, B);
    } else {
        hipLaunchKernelGGL(dwg_scan_kernel, dim3(1), dim3(1024), 0, st, w->cell_count, w->cell_start, keys);
        hipLaunchKernelGGL(dwg_scatter_kernel<false>, grid, dim3(kBlock), 0, st, src, n, G, w->visit, w->cell_count, cursors,
                           w->cell_start, w->order, srt, w->dw_force, B);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_swarm_bin launch");
    return 0;
}

int gpd_swarm_forces(const GpdParams* params, const GpdSwarm* w, int32_t build_lists, void* stream) {
    if (!params) return fail(GPD_EINVAL, "gpd_swarm_forces: NULL params");
    if (int rc = swarm_args("gpd_swarm_forces", w, true)) return rc;
    if (!w->dw_force) return fail(GPD_EINVAL, "gpd_swarm_forces: NULL dw_force");
    const bool lists = w->pair_list != nullptr;
    if (lists && (!w->pair_nb || !w->list_ok || w->list_cap < 4 || w->list_cap > 65535 || !(w->list_delta >= 0.0f)))
        return fail(GPD_EINVAL, "gpd_swarm_forces: pair_list needs pair_nb, list_ok, 4 <= list_cap <= 65535 and list_delta >= 0");
    // (an entry is lane << 26 | index and 0xffffffff marks an empty lane: index 2^26 - 1 of lane 63 must not exist)
    if (lists && w->n_rows >= (1 << 26)) return fail(GPD_ERANGE, "gpd_swarm_forces: wake lists address fewer than 2^26 rows");
    const DwGrid G{1.0f / w->cell, w->x0, w->y0, w->nx, w->ny, w->z0, w->nz > 1 ? 1.0f / w->zbin : 0.0f, w->nz};
    const float4* const p4 = reinterpret_cast<const float4*>(w->pos4);
    const DwWorld Wd{w->pos_sorted ? nullptr : p4, w->slot_key, p4, w->rank * w->slab, w->own_count, w->slab, w->world_size, w->meta_rows, w->cell,
                     w->drift, 1.0f / static_cast<float>(w->total_drones), w->n_rows};
    const DwLists Ls{w->pair_list, w->pair_nb, w->list_ok, w->list_cap, w->list_delta};
    const dim3 grid(static_cast<unsigned>((w->n_rows + 63) / 64) + 1u);        // (+ the workgroup that computes the drift)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float4* const srt = reinterpret_cast<const float4*>(w->pos_sorted);
#define GPD_FORCE_ARGS Ls.list, Ls.nb, Ls.ok, Wd.pos4, w->order, srt, Ls.cap, Wd.n_slots, *params, G, Wd, Ls, w->cell_start, w->dw_force, w->cell_count
    if (!lists) hipLaunchKernelGGL(dwg_force_kernel<0>, grid, dim3(kBlock), 0, st, GPD_FORCE_ARGS);
    else if (build_lists) hipLaunchKernelGGL(dwg_force_kernel<1>, grid, dim3(kBlock), 0, st, GPD_FORCE_ARGS);
    else hipLaunchKernelGGL(dwg_force_kernel<2>, grid, dim3(kBlock), 0, st, GPD_FORCE_ARGS);
#undef GPD_FORCE_ARGS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gpd_swarm_forces launch");
    return 0;
}

}  // extern "C"

