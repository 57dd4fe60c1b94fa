// bin.hip -- tile binning by COUNTING: the (tile, splat) pairs are written once, straight into their tile's region, and
// every tile's list is then sorted by depth on its own (gfx950).
//
// Replaces the same reference code as isect.hip + radix_sort.hip (gsplat/cuda/csrc/isect_tiles.cu:16-104 count / emit,
// 199 cumsum, 245-299 the 64-bit radix sort, 308-354 offset encode) with the same outputs, bit for bit:
//   isect_ids  [n_isects]  camera << (32 + tile_bits) | tile << 32 | float bits of the depth, ascending
//   flatten_ids[n_isects]  the element behind each id
//   offsets    [C, th, tw] first list position of every tile (what isect_offset_encode derives from the sorted ids)
// The reference sorts ALL pairs globally on 64-bit keys; the first design here sorted the splats by depth first and then
// moved the 4 M pairs through two stable radix passes (24 launches, ~220 us at bench config 2, almost all of them
// launch-latency bound).  But the order INSIDE a tile is fully determined by the pairs themselves -- depth, ties by
// element index, exactly what a stable sort over the emission order produces -- so the tile lists can be filled in any
// order and sorted independently:
//   1. count    workgroup (camera, chunk of 4096 elements) enumerates the pairs of its elements and counts them per tile in
//               LDS; the rows of counters form a matrix [C][chunks][n_tiles] (and tiles_per_gauss, and the per-workgroup
//               totals whose sum is n_isects);
//   2. offsets  per tile an exclusive prefix over the chunk rows (in place) and the tile's total; one exclusive scan over
//               the C x n_tiles totals = the tile offsets;
//   3. scatter  the same enumeration again; the LDS counters now start at offsets + prefix and every pair takes the next
//               slot of its tile with a returning LDS atomic: key = depth bits << 32 | element;
//   4. sort     one workgroup per tile sorts its keys (unique: no stability needed) with a bitonic network in LDS and
//               writes isect_ids / flatten_ids.  Lists that do not fit the LDS buffer are sorted in place in global memory
//               by the same network (slow, but correct for any length).
// Integer outputs from fp32 inputs through IEEE division / floor / ceil: bit-exact against the oracle; compiled with
// -ffp-contract=off.
#include "gs_common.h"
#include "isect_common.h"

namespace {


// ---------------------------------------------------------------------------------------------------------------------
// 1 + 3. histogram / scatter.  Workgroup (camera c, chunk w) owns CHUNK consecutive elements of camera c.  Its waves take
// them 16 at a time: the 16 lanes' tile boxes and pair counts, a wave prefix sum, then the pairs of those 16 elements are
// enumerated 64 at a time (binary search over the 16 starts).  HIST: every pair bumps its tile's counter in LDS, and the row
// of counters goes to matrix[c][w][:].  SCATTER: the LDS counters start at offsets[c][tile] + (pairs of chunks < w in that
// tile) and every pair takes the next slot with a returning LDS atomic -- there is NO global atomic anywhere (the first
// version took every slot from a global cursor: 555 us for 4 M pairs; +1 / -1 corners of a global difference array for the
// counts: 264 us for 1.2 M atomics -- contended global atomics run at a few per microsecond and address here).
constexpr uint32_t SPW = 16;
constexpr uint32_t BIN_THREADS = 1024, BIN_WAVES = BIN_THREADS / GS_WAVE;
constexpr uint32_t BIN_CHUNK = 4096; // elements per workgroup

struct PairRec {
    uint64_t key;  // depth bits << 32 | element
    int32_t tile0; // y0 * tw + x0 (inside the camera)
    int32_t w;
};

template <bool SCATTER>
__global__ void __launch_bounds__(BIN_THREADS) bin_pairs_kernel(
    uint32_t N, uint32_t chunks_per_cam, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
    const float *__restrict__ depths, float tile_size, int32_t tw, int32_t th, int32_t *__restrict__ tiles_per_gauss,
    int32_t *__restrict__ block_sums, int32_t *__restrict__ matrix /* [C][chunks][n_tiles] */, const int32_t *__restrict__ offsets,
    uint64_t *__restrict__ keys) {
    extern __shared__ int32_t s_bin[]; // [n_tiles]
    __shared__ PairRec s_rec[BIN_WAVES * SPW];
    __shared__ int32_t s_start[BIN_WAVES * (SPW + 1)];
    __shared__ int32_t s_tot[BIN_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid % GS_WAVE, wave = tid / GS_WAVE;
    const uint32_t cam = blockIdx.x / chunks_per_cam, chunk = blockIdx.x % chunks_per_cam;
    const int32_t n_tiles = tw * th;
    int32_t *row = matrix + ((size_t)cam * chunks_per_cam + chunk) * n_tiles;
    for (int32_t t = (int32_t)tid; t < n_tiles; t += BIN_THREADS)
        s_bin[t] = SCATTER ? offsets[(size_t)cam * n_tiles + t] + row[t] : 0;
    __syncthreads();
    const uint32_t e0 = chunk * BIN_CHUNK, e1 = min(e0 + BIN_CHUNK, N); // element range inside the camera
    PairRec *wrec = s_rec + wave * SPW;
    int32_t *wstart = s_start + wave * (SPW + 1);
    int32_t wave_pairs = 0;
    for (uint32_t first = e0 + wave * SPW; first < e1; first += BIN_WAVES * SPW) { // (wave-uniform)
        int32_t cnt = 0;
        PairRec rec = {0ull, 0, 1};
        const uint32_t n = first + lane;
        if (lane < SPW && n < e1) {
            const size_t i = (size_t)cam * N + n;
            const int32_t r = radii[i];
            if (r > 0) {
                const float2 m = reinterpret_cast<const float2 *>(means2d)[i];
                const TileBox b = tile_box(m.x, m.y, r, tile_size, tw, th);
                cnt = (b.y1 - b.y0) * (b.x1 - b.x0);
                // raw IEEE bits of the (positive) depth, as the reference's (int64_t)*(int32_t*)&depth  (isect_tiles.cu:91)
                if (SCATTER) rec.key = ((uint64_t)(uint32_t)__float_as_int(depths[i]) << 32) | (uint64_t)i;
                rec.tile0 = b.y0 * tw + b.x0;
                rec.w = max(b.x1 - b.x0, 1);
            }
            if (!SCATTER) tiles_per_gauss[i] = cnt;
        }
        int32_t inc = cnt; // inclusive prefix over the SPW lanes
#pragma unroll
        for (int off = 1; off < (int)SPW; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += o;
        }
        const int32_t total = __shfl(inc, SPW - 1, 64);
        wave_pairs += total;
        if (total == 0) continue; // wave-uniform
        if (lane < SPW) {
            wrec[lane] = rec;
            wstart[lane] = inc - cnt;
        }
        if (lane == 0) wstart[SPW] = total;
        __builtin_amdgcn_wave_barrier();
        for (int32_t t = (int32_t)lane; t < total; t += GS_WAVE) {
            int32_t sidx = 0; // largest s with start[s] <= t (zero-count lanes share their successor's start and are skipped)
#pragma unroll
            for (int step = SPW / 2; step > 0; step >>= 1)
                if (wstart[sidx + step] <= t) sidx += step;
            const PairRec o = wrec[sidx];
            const int32_t k = t - wstart[sidx];
            const int32_t dy = k / o.w, dx = k - dy * o.w;
            const int32_t tile = o.tile0 + dy * tw + dx;
            if (SCATTER) keys[atomicAdd(&s_bin[tile], 1)] = o.key;
            else atomicAdd(&s_bin[tile], 1);
        }
        __builtin_amdgcn_wave_barrier(); // the next group overwrites the wave's records
    }
    if (SCATTER) return;
    if (lane == 0) s_tot[wave] = wave_pairs;
    __syncthreads();
    for (int32_t t = (int32_t)tid; t < n_tiles; t += BIN_THREADS) row[t] = s_bin[t];
    if (tid == 0 && block_sums != nullptr) { // per-workgroup pair counts: their sum is n_isects (see isect_count_keys_kernel)
        int32_t s = 0;
        for (uint32_t w = 0; w < BIN_WAVES; ++w) s += s_tot[w];
        block_sums[blockIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. offsets.  (a) per (camera, tile): exclusive prefix over the chunk rows of the matrix, in place, and the tile's total;
// (b) one workgroup: exclusive scan of the C x n_tiles totals in (camera, tile) order = the tile offsets.
__global__ void __launch_bounds__(GS_BLOCK) bin_colscan_kernel(uint32_t chunks_per_cam, uint32_t n_tiles, int32_t *__restrict__ matrix,
                                                               int32_t *__restrict__ totals) {
    const uint32_t cam = blockIdx.y, t = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (t >= n_tiles) return;
    int32_t *col = matrix + (size_t)cam * chunks_per_cam * n_tiles + t;
    int32_t run = 0;
    for (uint32_t w = 0; w < chunks_per_cam; ++w) {
        const int32_t v = col[(size_t)w * n_tiles];
        col[(size_t)w * n_tiles] = run;
        run += v;
    }
    totals[(size_t)cam * n_tiles + t] = run;
}

constexpr int OFFS_BLOCK = 1024;
__global__ void __launch_bounds__(OFFS_BLOCK) bin_offsets_kernel(uint32_t n_tiles_all, const int32_t *__restrict__ totals,
                                                                 int32_t *__restrict__ offsets) {
    __shared__ int32_t s_wave[OFFS_BLOCK / GS_WAVE];
    __shared__ int32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles_all; base += OFFS_BLOCK) {
        const uint32_t t = base + tid;
        const int32_t v = t < n_tiles_all ? totals[t] : 0;
        int32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += o;
        }
        if (lane == 63u) s_wave[wave] = inc;
        __syncthreads();
        int32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < OFFS_BLOCK / GS_WAVE; ++w) {
            if ((uint32_t)w < wave) wbase += s_wave[w];
            total += s_wave[w];
        }
        const int32_t carry = s_carry;
        if (t < n_tiles_all) offsets[t] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == 0) s_carry = carry + total;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. sort.  An all-ascending bitonic network: for every block size k = 2, 4, ... the first step pairs i with its mirror image
// inside the block (i ^ (k - 1)), the following steps pair i with i ^ j, j = k/4 ... 1; every exchange puts the smaller key
// at the lower index, so an array of any length n sorts correctly by simply skipping the pairs whose upper index is >= n
// (the missing elements act as +infinity, which an ascending network never moves down).  Keys are unique
// (depth bits << 32 | element): no stability needed.
template <typename Arr>
GS_DEV void bitonic_step(Arr a, uint32_t n, uint32_t half_pairs, uint32_t j, uint32_t flip_mask, uint32_t tid, uint32_t n_threads) {
    for (uint32_t p = tid; p < half_pairs; p += n_threads) {
        const uint32_t lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)); // the p-th index with bit j clear
        const uint32_t hi = flip_mask ? (lo ^ flip_mask) : (lo | j);
        if (hi < n) {
            const uint64_t x = a[lo], y = a[hi];
            if (x > y) {
                a[lo] = y;
                a[hi] = x;
            }
        }
    }
}

template <int CAP, int THREADS, bool BIG>
__global__ void __launch_bounds__(THREADS) bin_sort_kernel(
    uint32_t n_tiles_all, uint32_t n_tiles, uint32_t n_isects, uint32_t tile_n_bits, int32_t min_len, const int32_t *__restrict__ offsets,
    uint64_t *__restrict__ keys, int64_t *__restrict__ isect_ids, int32_t *__restrict__ flatten_ids) {
    __shared__ uint64_t s_key[CAP];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const int32_t start = offsets[tile];
    const int32_t end = tile + 1 < n_tiles_all ? offsets[tile + 1] : (int32_t)n_isects;
    const uint32_t n = (uint32_t)(end - start);
    if ((int32_t)n <= min_len || (!BIG && n > (uint32_t)CAP)) return; // (block-uniform) not this launch's size class
    // id of this tile's pairs: camera << (32 + tile_bits) | tile << 32
    const uint64_t hi_bits = (((uint64_t)(tile / n_tiles) << tile_n_bits) | (uint64_t)(tile % n_tiles)) << 32;
    uint64_t *g = keys + start;
    uint32_t n_pad = 1;
    while (n_pad < n) n_pad <<= 1;
    const bool in_lds = n <= (uint32_t)CAP;
    if (in_lds) {
        for (uint32_t p = tid; p < n; p += THREADS) s_key[p] = g[p];
        __syncthreads();
        for (uint32_t k = 2; k <= n_pad; k <<= 1) {
            bitonic_step(s_key, n, n_pad / 2, k >> 1, k - 1, tid, THREADS);
            __syncthreads();
            for (uint32_t j = k >> 2; j > 0; j >>= 1) {
                bitonic_step(s_key, n, n_pad / 2, j, 0u, tid, THREADS);
                __syncthreads();
            }
        }
    } else {
        // longer than the LDS buffer: the same network in place in global memory (volatile: the workgroup re-reads what its
        // other waves wrote; slow, but correct for any length)
        volatile uint64_t *v = g;
        for (uint32_t k = 2; k <= n_pad; k <<= 1) {
            bitonic_step(v, n, n_pad / 2, k >> 1, k - 1, tid, THREADS);
            __threadfence_block();
            __syncthreads();
            for (uint32_t j = k >> 2; j > 0; j >>= 1) {
                bitonic_step(v, n, n_pad / 2, j, 0u, tid, THREADS);
                __threadfence_block();
                __syncthreads();
            }
        }
    }
    for (uint32_t p = tid; p < n; p += THREADS) {
        const uint64_t key = in_lds ? s_key[p] : ((volatile uint64_t *)g)[p];
        isect_ids[start + p] = (int64_t)(hi_bits | (key >> 32));
        flatten_ids[start + p] = (int32_t)(uint32_t)key;
    }
}

} // namespace

namespace {
constexpr size_t BIN_LDS_MAX = 150 * 1024; // dynamic LDS of bin_pairs_kernel: n_tiles counters
template <bool SCATTER>
int32_t set_bin_lds() {
    static bool done = false; // more than 64 KB of dynamic LDS has to be asked for once per kernel
    if (!done) {
        if (hipFuncSetAttribute((const void *)bin_pairs_kernel<SCATTER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BIN_LDS_MAX) !=
            hipSuccess) {
            gs_set_error("gs_bin: cannot reserve %zu bytes of LDS", BIN_LDS_MAX);
            return 2;
        }
        done = true;
    }
    return 0;
}
} // namespace

extern "C" uint32_t gs_bin_chunks(uint32_t N) { return gs_div_up(N, BIN_CHUNK); }
extern "C" uint32_t gs_bin_max_tiles(void) { return (uint32_t)(BIN_LDS_MAX / sizeof(int32_t)); }

extern "C" int32_t gs_bin_count(
    uint32_t C, uint32_t N, const float *means2d, const int32_t *radii, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t *tiles_per_gauss, int32_t *block_sums, int32_t *matrix, int32_t *totals, int32_t *offsets, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means2d && radii && tiles_per_gauss && matrix && totals && offsets, "null pointer");
    GS_CHECK_ARG(tile_size > 0, "tile_size must be > 0");
    const uint32_t n_tiles = tile_width * tile_height, chunks = gs_bin_chunks(N);
    GS_CHECK_ARG(n_tiles >= 1 && n_tiles <= gs_bin_max_tiles(), "tile grid too large for the counting path (gs_bin_max_tiles)");
    if (int32_t rc = set_bin_lds<false>()) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bin_pairs_kernel<false>, dim3(C * chunks), dim3(BIN_THREADS), (size_t)n_tiles * sizeof(int32_t), st, N, chunks, means2d,
                       radii, (const float *)nullptr, (float)tile_size, (int32_t)tile_width, (int32_t)tile_height, tiles_per_gauss, block_sums,
                       matrix, (const int32_t *)nullptr, (uint64_t *)nullptr);
    hipLaunchKernelGGL(bin_colscan_kernel, dim3(gs_div_up(n_tiles, GS_BLOCK), C), dim3(GS_BLOCK), 0, st, chunks, n_tiles, matrix, totals);
    hipLaunchKernelGGL(bin_offsets_kernel, dim3(1), dim3(OFFS_BLOCK), 0, st, C * n_tiles, totals, offsets);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_bin_scatter_sort(
    uint32_t C, uint32_t N, uint32_t n_isects, const float *means2d, const int32_t *radii, const float *depths, uint32_t tile_size,
    uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits, int32_t *matrix, const int32_t *offsets, uint64_t *keys,
    int64_t *isect_ids, int32_t *flatten_ids, gs_stream_t stream) {
    if (n_isects == 0 || C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means2d && radii && depths && matrix && offsets && keys && isect_ids && flatten_ids, "null pointer");
    GS_CHECK_ARG(tile_n_bits < 32, "tile_n_bits must be < 32");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t n_tiles = tile_width * tile_height, n_tiles_all = C * n_tiles, chunks = gs_bin_chunks(N);
    GS_CHECK_ARG(n_tiles >= 1 && n_tiles <= gs_bin_max_tiles(), "tile grid too large for the counting path (gs_bin_max_tiles)");
    if (int32_t rc = set_bin_lds<true>()) return rc;
    hipLaunchKernelGGL(bin_pairs_kernel<true>, dim3(C * chunks), dim3(BIN_THREADS), (size_t)n_tiles * sizeof(int32_t), st, N, chunks, means2d,
                       radii, depths, (float)tile_size, (int32_t)tile_width, (int32_t)tile_height, (int32_t *)nullptr, (int32_t *)nullptr, matrix,
                       offsets, keys);
    // two size classes: lists of up to 1024 keys with a small LDS buffer (many workgroups per CU), the rest with 64 KB
    hipLaunchKernelGGL((bin_sort_kernel<1024, 256, false>), dim3(n_tiles_all), dim3(256), 0, st, n_tiles_all, n_tiles, n_isects, tile_n_bits,
                       0, offsets, keys, isect_ids, flatten_ids);
    hipLaunchKernelGGL((bin_sort_kernel<8192, 1024, true>), dim3(n_tiles_all), dim3(1024), 0, st, n_tiles_all, n_tiles, n_isects, tile_n_bits,
                       1024, offsets, keys, isect_ids, flatten_ids);
    GS_CHECK_LAUNCH();
    return 0;
}
