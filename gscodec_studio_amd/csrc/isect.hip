// isect.hip -- R3/R4: gaussian/tile intersection, prefix sums, offset encode (gfx950).
//
// Replaces gsplat/cuda/csrc/isect_tiles.cu:16-104 (count + emit passes), the
// torch::cumsum between them (isect_tiles.cu:199) and isect_offset_encode
// (isect_tiles.cu:308-354).  The 64-bit radix sort lives in radix_sort.hip.
//
// All outputs of this file are integers derived from fp32 inputs with IEEE
// (correctly rounded) division / floor / ceil, so they are bit-exact against the
// oracle.  The file is compiled with -ffp-contract=off.
#include "gs_common.h"

namespace {

__global__ void __launch_bounds__(GS_BLOCK) isect_count_kernel(
    uint32_t n_elems, const float *__restrict__ means2d, uint32_t s_m2, const int32_t *__restrict__ radii,
    float tile_size, int32_t tw, int32_t th, int32_t *__restrict__ tiles_per_gauss) {
    uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i >= n_elems) return;
    int32_t r = radii[i];
    int32_t cnt = 0;
    if (r > 0) {
        float2 m = *reinterpret_cast<const float2 *>(means2d + (size_t)i * s_m2);
        TileBox b = tile_box(m.x, m.y, r, tile_size, tw, th);
        cnt = (b.y1 - b.y0) * (b.x1 - b.x0);
    }
    tiles_per_gauss[i] = cnt;
}

// Depth keys for the splat-level pre-sort: (float_bits(depth) << 32) | element for visible
// elements, a maximal positive key for culled ones (they sort last and emit nothing).
__global__ void __launch_bounds__(GS_BLOCK) isect_depth_keys_kernel(
    uint32_t n_elems, const int32_t *__restrict__ radii, const float *__restrict__ depths,
    int64_t *__restrict__ keys, int32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i >= n_elems) return;
    uint32_t d = radii[i] > 0 ? (uint32_t)__float_as_int(depths[i]) & 0x7fffffffu : 0x7fffffffu;
    keys[i] = (int64_t)(((uint64_t)d << 32) | (uint64_t)i);
    vals[i] = (int32_t)i;
}

// isect_count_kernel + isect_depth_keys_kernel in one launch (the sorted path needs both)
// block_sums (optional): the block's number of intersections.  Their sum is n_isects -- available right after THIS kernel,
// ~100 us of GPU work (depth pre-sort, prefix sum, SH colours) before the pipeline needs it on the host: the caller copies
// the partial sums to pinned memory here and adds them up on the host, so the one host read-back of the pipeline
// (isect_tiles.cu:200 in the reference) no longer leaves the GPU idle while the host wakes up and queues the rest.
// hist (optional): the digit histogram of the depth pre-sort's FIRST pass for this block's 1024 keys (digit = low 8 bits of
// the depth bits, culled keys not counted), in the sort's own [256][n_blocks] layout -- the sort then skips that launch.
constexpr int COUNT_ITEMS = 4; // elements per thread: a block covers the 1024 keys of one sort block
__global__ void __launch_bounds__(GS_BLOCK) isect_count_keys_kernel(
    uint32_t n_elems, const float *__restrict__ means2d, uint32_t s_m2, const int32_t *__restrict__ radii,
    const float *__restrict__ depths, float tile_size, int32_t tw, int32_t th,
    int32_t *__restrict__ tiles_per_gauss, int64_t *__restrict__ keys, int32_t *__restrict__ vals, int32_t *__restrict__ block_sums,
    uint32_t *__restrict__ hist, uint32_t n_blocks, const uint64_t *__restrict__ split) {
    __shared__ int32_t s_sum[GS_BLOCK / GS_WAVE], s_vis[GS_BLOCK / GS_WAVE];
    __shared__ uint32_t s_hist[256];
    __shared__ uint64_t s_split[GS_PRESORT_BUCKETS];
    if (hist != nullptr) s_hist[threadIdx.x] = 0u;
    if (split != nullptr) s_split[threadIdx.x] = split[threadIdx.x]; // bucketed pre-sort: digit = bucket of the whole key
    __syncthreads();
    int32_t cnt_sum = 0, vis_sum = 0;
#pragma unroll
    for (int k = 0; k < COUNT_ITEMS; ++k) {
        const uint32_t i = (blockIdx.x * COUNT_ITEMS + k) * GS_BLOCK + threadIdx.x;
        if (i >= n_elems) continue;
        const int32_t r = radii[i];
        uint32_t d = 0x7fffffffu;
        int32_t cnt = 0;
        if (r > 0) {
            if (means2d != nullptr) { // (uniform; NULL: tiles_per_gauss holds the counts already -- gs_projection_rows_fwd made them)
                float2 m = *reinterpret_cast<const float2 *>(means2d + (size_t)i * s_m2);
                TileBox b = tile_box(m.x, m.y, r, tile_size, tw, th);
                cnt = (b.y1 - b.y0) * (b.x1 - b.x0);
            }
            d = (uint32_t)__float_as_int(depths[i]) & 0x7fffffffu;
        }
        if (means2d != nullptr) tiles_per_gauss[i] = cnt;
        const uint64_t key = ((uint64_t)d << 32) | (uint64_t)i;
        keys[i] = (int64_t)key;
        if (vals != nullptr) vals[i] = (int32_t)i; // (NULL: the bucketed pre-sort reads the element off the key's low half)
        cnt_sum += cnt;
        vis_sum += r > 0 ? 1 : 0;
        if (hist != nullptr && d != 0x7fffffffu) atomicAdd(&s_hist[split != nullptr ? gs_bucket_of(s_split, key) : (d & 0xffu)], 1u);
    }
    if (block_sums != nullptr) { // (block-uniform)
        int32_t v = cnt_sum, u = vis_sum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v += __shfl_xor(v, off, 64);
            u += __shfl_xor(u, off, 64);
        }
        if ((threadIdx.x & 63u) == 0u) {
            s_sum[threadIdx.x >> 6] = v;
            s_vis[threadIdx.x >> 6] = u;
        }
        __syncthreads();
        if (threadIdx.x == 0) // (intersections, visible elements) as ONE 8-byte store: they reach the host together
            reinterpret_cast<int2 *>(block_sums)[blockIdx.x] = make_int2(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3], s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3]);
    }
    if (hist != nullptr) {
        __syncthreads();
        hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s_hist[threadIdx.x];
    }
}

__global__ void __launch_bounds__(GS_BLOCK) gather_i32_kernel(
    uint32_t n, const int32_t *__restrict__ src, const int32_t *__restrict__ idx, int32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// Wave-cooperative emission.  A wave owns a few consecutive positions of the emission order, whose pairs
// form ONE contiguous output range [cum[first-1], cum[last]).  The lanes first park their splat's
// record (output start, tile box, key base, id) in LDS, then walk the output range 64 slots at a time:
// slot o finds its owner by a binary search over the starts and derives its tile from its
// offset inside the owner's box.  Stores are fully coalesced (the one-thread-per-splat loop wrote 12 B
// at ~160 B strides: 51 us for 4 M pairs at config 2, 2.7x the compulsory traffic).
struct EmitRec {
    int64_t key_base; // camera bits | depth bits
    int32_t id;       // flatten id
    int32_t x0, y0, w;
};

// A wave owns EMIT_SPW consecutive positions of the emission order (16, not 64: the emission order is the depth order, so the big near splats
// sit next to each other and a 64-splat wave could own thousands of pairs while most own a few hundred -- the kernel
// lasted as long as that wave: 37 us at config 2 against 24 us with 16).
#ifndef GS_EMIT_SPW
#define GS_EMIT_SPW 16
#endif
constexpr uint32_t EMIT_SPW = GS_EMIT_SPW;
constexpr uint32_t EMIT_WAVES = GS_BLOCK / GS_WAVE;

// COMPACT: the pair is written as (32-bit key camera << tile_n_bits | tile, flatten id) = 8 bytes for the 32-bit pair sort
// (gs_sort_isect_pairs rebuilds the 64-bit id with the depth bits in its last pass); else the reference's 64-bit id.
template <bool COMPACT>
__global__ void __launch_bounds__(GS_BLOCK) isect_emit_kernel(
    uint32_t n_elems, uint32_t N, const int32_t *__restrict__ perm, const uint32_t *__restrict__ n_valid,
    const int64_t *__restrict__ camera_ids,
    const float *__restrict__ means2d, uint32_t s_m2, const int32_t *__restrict__ radii,
    const float *__restrict__ depths, const int64_t *__restrict__ cum_tiles,
    float tile_size, int32_t tw, int32_t th, uint32_t tile_n_bits,
    int64_t *__restrict__ isect_ids, uint32_t *__restrict__ keys32, int32_t *__restrict__ flatten_ids) {
    // `pos` = position in the emission order (identity, or depth-sorted when perm is given);
    // `i` = the element it refers to.  cum_tiles is indexed by position.
    __shared__ EmitRec s_rec[EMIT_WAVES * EMIT_SPW];
    __shared__ int32_t s_start[EMIT_WAVES * (EMIT_SPW + 1)]; // SPW + 1 starts per wave (last = total)
    const uint32_t lane = threadIdx.x % GS_WAVE, wave = threadIdx.x / GS_WAVE;
    const uint32_t wave_first = (blockIdx.x * EMIT_WAVES + wave) * EMIT_SPW;
    if (wave_first >= n_elems) return; // wave-uniform
    if (n_valid != nullptr && wave_first >= *n_valid) return; // behind the valid prefix (cum_tiles is not defined there)
    const uint32_t pos = wave_first + lane; // lanes >= EMIT_SPW carry no splat
    const uint32_t wave_last = min(wave_first + EMIT_SPW, n_elems) - 1;
    const int64_t out0 = (wave_first == 0) ? 0 : cum_tiles[wave_first - 1];
    const int64_t out1 = cum_tiles[wave_last];
    if (out1 == out0) return; // nothing visible in this wave (wave-uniform)
    EmitRec *wrec = s_rec + wave * EMIT_SPW;
    int32_t *wstart = s_start + wave * (EMIT_SPW + 1);
    if (lane < EMIT_SPW) {
        EmitRec rec = {0, 0, 0, 0, 1};
        int32_t start = (int32_t)(out1 - out0);
        if (pos < n_elems) {
            start = (int32_t)(((pos == 0) ? 0 : cum_tiles[pos - 1]) - out0);
            // positions behind *n_valid hold no element (perm is only defined for the kept ones)
            const bool has = n_valid == nullptr || pos < *n_valid;
            const uint32_t i = has ? (perm != nullptr ? (uint32_t)perm[pos] : pos) : 0u;
            const int32_t r = has ? radii[i] : 0;
            if (r > 0) {
                const float2 m = *reinterpret_cast<const float2 *>(means2d + (size_t)i * s_m2);
                const TileBox b = tile_box(m.x, m.y, r, tile_size, tw, th);
                const int64_t cid = camera_ids != nullptr ? camera_ids[i] : (int64_t)(i / N);
                // raw IEEE bits of the (positive) depth, sign-extended like the reference's
                // (int64_t)*(int32_t*)&depth  (isect_tiles.cu:91)
                rec.key_base = COMPACT ? (cid << tile_n_bits) : ((cid << (32 + tile_n_bits)) | (int64_t)__float_as_int(depths[i]));
                rec.id = (int32_t)i;
                rec.x0 = b.x0;
                rec.y0 = b.y0;
                rec.w = max(b.x1 - b.x0, 1);
            }
        }
        wrec[lane] = rec;
        wstart[lane] = start;
    }
    if (lane == 0) wstart[EMIT_SPW] = (int32_t)(out1 - out0);
    __builtin_amdgcn_wave_barrier();
    const int32_t total = (int32_t)(out1 - out0);
    for (int32_t t = (int32_t)lane; t < total; t += GS_WAVE) {
        // largest s with start[s] <= t (zero-count lanes share their successor's start and are skipped)
        int32_t sidx = 0;
#pragma unroll
        for (int step = EMIT_SPW / 2; step > 0; step >>= 1)
            if (wstart[sidx + step] <= t) sidx += step;
        const EmitRec o = wrec[sidx];
        const int32_t k = t - wstart[sidx];
        const int32_t dy = k / o.w, dx = k - dy * o.w;
        const int64_t tile_id = (int64_t)(o.y0 + dy) * tw + (o.x0 + dx);
        if (COMPACT) keys32[out0 + t] = (uint32_t)(o.key_base | tile_id);
        else isect_ids[out0 + t] = o.key_base | (tile_id << 32);
        flatten_ids[out0 + t] = o.id;
    }
}

// Emission for the PRE-SORTED path with the prefix scan folded in (round 3: the two launches of gs_cumsum_gather_i32 and the
// cum_tiles array are gone).  A workgroup owns EMIT_SCAN_TILE (= 128) consecutive positions of the emission order:
//   * its output offset = the sum of its predecessors' group sums (group_sums[g] = tiles of positions [128 g, 128 (g+1)),
//     left behind by the LAST pass of the depth pre-sort -- gs_presort_buckets / gs_sort_pairs_u64_i32_drop: side_sums);
//   * tile count, radius and mean of its positions are gathered through perm in ONE phase; the inclusive scan of the
//     counts and the emission records of all 128 positions live in LDS;
//   * its four waves then emit groups of EMIT_SPW positions exactly like isect_emit_kernel.
constexpr uint32_t EMIT_SCAN_SHIFT = 7; // 128 positions per workgroup: two groups of EMIT_SPW per wave (512: 58 us for 572
                                        // workgroups of eight sequential groups per wave -- too few waves in flight; the
                                        // one-group-per-wave kernel above needed 19 us + 15 us of prefix-sum launches)
constexpr uint32_t EMIT_SCAN_TILE = 1u << EMIT_SCAN_SHIFT;
static_assert(EMIT_SCAN_TILE <= GS_BLOCK && EMIT_SCAN_TILE % EMIT_SPW == 0, "one position per thread");

// MODE 0: 64-bit ids + flatten ids; 1 (compact): 32-bit (camera, tile) key + flatten id; 2 (packed): ONE 32-bit word per pair,
// key << pos_bits | emission position -- the position is the splat's depth rank, nothing else has to ride through the sort
template <int MODE>
__global__ void __launch_bounds__(GS_BLOCK) isect_emit_scan_kernel(
    uint32_t n_elems, uint32_t N, const int32_t *__restrict__ perm, const uint32_t *__restrict__ n_valid,
    const int64_t *__restrict__ camera_ids, const float *__restrict__ means2d, uint32_t s_m2, const int32_t *__restrict__ radii,
    const float *__restrict__ depths, const int32_t *__restrict__ tiles_per_gauss, const uint32_t *__restrict__ group_sums,
    const int64_t *__restrict__ group_prefix, float tile_size, int32_t tw, int32_t th, uint32_t tile_n_bits,
    int64_t *__restrict__ isect_ids, uint32_t *__restrict__ keys32, int32_t *__restrict__ flatten_ids, uint32_t pos_bits) {
    constexpr bool COMPACT = MODE != 0;
    __shared__ int32_t s_cum[EMIT_SCAN_TILE + 1]; // s_cum[j] = tiles of the block's positions [0, j)
    __shared__ int64_t s_red[GS_BLOCK / GS_WAVE];
    __shared__ int32_t s_wsum[GS_BLOCK / GS_WAVE];
    __shared__ EmitRec s_rec[EMIT_SCAN_TILE]; // the record of EVERY position of the block, gathered in one phase
    __shared__ int32_t s_start[EMIT_WAVES * (EMIT_SPW + 1)];
    const uint32_t tid = threadIdx.x, lane = tid % GS_WAVE, wave = tid / GS_WAVE;
    const uint32_t nv = n_valid != nullptr ? min(n_elems, *n_valid) : n_elems;
    const uint32_t block_first = blockIdx.x * EMIT_SCAN_TILE;
    if (block_first >= nv) return; // (block-uniform) nothing to emit here or behind
    // ---- offset of this block in the output: its predecessors' group sums
    // (every block adds up its predecessors itself -- quadratic, fine for the few thousand groups of a million splats; the
    // caller hands over the inclusive prefix instead when there are more)
    int64_t pre = 0;
    if (group_prefix != nullptr) pre = (tid == 0 && blockIdx.x > 0) ? group_prefix[blockIdx.x - 1] : 0;
    else
        for (uint32_t g = tid; g < blockIdx.x; g += GS_BLOCK) pre += (int64_t)group_sums[g];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
    if (lane == 0) s_red[wave] = pre;
    // ---- my position (threads beyond the tile only help with the offset above)
    int32_t c = 0, e = -1;
    if (tid < EMIT_SCAN_TILE) {
        const uint32_t pos = block_first + tid;
        e = pos < nv ? (perm != nullptr ? perm[pos] : (int32_t)pos) : -1;
        // the three gathers that hang off the element index go out together (they used to be two dependent phases: the tile
        // count here, radius and mean inside the per-group loop below -- once per group and wave)
        int32_t r = 0;
        float2 m = make_float2(0.f, 0.f);
        if (e >= 0) {
            c = tiles_per_gauss[e];
            r = radii[e];
            m = *reinterpret_cast<const float2 *>(means2d + (size_t)e * s_m2);
        }
        EmitRec rec = {0, 0, 0, 0, 1};
        if (r > 0) {
            const TileBox b = tile_box(m.x, m.y, r, tile_size, tw, th);
            const int64_t cid = camera_ids != nullptr ? camera_ids[e] : (int64_t)((uint32_t)e / N);
            rec.key_base = COMPACT ? (cid << tile_n_bits) : ((cid << (32 + tile_n_bits)) | (int64_t)__float_as_int(depths[e]));
            rec.id = MODE == 2 ? (int32_t)pos : e; // (packed: the emission position)
            rec.x0 = b.x0;
            rec.y0 = b.y0;
            rec.w = max(b.x1 - b.x0, 1);
        }
        s_rec[tid] = rec;
    }
    int32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == GS_WAVE - 1) s_wsum[wave] = inc;
    __syncthreads();
    int32_t wbase = 0;
#pragma unroll
    for (int w = 0; w < GS_BLOCK / GS_WAVE; ++w)
        if ((uint32_t)w < wave) wbase += s_wsum[w];
    if (tid < EMIT_SCAN_TILE) s_cum[tid] = wbase + inc - c;
    if (tid == EMIT_SCAN_TILE - 1) s_cum[EMIT_SCAN_TILE] = wbase + inc;
    const int64_t block_out0 = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();

    int32_t *wstart = s_start + wave * (EMIT_SPW + 1);
    for (uint32_t grp = wave; grp < EMIT_SCAN_TILE / EMIT_SPW; grp += EMIT_WAVES) { // groups of EMIT_SPW positions, one wave each
        const uint32_t j0 = grp * EMIT_SPW;
        if (block_first + j0 >= nv) break; // (wave-uniform)
        const int32_t g0 = s_cum[j0], g1 = s_cum[j0 + EMIT_SPW];
        if (g1 == g0) continue;
        const EmitRec *wrec = s_rec + j0;
        if (lane < EMIT_SPW) wstart[lane] = s_cum[j0 + lane] - g0;
        if (lane == 0) wstart[EMIT_SPW] = g1 - g0;
        __builtin_amdgcn_wave_barrier();
        const int32_t total = g1 - g0;
        const int64_t out0 = block_out0 + g0;
        for (int32_t t = (int32_t)lane; t < total; t += GS_WAVE) {
            int32_t sidx = 0;
#pragma unroll
            for (int step = EMIT_SPW / 2; step > 0; step >>= 1)
                if (wstart[sidx + step] <= t) sidx += step;
            const EmitRec o = wrec[sidx];
            const int32_t k = t - wstart[sidx];
            // k / w without the ~30-instruction integer division: float quotient, then one exact correction step (the float
            // result is off by at most one for any k < 2^24)
            int32_t dy = (int32_t)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)o.w));
            int32_t dx = k - dy * o.w;
            if (dx < 0) { dy -= 1; dx += o.w; }
            else if (dx >= o.w) { dy += 1; dx -= o.w; }
            const int64_t tile_id = (int64_t)(o.y0 + dy) * tw + (o.x0 + dx);
            if (MODE == 2) keys32[out0 + t] = ((uint32_t)(o.key_base | tile_id) << pos_bits) | (uint32_t)o.id;
            else if (MODE == 1) keys32[out0 + t] = (uint32_t)(o.key_base | tile_id);
            else isect_ids[out0 + t] = o.key_base | (tile_id << 32);
            if (MODE != 2) flatten_ids[out0 + t] = o.id;
        }
        __builtin_amdgcn_wave_barrier(); // the next group reuses wstart
    }
}

// isect_tiles.cu:308-354.  Four consecutive ids per thread (two 16-byte loads + the 8 bytes in front): the kernel streams the
// 8 I bytes of sorted ids once, and the tile arithmetic only runs at the ~T boundaries.
constexpr uint32_t OFFSET_ITEMS = 4;
__global__ void __launch_bounds__(GS_BLOCK) isect_offset_encode_kernel(
    uint32_t n_isects, const int64_t *__restrict__ isect_ids, uint32_t C, uint32_t n_tiles,
    uint32_t tile_n_bits, int32_t *__restrict__ offsets) {
    const uint32_t base = (blockIdx.x * GS_BLOCK + threadIdx.x) * OFFSET_ITEMS;
    if (base >= n_isects) return;
    const int64_t tmask = ((int64_t)1 << tile_n_bits) - 1;
    auto tile_index = [&](int64_t key) { return (key >> tile_n_bits) * (int64_t)n_tiles + (key & tmask); };
    int64_t k[OFFSET_ITEMS];
    if (base + OFFSET_ITEMS <= n_isects && (reinterpret_cast<uintptr_t>(isect_ids) & 15u) == 0u) { // (uniform but for the last thread)
        typedef long long v2ll __attribute__((ext_vector_type(2)));
        const v2ll a = *reinterpret_cast<const v2ll *>(isect_ids + base), b = *reinterpret_cast<const v2ll *>(isect_ids + base + 2);
        k[0] = a.x >> 32; k[1] = a.y >> 32; k[2] = b.x >> 32; k[3] = b.y >> 32;
    } else {
#pragma unroll
        for (uint32_t j = 0; j < OFFSET_ITEMS; ++j) k[j] = base + j < n_isects ? (isect_ids[base + j] >> 32) : 0;
    }
    int64_t prev = base > 0 ? (isect_ids[base - 1] >> 32) : 0;
#pragma unroll
    for (uint32_t j = 0; j < OFFSET_ITEMS; ++j) {
        const uint32_t idx = base + j;
        if (idx >= n_isects) break;
        const int64_t cur = k[j];
        if (idx == 0) {
            const int64_t id_cur = tile_index(cur);
            for (int64_t i = 0; i <= id_cur; ++i) offsets[i] = 0;
        } else if (prev != cur) {
            const int64_t id_prev = tile_index(prev), id_cur = tile_index(cur);
            for (int64_t i = id_prev + 1; i <= id_cur; ++i) offsets[i] = (int32_t)idx;
        }
        if (idx == n_isects - 1) {
            const int64_t id_cur = tile_index(cur);
            for (int64_t i = id_cur + 1; i < (int64_t)C * n_tiles; ++i) offsets[i] = (int32_t)n_isects;
        }
        prev = cur;
    }
}

// ---------------------------------------------------------------------------
// inclusive prefix sum (reduce-then-scan, 3 launches).  2048 items per block.
// ---------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = GS_BLOCK * SCAN_ITEMS;
// scan_apply leaves cum_tiles unwritten from the first scan block that lies entirely behind *n_valid, and an emit wave whose
// first position is valid reads cum_tiles up to its LAST position: an emit wave must never straddle a scan block
static_assert(SCAN_TILE % EMIT_SPW == 0, "GS_EMIT_SPW must divide the scan tile (2048)");

GS_DEV int64_t wave_inclusive_scan_i64(int64_t v) {
    uint32_t lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int64_t o = __shfl_up(v, off, 64);
        if (lane >= (uint32_t)off) v += o;
    }
    return v;
}

// block-wide exclusive scan of one int64 per thread; returns the exclusive prefix and
// the block total.
GS_DEV int64_t block_exclusive_scan_i64(int64_t v, int64_t &total, int64_t *s_wave /*[4]*/) {
    uint32_t lane = threadIdx.x % GS_WAVE, wave = threadIdx.x / GS_WAVE;
    int64_t inc = wave_inclusive_scan_i64(v);
    if (lane == GS_WAVE - 1) s_wave[wave] = inc;
    __syncthreads();
    int64_t base = 0, t = 0;
#pragma unroll
    for (int w = 0; w < GS_BLOCK / GS_WAVE; ++w) {
        if ((uint32_t)w < wave) base += s_wave[w];
        t += s_wave[w];
    }
    total = t;
    __syncthreads();
    return base + inc - v;
}

// idx != nullptr: the scanned value is in[idx[i]]; n_valid != nullptr: positions >= *n_valid count as 0
__global__ void __launch_bounds__(GS_BLOCK) scan_block_sums_kernel(
    uint64_t n, const int32_t *__restrict__ in, const int32_t *__restrict__ idx, const uint32_t *__restrict__ n_valid,
    int64_t *__restrict__ block_sums) {
    __shared__ int64_t s_wave[GS_BLOCK / GS_WAVE];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    const uint64_t nv = n_valid != nullptr ? min(n, (uint64_t)*n_valid) : n;
    if (n_valid != nullptr && base >= nv) { // behind the valid prefix: contributes nothing (the spine still sums every block)
        if (threadIdx.x == 0) block_sums[blockIdx.x] = 0;
        return;
    }
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + (uint64_t)k * GS_BLOCK + threadIdx.x;
        if (i < nv) s += idx != nullptr ? in[idx[i]] : in[i];
    }
    int64_t total;
    block_exclusive_scan_i64(s, total, s_wave);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: in-place exclusive scan of the block sums
__global__ void __launch_bounds__(GS_BLOCK) scan_spine_kernel(uint32_t n_blocks, int64_t *__restrict__ block_sums) {
    __shared__ int64_t s_wave[GS_BLOCK / GS_WAVE];
    int64_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += GS_BLOCK) {
        uint32_t i = base + threadIdx.x;
        int64_t v = i < n_blocks ? block_sums[i] : 0;
        int64_t total;
        int64_t ex = block_exclusive_scan_i64(v, total, s_wave);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += total;
    }
}

template <typename OutT>
__global__ void __launch_bounds__(GS_BLOCK) scan_apply_kernel(
    uint64_t n, const int32_t *__restrict__ in, const int32_t *__restrict__ idx, const uint32_t *__restrict__ n_valid,
    const int64_t *__restrict__ block_sums, OutT *__restrict__ out, int32_t raw_sums) {
    __shared__ int64_t s_wave[GS_BLOCK / GS_WAVE];
    const uint64_t nv = n_valid != nullptr ? min(n, (uint64_t)*n_valid) : n;
    // positions behind the valid prefix carry no element: `out` stays unwritten from the first block that lies entirely
    // behind it (gs_isect_emit* never reads there when it is handed the same n_valid)
    if (n_valid != nullptr && (uint64_t)blockIdx.x * SCAN_TILE >= nv) return;
    // raw_sums: block_sums holds the per-block totals themselves (no spine launch for up to a few thousand blocks):
    // the offset of this block is the sum of its predecessors' totals
    int64_t offset;
    if (raw_sums) {
        int64_t pre = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += GS_BLOCK) pre += block_sums[i];
        int64_t tot;
        block_exclusive_scan_i64(pre, tot, s_wave);
        offset = tot;
    } else {
        offset = block_sums[blockIdx.x];
    }
    // blocked arrangement: thread t owns items [t*ITEMS, (t+1)*ITEMS)
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    int32_t v[SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + k;
        v[k] = i < nv ? (idx != nullptr ? in[idx[i]] : in[i]) : 0;
        s += v[k];
    }
    int64_t total;
    int64_t run = block_exclusive_scan_i64(s, total, s_wave) + offset;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + k;
        run += v[k];
        if (i < n) out[i] = (OutT)run;
    }
}

template <typename OutT>
int32_t cumsum_impl(uint64_t n, const int32_t *in, const int32_t *idx, const uint32_t *n_valid, OutT *out, void *scratch,
                    size_t scratch_bytes, hipStream_t st) {
    if (n == 0) return 0;
    uint32_t n_blocks = gs_div_up(n, SCAN_TILE);
    if (scratch == nullptr || scratch_bytes < (size_t)n_blocks * sizeof(int64_t)) {
        gs_set_error("gs_cumsum: scratch too small (%zu < %zu)", scratch_bytes, (size_t)n_blocks * sizeof(int64_t));
        return 1;
    }
    int64_t *sums = (int64_t *)scratch;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(n_blocks), dim3(GS_BLOCK), 0, st, n, in, idx, n_valid, sums);
    const int32_t raw = n_blocks <= 4096u ? 1 : 0; // every block sums its predecessors' totals itself (<= 16 loads per thread)
    if (!raw) hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(GS_BLOCK), 0, st, n_blocks, sums);
    hipLaunchKernelGGL((scan_apply_kernel<OutT>), dim3(n_blocks), dim3(GS_BLOCK), 0, st, n, in, idx, n_valid, sums, out, raw);
    return 0;
}

} // namespace

extern "C" int32_t gs_isect_count(
    uint32_t n_elems, const float *means2d, uint32_t means2d_stride, const int32_t *radii, uint32_t tile_size,
    uint32_t tile_width, uint32_t tile_height, int32_t *tiles_per_gauss, gs_stream_t stream) {
    if (n_elems == 0) return 0;
    GS_CHECK_ARG(means2d && radii && tiles_per_gauss, "null pointer");
    GS_CHECK_ARG(tile_size > 0, "tile_size must be > 0");
    GS_CHECK_ARG(means2d_stride >= 2 && means2d_stride % 2 == 0, "means2d_stride must be even and >= 2");
    hipLaunchKernelGGL(isect_count_kernel, dim3(gs_div_up(n_elems, GS_BLOCK)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, n_elems, means2d, means2d_stride, radii, (float)tile_size, (int32_t)tile_width,
                       (int32_t)tile_height, tiles_per_gauss);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t gs_cumsum_scratch_bytes(uint64_t n) {
    return (size_t)(gs_div_up(n, SCAN_TILE) + 1) * sizeof(int64_t);
}

extern "C" int32_t gs_cumsum_i32(
    uint64_t n, const int32_t *in, int64_t *out, void *scratch, size_t scratch_bytes, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(in && out, "null pointer");
    int32_t rc = cumsum_impl<int64_t>(n, in, nullptr, nullptr, out, scratch, scratch_bytes, (hipStream_t)stream);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_cumsum_i32_i32(
    uint64_t n, const int32_t *in, int32_t *out, void *scratch, size_t scratch_bytes, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(in && out, "null pointer");
    int32_t rc = cumsum_impl<int32_t>(n, in, nullptr, nullptr, out, scratch, scratch_bytes, (hipStream_t)stream);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_cumsum_gather_i32(
    uint64_t n, const int32_t *in, const int32_t *idx, const uint32_t *n_valid, int64_t *out, void *scratch,
    size_t scratch_bytes, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(in && idx && out, "null pointer");
    int32_t rc = cumsum_impl<int64_t>(n, in, idx, n_valid, out, scratch, scratch_bytes, (hipStream_t)stream);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" uint32_t gs_isect_count_blocks(uint32_t n_elems) { return gs_div_up(n_elems, GS_BLOCK * COUNT_ITEMS); }

extern "C" int32_t gs_isect_count_keys(
    uint32_t n_elems, const float *means2d, uint32_t means2d_stride, const int32_t *radii, const float *depths, uint32_t tile_size,
    uint32_t tile_width, uint32_t tile_height, int32_t *tiles_per_gauss, int64_t *keys, int32_t *vals, int32_t *block_sums,
    void *sort_temp, size_t sort_temp_bytes, const int64_t *bucket_splitters, gs_stream_t stream) {
    if (n_elems == 0) return 0;
    GS_CHECK_ARG(radii && depths && tiles_per_gauss && keys && (vals || bucket_splitters), "null pointer (vals may be NULL with bucket_splitters only)");
    GS_CHECK_ARG(means2d != nullptr || block_sums == nullptr, "means2d NULL (tiles_per_gauss given): the block sums were made with the counts");
    GS_CHECK_ARG(bucket_splitters == nullptr || sort_temp != nullptr, "bucket_splitters come with sort_temp (the histogram's place)");
    GS_CHECK_ARG(tile_size > 0, "tile_size must be > 0");
    GS_CHECK_ARG(means2d_stride >= 2 && means2d_stride % 2 == 0, "means2d_stride must be even and >= 2");
    uint32_t n_blocks = 0;
    uint32_t *hist = nullptr;
    if (sort_temp != nullptr) {
        hist = sort_first_hist_slot(n_elems, sort_temp, sort_temp_bytes, &n_blocks);
        GS_CHECK_ARG(hist != nullptr && n_blocks == gs_isect_count_blocks(n_elems),
                     "sort_temp given, but gs_sort_first_hist_applicable(n_elems) is 0 or the buffer is too small");
    }
    hipLaunchKernelGGL(isect_count_keys_kernel, dim3(gs_isect_count_blocks(n_elems)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       n_elems, means2d, means2d_stride, radii, depths, (float)tile_size, (int32_t)tile_width, (int32_t)tile_height,
                       tiles_per_gauss, keys, vals, block_sums, hist, n_blocks, (const uint64_t *)bucket_splitters);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_isect_depth_keys(
    uint32_t n_elems, const int32_t *radii, const float *depths, int64_t *keys, int32_t *vals, gs_stream_t stream) {
    if (n_elems == 0) return 0;
    GS_CHECK_ARG(radii && depths && keys && vals, "null pointer");
    hipLaunchKernelGGL(isect_depth_keys_kernel, dim3(gs_div_up(n_elems, GS_BLOCK)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, n_elems, radii, depths, keys, vals);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_gather_i32(uint32_t n, const int32_t *src, const int32_t *idx, int32_t *out, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(src && idx && out, "null pointer");
    hipLaunchKernelGGL(gather_i32_kernel, dim3(gs_div_up(n, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, src, idx, out);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_isect_emit(
    uint32_t n_elems, uint32_t N, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids, const float *means2d,
    uint32_t means2d_stride, const int32_t *radii, const float *depths, const int64_t *cum_tiles_per_gauss,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits,
    int64_t *isect_ids, int32_t *flatten_ids, gs_stream_t stream) {
    if (n_elems == 0) return 0;
    GS_CHECK_ARG(means2d && radii && depths && cum_tiles_per_gauss, "null pointer");
    GS_CHECK_ARG(camera_ids != nullptr || N > 0, "N must be > 0 when camera_ids is NULL");
    GS_CHECK_ARG(tile_n_bits < 32, "tile_n_bits must be < 32");
    GS_CHECK_ARG(means2d_stride >= 2 && means2d_stride % 2 == 0, "means2d_stride must be even and >= 2");
    hipLaunchKernelGGL(isect_emit_kernel<false>, dim3(gs_div_up(n_elems, EMIT_WAVES * EMIT_SPW)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, n_elems, N, perm, n_valid, camera_ids, means2d, means2d_stride, radii, depths,
                       cum_tiles_per_gauss, (float)tile_size, (int32_t)tile_width, (int32_t)tile_height,
                       tile_n_bits, isect_ids, (uint32_t *)nullptr, flatten_ids);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_isect_emit_compact(
    uint32_t n_elems, uint32_t N, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids, const float *means2d,
    uint32_t means2d_stride, const int32_t *radii, const float *depths, const int64_t *cum_tiles_per_gauss,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits,
    uint32_t *keys32, int32_t *flatten_ids, gs_stream_t stream) {
    if (n_elems == 0) return 0;
    GS_CHECK_ARG(means2d && radii && depths && cum_tiles_per_gauss && keys32 && flatten_ids, "null pointer");
    GS_CHECK_ARG(camera_ids != nullptr || N > 0, "N must be > 0 when camera_ids is NULL");
    GS_CHECK_ARG(tile_n_bits < 32, "tile_n_bits must be < 32");
    GS_CHECK_ARG(means2d_stride >= 2 && means2d_stride % 2 == 0, "means2d_stride must be even and >= 2");
    hipLaunchKernelGGL(isect_emit_kernel<true>, dim3(gs_div_up(n_elems, EMIT_WAVES * EMIT_SPW)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, n_elems, N, perm, n_valid, camera_ids, means2d, means2d_stride, radii, depths,
                       cum_tiles_per_gauss, (float)tile_size, (int32_t)tile_width, (int32_t)tile_height,
                       tile_n_bits, (int64_t *)nullptr, keys32, flatten_ids);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" uint32_t gs_isect_emit_group_shift(void) { return EMIT_SCAN_SHIFT; }
// Above this many groups the emission wants the groups' prefix sum (group_prefix): without it every workgroup adds up its
// predecessors' sums itself, a number of loads quadratic in the number of groups (fine for a 1 M-element frame: 7.9 K groups).
extern "C" uint32_t gs_isect_emit_prefix_from_groups(void) { return 8192u; }

// mode 0 / 1 / 2: see isect_emit_scan_kernel
static int32_t emit_presorted_launch(
    int mode, uint32_t pos_bits, uint32_t n_elems, uint32_t N, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids,
    const float *means2d, uint32_t means2d_stride, const int32_t *radii, const float *depths, const int32_t *tiles_per_gauss,
    const uint32_t *group_sums, const int64_t *group_prefix, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    uint32_t tile_n_bits, int64_t *isect_ids, uint32_t *keys32, int32_t *flatten_ids, gs_stream_t stream) {
    const int32_t compact = mode != 0;
    if (n_elems == 0) return 0;
    GS_CHECK_ARG(means2d && radii && depths && tiles_per_gauss && (group_sums || group_prefix) && (flatten_ids || mode == 2), "null pointer");
    GS_CHECK_ARG(compact ? keys32 != nullptr : isect_ids != nullptr, "null output");
    GS_CHECK_ARG(camera_ids != nullptr || N > 0, "N must be > 0 when camera_ids is NULL");
    GS_CHECK_ARG(tile_n_bits < 32, "tile_n_bits must be < 32");
    GS_CHECK_ARG(means2d_stride >= 2 && means2d_stride % 2 == 0, "means2d_stride must be even and >= 2");
    GS_CHECK_ARG(n_elems < (1u << 31), "n_elems must be < 2^31");
    const dim3 grid(gs_div_up(n_elems, EMIT_SCAN_TILE));
    if (mode == 2)
        hipLaunchKernelGGL(isect_emit_scan_kernel<2>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, n_elems, N, perm, n_valid, camera_ids,
                           means2d, means2d_stride, radii, depths, tiles_per_gauss, group_sums, group_prefix, (float)tile_size, (int32_t)tile_width,
                           (int32_t)tile_height, tile_n_bits, (int64_t *)nullptr, keys32, (int32_t *)nullptr, pos_bits);
    else if (compact)
        hipLaunchKernelGGL(isect_emit_scan_kernel<1>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, n_elems, N, perm, n_valid, camera_ids,
                           means2d, means2d_stride, radii, depths, tiles_per_gauss, group_sums, group_prefix, (float)tile_size, (int32_t)tile_width,
                           (int32_t)tile_height, tile_n_bits, (int64_t *)nullptr, keys32, flatten_ids, 0u);
    else
        hipLaunchKernelGGL(isect_emit_scan_kernel<0>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, n_elems, N, perm, n_valid, camera_ids,
                           means2d, means2d_stride, radii, depths, tiles_per_gauss, group_sums, group_prefix, (float)tile_size, (int32_t)tile_width,
                           (int32_t)tile_height, tile_n_bits, isect_ids, (uint32_t *)nullptr, flatten_ids, 0u);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_isect_emit_presorted(
    uint32_t n_elems, uint32_t N, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids, const float *means2d,
    uint32_t means2d_stride, const int32_t *radii, const float *depths, const int32_t *tiles_per_gauss, const uint32_t *group_sums,
    const int64_t *group_prefix, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits, int32_t compact,
    int64_t *isect_ids, uint32_t *keys32, int32_t *flatten_ids, gs_stream_t stream) {
    return emit_presorted_launch(compact ? 1 : 0, 0u, n_elems, N, perm, n_valid, camera_ids, means2d, means2d_stride, radii, depths,
                                 tiles_per_gauss, group_sums, group_prefix, tile_size, tile_width, tile_height, tile_n_bits, isect_ids, keys32,
                                 flatten_ids, stream);
}

// Packed emission: words[i] = (camera << tile_n_bits | tile) << pos_bits | emission position of the splat (its index in perm:
// the depth rank).  Needs key bits + pos_bits <= 32; see gs_sort_isect_packed.
extern "C" int32_t gs_isect_emit_packed(
    uint32_t n_elems, uint32_t N, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids, const float *means2d,
    uint32_t means2d_stride, const int32_t *radii, const float *depths, const int32_t *tiles_per_gauss, const uint32_t *group_sums,
    const int64_t *group_prefix, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits, uint32_t pos_bits,
    uint32_t *words, gs_stream_t stream) {
    GS_CHECK_ARG(pos_bits + tile_n_bits <= 32, "tile bits + position bits must fit 32 bits");
    return emit_presorted_launch(2, pos_bits, n_elems, N, perm, n_valid, camera_ids, means2d, means2d_stride, radii, depths, tiles_per_gauss,
                                 group_sums, group_prefix, tile_size, tile_width, tile_height, tile_n_bits, nullptr, words, nullptr, stream);
}

extern "C" int32_t gs_isect_offset_encode(
    uint32_t n_isects, const int64_t *isect_ids_sorted, uint32_t C, uint32_t n_tiles,
    uint32_t tile_n_bits, int32_t *offsets, gs_stream_t stream) {
    GS_CHECK_ARG(offsets != nullptr, "null pointer");
    if ((uint64_t)C * n_tiles == 0) return 0;
    if (n_isects == 0) {
        // reference: offsets.fill_(0)  (isect_tiles.cu:385-387)
        hipError_t e = hipMemsetAsync(offsets, 0, (size_t)C * n_tiles * sizeof(int32_t), (hipStream_t)stream);
        if (e != hipSuccess) {
            gs_set_error("gs_isect_offset_encode: memset failed: %s", hipGetErrorString(e));
            return 2;
        }
        return 0;
    }
    GS_CHECK_ARG(isect_ids_sorted != nullptr, "null pointer");
    hipLaunchKernelGGL(isect_offset_encode_kernel, dim3(gs_div_up(n_isects, GS_BLOCK * OFFSET_ITEMS)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, n_isects, isect_ids_sorted, C, n_tiles, tile_n_bits, offsets);
    GS_CHECK_LAUNCH();
    return 0;
}

// emit (compact) -> 32-bit pair sort -> offsets, one call: see the header
static size_t finish_pairs_bytes(uint64_t n) { return ((size_t)n * 4 + 255) / 256 * 256; }

extern "C" size_t gs_isect_finish_work_bytes(uint64_t n_isects) {
    return 2 * finish_pairs_bytes(n_isects) + gs_sort_isect_temp_bytes(n_isects);
}

extern "C" int32_t gs_isect_finish_presorted(
    uint32_t n_elems, uint32_t N, uint64_t n_isects, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids,
    const float *means2d, uint32_t means2d_stride, const int32_t *radii, const float *depths, const int32_t *tiles_per_gauss,
    const uint32_t *group_sums, const int64_t *group_prefix, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    uint32_t tile_n_bits, uint32_t cam_n_bits, uint32_t C, int64_t *isect_ids, int32_t *flatten_ids, int32_t *offsets, void *work,
    size_t work_bytes, uint32_t n_kept_host, const int64_t *sorted_keys, gs_stream_t stream) {
    GS_CHECK_ARG(offsets != nullptr, "null pointer");
    if (n_isects > 0) {
        GS_CHECK_ARG(work != nullptr && (uintptr_t)work % 16 == 0 && work_bytes >= gs_isect_finish_work_bytes(n_isects),
                     "work buffer too small (gs_isect_finish_work_bytes) or misaligned");
        const size_t pb = finish_pairs_bytes(n_isects);
        uint32_t *keys32 = (uint32_t *)work;
        int32_t *vals = (int32_t *)((char *)work + pb);
        void *temp = (char *)work + 2 * pb;
        // PACKED route: the caller knows how many splats the pre-sort kept (n_kept_host, e.g. from the projection's block sums)
        // and (camera, tile) key + emission position fit ONE 32-bit word -- 4 bytes per pair through emission and sort instead
        // of 8, no value array.  The key bits that can be set: the tile bits + the bits of the largest camera index.
        uint32_t pos_bits = 0, cam_bits = 0;
        while (n_kept_host > 0 && ((uint64_t)1 << pos_bits) < (uint64_t)n_kept_host) ++pos_bits;
        while (((uint32_t)1 << cam_bits) < C) ++cam_bits;
        const uint32_t key_bits_eff = tile_n_bits + cam_bits;
        if (n_kept_host > 0 && perm != nullptr && camera_ids == nullptr && key_bits_eff >= 1 && key_bits_eff + pos_bits <= 32 &&
            key_bits_eff <= 31) {
            int32_t rc = gs_isect_emit_packed(n_elems, N, perm, n_valid, camera_ids, means2d, means2d_stride, radii, depths, tiles_per_gauss,
                                              group_sums, group_prefix, tile_size, tile_width, tile_height, tile_n_bits, pos_bits, keys32, stream);
            if (rc != 0) return rc;
            rc = gs_sort_isect_packed(n_isects, keys32, perm, sorted_keys, depths, (int32_t)key_bits_eff, pos_bits, isect_ids, flatten_ids,
                                      temp, work_bytes - 2 * pb, stream);
            if (rc != 0) return rc;
            return gs_isect_offset_encode((uint32_t)n_isects, isect_ids, C, tile_width * tile_height, tile_n_bits, offsets, stream);
        }
        int32_t rc = gs_isect_emit_presorted(n_elems, N, perm, n_valid, camera_ids, means2d, means2d_stride, radii, depths, tiles_per_gauss,
                                             group_sums, group_prefix, tile_size, tile_width, tile_height, tile_n_bits, 1, nullptr,
                                             keys32, vals, stream);
        if (rc != 0) return rc;
        // (only the bits a key can have set take part: 8 cameras are 3 bits, not the 4 of the id layout -- 16 key bits = two passes
        // instead of three; with all 32 bits in use the last pass also orders the ids as the signed values they are: left alone)
        const uint32_t sort_bits = (tile_n_bits + cam_n_bits < 32u && key_bits_eff >= 1u) ? key_bits_eff : tile_n_bits + cam_n_bits;
        rc = gs_sort_isect_pairs(n_isects, keys32, vals, depths, (int32_t)sort_bits, isect_ids, flatten_ids, temp,
                                 work_bytes - 2 * pb, stream);
        if (rc != 0) return rc;
    }
    return gs_isect_offset_encode((uint32_t)n_isects, isect_ids, C, tile_width * tile_height, tile_n_bits, offsets, stream);
}
