// radix_sort.hip -- stable LSD radix sort of (uint64 key, int32 value) pairs (gfx950).
//
// Replaces cub::DeviceRadixSort::SortPairs as used by the reference's tile binning
// (gsplat/cuda/csrc/isect_tiles.cu:245-299): keys are
//   camera_id << (32 + tile_bits) | tile_id << 32 | depth_bits
// and only bits [0, 32 + tile_bits + cam_bits) are significant.  Stability matters:
// equal (camera, tile, depth) keys must keep their emission order, as CUB's do.
//
// Structure per 8-bit digit pass (3 launches):
//   1. sort_hist_kernel   : per-block digit histogram -> hist[digit][block]
//   2. sort_scan_kernel   : one block per digit, exclusive scan over blocks (+ digit total)
//   3. sort_scatter_kernel: stable local ranking with wave64 ballot matching, scatter.
// A block owns 4096 (large inputs) or 1024 consecutive keys; wave w owns a quarter of them and walks them in 16 (4)
// rounds of 64, so (wave, round, lane) order == index order, which is what makes the
// ballot-based rank stable.  Keys stay in VGPRs between the ranking and scatter phases; the scatter
// itself goes through LDS (block-sorted order first, then coalesced runs per digit).
// Digits: 8 bits, except that the FIRST pass takes the remainder (14 bits -> 6 + 8).
//
// Two key widths share the kernels: 64-bit keys (the public gs_sort_pairs_u64_i32 entry points, and the splat-level
// depth pre-sort of the tile binning) and 32-bit (camera, tile) keys for the 4 M (tile, splat) pairs of the binning, whose
// LAST pass writes the reference's 64-bit intersection ids directly (gs_sort_isect_pairs): the pairs travel as 8 bytes
// instead of 12 through every pass, and the depth bits of the ids are gathered once, at the end.
// (A one-launch-per-pass "onesweep" form -- blocks publish their digit counts and pick up their predecessors' inside the
// launch, grouped so that no look-back chain forms -- was built and measured in round 2: correct, but 68 us per 4 M-pair
// pass against 45 us for the three launches; in-launch hand-offs cost more than kernel boundaries here.)
#include "gs_common.h"

namespace {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_WAVES = GS_BLOCK / GS_WAVE; // 4
// Keys per block = 256 * ROUNDS.  16 rounds (4096 keys) for large inputs; 4 rounds (1024 keys) up to SORT_SMALL_N
// keys: there a pass is bound by the LATENCY of one block (all blocks are resident at once: 17 us per scatter
// launch at 1 M keys with 4096-key blocks, whatever the number of live keys), so shorter blocks win.
constexpr int SORT_ROUNDS_BIG = 16, SORT_ROUNDS_SMALL = 4;
constexpr uint64_t SORT_SMALL_N = 2u << 20;
constexpr int sort_tile(int rounds) { return GS_BLOCK * rounds; }
inline int sort_rounds_for(uint64_t n) { return n <= SORT_SMALL_N ? SORT_ROUNDS_SMALL : SORT_ROUNDS_BIG; }

struct DigitSpec {
    uint32_t shift;
    uint32_t mask;
    uint32_t flip; // xor applied to the digit (sign bit handling when end_bit == 64)
    uint32_t drop; // != 0: keys whose upper 32 bits equal drop_hi are DROPPED by this pass (not counted, not written)
    uint32_t drop_hi;
    const uint64_t *split; // bucketed pre-sort (BUCKET kernels only): digit = gs_bucket_of(split, key) instead of key bits
};

GS_DEV bool key_kept(uint64_t key, DigitSpec d) { return !(d.drop != 0u && (uint32_t)(key >> 32) == d.drop_hi); }
GS_DEV bool key_kept(uint32_t, DigitSpec) { return true; } // (dropping is a feature of the 64-bit depth keys)

GS_DEV uint32_t digit_of(uint64_t key, DigitSpec d) { return ((uint32_t)(key >> d.shift) & d.mask) ^ d.flip; }
GS_DEV uint32_t digit_of(uint32_t key, DigitSpec d) { return ((key >> d.shift) & d.mask) ^ d.flip; }

// What the last pass over the binning's 32-bit keys writes instead of (key, value): the reference's 64-bit intersection id
// camera << (32 + tile_bits) | tile << 32 | depth bits (isect_tiles.cu:89-103) -- the 32-bit sort key IS its upper half --
// and the flatten id.
struct IsectEpilogue {
    const float *depths; // indexed by the flatten id
    int64_t *isect_ids;
    int32_t *flatten_ids;
    // PACKED pairs (KEYS_ONLY kernels): a pair is ONE word  key << pos_bits | emission position; the flatten id is perm[position]
    const int32_t *perm;
    uint32_t pos_bits;
    const uint64_t *sorted_keys; // or NULL: the pre-sort's sorted (depth bits << 32 | element) keys by position -- flatten id AND
                                 // depth bits in ONE 8-byte gather instead of the dependent pair perm[position] -> depths[id]
};

// What a scatter launch produces ON THE SIDE while it places its keys:
//   side_sums [n >> side_shift]  sums of side_vals[value] over groups of 2^side_shift consecutive OUTPUT positions -- the
//                                last pass of the splat-level pre-sort hands the tile counts per group of emission positions
//                                to gs_isect_emit_presorted this way (the block sums of its prefix scan: the two cumsum
//                                launches and the cum_tiles array are gone, 34 -> 25 us for count + emit).
// (Round 3 also had the scatter count the NEXT pass's digits per destination block, to drop the histogram launch of every
// pass: global atomics, merged over runs of equal counters inside a wave.  Measured: the four scatter launches went from
// 10.5 / 8.9 / 8.1 / 7.5 us to 25.0 / 23.9 / 47.8 / 10.6 us -- 293 K device-scope atomics cost more than a 6 us launch even
// without contention, and the concentrated upper digits serialise on a few hundred counters.  Removed.)
struct ScatterSide {
    const int32_t *side_vals;
    uint32_t *side_sums;
    uint32_t side_shift;
};

template <typename KeyT, int SORT_ROUNDS>
__global__ void __launch_bounds__(GS_BLOCK) sort_hist_kernel(
    uint64_t n, const uint32_t *__restrict__ n_dev, const KeyT *__restrict__ keys, DigitSpec d, uint32_t n_blocks,
    uint32_t *__restrict__ hist /* [RADIX][n_blocks] */) {
    constexpr int SORT_TILE = sort_tile(SORT_ROUNDS);
    __shared__ uint32_t s_hist[RADIX];
    s_hist[threadIdx.x] = 0;
    if (n_dev != nullptr) n = min(n, (uint64_t)*n_dev); // element count known only on the device
    if ((uint64_t)blockIdx.x * SORT_TILE >= n) { // nothing here (the grid is sized for the host-side upper bound)
        hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = 0;
        return;
    }
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
    if (sizeof(KeyT) == 4 && SORT_ROUNDS % 4 == 0 && base + SORT_TILE <= n && (uintptr_t)keys % 16 == 0) {
        // a full block of 32-bit keys: the order inside the block does not matter for a histogram, so every thread takes
        // four consecutive keys per 16-byte load and ALL its loads are in flight at once (the kernel is bound by the latency
        // of one block: 9.8 -> 8.1 us per 4 M-key launch)
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + base);
        uint4 v[SORT_ROUNDS / 4];
#pragma unroll
        for (int k = 0; k < SORT_ROUNDS / 4; ++k) v[k] = k4[k * GS_BLOCK + threadIdx.x];
#pragma unroll
        for (int k = 0; k < SORT_ROUNDS / 4; ++k) {
            atomicAdd(&s_hist[digit_of((KeyT)v[k].x, d)], 1u);
            atomicAdd(&s_hist[digit_of((KeyT)v[k].y, d)], 1u);
            atomicAdd(&s_hist[digit_of((KeyT)v[k].z, d)], 1u);
            atomicAdd(&s_hist[digit_of((KeyT)v[k].w, d)], 1u);
        }
    } else {
#pragma unroll 4
    for (int k = 0; k < SORT_TILE / GS_BLOCK; ++k) {
        uint64_t i = base + (uint64_t)k * GS_BLOCK + threadIdx.x;
        if (i < n) {
            const KeyT key = keys[i];
            if (key_kept(key, d)) atomicAdd(&s_hist[digit_of(key, d)], 1u);
        }
    }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s_hist[threadIdx.x];
}

// block = digit.  In-place exclusive scan of hist[digit][0..n_blocks) and digit total.
// zero_side (optional, n_side entries): the buffer the FOLLOWING scatter launch accumulates its side sums into -- this launch
// runs right before it, so the buffer is zeroed here instead of by a fill launch of its own.
__global__ void __launch_bounds__(GS_BLOCK) sort_scan_kernel(
    uint32_t n_blocks, uint32_t *__restrict__ hist, uint32_t *__restrict__ totals, uint32_t *__restrict__ zero_side, uint32_t n_side) {
    __shared__ uint32_t s_wave[SORT_WAVES];
    uint32_t *row = hist + (size_t)blockIdx.x * n_blocks;
    uint32_t lane = threadIdx.x % GS_WAVE, wave = threadIdx.x / GS_WAVE;
    uint32_t carry = 0;
    if (zero_side != nullptr)
        for (uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x; i < n_side; i += RADIX * GS_BLOCK) zero_side[i] = 0u;
    for (uint32_t base = 0; base < n_blocks; base += GS_BLOCK) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n_blocks ? row[i] : 0;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += o;
        }
        if (lane == GS_WAVE - 1) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            if ((uint32_t)w < wave) wbase += s_wave[w];
            total += s_wave[w];
        }
        if (i < n_blocks) row[i] = carry + wbase + inc - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the global
// STORES of the keys to be acknowledged before the values may be staged.
GS_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// BUCKET: the one partition pass of the bucketed depth pre-sort -- the digit of a (64-bit) key is its bucket among 255
// sampled splitters (table in LDS, 8 reads per lookup) instead of 8 key bits; everything else is the same stable scatter.
// KEYS_ONLY: no value array rides along (the binning's packed pairs: the low bits of the word ARE the value) -- the second
// LDS trip and half of the global traffic of a pass are gone.
template <typename KeyT, int SORT_ROUNDS, bool FINAL_ISECT, bool BUCKET = false, bool KEYS_ONLY = false>
__global__ void __launch_bounds__(GS_BLOCK) sort_scatter_kernel(
    uint64_t n, const uint32_t *__restrict__ n_dev, const KeyT *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
    KeyT *__restrict__ keys_out, int32_t *__restrict__ vals_out, DigitSpec d,
    uint32_t n_blocks, const uint32_t *__restrict__ hist_scan, const uint32_t *__restrict__ totals,
    uint32_t *__restrict__ n_kept_out /* or NULL: block 0 also publishes the number of keys this pass kept */, IsectEpilogue ep,
    ScatterSide side) {
    constexpr int SORT_TILE = sort_tile(SORT_ROUNDS);
    constexpr int SORT_WAVE_KEYS = GS_WAVE * SORT_ROUNDS;
    if (n_kept_out != nullptr && blockIdx.x == 0 && threadIdx.x < GS_WAVE) { // sum of the 256 digit totals (one wave)
        uint32_t v = totals[threadIdx.x] + totals[threadIdx.x + 64] + totals[threadIdx.x + 128] + totals[threadIdx.x + 192];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (threadIdx.x == 0) *n_kept_out = v;
    }
    __shared__ uint32_t s_cnt[SORT_WAVES][RADIX]; // per-wave digit counters -> per-wave prefix
    __shared__ uint32_t s_lbase[RADIX];           // base of the digit inside this block's sorted order
    __shared__ uint32_t s_gofs[RADIX];            // global base of (digit, this block) - s_lbase
    __shared__ uint32_t s_scan[SORT_WAVES];
    __shared__ KeyT s_keys[SORT_TILE];            // staging (keys, then values): 32 KB for 4096 64-bit keys
    __shared__ uint32_t s_count;
    __shared__ uint64_t s_split[BUCKET ? GS_PRESORT_BUCKETS : 1];
    const uint32_t tid = threadIdx.x, lane = tid % GS_WAVE, wave = tid / GS_WAVE;
    if (n_dev != nullptr) n = min(n, (uint64_t)*n_dev);
    if ((uint64_t)blockIdx.x * SORT_TILE >= n) return; // block-uniform
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) s_cnt[w][tid] = 0;
    if (BUCKET) s_split[BUCKET ? tid : 0] = d.split[tid];
    __syncthreads();
    auto digit = [&](KeyT k) -> uint32_t {
        if constexpr (BUCKET) return gs_bucket_of(s_split, (uint64_t)k);
        else return digit_of(k, d);
    };

    const uint64_t wave_base = (uint64_t)blockIdx.x * SORT_TILE + (uint64_t)wave * SORT_WAVE_KEYS;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    KeyT key[SORT_ROUNDS];
    uint32_t rank[SORT_ROUNDS];
    uint32_t kept = 0; // bit r: the key of round r takes part in this pass

    // phase 1: stable rank of every key within (wave, digit).  Per round, the lanes holding the same digit are matched by
    // ballots; the FIRST of them (the leader) adds the group's size to the wave's LDS counter with a returning atomic and
    // the others pick the returned base up from the leader's lane.  The LDS operations of one wave execute in order, so
    // the 16 rounds' atomics are issued back to back -- no wave barrier and no LDS round trip between rounds (the
    // read / barrier / write / barrier form cost two dependent LDS round trips per round: 32 per block).
    // the keys, the block's scanned histogram column and the digit totals are requested before anything waits
    // (also holding the VALUES in registers from here measured slower for the 4096-key blocks: 26.8 vs 25.0 us per pass)
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const uint64_t i = wave_base + (uint64_t)r * GS_WAVE + lane;
        key[r] = (i < n) ? keys_in[i] : (KeyT)0;
    }
    const uint32_t my_total = totals[tid], my_hist = hist_scan[(size_t)tid * n_blocks + blockIdx.x];
    uint32_t leader[SORT_ROUNDS];
    uint8_t dgv[BUCKET ? SORT_ROUNDS : 1]; // (a bucket lookup costs 8 LDS reads: kept for the placement below)
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const uint64_t i = wave_base + (uint64_t)r * GS_WAVE + lane;
        const bool valid = i < n && key_kept(key[r], d);
        kept |= valid ? (1u << r) : 0u;
        const uint32_t dg = digit(key[r]);
        if (BUCKET) dgv[BUCKET ? r : 0] = (uint8_t)dg;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            if (!BUCKET && ((d.mask >> b) & 1u) == 0u) break; // (uniform) narrower first digit: fewer ballots
            const bool bit = (dg >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = __popcll(peers & lt_mask);
        leader[r] = valid ? (uint32_t)__builtin_ctzll(peers) : lane;
        uint32_t base = 0;
        if (valid && before == 0) base = atomicAdd(&s_cnt[wave][dg], (uint32_t)__popcll(peers)); // ds_add_rtn_u32
        rank[r] = base + before; // leaders: final; the others add their leader's base below
    }
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const uint32_t lead_rank = __shfl(rank[r], (int)leader[r], 64); // leader: before == 0, so its rank IS the base
        if (leader[r] != lane) rank[r] += lead_rank;
    }
    __syncthreads();

    // phase 2: digit tid -> prefix over waves, the digit's base inside this block's sorted order, and
    // (global base of (digit, block)) - (local base): a key at sorted position j of the block goes to s_gofs[dg] + j
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            uint32_t c = s_cnt[w][tid];
            s_cnt[w][tid] = run;
            run += c;
        }
        auto block_excl_scan = [&](uint32_t t) { // exclusive scan over the 256 threads (digits)
            uint32_t inc = t;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = __shfl_up(inc, off, 64);
                if (lane >= (uint32_t)off) inc += o;
            }
            __syncthreads(); // s_scan may still be read by the previous call
            if (lane == GS_WAVE - 1) s_scan[wave] = inc;
            __syncthreads();
            uint32_t wbase = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w)
                if ((uint32_t)w < wave) wbase += s_scan[w];
            return wbase + inc - t;
        };
        const uint32_t gbase = block_excl_scan(my_total) + my_hist;
        const uint32_t lbase = block_excl_scan(run);
        s_lbase[tid] = lbase;
        if (tid == RADIX - 1) s_count = lbase + run; // keys of this block that take part
        s_gofs[tid] = gbase - lbase; // (unsigned wrap-around is fine: only s_gofs[dg] + j is used)
    }
    __syncthreads();

    // phase 3: scatter THROUGH LDS -- keys are first placed in the block's sorted order in LDS, then
    // written out with consecutive lanes on consecutive addresses (runs of one digit).  Writing straight
    // from the ranking registers made every lane of a store hit a different 8-byte location: the first
    // pass of the pair sort (random low tile bits) took 92 us for 4 M pairs against 26 us for the second.
    const uint32_t block_count = s_count;
    uint32_t lp[SORT_ROUNDS];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        uint32_t dg = BUCKET ? (uint32_t)dgv[BUCKET ? r : 0] : digit_of(key[r], d);
        lp[r] = s_lbase[dg] + s_cnt[wave][dg] + rank[r];
        if ((kept >> r) & 1u) s_keys[lp[r]] = key[r];
    }
    lds_barrier();
    uint32_t pos[SORT_ROUNDS];
    KeyT kept_key[FINAL_ISECT ? SORT_ROUNDS : 1]; // the final pass of the binning needs the key again next to the value
#pragma unroll
    for (int k = 0; k < SORT_ROUNDS; ++k) {
        const uint32_t j = (uint32_t)k * GS_BLOCK + tid;
        pos[k] = 0;
        if (j < block_count) {
            const KeyT kk = s_keys[j];
            pos[k] = s_gofs[digit(kk)] + j;
            if (FINAL_ISECT && KEYS_ONLY) {
                // the word's low bits are the emission position = the depth rank: flatten id and depth through perm
                const uint32_t w32 = (uint32_t)kk;
                const uint32_t at = w32 & ((1u << ep.pos_bits) - 1u);
                int32_t v;
                int64_t db;
                if (ep.sorted_keys != nullptr) { // (uniform) depths of visible splats are positive: their bits ARE the key's high half
                    const uint64_t sk = ep.sorted_keys[at];
                    v = (int32_t)(uint32_t)sk;
                    db = (int64_t)(sk >> 32);
                } else {
                    v = ep.perm[at];
                    db = (int64_t)__float_as_int(ep.depths[v]);
                }
                ep.isect_ids[pos[k]] = (int64_t)((uint64_t)(w32 >> ep.pos_bits) << 32) | db;
                ep.flatten_ids[pos[k]] = v;
            } else if (FINAL_ISECT) kept_key[FINAL_ISECT ? k : 0] = kk;
            else keys_out[pos[k]] = kk;
        }
    }
    if (KEYS_ONLY) return; // (block-uniform)
    lds_barrier();
    int32_t *s_vals = reinterpret_cast<int32_t *>(s_keys); // the same LDS, second trip for the values
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const uint64_t i = wave_base + (uint64_t)r * GS_WAVE + lane;
        if ((kept >> r) & 1u) s_vals[lp[r]] = vals_in[i];
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < SORT_ROUNDS; ++k) {
        const uint32_t j = (uint32_t)k * GS_BLOCK + tid;
        if (j < block_count) {
            const int32_t v = s_vals[j];
            if (FINAL_ISECT) {
                // (int64_t)*(int32_t*)&depth of the reference (isect_tiles.cu:91); depths of visible splats are positive
                const int64_t db = (int64_t)__float_as_int(ep.depths[v]);
                ep.isect_ids[pos[k]] = (int64_t)((uint64_t)(uint32_t)kept_key[FINAL_ISECT ? k : 0] << 32) | db;
                ep.flatten_ids[pos[k]] = v;
            } else {
                vals_out[pos[k]] = v;
            }
        }
        // side sums: consecutive lanes hold consecutive output positions inside a digit run, i.e. mostly the same group:
        // segmented wave sums, one atomic per (wave, run of equal groups) instead of one per key on a shared counter
        if (!FINAL_ISECT && side.side_sums != nullptr) { // (block-uniform)
            const bool on = j < block_count;
            const uint32_t grp = on ? (pos[k] >> side.side_shift) : 0xffffffffu;
            uint32_t inc = on ? (uint32_t)side.side_vals[s_vals[j]] : 0u;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, 64);
                if (lane >= (uint32_t)off) inc += o;
            }
            const uint32_t prevg = __shfl_up(grp, 1, 64);
            const bool head = lane == 0u || grp != prevg;
            const unsigned long long hm = __ballot(head);
            const unsigned long long above = lane == 63u ? 0ull : (hm >> (lane + 1u));
            const uint32_t last = above ? lane + (uint32_t)__builtin_ctzll(above) : 63u; // last lane of my run
            const uint32_t upto = __shfl(inc, (int)last, 64);
            const uint32_t before = __shfl_up(inc, 1, 64);
            if (head && on) atomicAdd(&side.side_sums[grp], upto - (lane == 0u ? 0u : before));
        }
    }
}

struct SortLayout {
    uint32_t n_blocks;
    size_t off_keys, off_vals, off_hist, off_totals, total;
};

SortLayout sort_layout(uint64_t n) {
    SortLayout L;
    L.n_blocks = gs_div_up(n, sort_tile(sort_rounds_for(n)));
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o += (bytes + 255) & ~(size_t)255;
        return r;
    };
    L.off_keys = take(n * sizeof(uint64_t));
    L.off_vals = take(n * sizeof(int32_t));
    L.off_hist = take((size_t)RADIX * L.n_blocks * sizeof(uint32_t));
    L.off_totals = take(RADIX * sizeof(uint32_t));
    L.total = o;
    return L;
}

} // namespace

extern "C" size_t gs_sort_temp_bytes(uint64_t n) { return sort_layout(n).total; }

// Where the FIRST pass of a sort of n 64-bit keys expects its per-block digit histogram inside `temp` ([256][n_blocks],
// blocks of 1024 consecutive keys), for a producer of the keys that can count the digits while it writes them
// (gs_isect_count_keys): nullptr when the sort would use its 4096-key blocks or temp is too small.
extern "C" int32_t gs_sort_first_hist_applicable(uint64_t n) { return (n > 0 && sort_rounds_for(n) == SORT_ROUNDS_SMALL) ? 1 : 0; }

uint32_t *sort_first_hist_slot(uint64_t n, void *temp, size_t temp_bytes, uint32_t *n_blocks) {
    if (n == 0 || temp == nullptr || sort_rounds_for(n) != SORT_ROUNDS_SMALL) return nullptr;
    const SortLayout L = sort_layout(n);
    if (temp_bytes < L.total) return nullptr;
    *n_blocks = L.n_blocks;
    return (uint32_t *)((char *)temp + L.off_hist);
}

static int32_t sort_impl(uint64_t n, const int64_t *keys_in, const int32_t *vals_in, int64_t *keys_out, int32_t *vals_out,
                         int32_t begin_bit, int32_t end_bit, bool drop, uint32_t drop_hi, uint32_t *n_valid_out, void *temp,
                         size_t temp_bytes, hipStream_t st, const char *who, bool first_hist_ready = false,
                         const int32_t *side_vals = nullptr, uint32_t *side_sums = nullptr, uint32_t side_shift = 0) {
    if (n == 0) return 0;
    int passes = (end_bit - begin_bit + RADIX_BITS - 1) / RADIX_BITS;
    if (passes == 0) {
        if (drop) {
            gs_set_error("%s: dropping keys needs at least one sorted bit", who);
            return 1;
        }
        (void)hipMemcpyAsync(keys_out, keys_in, n * sizeof(int64_t), hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(vals_out, vals_in, n * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
        return 0;
    }
    SortLayout L = sort_layout(n);
    if (temp == nullptr || temp_bytes < L.total) {
        gs_set_error("%s: temp too small (%zu < %zu)", who, temp_bytes, L.total);
        return 1;
    }
    char *tp = (char *)temp;
    uint64_t *tkeys = (uint64_t *)(tp + L.off_keys);
    int32_t *tvals = (int32_t *)(tp + L.off_vals);
    uint32_t *hist = (uint32_t *)(tp + L.off_hist);
    uint32_t *totals = (uint32_t *)(tp + L.off_totals);
    const uint32_t n_side = side_sums != nullptr ? (uint32_t)((n + (1ull << side_shift) - 1) >> side_shift) : 0u;

    // ping-pong between {temp, out}; the first destination is chosen so the last pass
    // lands in *_out without ever writing the inputs.
    const uint64_t *src_k = (const uint64_t *)keys_in;
    const int32_t *src_v = vals_in;
    bool to_out = (passes % 2) == 1;
    // the FIRST pass takes the remainder bits (e.g. 14 bits -> 6 + 8): it meets the data in its least
    // ordered state, and fewer bins there mean longer runs per bin and block
    const int first_bits = (end_bit - begin_bit) - (passes - 1) * RADIX_BITS;
    int shift = begin_bit;
    const uint32_t *n_dev = nullptr; // after a dropping first pass the element count lives on the device
    const bool small = sort_rounds_for(n) == SORT_ROUNDS_SMALL;
    auto digit_spec = [&](int p, int at_shift) {
        DigitSpec d;
        d.shift = (uint32_t)at_shift;
        const int bits = (p == 0) ? first_bits : RADIX_BITS;
        d.mask = (1u << bits) - 1u;
        // int64 keys: when the range includes bit 63 CUB orders them as signed values
        d.flip = (end_bit == 64 && p == passes - 1) ? (1u << (bits - 1)) : 0u;
        d.drop = (drop && p == 0) ? 1u : 0u;
        d.drop_hi = drop_hi;
        d.split = nullptr;
        return d;
    };
    for (int p = 0; p < passes; ++p) {
        const DigitSpec d = digit_spec(p, shift);
        shift += (p == 0) ? first_bits : RADIX_BITS;
        const bool last = p == passes - 1;
        uint64_t *dst_k = to_out ? (uint64_t *)keys_out : tkeys;
        int32_t *dst_v = to_out ? vals_out : tvals;
        if (p == 0 && first_hist_ready && small) {
            // the producer of the keys counted this pass's digits already (see sort_first_hist_slot)
        } else if (small)
            hipLaunchKernelGGL((sort_hist_kernel<uint64_t, SORT_ROUNDS_SMALL>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, n_dev, src_k, d, L.n_blocks, hist);
        else
            hipLaunchKernelGGL((sort_hist_kernel<uint64_t, SORT_ROUNDS_BIG>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, n_dev, src_k, d, L.n_blocks, hist);
        ScatterSide side = {nullptr, nullptr, 0u};
        if (last && side_sums != nullptr) side = {side_vals, side_sums, side_shift};
        hipLaunchKernelGGL(sort_scan_kernel, dim3(RADIX), dim3(GS_BLOCK), 0, st, L.n_blocks, hist, totals, side.side_sums, n_side);
        if (small)
            hipLaunchKernelGGL((sort_scatter_kernel<uint64_t, SORT_ROUNDS_SMALL, false>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, n_dev, src_k, src_v,
                               dst_k, dst_v, d, L.n_blocks, hist, totals, (drop && p == 0) ? n_valid_out : nullptr, IsectEpilogue{}, side);
        else
            hipLaunchKernelGGL((sort_scatter_kernel<uint64_t, SORT_ROUNDS_BIG, false>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, n_dev, src_k, src_v,
                               dst_k, dst_v, d, L.n_blocks, hist, totals, (drop && p == 0) ? n_valid_out : nullptr, IsectEpilogue{}, side);
        if (drop && p == 0) n_dev = n_valid_out;
        src_k = dst_k;
        src_v = dst_v;
        to_out = !to_out;
    }
    return 0;
}

extern "C" int32_t gs_sort_pairs_u64_i32(
    uint64_t n, const int64_t *keys_in, const int32_t *vals_in, int64_t *keys_out,
    int32_t *vals_out, int32_t begin_bit, int32_t end_bit, void *temp, size_t temp_bytes,
    gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(keys_in && vals_in && keys_out && vals_out, "null pointer");
    GS_CHECK_ARG(begin_bit >= 0 && end_bit <= 64 && begin_bit <= end_bit, "bad bit range");
    GS_CHECK_ARG(n < (1ull << 32), "n must be < 2^32");
    int32_t rc = sort_impl(n, keys_in, vals_in, keys_out, vals_out, begin_bit, end_bit, false, 0u, nullptr, temp, temp_bytes,
                           (hipStream_t)stream, "gs_sort_pairs_u64_i32");
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_sort_pairs_u64_i32_drop(
    uint64_t n, const int64_t *keys_in, const int32_t *vals_in, int64_t *keys_out,
    int32_t *vals_out, int32_t begin_bit, int32_t end_bit, uint32_t drop_hi32, uint32_t *n_kept, void *temp,
    size_t temp_bytes, int32_t first_hist_ready, const int32_t *side_vals, uint32_t *side_sums, uint32_t side_shift,
    gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(keys_in && vals_in && keys_out && vals_out && n_kept, "null pointer");
    GS_CHECK_ARG(begin_bit >= 0 && end_bit <= 64 && begin_bit < end_bit, "bad bit range");
    GS_CHECK_ARG(n < (1ull << 32), "n must be < 2^32");
    GS_CHECK_ARG(!first_hist_ready || (begin_bit == 32 && end_bit == 64), "a precomputed first histogram is defined for the bit range [32, 64)");
    GS_CHECK_ARG((side_vals == nullptr) == (side_sums == nullptr) && side_shift < 32, "side_vals and side_sums go together");
    int32_t rc = sort_impl(n, keys_in, vals_in, keys_out, vals_out, begin_bit, end_bit, true, drop_hi32, n_kept, temp, temp_bytes,
                           (hipStream_t)stream, "gs_sort_pairs_u64_i32_drop", first_hist_ready != 0, side_vals, side_sums, side_shift);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Sort of the binning's compact (32-bit (camera, tile) key, flatten id) pairs; the last pass writes the reference's outputs.
// temp layout: keys32[n] | vals[n] | hist[RADIX][n_blocks] | totals[RADIX]
namespace {

struct Sort32Layout {
    uint32_t n_blocks;
    size_t off_keys, off_vals, off_hist, off_totals, total;
};

Sort32Layout sort32_layout(uint64_t n) {
    Sort32Layout L;
    L.n_blocks = gs_div_up(n, sort_tile(sort_rounds_for(n)));
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o += (bytes + 255) & ~(size_t)255;
        return r;
    };
    L.off_keys = take(n * sizeof(uint32_t));
    L.off_vals = take(n * sizeof(int32_t));
    L.off_hist = take((size_t)RADIX * L.n_blocks * sizeof(uint32_t));
    L.off_totals = take(RADIX * sizeof(uint32_t));
    L.total = o;
    return L;
}

template <int ROUNDS>
void launch_pass32(uint64_t n, const uint32_t *src_k, const int32_t *src_v, uint32_t *dst_k, int32_t *dst_v, DigitSpec d,
                   const Sort32Layout &L, uint32_t *hist, uint32_t *totals, bool final, const IsectEpilogue &ep, hipStream_t st) {
    hipLaunchKernelGGL((sort_hist_kernel<uint32_t, ROUNDS>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, (const uint32_t *)nullptr, src_k, d, L.n_blocks, hist);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(RADIX), dim3(GS_BLOCK), 0, st, L.n_blocks, hist, totals, (uint32_t *)nullptr, 0u);
    const ScatterSide none = {nullptr, nullptr, 0u};
    if (final)
        hipLaunchKernelGGL((sort_scatter_kernel<uint32_t, ROUNDS, true>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, (const uint32_t *)nullptr, src_k, src_v,
                           dst_k, dst_v, d, L.n_blocks, hist, totals, (uint32_t *)nullptr, ep, none);
    else
        hipLaunchKernelGGL((sort_scatter_kernel<uint32_t, ROUNDS, false>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, (const uint32_t *)nullptr, src_k, src_v,
                           dst_k, dst_v, d, L.n_blocks, hist, totals, (uint32_t *)nullptr, ep, none);
}

template <int ROUNDS>
void launch_pass32_keys(uint64_t n, const uint32_t *src_k, uint32_t *dst_k, DigitSpec d, const Sort32Layout &L, uint32_t *hist,
                        uint32_t *totals, bool final, const IsectEpilogue &ep, hipStream_t st) {
    hipLaunchKernelGGL((sort_hist_kernel<uint32_t, ROUNDS>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, (const uint32_t *)nullptr, src_k, d, L.n_blocks, hist);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(RADIX), dim3(GS_BLOCK), 0, st, L.n_blocks, hist, totals, (uint32_t *)nullptr, 0u);
    const ScatterSide none = {nullptr, nullptr, 0u};
    if (final)
        hipLaunchKernelGGL((sort_scatter_kernel<uint32_t, ROUNDS, true, false, true>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, (const uint32_t *)nullptr,
                           src_k, (const int32_t *)nullptr, dst_k, (int32_t *)nullptr, d, L.n_blocks, hist, totals, (uint32_t *)nullptr, ep, none);
    else
        hipLaunchKernelGGL((sort_scatter_kernel<uint32_t, ROUNDS, false, false, true>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n, (const uint32_t *)nullptr,
                           src_k, (const int32_t *)nullptr, dst_k, (int32_t *)nullptr, d, L.n_blocks, hist, totals, (uint32_t *)nullptr, ep, none);
}

} // namespace

extern "C" size_t gs_sort_isect_temp_bytes(uint64_t n) { return sort32_layout(n).total; }

// Packed pairs: one 32-bit word per pair, key << pos_bits | emission position.  The positions are the depth ranks, so a stable
// sort on the key bits [pos_bits, pos_bits + key_bits) is all it takes, and nothing but the words moves; the last pass writes
// the reference's outputs through perm.  temp: gs_sort_isect_temp_bytes(n) (the value half stays unused).
extern "C" int32_t gs_sort_isect_packed(uint64_t n, uint32_t *words, const int32_t *perm, const int64_t *sorted_keys, const float *depths,
                                        int32_t key_bits, uint32_t pos_bits, int64_t *isect_ids, int32_t *flatten_ids, void *temp,
                                        size_t temp_bytes, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(words && ((perm && depths) || sorted_keys) && isect_ids && flatten_ids, "null pointer");
    GS_CHECK_ARG(key_bits >= 1 && key_bits <= 31 && (uint32_t)key_bits + pos_bits <= 32, "key_bits (<= 31) + pos_bits must fit 32 bits");
    GS_CHECK_ARG(n < (1ull << 32), "n must be < 2^32");
    const Sort32Layout L = sort32_layout(n);
    if (temp == nullptr || temp_bytes < L.total) {
        gs_set_error("gs_sort_isect_packed: temp too small (%zu < %zu)", temp_bytes, L.total);
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    char *tp = (char *)temp;
    uint32_t *tkeys = (uint32_t *)(tp + L.off_keys);
    uint32_t *hist = (uint32_t *)(tp + L.off_hist), *totals = (uint32_t *)(tp + L.off_totals);
    const int passes = (key_bits + RADIX_BITS - 1) / RADIX_BITS;
    const int first_bits = key_bits - (passes - 1) * RADIX_BITS;
    const IsectEpilogue ep = {depths, isect_ids, flatten_ids, perm, pos_bits, (const uint64_t *)sorted_keys};
    const bool small = sort_rounds_for(n) == SORT_ROUNDS_SMALL;
    const uint32_t *src_k = words;
    int shift = (int)pos_bits;
    for (int p = 0; p < passes; ++p) {
        DigitSpec d;
        const bool final = p == passes - 1;
        // (remainder bits FIRST, as for the pairs; remainder last -- longer store runs in the pass that writes 12 bytes per pair --
        // was measured again for the packed words: 0.7274 / 0.7475 / 0.7414 against 0.7276 / 0.7207 / 0.7207 ms/step: no)
        const int bits = (p == 0) ? first_bits : RADIX_BITS;
        d.shift = (uint32_t)shift;
        shift += bits;
        d.mask = (1u << bits) - 1u;
        d.drop = d.drop_hi = 0u;
        d.split = nullptr;
        d.flip = 0u; // (key_bits < 32 here: the id's sign bit is never set)
        uint32_t *dst_k = (p % 2 == 0) ? tkeys : words;
        if (small) launch_pass32_keys<SORT_ROUNDS_SMALL>(n, src_k, dst_k, d, L, hist, totals, final, ep, st);
        else launch_pass32_keys<SORT_ROUNDS_BIG>(n, src_k, dst_k, d, L, hist, totals, final, ep, st);
        src_k = dst_k;
    }
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_sort_isect_pairs(uint64_t n, uint32_t *keys32, int32_t *vals, const float *depths, int32_t key_bits,
                                       int64_t *isect_ids, int32_t *flatten_ids, void *temp, size_t temp_bytes, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(keys32 && vals && depths && isect_ids && flatten_ids, "null pointer");
    GS_CHECK_ARG(key_bits >= 1 && key_bits <= 32, "key_bits must be in [1, 32]");
    GS_CHECK_ARG(n < (1ull << 32), "n must be < 2^32");
    const Sort32Layout L = sort32_layout(n);
    if (temp == nullptr || temp_bytes < L.total) {
        gs_set_error("gs_sort_isect_pairs: temp too small (%zu < %zu)", temp_bytes, L.total);
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    char *tp = (char *)temp;
    uint32_t *tkeys = (uint32_t *)(tp + L.off_keys);
    int32_t *tvals = (int32_t *)(tp + L.off_vals);
    uint32_t *hist = (uint32_t *)(tp + L.off_hist), *totals = (uint32_t *)(tp + L.off_totals);
    const int passes = (key_bits + RADIX_BITS - 1) / RADIX_BITS;
    // the first pass takes the remainder (14 bits -> 6 + 8).  (Remainder LAST, for longer store runs in the pass that
    // writes 12 bytes per pair, measured 34.6 + 28.8 us against 36.0 + 25.0 us: no.)
    const int first_bits = key_bits - (passes - 1) * RADIX_BITS;
    const IsectEpilogue ep = {depths, isect_ids, flatten_ids, nullptr, 0u, nullptr};
    const bool small = sort_rounds_for(n) == SORT_ROUNDS_SMALL;
    const uint32_t *src_k = keys32;
    const int32_t *src_v = vals;
    int shift = 0;
    for (int p = 0; p < passes; ++p) { // ping-pong between the caller's pair (destroyed) and the temp pair
        DigitSpec d;
        const bool final = p == passes - 1;
        const int bits = (p == 0) ? first_bits : RADIX_BITS;
        d.shift = (uint32_t)shift;
        shift += bits;
        d.mask = (1u << bits) - 1u;
        d.drop = d.drop_hi = 0u;
        d.split = nullptr;
        // with all 32 key bits in use the id's bit 63 is set for the upper half of the cameras: the reference sorts int64
        // keys as SIGNED values (cub::DeviceRadixSort over [0, 64)), i.e. those come first
        d.flip = (key_bits == 32 && final) ? (1u << (bits - 1)) : 0u;
        uint32_t *dst_k = (p % 2 == 0) ? tkeys : keys32;
        int32_t *dst_v = (p % 2 == 0) ? tvals : vals;
        if (small) launch_pass32<SORT_ROUNDS_SMALL>(n, src_k, src_v, dst_k, dst_v, d, L, hist, totals, final, ep, st);
        else launch_pass32<SORT_ROUNDS_BIG>(n, src_k, src_v, dst_k, dst_v, d, L, hist, totals, final, ep, st);
        src_k = dst_k;
        src_v = dst_v;
    }
    GS_CHECK_LAUNCH();
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Bucketed depth pre-sort (round 4).  The splat-level depth pre-sort of the binning orders the ~0.3 M visible (depth, element)
// keys of a 1 M-splat frame.  As four 8-bit LSD passes that is 11 launches, each sitting at its ~5-10 us latency floor
// (76 us at BASELINE config 2, whatever the number of live keys).  Here it is ONE partition pass plus ONE launch of local sorts:
//   1. presort_sample_kernel + presort_split_kernel: 8192 elements sampled at a regular stride (32 workgroups fetch them);
//      the visible ones are histogrammed in LDS by one workgroup and 255 of them, at equal ranks, become SPLITTERS -- whatever the depth distribution, every bucket then holds ~n_kept / 256 keys (the gap
//      between splitters is a sum of ~15 sample gaps: more than 2.6x the mean happens about once in 10^6 buckets);
//   2. gs_isect_count_keys counts the keys of every 1024-element block per BUCKET (digit = number of splitters <= key);
//      sort_scan_kernel + sort_scatter_kernel<BUCKET> place them: a stable partition, the culled keys dropped;
//   3. presort_local_kernel: workgroup b (256 of them: one per CU) finishes bucket b with a stable LSD sort on the depth bits
//      that differ inside the bucket (typically 18: three 6-bit passes), entirely in LDS; it writes the permutation and the
//      emission's group sums.  A bucket that does not fit LDS (capacity 4096 keys against ~1150 expected: thousands of keys
//      within 2^-16 of the depth range, or adversarial input) is sorted by the same workgroup through global memory --
//      slower, same result.
// The order produced is exactly that of the stable radix sort: ascending depth bits, ties by element index.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t PS_CAP = 4096;                   // LDS capacity of a local sort (keys)

#ifdef PS_PROFILE
__device__ unsigned long long ps_stamps[64];
__device__ unsigned int ps_profile_block;
#define PS_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == ps_profile_block) ps_stamps[k] = wall_clock64(); } while (0)
#else
#define PS_STAMP(k) do { } while (0)
#endif

// The two kernels below are latency-bound workgroups (the splitter kernel is a single one): 16 waves each, four per SIMD, so
// that the serial chains of the ranking (LDS read -> ballots -> returning atomic -> shuffle) of different waves interleave.
constexpr int PS_WAVES = 16;
constexpr int PS_THREADS = PS_WAVES * GS_WAVE; // 1024

struct LdsSort {
    uint2 *a, *b;           // [capacity] each: (depth bits, element)
    uint16_t *rank;         // [capacity] rank of a key inside its (wave, digit) group
    uint32_t (*cnt)[RADIX]; // [PS_WAVES][RADIX]
    uint32_t *lbase;        // [RADIX]
    uint32_t *red;          // [2 * PS_WAVES + 8] reductions / scan scratch
};

GS_DEV LdsSort lds_sort_carve(unsigned char *lds, uint32_t capacity) {
    LdsSort L;
    L.a = reinterpret_cast<uint2 *>(lds);
    L.b = L.a + capacity;
    L.cnt = reinterpret_cast<uint32_t(*)[RADIX]>(L.b + capacity);
    L.lbase = reinterpret_cast<uint32_t *>(L.cnt + PS_WAVES);
    L.red = L.lbase + RADIX;
    L.rank = reinterpret_cast<uint16_t *>(L.red + 2 * PS_WAVES + 8);
    return L;
}
constexpr size_t lds_sort_bytes(uint32_t capacity) {
    return (size_t)capacity * (2 * sizeof(uint2) + 2) + PS_WAVES * RADIX * 4 + RADIX * 4 + (2 * PS_WAVES + 8) * 4 + 64;
}

// exclusive scan of one value per thread over the FIRST 256 threads of the workgroup (the digits); every thread calls it
GS_DEV uint32_t ps_scan256(uint32_t t, uint32_t *s_scan, uint32_t *total) {
    const uint32_t lane = threadIdx.x % GS_WAVE, wave = threadIdx.x / GS_WAVE;
    uint32_t inc = wave < 4u ? t : 0u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    __syncthreads();
    if (lane == GS_WAVE - 1 && wave < 4u) s_scan[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if ((uint32_t)w < wave) wbase += s_scan[w];
        tot += s_scan[w];
    }
    if (total != nullptr) *total = tot;
    return wbase + inc - t;
}

// Stable sort of the m pairs in L.a by their .x, in LDS; returns the buffer holding the result (L.a or L.b).
// Only the bits in which the keys differ are sorted (range taken over the m keys), in passes of equal width <= 8 bits.
// Same ranking as sort_scatter_kernel: wave w owns a contiguous share of the keys and walks it in rounds of 64, so
// (wave, round, lane) order is index order and the ballot-matched rank inside (wave, digit) is stable.  The ranks of a pass
// are parked in LDS (2 bytes per key) instead of registers: the round loops stay rolled and the routine serves any capacity.
// max_bits < 32: only the TOP max_bits of the differing bits are sorted (keys equal in them keep their order); *low_mask gets
// the mask of the ignored low bits.
GS_DEV uint2 *lds_stable_sort(const LdsSort &L, uint32_t m, uint32_t max_bits = 32, uint32_t *low_mask = nullptr) {
    const uint32_t tid = threadIdx.x, lane = tid % GS_WAVE, wave = tid / GS_WAVE;
    const uint32_t rounds = (m + PS_THREADS - 1) / PS_THREADS; // (uniform)
    const uint32_t wave_base = wave * rounds * GS_WAVE;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // key range
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    for (uint32_t j = tid; j < m; j += PS_THREADS) {
        const uint32_t k = L.a[j].x;
        kmin = min(kmin, k);
        kmax = max(kmax, k);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, off, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, off, 64));
    }
    __syncthreads();
    if (lane == 0) {
        L.red[wave] = kmin;
        L.red[PS_WAVES + wave] = kmax;
    }
    __syncthreads();
    kmin = 0xffffffffu;
    kmax = 0u;
#pragma unroll
    for (int w = 0; w < PS_WAVES; ++w) {
        kmin = min(kmin, L.red[w]);
        kmax = max(kmax, L.red[PS_WAVES + w]);
    }
    const uint32_t span = kmax >= kmin ? kmax - kmin : 0u;
    const uint32_t nb_all = span == 0u ? 0u : 32u - (uint32_t)__builtin_clz(span);
    const uint32_t drop = nb_all > max_bits ? nb_all - max_bits : 0u;
    const uint32_t nb = nb_all - drop;
    if (low_mask != nullptr) *low_mask = (1u << drop) - 1u;
    const uint32_t passes = (nb + RADIX_BITS - 1) / RADIX_BITS;
    const uint32_t wbits = passes ? (nb + passes - 1) / passes : 0u;
    const uint32_t mask = (1u << wbits) - 1u;
    uint2 *src = L.a, *dst = L.b;
    PS_STAMP(16);
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t shift = drop + p * wbits;
        PS_STAMP(17 + 4 * p);
        // (five workgroup barriers per pass: with 16 waves each costs a few hundred cycles)
        for (uint32_t c = tid; c < PS_WAVES * RADIX; c += PS_THREADS) (&L.cnt[0][0])[c] = 0; // (the previous pass ended on a barrier)
        __syncthreads();
#pragma unroll 1
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t j = wave_base + r * GS_WAVE + lane;
            const bool valid = j < m;
            const uint32_t dg = valid ? ((src[j].x - kmin) >> shift) & mask : 0u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < RADIX_BITS; ++b) {
                if ((uint32_t)b < wbits) { // (uniform)
                    const bool bit = (dg >> b) & 1u;
                    const unsigned long long mm = __ballot(bit);
                    peers &= bit ? mm : ~mm;
                }
            }
            const uint32_t before = __popcll(peers & lt_mask);
            const uint32_t leader = valid ? (uint32_t)__builtin_ctzll(peers) : lane;
            uint32_t base = 0;
            if (valid && before == 0) base = atomicAdd(&L.cnt[wave][dg], (uint32_t)__popcll(peers));
            base = __shfl(base, (int)leader, 64);
            if (valid) L.rank[j] = (uint16_t)(base + before);
        }
        PS_STAMP(18 + 4 * p);
        __syncthreads();
        uint32_t run = 0, inc = 0;
        if (tid < RADIX) { // digit tid: prefix over the waves (all 16 counts requested at once), then a wave-level scan of the totals
            uint32_t c[PS_WAVES];
#pragma unroll
            for (int w = 0; w < PS_WAVES; ++w) c[w] = L.cnt[w][tid];
#pragma unroll
            for (int w = 0; w < PS_WAVES; ++w) {
                L.cnt[w][tid] = run;
                run += c[w];
            }
            inc = run;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, 64);
                if (lane >= (uint32_t)off) inc += o;
            }
            if (lane == GS_WAVE - 1) L.red[wave] = inc;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t wbase = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if ((uint32_t)w < wave) wbase += L.red[w];
            L.lbase[tid] = wbase + inc - run;
        }
        __syncthreads();
        PS_STAMP(19 + 4 * p);
#pragma unroll 2
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t j = wave_base + r * GS_WAVE + lane;
            if (j < m) {
                const uint2 kv = src[j];
                const uint32_t dg = ((kv.x - kmin) >> shift) & mask;
                dst[L.lbase[dg] + L.cnt[wave][dg] + (uint32_t)L.rank[j]] = kv;
            }
        }
        __syncthreads();
        uint2 *t = src;
        src = dst;
        dst = t;
    }
    return src;
}

// 1. splitters.  radii / depths: the projection's dense outputs; split: 255 ascending keys (depth bits << 32) padded with UINT64_MAX in
// [0, 256), behind it the candidate slots (gs_presort_split_elems() int64 in all).
// Candidates: 8192 elements at a regular stride; the visible ones are the samples (every visible element has the same chance: the
// balance of the buckets does not depend on how visibility is distributed over the array; ~9 samples per bucket at 29 %
// visibility).  Until round 6 the candidates were 512 runs of 16 CONSECUTIVE elements fetched by the splitter workgroup itself
// (1 K cache lines): fine for an arbitrary splat order, but in a spatially sorted array (Morton / PLAS order, what a codec leaves
// behind) a run's 16 depths are nearly equal and visibility comes in runs too -- ~120 effective samples for 255 splitters, buckets
// overflowing LDS: the pre-sort of 2 M Morton-ordered splats took 175 us instead of 55.  8 K individual elements are 16 K
// distinct cache lines, which ONE workgroup needs 34-42 us to fetch (its CU's miss rate) and which several workgroups cannot
// hand to one of them inside a launch for less than ~10 us (agent-scope fence + ticket across 8 XCDs: measured 19-20 us for the
// kernel): so presort_sample_kernel (32 workgroups x 256 candidates, ~2 us) writes candidate slot c = (depth bits | invalid,
// element) and the splitter workgroup of the NEXT launch reads the 64 KB of slots as one coalesced stream.
constexpr uint32_t PS_CAND = 8192, PS_SAMPLE_BLOCKS = 32, PS_SPLIT_CAP = 8192, PS_ROUNDS = PS_CAND / PS_THREADS;
constexpr uint32_t PS_SPLIT_ELEMS = GS_PRESORT_BUCKETS + PS_CAND; // int64: table | candidate slots
constexpr uint32_t PS_INVALID = 0xffffffffu;

__global__ void __launch_bounds__(GS_BLOCK) presort_sample_kernel(uint32_t n_elems, const int32_t *__restrict__ radii,
                                                                  const float *__restrict__ depths, uint64_t *__restrict__ split) {
    const uint32_t c = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (c >= PS_CAND) return;
    const uint32_t i = (uint32_t)(((uint64_t)c * n_elems) / PS_CAND);
    const bool fresh = c == 0u || i != (uint32_t)(((uint64_t)(c - 1u) * n_elems) / PS_CAND); // (fewer elements than candidates)
    const uint32_t ic = i < n_elems ? i : 0u;
    const int32_t r = radii[ic];
    const float d = depths[ic]; // (both loads in flight together; undefined where culled: never used there)
    const bool k = fresh && i < n_elems && r > 0;
    reinterpret_cast<uint2 *>(split + GS_PRESORT_BUCKETS)[c] = make_uint2(k ? ((uint32_t)__float_as_int(d) & 0x7fffffffu) : PS_INVALID, i);
}

__global__ void __launch_bounds__(PS_THREADS) presort_split_kernel(uint64_t *__restrict__ split) {
    extern __shared__ __align__(16) unsigned char ps_lds[];
    struct {
        uint2 *a;      // [PS_SPLIT_CAP] samples (depth bits, element)
        uint2 *b;      // histogram cells (4096 x u32) live here
        uint32_t *red; // [2 * PS_WAVES + 8]
    } L;
    L.a = reinterpret_cast<uint2 *>(ps_lds);
    L.b = L.a + PS_SPLIT_CAP;
    L.red = reinterpret_cast<uint32_t *>(ps_lds + PS_SPLIT_CAP * sizeof(uint2) + 4096 * sizeof(uint32_t));
    const uint32_t tid = threadIdx.x, lane = tid % GS_WAVE, wave = tid / GS_WAVE;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint2 *g_samp = reinterpret_cast<const uint2 *>(split + GS_PRESORT_BUCKETS);
    PS_STAMP(0);
    uint2 cand[PS_ROUNDS];
#pragma unroll
    for (uint32_t r = 0; r < PS_ROUNDS; ++r) cand[r] = g_samp[r * PS_THREADS + tid]; // (64 KB, coalesced)
    PS_STAMP(1);
    // the valid candidates become the samples, in any order (a histogram follows): per round one ballot and one LDS atomic of the
    // wave's first valid lane reserve the slots
    uint32_t *s_count = L.red + 2 * PS_WAVES;
    if (tid == 0) *s_count = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < PS_ROUNDS; ++r) {
        const bool k = cand[r].x != PS_INVALID;
        const unsigned long long bl = __ballot(k);
        if (bl != 0ull) { // (wave-uniform)
            const uint32_t first = (uint32_t)__builtin_ctzll(bl);
            uint32_t base = 0;
            if (lane == first) base = atomicAdd(s_count, (uint32_t)__popcll(bl));
            base = __shfl(base, (int)first, 64);
            const uint32_t slot = base + (uint32_t)__popcll(bl & lt_mask);
            if (k && slot < PS_SPLIT_CAP) L.a[slot] = cand[r];
        }
    }
    __syncthreads();
    const uint32_t m = min(*s_count, PS_SPLIT_CAP);
    __syncthreads();
    PS_STAMP(2);
    // Splitters = equal-frequency quantiles of the samples, read off a 4096-cell histogram over the samples' depth range (no
    // sort: this is ONE workgroup, every sorting pass is serial latency -- 9.6 us for 2.4 K samples against ~2 us here).
    // A splitter is the lower edge of the cell holding the sample of rank (j + 1) m / 256, element 0: the table is ascending,
    // which is all the bucket function needs (splitters that coincide just leave buckets empty; it takes thousands of keys
    // within 2^-12 of the depth range to overfill one bucket, and then the local sort's global-memory route takes it).
    constexpr uint32_t CELLS = 4096, CPT = CELLS / PS_THREADS; // 4 cells per thread
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(L.b);      // [CELLS] counts, then inclusive prefix
    uint32_t smin = 0xffffffffu, smax = 0u;
    for (uint32_t j = tid; j < m; j += PS_THREADS) {
        const uint32_t k = L.a[j].x;
        smin = min(smin, k);
        smax = max(smax, k);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        smin = min(smin, (uint32_t)__shfl_xor(smin, off, 64));
        smax = max(smax, (uint32_t)__shfl_xor(smax, off, 64));
    }
    if (lane == 0) {
        L.red[wave] = smin;
        L.red[PS_WAVES + wave] = smax;
    }
#pragma unroll
    for (uint32_t i = 0; i < CPT; ++i) s_hist[tid * CPT + i] = 0u;
    __syncthreads();
    smin = 0xffffffffu;
    smax = 0u;
#pragma unroll
    for (int w = 0; w < PS_WAVES; ++w) {
        smin = min(smin, L.red[w]);
        smax = max(smax, L.red[PS_WAVES + w]);
    }
    const uint32_t span = smax >= smin ? smax - smin : 0u;
    const uint32_t nb = span == 0u ? 0u : 32u - (uint32_t)__builtin_clz(span);
    const uint32_t shift = nb > 12u ? nb - 12u : 0u;
    for (uint32_t j = tid; j < m; j += PS_THREADS) atomicAdd(&s_hist[(L.a[j].x - smin) >> shift], 1u);
    __syncthreads();
    // inclusive prefix over the cells: 4 consecutive cells per thread, wave scans of the thread totals, 16 wave totals
    uint32_t c4[CPT], tsum = 0;
#pragma unroll
    for (uint32_t i = 0; i < CPT; ++i) {
        tsum += s_hist[tid * CPT + i];
        c4[i] = tsum;
    }
    uint32_t inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    __syncthreads();
    if (lane == GS_WAVE - 1) L.red[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (int w = 0; w < PS_WAVES; ++w)
        if ((uint32_t)w < wave) wbase += L.red[w];
    const uint32_t excl = wbase + inc - tsum;
#pragma unroll
    for (uint32_t i = 0; i < CPT; ++i) s_hist[tid * CPT + i] = excl + c4[i];
    __syncthreads();
    PS_STAMP(3);
    if (tid < GS_PRESORT_BUCKETS) {
        uint64_t out = ~0ull;
        if (tid < GS_PRESORT_BUCKETS - 1 && m > 0) {
            const uint32_t rank = (uint32_t)(((uint64_t)(tid + 1) * m) / GS_PRESORT_BUCKETS); // sample rank of this splitter (< m)
            uint32_t lo = 0, hi = CELLS - 1; // smallest cell whose inclusive prefix exceeds the rank
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_hist[mid] > rank) hi = mid;
                else lo = mid + 1;
            }
            out = (uint64_t)(smin + (lo << shift)) << 32;
        }
        split[tid] = out;
    }
    PS_STAMP(4);
}

// segmented wave sums of `inc` over runs of equal `grp` among consecutive lanes, one atomic per run (as in the scatter's side job)
GS_DEV void ps_side_add(uint32_t *side_sums, bool on, uint32_t grp, uint32_t inc) {
    const uint32_t lane = threadIdx.x % GS_WAVE;
    if (!on) { grp = 0xffffffffu; inc = 0u; }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    const uint32_t prevg = __shfl_up(grp, 1, 64);
    const bool head = lane == 0u || grp != prevg;
    const unsigned long long hm = __ballot(head);
    const unsigned long long above = lane == 63u ? 0ull : (hm >> (lane + 1u));
    const uint32_t last = above ? lane + (uint32_t)__builtin_ctzll(above) : 63u;
    const uint32_t upto = __shfl(inc, (int)last, 64);
    const uint32_t before = __shfl_up(inc, 1, 64);
    if (head && on) atomicAdd(&side_sums[grp], upto - (lane == 0u ? 0u : before));
}

// 3. local sorts.  keys [*n_kept]: the partitioned keys (bucket order, stable); totals [256]: keys per bucket; alt [n]: scratch
// for a range that does not fit LDS.  perm [*n_kept] out; side_sums[p >> side_shift] += side_vals[perm[p]].
__global__ void __launch_bounds__(PS_THREADS) presort_local_kernel(const uint32_t *__restrict__ n_kept_p, const uint32_t *__restrict__ totals,
                                                                   uint64_t *__restrict__ keys, uint64_t *__restrict__ alt,
                                                                   int32_t *__restrict__ perm, uint64_t *__restrict__ sorted_out,
                                                                   const int32_t *__restrict__ side_vals,
                                                                   uint32_t *__restrict__ side_sums, uint32_t side_shift, uint32_t cap) {
    extern __shared__ __align__(16) unsigned char ps_lds[];
    const LdsSort L = lds_sort_carve(ps_lds, PS_CAP);
    uint32_t *s_start = reinterpret_cast<uint32_t *>(ps_lds + lds_sort_bytes(PS_CAP)); // [257]
    const uint32_t tid = threadIdx.x, lane = tid % GS_WAVE, wave = tid / GS_WAVE;
    PS_STAMP(8);
    // workgroup b finishes bucket b: 256 buckets, one workgroup per CU, all of them resident at once
    {
        const uint32_t tot = tid < GS_PRESORT_BUCKETS ? totals[tid] : 0u;
        const uint32_t st = ps_scan256(tot, L.red, nullptr);
        if (tid < GS_PRESORT_BUCKETS) {
            s_start[tid] = st;
            if (tid == GS_PRESORT_BUCKETS - 1) s_start[GS_PRESORT_BUCKETS] = st + tot;
        }
    }
    __syncthreads();
    const uint32_t lo = s_start[blockIdx.x], hi = s_start[blockIdx.x + 1];
    const uint32_t m = hi - lo;
    PS_STAMP(9);
    if (m == 0) return;
    if (m <= cap) {
        for (uint32_t j = tid; j < m; j += PS_THREADS) {
            const uint64_t k = keys[lo + j];
            L.a[j] = make_uint2((uint32_t)(k >> 32), (uint32_t)k);
        }
        __syncthreads();
        PS_STAMP(10);
        const uint2 *sorted = lds_stable_sort(L, m);
        PS_STAMP(11);
        for (uint32_t j0 = 0; j0 < m; j0 += PS_THREADS) { // (whole waves take part in the segmented sums)
            const uint32_t j = j0 + tid;
            const bool on = j < m;
            const uint32_t e = on ? sorted[j].y : 0u;
            if (on) perm[lo + j] = (int32_t)e;
            if (on && sorted_out != nullptr) sorted_out[lo + j] = ((uint64_t)sorted[j].x << 32) | (uint64_t)e;
            if (side_sums != nullptr) ps_side_add(side_sums, on, (lo + j) >> side_shift, on ? (uint32_t)side_vals[e] : 0u);
        }
        PS_STAMP(12);
        return;
    }
    // The range does not fit LDS: the same stable LSD sort, by this workgroup alone, through global memory -- per pass a digit
    // histogram over the range, then 1024-key tiles in order (one key per thread), each ranked like a scatter block and placed
    // behind its predecessors.  keys <-> alt ping-pong inside [lo, hi) (nobody else touches that range).
    uint32_t *s_hist = L.lbase;
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    for (uint32_t j = tid; j < m; j += PS_THREADS) {
        const uint32_t k = (uint32_t)(keys[lo + j] >> 32);
        kmin = min(kmin, k);
        kmax = max(kmax, k);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, off, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, off, 64));
    }
    __syncthreads();
    if (lane == 0) {
        L.red[wave] = kmin;
        L.red[PS_WAVES + wave] = kmax;
    }
    __syncthreads();
    kmin = 0xffffffffu;
    kmax = 0u;
#pragma unroll
    for (int w = 0; w < PS_WAVES; ++w) {
        kmin = min(kmin, L.red[w]);
        kmax = max(kmax, L.red[PS_WAVES + w]);
    }
    __syncthreads();
    const uint32_t span = kmax - kmin;
    const uint32_t nb = span == 0u ? 0u : 32u - (uint32_t)__builtin_clz(span);
    const uint32_t passes = (nb + RADIX_BITS - 1) / RADIX_BITS;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint64_t *src = keys, *dst = alt;
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t shift = p * RADIX_BITS;
        if (tid < RADIX) s_hist[tid] = 0;
        __syncthreads();
        for (uint32_t j = tid; j < m; j += PS_THREADS) atomicAdd(&s_hist[(((uint32_t)(src[lo + j] >> 32) - kmin) >> shift) & 0xffu], 1u);
        __syncthreads();
        const uint32_t mine = tid < RADIX ? s_hist[tid] : 0u;
        const uint32_t base0 = ps_scan256(mine, L.red, nullptr);
        __syncthreads();
        if (tid < RADIX) s_hist[tid] = base0; // running base of digit tid
        __syncthreads();
        for (uint32_t t0 = 0; t0 < m; t0 += PS_THREADS) {
            for (uint32_t c = tid; c < PS_WAVES * RADIX; c += PS_THREADS) (&L.cnt[0][0])[c] = 0;
            __syncthreads();
            const uint32_t j = t0 + tid; // (wave, lane) order is index order inside the tile
            const bool valid = j < m;
            const uint64_t kk = valid ? src[lo + j] : 0ull;
            const uint32_t dg = (((uint32_t)(kk >> 32) - kmin) >> shift) & 0xffu;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < RADIX_BITS; ++b) {
                const bool bit = (dg >> b) & 1u;
                const unsigned long long mm = __ballot(bit);
                peers &= bit ? mm : ~mm;
            }
            const uint32_t before = __popcll(peers & lt_mask);
            const uint32_t leader = valid ? (uint32_t)__builtin_ctzll(peers) : lane;
            uint32_t base = 0;
            if (valid && before == 0) base = atomicAdd(&L.cnt[wave][dg], (uint32_t)__popcll(peers));
            const uint32_t rank = __shfl(base, (int)leader, 64) + before;
            __syncthreads();
            if (tid < RADIX) {
                uint32_t run = s_hist[tid];
                uint32_t c[PS_WAVES];
#pragma unroll
                for (int w = 0; w < PS_WAVES; ++w) c[w] = L.cnt[w][tid];
#pragma unroll
                for (int w = 0; w < PS_WAVES; ++w) {
                    L.cnt[w][tid] = run;
                    run += c[w];
                }
                s_hist[tid] = run;
            }
            __syncthreads();
            if (valid) dst[lo + L.cnt[wave][dg] + rank] = kk;
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        uint64_t *t = src;
        src = dst;
        dst = t;
    }
    for (uint32_t j0 = 0; j0 < m; j0 += PS_THREADS) {
        const uint32_t j = j0 + tid;
        const bool on = j < m;
        const uint64_t sk = on ? src[lo + j] : 0ull;
        const uint32_t e = (uint32_t)sk;
        if (on) perm[lo + j] = (int32_t)e;
        if (on && sorted_out != nullptr) sorted_out[lo + j] = sk;
        if (side_sums != nullptr) ps_side_add(side_sums, on, (lo + j) >> side_shift, on ? (uint32_t)side_vals[e] : 0u);
    }
}

constexpr size_t PS_LOCAL_LDS = lds_sort_bytes(PS_CAP) + (GS_PRESORT_BUCKETS + 1) * 4 + 60;
constexpr size_t PS_SPLIT_LDS = PS_SPLIT_CAP * sizeof(uint2) + 4096 * sizeof(uint32_t) + (2 * PS_WAVES + 8) * sizeof(uint32_t) + 64;

} // namespace

// the bucketed pre-sort serves the sizes the 1024-key sort blocks serve (its partition pass is one of them); above that the
// passes of the plain radix sort are bandwidth-bound and there is nothing to gain
extern "C" int32_t gs_presort_applicable(uint64_t n) { return (n > 0 && sort_rounds_for(n) == SORT_ROUNDS_SMALL) ? 1 : 0; }
extern "C" uint32_t gs_presort_capacity(void) { return PS_CAP; }

extern "C" uint32_t gs_presort_split_elems(void) { return PS_SPLIT_ELEMS; }

extern "C" int32_t gs_presort_split(uint32_t n_elems, const int32_t *radii, const float *depths, int64_t *splitters, gs_stream_t stream) {
    GS_CHECK_ARG(radii && depths && splitters, "null pointer");
    GS_CHECK_ARG(n_elems > 0, "n_elems must be > 0");
    GS_CHECK_ARG((uintptr_t)splitters % 8 == 0, "splitters must be 8-byte aligned");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(presort_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)PS_SPLIT_LDS);
    GS_CHECK_ARG(e == hipSuccess, "cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(presort_sample_kernel, dim3(PS_SAMPLE_BLOCKS), dim3(GS_BLOCK), 0, (hipStream_t)stream, n_elems, radii, depths,
                       (uint64_t *)splitters);
    hipLaunchKernelGGL(presort_split_kernel, dim3(1), dim3(PS_THREADS), PS_SPLIT_LDS, (hipStream_t)stream, (uint64_t *)splitters);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_presort_buckets(uint64_t n, int64_t *keys_in, const int32_t *vals_in, const int64_t *splitters,
                                      int32_t *perm, uint32_t *n_kept, void *temp, size_t temp_bytes, const int32_t *side_vals,
                                      uint32_t *side_sums, uint32_t side_shift, uint32_t lds_capacity, gs_stream_t stream) {
    if (n == 0) return 0;
    (void)vals_in; // (the element index is the low half of the key: nothing rides along with the keys)
    GS_CHECK_ARG(keys_in && splitters && perm && n_kept, "null pointer");
    GS_CHECK_ARG(gs_presort_applicable(n), "gs_presort_applicable(n) is 0: use gs_sort_pairs_u64_i32_drop");
    GS_CHECK_ARG((side_vals == nullptr) == (side_sums == nullptr) && side_shift < 32, "side_vals and side_sums go together");
    GS_CHECK_ARG(lds_capacity <= PS_CAP, "lds_capacity exceeds gs_presort_capacity()");
    const SortLayout L = sort_layout(n);
    // temp: the radix sort's layout (keys | vals | hist | totals: the histogram is where gs_isect_count_keys put it) + n more keys
    const size_t alt_off = (L.total + 255) & ~(size_t)255;
    GS_CHECK_ARG(temp != nullptr && temp_bytes >= alt_off + n * sizeof(uint64_t), "temp too small (gs_presort_temp_bytes)");
    hipStream_t st = (hipStream_t)stream;
    char *tp = (char *)temp;
    uint64_t *tkeys = (uint64_t *)(tp + L.off_keys);
    uint32_t *hist = (uint32_t *)(tp + L.off_hist), *totals = (uint32_t *)(tp + L.off_totals);
    uint64_t *alt = (uint64_t *)(tp + alt_off);
    const uint32_t n_side = side_sums != nullptr ? (uint32_t)((n + (1ull << side_shift) - 1) >> side_shift) : 0u;
    DigitSpec d;
    d.shift = 32; d.mask = 0xffu; d.flip = 0u; d.drop = 1u; d.drop_hi = 0x7fffffffu;
    d.split = (const uint64_t *)splitters;
    hipLaunchKernelGGL(sort_scan_kernel, dim3(RADIX), dim3(GS_BLOCK), 0, st, L.n_blocks, hist, totals, side_sums, n_side);
    hipLaunchKernelGGL((sort_scatter_kernel<uint64_t, SORT_ROUNDS_SMALL, false, true, true>), dim3(L.n_blocks), dim3(GS_BLOCK), 0, st, n,
                       (const uint32_t *)nullptr, (const uint64_t *)keys_in, (const int32_t *)nullptr, tkeys, (int32_t *)nullptr, d, L.n_blocks, hist,
                       totals, n_kept, IsectEpilogue{}, ScatterSide{nullptr, nullptr, 0u});
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(presort_local_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)PS_LOCAL_LDS);
    GS_CHECK_ARG(e == hipSuccess, "cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(presort_local_kernel, dim3(GS_PRESORT_BUCKETS), dim3(PS_THREADS), PS_LOCAL_LDS, st, n_kept, totals, tkeys, alt, perm,
                       (uint64_t *)keys_in /* dead after the partition pass: receives the sorted keys */, side_vals, side_sums, side_shift, lds_capacity ? lds_capacity : PS_CAP);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t gs_presort_temp_bytes(uint64_t n) { return ((sort_layout(n).total + 255) & ~(size_t)255) + n * sizeof(uint64_t); }
