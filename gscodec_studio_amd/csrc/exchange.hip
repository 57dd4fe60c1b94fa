// exchange.hip -- row packing around the multi-GPU exchange of projected splats (gfx950).
//
// The gaussian-sharded mode (reference gsplat/rendering.py:397-478) moves, per (camera, gaussian) pair, radii |
// means2d | depths | conics | opacities | colours between ranks.  The reference concatenates them with torch.cat
// before each all-to-all and splits them afterwards; here ONE streaming kernel gathers the column blocks of up to
// eight row-major arrays (each with its own row stride, so column views of wider buffers -- e.g. the packed 16-float
// gradient rows of gs_rasterize_bwd -- are read in place) into the wire rows, and one scatters wire rows back into
// separate arrays, 256 rows per workgroup staged through LDS so that both sides move as contiguous chunks.
// HBM-bound: 8 B per element.
#include "gs_common.h"

#include <algorithm>

namespace {

constexpr int MAX_PARTS = 8;

constexpr int ROWS_PER_BLOCK = 256;   // rows per workgroup for wire rows of up to 64 elements ...
constexpr int TILE_ELEMS = 15360;     // ... wider rows get fewer rows per workgroup: the LDS tile stays at 60 KB (+ 1 KB of indices)
constexpr int MAX_WIDTH = TILE_ELEMS; // (the reference's distributed path takes any channel count; 513 + 10 columns is the widest real row)

struct RowParts {
    uint32_t *ptr[MAX_PARTS];
    int64_t stride[MAX_PARTS];  // row stride in 4-byte elements
    int32_t begin[MAX_PARTS + 1];  // first wire column of part k; begin[n] = wire width
    uint64_t inv[MAX_PARTS];  // floor(2^32 / width) + 1: j / width == (j * inv) >> 32 for the j < 2^14 used here
    int32_t n;
    uint32_t indexed;  // bit k: part k's row is row_index[wire row], not the wire row itself
};

// One workgroup moves 256 rows through LDS: the wire side is one contiguous chunk (coalesced), and each part is
// walked in its own element order, so a part whose rows are dense in memory is one contiguous chunk too.
template <bool PACK>
__global__ void __launch_bounds__(GS_BLOCK) rows_kernel(uint64_t n_rows, uint32_t width, uint32_t rpb, RowParts t, uint32_t *__restrict__ wire,
                                                        const int32_t *__restrict__ row_index, int64_t index_stride) {
    extern __shared__ uint32_t tile[];  // [rows][width]
    __shared__ int32_t s_index[ROWS_PER_BLOCK];
    const uint64_t row0 = (uint64_t)blockIdx.x * rpb;
    const uint32_t nr = (uint32_t)min((uint64_t)rpb, n_rows - row0);
    uint32_t *w0 = wire + row0 * width;
    if (t.indexed != 0u) {
        if (threadIdx.x < nr) s_index[threadIdx.x] = row_index[(int64_t)(row0 + threadIdx.x) * index_stride];
        __syncthreads();
    }
    if (!PACK) {
        for (uint32_t j = threadIdx.x; j < nr * width; j += GS_BLOCK) tile[j] = w0[j];
        __syncthreads();
    }
    for (int k = 0; k < t.n; ++k) {  // uniform
        uint32_t *p = t.ptr[k];
        if (!PACK && p == nullptr) continue;
        const uint32_t b = (uint32_t)t.begin[k], w = (uint32_t)t.begin[k + 1] - b;
        const uint64_t inv = t.inv[k];
        const int64_t s = t.stride[k];
        const bool idx = (t.indexed >> k) & 1u;  // uniform
        if (!idx && p != nullptr) p += row0 * s;
        for (uint32_t j = threadIdx.x; j < nr * w; j += GS_BLOCK) {
            const uint32_t r = (uint32_t)(((uint64_t)j * inv) >> 32), c = j - r * w;
            const int64_t row = idx ? (int64_t)s_index[r] : (int64_t)r;  // a negative index = no row: zeros in, skipped out
            if (PACK)
                tile[r * width + b + c] = (p != nullptr && row >= 0) ? p[row * s + c] : 0u;
            else if (row >= 0)
                p[row * s + c] = tile[r * width + b + c];
        }
    }
    if (PACK) {
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < nr * width; j += GS_BLOCK) w0[j] = tile[j];
    }
}

template <bool PACK>
int32_t rows_launch(uint64_t n_rows, int32_t n_parts, void *const *parts, const int32_t *widths, const int64_t *strides, void *wire,
                    gs_stream_t stream, const int32_t *indexed = nullptr, const int32_t *row_index = nullptr, int64_t index_stride = 1) {
    GS_CHECK_ARG(n_parts >= 1 && n_parts <= MAX_PARTS, "between 1 and 8 parts");
    GS_CHECK_ARG(parts && widths && strides, "null table");
    RowParts t = {};
    t.n = n_parts;
    int32_t w = 0;
    for (int k = 0; k < n_parts; ++k) {
        GS_CHECK_ARG(widths[k] >= 1 && strides[k] >= widths[k], "part width / row stride");
        t.ptr[k] = (uint32_t *)parts[k];
        t.stride[k] = strides[k];
        t.begin[k] = w;
        t.inv[k] = (1ull << 32) / (uint64_t)widths[k] + 1ull;
        if (indexed != nullptr && indexed[k] != 0) t.indexed |= 1u << k;
        w += widths[k];
    }
    GS_CHECK_ARG(t.indexed == 0u || (row_index != nullptr && index_stride >= 1), "indexed parts need row_index");
    for (int k = n_parts; k <= MAX_PARTS; ++k) t.begin[k] = w;
    if (n_rows == 0) return 0;
    GS_CHECK_ARG(wire != nullptr, "null pointer");
    GS_CHECK_ARG(w <= MAX_WIDTH, "wire rows of at most 15360 elements");
    // rows per workgroup: 256 up to 64 columns, fewer for wider rows (rpb * w < 2^14 keeps the LDS tile under 64 KB and the
    // multiply-shift division of the kernel exact)
    const uint32_t rpb = (uint32_t)std::max(1, std::min(ROWS_PER_BLOCK, TILE_ELEMS / w));
    const uint64_t blocks = (n_rows + rpb - 1) / rpb;
    GS_CHECK_ARG(blocks < (1ull << 31), "too many rows");
    hipLaunchKernelGGL(rows_kernel<PACK>, dim3((uint32_t)blocks), dim3(GS_BLOCK), (size_t)rpb * w * sizeof(uint32_t),
                       (hipStream_t)stream, n_rows, (uint32_t)w, rpb, t, (uint32_t *)wire, row_index, index_stride);
    GS_CHECK_LAUNCH();
    return 0;
}

// Compaction of the visible (camera, gaussian) rows for the sparse exchange.  Destination rank d gets a chunk of `cap` row
// slots plus one header row; slot order inside a chunk is arbitrary (the receiver scatters by the destination index).
//   src_index[d * (cap + 1) + p]     = c * N + n                       (pre-filled with -1)
//   hdr      [d * (cap + 1) + p]     = (destination row (c % C_local) * N_total + N_off + n, 0)   (pre-filled with -1)
//   counters [d]                     = rows wanted by d (may exceed cap: then rows were dropped -> overflow)
constexpr int COMPACT_BLOCK = 1024;

__global__ void __launch_bounds__(COMPACT_BLOCK) exchange_compact_kernel(uint32_t N, uint32_t C_local, uint32_t cap, uint32_t N_total,
                                                                         uint32_t N_off, const int32_t *__restrict__ radii,
                                                                         int32_t *__restrict__ src_index, int2 *__restrict__ hdr,
                                                                         uint32_t *__restrict__ counters) {
    // one atomic per 1024 elements (all chunks of one destination share ONE counter: per-wave atomics on it took 95 us)
    __shared__ uint32_t s_cnt[COMPACT_BLOCK / GS_WAVE];
    __shared__ uint32_t s_base;
    const uint32_t c = blockIdx.y, n = blockIdx.x * COMPACT_BLOCK + threadIdx.x, d = c / C_local;  // d is block-uniform
    const bool vis = n < N && radii[(size_t)c * N + n] > 0;
    const unsigned long long m = __ballot(vis);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < COMPACT_BLOCK / GS_WAVE; ++w) {
            const uint32_t k = s_cnt[w];
            s_cnt[w] = tot;
            tot += k;
        }
        s_base = tot ? atomicAdd(&counters[d], tot) : 0u;
    }
    __syncthreads();
    const uint32_t p = s_base + s_cnt[wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (vis && p < cap) {
        const size_t o = (size_t)d * (cap + 1) + p;
        src_index[o] = (int32_t)(c * N + n);
        hdr[o] = make_int2((int32_t)((c % C_local) * N_total + N_off + n), 0);
    }
}

// header row of every chunk: (-1, count | overflow << 30); stats[0] = max count, stats[1] = 1 if any chunk overflowed
__global__ void exchange_header_kernel(uint32_t world, uint32_t cap, const uint32_t *__restrict__ counters, int2 *__restrict__ hdr,
                                       uint32_t *__restrict__ stats) {
    // the flag in EVERY header says "some chunk of this rank overflowed": a receiver only sees the chunks addressed to it,
    // and all ranks must take the same decision about repeating the exchange
    __shared__ uint32_t s_over;
    const uint32_t d = threadIdx.x;
    if (d == 0) s_over = 0u;
    __syncthreads();
    const uint32_t cnt = d < world ? counters[d] : 0u;
    if (cnt > cap) atomicOr(&s_over, 1u);
    __syncthreads();
    if (d >= world) return;
    const uint32_t over = s_over;
    hdr[(size_t)d * (cap + 1) + cap] = make_int2(-1, (int32_t)(min(cnt, cap) | (over << 30)));
    atomicMax(&stats[0], cnt);
    if (over) atomicMax(&stats[1], 1u);
}

// pre-fill of the compaction's outputs in ONE launch (four separate memsets cost a launch and a ~6 us gap each on the
// host-bound exchange path): src_index = -1, hdr = (-1, -1), counters = 0, stats = 0
__global__ void __launch_bounds__(GS_BLOCK) exchange_init_kernel(size_t rows, uint32_t world, int32_t *__restrict__ src_index,
                                                                 int2 *__restrict__ hdr, uint32_t *__restrict__ counters,
                                                                 uint32_t *__restrict__ stats) {
    const size_t i = (size_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i < rows) {
        src_index[i] = -1;
        hdr[i] = make_int2(-1, -1);
    }
    if (i < world) counters[i] = 0u;
    if (i < 2) stats[i] = 0u;
}

// After the all-to-all: the overflow flags of ALL senders (bit 30 of the count in the header row of every received chunk)
// and this rank's own statistics, written where the host can read them -- `out` may be pinned host memory.
__global__ void exchange_flags_kernel(uint32_t world, const int32_t *__restrict__ recv, uint32_t width,
                                      const int64_t *__restrict__ hdr_rows, const uint32_t *__restrict__ stats,
                                      int32_t *__restrict__ out) {
    __shared__ uint32_t s_over;
    if (threadIdx.x == 0) s_over = 0u;
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < world; d += blockDim.x)
        if ((recv[(size_t)hdr_rows[d] * width + 1] >> 30) & 1) atomicOr(&s_over, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (int32_t)s_over;
        out[1] = (int32_t)stats[0];
        out[2] = (int32_t)stats[1];
    }
}

}  // namespace

extern "C" int32_t gs_exchange_flags(uint32_t world, const int32_t *recv, uint32_t row_width, const int64_t *hdr_rows,
                                     const uint32_t *stats, int32_t *out3, gs_stream_t stream) {
    GS_CHECK_ARG(world >= 1 && recv && hdr_rows && stats && out3 && row_width >= 2, "null pointer / row width");
    hipLaunchKernelGGL(exchange_flags_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, world, recv, row_width, hdr_rows, stats, out3);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_exchange_compact(uint32_t C_total, uint32_t N, uint32_t C_local, uint32_t world, uint32_t cap, uint32_t N_total,
                                       uint32_t N_off, const int32_t *radii, int32_t *src_index, int32_t *hdr, uint32_t *counters,
                                       uint32_t *stats, gs_stream_t stream) {
    GS_CHECK_ARG(C_local >= 1 && world >= 1 && C_total == C_local * world && world <= 1024, "C_total = C_local * world, world <= 1024");
    GS_CHECK_ARG(src_index && hdr && counters && stats, "null pointer");
    GS_CHECK_ARG((uint64_t)C_total * N < (1ull << 31) && (uint64_t)C_local * N_total < (1ull << 31), "row indices must fit 31 bits");
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)world * (cap + 1);
    hipLaunchKernelGGL(exchange_init_kernel, dim3(gs_div_up(std::max(rows, (size_t)world), GS_BLOCK)), dim3(GS_BLOCK), 0, st, rows, world,
                       src_index, (int2 *)hdr, counters, stats);
    if (N > 0) {
        GS_CHECK_ARG(radii != nullptr, "null pointer");
        hipLaunchKernelGGL(exchange_compact_kernel, dim3(gs_div_up(N, COMPACT_BLOCK), C_total), dim3(COMPACT_BLOCK), 0, st, N, C_local, cap, N_total,
                           N_off, radii, src_index, (int2 *)hdr, counters);
        GS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(exchange_header_kernel, dim3(1), dim3(1024), 0, st, world, cap, counters, (int2 *)hdr, stats);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_rows_pack(uint64_t n_rows, int32_t n_parts, const void *const *parts, const int32_t *widths,
                                const int64_t *row_strides, void *wire, gs_stream_t stream) {
    return rows_launch<true>(n_rows, n_parts, (void *const *)parts, widths, row_strides, wire, stream);
}

extern "C" int32_t gs_rows_unpack(uint64_t n_rows, int32_t n_parts, void *const *parts, const int32_t *widths,
                                  const int64_t *row_strides, const void *wire, gs_stream_t stream) {
    return rows_launch<false>(n_rows, n_parts, parts, widths, row_strides, (void *)wire, stream);
}

extern "C" int32_t gs_rows_pack_indexed(uint64_t n_rows, int32_t n_parts, const void *const *parts, const int32_t *widths,
                                        const int64_t *row_strides, const int32_t *indexed, const int32_t *row_index,
                                        int64_t index_stride, void *wire, gs_stream_t stream) {
    return rows_launch<true>(n_rows, n_parts, (void *const *)parts, widths, row_strides, wire, stream, indexed, row_index, index_stride);
}

extern "C" int32_t gs_rows_unpack_indexed(uint64_t n_rows, int32_t n_parts, void *const *parts, const int32_t *widths,
                                          const int64_t *row_strides, const int32_t *indexed, const int32_t *row_index,
                                          int64_t index_stride, const void *wire, gs_stream_t stream) {
    return rows_launch<false>(n_rows, n_parts, parts, widths, row_strides, (void *)wire, stream, indexed, row_index, index_stride);
}

// ---------------------------------------------------------------------------------------------------------------------
// Row gather and its adjoint for the packed (COO) pipeline: `opacities[gaussian_ids]`, `colors[gaussian_ids]`,
// `means[gaussian_ids]` of reference gsplat/rendering.py:325, 365-380.  torch's backward of such an index sorts the ids
// and runs ~45 small kernels (0.45 ms per step at 2.8 M splats); here it is one pass of float atomics (ids repeat only
// across cameras).  One lane per element, the packed side is coalesced.
namespace {

__global__ void __launch_bounds__(GS_BLOCK) gather_rows_kernel(uint64_t total, uint32_t width, const float *__restrict__ src,
                                                               const int64_t *__restrict__ ids, float *__restrict__ out) {
    const uint64_t e = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (e >= total) return;
    const uint64_t r = e / width;
    const uint32_t c = (uint32_t)(e - r * width);
    out[e] = src[(uint64_t)ids[r] * width + c];
}

__global__ void __launch_bounds__(GS_BLOCK) scatter_add_rows_kernel(uint64_t total, uint32_t width, const float *__restrict__ v_out,
                                                                    const int64_t *__restrict__ ids, float *__restrict__ v_src) {
    const uint64_t e = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (e >= total) return;
    const uint64_t r = e / width;
    const uint32_t c = (uint32_t)(e - r * width);
    unsafeAtomicAdd(v_src + (uint64_t)ids[r] * width + c, v_out[e]);
}

// Wire rows that carry their own destination: column 0 of a [n_rows, 1 + width] wire row is a global row index (int32 bit
// pattern; negative = no row), the rest its values.  acc[map[index - lo]][1 + c] += scale * wire[r][1 + c]: the owner side of
// the sparse gradient reduction (distributed.py) adds the rows it received for its block into the block's compact accumulator.
__global__ void __launch_bounds__(GS_BLOCK) scatter_add_wire_rows_kernel(uint64_t total, uint32_t width, const float *__restrict__ wire,
                                                                         const int32_t *__restrict__ map, int32_t lo, float scale,
                                                                         float *__restrict__ acc) {
    const uint64_t e = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (e >= total) return;
    const uint64_t r = e / width;
    const uint32_t c = (uint32_t)(e - r * width);
    const float *w = wire + r * (width + 1u);
    const int32_t idx = __float_as_int(w[0]);
    if (idx < 0) return;
    unsafeAtomicAdd(acc + (uint64_t)map[idx - lo] * (width + 1u) + 1u + c, w[1u + c] * scale);
}

}  // namespace

extern "C" int32_t gs_scatter_add_wire_rows(uint64_t n_rows, uint32_t width, const float *wire, const int32_t *map, int32_t lo,
                                            float scale, float *acc, gs_stream_t stream) {
    if (n_rows == 0 || width == 0) return 0;
    GS_CHECK_ARG(wire && map && acc, "null pointer");
    const uint64_t total = n_rows * width;
    GS_CHECK_ARG(total / GS_BLOCK < (1ull << 31), "too many elements");
    hipLaunchKernelGGL(scatter_add_wire_rows_kernel, dim3((uint32_t)((total + GS_BLOCK - 1) / GS_BLOCK)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, total, width, wire, map, lo, scale, acc);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_gather_rows_f32(uint64_t n_rows, uint32_t width, const float *src, const int64_t *ids, float *out,
                                      gs_stream_t stream) {
    if (n_rows == 0 || width == 0) return 0;
    GS_CHECK_ARG(src && ids && out, "null pointer");
    const uint64_t total = n_rows * width;
    GS_CHECK_ARG(total / GS_BLOCK < (1ull << 31), "too many elements");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)((total + GS_BLOCK - 1) / GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       total, width, src, ids, out);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_scatter_add_rows_f32(uint64_t n_rows, uint32_t width, const float *v_out, const int64_t *ids, float *v_src,
                                           gs_stream_t stream) {
    if (n_rows == 0 || width == 0) return 0;
    GS_CHECK_ARG(v_out && ids && v_src, "null pointer");
    const uint64_t total = n_rows * width;
    GS_CHECK_ARG(total / GS_BLOCK < (1ull << 31), "too many elements");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((uint32_t)((total + GS_BLOCK - 1) / GS_BLOCK)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, total, width, v_out, ids, v_src);
    GS_CHECK_LAUNCH();
    return 0;
}
