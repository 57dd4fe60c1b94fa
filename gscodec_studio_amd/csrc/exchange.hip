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

// Indexed rows of up to 64 elements (the sparse gradient reduction: 60-float rows gathered from / scattered to five gradient
// tensors at the splat's index): ONE WAVE PER ROW, lane = wire column.  The tile kernel above keeps 60 KB of LDS per workgroup
// (two workgroups per CU) and walks every part element by element behind an LDS index read: 96 us to pack 293 K rows of 60
// floats; here nothing is staged, the index is a wave-uniform load, the wire side of a row is one 240-byte access and the
// waves (eight per SIMD) keep four rows in flight each.
template <bool PACK>
__global__ void __launch_bounds__(GS_BLOCK) rows_wave_kernel(uint64_t n_rows, uint32_t width, RowParts t, uint32_t *__restrict__ wire,
                                                             const int32_t *__restrict__ row_index, int64_t index_stride) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (GS_BLOCK / GS_WAVE) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (GS_BLOCK / GS_WAVE);
    // my column's part
    int k = 0;
    for (int i = 1; i < t.n; ++i)
        if ((int32_t)lane >= t.begin[i]) k = i;
    const bool on = lane < width;
    uint32_t *p = on ? t.ptr[k] : nullptr;
    const int64_t s = t.stride[k];
    const uint32_t c = lane - (uint32_t)t.begin[k];
    const bool idx = (t.indexed >> k) & 1u;
    constexpr int UNROLL = 4;
    for (uint64_t r0 = wave * UNROLL; r0 < n_rows; r0 += n_waves * UNROLL) {
        int64_t row[UNROLL];
        uint32_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t r = r0 + u;
            row[u] = -1;
            if (r < n_rows) row[u] = idx ? (int64_t)row_index[(int64_t)r * index_stride] : (int64_t)r;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t r = r0 + u;
            v[u] = 0u;
            if (r < n_rows && on) {
                if (PACK) v[u] = (p != nullptr && row[u] >= 0) ? p[row[u] * s + c] : 0u;
                else v[u] = wire[r * width + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t r = r0 + u;
            if (r < n_rows && on) {
                if (PACK) wire[r * width + lane] = v[u];
                else if (p != nullptr && row[u] >= 0) p[row[u] * s + c] = v[u];
            }
        }
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
    if (t.indexed != 0u && w <= GS_WAVE) { // narrow indexed rows: one wave per row (see rows_wave_kernel)
        const uint64_t waves = (n_rows + 3) / 4;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((waves + 3) / 4, 256ull * 8ull);
        hipLaunchKernelGGL(rows_wave_kernel<PACK>, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, n_rows, (uint32_t)w, t, (uint32_t *)wire,
                           row_index, index_stride);
        GS_CHECK_LAUNCH();
        return 0;
    }
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
// (+ optionally the RECEIVER's radii, zero before the scatter of the received rows fills in the visible ones: the same
// process allocates them before the all-to-all, and this launch has the GPU to itself -- a memset behind the collective
// was two more launches on the critical path)
__global__ void __launch_bounds__(GS_BLOCK) exchange_init_kernel(size_t rows, uint32_t world, int32_t *__restrict__ src_index,
                                                                 int2 *__restrict__ hdr, uint32_t *__restrict__ counters,
                                                                 uint32_t *__restrict__ stats, int32_t *__restrict__ zero, size_t n_zero) {
    const size_t i = (size_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i < rows) {
        src_index[i] = -1;
        hdr[i] = make_int2(-1, -1);
    }
    if (i < world) counters[i] = 0u;
    if (i < 2) stats[i] = 0u;
    const size_t stride = (size_t)gridDim.x * GS_BLOCK;
    for (size_t j = i; j < n_zero; j += stride) zero[j] = 0;
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

// Splat rows (16 floats, include/gsplat_hip.h) around the exchange: four lanes move one 64-byte row as four float4.
// gather: out[r] = src[index[r]] (zeros for a negative index); columns 12 / 13 carry tag[r] (the destination row and the
// chunk header of gs_exchange_compact) when given.
__global__ void __launch_bounds__(GS_BLOCK) rows16_gather_kernel(uint64_t n_rows, const int32_t *__restrict__ index, int64_t index_stride,
                                                                 const float4 *__restrict__ src, const int2 *__restrict__ tag,
                                                                 float4 *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    const uint64_t r = t >> 2;
    const uint32_t q = (uint32_t)t & 3u;
    if (r >= n_rows) return;
    const int32_t i = index[r * index_stride];
    float4 v = (i >= 0 && src != nullptr) ? src[(size_t)i * 4u + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (q == 3u && tag != nullptr) {
        const int2 g = tag[r];
        v.x = __int_as_float(g.x);
        v.y = __int_as_float(g.y);
    }
    out[r * 4u + q] = v;
}

// scatter: dst[index[r]] = wire[r] for index[r] >= 0 (index may be a column of the wire itself); radii / depths (optional) get
// the row's columns 10 / 9 at the same element -- the dense arrays the binning kernels stream through.
__global__ void __launch_bounds__(GS_BLOCK) rows16_scatter_kernel(uint64_t n_rows, const int32_t *__restrict__ index, int64_t index_stride,
                                                                  const float4 *__restrict__ wire, float4 *__restrict__ dst,
                                                                  int32_t *__restrict__ radii, float *__restrict__ depths) {
    const uint64_t t = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    const uint64_t r = t >> 2;
    const uint32_t q = (uint32_t)t & 3u;
    if (r >= n_rows) return;
    const int32_t i = index[r * index_stride];
    if (i < 0 || dst == nullptr) return;
    const float4 v = wire[r * 4u + q];
    dst[(size_t)i * 4u + q] = v;
    if (q == 2u) { // columns 8..11: colour 2 | depth | radius bits | compensation
        if (radii != nullptr) radii[i] = __float_as_int(v.z);
        if (depths != nullptr) depths[i] = v.y;
    }
}

}  // namespace

extern "C" int32_t gs_rows16_gather(uint64_t n_rows, const int32_t *index, int64_t index_stride, const float *src_rows,
                                    const int32_t *tag, float *out_rows, gs_stream_t stream) {
    if (n_rows == 0) return 0;
    GS_CHECK_ARG(index && out_rows && index_stride >= 1, "null pointer / index stride"); // (src_rows NULL: an empty shard, every index negative)
    GS_CHECK_ARG((uintptr_t)src_rows % 16 == 0 && (uintptr_t)out_rows % 16 == 0 && (uintptr_t)tag % 8 == 0, "rows must be 16-byte aligned");
    hipLaunchKernelGGL(rows16_gather_kernel, dim3((uint32_t)gs_div_up(n_rows * 4u, (uint64_t)GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       n_rows, index, index_stride, (const float4 *)src_rows, (const int2 *)tag, (float4 *)out_rows);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_rows16_scatter(uint64_t n_rows, const int32_t *index, int64_t index_stride, const float *wire_rows,
                                     float *dst_rows, int32_t *radii, float *depths, gs_stream_t stream) {
    if (n_rows == 0) return 0;
    GS_CHECK_ARG(index && wire_rows && index_stride >= 1, "null pointer / index stride"); // (dst_rows NULL: an empty shard)
    GS_CHECK_ARG((uintptr_t)wire_rows % 16 == 0 && (uintptr_t)dst_rows % 16 == 0, "rows must be 16-byte aligned");
    hipLaunchKernelGGL(rows16_scatter_kernel, dim3((uint32_t)gs_div_up(n_rows * 4u, (uint64_t)GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       n_rows, index, index_stride, (const float4 *)wire_rows, (float4 *)dst_rows, radii, depths);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_exchange_flags(uint32_t world, const int32_t *recv, uint32_t row_width, const int64_t *hdr_rows,
                                     const uint32_t *stats, int32_t *out3, gs_stream_t stream) {
    GS_CHECK_ARG(world >= 1 && recv && hdr_rows && stats && out3 && row_width >= 2, "null pointer / row width");
    hipLaunchKernelGGL(exchange_flags_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, world, recv, row_width, hdr_rows, stats, out3);
    GS_CHECK_LAUNCH();
    return 0;
}

static int32_t exchange_compact_impl(uint32_t C_total, uint32_t N, uint32_t C_local, uint32_t world, uint32_t cap, uint32_t N_total,
                                     uint32_t N_off, const int32_t *radii, int32_t *src_index, int32_t *hdr, uint32_t *counters,
                                     uint32_t *stats, int32_t *zero, size_t n_zero, gs_stream_t stream) {
    GS_CHECK_ARG(C_local >= 1 && world >= 1 && C_total == C_local * world && world <= 1024, "C_total = C_local * world, world <= 1024");
    GS_CHECK_ARG(src_index && hdr && counters && stats, "null pointer");
    GS_CHECK_ARG((uint64_t)C_total * N < (1ull << 31) && (uint64_t)C_local * N_total < (1ull << 31), "row indices must fit 31 bits");
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)world * (cap + 1);
    hipLaunchKernelGGL(exchange_init_kernel, dim3(gs_div_up(std::max(rows, (size_t)world), GS_BLOCK)), dim3(GS_BLOCK), 0, st, rows, world,
                       src_index, (int2 *)hdr, counters, stats, zero, zero ? n_zero : (size_t)0);
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

extern "C" int32_t gs_exchange_compact(uint32_t C_total, uint32_t N, uint32_t C_local, uint32_t world, uint32_t cap, uint32_t N_total,
                                       uint32_t N_off, const int32_t *radii, int32_t *src_index, int32_t *hdr, uint32_t *counters,
                                       uint32_t *stats, gs_stream_t stream) {
    return exchange_compact_impl(C_total, N, C_local, world, cap, N_total, N_off, radii, src_index, hdr, counters, stats, nullptr, 0, stream);
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

// ---------------------------------------------------------------------------------------------------------------------
// Plan of the sparse gradient reduction of the camera-sharded mode (distributed.py: plan_sparse_grad_exchange), built in the
// forward pass from the visibility masks of all ranks.  The torch formulation (any / sum / cat / nonzero_static x 2 / cumsum /
// where / casts) was ~35 launches of a few microseconds each: 0.34 ms of host-bound work per step.  Here: one kernel for the
// mask, two for everything else.
//   masks   uint8 [world, n_pad]   (n_pad = world * block; splat n belongs to owner n / block)
//   counts  int32 [world * world + world], zero-filled by the caller:  rows[r][o] = splats of owner o that rank r saw,
//                                                                     then urows[o] = splats of owner o that ANY rank saw
//   send_idx [n_pad]  my visible splats, ascending (entries behind their count: undefined)
//   urank    [n_pad]  position of every union splat in the ascending list of ALL union splats
//   uidx     [n_pad]  that list
constexpr int PLAN_ITEMS = 8;
constexpr int PLAN_TILE = GS_BLOCK * PLAN_ITEMS; // 2048 splats per workgroup
constexpr int PLAN_MAX_WORLD = 16;

__global__ void __launch_bounds__(GS_BLOCK) dp_vis_kernel(uint32_t C, uint32_t N, uint32_t n_pad, const int32_t *__restrict__ radii,
                                                          uint8_t *__restrict__ vis) {
    const uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (n >= n_pad) return;
    uint8_t v = 0;
    if (n < N)
        for (uint32_t c = 0; c < C; ++c) v |= radii[(size_t)c * N + n] > 0 ? 1 : 0;
    vis[n] = v;
}

__global__ void __launch_bounds__(GS_BLOCK) dp_plan_count_kernel(uint32_t world, uint32_t rank, uint32_t n_pad, uint32_t block,
                                                                 const uint8_t *__restrict__ masks, uint2 *__restrict__ tile_counts,
                                                                 int32_t *__restrict__ counts) {
    __shared__ int32_t s_cnt[PLAN_MAX_WORLD + 1][PLAN_MAX_WORLD]; // [rank | union][owner]
    __shared__ uint32_t s_red[2][GS_BLOCK / GS_WAVE];
    for (uint32_t i = threadIdx.x; i < (PLAN_MAX_WORLD + 1) * PLAN_MAX_WORLD; i += GS_BLOCK) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t n0 = blockIdx.x * PLAN_TILE + threadIdx.x * PLAN_ITEMS;
    uint32_t mine = 0, uni = 0;
    for (int k = 0; k < PLAN_ITEMS; ++k) {
        const uint32_t n = n0 + k;
        if (n >= n_pad) break;
        const uint32_t o = n / block;
        bool any = false;
        for (uint32_t r = 0; r < world; ++r) {
            if (masks[(size_t)r * n_pad + n]) {
                any = true;
                atomicAdd(&s_cnt[r][o], 1);
                if (r == rank) ++mine;
            }
        }
        if (any) {
            ++uni;
            atomicAdd(&s_cnt[PLAN_MAX_WORLD][o], 1);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mine += __shfl_xor(mine, off, 64);
        uni += __shfl_xor(uni, off, 64);
    }
    if ((threadIdx.x & 63u) == 0u) {
        s_red[0][threadIdx.x >> 6] = mine;
        s_red[1][threadIdx.x >> 6] = uni;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        tile_counts[blockIdx.x] = make_uint2(s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3], s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]);
    for (uint32_t i = threadIdx.x; i < (world + 1) * world; i += GS_BLOCK) {
        const uint32_t r = i / world, o = i % world;
        const int32_t c = r < world ? s_cnt[r][o] : s_cnt[PLAN_MAX_WORLD][o];
        if (c != 0) atomicAdd(&counts[i], c);
    }
}

__global__ void __launch_bounds__(GS_BLOCK) dp_plan_fill_kernel(uint32_t world, uint32_t rank, uint32_t n_pad, const uint8_t *__restrict__ masks,
                                                                const uint2 *__restrict__ tile_counts, int32_t *__restrict__ send_idx,
                                                                int32_t *__restrict__ urank, int32_t *__restrict__ uidx) {
    __shared__ uint32_t s_pre[2][GS_BLOCK / GS_WAVE];
    __shared__ uint32_t s_w[2][GS_BLOCK / GS_WAVE];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // offsets of this tile: its predecessors' counts (every tile adds them up itself: a few thousand tiles at most)
    uint32_t pm = 0, pu = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += GS_BLOCK) {
        const uint2 c = tile_counts[b];
        pm += c.x;
        pu += c.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pm += __shfl_xor(pm, off, 64);
        pu += __shfl_xor(pu, off, 64);
    }
    if (lane == 0) {
        s_pre[0][wave] = pm;
        s_pre[1][wave] = pu;
    }
    const uint32_t n0 = blockIdx.x * PLAN_TILE + threadIdx.x * PLAN_ITEMS;
    uint32_t fm = 0, fu = 0; // flags of my 8 splats
    for (int k = 0; k < PLAN_ITEMS; ++k) {
        const uint32_t n = n0 + k;
        if (n >= n_pad) break;
        bool any = false;
        for (uint32_t r = 0; r < world; ++r) any |= masks[(size_t)r * n_pad + n] != 0;
        if (masks[(size_t)rank * n_pad + n]) fm |= 1u << k;
        if (any) fu |= 1u << k;
    }
    uint32_t im = (uint32_t)__popc(fm), iu = (uint32_t)__popc(fu);
    const uint32_t cm = im, cu = iu;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = __shfl_up(im, off, 64), b = __shfl_up(iu, off, 64);
        if (lane >= (uint32_t)off) {
            im += a;
            iu += b;
        }
    }
    if (lane == 63u) {
        s_w[0][wave] = im;
        s_w[1][wave] = iu;
    }
    __syncthreads();
    uint32_t bm = s_pre[0][0] + s_pre[0][1] + s_pre[0][2] + s_pre[0][3] + im - cm;
    uint32_t bu = s_pre[1][0] + s_pre[1][1] + s_pre[1][2] + s_pre[1][3] + iu - cu;
    for (uint32_t w = 0; w < wave; ++w) {
        bm += s_w[0][w];
        bu += s_w[1][w];
    }
    for (int k = 0; k < PLAN_ITEMS; ++k) {
        const uint32_t n = n0 + k;
        if (n >= n_pad) break;
        if ((fm >> k) & 1u) send_idx[bm++] = (int32_t)n;
        if ((fu >> k) & 1u) {
            urank[n] = (int32_t)bu;
            uidx[bu++] = (int32_t)n;
        }
    }
}

// The owner side of the sparse gradient reduction WITHOUT atomics.  The rows an owner receives (wire, [n_recv, 1 + width],
// column 0 = global splat index as an int32 bit pattern) come in one chunk per sender; a union splat of the owner's block gets
// at most one row from every sender.  Scatter-adding them into a zero-filled accumulator took a 36 us fill + 82 us of float
// atomics (17 M of them) for 293 K rows of 59 floats; instead
//   dp_inv_kernel      notes, per (sender, accumulator row), WHICH received row belongs there (plain stores: the pair is unique),
//   dp_reduce_kernel   then writes every accumulator row once: its index column, and the sum of its <= world received rows.
struct ChunkStarts {
    int64_t v[PLAN_MAX_WORLD + 1]; // first received row of every sender's chunk; v[world] = n_recv
};

__global__ void __launch_bounds__(GS_BLOCK) dp_inv_kernel(uint64_t n_recv, uint32_t row_w, uint32_t world, ChunkStarts cs, const float *__restrict__ wire,
                                                          const int32_t *__restrict__ map, int32_t map_offset, uint64_t umax,
                                                          int32_t *__restrict__ inv) {
    const uint64_t r = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (r >= n_recv) return;
    const int32_t idx = __float_as_int(wire[r * row_w]);
    if (idx < 0) return;
    uint32_t snd = 0;
    for (uint32_t k = 1; k < world; ++k)
        if ((int64_t)r >= cs.v[k]) snd = k;
    inv[(uint64_t)snd * umax + (uint64_t)(map[idx] - map_offset)] = (int32_t)r;
}

__global__ void __launch_bounds__(GS_BLOCK) dp_reduce_kernel(uint64_t umax, uint32_t width, uint32_t world, const float *__restrict__ wire,
                                                             const int32_t *__restrict__ inv, const int32_t *__restrict__ uidx, uint32_t n_valid,
                                                             float scale, float *__restrict__ acc) {
    // one wave per accumulator row, lane = column; the per-sender row numbers are wave-uniform loads.  Four rows per
    // iteration: a row is a dependent chain (row number -> received row -> store), one at a time it took 67 us for 293 K rows
    const uint32_t lane = threadIdx.x & 63u, row_w = width + 1u;
    const uint64_t wave = (uint64_t)blockIdx.x * (GS_BLOCK / GS_WAVE) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (GS_BLOCK / GS_WAVE);
    constexpr int UNROLL = 4;
    for (uint64_t u0 = wave * UNROLL; u0 < umax; u0 += n_waves * UNROLL) {
        for (uint32_t c0 = 0; c0 < width; c0 += GS_WAVE) { // (one trip for rows of up to 64 values)
            const uint32_t c = c0 + lane;
            float sum[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) sum[k] = 0.f;
            for (uint32_t s = 0; s < world; ++s) {
                int32_t r[UNROLL];
#pragma unroll
                for (int k = 0; k < UNROLL; ++k) r[k] = (u0 + k < umax) ? inv[(uint64_t)s * umax + u0 + k] : -1;
#pragma unroll
                for (int k = 0; k < UNROLL; ++k)
                    if (r[k] >= 0 && c < width) sum[k] += wire[(uint64_t)r[k] * row_w + 1u + c];
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
                if (u0 + k < umax && c < width) acc[(u0 + k) * row_w + 1u + c] = sum[k] * scale;
        }
        if (lane < (uint32_t)UNROLL && u0 + lane < umax) acc[(u0 + lane) * row_w] = __int_as_float(u0 + lane < n_valid ? uidx[u0 + lane] : -1);
    }
}

}  // namespace

extern "C" uint32_t gs_dp_plan_tiles(uint32_t n_pad) { return gs_div_up(n_pad, PLAN_TILE); }

extern "C" int32_t gs_dp_visibility(uint32_t C, uint32_t N, uint32_t n_pad, const int32_t *radii, uint8_t *vis, gs_stream_t stream) {
    if (n_pad == 0) return 0;
    GS_CHECK_ARG(vis && (radii || C * N == 0) && n_pad >= N, "null pointer / n_pad < N");
    hipLaunchKernelGGL(dp_vis_kernel, dim3(gs_div_up(n_pad, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, n_pad, radii, vis);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_dp_plan(uint32_t world, uint32_t rank, uint32_t n_pad, uint32_t block, const uint8_t *masks, void *tile_counts,
                              int32_t *counts, int32_t *send_idx, int32_t *urank, int32_t *uidx, gs_stream_t stream) {
    if (n_pad == 0) return 0;
    GS_CHECK_ARG(masks && tile_counts && counts && send_idx && urank && uidx, "null pointer");
    GS_CHECK_ARG(world >= 1 && world <= (uint32_t)PLAN_MAX_WORLD && rank < world && block >= 1 && (uint64_t)block * world == n_pad,
                 "world in 1..16, n_pad = world * block");
    const uint32_t tiles = gs_div_up(n_pad, PLAN_TILE);
    GS_CHECK_ARG(tiles <= 65536u, "too many splats for the two-launch plan (the caller falls back to the torch formulation)");
    hipLaunchKernelGGL(dp_plan_count_kernel, dim3(tiles), dim3(GS_BLOCK), 0, (hipStream_t)stream, world, rank, n_pad, block, masks,
                       (uint2 *)tile_counts, counts);
    hipLaunchKernelGGL(dp_plan_fill_kernel, dim3(tiles), dim3(GS_BLOCK), 0, (hipStream_t)stream, world, rank, n_pad, masks,
                       (const uint2 *)tile_counts, send_idx, urank, uidx);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_dp_reduce_rows(uint64_t n_recv, uint32_t width, uint32_t world, const float *wire, const int64_t *chunk_starts,
                                     const int32_t *map, int32_t map_offset, uint64_t umax, uint32_t n_valid, const int32_t *uidx, float scale,
                                     int32_t *inv, float *acc, gs_stream_t stream) {
    if (umax == 0) return 0;
    GS_CHECK_ARG(acc && inv && chunk_starts && (wire || n_recv == 0) && (map || n_recv == 0) && (uidx || n_valid == 0), "null pointer");
    GS_CHECK_ARG(world >= 1 && world <= (uint32_t)PLAN_MAX_WORLD && width >= 1, "world in 1..16");
    GS_CHECK_ARG(n_recv < (1ull << 31) && umax < (1ull << 31), "too many rows");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(inv, 0xff, (size_t)world * umax * sizeof(int32_t), st) != hipSuccess) { gs_set_error("gs_dp_reduce_rows: memset failed"); return 1; }
    ChunkStarts cs;
    for (uint32_t k = 0; k <= world; ++k) cs.v[k] = chunk_starts[k];
    if (n_recv > 0)
        hipLaunchKernelGGL(dp_inv_kernel, dim3(gs_div_up(n_recv, GS_BLOCK)), dim3(GS_BLOCK), 0, st, n_recv, width + 1u, world, cs, wire, map, map_offset,
                           umax, inv);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((umax + 15) / 16, 256ull * 8ull);
    hipLaunchKernelGGL(dp_reduce_kernel, dim3(blocks), dim3(GS_BLOCK), 0, st, umax, width, world, wire, inv, uidx, n_valid, scale, acc);
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

extern "C" int32_t gs_exchange_rows_send(uint32_t C_total, uint32_t N, uint32_t C_local, uint32_t world, uint32_t cap, uint32_t N_total,
                                         uint32_t N_off, const int32_t *radii, const float *rows, int32_t *src_index, int32_t *hdr,
                                         uint32_t *counters, uint32_t *stats, float *send_rows, int32_t *zero_radii, uint64_t n_zero,
                                         gs_stream_t stream) {
    if (int32_t rc = exchange_compact_impl(C_total, N, C_local, world, cap, N_total, N_off, radii, src_index, hdr, counters, stats, zero_radii,
                                           (size_t)n_zero, stream))
        return rc;
    return gs_rows16_gather((uint64_t)world * (cap + 1), src_index, 1, rows, hdr, send_rows, stream);
}

extern "C" int32_t gs_exchange_rows_recv(uint64_t n_recv, const float *recv_rows, uint64_t n_dst, float *dst_rows, int32_t *radii,
                                         float *depths, uint32_t world, const int64_t *hdr_rows, const uint32_t *stats, int32_t *out3,
                                         int32_t radii_zeroed, gs_stream_t stream) {
    GS_CHECK_ARG(recv_rows != nullptr || n_recv == 0, "null pointer");
    if (n_dst > 0 && !radii_zeroed) {
        GS_CHECK_ARG(radii != nullptr, "null pointer");
        if (hipMemsetAsync(radii, 0, n_dst * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) { gs_set_error("gs_exchange_rows_recv: memset failed"); return 1; }
    }
    const int32_t *index = reinterpret_cast<const int32_t *>(recv_rows) + 12;
    if (int32_t rc = gs_rows16_scatter(n_recv, index, 16, recv_rows, dst_rows, radii, depths, stream)) return rc;
    return gs_exchange_flags(world, index, 16u, hdr_rows, stats, out3, stream);
}
