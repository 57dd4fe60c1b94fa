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

namespace {

constexpr int MAX_PARTS = 8;

constexpr int ROWS_PER_BLOCK = 256;
constexpr int MAX_WIDTH = 64;

struct RowParts {
    uint32_t *ptr[MAX_PARTS];
    int64_t stride[MAX_PARTS];  // row stride in 4-byte elements
    int32_t begin[MAX_PARTS + 1];  // first wire column of part k; begin[n] = wire width
    uint64_t inv[MAX_PARTS];  // floor(2^32 / width) + 1: j / width == (j * inv) >> 32 for the j < 2^14 used here
    int32_t n;
};

// One workgroup moves 256 rows through LDS: the wire side is one contiguous chunk (coalesced), and each part is
// walked in its own element order, so a part whose rows are dense in memory is one contiguous chunk too.
template <bool PACK>
__global__ void __launch_bounds__(GS_BLOCK) rows_kernel(uint64_t n_rows, uint32_t width, RowParts t, uint32_t *__restrict__ wire) {
    extern __shared__ uint32_t tile[];  // [rows][width]
    const uint64_t row0 = (uint64_t)blockIdx.x * ROWS_PER_BLOCK;
    const uint32_t nr = (uint32_t)min((uint64_t)ROWS_PER_BLOCK, n_rows - row0);
    uint32_t *w0 = wire + row0 * width;
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
        p += row0 * s;
        for (uint32_t j = threadIdx.x; j < nr * w; j += GS_BLOCK) {
            const uint32_t r = (uint32_t)(((uint64_t)j * inv) >> 32), c = j - r * w;
            if (PACK)
                tile[r * width + b + c] = p != nullptr ? p[(int64_t)r * s + c] : 0u;
            else
                p[(int64_t)r * s + c] = tile[r * width + b + c];
        }
    }
    if (PACK) {
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < nr * width; j += GS_BLOCK) w0[j] = tile[j];
    }
}

template <bool PACK>
int32_t rows_launch(uint64_t n_rows, int32_t n_parts, void *const *parts, const int32_t *widths, const int64_t *strides, void *wire,
                    gs_stream_t stream) {
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
        w += widths[k];
    }
    for (int k = n_parts; k <= MAX_PARTS; ++k) t.begin[k] = w;
    if (n_rows == 0) return 0;
    GS_CHECK_ARG(wire != nullptr, "null pointer");
    GS_CHECK_ARG(w <= MAX_WIDTH, "wire rows of at most 64 elements");
    const uint64_t blocks = (n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    GS_CHECK_ARG(blocks < (1ull << 31), "too many rows");
    hipLaunchKernelGGL(rows_kernel<PACK>, dim3((uint32_t)blocks), dim3(GS_BLOCK), ROWS_PER_BLOCK * w * sizeof(uint32_t),
                       (hipStream_t)stream, n_rows, (uint32_t)w, t, (uint32_t *)wire);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int32_t gs_rows_pack(uint64_t n_rows, int32_t n_parts, const void *const *parts, const int32_t *widths,
                                const int64_t *row_strides, void *wire, gs_stream_t stream) {
    return rows_launch<true>(n_rows, n_parts, (void *const *)parts, widths, row_strides, wire, stream);
}

extern "C" int32_t gs_rows_unpack(uint64_t n_rows, int32_t n_parts, void *const *parts, const int32_t *widths,
                                  const int64_t *row_strides, const void *wire, gs_stream_t stream) {
    return rows_launch<false>(n_rows, n_parts, parts, widths, row_strides, (void *)wire, stream);
}
