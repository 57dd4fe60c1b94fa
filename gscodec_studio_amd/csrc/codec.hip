// codec.hip -- decode of the on-disk quantized attribute format straight into the rasterizer's inputs (gfx950).
//
// "The step after the path" of SURVEY.md section 8(f) rank 3: the reference's eval / viewer path reads the compressed planes
// (gsplat/compression/png_compression.py:166-236 decompress, 312-392 _decompress_png*, 487-520 _decompress_kmeans), builds
// fp32 attribute tensors on the HOST (numpy / torch CPU), moves them to the GPU, undoes the log transform of the means
// (png_compression.py:228-230) and then applies the trainer's activations in front of rasterization()
// (examples/simple_trainer.py:779-786: exp on the scales, sigmoid on the opacities, cat(sh0, shN)).  Here the uint8 planes
// go to the GPU as they are and ONE kernel per splat writes means / quats / scales / opacities / sh0 ready for
// rasterization(): dequantize (same arithmetic as gs_grid_dequantize: float64 like the reference's numpy / torch mix),
// inverse log transform sign(y) expm1(|y|), optional quaternion normalisation, exp, sigmoid.  5 + 3 + 4 + 1 + 3 = 16 bytes
// read and 14 floats written per splat: HBM-bound streaming.  The K-means decode of the higher SH bands is a codebook
// gather (labels -> dequantized centroid rows).
#include "gs_common.h"

namespace {

struct DecodeArgs {
    const uint8_t *means_lo, *means_hi; // [n,3] each (16-bit grid)
    const uint8_t *scales, *quats, *opacities, *sh0; // [n,3] [n,4] [n] [n,3]
    float mins[14], maxs[14];           // channels: means 0..2 | scales 3..5 | quats 6..9 | opacity 10 | sh0 11..13
    uint32_t shift[5];                  // right shift of the stored byte per attribute (k-bit planes keep the value in the top bits)
    double levels[5];                   // 2^bits - 1 per attribute
    float *means, *scales_out, *quats_out, *opac_out, *sh0_out;
    int32_t normalize_quats, activate;
};

// the reference's decode of one value: q / levels (float64) * (maxs - mins as fp32) + mins, rounded to fp32
GS_DEV float dequant(uint32_t q, double levels, float mn, float mx) {
    const float range = __fsub_rn(mx, mn);
    return (float)((double)q / levels * (double)range + (double)mn);
}

__global__ void __launch_bounds__(GS_BLOCK) decode_splats_kernel(uint64_t n, DecodeArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i >= n) return;
    // means: 16-bit grid, then the inverse log transform sign(y) * expm1(|y|)  (gsplat/utils.py:40-41)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t q = ((uint32_t)a.means_hi[i * 3 + c] << 8) + (uint32_t)a.means_lo[i * 3 + c];
        const float y = dequant(q, a.levels[0], a.mins[c], a.maxs[c]);
        const float m = expm1f(fabsf(y));
        a.means[i * 3 + c] = y > 0.f ? m : (y < 0.f ? -m : 0.f * y); // sign(0) * x = 0 (NaN stays NaN)
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = dequant((uint32_t)a.scales[i * 3 + c] >> a.shift[1], a.levels[1], a.mins[3 + c], a.maxs[3 + c]);
        a.scales_out[i * 3 + c] = a.activate ? expf(v) : v;
    }
    float q4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = dequant((uint32_t)a.quats[i * 4 + c] >> a.shift[2], a.levels[2], a.mins[6 + c], a.maxs[6 + c]);
    if (a.normalize_quats) { // F.normalize(dim=-1): x / max(||x||, 1e-12)
        const float nrm = fmaxf(sqrtf(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3]), 1e-12f);
#pragma unroll
        for (int c = 0; c < 4; ++c) q4[c] = q4[c] / nrm;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) a.quats_out[i * 4 + c] = q4[c];
    {
        const float v = dequant((uint32_t)a.opacities[i] >> a.shift[3], a.levels[3], a.mins[10], a.maxs[10]);
        a.opac_out[i] = a.activate ? 1.f / (1.f + expf(-v)) : v;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        a.sh0_out[i * 3 + c] = dequant((uint32_t)a.sh0[i * 3 + c] >> a.shift[4], a.levels[4], a.mins[11 + c], a.maxs[11 + c]);
}

// K-means decode (png_compression.py:487-520): out[r, :] = centroids_quant[labels[r], :] / levels * (maxs - mins) + mins,
// one scalar range for the whole codebook; float64 like the reference (numpy division, 0-dim fp32 range and offset).
__global__ void __launch_bounds__(GS_BLOCK) kmeans_decode_kernel(uint64_t total, uint32_t width, const int32_t *__restrict__ labels,
                                                                 const uint8_t *__restrict__ centroids, uint32_t n_centroids, double levels,
                                                                 float mn, float mx, float *__restrict__ out, uint32_t *__restrict__ n_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * GS_BLOCK;
    for (uint64_t e = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x; e < total; e += stride) {
        const uint64_t r = e / width;
        const uint32_t c = (uint32_t)(e - r * width);
        // labels come from a file: one outside the codebook (truncated / mismatched shN.npz) must not become a wild read.
        // The reference's centroids[labels] raises IndexError there; here the row decodes to NaN and the count of such
        // labels is reported (n_bad), which the host wrapper turns into the same error.
        const uint32_t lab = (uint32_t)labels[r];
        if (lab >= n_centroids) {
            out[e] = __builtin_nanf("");
            if (c == 0 && n_bad != nullptr) atomicAdd(n_bad, 1u);
            continue;
        }
        out[e] = dequant((uint32_t)centroids[(uint64_t)lab * width + c], levels, mn, mx);
    }
}

} // namespace

extern "C" int32_t gs_decode_splats(
    uint64_t n, const uint8_t *means_lo, const uint8_t *means_hi, const uint8_t *scales, const uint8_t *quats,
    const uint8_t *opacities, const uint8_t *sh0, const float *mins14, const float *maxs14, const uint32_t *bits5,
    int32_t normalize_quats, int32_t activate, float *means_out, float *scales_out, float *quats_out, float *opacities_out,
    float *sh0_out, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(means_lo && means_hi && scales && quats && opacities && sh0 && mins14 && maxs14 && bits5, "null pointer");
    GS_CHECK_ARG(means_out && scales_out && quats_out && opacities_out && sh0_out, "null pointer");
    GS_CHECK_ARG(bits5[0] == 16, "the means are a 16-bit grid");
    DecodeArgs a;
    a.means_lo = means_lo; a.means_hi = means_hi; a.scales = scales; a.quats = quats; a.opacities = opacities; a.sh0 = sh0;
    for (int c = 0; c < 14; ++c) { a.mins[c] = mins14[c]; a.maxs[c] = maxs14[c]; } // HOST arrays (14 floats of meta.json)
    for (int k = 0; k < 5; ++k) {
        GS_CHECK_ARG(k == 0 || (bits5[k] >= 1 && bits5[k] <= 8), "plane bit depths must be in 1..8");
        a.levels[k] = (double)((1u << bits5[k]) - 1u);
        a.shift[k] = k == 0 ? 0u : 8u - bits5[k];
    }
    a.means = means_out; a.scales_out = scales_out; a.quats_out = quats_out; a.opac_out = opacities_out; a.sh0_out = sh0_out;
    a.normalize_quats = normalize_quats; a.activate = activate;
    GS_CHECK_ARG(gs_div_up(n, GS_BLOCK) < (1ull << 31), "too many splats");
    hipLaunchKernelGGL(decode_splats_kernel, dim3((uint32_t)gs_div_up(n, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, a);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_kmeans_decode(uint64_t n_rows, uint32_t width, const int32_t *labels, const uint8_t *centroids_quant,
                                    uint32_t n_centroids, uint32_t bits, float mins, float maxs, float *out, uint32_t *n_bad,
                                    gs_stream_t stream) {
    if (n_rows == 0 || width == 0) return 0;
    GS_CHECK_ARG(labels && centroids_quant && out, "null pointer");
    GS_CHECK_ARG(bits >= 1 && bits <= 8, "bits must be in 1..8");
    const uint64_t total = n_rows * width;
    const uint32_t blocks = (uint32_t)(gs_div_up(total, GS_BLOCK) < 16384u ? gs_div_up(total, GS_BLOCK) : 16384u);
    hipLaunchKernelGGL(kmeans_decode_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, total, width, labels, centroids_quant,
                       n_centroids, (double)((1u << bits) - 1u), mins, maxs, out, n_bad);
    GS_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// PNG scanline reconstruction (host code; the inverse of the five PNG filter types, PNG specification section 9): the
// sequential-in-x part of reading the reference's image grids (imageio / Pillow write them with adaptive filters).
// data: h rows of (1 + stride) bytes -- filter type, then the filtered scanline; out: h rows of stride bytes.
extern "C" int32_t gs_png_unfilter(const uint8_t *data, uint32_t h, uint32_t stride, uint32_t bpp, uint8_t *out) {
    GS_CHECK_ARG(data && out && bpp >= 1 && bpp <= 8 && stride >= bpp, "null pointer / bytes per pixel");
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *src = data + (size_t)y * (stride + 1);
        const uint8_t ft = src[0];
        const uint8_t *in = src + 1, *up = y ? out + (size_t)(y - 1) * stride : nullptr;
        uint8_t *o = out + (size_t)y * stride;
        GS_CHECK_ARG(ft <= 4, "unknown PNG filter type");
        for (uint32_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? o[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            o[x] = (uint8_t)(in[x] + pred);
        }
    }
    return 0;
}
