// Where the time of the bucketed pre-sort's two new kernels goes (wall_clock64 stamps of thread 0 of one workgroup, 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPS_PROFILE -DPS_PROFILE_BLOCK=<b> -I gscodec_studio_amd/csrc -o presort_bench tools/presort_bench.hip
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <cmath>
#include <cstring>
void gs_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#include "../gscodec_studio_amd/csrc/radix_sort.hip"
#include "../gscodec_studio_amd/csrc/isect.hip"

int main() {
    const uint32_t n = 1006065;
    std::vector<int32_t> radii(n);
    std::vector<float> depths(n);
    srand(1);
    for (uint32_t i = 0; i < n; ++i) {
        const bool region = ((i / 111785) % 3) == 1; // visibility is spatially clustered, like the tiled garden scene
        radii[i] = (rand() % 100) < (region ? 80 : 4) ? 5 : 0;
        depths[i] = 0.6f + 7.7f * (float)rand() / RAND_MAX * (float)rand() / RAND_MAX;
    }
    int32_t *d_r; float *d_d; int64_t *d_split, *d_keys; int32_t *d_vals, *d_perm, *d_tiles; uint32_t *d_nk, *d_gs; void *temp;
    hipMalloc(&d_r, n * 4); hipMalloc(&d_d, n * 4); hipMalloc(&d_split, (size_t)gs_presort_split_elems() * 8); hipMalloc(&d_keys, n * 8); hipMalloc(&d_vals, n * 4);
    hipMalloc(&d_perm, n * 4); hipMalloc(&d_tiles, n * 4); hipMalloc(&d_nk, 4); hipMalloc(&d_gs, (n / 128 + 1) * 4);
    const size_t tb = gs_presort_temp_bytes(n);
    hipMalloc(&temp, tb);
    hipMemcpy(d_r, radii.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_d, depths.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_tiles, 0, n * 4);
    std::vector<int64_t> keys(n);
    std::vector<int32_t> vals(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t db; memcpy(&db, &depths[i], 4);
        keys[i] = (int64_t)(((uint64_t)(radii[i] > 0 ? db : 0x7fffffffu) << 32) | i);
        vals[i] = (int32_t)i;
    }
    hipMemcpy(d_keys, keys.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_vals, vals.data(), n * 4, hipMemcpyHostToDevice);
    float *d_m2; int32_t *d_tpg; int32_t *d_bs;
    hipMalloc(&d_m2, n * 8); hipMemset(d_m2, 0, n * 8); hipMalloc(&d_tpg, n * 4); hipMalloc(&d_bs, 4096 * 4);
    hipEvent_t e[5];
    for (auto &x : e) hipEventCreate(&x);
    float t_split = 0, t_count = 0, t_bucket = 0;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e[0]);
        gs_presort_split(n, d_r, d_d, d_split, nullptr);
        hipEventRecord(e[1]);
        gs_isect_count_keys(n, d_m2, 2, d_r, d_d, 16, 120, 68, d_tpg, d_keys, d_vals, d_bs, temp, tb, d_split, nullptr);
        hipEventRecord(e[2]);
        gs_presort_buckets(n, d_keys, d_vals, d_split, d_perm, d_nk, temp, tb, d_tpg, d_gs, 7, 0, nullptr);
        hipEventRecord(e[3]);
        hipDeviceSynchronize();
        hipEventElapsedTime(&t_split, e[0], e[1]); hipEventElapsedTime(&t_count, e[1], e[2]); hipEventElapsedTime(&t_bucket, e[2], e[3]);
    }
    printf("events (ms): split %.4f  count %.4f  scan+scatter+local %.4f\n", t_split, t_count, t_bucket);
    // the slowest local-sort workgroup: the largest bucket
    {
        const SortLayout L = sort_layout(n);
        std::vector<uint32_t> tot(256);
        hipMemcpy(tot.data(), (char *)temp + L.off_totals, 256 * 4, hipMemcpyDeviceToHost);
        unsigned int arg = (unsigned int)(std::max_element(tot.begin(), tot.end()) - tot.begin());
        printf("largest bucket: #%u with %u keys (mean %.0f)\n", arg, tot[arg], 0.0 + std::accumulate(tot.begin(), tot.end(), 0.0) / 256);
        hipMemcpyToSymbol(HIP_SYMBOL(ps_profile_block), &arg, 4);
        gs_presort_split(n, d_r, d_d, d_split, nullptr);
        gs_isect_count_keys(n, d_m2, 2, d_r, d_d, 16, 120, 68, d_tpg, d_keys, d_vals, d_bs, temp, tb, d_split, nullptr);
        gs_presort_buckets(n, d_keys, d_vals, d_split, d_perm, d_nk, temp, tb, d_tpg, d_gs, 7, 0, nullptr);
        hipDeviceSynchronize();
        unsigned long long s2[64];
        hipMemcpyFromSymbol(s2, HIP_SYMBOL(ps_stamps), sizeof(s2));
        auto us2 = [&](int a, int b) { return (double)(s2[b] - s2[a]) / 100.0; };
        printf("local kernel, largest bucket (us): prelude %.1f  load %.1f  sort %.1f  output %.1f  total %.1f\n", us2(8, 9), us2(9, 10), us2(10, 11), us2(11, 12), us2(8, 12));
        unsigned int zero = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(ps_profile_block), &zero, 4);
    }
    unsigned long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(ps_stamps), sizeof(st));
    auto us = [&](int a, int b) { return (double)(st[b] - st[a]) / 100.0; };
    printf("split kernel (us): gather %.1f  scan+compaction %.1f  histogram+prefix %.1f  splitters %.1f  total %.1f\n", us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(0, 4));
    printf("  block-0 in-LDS sort (us):");
    for (int p = 0; p < 2; ++p) printf(" pass %d: rank %.1f scan %.1f%s", p, us(17 + 4 * p, 18 + 4 * p), us(18 + 4 * p, 19 + 4 * p), p ? "\n" : " |");
    printf("local kernel, block %d (us): prelude %.1f  load %.1f  sort %.1f  output %.1f  total %.1f\n", 0, us(8, 9), us(9, 10), us(10, 11), us(11, 12), us(8, 12));
    return 0;
}
