// Workgroup dispatch rate on MI355X: empty workgroups of 256 threads with various LDS sizes / counts.
// hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o build_abl/dispatch_rate && ./build_abl/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) empty_kernel(int *out, int spin) {
    extern __shared__ int lds[];
    if (threadIdx.x == 0) lds[0] = blockIdx.x;
    __syncthreads();
    int v = lds[0];
    for (int i = 0; i < spin; ++i) v = v * 1664525 + 1013904223;
    if (v == 12345 && out) out[0] = v;
}
int main() {
    int *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {1280, 8160, 16320, 32640};
    const int ldss[] = {0, 8192, 29520, 65536};
    const int spins[] = {0, 2000};
    for (int spin : spins) for (int lds : ldss) for (int g : grids) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), lds, 0, d, spin);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), lds, 0, d, spin);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("spin %5d lds %6d grid %6d: %8.1f us/launch  %6.1f ns/WG\n", spin, lds, g, ms * 100.f, ms * 1e5f / g);
    }
    return 0;
}
