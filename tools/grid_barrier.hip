// grid_barrier.hip -- what does a grid-wide barrier cost on MI355X?  (round-5 verdict item 4: a cooperative launch of the
// pre-sort / a radix pass is only worth building if the barrier between its phases stays below ~3 us.)
// G co-resident workgroups of T threads run K barriers back to back: arrive = one agent-scope atomic add by thread 0 (release),
// wait = spin on the counter with s_sleep until it reaches the phase's target (acquire), workgroup barrier on both sides.
// Every spin is BOUNDED (a flag is set and the kernel leaves when a phase does not complete): a grid that is not co-resident
// must not hang the GPU.   build: hipcc --offload-arch=gfx950 -O2 -o build_abl/grid_barrier tools/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, unsigned *fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; *fail = 1u; break; }
        }
    }
    __syncthreads();
    return ok;
}

// variant 2: relaxed polling with a longer sleep, one acquire fence at the end
__device__ __forceinline__ bool grid_barrier_v2(unsigned *counter, unsigned target, unsigned *fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 20)) { ok = false; *fail = 1u; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

// variant 3: two levels -- 16 group counters on their own 128-byte lines (workgroup b arrives at group b % 16), the last
// arriver of a group arrives at the top counter, the last one there publishes the phase in a word everybody polls
__device__ __forceinline__ bool grid_barrier_v3(unsigned *mem, unsigned phase, unsigned *fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned NG = 16u, g = blockIdx.x % NG;
        const unsigned members = (gridDim.x - g + NG - 1u) / NG, groups = gridDim.x < NG ? gridDim.x : NG;
        unsigned *grp = mem + 32u * (1u + g), *top = mem + 32u * 17u, *rel = mem;
        const unsigned a = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1u == members * phase) {
            const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1u == groups * phase) __hip_atomic_store(rel, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 21)) { ok = false; *fail = 1u; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

template <int V>
__global__ void barrier_loop_v(unsigned *mem, unsigned *fail, int K, float *sink) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        acc += (float)k;
        const bool ok = V == 2 ? grid_barrier_v2(mem, (unsigned)(k + 1) * gridDim.x, fail) : grid_barrier_v3(mem, (unsigned)(k + 1), fail);
        if (!ok) break;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = acc;
}

__global__ void barrier_loop(unsigned *counter, unsigned *fail, int K, float *sink) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        acc += (float)k; // (something between the barriers)
        if (!grid_barrier(counter, (unsigned)(k + 1) * gridDim.x, fail)) break;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = acc;
}

__global__ void empty_kernel(float *sink) {
    if (threadIdx.x == 1024) *sink = 0.f;
}

int main() {
    unsigned *counter, *fail;
    float *sink;
    hipMalloc(&counter, 4); hipMalloc(&fail, 4); hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int K = 200;
    for (int T : {256, 1024}) {
        for (int G : {64, 256, 512, 1024, 2048}) {
            if ((long)G * T > 256L * 2048) continue; // beyond the chip's resident threads
            float best = 1e9f;
            unsigned hfail = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(fail, 0, 4);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(barrier_loop, dim3(G), dim3(T), 0, 0, counter, fail, K, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost);
                if (hfail) break;
                if (ms < best) best = ms;
            }
            if (hfail) printf("G = %4d x %4d threads: NOT co-resident (bounded spin gave up)\n", G, T);
            else printf("G = %4d x %4d threads: %d barriers in %.3f ms -> %.2f us per barrier (incl. one launch of ~5 us over %d)\n", G, T, K, best,
                        1e3f * best / K, K);
        }
    }
    unsigned *mem;
    hipMalloc(&mem, 4096);
    for (int V : {2, 3}) {
        for (int G : {64, 256, 512, 1024}) {
            float best = 1e9f;
            unsigned hfail = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(mem, 0, 4096); hipMemset(fail, 0, 4);
                hipEventRecord(e0, 0);
                if (V == 2) hipLaunchKernelGGL(barrier_loop_v<2>, dim3(G), dim3(256), 0, 0, mem, fail, K, sink);
                else hipLaunchKernelGGL(barrier_loop_v<3>, dim3(G), dim3(256), 0, 0, mem, fail, K, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost);
                if (hfail) break;
                if (ms < best) best = ms;
            }
            if (hfail) printf("variant %d, G = %4d x 256: bounded spin gave up\n", V, G);
            else printf("variant %d, G = %4d x 256 threads: %.2f us per barrier\n", V, G, 1e3f * best / K);
        }
    }
    // the launch floor for comparison: back-to-back empty kernels
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("200 empty launches back to back: %.2f us each\n", 1e3f * ms / 200);
    return 0;
}
