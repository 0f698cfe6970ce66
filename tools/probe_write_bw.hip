// What the part sustains for a pure WRITE stream, a pure read stream and a copy (the rollout kernels at HBM-sized batches write 93 % of
// their bytes: obs + state, 340 of the 366 algorithmic bytes per env-step at 5 agents): float4 per lane, fully coalesced 1 KB per
// wavefront-instruction, grid-stride over a buffer far larger than the 256 MiB Infinity Cache; non-temporal and plain stores.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_write_bw.hip -o build/probe/write_bw && build/probe/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_write(v4f *dst, size_t n4, float seed) {
    const v4f v = {seed, seed + 1.f, seed + 2.f, seed + 3.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_read(const v4f *src, size_t n4, float *sink) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4f *src, v4f *dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const v4f v = src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
template <class F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; r++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 4096) << 20;   // MiB
    const size_t n4 = bytes / 16;
    v4f *a, *b;
    float *sink;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&sink, 4);
    (void)hipMemset(a, 0, bytes);
    printf("buffer %zu MiB per stream; GB/s = bytes moved / time (copy: read + written)\n", bytes >> 20);
    for (int wgs_per_cu : {2, 4, 8, 16}) {
        const int grid = 256 * wgs_per_cu;
        const double w_nt = time_ms([&] { hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, b, n4, 1.f); }, 5);
        const double w_pl = time_ms([&] { hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, b, n4, 2.f); }, 5);
        const double rd = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n4, sink); }, 5);
        const double cp_nt = time_ms([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, a, b, n4); }, 5);
        const double cp_pl = time_ms([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, a, b, n4); }, 5);
        printf("%2d workgroups of 256 per CU: write nt %.0f GB/s | write plain %.0f | read %.0f | copy nt %.0f | copy plain %.0f\n", wgs_per_cu,
               bytes / w_nt / 1e6, bytes / w_pl / 1e6, bytes / rd / 1e6, 2.0 * bytes / cp_nt / 1e6, 2.0 * bytes / cp_pl / 1e6);
    }
    return 0;
}
