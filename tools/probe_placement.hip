// Where do the two wavefronts of a 128-thread workgroup land?  A stand-in for k_rollout_od's launch shape (20 KB of LDS, 128 VGPRs,
// one resident round of 2048 workgroups): every wavefront records HW_REG_HW_ID / XCC_ID, the host prints how many first / second
// wavefronts each SIMD of each CU received.   hipcc --offload-arch=gfx950 -O2 tools/probe_placement.hip -o build/probe/placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include <vector>
__global__ __launch_bounds__(128) void probe(unsigned *out, int spin) {
    __shared__ unsigned pad[5000];
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pad[threadIdx.x] = hw;
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 0] = hw;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (pad[(threadIdx.x * 7) % 5000] == 0xdeadbeefu) out[0] = 1;
}
int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 2048;
    unsigned *d;
    hipMalloc(&d, wgs * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(128), 0, 0, d, 200);
    hipDeviceSynchronize();
    std::vector<unsigned> h(wgs * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx90a+: se_id [14:13]...)
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, std::pair<int, int>> cnt;   // (xcc, se, cu, simd) -> (first, second)
    for (int w = 0; w < wgs * 2; w++) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, se = (hw >> 13) & 7;
        auto &c = cnt[{xcc, se, cu, simd}];
        (w & 1) ? c.second++ : c.first++;
    }
    int hist[5][5] = {};
    for (auto &kv : cnt) hist[kv.second.first > 4 ? 4 : kv.second.first][kv.second.second > 4 ? 4 : kv.second.second]++;
    printf("%zu SIMDs seen; histogram of (first wavefronts, second wavefronts) per SIMD:\n", cnt.size());
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            if (hist[a][b]) printf("  %d first + %d second: %d SIMDs\n", a, b, hist[a][b]);
    // the first few workgroups: where their two wavefronts went
    for (int g = 0; g < 12 && g < wgs; g++)
        printf("wg %d: xcc %u se %u cu %u simd %u | simd %u\n", g, h[4 * g + 1] & 0xf, (h[4 * g] >> 13) & 7, (h[4 * g] >> 8) & 0xf, (h[4 * g] >> 4) & 3, (h[4 * g + 2] >> 4) & 3);
    return 0;
}
