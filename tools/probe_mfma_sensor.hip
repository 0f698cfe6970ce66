// The sensor tests of the octet detection pass on the matrix pipe against the VALU form: one wavefront = 8 envs x 8 lanes,
// 5 agents and 16 target slots per env, positions out of a four-slot LDS ring like D's.  Per wavefront-step
//   valu : every lane tests its two targets against the 5 agents in fp64 (2 sub, 2 mul, add, compare: what oct_detect_impl does) -> 10
//          lane masks, their population counts summed on the scalar side;
//   mfma : d2 = |a|^2 + |t|^2 - 2 a.t for 16 agent slots x 16 targets of ONE env per v_mfma_f64_16x16x4_f64 (A rows (x, y, |a|^2, 1) read
//          from the ring by a per-lane address -- K would publish them: one ds_read_b64 per env, no VALU --, B columns (-2tx, -2ty, 1, |t|^2) resident: one double per lane and env), 8 MFMAs per
//          wavefront-step, two compares per env against the threshold (agents 0-3: register 0, agent 4: register 1, lanes 0-15) and two
//          band compares (|d2 - thr| <= 1e-9: the pairs an exact pass would have to redo) -> the env's agent-major pair mask as a scalar.
// Both kernels do ONLY this stage, ITER times, with W wavefronts per SIMD (W x 1024 workgroups of 64 threads, all resident); the host
// reports ns and cycles (2.4 GHz) per wavefront-step of a SIMD's W wavefronts, and checks that the two forms count the same pairs in range
// (pairs inside the band reported separately).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_mfma_sensor.hip -o build/probe/mfma_sensor && build/probe/mfma_sensor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ENVS = 8, NA = 5, RING = 4, AP = 9;
typedef double v4d __attribute__((ext_vector_type(4)));

struct Shared {
    double2 pos[RING][ENVS][AP];   // agent positions of step (it & 3), env g, agent i (columns 5..8: padding agents far outside the map)
    double row4[RING][ENVS][AP][4];  // the same agents as A-operand rows (x, y, x^2 + y^2, 1): what K would publish for the matrix form
};

__device__ __forceinline__ void fill(Shared &sh, int lane, unsigned seed) {
    // deterministic pseudo-random positions in [0, 50); padding agents at 1e4
    for (int k = lane; k < RING * ENVS * AP; k += 64) {
        const int i = k % AP;
        unsigned h = (unsigned)k * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        unsigned h2 = h * 3266489917u; h2 ^= h2 >> 16;
        const double x = i < NA ? (h & 0xffffff) * (50.0 / 16777216.0) : 1.0e4, y = i < NA ? (h2 & 0xffffff) * (50.0 / 16777216.0) : 1.0e4;
        (&sh.pos[0][0][0])[k] = make_double2(x, y);
        double *r4 = &sh.row4[0][0][0][0] + 4 * k;
        r4[0] = x; r4[1] = y; r4[2] = x * x + y * y; r4[3] = 1.0;
    }
}
__device__ __forceinline__ double2 target_of(int g, int j, unsigned seed) {
    unsigned h = (unsigned)(g * 16 + j) * 40503u ^ (seed * 2246822519u);
    h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
    unsigned h2 = h * 3266489917u; h2 ^= h2 >> 16;
    return make_double2((h & 0xffffff) * (50.0 / 16777216.0), (h2 & 0xffffff) * (50.0 / 16777216.0));
}

__global__ __launch_bounds__(64) void k_valu(int iters, double thr, unsigned long long *out) {
    __shared__ Shared sh;
    const int lane = threadIdx.x, o = lane >> 3, t = lane & 7;
    const unsigned seed = blockIdx.x + 1;
    fill(sh, lane, seed);
    __syncthreads();
    const double2 t0 = target_of(o, t, seed), t1 = target_of(o, t + 8, seed);
    unsigned long long total = 0;
    for (int it = 0; it < iters; it++) {
        const double2(*pos)[AP] = sh.pos[it & (RING - 1)];
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const double2 a = pos[o][i];
            const double dx0 = t0.x - a.x, dy0 = t0.y - a.y, dx1 = t1.x - a.x, dy1 = t1.y - a.y;
            cnt += __popcll(__ballot(dx0 * dx0 + dy0 * dy0 <= thr)) + __popcll(__ballot(dx1 * dx1 + dy1 * dy1 <= thr));
        }
        total += (unsigned long long)cnt;
        asm volatile("" ::: "memory");
    }
    if (lane == 0) out[blockIdx.x] = total;
}

__global__ __launch_bounds__(64) void k_mfma(int iters, double thr, unsigned long long *out, unsigned long long *band_out) {
    __shared__ Shared sh;
    const int lane = threadIdx.x;
    const unsigned seed = blockIdx.x + 1;
    fill(sh, lane, seed);
    __syncthreads();
    const int j = lane & 15, kk = lane >> 4;   // B[k][j]: column j = target, row k of (-2tx, -2ty, 1, |t|^2); A[i][k]: i = lane & 15 = agent slot
    double b[ENVS];
#pragma unroll
    for (int g = 0; g < ENVS; g++) {
        const double2 tg = target_of(g, j, seed);
        b[g] = kk == 0 ? -2.0 * tg.x : (kk == 1 ? -2.0 * tg.y : (kk == 2 ? 1.0 : tg.x * tg.x + tg.y * tg.y));
    }
    const int ai = (lane & 15) < AP ? (lane & 15) : AP - 1;   // agent slots 9..15 repeat the last padding agent
    unsigned long long total = 0, band = 0;
    for (int it = 0; it < iters; it++) {
        const double *rows = &sh.row4[it & (RING - 1)][0][0][0] + 4 * ai + kk;   // this lane's component of its agent slot: ONE ds_read_b64 per env
        int cnt = 0, bcnt = 0;
#pragma unroll
        for (int g = 0; g < ENVS; g++) {
            const double av = rows[g * AP * 4];
            v4d c = {0.0, 0.0, 0.0, 0.0};
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[g], c, 0, 0, 0);
            // rows (lane >> 4) + 4 r: register 0 = agents 0..3, register 1 = agents 4..7 (agent 4 in lanes 0..15)
            const unsigned long long m0 = __ballot(c[0] <= thr), m1 = __ballot(c[1] <= thr) & 0xffffull;
            const unsigned long long f0 = __ballot(__builtin_fabs(c[0] - thr) <= 1.0e-9), f1 = __ballot(__builtin_fabs(c[1] - thr) <= 1.0e-9) & 0xffffull;
            cnt += __popcll(m0) + __popcll(m1);
            bcnt += __popcll(f0) + __popcll(f1);
        }
        total += (unsigned long long)cnt;
        band += (unsigned long long)bcnt;
        asm volatile("" ::: "memory");
    }
    if (lane == 0) {
        out[blockIdx.x] = total;
        band_out[blockIdx.x] = band;
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const double thr = 49.0;   // view_range 7
    printf("sensor stage of one detection wavefront (8 envs x 5 agents x 16 target slots), %d steps per launch\n", iters);
    for (int W : {1, 2, 4}) {
        const int wgs = 1024 * W;
        unsigned long long *o1, *o2, *o3;
        hipMalloc(&o1, wgs * 8); hipMalloc(&o2, wgs * 8); hipMalloc(&o3, wgs * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms_v = 0, ms_m = 0;
        for (int rep = 0; rep < 3; rep++) {   // last repetition counts
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_valu, dim3(wgs), dim3(64), 0, 0, iters, thr, o1);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_v, e0, e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(64), 0, 0, iters, thr, o2, o3);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_m, e0, e1);
        }
        std::vector<unsigned long long> h1(wgs), h2(wgs), h3(wgs);
        hipMemcpy(h1.data(), o1, wgs * 8, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), o2, wgs * 8, hipMemcpyDeviceToHost);
        hipMemcpy(h3.data(), o3, wgs * 8, hipMemcpyDeviceToHost);
        unsigned long long pairs_v = 0, pairs_m = 0, band = 0, differ = 0;
        for (int k = 0; k < wgs; k++) { pairs_v += h1[k]; pairs_m += h2[k]; band += h3[k]; differ += h1[k] != h2[k]; }
        const double ns_v = ms_v * 1e6 / iters, ns_m = ms_m * 1e6 / iters;
        printf("W = %d wavefronts per SIMD: valu %.1f ns = %.0f cycles per step of a SIMD's %d wavefronts (%.0f per wavefront-step) | mfma %.1f ns = %.0f cycles (%.0f) | "
               "pairs in range %llu / %llu, workgroups whose counts differ %llu, pairs inside the 1e-9 band %llu\n",
               W, ns_v, ns_v * 2.4, W, ns_v * 2.4 / W, ns_m, ns_m * 2.4, ns_m * 2.4 / W, pairs_v, pairs_m, differ, band);
        hipFree(o1); hipFree(o2); hipFree(o3);
    }
    return 0;
}
