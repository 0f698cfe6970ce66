// cooperative-search_amd/csrc/policy.hip -- fused recurrent-policy forward for the batched collector (gfx950).
//
// "Next" row f3 of SURVEY.md section 8(f): the reference picks actions one agent at a time with a batch-1 forward
// of its RNN (agent/agent.py:33-75, network/base_net.py:5-46: fc1 -> ReLU -> GRUCell -> fc2[Linear, ReLU, Linear])
// and an epsilon-greedy choice on the host.  Here ONE launch does, for all R = B*n (env, agent) rows:
//     x  = [obs(4) | one_hot(last_action)(A) | one_hot(agent_id)(n)]        (agent.py:41-52)
//     h1 = relu(W1 x + b1);  h' = GRUCell(h1, h);  q = W3 relu(W2 h' + b2) + b3
//     action = argmax_a q  (greedy)  or uniform over actions with probability epsilon
// The GEMM-shaped products run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products and sums,
// 32 cycles per 16x16x4 block).  Non-conv (flight_easy) networks only; flight's conv front end stays in torch.
//
// Decomposition (measured alternatives are listed in DESIGN.md section 8): a BLOCK of 4 wavefronts owns a tile of 16
// rows and wavefront w computes column tile w (16 of the 64 hidden columns) of every layer, 120 MFMAs per tile:
//   * its 120 weight fragments (one float per lane each, pre-packed on the host in lane order) are loaded ONCE into
//     registers; the grid is persistent (as many blocks as the device holds) and loops over row tiles;
//   * activations move from the C/D layout (col = lane&15, row = 4*(lane>>4)+reg) to the A layout (row = lane&15,
//     k = lane>>4) through block-shared LDS, one barrier per layer;
//   * the next tile's inputs are fetched into registers while the current one computes, and nothing inside the loop
//     issues a global LOAD other than that prefetch (loads and stores share one in-order counter on this hardware:
//     waiting for any load also waits for every store in flight);
//   * the six GRU accumulator chains are interleaved so that consecutive MFMAs are independent (a dependent pair
//     costs 40 cycles instead of 32); the gate nonlinearities use the hardware exp2 / rcp units.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>

#include "coopsearch.h"

namespace {

constexpr int H = 64;            // rnn_hidden_dim of the reference (common/arguments.py:58)
constexpr int KIN = 16;          // padded input width (4 + n_actions + n_agents <= 16)
constexpr int LDW = 68;          // LDS row stride in floats (68 % 32 = 4: 2-way conflicts at most on the A reads)
constexpr int PBLOCK = 256;      // 4 wavefronts, one 16-column tile each

using f32x4 = __attribute__((ext_vector_type(4))) float;

// packed weight fragments, in floats: fragment (column tile nt, k-step kk) holds, for lane l,
// W[16*nt + (l & 15)][4*kk + (l >> 4)] -- the B operand of one 16x16x4 MFMA
constexpr int FR = 64;
constexpr int OFF_W1 = 0;                           // [4 col tiles][4 k-steps][64]
constexpr int OFF_WIH = OFF_W1 + 4 * 4 * FR;        // [12][16][64]
constexpr int OFF_WHH = OFF_WIH + 12 * 16 * FR;     // [12][16][64]
constexpr int OFF_W2 = OFF_WHH + 12 * 16 * FR;      // [4][16][64]
constexpr int OFF_W3 = OFF_W2 + 4 * 16 * FR;        // [1][16][64]
constexpr int OFF_B1 = OFF_W3 + 16 * FR;            // 64
constexpr int OFF_BIH = OFF_B1 + 64;                // 192
constexpr int OFF_BHH = OFF_BIH + 192;              // 192
constexpr int OFF_B2 = OFF_BHH + 192;               // 64
constexpr int OFF_B3 = OFF_B2 + 64;                 // 16
constexpr int PACKED_FLOATS = OFF_B3 + 16;

struct PolicyParams {
    int rows, n_agents, n_actions, obs_stride, obs_offset;  // obs row r starts at obs + r*obs_stride + obs_offset (4 floats)
    float epsilon;
    unsigned long long seed;
    unsigned step;
    const float *w;          // packed
    const float *obs;
    const int64_t *last;     // [rows] last action index, < 0 = none (all-zero one-hot); null = raw input rows
    float *hidden;           // [rows][64] in/out
    float *q;                // [rows][n_actions] or null
    int64_t *actions;        // [rows]
};

// Gate nonlinearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each): absolute error ~1e-7 on
// outputs in [-1, 1], well inside the fp32 tolerance of the parity tests; the libm versions cost ~50 VALU
// instructions each, and VALU work competes with the co-resident block's MFMAs for the SIMD.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// splitmix64: per-row uniform for the epsilon-greedy choice (the reference draws from numpy's global stream on the
// host; any iid uniform source is equivalent)
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

#ifdef POL_TIMELINE
// debug build only: per-phase s_memtime stamps of thread 0 / block 0 (tools/exp_policy_timeline.py)
__device__ unsigned long long g_pstamps[32][8];
#define POL_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && iter < 32) g_pstamps[iter][k] = __builtin_readcyclecounter(); } while (0)
#else
#define POL_STAMP(k) do {} while (0)
#endif

// acc += A(16 x 4*KSTEPS, LDS row-major with stride LDW) * B(register-resident fragments)
template <int KSTEPS>
__device__ __forceinline__ f32x4 mfma_chain(const float *a_lds, const float (&bfrag)[KSTEPS], int lane, f32x4 acc) {
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
        const float a = a_lds[(lane & 15) * LDW + 4 * kk + (lane >> 4)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfrag[kk], acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(PBLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_policy(PolicyParams p) {
    __shared__ float s_a[16 * LDW];      // x, then h'
    __shared__ float s_b[16 * LDW];      // h1, then f
    __shared__ float s_h[16 * LDW];      // previous hidden state
    __shared__ float s_q[4][16 * 17];    // partial q of the four wavefronts
    __shared__ float s_b3[16];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // w is wave-uniform
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // every weight fragment this wavefront will use (120 floats per lane) and its biases, once
    float b1[KIN / 4], bg[6][16], b2[16], b3f[4];
    const unsigned ulane = lane;  // uniform base + 32-bit lane offset
#pragma unroll
    for (int kk = 0; kk < KIN / 4; kk++) b1[kk] = (p.w + OFF_W1 + (w * (KIN / 4) + kk) * FR)[ulane];
#pragma unroll
    for (int g = 0; g < 3; g++)   // torch.nn.GRUCell: gates ordered r, z, n in weight_ih / weight_hh
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            bg[2 * g][kk] = (p.w + OFF_WIH + ((w + 4 * g) * 16 + kk) * FR)[ulane];
            bg[2 * g + 1][kk] = (p.w + OFF_WHH + ((w + 4 * g) * 16 + kk) * FR)[ulane];
        }
#pragma unroll
    for (int kk = 0; kk < 16; kk++) b2[kk] = (p.w + OFF_W2 + (w * 16 + kk) * FR)[ulane];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) b3f[kk] = (p.w + OFF_W3 + (4 * w + kk) * FR)[ulane];
    if (threadIdx.x < 16) s_b3[threadIdx.x] = p.w[OFF_B3 + threadIdx.x];  // visible after the first barrier
    const int col = 16 * w + ccol;
    const float bias1 = p.w[OFF_B1 + col], bias2 = p.w[OFF_B2 + col];
    const float bir = p.w[OFF_BIH + col], biz = p.w[OFF_BIH + 64 + col], bin = p.w[OFF_BIH + 128 + col];
    const float bhr = p.w[OFF_BHH + col], bhz = p.w[OFF_BHH + 64 + col], bhn = p.w[OFF_BHH + 128 + col];

    // staging: 16 threads per row, one input column and four hidden values each
    const int tiles = (p.rows + 15) / 16;
    const int in_dim = 4 + p.n_actions + p.n_agents;
    const int srow = threadIdx.x >> 4, kcol = threadIdx.x & 15;
    const bool col_float = p.last ? kcol < 4 : kcol < in_dim, col_last = p.last && kcol >= 4 && kcol < 4 + p.n_actions;
    float xv;     // raw input value (obs column or caller-assembled row), where the column is a float input
    int lav;      // last action of the row, in the one-hot(last action) columns
    float4 hv;
    auto fetch = [&](int tile) {   // loads only: nothing here waits for memory
        const int row = 16 * tile + srow < p.rows ? 16 * tile + srow : p.rows - 1;
        xv = col_float ? p.obs[(size_t)row * p.obs_stride + p.obs_offset + kcol] : 0.0f;
        lav = col_last ? (int)p.last[row] : -1;
        hv = *reinterpret_cast<const float4 *>(p.hidden + (size_t)row * H + 4 * kcol);
    };
    auto input_value = [&](int row) {   // obs ++ one_hot(last action) ++ one_hot(agent id), agent.py:41-52
        if (col_float) return xv;
        if (col_last) return kcol - 4 == lav ? 1.0f : 0.0f;
        return (p.last && kcol < in_dim && kcol - 4 - p.n_actions == row % p.n_agents) ? 1.0f : 0.0f;
    };

    int tile = blockIdx.x;
    if (tile < tiles) fetch(tile);
    for (int iter = 0; tile < tiles; tile += gridDim.x, iter++) {
        const int row0 = 16 * tile;
        POL_STAMP(0);
        s_a[srow * LDW + kcol] = input_value(row0 + srow < p.rows ? row0 + srow : p.rows - 1);
        *reinterpret_cast<float4 *>(s_h + srow * LDW + 4 * kcol) = hv;
        if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
        __syncthreads();
        POL_STAMP(1);

        {   // h1 = relu(W1 x + b1), columns 16w..16w+15
            const f32x4 acc = mfma_chain<KIN / 4>(s_a, b1, lane, zero);
#pragma unroll
            for (int r = 0; r < 4; r++) s_b[(crow + r) * LDW + col] = fmaxf(acc[r] + bias1, 0.0f);
        }
        __syncthreads();
        POL_STAMP(2);

        {   // GRUCell, columns 16w..16w+15: six independent accumulator chains, interleaved
            f32x4 ir = zero, iz = zero, in_ = zero, hr = zero, hz = zero, hn_ = zero;
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                const float ax = s_b[(lane & 15) * LDW + 4 * kk + (lane >> 4)];
                const float ah = s_h[(lane & 15) * LDW + 4 * kk + (lane >> 4)];
                ir = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bg[0][kk], ir, 0, 0, 0);
                hr = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, bg[1][kk], hr, 0, 0, 0);
                iz = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bg[2][kk], iz, 0, 0, 0);
                hz = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, bg[3][kk], hz, 0, 0, 0);
                in_ = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bg[4][kk], in_, 0, 0, 0);
                hn_ = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, bg[5][kk], hn_, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float rg = sigmoidf_((ir[r] + bir) + (hr[r] + bhr));
                const float zg = sigmoidf_((iz[r] + biz) + (hz[r] + bhz));
                const float ng = tanhf_((in_[r] + bin) + rg * (hn_[r] + bhn));
                const float hnew = (1.0f - zg) * ng + zg * s_h[(crow + r) * LDW + col];
                s_a[(crow + r) * LDW + col] = hnew;  // s_a (x) was last read before the previous barrier
                if (row0 + crow + r < p.rows) p.hidden[(size_t)(row0 + crow + r) * H + col] = hnew;
            }
        }
        __syncthreads();
        POL_STAMP(3);

        {   // f = relu(W2 h' + b2), columns 16w..16w+15
            const f32x4 acc = mfma_chain<16>(s_a, b2, lane, zero);
#pragma unroll
            for (int r = 0; r < 4; r++) s_b[(crow + r) * LDW + col] = fmaxf(acc[r] + bias2, 0.0f);
        }
        __syncthreads();
        POL_STAMP(4);

        {   // q = W3 f + b3: the 64-long reduction is split four ways over the wavefronts
            const f32x4 acc = mfma_chain<4>(s_b + 16 * w, b3f, lane, zero);
#pragma unroll
            for (int r = 0; r < 4; r++) s_q[w][(crow + r) * 17 + ccol] = acc[r];
        }
        __syncthreads();
        POL_STAMP(5);

        // final sum, argmax and epsilon-greedy: one thread per row (wavefront 0 only; the others run ahead to the
        // next tile's staging)
        if (threadIdx.x < 16) {
            const int row = row0 + threadIdx.x;
            float best = -3.0e38f;
            int arg = 0;
            for (int a = 0; a < p.n_actions; a++) {
                const int o = threadIdx.x * 17 + a;
                const float qv = ((s_q[0][o] + s_q[1][o]) + (s_q[2][o] + s_q[3][o])) + s_b3[a];
                if (p.q && row < p.rows) p.q[(size_t)row * p.n_actions + a] = qv;
                if (qv > best) {   // strict: the first maximum wins, like torch.argmax
                    best = qv;
                    arg = a;
                }
            }
            if (row < p.rows) {
                int act = arg;
                if (p.epsilon > 0.0f) {
                    const unsigned long long h = mix64(p.seed ^ mix64(((unsigned long long)p.step << 32) | (unsigned)row));
                    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
                    if (u < p.epsilon) act = (int)((h & 0xffffffull) % (unsigned)p.n_actions);
                }
                p.actions[row] = act;
            }
        }
        POL_STAMP(6);
    }
}

thread_local char g_perr[200] = "";

// persistent grid: as many blocks as the device holds at once (queried once)
int resident_blocks() {
    static const int resident = [] {
        int dev = 0, cus = 256, per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess) return 512;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_policy, PBLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        return cus * per_cu;
    }();
    return resident;
}

}  // namespace

extern "C" {

size_t cs_policy_packed_floats(void) { return (size_t)PACKED_FLOATS; }

// Host-side packing of torch-layout weights (network/base_net.py parameter names) into MFMA B-fragment order.
// fc1_w [64][in_dim], w_ih / w_hh [192][64], fc2a_w [64][64], fc2b_w [n_actions][64]; biases alongside.
int cs_policy_pack(const float *fc1_w, const float *fc1_b, const float *w_ih, const float *b_ih, const float *w_hh,
                   const float *b_hh, const float *fc2a_w, const float *fc2a_b, const float *fc2b_w, const float *fc2b_b,
                   int in_dim, int n_actions, float *packed) {
    if (in_dim < 1 || in_dim > KIN || n_actions < 1 || n_actions > 16) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_pack: in_dim must be 1..16 and n_actions 1..16");
        return CS_E_ARG;
    }
    for (int i = 0; i < PACKED_FLOATS; i++) packed[i] = 0.0f;
    auto frag = [&](int off, int tiles, int ksteps, const float *w, int n_out, int k_in) {
        for (int nt = 0; nt < tiles; nt++)
            for (int kk = 0; kk < ksteps; kk++)
                for (int l = 0; l < 64; l++) {
                    const int n = 16 * nt + (l & 15), k = 4 * kk + (l >> 4);  // B[k][n] = W[n][k]
                    packed[off + (nt * ksteps + kk) * FR + l] = (n < n_out && k < k_in) ? w[(size_t)n * k_in + k] : 0.0f;
                }
    };
    frag(OFF_W1, 4, KIN / 4, fc1_w, 64, in_dim);
    frag(OFF_WIH, 12, 16, w_ih, 192, 64);
    frag(OFF_WHH, 12, 16, w_hh, 192, 64);
    frag(OFF_W2, 4, 16, fc2a_w, 64, 64);
    frag(OFF_W3, 1, 16, fc2b_w, n_actions, 64);
    for (int i = 0; i < 64; i++) packed[OFF_B1 + i] = fc1_b[i];
    for (int i = 0; i < 192; i++) packed[OFF_BIH + i] = b_ih[i];
    for (int i = 0; i < 192; i++) packed[OFF_BHH + i] = b_hh[i];
    for (int i = 0; i < 64; i++) packed[OFF_B2 + i] = fc2a_b[i];
    for (int i = 0; i < n_actions; i++) packed[OFF_B3 + i] = fc2b_b[i];
    return CS_OK;
}

// One forward of the shared agent network over `rows` = B*n rows (row r = env r / n_agents, agent r % n_agents).
// obs row r = 4 floats at obs_dev + r*obs_stride + obs_offset; last_dev[r] = last action (< 0: none);
// hidden_dev [rows][64] is updated in place; q_dev (nullable) [rows][n_actions]; actions_dev [rows] int64.
int cs_policy_forward(const float *packed_dev, const float *obs_dev, int obs_stride, int obs_offset, const int64_t *last_dev,
                      float *hidden_dev, float *q_dev, int64_t *actions_dev, int rows, int n_agents, int n_actions,
                      float epsilon, uint64_t seed, uint32_t step, void *stream) {
    if (!packed_dev || !obs_dev || !hidden_dev || !actions_dev || rows < 1 || n_agents < 1 || n_actions < 1 ||
        4 + n_actions + n_agents > KIN) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_forward: bad argument (need 4 + n_actions + n_agents <= 16)");
        return CS_E_ARG;
    }
    PolicyParams p{rows, n_agents, n_actions, obs_stride, obs_offset, epsilon, seed, step, packed_dev, obs_dev, last_dev,
                   hidden_dev, q_dev, actions_dev};
    const int tiles = (rows + 15) / 16, resident = resident_blocks();
    hipLaunchKernelGGL(k_policy, dim3(tiles < resident ? tiles : resident), dim3(PBLOCK), 0, (hipStream_t)stream, p);
    if (hipGetLastError() != hipSuccess) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_forward: kernel launch failed");
        return CS_E_LAUNCH;
    }
    return CS_OK;
}

const char *cs_policy_last_error(void) { return g_perr; }

#ifdef POL_TIMELINE
int cs_policy_debug_read_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * 32 * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
