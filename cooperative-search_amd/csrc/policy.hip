// cooperative-search_amd/csrc/policy.hip -- fused recurrent-policy forward for the batched collector (gfx950).
//
// "Next" row f3 of SURVEY.md section 8(f): the reference picks actions one agent at a time with a batch-1 forward
// of its RNN (agent/agent.py:33-75, network/base_net.py:5-46: fc1 -> ReLU -> GRUCell -> fc2[Linear, ReLU, Linear])
// and an epsilon-greedy choice on the host.  Here ONE launch does, for all R = B*n (env, agent) rows:
//     x  = [obs(4) | one_hot(last_action)(A) | one_hot(agent_id)(n)]        (agent.py:41-52)
//     h1 = relu(W1 x + b1);  h' = GRUCell(h1, h);  q = W3 relu(W2 h' + b2) + b3
//     action = argmax_a q  (greedy)  or uniform over actions with probability epsilon
// The GEMM-shaped products run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products and sums,
// 32 cycles per 16x16x4 block).  flight adds k_conv_features (below) for the conv front end of its network.
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

#include "policy_dev.h"

struct PolicyParams {
    int rows, n_agents, n_actions, obs_stride, obs_offset;  // obs row r starts at obs + r*obs_stride + obs_offset (4 floats)
    float epsilon;
    const double *eps_env;   // null, or [rows / n_agents]: per-env epsilon (cs_epsilon schedule) instead of `epsilon`
    unsigned long long seed;
    unsigned step;
    unsigned long long row0; // global index of row 0 (sharded batches: env_offset * n_agents)
    int select;              // CS_SELECT_*
    const float *w;          // packed
    const float *obs;
    const int64_t *last;     // [rows] last action index, < 0 = none (all-zero one-hot); null = raw input rows
    const float *feat;       // [rows / rows_per_feat][16] conv features placed in front of the obs columns, or null
    int rows_per_feat;
    float *hidden;           // [rows][64] in/out
    float *q;                // [rows][n_actions] or null
    int64_t *actions;        // [rows]
};

#ifdef POL_TIMELINE
// debug build only: per-phase s_memtime stamps of thread 0 / block 0 (read by tools/exp_policy_timeline.py of round 4: git history)
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

template <int KS1>   // k-steps of fc1: 4 (inputs <= 16 wide) or 8 (<= 32 wide: conv features in front)
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
    float b1[KS1], bg[6][16], b2[16], b3f[4];
    const unsigned ulane = lane;  // uniform base + 32-bit lane offset
#pragma unroll
    for (int kk = 0; kk < KS1; kk++) b1[kk] = (p.w + OFF_W1 + (w * (KIN_MAX / 4) + kk) * FR)[ulane];
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

    // staging: 16 threads per row, NCOL input columns and four hidden values each.  Input row (agent.py:41-52,
    // base_net.py:31-39): [16 conv features |] obs(4) | one_hot(last action) | one_hot(agent id); raw mode (no `last`):
    // [16 conv features |] the caller's own columns.
    constexpr int NCOL = KS1 / 4;
    const int tiles = (p.rows + 15) / 16;
    const int fbase = p.feat ? NFEAT : 0, in_dim = fbase + 4 + p.n_actions + p.n_agents;
    const int srow = threadIdx.x >> 4, kcol = threadIdx.x & 15;
    enum { ZERO, OBS, FEAT, LAST, AGENT };
    int kind[NCOL], kidx[NCOL];
#pragma unroll
    for (int j = 0; j < NCOL; j++) {
        const int k = kcol + 16 * j, ko = k - fbase;
        kind[j] = k >= in_dim ? ZERO : k < fbase ? FEAT : (!p.last || ko < 4) ? OBS : ko < 4 + p.n_actions ? LAST : AGENT;
        kidx[j] = kind[j] == FEAT ? k : kind[j] == OBS ? ko : kind[j] == LAST ? ko - 4 : ko - 4 - p.n_actions;
    }
    float xv[NCOL];   // float inputs (obs / feature columns)
    int lav[NCOL];    // last action of the row (one-hot(last action) columns)
    float4 hv;
    auto fetch = [&](int tile) {   // loads only: nothing here waits for memory
        const int row = 16 * tile + srow < p.rows ? 16 * tile + srow : p.rows - 1;
#pragma unroll
        for (int j = 0; j < NCOL; j++) {
            xv[j] = kind[j] == OBS ? p.obs[(size_t)row * p.obs_stride + p.obs_offset + kidx[j]]
                  : kind[j] == FEAT ? p.feat[(size_t)(row / p.rows_per_feat) * NFEAT + kidx[j]] : 0.0f;
            lav[j] = kind[j] == LAST ? (int)p.last[row] : -1;
        }
        hv = *reinterpret_cast<const float4 *>(p.hidden + (size_t)row * H + 4 * kcol);
    };
    auto input_value = [&](int j, int row) {
        if (kind[j] == OBS || kind[j] == FEAT) return xv[j];
        if (kind[j] == LAST) return kidx[j] == lav[j] ? 1.0f : 0.0f;
        return (kind[j] == AGENT && kidx[j] == row % p.n_agents) ? 1.0f : 0.0f;
    };

    int tile = blockIdx.x;
    if (tile < tiles) fetch(tile);
    for (int iter = 0; tile < tiles; tile += gridDim.x, iter++) {
        const int row0 = 16 * tile;
        POL_STAMP(0);
#pragma unroll
        for (int j = 0; j < NCOL; j++)
            s_a[srow * LDW + kcol + 16 * j] = input_value(j, row0 + srow < p.rows ? row0 + srow : p.rows - 1);
        *reinterpret_cast<float4 *>(s_h + srow * LDW + 4 * kcol) = hv;
        if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
        __syncthreads();
        POL_STAMP(1);

        {   // h1 = relu(W1 x + b1), columns 16w..16w+15
            const f32x4 acc = mfma_chain<KS1>(s_a, b1, lane, zero);
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
            auto qf = [&](int a) {
                const int o = threadIdx.x * 17 + a;
                return ((s_q[0][o] + s_q[1][o]) + (s_q[2][o] + s_q[3][o])) + s_b3[a];
            };
            if (p.q && row < p.rows)
                for (int a = 0; a < p.n_actions; a++) p.q[(size_t)row * p.n_actions + a] = qf(a);
            const float eps = p.eps_env ? (float)p.eps_env[(row < p.rows ? row : p.rows - 1) / p.n_agents] : p.epsilon;
            const int act = select_action(qf, p.n_actions, p.select, eps, p.seed, p.step, p.row0 + (unsigned long long)row);
            if (row < p.rows) p.actions[row] = act;
        }
        POL_STAMP(6);
    }
}


#if CS_POLICY_F16
// k_policy on the 16-bit matrix pipe with split-fp16 operands (policy_dev.h): same decomposition -- a block of 4 wavefronts owns a
// 16-row tile, wavefront w the hidden columns 16w..16w+15 of every layer, weights resident in registers (32 B fragments: 128
// VGPRs), activations through LDS as (hi, lo) plane pairs, one barrier per layer, the next tile's inputs prefetched -- with 48
// matrix instructions of ~17 cycles per tile and wavefront instead of 120 of 32.  The fc1 input is padded to 32 columns (one k-step)
// whether or not the conv features are present.
__global__ __launch_bounds__(PBLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_policy_h(PolicyParams p) {
    __shared__ __attribute__((aligned(16))) _Float16 s_a[2][16 * HST];   // x, then h'      (hi, lo planes)
    __shared__ __attribute__((aligned(16))) _Float16 s_b[2][16 * HST];   // h1, then f
    __shared__ __attribute__((aligned(16))) _Float16 s_hs[2][16 * HST];  // previous hidden state, split
    __shared__ float s_h[16 * LDW];      // previous hidden state, fp32 (the GRU blend)
    __shared__ float s_q[4][16 * 17];    // partial q of the four wavefronts
    __shared__ float s_b3[16];
    __shared__ uint4 s_w3[2 * 2 * 64];   // the fc3 fragments ([k-step][hi, lo][lane]): only wavefront 0 multiplies by them, once per tile --
    //                                      in registers they cost every wavefront 16 VGPRs, and the kernel sits at its 256-register limit
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // w is wave-uniform
    const int crow = (lane >> 4) * 4, ccol = lane & 15;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const unsigned ulane = lane;

    // every weight fragment this wavefront will use and its biases, once
    const BFrag b1 = load_bfrag(p.w, HOFF_W1, w, ulane);
    BFrag bg[6][2];
#pragma unroll
    for (int g = 0; g < 3; g++)   // torch.nn.GRUCell: gates ordered r, z, n in weight_ih / weight_hh
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bg[2 * g][ks] = load_bfrag(p.w, HOFF_WIH, (w + 4 * g) * 2 + ks, ulane);
            bg[2 * g + 1][ks] = load_bfrag(p.w, HOFF_WHH, (w + 4 * g) * 2 + ks, ulane);
        }
    BFrag b2[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) b2[ks] = load_bfrag(p.w, HOFF_W2, w * 2 + ks, ulane);
    s_w3[threadIdx.x] = reinterpret_cast<const uint4 *>(p.w + HOFF_W3)[threadIdx.x];   // 256 threads x 16 B = both fragments (visible after the first barrier)
    if (threadIdx.x < 16) s_b3[threadIdx.x] = p.w[HOFF_B3 + threadIdx.x];  // visible after the first barrier
    const int col = 16 * w + ccol;
    const float bias1 = p.w[HOFF_B1 + col], bias2 = p.w[HOFF_B2 + col];
    const float bir = p.w[HOFF_BIH + col], biz = p.w[HOFF_BIH + 64 + col], bin = p.w[HOFF_BIH + 128 + col];
    const float bhr = p.w[HOFF_BHH + col], bhz = p.w[HOFF_BHH + 64 + col], bhn = p.w[HOFF_BHH + 128 + col];
    const float b_r = bir + bhr, b_z = biz + bhz;   // the r and z gates run as one chain over [x | h] with one bias (gru_cell)

    // staging: 16 threads per row, two input columns (k, k + 16) and four hidden values each.  Input row (agent.py:41-52,
    // base_net.py:31-39): [16 conv features |] obs(4) | one_hot(last action) | one_hot(agent id), zero up to column 32; raw mode
    // (no `last`): [16 conv features |] the caller's own columns.
    constexpr int NCOL = 2;
    const int tiles = (p.rows + 15) / 16;
    const int fbase = p.feat ? NFEAT : 0, in_dim = fbase + 4 + p.n_actions + p.n_agents;
    const int srow = threadIdx.x >> 4, kcol = threadIdx.x & 15;
    enum { ZERO, OBS, FEAT, LAST, AGENT };
    int kind[NCOL], kidx[NCOL];
#pragma unroll
    for (int j = 0; j < NCOL; j++) {
        const int k = kcol + 16 * j, ko = k - fbase;
        kind[j] = k >= in_dim ? ZERO : k < fbase ? FEAT : (!p.last || ko < 4) ? OBS : ko < 4 + p.n_actions ? LAST : AGENT;
        kidx[j] = kind[j] == FEAT ? k : kind[j] == OBS ? ko : kind[j] == LAST ? ko - 4 : ko - 4 - p.n_actions;
    }
    // the inputs of the CURRENT tile (xv, lav, hv) and of the NEXT one (n*): the next tile's loads are issued right after the
    // current tile has been staged and are first touched at the very end of the loop body (the empty asm below pins that use
    // there), four phases later.  With one set of variables the compiler copied the freshly loaded values into the loop-carried
    // registers straight after issuing the loads and waited for them in front of the first barrier: a full memory round trip per tile.
    float xv[NCOL], nxv[NCOL];
    int lav[NCOL], nlav[NCOL];
    float4 hv, nhv;
    auto fetch = [&](int tile, float (&fx)[NCOL], int (&fl)[NCOL], float4 &fh) {   // loads only: nothing here waits for memory
        const int row = 16 * tile + srow < p.rows ? 16 * tile + srow : p.rows - 1;
#pragma unroll
        for (int j = 0; j < NCOL; j++) {
            fx[j] = kind[j] == OBS ? p.obs[(size_t)row * p.obs_stride + p.obs_offset + kidx[j]]
                  : kind[j] == FEAT ? p.feat[(size_t)(row / p.rows_per_feat) * NFEAT + kidx[j]] : 0.0f;
            fl[j] = kind[j] == LAST ? (int)p.last[row] : -1;
        }
        fh = *reinterpret_cast<const float4 *>(p.hidden + (size_t)row * H + 4 * kcol);
    };
    auto input_value = [&](int j, int row) {
        if (kind[j] == OBS || kind[j] == FEAT) return xv[j];
        if (kind[j] == LAST) return kidx[j] == lav[j] ? 1.0f : 0.0f;
        return (kind[j] == AGENT && kidx[j] == row % p.n_agents) ? 1.0f : 0.0f;
    };

    int tile = blockIdx.x;
    if (tile < tiles) fetch(tile, xv, lav, hv);
#pragma unroll
    for (int j = 0; j < NCOL; j++) {
        nxv[j] = 0.0f;
        nlav[j] = -1;
    }
    nhv = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iter = 0; tile < tiles; tile += gridDim.x, iter++) {
        const int row0 = 16 * tile;
        POL_STAMP(0);
#pragma unroll
        for (int j = 0; j < NCOL; j++)
            split_store(s_a[0], s_a[1], srow * HST + kcol + 16 * j, input_value(j, row0 + srow < p.rows ? row0 + srow : p.rows - 1));
        *reinterpret_cast<float4 *>(s_h + srow * LDW + 4 * kcol) = hv;
        split_store(s_hs[0], s_hs[1], srow * HST + 4 * kcol + 0, hv.x);
        split_store(s_hs[0], s_hs[1], srow * HST + 4 * kcol + 1, hv.y);
        split_store(s_hs[0], s_hs[1], srow * HST + 4 * kcol + 2, hv.z);
        split_store(s_hs[0], s_hs[1], srow * HST + 4 * kcol + 3, hv.w);
        const bool more = tile + (int)gridDim.x < tiles;   // block-uniform
        if (more) fetch(tile + gridDim.x, nxv, nlav, nhv);
        __syncthreads();
        POL_STAMP(1);

        {   // h1 = relu(W1 x + b1), columns 16w..16w+15 (the bias enters the accumulator)
            f32x4 hi = splat4(bias1), lo = zero;
            h8 ah, al;
            load_afrag(s_a[0], s_a[1], 0, 0, lane, ah, al);
            mfma_split(ah, al, b1, hi, lo);
#pragma unroll
            for (int r = 0; r < 4; r++) split_store(s_b[0], s_b[1], (crow + r) * HST + col, fmaxf(split_sum(hi[r], lo[r]), 0.0f));
        }
        __syncthreads();
        POL_STAMP(2);

        {   // GRUCell, columns 16w..16w+15 (gru_products / gru_cell of policy_dev.h: the very code the fused closed loop runs)
            h8 xh[2], xl[2], hh[2], hl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                load_afrag(s_b[0], s_b[1], 0, ks, lane, xh[ks], xl[ks]);
                load_afrag(s_hs[0], s_hs[1], 0, ks, lane, hh[ks], hl[ks]);
            }
            GruAcc acc;
            gru_products(xh, xl, hh, hl, bg, b_r, b_z, bin, bhn, acc);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float hnew = gru_cell(acc, r, s_h[(crow + r) * LDW + col]);
                split_store(s_a[0], s_a[1], (crow + r) * HST + col, hnew);  // s_a (x) was last read before the previous barrier
                if (row0 + crow + r < p.rows) p.hidden[(size_t)(row0 + crow + r) * H + col] = hnew;
            }
        }
        __syncthreads();
        POL_STAMP(3);

        {   // f = relu(W2 h' + b2), columns 16w..16w+15
            f32x4 hi = splat4(bias2), lo = zero;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                h8 ah, al;
                load_afrag(s_a[0], s_a[1], 0, ks, lane, ah, al);
                mfma_split(ah, al, b2[ks], hi, lo);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) split_store(s_b[0], s_b[1], (crow + r) * HST + col, fmaxf(split_sum(hi[r], lo[r]), 0.0f));
        }
        __syncthreads();
        POL_STAMP(4);

        // q = W3 f + b3 and the choice, by ONE wavefront (no K split, no exchange of partial sums, no barrier: the other wavefronts go
        // on to the next tile's staging, which touches neither s_b nor s_q; wavefront 0 joins the next barrier when it is done)
        if (w == 0) {
            f32x4 hi = splat4(s_b3[ccol]), lo = zero;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                h8 ah, al;
                load_afrag(s_b[0], s_b[1], 0, ks, lane, ah, al);
                BFrag b3;
                const uint4 wh = s_w3[(2 * ks) * 64 + lane], wl = s_w3[(2 * ks + 1) * 64 + lane];
                __builtin_memcpy(&b3.hi, &wh, 16);
                __builtin_memcpy(&b3.lo, &wl, 16);
                mfma_split(ah, al, b3, hi, lo);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) s_q[0][(crow + r) * 17 + ccol] = split_sum(hi[r], lo[r]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            POL_STAMP(5);
            // argmax and epsilon-greedy: one lane per row
            if (lane < 16) {
                const int row = row0 + lane;
                auto qf = [&](int a) { return s_q[0][lane * 17 + a]; };
                if (p.q && row < p.rows)
                    for (int a = 0; a < p.n_actions; a++) p.q[(size_t)row * p.n_actions + a] = qf(a);
                const float eps = p.eps_env ? (float)p.eps_env[(row < p.rows ? row : p.rows - 1) / p.n_agents] : p.epsilon;
                const int act = select_action(qf, p.n_actions, p.select, eps, p.seed, p.step, p.row0 + (unsigned long long)row);
                if (row < p.rows) p.actions[row] = act;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // s_q[0] is free again (this wavefront's next tile)
        }
        POL_STAMP(6);
        // first use of the next tile's inputs: here, not earlier (see above)
#pragma unroll
        for (int j = 0; j < NCOL; j++) {
            asm volatile("" : "+v"(nxv[j]), "+v"(nlav[j]));
            xv[j] = nxv[j];
            lav[j] = nlav[j];
        }
        asm volatile("" : "+v"(nhv.x), "+v"(nhv.y), "+v"(nhv.z), "+v"(nhv.w));
        hv = nhv;
    }
}
#endif   // CS_POLICY_F16

// ---- flight: conv front end of the agent network (network/base_net.py:9-18,31-36) ---------------------------------
// Conv2d(1, 4, k=4, s=2) -> ReLU -> Conv2d(4, 1, k=3, s=1, p=1) -> ReLU -> Linear(576, 16) on the 50 x 50 probability
// map (the reference's flight hyper-parameters, common/arguments.py:256-265; anything else stays on the torch path).
// One block per map: the map and both conv outputs live in LDS (23 KB), the 16 x 576 linear weights in registers (36
// per thread); persistent blocks loop over maps.  All agents of an env observe the same map (flight_env.py:223-230), so the
// batched caller runs this once per ENV and k_policy fans the 16 features out to the env's rows.
constexpr int MAPW = 50, C1W = 24, C1P = 26, C1CH = 4, NPOS = C1W * C1W;

struct ConvParams {
    const float *c1w, *c1b, *c2w, *c2b, *lw, *lb;  // torch layouts: [4][1][4][4], [4], [1][4][3][3], [1], [16][576], [16]
    const float *maps;                             // map m at maps + m * map_stride, 2500 floats ([50][50] row-major)
    long long map_stride;
    int n_maps, vec4;                              // vec4: every map is 16-byte aligned
    float *feat;                                   // [n_maps][16]
};

// Round 3: register-tiled.  The first version computed one output position per thread and pass and read every tap from LDS
// on its own: 16 ds_read_b32 per conv1 position, 36 per conv2 position, 36 strided ones per thread in the linear layer --
// ~40 000 LDS instructions per map against ~33 000 packed FMAs, LDS-issue bound at 25 TFLOP/s.  Now a thread owns a STRIP of
// three horizontally adjacent outputs (24 x 8 = 192 strips: one pass) and fetches the strip's input window in wide reads
// (conv1: 4 rows x 8 floats as float2; conv2: conv1's channels stored as two planes of PAIRS, {c0, c2} and {c1, c3}, so one
// ds_read_b64 brings the pair a packed FMA consumes -- interleaving all four per position put the 64 lanes of a read on 8 bank
// groups, 62 % of the LDS's busy cycles were bank conflicts); conv2's output is stored transposed ((pos % 16) * 36 + pos / 16)
// so that the linear layer's 36 terms of a thread are contiguous (9 float4).  Every output's FMA chain -- taps in the same
// order, the same channel pairing, the same butterfly -- is unchanged: the features are bit-identical to the first version's.
// four workgroups per CU (<= 128 VGPRs)
__global__ __launch_bounds__(PBLOCK, 4) void k_conv_features(ConvParams p) {
    using v2f = __attribute__((ext_vector_type(2))) float;
    // Row pitches chosen for the LDS banks: a 64-bit read is served 16 lanes at a time (16 x 8 B = all 32 banks), and 16 lanes
    // are two rows of 8 strips whose windows start 6 dwords apart -- {0, 6, .., 30, 4, 10} mod 32; the second row's starts are
    // the OTHER eight even residues exactly when two map rows / one conv1 row are 16 dwords apart mod 32
    constexpr int MAPP = 56;   // floats per staged map row (50 used): 2 * 56 = 112 = 16 mod 32
    constexpr int C1Q = 40;    // channel pairs per conv1 row (26 used): 2 * 40 = 80 = 16 mod 32
    __shared__ __attribute__((aligned(16))) float s_map[MAPW * MAPP];
    __shared__ __attribute__((aligned(16))) float s_c1[2 * C1P * C1Q * 2];   // two planes of [26][40] channel PAIRS: {c0, c2}, then {c1, c3}; zero border (conv2's padding)
    __shared__ __attribute__((aligned(16))) float s_c2[NPOS];               // transposed: element pos at (pos % 16) * 36 + pos / 16
    const int t = threadIdx.x;
    static_assert(C1CH == 4, "channel pairs (0,1) and (2,3)");
    static_assert(PBLOCK >= 192 && C1W % 3 == 0, "one strip of three outputs per thread");
    // 105 uniform weights do not fit the ~100 SGPRs a wavefront has, and VGPRs are short too (36 linear weights per thread are
    // resident): conv2's 37 stay scalar, conv1's 64 sit in LDS as channel pairs and are read where they are used (one
    // address for the whole wavefront: a broadcast read)
    __shared__ v2f s_w1[2][16];
    v2f bias1[2], w2[2][9];
    float bias2;
    if (t < 32) s_w1[t >> 4][t & 15] = v2f{p.c1w[(2 * (t >> 4)) * 16 + (t & 15)], p.c1w[(2 * (t >> 4) + 1) * 16 + (t & 15)]};
#pragma unroll
    for (int h = 0; h < 2; h++) bias1[h] = v2f{p.c1b[2 * h], p.c1b[2 * h + 1]};
#pragma unroll
    for (int h = 0; h < 2; h++)   // pair h: input channels h and h + 2
#pragma unroll
        for (int k = 0; k < 9; k++) w2[h][k] = v2f{p.c2w[h * 9 + k], p.c2w[(h + 2) * 9 + k]};
    bias2 = p.c2b[0];
    for (int i = t; i < 2 * C1P * C1Q * 2; i += PBLOCK) s_c1[i] = 0.0f;
    const int j = t >> 4, sl = t & 15;   // linear: 16 threads per output feature, its 36 weights in registers
    const float lbias = p.lb[j];
    float lwr[NPOS / 16];
#pragma unroll
    for (int i = 0; i < NPOS / 16; i++) lwr[i] = p.lw[j * NPOS + sl + 16 * i];
    // this thread's strip: outputs (oy, ox0 .. ox0 + 2)
    const bool strip = t < (C1W / 3) * C1W;
    const int ts = strip ? t : 0;
    const int oy = ts / (C1W / 3), ox0 = 3 * (ts % (C1W / 3));

    // The next map is fetched into registers while this one computes, and a map's 16 features are stored one
    // iteration late: at the top of an iteration the only memory operations in flight are then the prefetch loads
    // (loads and stores share one in-order counter, see k_policy).
    constexpr int NV = (MAPW * MAPW / 4 + PBLOCK - 1) / PBLOCK;   // float4 per thread and map
    static_assert(NV == 3, "three float4 per thread");
    float4 pre0, pre1, pre2;   // plain scalars: an indexed private array captured by a lambda ends up in LDS
    pre0 = pre1 = pre2 = make_float4(0.f, 0.f, 0.f, 0.f);
#define CONV_FETCH(m_)                                                                                        \
    do {                                                                                                      \
        const float4 *src4 = reinterpret_cast<const float4 *>(p.maps + (size_t)(m_) * p.map_stride);          \
        pre0 = src4[t];                                                                                       \
        pre1 = src4[t + PBLOCK];                                                                              \
        if (t + 2 * PBLOCK < MAPW * MAPW / 4) pre2 = src4[t + 2 * PBLOCK];                                    \
    } while (0)
    if (p.vec4 && blockIdx.x < p.n_maps) CONV_FETCH(blockIdx.x);
    float pend = 0.0f;
    int pend_m = -1;
    for (int m = blockIdx.x; m < p.n_maps; m += gridDim.x) {
        if (p.vec4) {   // (50 is even: a pair of floats never straddles two rows)
            auto put = [&](int chunk, const float4 &v) __attribute__((always_inline)) {
                const int e0 = 4 * chunk, e1 = e0 + 2;
                *reinterpret_cast<v2f *>(s_map + (e0 / MAPW) * MAPP + e0 % MAPW) = v2f{v.x, v.y};
                *reinterpret_cast<v2f *>(s_map + (e1 / MAPW) * MAPP + e1 % MAPW) = v2f{v.z, v.w};
            };
            put(t, pre0);
            put(t + PBLOCK, pre1);
            if (t + 2 * PBLOCK < MAPW * MAPW / 4) put(t + 2 * PBLOCK, pre2);
            if (m + (int)gridDim.x < p.n_maps) CONV_FETCH(m + gridDim.x);
        } else {   // unaligned maps (caller-assembled rows): plain loads
            const float *src = p.maps + (size_t)m * p.map_stride;
            for (int i = t; i < MAPW * MAPW; i += PBLOCK) s_map[(i / MAPW) * MAPP + i % MAPW] = src[i];
        }
        if (pend_m >= 0 && sl == 0) p.feat[(size_t)pend_m * NFEAT + j] = pend;
        __syncthreads();
        if (strip) {   // conv1 (k = 4, stride 2) + ReLU: three positions x four channels from a 4 x 8 input window
            v2f acc[3][2];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                acc[d][0] = bias1[0];
                acc[d][1] = bias1[1];
            }
#pragma unroll
            for (int ky = 0; ky < 4; ky++) {
                const v2f *r = reinterpret_cast<const v2f *>(s_map + (2 * oy + ky) * MAPP + 2 * ox0);   // even index: 8-byte aligned
                const v2f a0 = r[0], a1 = r[1], a2 = r[2], a3 = r[3];
                const float in[8] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y};
#pragma unroll
                for (int kx = 0; kx < 4; kx++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {   // (per output and channel pair: taps in the order ky * 4 + kx, as before)
                        const v2f w = s_w1[h][ky * 4 + kx];
#pragma unroll
                        for (int d = 0; d < 3; d++) acc[d][h] = __builtin_elementwise_fma(w, v2f{in[2 * d + kx], in[2 * d + kx]}, acc[d][h]);
                    }
            }
#pragma unroll
            for (int d = 0; d < 3; d++) {   // plane 0: channels {0, 2}, plane 1: {1, 3} -- the pairs conv2's packed FMAs take
                v2f *o = reinterpret_cast<v2f *>(s_c1) + (oy + 1) * C1Q + ox0 + d + 1;
                o[0] = v2f{fmaxf(acc[d][0].x, 0.0f), fmaxf(acc[d][1].x, 0.0f)};
                o[C1P * C1Q] = v2f{fmaxf(acc[d][0].y, 0.0f), fmaxf(acc[d][1].y, 0.0f)};
            }
        }
        __syncthreads();
        if (strip) {   // conv2 (k = 3, pad 1) + ReLU: channels (0, 2) then (1, 3), halves added
            v2f acc[3];
#pragma unroll
            for (int d = 0; d < 3; d++) acc[d] = v2f{bias2, 0.0f};
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {   // one window row of the pair (h, h + 2) at a time: 10 registers, not 30
                    v2f q[5];
#pragma unroll
                    for (int c = 0; c < 5; c++) q[c] = reinterpret_cast<const v2f *>(s_c1)[h * C1P * C1Q + (oy + ky) * C1Q + ox0 + c];
#pragma unroll
                    for (int d = 0; d < 3; d++)   // (per output: taps in the order h, ky, kx, as before)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) acc[d] = __builtin_elementwise_fma(w2[h][ky * 3 + kx], q[d + kx], acc[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const int pos = oy * C1W + ox0 + d;
                s_c2[(pos & 15) * (NPOS / 16) + (pos >> 4)] = fmaxf(acc[d].x + acc[d].y, 0.0f);
            }
        }
        __syncthreads();
        v2f acc2 = {0.0f, 0.0f};   // linear 576 -> 16: feature j by 16 threads, 36 terms each, butterfly over the 16 lanes
        const float4 *c2v = reinterpret_cast<const float4 *>(s_c2 + sl * (NPOS / 16));
#pragma unroll
        for (int i = 0; i < NPOS / 16; i += 4) {
            const float4 v = c2v[i / 4];
            acc2 = __builtin_elementwise_fma(v2f{lwr[i], lwr[i + 1]}, v2f{v.x, v.y}, acc2);
            acc2 = __builtin_elementwise_fma(v2f{lwr[i + 2], lwr[i + 3]}, v2f{v.z, v.w}, acc2);
        }
        float acc = acc2.x + acc2.y;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        pend = acc + lbias;
        pend_m = m;
        // the next map's staging only writes s_map (last read before the second barrier above)
    }
    if (pend_m >= 0 && sl == 0) p.feat[(size_t)pend_m * NFEAT + j] = pend;
#undef CONV_FETCH
}

thread_local char g_perr[200] = "";

// persistent grid: as many blocks as the device holds at once (queried once)
template <typename K>
int resident_blocks(K kernel) {
    int dev = 0, cus = 256, per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess) return 512;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, PBLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    return cus * per_cu;
}

}  // namespace

extern "C" {

size_t cs_policy_packed_floats(void) { return (size_t)PACKED_FLOATS; }

// Host-side packing of torch-layout weights (network/base_net.py parameter names) into MFMA B-fragment order.
// fc1_w [64][in_dim], w_ih / w_hh [192][64], fc2a_w [64][64], fc2b_w [n_actions][64]; biases alongside.
int cs_policy_pack(const float *fc1_w, const float *fc1_b, const float *w_ih, const float *b_ih, const float *w_hh,
                   const float *b_hh, const float *fc2a_w, const float *fc2a_b, const float *fc2b_w, const float *fc2b_b,
                   int in_dim, int n_actions, float *packed) {
    if (in_dim < 1 || in_dim > KIN_MAX || n_actions < 1 || n_actions > 16) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_pack: in_dim must be 1..32 and n_actions 1..16");
        return CS_E_ARG;
    }
    for (int i = 0; i < PACKED_FLOATS; i++) packed[i] = 0.0f;
#if CS_POLICY_F16
    // split_f16 carries a value as hi = fp16(v), lo = fp16((v - hi) * 2048): above fp16's range hi is inf and lo NaN.  A weight
    // there is refused here; activations are bounded by the weights and the inputs (|obs| <= 1, |h| < 1): the bound
    // an activation must keep is documented in include/coopsearch.h.
    {
        const struct { const float *w; int n; const char *name; } ws[] = {
            {fc1_w, 64 * in_dim, "fc1.weight"}, {w_ih, 192 * 64, "rnn.weight_ih"}, {w_hh, 192 * 64, "rnn.weight_hh"},
            {fc2a_w, 64 * 64, "fc2.0.weight"}, {fc2b_w, n_actions * 64, "fc2.2.weight"}};
        for (const auto &t : ws)
            for (int i = 0; i < t.n; i++)
                if (!(fabsf(t.w[i]) <= 65504.0f)) {
                    snprintf(g_perr, sizeof(g_perr), "cs_policy_pack: %s[%d] = %g is outside the fp16 range (+-65504) of the split-fp16 matrix path",
                             t.name, i, (double)t.w[i]);
                    return CS_E_ARG;
                }
    }
    // split-fp16 fragments (policy_dev.h): fragment (column tile nt, k-step ks of 32) = hi plane | lo plane, each [64 lanes][8 halves];
    // lane l, j: W[16 nt + (l & 15)][k0 + 32 ks + 8 (l >> 4) + j] for k-blocks (l >> 4) < kblocks, zero beyond
    auto hfrag = [&](int off, int frag, const float *w, int n_out, int k_in, int nt, int k0, int kblocks) {
        _Float16 *hi = reinterpret_cast<_Float16 *>(packed + off + (size_t)frag * FRAG_DW), *lo = hi + 64 * 8;
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 8; j++) {
                const int n = 16 * nt + (l & 15), k = k0 + 8 * (l >> 4) + j;
                const float v = (n < n_out && k < k_in && (l >> 4) < kblocks) ? w[(size_t)n * k_in + k] : 0.0f;
                split_f16(v, hi[l * 8 + j], lo[l * 8 + j]);
            }
    };
    for (int nt = 0; nt < 4; nt++) hfrag(HOFF_W1, nt, fc1_w, 64, in_dim, nt, 0, 4);
    for (int nt = 0; nt < 12; nt++)
        for (int ks = 0; ks < 2; ks++) {
            hfrag(HOFF_WIH, nt * 2 + ks, w_ih, 192, 64, nt, 32 * ks, 4);
            hfrag(HOFF_WHH, nt * 2 + ks, w_hh, 192, 64, nt, 32 * ks, 4);
        }
    for (int nt = 0; nt < 4; nt++)
        for (int ks = 0; ks < 2; ks++) hfrag(HOFF_W2, nt * 2 + ks, fc2a_w, 64, 64, nt, 32 * ks, 4);
    for (int ks = 0; ks < 2; ks++) hfrag(HOFF_W3, ks, fc2b_w, n_actions, 64, 0, 32 * ks, 4);
    for (int i = 0; i < 64; i++) packed[HOFF_B1 + i] = fc1_b[i];
    for (int i = 0; i < 192; i++) packed[HOFF_BIH + i] = b_ih[i];
    for (int i = 0; i < 192; i++) packed[HOFF_BHH + i] = b_hh[i];
    for (int i = 0; i < 64; i++) packed[HOFF_B2 + i] = fc2a_b[i];
    for (int i = 0; i < n_actions; i++) packed[HOFF_B3 + i] = fc2b_b[i];
    return CS_OK;
#endif
    auto frag = [&](int off, int tiles, int ksteps, const float *w, int n_out, int k_in) {
        for (int nt = 0; nt < tiles; nt++)
            for (int kk = 0; kk < ksteps; kk++)
                for (int l = 0; l < 64; l++) {
                    const int n = 16 * nt + (l & 15), k = 4 * kk + (l >> 4);  // B[k][n] = W[n][k]
                    packed[off + (nt * ksteps + kk) * FR + l] = (n < n_out && k < k_in) ? w[(size_t)n * k_in + k] : 0.0f;
                }
    };
    frag(OFF_W1, 4, KIN_MAX / 4, fc1_w, 64, in_dim);
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
// feat_dev (nullable): 16 conv features per group of rows_per_feat rows, placed in front of the obs columns;
// hidden_dev [rows][64] is updated in place; q_dev (nullable) [rows][n_actions]; actions_dev [rows] int64.
int cs_policy_forward(const float *packed_dev, const float *obs_dev, int obs_stride, int obs_offset, const int64_t *last_dev,
                      const float *feat_dev, int rows_per_feat, float *hidden_dev, float *q_dev, int64_t *actions_dev,
                      int rows, int n_agents, int n_actions, float epsilon, const double *eps_env_dev, uint64_t seed, uint32_t step,
                      uint64_t row0, int select, void *stream) {
    const int in_dim = (feat_dev ? NFEAT : 0) + 4 + n_actions + n_agents;
    if (!packed_dev || !obs_dev || !hidden_dev || !actions_dev || rows < 1 || n_agents < 1 || n_actions < 1 ||
        in_dim > KIN_MAX || (feat_dev && rows_per_feat < 1)) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_forward: bad argument (input width %d, limit %d)", in_dim, KIN_MAX);
        return CS_E_ARG;
    }
    PolicyParams p{rows, n_agents, n_actions, obs_stride, obs_offset, epsilon, eps_env_dev, seed, step, row0, select, packed_dev, obs_dev, last_dev,
                   feat_dev, rows_per_feat, hidden_dev, q_dev, actions_dev};
    const int tiles = (rows + 15) / 16;
#if CS_POLICY_F16
    {
        static const int resident = resident_blocks(k_policy_h);   // persistent grid: what the device holds at once
        hipLaunchKernelGGL(k_policy_h, dim3(tiles < resident ? tiles : resident), dim3(PBLOCK), 0, (hipStream_t)stream, p);
        if (hipGetLastError() != hipSuccess) {
            snprintf(g_perr, sizeof(g_perr), "cs_policy_forward: kernel launch failed");
            return CS_E_LAUNCH;
        }
        return CS_OK;
    }
#endif
    if (in_dim <= 16) {
        static const int resident = resident_blocks(k_policy<4>);   // persistent grid: what the device holds at once
        hipLaunchKernelGGL(k_policy<4>, dim3(tiles < resident ? tiles : resident), dim3(PBLOCK), 0, (hipStream_t)stream, p);
    } else {
        static const int resident = resident_blocks(k_policy<8>);
        hipLaunchKernelGGL(k_policy<8>, dim3(tiles < resident ? tiles : resident), dim3(PBLOCK), 0, (hipStream_t)stream, p);
    }
    if (hipGetLastError() != hipSuccess) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_forward: kernel launch failed");
        return CS_E_LAUNCH;
    }
    return CS_OK;
}

// flight: conv features of n_maps probability maps (map m = 2500 floats at maps_dev + m*map_stride) -> feat_dev
// [n_maps][16].  Weights are the torch tensors of network/base_net.py's `conv` / `linear` as they are (device pointers).
int cs_policy_conv_features(const float *conv1_w_dev, const float *conv1_b_dev, const float *conv2_w_dev,
                            const float *conv2_b_dev, const float *lin_w_dev, const float *lin_b_dev,
                            const float *maps_dev, int64_t map_stride, int n_maps, float *feat_dev, void *stream) {
    if (!conv1_w_dev || !conv1_b_dev || !conv2_w_dev || !conv2_b_dev || !lin_w_dev || !lin_b_dev || !maps_dev || !feat_dev ||
        n_maps < 1) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_conv_features: bad argument");
        return CS_E_ARG;
    }
    const int vec4 = ((uintptr_t)maps_dev % 16 == 0) && (map_stride % 4 == 0);
    ConvParams p{conv1_w_dev, conv1_b_dev, conv2_w_dev, conv2_b_dev, lin_w_dev, lin_b_dev, maps_dev, (long long)map_stride,
                 n_maps, vec4, feat_dev};
    static const int resident = resident_blocks(k_conv_features);
    hipLaunchKernelGGL(k_conv_features, dim3(n_maps < resident ? n_maps : resident), dim3(PBLOCK), 0, (hipStream_t)stream, p);
    if (hipGetLastError() != hipSuccess) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_conv_features: kernel launch failed");
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
