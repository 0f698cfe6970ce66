// cooperative-search_amd/csrc/policy.hip -- fused recurrent-policy forward for the batched collector (gfx950).
//
// "Next" row f3 of SURVEY.md section 8(f): the reference picks actions one agent at a time with a batch-1 forward
// of its RNN (agent/agent.py:33-75, network/base_net.py:5-46: fc1 -> ReLU -> GRUCell -> fc2[Linear, ReLU, Linear])
// and an epsilon-greedy choice on the host.  Here ONE launch does, for all R = B*n (env, agent) rows:
//     x  = [obs(4) | one_hot(last_action)(A) | one_hot(agent_id)(n)]        (agent.py:41-52)
//     h1 = relu(W1 x + b1);  h' = GRUCell(h1, h);  q = W3 relu(W2 h' + b2) + b3
//     action = argmax_a q  (greedy)  or uniform over actions with probability epsilon
// The three GEMM-shaped products run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, same rate as
// the fp32 VALU but with register-level operand reuse): one wavefront owns 16 rows, the weight fragments are
// pre-packed on the host in the exact per-lane order (one coalesced 256-byte load per MFMA), activations go
// through a small per-wave LDS tile to move from the C/D layout (col = lane&15, row = 4*(lane>>4)+reg) to the A
// layout (row = lane&15, k = lane>>4).  Non-conv (flight_easy) networks only; flight's conv front end stays in torch.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>

#include "coopsearch.h"

namespace {

constexpr int H = 64;            // rnn_hidden_dim of the reference (common/arguments.py:58)
constexpr int KIN = 16;          // padded input width (4 + n_actions + n_agents <= 16)
constexpr int LDW = 68;          // LDS row stride in floats (68 % 32 = 4: 2-way conflicts at most on the A reads)
constexpr int PBLOCK = 256;      // 4 wavefronts x 16 rows

using f32x4 = __attribute__((ext_vector_type(4))) float;

// packed weight fragments, in floats
constexpr int FR = 64;                              // one MFMA B fragment = 64 lanes x 1 float
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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// splitmix64: per-row uniform for the epsilon-greedy choice (the reference draws from numpy's global stream on the
// host; any iid uniform source is equivalent)
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// acc[tile] += A(16 x K from LDS, row-major stride LDW) * B(packed fragments): K/4 MFMAs per column tile
template <int KSTEPS>
__device__ __forceinline__ f32x4 mfma_tile(const float *a_lds, const float *wfrag, int lane, f32x4 acc) {
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
        const float a = a_lds[(lane & 15) * LDW + 4 * kk + (lane >> 4)];
        const float b = wfrag[kk * FR + lane];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(PBLOCK) void k_policy(PolicyParams p) {
    __shared__ float s_x[PBLOCK / 64][16 * LDW];   // layer input in A-readable form (x, then h1, then h', then f)
    __shared__ float s_h[PBLOCK / 64][16 * LDW];   // previous hidden state
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *xa = s_x[wave], *hp = s_h[wave];
    const int row0 = (blockIdx.x * (PBLOCK / 64) + wave) * 16;
    if (row0 >= p.rows) return;
    const int crow = (lane >> 4) * 4;   // C/D layout: this lane holds rows crow..crow+3 of column (lane & 15) + 16*tile
    const int ccol = lane & 15;

    // ---- stage x (16 rows x 16) and h (16 rows x 64) in LDS
    {
        const int r = lane >> 2, part = lane & 3;  // 4 lanes per row
        const int row = row0 + r < p.rows ? row0 + r : p.rows - 1;
        float v[4];
        if (!p.last) {  // raw mode: the caller assembled the input rows itself
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int k = 4 * part + c;
                v[c] = k < 4 + p.n_actions + p.n_agents ? p.obs[(size_t)row * p.obs_stride + p.obs_offset + k] : 0.0f;
            }
        } else if (part == 0) {
            const float4 o = *reinterpret_cast<const float4 *>(p.obs + (size_t)row * p.obs_stride + p.obs_offset);
            v[0] = o.x; v[1] = o.y; v[2] = o.z; v[3] = o.w;
        } else {
            const int la = (int)p.last[row], ag = row % p.n_agents;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int k = 4 * part + c;  // input column
                float f = 0.0f;
                if (k < 4 + p.n_actions) f = (k - 4 == la) ? 1.0f : 0.0f;
                else if (k < 4 + p.n_actions + p.n_agents) f = (k - 4 - p.n_actions == ag) ? 1.0f : 0.0f;
                v[c] = f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) xa[r * LDW + 4 * part + c] = v[c];
#pragma unroll
        for (int c = 0; c < 4; c++) {  // 16 floats of h per lane, as 4 float4
            const float4 hv = *reinterpret_cast<const float4 *>(p.hidden + (size_t)row * H + 16 * part + 4 * c);
            float *d = hp + r * LDW + 16 * part + 4 * c;
            d[0] = hv.x; d[1] = hv.y; d[2] = hv.z; d[3] = hv.w;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- h1 = relu(W1 x + b1)
    f32x4 h1[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma_tile<KIN / 4>(xa, p.w + OFF_W1 + nt * (KIN / 4) * FR, lane, acc);
        const float b = p.w[OFF_B1 + 16 * nt + ccol];
#pragma unroll
        for (int r = 0; r < 4; r++) h1[nt][r] = fmaxf(acc[r] + b, 0.0f);
    }
    __builtin_amdgcn_wave_barrier();  // all A reads of x are done before xa is overwritten
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) xa[(crow + r) * LDW + 16 * nt + ccol] = h1[nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- GRUCell (torch.nn.GRUCell: gates ordered r, z, n in weight_ih / weight_hh)
    f32x4 hn[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        f32x4 ir = {0.f, 0.f, 0.f, 0.f}, iz = ir, in_ = ir, hr = ir, hz = ir, hn_ = ir;
        ir = mfma_tile<16>(xa, p.w + OFF_WIH + (nt + 0) * 16 * FR, lane, ir);
        iz = mfma_tile<16>(xa, p.w + OFF_WIH + (nt + 4) * 16 * FR, lane, iz);
        in_ = mfma_tile<16>(xa, p.w + OFF_WIH + (nt + 8) * 16 * FR, lane, in_);
        hr = mfma_tile<16>(hp, p.w + OFF_WHH + (nt + 0) * 16 * FR, lane, hr);
        hz = mfma_tile<16>(hp, p.w + OFF_WHH + (nt + 4) * 16 * FR, lane, hz);
        hn_ = mfma_tile<16>(hp, p.w + OFF_WHH + (nt + 8) * 16 * FR, lane, hn_);
        const int col = 16 * nt + ccol;
        const float bir = p.w[OFF_BIH + col], biz = p.w[OFF_BIH + 64 + col], bin = p.w[OFF_BIH + 128 + col];
        const float bhr = p.w[OFF_BHH + col], bhz = p.w[OFF_BHH + 64 + col], bhn = p.w[OFF_BHH + 128 + col];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float rg = sigmoidf_((ir[r] + bir) + (hr[r] + bhr));
            const float zg = sigmoidf_((iz[r] + biz) + (hz[r] + bhz));
            const float ng = tanhf((in_[r] + bin) + rg * (hn_[r] + bhn));
            const float hprev = hp[(crow + r) * LDW + col];
            hn[nt][r] = (1.0f - zg) * ng + zg * hprev;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = row0 + crow + r;
            xa[(crow + r) * LDW + 16 * nt + ccol] = hn[nt][r];
            if (row < p.rows) p.hidden[(size_t)row * H + 16 * nt + ccol] = hn[nt][r];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- f = relu(W2 h' + b2)
    f32x4 f[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma_tile<16>(xa, p.w + OFF_W2 + nt * 16 * FR, lane, acc);
        const float b = p.w[OFF_B2 + 16 * nt + ccol];
#pragma unroll
        for (int r = 0; r < 4; r++) f[nt][r] = fmaxf(acc[r] + b, 0.0f);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) xa[(crow + r) * LDW + 16 * nt + ccol] = f[nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- q = W3 f + b3 (n_actions <= 16 columns), argmax / epsilon-greedy
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    q = mfma_tile<16>(xa, p.w + OFF_W3, lane, q);
    const float b3 = p.w[OFF_B3 + ccol];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float qv = q[r] + b3;
        const int row = row0 + crow + r;
        if (p.q && ccol < p.n_actions && row < p.rows) p.q[(size_t)row * p.n_actions + ccol] = qv;
        // first maximal action of the row: columns live in lanes (lane & ~15) + a
        float best = -3.0e38f;
        int arg = 0;
        for (int a = 0; a < p.n_actions; a++) {
            const float v = __shfl(qv, (lane & ~15) + a, 64);
            if (v > best) {
                best = v;
                arg = a;
            }
        }
        if (ccol == 0 && row < p.rows) {
            int act = arg;
            if (p.epsilon > 0.0f) {
                const unsigned long long h = mix64(p.seed ^ mix64(((unsigned long long)p.step << 32) | (unsigned)row));
                const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
                if (u < p.epsilon) act = (int)((h & 0xffffffull) % (unsigned)p.n_actions);
            }
            p.actions[row] = act;
        }
    }
}

thread_local char g_perr[200] = "";

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
    if (!packed_dev || !obs_dev || !hidden_dev || !actions_dev || rows < 1 || 4 + n_actions + n_agents > KIN) {
        snprintf(g_perr, sizeof(g_perr), "cs_policy_forward: bad argument");
        return CS_E_ARG;
    }
    PolicyParams p{rows, n_agents, n_actions, obs_stride, obs_offset, epsilon, seed, step, packed_dev, obs_dev, last_dev,
                   hidden_dev, q_dev, actions_dev};
    const int rows_per_block = (PBLOCK / 64) * 16;
    hipLaunchKernelGGL(k_policy, dim3((rows + rows_per_block - 1) / rows_per_block), dim3(PBLOCK), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? CS_OK : CS_E_LAUNCH;
}

const char *cs_policy_last_error(void) { return g_perr; }

}  // extern "C"
