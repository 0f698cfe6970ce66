// cooperative-search_amd/csrc/policy_dev.h -- device-side definitions of the agent-network kernels shared by
// policy.hip (k_policy, k_conv_features) and coopsearch.hip (k_rollout_policy, the fused policy + env rollout).
// Include inside the translation unit's anonymous namespace.
#pragma once

constexpr int H = 64;            // rnn_hidden_dim of the reference (common/arguments.py:58)
constexpr int KIN_MAX = 32;      // padded input width: 16 (obs ++ last ++ id) or 32 (+ 16 conv features in front)
constexpr int NFEAT = 16;        // conv_out_dim of the reference's flight network (common/arguments.py:265)
constexpr int LDW = 68;          // LDS row stride in floats (68 % 32 = 4: 2-way conflicts at most on the A reads)
constexpr int PBLOCK = 256;      // 4 wavefronts, one 16-column tile each

using f32x4 = __attribute__((ext_vector_type(4))) float;

// packed weight fragments, in floats: fragment (column tile nt, k-step kk) holds, for lane l,
// W[16*nt + (l & 15)][4*kk + (l >> 4)] -- the B operand of one 16x16x4 MFMA
constexpr int FR = 64;
constexpr int OFF_W1 = 0;                           // [4 col tiles][8 k-steps][64] (k-steps 4..7 zero when in_dim <= 16)
constexpr int OFF_WIH = OFF_W1 + 4 * (KIN_MAX / 4) * FR;  // [12][16][64]
constexpr int OFF_WHH = OFF_WIH + 12 * 16 * FR;     // [12][16][64]
constexpr int OFF_W2 = OFF_WHH + 12 * 16 * FR;      // [4][16][64]
constexpr int OFF_W3 = OFF_W2 + 4 * 16 * FR;        // [1][16][64]
constexpr int OFF_B1 = OFF_W3 + 16 * FR;            // 64
constexpr int OFF_BIH = OFF_B1 + 64;                // 192
constexpr int OFF_BHH = OFF_BIH + 192;              // 192
constexpr int OFF_B2 = OFF_BHH + 192;               // 64
constexpr int OFF_B3 = OFF_B2 + 64;                 // 16
constexpr int PACKED_FLOATS = OFF_B3 + 16;


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


// epsilon-greedy on top of the greedy choice `arg` (agent/agent.py:70-75): with probability epsilon a uniform action,
// from a counter-based generator keyed by (seed, step, row)
__device__ __forceinline__ int epsilon_greedy(int arg, float epsilon, unsigned long long seed, unsigned step, int row,
                                              int n_actions) {
    if (epsilon > 0.0f) {
        const unsigned long long h = mix64(seed ^ mix64(((unsigned long long)step << 32) | (unsigned)row));
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
        if (u < epsilon) return (int)((h & 0xffffffull) % (unsigned)n_actions);
    }
    return arg;
}
