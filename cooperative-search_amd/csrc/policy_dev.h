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
constexpr int PACKED_FLOATS_F32 = OFF_B3 + 16;

// ---------------------------------------------------------------------------------------------------------------------------
// Split-fp16 matrix path (default).  The fp32 matrix pipe (v_mfma_f32_16x16x4_f32) runs at the vector rate, 1/16 of the 16-bit
// rate, and at 92 kFLOP per env-step it caps the closed loop at ~1.7e9 env-steps/s however fast the env is.  Here every fp32
// operand v is carried as TWO fp16 numbers, v = hi + lo / 2048 with hi = fp16(v) and lo = fp16((v - hi) * 2048) -- 22 significant
// bits, the low part pre-scaled so that it stays in fp16's normal range wherever hi is normal -- and a product a*b is evaluated as
//     a_hi*b_hi + (a_lo*b_hi + a_hi*b_lo) / 2048
// with three v_mfma_f32_16x16x32_f16 (exact fp16 products, fp32 accumulation; the dropped a_lo*b_lo term is 2^-22 relative):
// three matrix instructions per K = 32 instead of eight fp32 ones per K = 32, each ~17 instead of 32 cycles -- 5x the rate at
// an error of ~2.4e-7 per product, two orders below the 2e-5 parity bar with the reference network (network/base_net.py:31-46).
// Values below fp16's normal range (6.1e-5) become subnormal halves; the matrix pipe multiplies those exactly (see split_f16).
// ---------------------------------------------------------------------------------------------------------------------------
#ifndef CS_POLICY_F16
#define CS_POLICY_F16 1
#endif
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int HST = 72;                 // halves per LDS row of a split activation plane: rows 144 B apart, conflict-free 16-byte reads
constexpr int HXS = 40;                 // ... of the 32-column input planes of the fused closed loop: rows 80 B apart, conflict-free too
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
constexpr float F16_MIN_NORMAL = 6.2e-5f;
// packed weights, in dwords: fragment f = 2 planes (hi, lo) x 64 lanes x 4 dwords (8 halves): lane l of fragment (column tile nt,
// k-step ks of 32) holds W[16 nt + (l & 15)][32 ks + 8 (l >> 4) + j], j = 0..7 -- the B operand of one 16x16x32 MFMA
constexpr int FRAG_DW = 2 * 64 * 4;
constexpr int HOFF_W1 = 0;                               // [4 col tiles][1 k-step]   (inputs padded to 32 columns)
constexpr int HOFF_WIH = HOFF_W1 + 4 * 1 * FRAG_DW;      // [12][2]
constexpr int HOFF_WHH = HOFF_WIH + 12 * 2 * FRAG_DW;    // [12][2]
constexpr int HOFF_W2 = HOFF_WHH + 12 * 2 * FRAG_DW;     // [4][2]
constexpr int HOFF_W3 = HOFF_W2 + 4 * 2 * FRAG_DW;       // [1 col tile][2]: the whole K = 64 (one wavefront computes a tile's q and selects)
constexpr int HOFF_B1 = HOFF_W3 + 2 * FRAG_DW;           // biases, fp32
constexpr int HOFF_BIH = HOFF_B1 + 64;
constexpr int HOFF_BHH = HOFF_BIH + 192;
constexpr int HOFF_B2 = HOFF_BHH + 192;
constexpr int HOFF_B3 = HOFF_B2 + 64;
constexpr int PACKED_FLOATS_F16 = HOFF_B3 + 16;
constexpr int PACKED_FLOATS = CS_POLICY_F16 ? PACKED_FLOATS_F16 : PACKED_FLOATS_F32;

// v -> (hi, lo) of the split representation (same code on the host for the weights, cs_policy_pack)
// hi = fp16(v) whatever v's size: the gfx950 matrix pipe takes SUBNORMAL fp16 inputs exactly (tools/mfma_f16_denorm.hip, run by
// tests/test_gpu_mfma_denorm.py: products of subnormal halves against the fp64 sum, 1.4e-7 = the fp32 accumulation).  The first
// version zeroed hi below fp16's normal range "so that no subnormal is handed to the matrix pipe": a compare and a select per stored
// element in phases that are bound by VALU issue (closed loop 5.90 -> 6.11e8 without them).  -DCS_SPLIT_GUARD=1 brings the guard back
// for a pipe that flushes.
#ifndef CS_SPLIT_GUARD
#define CS_SPLIT_GUARD 0
#endif
// float -> half, round to nearest even, subnormal results kept: the instruction is PINNED on the device.  Written as a C cast, the
// conversions of split_f16 came out of the compiler differently in k_policy_h and in the fused loop once the subnormal guard was
// gone, and the two kernels' hidden states -- which must agree bit for bit -- differed by an ulp in rare elements after ~50 steps
// (values below fp16's normal range; v_cvt_f16_f32 and v_cvt_pk_f16_f32 themselves agree on every input tried).  With the
// instruction pinned they agree again (tests/test_gpu_policy.py::test_fused_closed_loop_rollout_equals_stepwise, 200 steps).
__host__ __device__ __forceinline__ _Float16 cvt_half(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16 h;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(v));
    return h;
#else
    return (_Float16)v;
#endif
}
// Values beyond fp16's range SATURATE at +-65504 (round 6): unclamped, hi would be inf and lo = (v - inf) * 2048 = NaN,
// and one such activation would turn every q-value of its row into NaN.  Weights are refused by cs_policy_pack instead (a saturated
// weight is a different network); an ACTIVATION of that size -- only relu(fc1 x) can reach it, the recurrent state lies in (-1, 1) --
// behaves like min(relu(.), 65504).  One v_med3_f32 per conversion; values inside the range are unchanged bit for bit.
constexpr float F16_MAX = 65504.0f;
__host__ __device__ __forceinline__ float clamp_f16_range(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, -F16_MAX, F16_MAX);
#else
    return v < -F16_MAX ? -F16_MAX : (v > F16_MAX ? F16_MAX : v);
#endif
}
__host__ __device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    v = clamp_f16_range(v);
#if CS_SPLIT_GUARD
    const float a = v < 0.0f ? -v : v;
    hi = a < F16_MIN_NORMAL ? (_Float16)0.0f : (_Float16)v;
#else
    hi = cvt_half(v);
#endif
    lo = cvt_half((v - (float)hi) * LO_SCALE);
}
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// one element of a split activation plane pair in LDS (planes: [rows][HST] halves)
__device__ __forceinline__ void split_store(_Float16 *hi, _Float16 *lo, int idx, float v) {
    _Float16 h, l;
    split_f16(v, h, l);
    hi[idx] = h;
    lo[idx] = l;
}
// B fragment (hi, lo) of one (column tile, k-step) for this lane
struct BFrag {
    h8 hi, lo;
};
__device__ __forceinline__ BFrag load_bfrag(const float *packed, int off_dw, int frag, unsigned lane) {
    const uint4 *base = reinterpret_cast<const uint4 *>(packed + off_dw + (size_t)frag * FRAG_DW);
    const uint4 a = base[lane], b = base[64 + lane];
    BFrag f;
    __builtin_memcpy(&f.hi, &a, 16);
    __builtin_memcpy(&f.lo, &b, 16);
    return f;
}
// A fragment of k-step ks from a split plane pair (16 rows of one tile starting at `row0` of the planes; row pitch ST halves)
template <int ST = HST>
__device__ __forceinline__ void load_afrag(const _Float16 *hi, const _Float16 *lo, int row0, int ks, int lane, h8 &ah, h8 &al) {
    const int idx = (row0 + (lane & 15)) * ST + 32 * ks + 8 * (lane >> 4);
    ah = *reinterpret_cast<const h8 *>(hi + idx);
    al = *reinterpret_cast<const h8 *>(lo + idx);
}
// one k-step of the split product into the (hi, lo) accumulator pair: ALWAYS in this order, in every kernel
__device__ __forceinline__ void mfma_split(const h8 ah, const h8 al, const BFrag &b, f32x4 &hi, f32x4 &lo) {
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b.hi, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b.lo, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b.hi, hi, 0, 0, 0);
}
// hi + lo / 2048 in ONE instruction: the scaling by a power of two is exact, so the fused form rounds exactly like the product followed
// by the sum (the translation unit is compiled with -ffp-contract=off: written as a sum it stays two instructions)
__device__ __forceinline__ float split_sum(float hi, float lo) { return __builtin_fmaf(lo, LO_INV, hi); }
#endif


// Gate nonlinearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each): absolute error ~1e-7 on
// outputs in [-1, 1], well inside the fp32 tolerance of the parity tests; the libm versions cost ~50 VALU
// instructions each, and VALU work competes with the co-resident block's MFMAs for the SIMD.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

#if CS_POLICY_F16 && (defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__))
// GRUCell (torch.nn.GRUCell: gates r, z, n in weight_ih / weight_hh) of one 16-row tile for a wavefront's 16 hidden columns, shared by
// k_policy_h and the fused closed loop so that both walk the SAME products in the SAME order (their actions must agree bit for bit).
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr), z likewise: ONE accumulator chain each over [x | h] (K = 128: x's two k-steps, then
//   h's) and ONE bias (b_i + b_h, summed once per kernel) -- the first version kept W_i x and W_h h apart and paid two accumulator
//   reads, a split sum and two additions more per output element and gate, in a phase that is bound by VALU issue;
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn)): the two halves stay apart, r multiplies the hidden half only.
// bg[2 g][ks] / bg[2 g + 1][ks]: fragments of W_i / W_h of gate g, k-step ks.
struct GruAcc {
    f32x4 r_hi, r_lo, z_hi, z_lo, in_hi, in_lo, hn_hi, hn_lo;
};
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
// b_r = b_ir + b_hr, b_z = b_iz + b_hz.  Every bias ENTERS its accumulator (a lane's four elements are four rows of one column: one
// bias value) instead of being added to the finished sum: no bias addition in the epilogue at all.
__device__ __forceinline__ void gru_products(const h8 (&xh)[2], const h8 (&xl)[2], const h8 (&hh)[2], const h8 (&hl)[2],
                                             const BFrag (&bg)[6][2], float b_r, float b_z, float b_in, float b_hn, GruAcc &a) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    a.r_hi = splat4(b_r);
    a.z_hi = splat4(b_z);
    a.in_hi = splat4(b_in);
    a.hn_hi = splat4(b_hn);
    a.r_lo = a.z_lo = a.in_lo = a.hn_lo = zero;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) mfma_split(xh[ks], xl[ks], bg[0][ks], a.r_hi, a.r_lo);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) mfma_split(hh[ks], hl[ks], bg[1][ks], a.r_hi, a.r_lo);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) mfma_split(xh[ks], xl[ks], bg[2][ks], a.z_hi, a.z_lo);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) mfma_split(hh[ks], hl[ks], bg[3][ks], a.z_hi, a.z_lo);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) mfma_split(xh[ks], xl[ks], bg[4][ks], a.in_hi, a.in_lo);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) mfma_split(hh[ks], hl[ks], bg[5][ks], a.hn_hi, a.hn_lo);
}
// new hidden value of element r (0..3: the lane's four rows) from the tile's accumulators: h' = n + z (h - n), n = tanh(i_n + r h_n)
// (each a single fma: the translation unit is compiled without contraction, so they are written out)
__device__ __forceinline__ float gru_cell(const GruAcc &a, int r, float h_prev) {
    const float rg = sigmoidf_(split_sum(a.r_hi[r], a.r_lo[r]));
    const float zg = sigmoidf_(split_sum(a.z_hi[r], a.z_lo[r]));
    const float ng = tanhf_(__builtin_fmaf(rg, split_sum(a.hn_hi[r], a.hn_lo[r]), split_sum(a.in_hi[r], a.in_lo[r])));
    return __builtin_fmaf(zg, h_prev - ng, ng);
}
#endif

// splitmix64: per-row uniform for the epsilon-greedy choice (the reference draws from numpy's global stream on the
// host; any iid uniform source is equivalent)
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}


// per-row random bits of one selection: counter-based, keyed by (seed, step, GLOBAL row) -- global, so that a sharded
// batch draws the same exploration noise whatever the split (row = (env_offset + env) * n_agents + agent)
__device__ __forceinline__ unsigned long long row_bits(unsigned long long seed, unsigned step, unsigned long long grow) {
    // the fields are CHAINED through the mixer, each entering a value that already depends on the ones before it: XOR-ing
    // separately hashed fields (the first version) let (step = a, row = b) collide with (step = b, row = a) whenever the two hashed
    // to swapped values, and cancelled for equal ones.  seed and step are wave-uniform: two of the three mixes run on the scalar unit.
    return mix64(mix64(mix64(seed) + (unsigned long long)step) + grow * 0x9e3779b97f4a7c15ull);
}

// epsilon-greedy on top of the greedy choice `arg` (agent/agent.py:70-75): with probability epsilon a uniform action
__device__ __forceinline__ int epsilon_greedy(int arg, float epsilon, unsigned long long seed, unsigned step,
                                              unsigned long long grow, int n_actions) {
    if (epsilon > 0.0f) {
        const unsigned long long h = row_bits(seed, step, grow);
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
        if (u < epsilon) return (int)((h & 0xffffffull) % (unsigned)n_actions);
    }
    return arg;
}

// Action of one row from its q values (qf(a) = q value of action a), all actions available (flight_env_easy.py:184-188).
//   sel = 0                      agent/agent.py:68-75: argmax (first maximum), epsilon-greedy
//   sel & CS_SELECT_SOFTMAX      agent/agent.py:77-97 (_choose_action_from_softmax, alg == 'reinforce'):
//                                prob = (1 - eps) * softmax(q) + eps / |A|; argmax(prob) when the caller asked for the
//                                deterministic choice (epsilon == 0 and evaluate: no CS_SELECT_SAMPLE), otherwise one
//                                draw from Categorical(prob) by inverse CDF on the row's uniform
template <class QF>
__device__ __forceinline__ int select_action(QF qf, int n_actions, int sel, float epsilon, unsigned long long seed,
                                             unsigned step, unsigned long long grow) {
    float best = -3.0e38f;
    int arg = 0;
    for (int a = 0; a < n_actions; a++) {
        const float qv = qf(a);
        if (qv > best) {   // strict: the first maximum wins, like torch.argmax
            best = qv;
            arg = a;
        }
    }
    if (!(sel & CS_SELECT_SOFTMAX)) return epsilon_greedy(arg, epsilon, seed, step, grow, n_actions);
    float sum = 0.0f;
    for (int a = 0; a < n_actions; a++) sum += __expf(qf(a) - best);
    if (!(sel & CS_SELECT_SAMPLE)) {
        // argmax over the float32 prob, like agent.py:93 -- not over q: two q values close enough to round to the same
        // prob tie there, and the first of them wins.  (The probabilities here come from the hardware exp2 and a multiplication
        // by the reciprocal of the sum, torch's from expf and a division: the SAME q values can round to a tie in one and not in
        // the other.  The parity tests therefore treat near-ties -- top two outputs within 1e-3 -- as either way.)
        const float inv0 = (1.0f - epsilon) / sum, uni0 = epsilon / (float)n_actions;
        float pbest = -1.0f;
        int parg = 0;
        for (int a = 0; a < n_actions; a++) {
            const float pa = __expf(qf(a) - best) * inv0 + uni0;
            if (pa > pbest) {
                pbest = pa;
                parg = a;
            }
        }
        return parg;
    }
    const float u = (float)(row_bits(seed, step, grow) >> 40) * (1.0f / 16777216.0f);   // [0, 1)
    const float inv = (1.0f - epsilon) / sum, uni = epsilon / (float)n_actions;
    float cum = 0.0f;
    int pick = n_actions - 1;
    for (int a = 0; a < n_actions; a++) {
        cum += __expf(qf(a) - best) * inv + uni;
        if (u < cum) {
            pick = a;
            break;
        }
    }
    return pick;
}
