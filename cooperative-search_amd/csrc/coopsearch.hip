// cooperative-search_amd/csrc/coopsearch.hip -- gfx950 (MI355X) kernels + the C ABI of include/coopsearch.h.
//
// Batched flight_easy / flight environment path of WZN1ng/Cooperative-Search, written for CDNA4:
//   * one environment = one 16-lane group of a wave64 (4 envs per wavefront): lane t owns target t, the <= 8
//     agents live replicated in every lane's registers, so the Gauss-Seidel kinematics need no communication
//     and the 15 x n sensor tests of the detection pass run across lanes;
//   * the detection mask is a wavefront ballot; a prefix popcount of the group's 16-bit slice gives every
//     in-range (agent, target) pair its offset in the env's private NumPy-compatible MT19937 stream, in the
//     reference's agent-major order;
//   * MT19937 is kept in its circular (incremental) form, so a draw touches 5 state words and there is no
//     624-word twist spike: word k is regenerated from words k, k+1, k+397 at the moment it is consumed;
//   * fp64 for everything that decides an integer outcome (positions, yaw, distance tests), fp32 only for
//     the emitted obs / state / reward tensors.  Compiled with -ffp-contract=off: the reference's CPython
//     arithmetic never fuses, and its wall test is knife-edged at the 1-ulp level.
//
// Reference citations are relative to the reference repo root.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

#include "coopsearch.h"

namespace {

#include "policy_dev.h"

constexpr int G = 16;        // lanes per environment
constexpr int BLOCK = 256;   // 4 wavefronts, 16 environments
constexpr int MT_N = 624;
constexpr int MT_PAD = CS_MT_PAD;        // words 0..31 of each env's state are mirrored at 624..655, so a 32-word
constexpr int MT_STRIDE = CS_MT_STRIDE;  // window starting anywhere in 0..623 never wraps (row padded to 21 x 128 B)
constexpr int MT_CANON = MT_N;           // cs_mt_canonical: every word twisted ahead of the cursor
static_assert(MT_STRIDE >= MT_N + MT_PAD, "MT row too short for its mirror");
constexpr int MT_M = 397;
constexpr int TRIG_ROWS = 37, TRIG_COLS = 7;
constexpr int FLAG_WIN = 1, FLAG_DIRTY = 2, FLAG_RESET_PASS = 4;

// {A_hi, A_lo, A_lo2, S_hi, S_lo, C_hi, C_lo} for A = k*pi/18 (gen_trig_table.py)
__device__ const double g_trig[TRIG_ROWS][TRIG_COLS] = {
#include "trig_table.inc"
};

typedef float v4f __attribute__((ext_vector_type(4)));

struct DevParams {
    int B, n_targets, map_size, cells, time_limit, agent_mode, target_mode, variant;
    double velocity, force_k, force_d2, view_r2, L, q, mid, inv_half;
    unsigned long long detect_K;  // U <= detect_prob  <=>  53-bit integer draw <= floor(detect_prob * 2^53)
    double tx0[CS_MAX_TARGETS], ty0[CS_MAX_TARGETS], jx2[CS_MAX_TARGETS], jy2[CS_MAX_TARGETS];
    unsigned deter_mask;
    double *tgt;     // [B][16][2]
    double *agent;   // [B][8][4]
    int *hdr;        // [B][16]
    unsigned *mt;    // [B][640]
    int *ahead;      // [B] words at the cursor that are already twisted
    unsigned *tape;  // [B][16] hit bits of the twisted words (lane kernel), see k_mt_advance
    float *prob;     // [B][cells]
    char *job;       // [2][B] MapJob records (flight): what k_map needs of the step that ran before it
    int obs_row_w;   // floats per (env, agent) row of the obs output: 4, flight: cells + 4 (map first, flight_env.py:223-230)
    int obs_feat_off;  // where the agent's own 4 floats sit in its row: 0, flight: cells
    float thr32, eps32;  // lane kernel's fp32 pre-filter of the sensor test, in normalised coordinates
    double start_x[CS_MAX_AGENTS], start_y[CS_MAX_AGENTS], start_yaw;  // start_pose() of every agent, evaluated once on the host
};

// The kernel's own DevParams as it sits in the kernarg segment (every kernel here takes it as its FIRST argument), behind
// a pointer the compiler cannot see through.  Cold paths (resets, row top-ups, epilogues) read their fields through this
// view: the loads then happen where they are written, instead of every field being loaded at kernel entry and held in
// SGPRs across the rollout loops, where the hot paths' own uniforms already fill the scalar file (spills show up as
// v_readlane / v_writelane traffic inside the loops).
__device__ __forceinline__ const DevParams &cold_params() {
#if defined(__HIP_DEVICE_COMPILE__)
    const void *q = (const void *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
#else
    const void *q = nullptr;   // host pass: never executed
#endif
    return *reinterpret_cast<const DevParams *>(q);
}

// ---------------------------------------------------------------------------------------------------------
// MT19937, circular form.  At cursor k, entries < k belong to the next block, entries >= k to the current
// one -- exactly the intermediate states of NumPy's in-place block twist -- so outputs are bit-identical to
// numpy.random.RandomState.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mt_mix(unsigned cur, unsigned nxt, unsigned far) {
    unsigned y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
// Ends a rarely-taken branch that issued vector memory operations: with nothing left in flight on that arm, the
// compiler's wait-count bookkeeping at the join is exactly the common path's (otherwise it drains everything --
// including the step's own stores -- before the next use of any loaded value).
__device__ __forceinline__ void drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0), gfx9 encoding

__device__ __forceinline__ int wrap624(int v) { return v >= MT_N ? v - MT_N : v; }  // v < 2*624

// every store to the circular state also refreshes the mirror of words 0..15 (second store duplicates the first
// when the word has no mirror: unconditional, so no branch is introduced around vector memory operations)
__device__ __forceinline__ void mt_store(unsigned *mt, int idx, unsigned v) {
    mt[idx] = v;
    mt[idx < MT_PAD ? MT_N + idx : idx] = v;
}

// Reset draws its random numbers 16 "attempts" at a time: lane l of the group rebuilds the four stream words
// pos+4l .. pos+4l+3 (one memory round trip for the whole group; 64 words < 227, so they are independent of each
// other) and turns them into the two uniforms one polar-gaussian attempt (or one uniform target) consumes.  Only
// the prefix of the batch that the reference's sequential algorithm would have consumed is committed.
struct AttemptBatch {
    unsigned nw[4];
    double u1, u2;  // np.random.rand() #2l and #2l+1 of the batch

    // `ahead` = words at `pos` that are already twisted (their stored value IS the new word)
    // use_pre (group-uniform; only with ahead >= 4 * G): the batch's words were fetched ahead of time into pre[0..3]
    __device__ __forceinline__ void generate(const unsigned *mt, int pos, int l, int ahead, bool use_pre = false,
                                             const unsigned *pre = nullptr) {
        const int i0 = wrap624(pos + 4 * l);
        unsigned tw[4];
        if (ahead >= 4 * G) {   // group-uniform, the usual case: the whole batch was twisted ahead of time
#pragma unroll
            for (int q = 0; q < 4; q++) {
                nw[q] = use_pre ? pre[q] : mt[wrap624(i0 + q)];
                tw[q] = mt_temper(nw[q]);
            }
        } else {
            unsigned cur[5], far[4];
#pragma unroll
            for (int q = 0; q < 5; q++) cur[q] = mt[wrap624(i0 + q)];
#pragma unroll
            for (int q = 0; q < 4; q++) far[q] = mt[wrap624(wrap624(i0 + MT_M) + q)];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                nw[q] = 4 * l + q < ahead ? cur[q] : mt_mix(cur[q], cur[q + 1], far[q]);
                tw[q] = mt_temper(nw[q]);
            }
        }
        // numpy random_sample: 53-bit double from two words
        u1 = ((double)(tw[0] >> 5) * 67108864.0 + (double)(tw[1] >> 6)) / 9007199254740992.0;
        u2 = ((double)(tw[2] >> 5) * 67108864.0 + (double)(tw[3] >> 6)) / 9007199254740992.0;
    }
    // write back the first `words` words of the batch (those that were twisted here: the first `ahead` are in place)
    __device__ __forceinline__ void commit(unsigned *mt, int pos, int l, int words, int ahead) const {
        if (ahead >= words) return;   // group-uniform
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (4 * l + q < words && 4 * l + q >= ahead) mt_store(mt, wrap624(wrap624(pos + 4 * l) + q), nw[q]);
    }
};

// index of the k-th (0-based) set bit of a 16-bit mask, 16 if there is none: binary search on popcounts (24 instructions;
// the 16-step scan it replaces was 64, twice per attempt batch of every reset)
__device__ __forceinline__ int kth_set_bit16(unsigned mask, int k) {
    const bool none = k < 0 || __popc(mask & 0xffffu) <= k;
    unsigned m = mask & 0xffffu;
    int sel = 0;
    int c = __popc(m & 0xffu);
    bool hi = k >= c;
    sel += hi ? 8 : 0;
    k -= hi ? c : 0;
    m = hi ? m >> 8 : m;
    c = __popc(m & 0xfu);
    hi = k >= c;
    sel += hi ? 4 : 0;
    k -= hi ? c : 0;
    m = hi ? m >> 4 : m;
    c = __popc(m & 0x3u);
    hi = k >= c;
    sel += hi ? 2 : 0;
    k -= hi ? c : 0;
    m = hi ? m >> 2 : m;
    c = (int)(m & 1u);
    hi = k >= c;
    sel += hi ? 1 : 0;
    return none ? 16 : sel;
}

struct __attribute__((packed, aligned(4))) U4 { unsigned x, y, z, w; };  // 4-byte-aligned 16-byte access
struct __attribute__((packed, aligned(4))) U2 { unsigned x, y; };

// ---------------------------------------------------------------------------------------------------------
// Hit tape.  The detection pass only ever asks of a draw whether `rand() <= detect_prob`, so the 312 draws (word
// pairs) of an MT19937 row that has been twisted ahead of its cursor boil down to 312 bits.  cs_mt_advance (the pre-pass
// of the rollout kernels) writes them next to the row (cs_layout.tape_off); the rollout kernels keep the tape of an env
// in ten registers, read draw r as bit r and shift the tape by the number of draws a step consumed: no MT19937 word
// is loaded, mixed or tempered inside their loops.  Resets, which need the uniforms themselves, read the twisted words.
// ---------------------------------------------------------------------------------------------------------
constexpr int TAPE_DW = 10;        // 320 hit bits >= the 312 draw slots (word pairs) of one MT19937 row
constexpr int TAPE_STRIDE = CS_TAPE_STRIDE;  // dwords per env: 10 of bits | base lo, hi | K lo, hi | 2 unused

// np.random.rand() <= detect_prob for the draw made of stream words (wa, wb), exactly, in integers
__device__ __forceinline__ bool draw_hits(const DevParams &p, unsigned wa, unsigned wb) {
    // u = (a >> 5) * 2^26 + (b >> 6) <= K: decided by the first word unless its 27 bits equal K's top 27 (2^-27 of draws)
    const unsigned hi = mt_temper(wa) >> 5, khi = (unsigned)(p.detect_K >> 26);
    if (hi != khi) return hi < khi;
    return (mt_temper(wb) >> 6) <= (unsigned)(p.detect_K & 0x3ffffffull);
}

// One wavefront, one env's MT19937 row held in LDS (`row`, 624 words): twist every word that is not yet twisted ahead of
// the cursor -- words pos + a .. pos + 623 -- in place, and store the new words to the state blob `m` (mirror included).
// Super-batches of 192 words: word j needs the stored words j, j+1, j+397, none of which another word of the same
// super-batch writes (192 <= 227); within a wavefront LDS operations complete in order.
__device__ __forceinline__ void row_twist_ahead(unsigned *row, unsigned *m, int pos, int a, int lane) {
    while (a < MT_N) {   // wave-uniform
        const int r = MT_N - a < 192 ? MT_N - a : 192;
        const int g = wrap624(pos + a);
        unsigned nw[3];
        int idx[3];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int j = wrap624(g + 64 * c + lane);
            idx[c] = j;
            nw[c] = mt_mix(row[j], row[wrap624(j + 1)], row[wrap624(j + MT_M)]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (64 * c + lane < r) {
                row[idx[c]] = nw[c];
                mt_store(m, idx[c], nw[c]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        a += r;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct RowRegs {
    unsigned w[10];   // lane l: words l + 64 i of the row
};

__device__ __forceinline__ void row_load(const unsigned *m, int lane, RowRegs &r) {
#pragma unroll
    for (int i = 0; i < 10; i++) r.w[i] = lane + 64 * i < MT_N ? m[lane + 64 * i] : 0u;
}

// row registers -> LDS (`row`: 624 words owned by this wavefront)
__device__ __forceinline__ void row_to_lds(const RowRegs &r, unsigned *row, int lane) {
#pragma unroll
    for (int i = 0; i < 10; i++)
        if (lane + 64 * i < MT_N) row[lane + 64 * i] = r.w[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// hit bits of draw slots 64 * it .. 64 * it + 63 from the cursor of a fully twisted row in LDS (slot r = words pos + 2r,
// pos + 2r + 1; pos is even, so the pair never straddles the end of the row)
__device__ __forceinline__ unsigned long long row_slot_hits(const DevParams &p, const unsigned *row, int pos, int it, int lane) {
    const int r = 64 * it + lane;
    bool hit = false;
    if (2 * r < MT_N) {
        const int i0 = wrap624(pos + 2 * r);
        hit = draw_hits(p, row[i0], row[i0 + 1]);
    }
    return __ballot(hit);
}

// The hit tape of one lane: bit r = "draw slot r from the cursor hits".  Shift by n slots (n < 320), dword barrel first.
template <int MAX_DW>
__device__ __forceinline__ void tape_shift(unsigned (&t)[TAPE_DW], int n) {
    const int dw = n >> 5, bit = n & 31;
#pragma unroll
    for (int st = 1; st <= MAX_DW; st <<= 1) {
#pragma unroll
        for (int k = 0; k < TAPE_DW; k++) {
            const unsigned from = k + st < TAPE_DW ? t[k + st] : 0u;
            t[k] = (dw & st) ? from : t[k];
        }
    }
#pragma unroll
    for (int k = 0; k < TAPE_DW; k++)
        t[k] = __builtin_amdgcn_alignbit(k + 1 < TAPE_DW ? t[k + 1] : 0u, t[k], (unsigned)bit);
}

// An env's hit tape from the state blob, shifted to the env's cursor; returns false when the stored tape does not
// describe the words twisted ahead of this cursor (never built, other detect_prob, ...): the caller rebuilds it.
// the env's stored tape record: requested early (with the rest of the state), interpreted once it is needed
struct TapeRaw {
    U4 t0, t1, t2, t3;
};
__device__ __forceinline__ TapeRaw tape_fetch(const DevParams &p, int b) {
    const U4 *tp = reinterpret_cast<const U4 *>(p.tape + (size_t)b * TAPE_STRIDE);
    return TapeRaw{tp[0], tp[1], tp[2], tp[3]};
}
template <class EnvT>
__device__ __forceinline__ bool tape_finish(const DevParams &p, const TapeRaw &r, const EnvT &e, unsigned (&tape)[TAPE_DW]) {
    tape[0] = r.t0.x; tape[1] = r.t0.y; tape[2] = r.t0.z; tape[3] = r.t0.w;
    tape[4] = r.t1.x; tape[5] = r.t1.y; tape[6] = r.t1.z; tape[7] = r.t1.w;
    tape[8] = r.t2.x; tape[9] = r.t2.y;
    const unsigned long long base = (unsigned long long)r.t2.z | ((unsigned long long)r.t2.w << 32);
    const unsigned long long K = (unsigned long long)r.t3.x | ((unsigned long long)r.t3.y << 32);
    const unsigned long long used = e.words - base;   // words consumed since the tape was written
    const bool ok = K == p.detect_K && e.words >= base && used + (unsigned long long)e.ahead <= (unsigned long long)MT_N;
    tape_shift<8>(tape, ok ? (int)(used >> 1) : 0);
    return ok;
}
template <class EnvT>
__device__ __forceinline__ bool tape_load(const DevParams &p, int b, const EnvT &e, unsigned (&tape)[TAPE_DW]) {
    return tape_finish(p, tape_fetch(p, b), e, tape);
}

// ---------------------------------------------------------------------------------------------------------
// Correctly rounded sin/cos of an accumulated heading (see gen_trig_table.py).  T points at the LDS copy.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void trig_heading(const double *T, double yaw, double &s, double &c) {
    int k = (int)(yaw * 5.729577951308232 + 0.5);  // 18/pi
    k = k < 0 ? 0 : (k > 36 ? 36 : k);
    const double *r = T + k * TRIG_COLS;
    double t = yaw - r[0];  // exact (Sterbenz) for headings on the pi/18 grid
    double dh = t - r[1];
    double bb = dh - t;
    double err = (t - (dh - bb)) + ((-r[1]) - bb);  // TwoSum tail
    double dl = err - r[2];
    s = r[3] + ((r[4] + dh * (r[5] - 0.5 * dh * r[3])) + dl * r[5]);
    c = r[5] + ((r[6] - dh * (r[3] + 0.5 * dh * r[5])) - dl * r[3]);
    if (fabs(dh) > 1e-6) {
        // off-grid heading (only reachable by editing the raw state; |dh| <= pi/36): angle-addition about the
        // nearest grid heading with Taylor series in dh, ~1 ulp
        const double d2 = dh * dh;
        const double sd = dh * (1.0 + d2 * (-1.0 / 6 + d2 * (1.0 / 120 + d2 * (-1.0 / 5040 + d2 * (1.0 / 362880)))));
        const double cd = 1.0 + d2 * (-0.5 + d2 * (1.0 / 24 + d2 * (-1.0 / 720 + d2 * (1.0 / 40320 + d2 * (-1.0 / 3628800)))));
        s = r[3] * cd + r[5] * sd;
        c = r[5] * cd - r[3] * sd;
    }
}

__device__ __forceinline__ void load_trig_to_lds(double *T) {
    // every launch starts with this round trip: all of a thread's loads in flight together (workgroups of >= 128 threads)
    constexpr int NT = TRIG_ROWS * TRIG_COLS, PER = (NT + 127) / 128;
    const double *src = &g_trig[0][0];
    double v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = threadIdx.x + k * blockDim.x;
        v[k] = src[i < NT ? i : NT - 1];
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = threadIdx.x + k * blockDim.x;
        if (i < NT) T[i] = v[k];
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// Per-env register state (group-uniform values are replicated in all 16 lanes).
// ---------------------------------------------------------------------------------------------------------
template <int N>
struct Env {
    double ax[N], ay[N], yaw[N], cs[N], sn[N];  // cs/sn: cos/sin of the CURRENT yaw (what get_obs emits)
    double tx, ty;                               // this lane's target
    float ntx, nty;                              // its normalised coordinates as get_state emits them
    unsigned found, newly, newly_reset;          // bit masks over targets
    int target_find, flags, time_step, total_reward, mt_pos, episodes, curr_reward;
    int ahead;                                   // pre-twisted words at mt_pos (cs_layout.ahead_off)
    unsigned long long words;
};

template <int N>
__device__ __forceinline__ void norm_target(const DevParams &p, Env<N> &e) {
    // (t - 0.5*map_size)/(map_size/2), flight_env_easy.py:211; fp32 output, so the fp64 quotient is replaced by a
    // product with the reciprocal (differs from the quotient's fp32 rounding in ~1e-9 of cases, tolerance 1e-6)
    e.ntx = (float)((e.tx - p.mid) * p.inv_half);
    e.nty = (float)((e.ty - p.mid) * p.inv_half);
}

// The group's window into the circular MT19937 state: lane l holds words pos+l and pos+397+l.

struct MtWin {
    unsigned cur, far;
};
__device__ __forceinline__ MtWin mt_prefetch(const unsigned *mt, int pos, int l) {
    MtWin w;
    w.cur = mt[wrap624(pos + l)];
    w.far = mt[wrap624(wrap624(pos + MT_M) + l)];
    return w;
}

template <int N>
__device__ __forceinline__ void env_trig(const double *T, Env<N> &e) {
#pragma unroll
    for (int i = 0; i < N; i++) trig_heading(T, e.yaw[i], e.sn[i], e.cs[i]);
}

template <int N>
__device__ __forceinline__ void env_load(const DevParams &p, int b, int t, Env<N> &e) {
    const int4 *h4 = reinterpret_cast<const int4 *>(p.hdr + (size_t)b * CS_H_WORDS);
    int4 h0 = h4[0], h1 = h4[1], h2 = h4[2];
    e.found = (unsigned)h0.x;
    e.newly = (unsigned)h0.y;
    e.target_find = h0.z;
    e.flags = h0.w;
    e.time_step = h1.x;
    e.total_reward = h1.y;
    e.mt_pos = h1.z;
    e.episodes = h1.w;
    e.words = (unsigned long long)(unsigned)h2.x | ((unsigned long long)(unsigned)h2.y << 32);
    e.curr_reward = h2.z;
    e.newly_reset = (unsigned)h2.w;
    e.ahead = p.ahead[b];
    const double4 *a4 = reinterpret_cast<const double4 *>(p.agent + (size_t)b * CS_MAX_AGENTS * 4);
#pragma unroll
    for (int i = 0; i < N; i++) {
        double4 a = a4[i];
        e.ax[i] = a.x;
        e.ay[i] = a.y;
        e.yaw[i] = a.z;
        e.cs[i] = 0.0;  // filled by kinematics / env_reset / env_trig before any emission
        e.sn[i] = 0.0;
    }
    const double2 *t2 = reinterpret_cast<const double2 *>(p.tgt + (size_t)b * G * 2);
    double2 tt = t2[t];
    e.tx = tt.x;
    e.ty = tt.y;
    norm_target(p, e);
}

template <int N>
__device__ __forceinline__ void env_store(const DevParams &p, int b, int t, const Env<N> &e, bool store_targets) {
    if (t == 0) {
        int4 *h4 = reinterpret_cast<int4 *>(p.hdr + (size_t)b * CS_H_WORDS);
        h4[0] = make_int4((int)e.found, (int)e.newly, e.target_find, e.flags);
        h4[1] = make_int4(e.time_step, e.total_reward, e.mt_pos, e.episodes);
        h4[2] = make_int4((int)(unsigned)(e.words & 0xffffffffull), (int)(unsigned)(e.words >> 32), e.curr_reward,
                          (int)e.newly_reset);
        p.ahead[b] = e.ahead;
    }
    double4 *a4 = reinterpret_cast<double4 *>(p.agent + (size_t)b * CS_MAX_AGENTS * 4);
#pragma unroll
    for (int i = 0; i < N; i++)
        if (t == i) a4[i] = make_double4(e.ax[i], e.ay[i], e.yaw[i], 0.0);
    if (store_targets) {
        double2 *t2 = reinterpret_cast<double2 *>(p.tgt + (size_t)b * G * 2);
        t2[t] = make_double2(e.tx, e.ty);
    }
}

// flight: everything the map sweep needs of the step (or reset) that ran before it, as a record of its own, so that a
// sweep for step t can run beside the kinematics / detection of step t + 1 (which overwrite hdr / agent / tgt).  Two
// records per env, selected by the launch's parity.
struct MapJob {
    int flags;                  // FLAG_DIRTY / FLAG_RESET_PASS of this step
    unsigned newly, newly_reset;
    int pad;
    int cell[CS_MAX_TARGETS];   // flat map cell of every target (int() truncation, clamped: flight_env.py:279), -1 = none
    double axy[CS_MAX_AGENTS][2];
    char fill[CS_JOB_BYTES - 16 - 4 * CS_MAX_TARGETS - 16 * CS_MAX_AGENTS];
};
static_assert(sizeof(MapJob) == CS_JOB_BYTES, "MapJob layout");

__device__ __forceinline__ MapJob *job_ptr(const DevParams &p, int parity, int b) {
    return reinterpret_cast<MapJob *>(p.job) + (size_t)parity * p.B + b;
}

template <int N>
__device__ __forceinline__ void job_store(const DevParams &p, int parity, int b, int t, const Env<N> &e) {
    MapJob *j = job_ptr(p, parity, b);
    if (t == 0) *reinterpret_cast<int4 *>(j) = make_int4(e.flags & (FLAG_DIRTY | FLAG_RESET_PASS), (int)e.newly, (int)e.newly_reset, 0);
    int cell = -1;
    if (t < p.n_targets) {
        int ix = (int)e.tx, iy = (int)e.ty;  // int(): truncation toward zero, flight_env.py:279
        ix = ix < p.map_size - 1 ? ix : p.map_size - 1;
        iy = iy < p.map_size - 1 ? iy : p.map_size - 1;
        cell = (ix >= 0 && iy >= 0) ? ix * p.map_size + iy : -1;
    }
    j->cell[t] = cell;
#pragma unroll
    for (int i = 0; i < N; i++)
        if (t == i) *reinterpret_cast<double2 *>(j->axy[i]) = make_double2(e.ax[i], e.ay[i]);
}

// ---------------------------------------------------------------------------------------------------------
// Detection pass + reward: flight_env_easy.py:223-253 (_update_obs), flight_env.py:232-266.
// Returns curr_reward.  gshift = 16 * (group index inside the wavefront); win = the group's prefetched window
// at e.mt_pos.
//
// One np.random.rand() is consumed per in-range (agent, target) pair, found or not (quirk Q4), in agent-major
// order: a wavefront ballot gives the in-range mask, a prefix popcount of the group's 16-bit slice gives each
// pair its rank r, i.e. stream words pos+2r and pos+2r+1.  Circular MT19937: word k is rebuilt from words k,
// k+1 and k+397 (== k-227), so the <= 2*n*m <= 226 words of one pass (n <= 7; n = 8 draws in two halves on the
// slow path) can all be rebuilt from the state as it stood before the pass.  Lane l rebuilds word l of the
// window once; ranks 0..6 are served by two wave shuffles each, and the consumed prefix is committed with one
// coalesced store.  Ranks >= 7 (more than 7 pairs in range at once) take the direct-load path.
// ---------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ int detect_finish(const DevParams &p, int t, int gshift, Env<N> &e, bool hit);

template <int N>
__device__ __forceinline__ int detect_pass(const DevParams &p, int b, int t, int gshift, Env<N> &e, MtWin win) {
    const bool is_tgt = t < p.n_targets;
    unsigned *mt = p.mt + (size_t)b * MT_STRIDE;
    bool inr[N];
    int rank[N];
    int base = 0;
    const unsigned below = (1u << t) - 1u;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double ddx = e.tx - e.ax[i], ddy = e.ty - e.ay[i];
        inr[i] = is_tgt && (ddx * ddx + ddy * ddy <= p.view_r2);  // (t_x-x)**2 + (t_y-y)**2 <= view_range**2
        unsigned gm = (unsigned)((__ballot(inr[i]) >> gshift) & 0xffffull);
        rank[i] = base + __popc(gm & below);  // agent-major order of the reference's double loop
        base += __popc(gm);
    }
    // window words 0..14 rebuilt in lanes 0..14
    // (words below e.ahead were twisted ahead of time by the lane kernel: their stored value is the new word)
    const unsigned nxt = (unsigned)__shfl((int)win.cur, t + 1, G);
    const unsigned nw = t < e.ahead ? win.cur : mt_mix(win.cur, nxt, win.far);
    const unsigned tw = mt_temper(nw);
    bool hit = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int src = rank[i] < 7 ? 2 * rank[i] : 0;
        const unsigned wa = (unsigned)__shfl((int)tw, src, G);
        const unsigned wb = (unsigned)__shfl((int)tw, src + 1, G);
        const unsigned long long u = ((unsigned long long)(wa >> 5) << 26) | (unsigned long long)(wb >> 6);
        hit = hit || (inr[i] && rank[i] < 7 && u <= p.detect_K);  // prob <= self.detect_prob, exact in integers
    }
    const int fast_words = base < 7 ? 2 * base : 14;
    if (t < fast_words) mt_store(mt, wrap624(e.mt_pos + t), nw);
    if (base > 7) {  // rare: direct loads for the ranks the window does not cover
        constexpr int PHASES = (N * G * 2 > 226) ? 2 : 1;
        constexpr int PER = (N + PHASES - 1) / PHASES;
#pragma unroll
        for (int ph = 0; ph < PHASES; ph++) {
            unsigned w[PER][5];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int i = ph * PER + k;
                if (i < N && inr[i] && rank[i] >= 7) {
                    int i0 = wrap624(e.mt_pos + 2 * rank[i]);
                    int i1 = wrap624(i0 + 1), i2 = wrap624(i0 + 2);
                    w[k][0] = mt[i0];
                    w[k][1] = mt[i1];
                    w[k][2] = mt[i2];
                    w[k][3] = mt[wrap624(i0 + MT_M)];
                    w[k][4] = mt[wrap624(i1 + MT_M)];
                }
            }
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int i = ph * PER + k;
                if (i < N && inr[i] && rank[i] >= 7) {
                    int i0 = wrap624(e.mt_pos + 2 * rank[i]);
                    int i1 = wrap624(i0 + 1);
                    const int ah = e.ahead;  // still the value on entry: positions are relative to the pass's cursor
                    unsigned n0 = 2 * rank[i] < ah ? w[k][0] : mt_mix(w[k][0], w[k][1], w[k][3]);
                    unsigned n1 = 2 * rank[i] + 1 < ah ? w[k][1] : mt_mix(w[k][1], w[k][2], w[k][4]);
                    mt_store(mt, i0, n0);
                    mt_store(mt, i1, n1);
                    unsigned long long u =
                        ((unsigned long long)(mt_temper(n0) >> 5) << 26) | (unsigned long long)(mt_temper(n1) >> 6);
                    hit = hit || (u <= p.detect_K);
                }
            }
        }
        drain_vmem();
    }
    e.mt_pos = wrap624(e.mt_pos + 2 * base);
    e.words += (unsigned long long)(2 * base);
    e.ahead = e.ahead > 2 * base ? e.ahead - 2 * base : 0;
    return detect_finish<N>(p, t, gshift, e, hit);
}

// Second half of a detection pass: `hit` = this lane's target was detected by some agent (flight_env_easy.py:238-247).
template <int N>
__device__ __forceinline__ int detect_finish(const DevParams &p, int t, int gshift, Env<N> &e, bool hit) {
    bool lane_new = hit && !((e.found >> t) & 1u);
    unsigned newly = (unsigned)((__ballot(lane_new) >> gshift) & 0xffffull);
    int cnt = __popc(newly);
    int r = -1;  // MOVE_COST
    r += 10 * cnt;  // FIND_ONE_TGT
    e.found |= newly;
    e.newly = newly;
    e.target_find += cnt;
    if (cnt > 0 && e.target_find == p.n_targets && !(e.flags & FLAG_WIN)) {
        r += 100;  // FIND_ALL_TGT
        e.flags |= FLAG_WIN;
    }
    r -= __popc(((unsigned)e.flags >> 8) & 0xffu);  // OUT_PUNISH per agent with out_flag set
    e.curr_reward = r;
    e.flags |= FLAG_DIRTY;
    return r;
}

// Detection pass of the rollout kernels: same contract as detect_pass, draws read from the env's hit tape (replicated
// in the 16 lanes of the group).  The pair of rank r takes draw slot r, i.e. bit r of the tape; afterwards the tape
// is shifted by the number of draws.  When the tape does not cover the pass (no pre-pass ran, or the env has drawn more
// than a row's worth since), the pass runs the on-demand path above on a freshly loaded window.
template <int N>
__device__ __forceinline__ int detect_pass_tape(const DevParams &p, int b, int t, int gshift, Env<N> &e,
                                                unsigned (&tape)[TAPE_DW], bool tape_ok) {
    constexpr int MAXDW = (N * CS_MAX_TARGETS) / 32 < 1 ? 1 : (N * CS_MAX_TARGETS) / 32;   // draws of one pass, in dwords
    const bool is_tgt = t < p.n_targets;
    bool inr[N];
    int rank[N];
    int base = 0;
    const unsigned below = (1u << t) - 1u;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double ddx = e.tx - e.ax[i], ddy = e.ty - e.ay[i];
        inr[i] = is_tgt && (ddx * ddx + ddy * ddy <= p.view_r2);  // (t_x-x)**2 + (t_y-y)**2 <= view_range**2
        unsigned gm = (unsigned)((__ballot(inr[i]) >> gshift) & 0xffffull);
        rank[i] = base + __popc(gm & below);  // agent-major order of the reference's double loop
        base += __popc(gm);
    }
    if (!(tape_ok && 2 * base <= e.ahead)) {   // group-uniform
        const int r = detect_pass<N>(p, b, t, gshift, e, mt_prefetch(p.mt + (size_t)b * MT_STRIDE, e.mt_pos, t));
        drain_vmem();
        tape_shift<MAXDW>(tape, base);
        return r;
    }
    bool hit = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        unsigned w = tape[0];
#pragma unroll
        for (int d = 1; d < (N * CS_MAX_TARGETS + 31) / 32; d++) w = (rank[i] >> 5) == d ? tape[d] : w;
        hit = hit || (inr[i] && ((w >> (rank[i] & 31)) & 1u));
    }
    e.mt_pos = wrap624(e.mt_pos + 2 * base);
    e.words += (unsigned long long)(2 * base);
    e.ahead -= 2 * base;
    tape_shift<MAXDW>(tape, base);
    return detect_finish<N>(p, t, gshift, e, hit);
}

// Prologue of the 16-lane rollout kernels: the wavefront tops up the MT19937 rows of its (up to) four envs that have
// fewer than `min_ahead` twisted words left or no matching tape -- whole wavefront on one row at a time, exactly what
// k_mt_advance does, but without a launch of its own -- and hands the new tape to the env's 16 lanes by ballot.
template <int N>
__device__ __forceinline__ void group_wave_advance(const DevParams &p, int wave_b0, int nvalid, int lane, int min_ahead,
                                                   unsigned *rowbuf, Env<N> &e, unsigned (&tape)[TAPE_DW], bool &tape_ok) {
    const int grp = lane >> 4;
#pragma unroll 1
    for (int g = 0; g < 4; g++) {
        const int pos = __shfl(e.mt_pos, 16 * g), a = __shfl(e.ahead, 16 * g);
        const int ok = __shfl(tape_ok ? 1 : 0, 16 * g);
        if (g >= nvalid || (ok && a >= min_ahead)) continue;   // wave-uniform
        unsigned *m = p.mt + (size_t)(wave_b0 + g) * MT_STRIDE;
        RowRegs rr;
        row_load(m, lane, rr);
        row_to_lds(rr, rowbuf, lane);
        row_twist_ahead(rowbuf, m, pos, a < 0 ? 0 : a, lane);
#pragma unroll
        for (int it = 0; it < TAPE_DW / 2; it++) {
            const unsigned long long bm = row_slot_hits(p, rowbuf, pos, it, lane);
            if (grp == g) {
                tape[2 * it] = (unsigned)(bm & 0xffffffffull);
                tape[2 * it + 1] = (unsigned)(bm >> 32);
            }
        }
        if (grp == g) {
            e.ahead = MT_N;
            tape_ok = true;
        }
    }
    drain_vmem();
}

// ... and their epilogue: the tape (in registers, aligned to the cursor) goes back to the state blob for the next launch
template <int N>
__device__ __forceinline__ void group_tape_store(const DevParams &p, int b, int t, const Env<N> &e, const unsigned (&tape)[TAPE_DW]) {
    U4 *tp = reinterpret_cast<U4 *>(p.tape + (size_t)b * TAPE_STRIDE);
    if (t == 0) tp[0] = U4{tape[0], tape[1], tape[2], tape[3]};
    if (t == 1) tp[1] = U4{tape[4], tape[5], tape[6], tape[7]};
    if (t == 2) tp[2] = U4{tape[8], tape[9], (unsigned)(e.words & 0xffffffffull), (unsigned)(e.words >> 32)};
    if (t == 3) tp[3] = U4{(unsigned)(p.detect_K & 0xffffffffull), (unsigned)(p.detect_K >> 32), 0u, 0u};
}

// ---------------------------------------------------------------------------------------------------------
// Kinematics: flight_env_easy.py:255-301 (_agent_step + _potential_energy_force), flight_env.py:305-355.
// The reference is sequential over agents (quirk Q7: agent i is repelled from the already-moved agents j < i),
// but only the force and the wall test carry that dependency.  Phase 1 therefore computes, for all agents at
// once and branch-free (instruction-level parallelism instead of 2n dependent chains), the new heading, its
// sin/cos, the unforced move, and the sin/cos of the reflected heading a wall hit would select.  Phase 2 is the
// short sequential part: force (a rare branch with the two fp64 divisions), wall test, selects.  Every lane of
// the group computes the same values.  Operation order per coordinate is the reference's: (x + v*cos) + f_x.
// ---------------------------------------------------------------------------------------------------------
template <int N, int VARIANT, class EnvT>
__device__ __forceinline__ void kinematics(const DevParams &p, const double *T, const int (&act)[N], EnvT &e) {
    const double PI = 3.141592653589793, TWO_PI = 2.0 * 3.141592653589793, THREE_PI = 3.0 * 3.141592653589793;
    const double DYAW = 3.141592653589793 / 18.0;
    double yw[N], s1[N], c1[N], yr[N], s2[N], c2[N], xt[N], yt[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        double yaw = e.yaw[i];
        yaw = act[i] == 1 ? yaw + DYAW : (act[i] == 2 ? yaw + -DYAW : yaw);  // dyaw = [0, pi/18, -pi/18][act]
        yaw = yaw > TWO_PI ? yaw - TWO_PI : (yaw < 0.0 ? yaw + TWO_PI : yaw);
        yw[i] = yaw;
        trig_heading(T, yaw, s1[i], c1[i]);
        yr[i] = (yaw <= PI) ? PI - yaw : THREE_PI - yaw;
        trig_heading(T, yr[i], s2[i], c2[i]);
        xt[i] = e.ax[i] + p.velocity * c1[i];
        yt[i] = e.ay[i] + p.velocity * s1[i];
    }
    unsigned out = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double x0 = e.ax[i], y0 = e.ay[i];
        double fx = 0.0, fy = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            double xa = e.ax[j], ya = e.ay[j];  // already moved if j < i
            double d2 = (xa - x0) * (xa - x0) + (ya - y0) * (ya - y0);
            if (d2 < p.force_d2 && (xa != x0 || ya != y0)) {
                double den = (x0 - xa) * (x0 - xa) + (y0 - ya) * (y0 - ya);
                fx += p.force_k * (x0 - xa) / den;
                fy += p.force_k * (y0 - ya) / den;
            }
        }
        double x = xt[i] + fx;
        double y = yt[i] + fy;
        const bool hit = VARIANT == 1 ? (x < 0.0 || x >= p.L || y < 0.0 || y >= p.L)   // flight_env.py:328
                                      : (x < 0.0 || x > p.L || y < 0.0 || y > p.L);    // flight_env_easy.py:278
        e.ax[i] = hit ? fmin(fmax(x, 0.0), p.L) : x;
        e.ay[i] = hit ? fmin(fmax(y, 0.0), p.L) : y;
        e.yaw[i] = hit ? yr[i] : yw[i];
        e.cs[i] = hit ? c2[i] : c1[i];
        e.sn[i] = hit ? s2[i] : s1[i];
        out |= hit ? (1u << i) : 0u;
    }
    e.flags = (e.flags & ~0xff00) | (int)(out << 8);
}

// Per-wavefront LDS staging tile for the group kernels: the 4 envs of a wavefront deposit their get_state rows,
// obs features and step outputs here, then all 64 lanes write them out as contiguous dwords.  Every global store
// of a step is thereby unconditional and sits in one straight-line block, so the compiler's vmcnt bookkeeping is
// exact and a prefetched load is never waited for together with the step's own stores.
constexpr int TILE_W = 4 * CS_MAX_AGENTS + 3 * CS_MAX_TARGETS;  // widest get_state row (80 floats)
struct WaveTile {
    float row[4][TILE_W];
    float reward[4];
    int term[4], win[4];
    int pad[4];
    double2 trig[4][2 * CS_MAX_AGENTS];  // (sin, cos) of the 2n headings of a step, per group (kinematics_group)
};

// ---------------------------------------------------------------------------------------------------------
// Group version of the kinematics.
//  * The 2n heading evaluations of a step (new heading and its wall reflection, per agent) are spread over the
//    group's lanes -- lane 2i takes agent i's heading, lane 2i+1 its reflection -- and published through the
//    wavefront's LDS tile, instead of every lane evaluating all 2n (2n <= 16 = lanes).
//  * Fast path: the repulsion is zero unless two agents are within force_dist (3-14 % of env-steps), so all agents
//    are first moved as if it were zero -- branch-free, all agents in parallel -- and every ordered pair (i, j) is
//    then tested exactly as the reference would test it (agent i's PRE-move position against j's already-moved
//    position if j < i, else j's old one).  If no pair is in range the reference's sequential loop would have
//    added f = 0 everywhere and the tentative result IS its result (x + 0.0 kept, so even signed zeros agree);
//    otherwise the group falls back to the sequential code.  Operation order per coordinate is the reference's:
//    (x + v*cos) + f_x.
// ---------------------------------------------------------------------------------------------------------
template <int N, int VARIANT>
__device__ __forceinline__ void kinematics_group(const DevParams &p, const double *T, WaveTile &tile, const int (&act)[N],
                                                 int t, int grp, Env<N> &e) {
    const double PI = 3.141592653589793, TWO_PI = 2.0 * 3.141592653589793, THREE_PI = 3.0 * 3.141592653589793;
    const double DYAW = 3.141592653589793 / 18.0;
    double yw[N], yr[N];
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double yaw = e.yaw[i];
        yaw = act[i] == 1 ? yaw + DYAW : (act[i] == 2 ? yaw + -DYAW : yaw);  // dyaw = [0, pi/18, -pi/18][act]
        yaw = yaw > TWO_PI ? yaw - TWO_PI : (yaw < 0.0 ? yaw + TWO_PI : yaw);
        yw[i] = yaw;
        yr[i] = (yaw <= PI) ? PI - yaw : THREE_PI - yaw;
        mine = t == 2 * i ? yw[i] : (t == 2 * i + 1 ? yr[i] : mine);
    }
    double ms, mc;
    trig_heading(T, mine, ms, mc);
    if (t < 2 * N) tile.trig[grp][t] = make_double2(ms, mc);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double s1[N], c1[N], s2[N], c2[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double2 a = tile.trig[grp][2 * i], r = tile.trig[grp][2 * i + 1];
        s1[i] = a.x; c1[i] = a.y; s2[i] = r.x; c2[i] = r.y;
    }
    // ---- tentative move of every agent with zero repulsion
    double xf[N], yf[N];
    bool hitf[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double x = (e.ax[i] + p.velocity * c1[i]) + 0.0;
        const double y = (e.ay[i] + p.velocity * s1[i]) + 0.0;
        const bool hit = VARIANT == 1 ? ((x < 0.0) | (x >= p.L) | (y < 0.0) | (y >= p.L))   // flight_env.py:328
                                      : ((x < 0.0) | (x > p.L) | (y < 0.0) | (y > p.L));    // flight_env_easy.py:278
        xf[i] = hit ? fmin(fmax(x, 0.0), p.L) : x;
        yf[i] = hit ? fmin(fmax(y, 0.0), p.L) : y;
        hitf[i] = hit;
    }
    // ---- would the reference have found any pair within force_dist?  (n >= 5: the fallback would run on ~45 % of
    // wavefront-steps, so larger teams go straight to the sequential loop)
    bool need = N > 4;
#pragma unroll
    for (int i = 0; i < (N > 4 ? 0 : N); i++) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            const double xa = j < i ? xf[j] : e.ax[j], ya = j < i ? yf[j] : e.ay[j];
            const double dx = xa - e.ax[i], dy = ya - e.ay[i];
            need = need | ((dx * dx + dy * dy < p.force_d2) & ((xa != e.ax[i]) | (ya != e.ay[i])));
        }
    }
    unsigned out = 0;
    if (!need) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            e.ax[i] = xf[i];
            e.ay[i] = yf[i];
            e.yaw[i] = hitf[i] ? yr[i] : yw[i];
            e.cs[i] = hitf[i] ? c2[i] : c1[i];
            e.sn[i] = hitf[i] ? s2[i] : s1[i];
            out |= hitf[i] ? (1u << i) : 0u;
        }
    } else {  // the reference's sequential loop (quirk Q7); one branch per agent: its pairs are tested branch-free first
#pragma unroll
        for (int i = 0; i < N; i++) {
            const double x0 = e.ax[i], y0 = e.ay[i];
            double fx = 0.0, fy = 0.0;
            bool any = false;
#pragma unroll
            for (int j = 0; j < N; j++) {
                if (j == i) continue;
                const double dx = e.ax[j] - x0, dy = e.ay[j] - y0;  // e.ax[j] already moved if j < i
                any = any | ((dx * dx + dy * dy < p.force_d2) & ((e.ax[j] != x0) | (e.ay[j] != y0)));
            }
            if (any) {
#pragma unroll
                for (int j = 0; j < N; j++) {
                    if (j == i) continue;
                    double xa = e.ax[j], ya = e.ay[j];
                    double d2 = (xa - x0) * (xa - x0) + (ya - y0) * (ya - y0);
                    if (d2 < p.force_d2 && (xa != x0 || ya != y0)) {
                        double den = (x0 - xa) * (x0 - xa) + (y0 - ya) * (y0 - ya);
                        fx += p.force_k * (x0 - xa) / den;
                        fy += p.force_k * (y0 - ya) / den;
                    }
                }
            }
            double x = (x0 + p.velocity * c1[i]) + fx;
            double y = (y0 + p.velocity * s1[i]) + fy;
            const bool hit = VARIANT == 1 ? (x < 0.0 || x >= p.L || y < 0.0 || y >= p.L)
                                          : (x < 0.0 || x > p.L || y < 0.0 || y > p.L);
            e.ax[i] = hit ? fmin(fmax(x, 0.0), p.L) : x;
            e.ay[i] = hit ? fmin(fmax(y, 0.0), p.L) : y;
            e.yaw[i] = hit ? yr[i] : yw[i];
            e.cs[i] = hit ? c2[i] : c1[i];
            e.sn[i] = hit ? s2[i] : s1[i];
            out |= hit ? (1u << i) : 0u;
        }
    }
    e.flags = (e.flags & ~0xff00) | (int)(out << 8);
}

// start position / heading of agent i (flight_env_easy.py:139-180): the same for every env
template <int N>
__device__ __forceinline__ void start_pose(const DevParams &p, int i, double &x, double &y, double &yaw) {
    const double s = N != 1 ? (double)(i * p.map_size) / (double)(N - 1) : p.L / 2.0;
    switch (p.agent_mode) {
    case 0: x = s; y = 0.0; yaw = 3.141592653589793 / 2.0; break;
    case 1: x = s; y = p.L / 2.0; yaw = 3.141592653589793 / 2.0; break;
    case 2: x = 0.0; y = s; yaw = 0.0; break;
    default: x = p.L; y = s; yaw = 3.141592653589793; break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// reset: flight_env_easy.py:79-182 / flight_env.py:83-191.  Group-cooperative; ends with the reset-time
// detection pass (quirk Q3) whose reward is discarded.
// ---------------------------------------------------------------------------------------------------------
#ifdef CS_TIMELINE
__device__ unsigned long long g_blk[1024][8];   // per workgroup: K entry / loop / loop end / exit, D the same
#define BLK_STAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_blk[blockIdx.x][k] = __builtin_readcyclecounter(); } while (0)
__device__ unsigned g_spin[1024][4];   // per workgroup: polls K / D / E spent waiting for the other side
#define SPIN_DECL unsigned spin_count = 0
#define SPIN_TICK (++spin_count)
#define SPIN_STORE(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_spin[blockIdx.x][k] = spin_count; } while (0)
extern __device__ unsigned long long g_stamps[64][16];
#define RT_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_stamps[63][3 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define RT_STAMP(k) do {} while (0)
#define BLK_STAMP(k) do {} while (0)
#define SPIN_DECL do {} while (0)
#define SPIN_TICK ((void)0)
#define SPIN_STORE(k) do {} while (0)
#endif
// Target placement of a reset (flight_env_easy.py:95-134) for the env whose 16-lane group this is: lane t gets target t's
// position in (mx, my); the env's MT19937 cursor / word count / pre-twisted count advance by what the reference's
// sequential algorithm consumes.  No agent state involved: the octet kernel calls this alone.
// use_pre: the first batch's four stream words (per lane) were prefetched into pre[] (octet pair kernel).
__device__ __forceinline__ void reset_targets(const DevParams &p, unsigned *mt, int t, int gshift, int &mt_pos,
                                              unsigned long long &words_total, int &ahead, double &mx, double &my,
                                              bool use_pre = false, const unsigned *pre = nullptr) {
    const unsigned tmask = p.n_targets >= 32 ? ~0u : ((1u << p.n_targets) - 1u);
    mx = 0.0;
    my = 0.0;
    if (p.target_mode == 0) {
        // x = a*cx (+ dx*2*(randn-0.5) for the 'f' rows), flight_env_easy.py:95-113.  np.random.randn is the legacy
        // polar method: attempts (x1, x2) are drawn until 0 < r2 < 1; the pair's SECOND value f*x2 is returned
        // first, f*x1 is cached for the next call -- so the j-th accepted attempt serves the j-th 'f' target.
        double jx = 0.0, jy = 0.0;
        // this lane's entries of the kernel-argument tables, read where they are (the kernarg segment, indexed by lane):
        // as 64 selects the tables sat in SGPRs across the rollout loops and spilled in every kernel that can reset
        // (k_rollout<5>: 1244 -> 441 v_readlane / v_writelane, 253 -> 250 VGPRs)
        RT_STAMP(0);
        mx = p.tx0[t];
        my = p.ty0[t];
        jx = p.jx2[t];
        jy = p.jy2[t];
        asm volatile("" : "+v"(mx), "+v"(my), "+v"(jx), "+v"(jy));
        RT_STAMP(1);
        const unsigned fmask = ~p.deter_mask & tmask;        // jittered targets
        const int need_total = __popc(fmask);
        const bool mine = (fmask >> t) & 1u;
        const int my_rank = __popc(fmask & ((1u << t) - 1u));  // which accepted attempt is mine
        int taken = 0;
        while (taken < need_total) {  // group-uniform; one batch suffices ~99 % of the time for 9 jittered targets
            AttemptBatch ab;
            ab.generate(mt, mt_pos, t, ahead, use_pre && taken == 0, pre);
            asm volatile("" : "+v"(ab.u1), "+v"(ab.u2));
            RT_STAMP(2);
            const double x1 = 2.0 * ab.u1 - 1.0, x2 = 2.0 * ab.u2 - 1.0;
            const double r2 = x1 * x1 + x2 * x2;
            const bool accept = !(r2 >= 1.0 || r2 == 0.0);
            const double f = sqrt(-2.0 * log(accept ? r2 : 0.5) / (accept ? r2 : 0.5));
            double g1 = f * x2, g2 = f * x1;
            asm volatile("" : "+v"(g1), "+v"(g2));
            RT_STAMP(3);
            const unsigned amask = (unsigned)((__ballot(accept) >> gshift) & 0xffffull);
            const int have = __popc(amask);
            const int want = need_total - taken;
            const int k = my_rank - taken;                                  // my index within this batch's accepts
            const int sel = kth_set_bit16(amask, (k >= 0 && k < 16) ? k : 0);
            const double s1 = __shfl(g1, sel & 15, G), s2 = __shfl(g2, sel & 15, G);
            if (mine && k >= 0 && k < have && k < want) {
                mx += jx * (s1 - 0.5);  // dx*2*(randn-0.5)
                my += jy * (s2 - 0.5);
            }
            // words consumed: up to and including the attempt that supplied the last needed pair, else the batch
            const int last = have >= want ? kth_set_bit16(amask, want - 1) : 15;
            const int words = 4 * (last + 1);
            ab.commit(mt, mt_pos, t, words, ahead);
            mt_pos = wrap624(mt_pos + words);
            words_total += (unsigned long long)words;
            ahead = ahead > words ? ahead - words : 0;
            taken += have < want ? have : want;
            RT_STAMP(4);
        }
    } else {
        // x, y = map_size*np.random.rand() per target, flight_env_easy.py:122-127
        AttemptBatch ab;
        ab.generate(mt, mt_pos, t, ahead, use_pre, pre);
        mx = p.L * ab.u1;
        my = p.L * ab.u2;
        const int words = 4 * p.n_targets;
        ab.commit(mt, mt_pos, t, words, ahead);
        mt_pos = wrap624(mt_pos + words);
        words_total += (unsigned long long)words;
        ahead = ahead > words ? ahead - words : 0;
    }
}

#define CS_AS1 __attribute__((address_space(1)))
#define CS_AS4 __attribute__((address_space(4)))
__device__ __forceinline__ const CS_AS4 DevParams *cold_params4() {
#if defined(__HIP_DEVICE_COMPILE__)
    auto q = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return (const CS_AS4 DevParams *)q;
#else
    return nullptr;   // host pass: never executed
#endif
}

// the reset's target tables into LDS: rtab[0..15] = a*cx, [16..31] = a*cy, [32..47] = 2*a*dx, [48..63] = 2*a*dy
// (DevParams::tx0, ty0, jx2, jy2: adjacent in the kernarg segment); one lane per entry
__device__ __forceinline__ void load_reset_tab(double *rtab, int lane) {
    const CS_AS4 DevParams *q = cold_params4();
    static_assert(CS_MAX_TARGETS == G, "one table row per 16 lanes");
    static_assert(offsetof(DevParams, jy2) - offsetof(DevParams, tx0) == 3 * G * sizeof(double), "tables are adjacent");
    rtab[lane] = q->tx0[lane];   // lane 0..63 runs through tx0, ty0, jx2, jy2
}

template <int N>
struct StartTab {
    double x[N], y[N], yaw;
};
template <int N>
__device__ __forceinline__ StartTab<N> start_tab() {
    const CS_AS4 DevParams *q = cold_params4();
    StartTab<N> st;
#pragma unroll
    for (int i = 0; i < N; i++) {
        st.x[i] = q->start_x[i];
        st.y[i] = q->start_y[i];
    }
    st.yaw = q->start_yaw;
    return st;
}
template <int N>
__device__ __forceinline__ void start_pick(const StartTab<N> &st, int i, double &x, double &y) {
    x = st.x[0];
    y = st.y[0];
#pragma unroll
    for (int k = 1; k < N; k++) {
        x = i == k ? st.x[k] : x;
        y = i == k ? st.y[k] : y;
    }
}

// One attempt batch of a reset's target placement from TWISTED words only, for the env whose 16-lane group this is (lane =
// polar attempt / uniform target): the 16-lane form of what oct_place_targets does per round.  `w`: this lane's four stream
// words at cursor + 4 * t16.  Returns true if the batch completes the placement (always for uniform targets; ~99 % of the
// time for the jittered ones): (mx, my) is then target t16's position and `words` the stream words consumed.  Nothing is
// written: a caller whose batch does not suffice falls back to reset_targets() from the untouched cursor.
__device__ __forceinline__ bool reset_batch_twisted(const unsigned (&w_in)[4], double tx0, double ty0, double jx2, double jy2,
                                                    unsigned fmask, int n_targets, int target_mode, double L, int t16,
                                                    int gshift16, double &mx, double &my, int &words) {
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = mt_temper(w_in[k]);
    // numpy random_sample: 53-bit double from two words
    const double u1 = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
    const double u2 = ((double)(w[2] >> 5) * 67108864.0 + (double)(w[3] >> 6)) / 9007199254740992.0;
    mx = tx0;
    my = ty0;
    if (target_mode != 0) {   // x, y = map_size*np.random.rand() per target, flight_env_easy.py:122-127
        mx = L * u1;
        my = L * u2;
        words = 4 * n_targets;
        return true;
    }
    const int need_total = __popc(fmask);
    words = 0;
    if (need_total == 0) return true;
    const double x1 = 2.0 * u1 - 1.0, x2 = 2.0 * u2 - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    const bool accept = !(r2 >= 1.0 || r2 == 0.0);
    const double f = sqrt(-2.0 * log(accept ? r2 : 0.5) / (accept ? r2 : 0.5));
    const double g1 = f * x2, g2 = f * x1;
    const unsigned amask = (unsigned)((__ballot(accept) >> gshift16) & 0xffffull);
    const int have = __popc(amask);
    if (have < need_total) return false;   // group-uniform
    const bool jit = (fmask >> t16) & 1u;
    const int k = __popc(fmask & ((1u << t16) - 1u));   // which accepted attempt is this target's
    const int sel = kth_set_bit16(amask, k < 16 ? k : 0);
    const double s1 = __shfl(g1, sel & 15, G), s2 = __shfl(g2, sel & 15, G);
    if (jit) {
        mx += jx2 * (s1 - 0.5);  // dx*2*(randn-0.5)
        my += jy2 * (s2 - 0.5);
    }
    words = 4 * (kth_set_bit16(amask, need_total - 1) + 1);   // up to and including the attempt that supplied the last needed pair
    return true;
}

// TRIG = false: the caller steps the env right away (fused auto-reset), so the headings' sin / cos -- recomputed by the
// kinematics of that step -- are not evaluated here.
template <int N, bool TRIG = true>
__device__ __forceinline__ void env_reset(const DevParams &p, const double *T, int b, int t, int gshift, int init,
                                          Env<N> &e) {
    if (p.variant == 1 && init) {  // flight_env.py:84-86
        float4 *m4 = reinterpret_cast<float4 *>(p.prob + (size_t)b * p.cells);
        for (int c = t; c < p.cells / 4; c += G) m4[c] = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
    }
    unsigned *mt = p.mt + (size_t)b * MT_STRIDE;
    double mx, my;
    reset_targets(p, mt, t, gshift, e.mt_pos, e.words, e.ahead, mx, my);
    e.tx = mx;
    e.ty = my;
    norm_target(p, e);
    e.found = 0;
    e.newly = 0;
    e.target_find = 0;
    e.time_step = 0;
    e.total_reward = 0;
    e.flags = 0;
    e.episodes += 1;
    bool any_in_range = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        start_pose<N>(p, i, e.ax[i], e.ay[i], e.yaw[i]);
        if (TRIG) trig_heading(T, e.yaw[i], e.sn[i], e.cs[i]);
        const double ddx = e.tx - e.ax[i], ddy = e.ty - e.ay[i];
        any_in_range = any_in_range | (t < p.n_targets && ddx * ddx + ddy * ddy <= p.view_r2);
    }
    // the reset-time pass (quirk Q3) draws nothing unless a target is within view of a start position (never for
    // agent_mode 0 with the shipped target file): request the MT window only then
    if ((__ballot(any_in_range) >> gshift) & 0xffffull) {
        detect_pass<N>(p, b, t, gshift, e, mt_prefetch(mt, e.mt_pos, t));
    } else {
        detect_finish<N>(p, t, gshift, e, false);   // what the pass does when no pair is in range: no draw, reward -1
    }
}

// env_reset(init = 0) for the 16-lane kernels' fused auto-resets, the usual case on a lean path: when the first attempt batch lies
// within the twisted words and its 16 attempts suffice (~99 %), the placement is ONE reset_batch_twisted on parameters read
// through the constant address space (scalar loads; the lane's table entries requested first, used last), the start poses come
// from the host's table, and the reset-time pass (quirk Q3) is two assignments unless a target landed within view of a start
// pose.  Everything else falls back to env_reset from the untouched state.  Same results, value for value.
// `rtab`: the target tables in LDS (load_reset_tab) or nullptr (then this lane's four entries are loaded from the kernarg segment).
template <int N, bool TRIG>
__device__ __forceinline__ void env_reset_fast(const DevParams &cp, const double *T, const double *rtab, int b, int t, int gshift,
                                               Env<N> &e) {
    bool lean = e.ahead >= 4 * G;   // group-uniform
    if (lean) {
        const CS_AS4 DevParams *q4 = cold_params4();
        double tx0, ty0, jx2, jy2;
        if (rtab) {
            tx0 = rtab[t];
            ty0 = rtab[G + t];
            jx2 = rtab[2 * G + t];
            jy2 = rtab[3 * G + t];
        } else {
            tx0 = q4->tx0[t];
            ty0 = q4->ty0[t];
            jx2 = q4->jx2[t];
            jy2 = q4->jy2[t];
        }
        const int nt = q4->n_targets, tm = q4->target_mode;
        const unsigned tmask = nt >= 32 ? ~0u : ((1u << nt) - 1u);
        const unsigned fm = tm == 0 ? ~q4->deter_mask & tmask : 0u;
        const CS_AS1 unsigned *mtb = (const CS_AS1 unsigned *)q4->mt;
        const CS_AS1 unsigned *wrow = mtb + (size_t)b * MT_STRIDE + wrap624(e.mt_pos + 4 * t);
        unsigned w4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w4[k] = wrow[k];   // (words 0..31 are mirrored behind the row)
        double mx, my;
        int words;
        lean = reset_batch_twisted(w4, tx0, ty0, jx2, jy2, fm, nt, tm, q4->L, t, gshift, mx, my, words);
        if (lean) {
            const StartTab<N> st = start_tab<N>();
            const double vr2 = q4->view_r2;
            bool near = false;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const double ddx = mx - st.x[i], ddy = my - st.y[i];
                near = near | ((t < nt) & (ddx * ddx + ddy * ddy <= vr2));
                e.ax[i] = st.x[i];
                e.ay[i] = st.y[i];
                e.yaw[i] = st.yaw;
            }
            if (TRIG) {   // every agent starts with the same heading: one evaluation
                double s0, c0;
                trig_heading(T, st.yaw, s0, c0);
#pragma unroll
                for (int i = 0; i < N; i++) {
                    e.sn[i] = s0;
                    e.cs[i] = c0;
                }
            }
            e.tx = mx;
            e.ty = my;
            e.ntx = (float)((mx - q4->mid) * q4->inv_half);   // norm_target
            e.nty = (float)((my - q4->mid) * q4->inv_half);
            e.mt_pos = wrap624(e.mt_pos + words);
            e.words += (unsigned long long)words;
            e.ahead -= words;
            e.episodes += 1;
            e.found = 0u;
            e.newly = 0u;
            e.target_find = 0;
            e.time_step = 0;
            e.total_reward = 0;
            e.curr_reward = -1;      // the reset-time pass with no pair in range: no draw, reward -1
            e.flags = FLAG_DIRTY;
            if ((__ballot(near) >> gshift) & 0xffffull) {   // group-uniform: the pass draws
                e.flags = 0;
                detect_pass<N>(cp, b, t, gshift, e, mt_prefetch(cp.mt + (size_t)b * MT_STRIDE, e.mt_pos, t));
            }
            return;
        }
    }
    env_reset<N, TRIG>(cp, T, b, t, gshift, 0, e);
}

// ---------------------------------------------------------------------------------------------------------
// Emission: get_obs (flight_env_easy.py:218-221; flight_env.py:223-230 writes the 4 features after the map)
// and get_state (flight_env_easy.py:190-216).
// ---------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void emit(const DevParams &p, int t, const Env<N> &e, float *obs_row, float *state_row) {
    const int obs_w = p.obs_row_w, feat_off = p.obs_feat_off;
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (t == i) {
            float4 f = make_float4((float)((e.ax[i] - p.mid) * p.inv_half), (float)((e.ay[i] - p.mid) * p.inv_half),
                                   (float)e.cs[i], (float)e.sn[i]);
            if (obs_row) *reinterpret_cast<float4 *>(obs_row + (size_t)i * obs_w + feat_off) = f;
            if (state_row) {
                state_row[4 * i + 0] = f.x;
                state_row[4 * i + 1] = f.y;
                state_row[4 * i + 2] = f.z;
                state_row[4 * i + 3] = f.w;
            }
        }
    }
    if (state_row && t < p.n_targets) {
        float *s = state_row + 4 * N + 3 * t;
        s[0] = e.ntx;
        s[1] = e.nty;
        s[2] = ((e.found >> t) & 1u) ? 1.0f : 0.0f;
    }
}

#ifdef CS_TIMELINE
// debug build only: per-stage s_memtime stamps of wavefront 0 / block 0 (read by tools/exp_*timeline.py of rounds 1-4: git history)
__device__ unsigned long long g_stamps[64][16];
#define CS_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_tl_step >= 0 && g_tl_step < 64) g_stamps[g_tl_step][k] = __builtin_readcyclecounter(); } while (0)
__device__ int g_tl_step_dummy;
#define LANE_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && s < 64) g_stamps[s][k] = __builtin_readcyclecounter(); } while (0)
// the constant 100 MHz counter beside the shader-clock stamps: (delta s_memtime) / (delta s_memrealtime) x 100 MHz = the clock the
// kernel actually ran at (the chip clocks to its power budget: fp64-dense kernels run well below 2.4 GHz)
#define REAL_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && s < 64) g_stamps[s][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define DUO_STAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && s < 64) g_stamps[s][k] = __builtin_readcyclecounter(); } while (0)
#define DUO_MARK(row, k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_stamps[row][k] = __builtin_readcyclecounter(); } while (0)
#define OCT_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && s < 64) g_stamps[s][k] = __builtin_readcyclecounter(); } while (0)
#define KIN_STAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && tl_step >= 0 && tl_step < 64) g_stamps[tl_step][k] = __builtin_readcyclecounter(); } while (0)
#define KIN_STAMP_SP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && sp >= 0 && sp < 64) g_stamps[sp][k] = __builtin_readcyclecounter(); } while (0)
#else
#define KIN_STAMP(k) do {} while (0)
#define KIN_STAMP_SP(k) do {} while (0)
#define OCT_STAMP(k) do {} while (0)
#define DUO_STAMP(k) do {} while (0)
#define DUO_MARK(row, k) do {} while (0)
#define CS_STAMP(k) do {} while (0)
#define LANE_STAMP(k) do {} while (0)
#define REAL_STAMP(k) do {} while (0)
#endif

struct StepIO {
    const void *actions;  // [T][B][N] int32 / int64
    float *reward;        // [T][B]
    uint8_t *terminated, *win;
    float *obs, *state;   // [T][B][...]
    int flags, T;
    int env0, env_n;      // lane kernel: this launch covers envs [env0, env0 + env_n)
    int min_ahead;        // 16-lane rollout kernels: rows with fewer twisted words than this are topped up in the prologue
    int job_parity;       // flight: which of the env's two MapJob records this step writes
};

// int32 actions, or the low dword of little-endian int64 actions (values 0..2): one branch-free strided read
template <int N>
__device__ __forceinline__ void load_actions(const StepIO &io, size_t row, int (&act)[N]) {
    const int stride = (io.flags & CS_ACTIONS_I64) ? 2 : 1;
    const int *a = reinterpret_cast<const int *>(io.actions) + row * N * stride;
#pragma unroll
    for (int i = 0; i < N; i++) act[i] = a[i * stride];
}

// Loop-invariant part of the wave-level write-out: which tile element / output dword each lane moves.
template <int N>
struct EmitPlan {
    static constexpr int K = (4 * (4 * N + 3 * CS_MAX_TARGETS) + 63) / 64;
    int st_lds[K];   // float index into tile.row (flattened [4][TILE_W])
    int st_out[K];   // dword index relative to state_out + slot0 * W
    int obs_lds;     // float index of this lane's float4 in tile.row
    int obs_out;     // float index relative to obs + slot0 * N * obs_w
    int rtw;         // env (0..3) whose reward / terminated / win this lane writes
};

template <int N>
__device__ __forceinline__ EmitPlan<N> make_emit_plan(const DevParams &p, int lane, int nvalid) {
    EmitPlan<N> pl;
    const int W = 4 * N + 3 * p.n_targets;
    const int total = nvalid * W;
    const float inv_w = 1.0f / (float)W;
#pragma unroll
    for (int k = 0; k < EmitPlan<N>::K; k++) {
        int idx = lane + 64 * k;
        idx = idx < total ? idx : total - 1;  // surplus lanes repeat the last element (same value, same address)
        const int r = (int)(((float)idx + 0.5f) * inv_w);
        pl.st_lds[k] = r * TILE_W + (idx - r * W);
        pl.st_out[k] = idx;
    }
    const int l = lane < nvalid * N ? lane : nvalid * N - 1;
    const int r = l / N, i = l - r * N;
    pl.obs_lds = r * TILE_W + 4 * i;
    pl.obs_out = (r * N + i) * p.obs_row_w + p.obs_feat_off;
    pl.rtw = (lane & 3) < nvalid ? (lane & 3) : nvalid - 1;
    return pl;
}

// Wave-level write-out of one step (see WaveTile), in two halves so that the rollout loop can overlap the LDS round
// trip and the stores of step s with the arithmetic of step s+1:
//   emit_deposit: every live lane writes its pieces of step s into the tile (end of step s);
//   emit_flush:   all 64 lanes read the tile back and store it (called at the start of step s+1, or right away by
//                 the single-step kernel).  slot0 = output slot of the wavefront's first env for the deposited step.
template <int N, bool SELECT = true>
__device__ __forceinline__ void emit_deposit(const DevParams &p, WaveTile &tile, int t, int grp, bool live, const Env<N> &e,
                                             int reward, bool term) {
    if (live) {
        if (SELECT) {
            // lane i < N deposits agent i, picked with selects: in the two-role kernel the compiler turns the branchy
            // form below into an indexed read (agent arrays in scratch), and k_rollout<5> needs 5 VGPRs more with it
            // (258: one wavefront per SIMD instead of two)
            double mx = 0.0, my = 0.0, mc = 0.0, ms = 0.0;
#pragma unroll
            for (int i = 0; i < N; i++) {
                mx = t == i ? e.ax[i] : mx;
                my = t == i ? e.ay[i] : my;
                mc = t == i ? e.cs[i] : mc;
                ms = t == i ? e.sn[i] : ms;
            }
            if (t < N)
                *reinterpret_cast<float4 *>(&tile.row[grp][4 * t]) =
                    make_float4((float)((mx - p.mid) * p.inv_half), (float)((my - p.mid) * p.inv_half), (float)mc, (float)ms);
        } else {
#pragma unroll
            for (int i = 0; i < N; i++)
                if (t == i)
                    *reinterpret_cast<float4 *>(&tile.row[grp][4 * i]) =
                        make_float4((float)((e.ax[i] - p.mid) * p.inv_half), (float)((e.ay[i] - p.mid) * p.inv_half),
                                    (float)e.cs[i], (float)e.sn[i]);
        }
        if (t < p.n_targets) {
            tile.row[grp][4 * N + 3 * t + 0] = e.ntx;
            tile.row[grp][4 * N + 3 * t + 1] = e.nty;
            tile.row[grp][4 * N + 3 * t + 2] = ((e.found >> t) & 1u) ? 1.0f : 0.0f;
        }
        if (t == 0) {
            tile.reward[grp] = (float)reward;
            tile.term[grp] = term ? 1 : 0;
            tile.win[grp] = (e.flags & FLAG_WIN) ? 1 : 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int N>
struct FlushRegs {
    float st[EmitPlan<N>::K];
    float4 obs;
    float reward;
    int term, win;
};

template <int N>
__device__ __forceinline__ void emit_flush_load(const WaveTile &tile, const EmitPlan<N> &pl, FlushRegs<N> &f) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float *flat = &tile.row[0][0];
    f.reward = tile.reward[pl.rtw];
    f.term = tile.term[pl.rtw];
    f.win = tile.win[pl.rtw];
    f.obs = *reinterpret_cast<const float4 *>(flat + pl.obs_lds);
#pragma unroll
    for (int k = 0; k < EmitPlan<N>::K; k++) f.st[k] = flat[pl.st_lds[k]];
}

template <int N>
__device__ __forceinline__ void emit_flush_store(const DevParams &p, const StepIO &io, const EmitPlan<N> &pl,
                                                 const FlushRegs<N> &f, size_t slot0) {
    io.reward[slot0 + pl.rtw] = f.reward;  // duplicates write the same value
    io.terminated[slot0 + pl.rtw] = (uint8_t)f.term;
    io.win[slot0 + pl.rtw] = (uint8_t)f.win;
#ifndef CS_EMIT_NT
#define CS_EMIT_NT 1
#endif
    if (io.obs) {  // one float4 per (env, agent)
        float4 *dst = reinterpret_cast<float4 *>(io.obs + slot0 * N * (size_t)p.obs_row_w + pl.obs_out);
#if CS_EMIT_NT
        const v4f nv = {f.obs.x, f.obs.y, f.obs.z, f.obs.w};
        __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst));
#else
        *dst = f.obs;
#endif
    }
    if (io.state) {  // the wavefront's rows are contiguous in get_state's [B][W] layout
        float *dst = io.state + slot0 * (size_t)(4 * N + 3 * p.n_targets);
#pragma unroll
        for (int k = 0; k < EmitPlan<N>::K; k++) {
#if CS_EMIT_NT
            __builtin_nontemporal_store(f.st[k], dst + pl.st_out[k]);
#else
            dst[pl.st_out[k]] = f.st[k];
#endif
        }
    }
}

// One env.step for the group's env, state in registers.  `act` are this step's actions, `win` the MT window at
// e.mt_pos (both already loaded); called by all 64 lanes of the wavefront (`live` = the lane's env exists).
template <int N, int VARIANT, bool FASTRESET = (VARIANT == 0)>
__device__ __forceinline__ void step_once(const DevParams &p, const double *T, const StepIO &io, WaveTile &tile, int b,
                                          int lane, size_t slot0, const EmitPlan<N> &plan, bool live, const int (&act)[N],
                                          MtWin &win, bool prefetch_next, bool flush_prev, size_t prev_slot0,
                                          bool defer_flush, Env<N> &e, unsigned (&tape)[TAPE_DW], const bool use_tape,
                                          bool tape_ok) {
    const int t = lane & (G - 1), grp = lane >> 4, gshift = lane & ~(G - 1);
    int reward = 0;
    bool term = true;
    // the previous step's rows: LDS reads now, global stores after this step's arithmetic
    FlushRegs<N> fr;
    if (flush_prev) emit_flush_load<N>(tile, plan, fr);
#ifdef CS_TIMELINE
    const int g_tl_step = (int)(slot0 / (size_t)p.B);
#endif
    CS_STAMP(0);
    if (live) {
        bool done = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
        e.flags &= ~(FLAG_DIRTY | FLAG_RESET_PASS);  // pending-map-update flags describe THIS launch only
        if (done && (io.flags & CS_AUTO_RESET)) {
            const unsigned long long words_before = e.words;
            // cold path: parameters read where they are needed.  (FASTRESET off: the step role of k_flight_pipe is held to 128
            // VGPRs; with the lean path compiled in it spills 12 of them and the pipelined sweep loses 3 %: env_reset as it was)
            if (FASTRESET) env_reset_fast<N, false>(cold_params(), T, nullptr, b, t, gshift, e);
            else env_reset<N, false>(cold_params(), T, b, t, gshift, 0, e);
            if (VARIANT == 1) {  // flight: the map kernel must replay the reset-time update before this step's
                e.newly_reset = e.newly;
                e.flags |= FLAG_RESET_PASS;
            }
            env_store<N>(p, b, t, e, true);  // targets changed
            if (use_tape) {   // the draw slots the reset consumed leave the tape
                const unsigned long long used = e.words - words_before;
                tape_shift<8>(tape, used < 2ull * 319ull ? (int)(used >> 1) : 319);
            } else {
                win = mt_prefetch(p.mt + (size_t)b * MT_STRIDE, e.mt_pos, t);
            }
            drain_vmem();
            done = false;
        }
        if (!(done && (io.flags & CS_FREEZE_DONE))) {
            CS_STAMP(1);
            kinematics_group<N, VARIANT>(p, T, tile, act, t, grp, e);
            CS_STAMP(2);
            reward = use_tape ? detect_pass_tape<N>(p, b, t, gshift, e, tape, tape_ok) : detect_pass<N>(p, b, t, gshift, e, win);
            CS_STAMP(3);
            e.total_reward += reward;
            e.time_step += 1;
            term = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
        } else {
            env_trig<N>(T, e);  // frozen env: re-emit the unchanged observation
        }
        // the next step's window does not overlap the words just committed: request it before this step's stores
        if (prefetch_next && !use_tape) win = mt_prefetch(p.mt + (size_t)b * MT_STRIDE, e.mt_pos, t);
    }
    CS_STAMP(4);
    if (flush_prev) emit_flush_store<N>(p, io, plan, fr, prev_slot0);
    emit_deposit<N>(p, tile, t, grp, live, e, reward, term);
    if (!defer_flush) {
        emit_flush_load<N>(tile, plan, fr);
        emit_flush_store<N>(p, io, plan, fr, slot0);
    }
    CS_STAMP(5);
}

// One launch's share of a single step: workgroup `blk` of BLOCK threads = 16 envs.
template <int N, int VARIANT, bool TAPE = true, bool FASTRESET = true>
__device__ __forceinline__ void step_block(const DevParams &p, const StepIO &io, double *T, WaveTile *tiles, int blk) {
    const int gid = blk * BLOCK + threadIdx.x;
    const int b = gid / G, t = gid % G;
    const int lane = threadIdx.x & 63;
    const bool live = b < p.B;
    // issue every independent global load before the barrier that publishes the trig table
#ifndef CS_STEP_TAPE
#define CS_STEP_TAPE 1
#endif
    // The draws of a single step come from the env's hit tape while it is valid (left by a rollout call or cs_mt_advance;
    // it is never written here: tape_finish rebases it by the words consumed since), which takes the MT19937 window --
    // a load that depends on the header's cursor -- off the launch's critical path; otherwise words are twisted on demand.
    // (the rollout call of flight passes TAPE = false: nothing refreshes the tapes between its launches, and the step
    // role there is hidden behind the map sweep either way)
    constexpr bool STEP_TAPE = CS_STEP_TAPE && TAPE && N <= 5;
    Env<N> e;
    int act[N];
    TapeRaw traw = {};
    if (live) {
        env_load<N>(p, b, t, e);
        load_actions<N>(io, (size_t)b, act);
        if (STEP_TAPE) traw = tape_fetch(p, b);
    }
    load_trig_to_lds(T);
    const int wave_b0 = (blk * BLOCK + (threadIdx.x & ~63)) / G;
    if (wave_b0 >= p.B) return;
    const int nvalid = p.B - wave_b0 < 4 ? p.B - wave_b0 : 4;
    MtWin win = {0u, 0u};
    unsigned tape[TAPE_DW];
    bool tape_ok = false;
    if (STEP_TAPE) {
        if (live) tape_ok = tape_finish(p, traw, e, tape);
    } else if (live) {
        win = mt_prefetch(p.mt + (size_t)b * MT_STRIDE, e.mt_pos, t);
    }
    const EmitPlan<N> plan = make_emit_plan<N>(p, lane, nvalid);
    step_once<N, VARIANT, FASTRESET>(p, T, io, tiles[threadIdx.x >> 6], b, lane, (size_t)wave_b0, plan, live, act, win, false, false, 0, false, e,
                          tape, STEP_TAPE, tape_ok);
    if (live) {
        env_store<N>(p, b, t, e, false);
        if (VARIANT == 1) job_store<N>(p, io.job_parity, b, t, e);
    }
}

template <int N, int VARIANT>
__global__ __launch_bounds__(BLOCK) void k_step(DevParams p, StepIO io) {
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    __shared__ WaveTile tiles[BLOCK / 64];
    step_block<N, VARIANT>(p, io, T, tiles, blockIdx.x);
}

// T steps per launch, env resident in registers between steps (flight_easy).  The next step's actions and MT
// window are requested before the current step's stores so their latency hides behind the arithmetic.
template <int N>
__global__ __launch_bounds__(BLOCK) void k_rollout(DevParams p, StepIO io) {
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    __shared__ WaveTile tiles[BLOCK / 64];
    __shared__ unsigned rowbufs[N <= 4 ? BLOCK / 64 : 1][N <= 4 ? MT_N : 1];   // one MT19937 row per wavefront (prologue)
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    const int b = gid / G, t = gid % G;
    const int lane = threadIdx.x & 63;
    const bool live = b < p.B;
    // the env's hit tape (cs_mt_advance), replicated in the group's lanes; teams of 5 and more keep the on-demand
    // window instead: the ten tape registers would cost them their second wavefront per SIMD
    constexpr bool USE_TAPE = N <= 4;
    Env<N> e;
    int act[N];
    TapeRaw traw = {};
    if (live) {   // everything the first step waits for is requested before the barrier that publishes the trig table
        env_load<N>(p, b, t, e);
        load_actions<N>(io, (size_t)b, act);
        if (USE_TAPE) traw = tape_fetch(p, b);
    }
    load_trig_to_lds(T);
    const int wave_b0 = (blockIdx.x * BLOCK + (threadIdx.x & ~63)) / G;
    if (wave_b0 >= p.B) return;
    const int nvalid = p.B - wave_b0 < 4 ? p.B - wave_b0 : 4;
    WaveTile &tile = tiles[threadIdx.x >> 6];
    const EmitPlan<N> plan = make_emit_plan<N>(p, lane, nvalid);
    constexpr bool PIPE = N <= 4;
    MtWin win = {0u, 0u};
    unsigned tape[TAPE_DW];
    bool tape_ok = false;
    if (USE_TAPE) {
        if (live) tape_ok = tape_finish(p, traw, e, tape);
        group_wave_advance<N>(p, wave_b0, nvalid, lane, io.min_ahead, rowbufs[N <= 4 ? threadIdx.x >> 6 : 0], e, tape, tape_ok);
    }
    if (!USE_TAPE && live) win = mt_prefetch(p.mt + (size_t)b * MT_STRIDE, e.mt_pos, t);
    for (int s = 0; s < io.T; s++) {
        // (no in-loop top-up here: it would push this kernel past 256 VGPRs and cost its second wavefront per SIMD; an env
        // that exhausts its row falls back to twisting on demand until the next launch's prologue)
        int act_next[N];
        const int sn = s + 1 < io.T ? s + 1 : s;
        load_actions<N>(io, (size_t)sn * p.B + (live ? b : 0), act_next);
        // (lane predicates -- t == i, t < n_targets ... -- are recomputed every step: hoisted out of the loop each of them is
        // an SGPR pair the scalar file has no room for, and they came back as v_readlane pairs at every use)
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        // n <= 4: the rows of step s are stored while step s+1 computes (costs ~12 VGPRs; larger teams have none spare)
        step_once<N, 0>(p, T, io, tile, b, lane_s, (size_t)s * p.B + wave_b0, plan, live, act, win, s + 1 < io.T,
                        PIPE && s > 0, (size_t)(s - 1) * p.B + wave_b0, PIPE, e, tape, USE_TAPE, tape_ok);
#pragma unroll
        for (int i = 0; i < N; i++) act[i] = act_next[i];
    }
    if (PIPE) {  // rows of the last step
        FlushRegs<N> fr;
        emit_flush_load<N>(tile, plan, fr);
        emit_flush_store<N>(p, io, plan, fr, (size_t)(io.T - 1) * p.B + wave_b0);
    }
    if (live) {
        env_store<N>(p, b, t, e, false);
        if (USE_TAPE && tape_ok) group_tape_store<N>(p, b, t, e, tape);
    }
}

// =========================================================================================================
// Two-role rollout (flight_easy): the default T-step kernel of the 16-lanes-per-env path.
//
// One env.step is a dependent chain -- kinematics (trig lookup, move, wall test), then the detection pass over the
// new positions, reward, emission -- of ~6000 cycles on one wavefront, and at the batch sizes this path serves
// (B = 4096: one wavefront per SIMD) nothing else is there to fill its stalls.  But the kinematics of step s + 1
// need nothing from the detection pass of step s: the actions are an open-loop table, and the only coupling is
// termination (auto-reset / freeze), which is predictable from the step counter except when an env finds its last
// target.  So every group of four envs gets TWO wavefronts: wave K runs the kinematics of step s + 1 while its
// partner wave D runs detection + reward + emission of step s on the positions K left in a two-slot LDS ring; one
// workgroup barrier per step.  When D sees a termination K could not predict (a win before the time limit) it
// flags the group, and after the barrier K restores that env from the ring, applies the reset / freeze and redoes
// the step (one extra barrier, a few times per episode batch).  Arithmetic per env is exactly k_rollout's (same
// functions), so results are bit-identical; the step time drops from kinematics + detection + emission to
// max(kinematics, detection + emission), and B = 4096 fills both wave slots of every SIMD.
// =========================================================================================================
template <int N>
struct KinSlot {   // agents of one env after a step: K -> D
    double x[N], y[N], yaw[N];
    float cs[N], sn[N];
    unsigned out;   // out_flag bits (OUT_PUNISH)
    int pad;
};

// Pairs per workgroup share the per-step barrier, so a pair waits for the slowest of its neighbours every step.  Measured
// (flight_easy 3a15t, B = 4096): 4 / 2 / 1 pairs -> 1.78 / 1.77 / 2.02e9 env-steps/s; 5 agents: 1.17 -> 1.32e9.
#ifndef CS_DUO_PAIRS
#define CS_DUO_PAIRS 1
#endif
constexpr int DUO_PAIRS = CS_DUO_PAIRS;        // wavefront pairs per workgroup
constexpr int DUO_ENVS = 4 * DUO_PAIRS;        // envs per workgroup
constexpr int DUO_BLOCK = 128 * DUO_PAIRS;     // DUO_PAIRS K wavefronts, then DUO_PAIRS D wavefronts

template <int N>
__global__ __launch_bounds__(DUO_BLOCK, 2) void k_rollout_duo(DevParams p, StepIO io) {
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    __shared__ WaveTile tiles[DUO_BLOCK / 64];   // K waves use .trig, D waves the emission rows
    __shared__ KinSlot<N> slots[2][DUO_ENVS];
    __shared__ unsigned fix[2][DUO_PAIRS];               // [step parity][pair]: groups whose termination K mispredicted
    __shared__ unsigned rowbufs[DUO_PAIRS][MT_N];        // one MT19937 row per D wavefront (prologue top-up)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_k = wave < DUO_PAIRS;
    DUO_MARK(63, is_k ? 3 : 13);   // entry
    const int pw = wave % DUO_PAIRS;             // wave pair = 4 envs
    int t = lane & (G - 1);   // (made opaque once per step: its predicates are recomputed instead of living in SGPR pairs)
    const int grp = lane >> 4, gshift = lane & ~(G - 1);
    const int el = 4 * pw + grp;                 // env within the block
    const int b = blockIdx.x * DUO_ENVS + el;
    const bool live = b < p.B;
    const int wave_b0 = blockIdx.x * DUO_ENVS + 4 * pw;
    const int nvalid = p.B - wave_b0 < 4 ? p.B - wave_b0 : 4;   // <= 0: a pair without envs (it still meets the barriers)
    const bool auto_reset = io.flags & CS_AUTO_RESET, freeze = io.flags & CS_FREEZE_DONE;
    Env<N> e;
    if (live) env_load<N>(p, b, t, e);
    // everything either role will wait for first is requested before the barrier that publishes the trig table
    const size_t arow = live ? (size_t)b : 0;
    int act[N], act_next[N];
    TapeRaw traw = {};
    if (is_k) {
        load_actions<N>(io, arow, act);
        load_actions<N>(io, (size_t)(1 < io.T ? 1 : 0) * p.B + arow, act_next);   // one step ahead of its use
    } else if (live) {
        traw = tape_fetch(p, b);
    }
    load_trig_to_lds(T);
    DUO_MARK(63, is_k ? 4 : 14);   // state requested, trig table in LDS
    WaveTile &tile = tiles[wave];

    if (is_k) {
        // ------------------------------------------------------------------------------------------ K: kinematics
        // the longer half of the pair gets the issue priority: K from four agents up (5 agents, B = 4096: 1.35 -> 1.43e9
        // env-steps/s; raising D instead: 1.36e9), D for smaller teams (below)
#ifndef CS_DUO_K_PRIO_FROM
#define CS_DUO_K_PRIO_FROM 4
#endif
        if (N >= CS_DUO_K_PRIO_FROM) __builtin_amdgcn_s_setprio(1);
        bool k_done = live && (e.target_find >= p.n_targets || e.time_step >= p.time_limit);   // exact at launch
        int k_time = e.time_step;
        if (live && freeze) env_trig<N>(T, e);   // what a frozen env keeps emitting (every other path recomputes cs / sn)
        // produces the state after step `sp` from the state after step sp - 1 and writes it to the ring
        auto produce = [&](int sp, const int (&a)[N]) __attribute__((always_inline)) {
            if (live) {
                bool frozen = false;
                if (k_done && auto_reset) {
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        start_pose<N>(p, i, e.ax[i], e.ay[i], e.yaw[i]);
                        trig_heading(T, e.yaw[i], e.sn[i], e.cs[i]);
                    }
                    e.flags &= ~0xff00;
                    k_time = 0;
                    k_done = false;
                } else if (k_done && freeze) {
                    frozen = true;
                }
                if (!frozen) {
                    kinematics_group<N, 0>(p, T, tile, a, t, grp, e);
                    k_time += 1;
                    k_done = k_time >= p.time_limit;   // a win is D's knowledge: see the fix-up below
                }
                KinSlot<N> &sl = slots[sp & 1][el];
                double mx = 0.0, my = 0.0, mw = 0.0, mc = 0.0, ms = 0.0;   // lane i < N publishes agent i
#pragma unroll
                for (int i = 0; i < N; i++) {
                    mx = t == i ? e.ax[i] : mx;
                    my = t == i ? e.ay[i] : my;
                    mw = t == i ? e.yaw[i] : mw;
                    mc = t == i ? e.cs[i] : mc;
                    ms = t == i ? e.sn[i] : ms;
                }
                if (t < N) {
                    sl.x[t] = mx;
                    sl.y[t] = my;
                    sl.yaw[t] = mw;
                    sl.cs[t] = (float)mc;
                    sl.sn[t] = (float)ms;
                }
                if (t == 0) sl.out = ((unsigned)e.flags >> 8) & 0xffu;
            }
        };
        produce(0, act);
        __syncthreads();
        for (int s = 0; s < io.T; s++) {
            asm volatile("" : "+v"(t));
            const bool more = s + 1 < io.T;
            int act_after[N];
            DUO_STAMP(0);
            load_actions<N>(io, (size_t)(s + 2 < io.T ? s + 2 : io.T - 1) * p.B + arow, act_after);
            if (more) produce(s + 1, act_next);
            DUO_STAMP(1);
            __syncthreads();
            DUO_STAMP(2);
            unsigned any_fix = 0, mine = 0;
#pragma unroll
            for (int q = 0; q < DUO_PAIRS; q++) {
                const unsigned f = fix[s & 1][q];
                any_fix |= f;
                mine = pw == q ? f : mine;
            }
            if (any_fix) {   // block-uniform, rare: an env of the block terminated by finding its last target
                if (more && live && ((mine >> grp) & 1u)) {
                    const KinSlot<N> &sl = slots[s & 1][el];   // the env as it was after step s
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        e.ax[i] = sl.x[i];
                        e.ay[i] = sl.y[i];
                        e.yaw[i] = sl.yaw[i];
                    }
                    env_trig<N>(T, e);
                    e.flags = (e.flags & ~0xff00) | (int)(sl.out << 8);
                    k_done = true;
                    k_time -= 1;           // the speculative step s + 1 is undone (a frozen env never gets here)
                    produce(s + 1, act_next);
                }
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < N; i++) act_next[i] = act_after[i];
        }
        if (live) {   // agents are K's part of the state
            double4 *a4 = reinterpret_cast<double4 *>(p.agent + (size_t)b * CS_MAX_AGENTS * 4);
#pragma unroll
            for (int i = 0; i < N; i++)
                if (t == i) a4[i] = make_double4(e.ax[i], e.ay[i], e.yaw[i], 0.0);
        }
        DUO_MARK(63, 5);
        return;
    }

    // ---------------------------------------------------------------------------------------------- D: detection
    // D is the longer half of the pair and the YOUNGER wavefront of its SIMD (K waves are dispatched first): at equal
    // priority the issue arbiter serves the older wave first and D gets the leftover slots (timeline: its detection +
    // emission take 1.8x what they take alone).  Raised priority gives the slots to the longer half (measured, B = 4096:
    // prio 0 / 1 / 2 / 3 -> 1.73 / 1.80 / 1.75 / 1.78e9 env-steps/s; at 2 K becomes the slower half: produce 2600 -> 3300).
#ifndef CS_DUO_D_PRIO
#define CS_DUO_D_PRIO 1
#endif
    if (N <= 3) __builtin_amdgcn_s_setprio(CS_DUO_D_PRIO);   // larger teams: K (n agents' kinematics) is the longer half

    const bool wave_valid = nvalid > 0;
    const EmitPlan<N> plan = make_emit_plan<N>(p, lane, wave_valid ? nvalid : 1);
    constexpr bool PIPE = N <= 4;
    unsigned tape[TAPE_DW];   // the env's hit tape, replicated in the group's lanes
    bool tape_ok = false;
    if (live) tape_ok = tape_finish(p, traw, e, tape);
    group_wave_advance<N>(p, wave_b0, nvalid, lane, io.min_ahead, rowbufs[pw], e, tape, tape_ok);   // while K produces step 0
    if (threadIdx.x == DUO_PAIRS * 64) {
#pragma unroll
        for (int q = 0; q < 2 * DUO_PAIRS; q++) (&fix[0][0])[q] = 0u;
    }
    __syncthreads();   // the ring holds step 0
    for (int s = 0; s < io.T; s++) {
        asm volatile("" : "+v"(t));
        const size_t slot0 = (size_t)s * p.B + wave_b0;
        int reward = 0;
        bool term = true, mispredicted = false;
        FlushRegs<N> fr;
        DUO_STAMP(8);
        // a row that is about to run out of twisted words is topped up in place (about once per env and 80 steps)
        if (__ballot(live && tape_ok && e.ahead < 2 * N * CS_MAX_TARGETS))
            group_wave_advance<N>(p, wave_b0, nvalid, lane, 2 * N * CS_MAX_TARGETS, rowbufs[pw], e, tape, tape_ok);
        if (PIPE && s > 0 && wave_valid) emit_flush_load<N>(tile, plan, fr);
        if (live) {
            bool done = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
            e.flags &= ~(FLAG_DIRTY | FLAG_RESET_PASS);
            if (done && auto_reset) {
                const unsigned long long words_before = e.words;
                const DevParams &cp = cold_params();
                env_reset_fast<N, false>(cp, T, nullptr, b, t, gshift, e);
                reinterpret_cast<double2 *>(cp.tgt + (size_t)b * G * 2)[t] = make_double2(e.tx, e.ty);
                const unsigned long long used = e.words - words_before;   // its draw slots leave the tape
                tape_shift<8>(tape, used < 2ull * 319ull ? (int)(used >> 1) : 319);
                drain_vmem();
                done = false;
            }
            const KinSlot<N> &sl = slots[s & 1][el];
#pragma unroll
            for (int i = 0; i < N; i++) {
                e.ax[i] = sl.x[i];
                e.ay[i] = sl.y[i];
                e.yaw[i] = sl.yaw[i];
                e.cs[i] = (double)sl.cs[i];
                e.sn[i] = (double)sl.sn[i];
            }
            e.flags = (e.flags & ~0xff00) | (int)(sl.out << 8);
            DUO_STAMP(9);
            if (!(done && freeze)) {
                reward = detect_pass_tape<N>(p, b, t, gshift, e, tape, tape_ok);
                DUO_STAMP(10);
                e.total_reward += reward;
                e.time_step += 1;
                term = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
                // K steps on unless the step counter says otherwise
                mispredicted = (auto_reset || freeze) && term && e.time_step < p.time_limit;
            }
        }
        {
            const unsigned long long mb = __ballot(mispredicted && t == 0);
            const unsigned m4 = (unsigned)((mb >> 0) & 1ull) | (unsigned)((mb >> 15) & 2ull) | (unsigned)((mb >> 30) & 4ull) |
                                (unsigned)((mb >> 45) & 8ull);
            if (lane == 0) fix[s & 1][pw] = m4;
        }
        if (PIPE && s > 0 && wave_valid) emit_flush_store<N>(p, io, plan, fr, (size_t)(s - 1) * p.B + wave_b0);
        emit_deposit<N, true>(p, tile, t, grp, live, e, reward, term);
        if (!PIPE && wave_valid) {
            emit_flush_load<N>(tile, plan, fr);
            emit_flush_store<N>(p, io, plan, fr, slot0);
        }
        DUO_STAMP(11);
        __syncthreads();
        DUO_STAMP(12);
        unsigned any_fix = 0;
#pragma unroll
        for (int q = 0; q < DUO_PAIRS; q++) any_fix |= fix[s & 1][q];
        if (any_fix) __syncthreads();   // K redoes step s + 1 of the flagged envs
    }
    if (PIPE && wave_valid) {
        FlushRegs<N> fr;
        emit_flush_load<N>(tile, plan, fr);
        emit_flush_store<N>(p, io, plan, fr, (size_t)(io.T - 1) * p.B + wave_b0);
    }
    if (live && t == 0) {   // header (and cursor) are D's part of the state; targets were stored at each reset
        int4 *h4 = reinterpret_cast<int4 *>(p.hdr + (size_t)b * CS_H_WORDS);
        h4[0] = make_int4((int)e.found, (int)e.newly, e.target_find, e.flags);
        h4[1] = make_int4(e.time_step, e.total_reward, e.mt_pos, e.episodes);
        h4[2] = make_int4((int)(unsigned)(e.words & 0xffffffffull), (int)(unsigned)(e.words >> 32), e.curr_reward,
                          (int)e.newly_reset);
        p.ahead[b] = e.ahead;
    }
    if (live && tape_ok) group_tape_store<N>(p, b, t, e, tape);
    DUO_MARK(63, 15);
}

// =========================================================================================================
// Fused closed-loop rollout (flight_easy): T x (agent network forward -> env.step) in ONE launch.
//
// The caller-side row f3 (csrc/policy.hip) and the env step are both latency-bound at the batch sizes a collector
// uses (B = 4096: one wavefront per SIMD), and two launches per step cost ~20 us.  Here a block keeps its 16 envs
// (4 wavefronts x 4 groups, as k_rollout) AND their 16*N network rows resident: the N row tiles of 16 rows go
// through fc1 -> GRUCell -> fc2 on the fp32 matrix cores exactly as in k_policy (wavefront w owns hidden columns
// 16w..16w+15; same fragment order, same summation order, so the actions are bit-identical to the two-kernel loop),
// the hidden state never leaves LDS between steps, the chosen actions go through LDS to the env groups, and the
// env step is step_once of k_rollout (same MT19937 order, same emission).
// =========================================================================================================
struct PolicyIO {
    const float *w;          // packed weights (cs_policy_pack)
    float *hidden;           // [B*N][64] in/out
    const int64_t *last;     // [B][N] action before the first step (< 0 = none)
    int64_t *actions;        // [T][B][N] chosen actions
    double epsilon;          // exploration schedule (cs_epsilon): start value when eps_dev is null,
    double anneal, min_eps;  //   the step-scale rule of common/rollout.py:75-76,
    int per_step;            //   applied after every executed env step if set,
    double *eps_dev;         //   per-env values [B] in / out (null: `epsilon` throughout),
    double *trace;           //   the value every env's selection used at every step [T][B] (null: none)
    unsigned long long seed;
    unsigned step0;          // epsilon-greedy counter of the first step (one per step, as one cs_policy_forward call each)
    unsigned long long row0; // global index of network row 0 (sharded batches)
    int select;              // CS_SELECT_*
};

#ifndef CS_RP_WAVES
#define CS_RP_WAVES 1   /* wavefronts per SIMD the fused closed-loop kernel is compiled for at N <= 3 (2: 256 registers) */
#endif
#ifndef CS_RP_TAPE
#define CS_RP_TAPE 1    /* teams of up to 3 read their draws from the hit tape (10 KB of row buffers per workgroup) */
#endif
template <int N>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((N <= 3 ? CS_RP_WAVES : 1), (N <= 3 ? CS_RP_WAVES : 1))))
void k_rollout_policy(DevParams p, StepIO io, PolicyIO pio) {
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    __shared__ WaveTile tiles[BLOCK / 64];
    __shared__ int s_act[16 * N];          // last / chosen action per row (row = env_in_block * N + agent)
    __shared__ float s_b3[16];
    __shared__ double s_eps[BLOCK / G];    // the block's 16 envs' epsilon (cs_epsilon: annealed env by env, rollout.py:75-76)
    extern __shared__ __attribute__((aligned(16))) float pol_lds[];
    constexpr int ROWS = 16 * N, NA = 3;   // the env has three actions (flight_env_easy.py:32)
#if CS_POLICY_F16
    // split-fp16 activations (policy_dev.h), all as (hi, lo) plane pairs of halves:
    //   x  [ROWS][HXS]   the network input of the NEXT forward, kept current in place: the env lanes write the four observation
    //                    columns after every step, the selecting lanes the one-hot of the chosen action; the agent-id columns and
    //                    the zero padding never change (no assembly phase, no barrier for it)
    //   b  [ROWS][HST]   h1, then f (scratch of one forward)
    //   hs [2][...]      the hidden state as A-operand planes, PING-PONG by step parity: the GRU of step s reads hs[s & 1] (every
    //                    wavefront reads all rows) and writes hs[(s + 1) & 1] -- no barrier between its reads and its writes; fc2 reads
    //                    h' from there too, and the q values of step s take the space of hs[s & 1] once the GRU has consumed it
    //   s_h [ROWS][LDW]  the hidden state in fp32 (the GRU blend): element (row, col) is read and written by ONE thread only
    _Float16 *x_hi = reinterpret_cast<_Float16 *>(pol_lds), *x_lo = x_hi + ROWS * HXS;
    _Float16 *b_hi = x_lo + ROWS * HXS, *b_lo = b_hi + ROWS * HST;
    _Float16 *hs_base = b_lo + ROWS * HST;                 // [2][2 planes][ROWS][HST]
    float *s_h = reinterpret_cast<float *>(hs_base + 4 * ROWS * HST);
#else
    // s_a | s_b | s_h, each [16N][LDW]; the partial q of fc2 aliases s_a
    float *s_a = pol_lds, *s_b = pol_lds + ROWS * LDW, *s_h = pol_lds + 2 * ROWS * LDW;
    float *s_q = s_a;                      // [4][ROWS * 17] <= ROWS * LDW floats
#endif
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    const int b = gid / G, t = gid % G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane >> 4;
    const bool live = b < p.B;
    const int b0 = blockIdx.x * (BLOCK / G);                 // first env of the block
    const int rows_valid = (p.B - b0 < 16 ? p.B - b0 : 16) * N;
    Env<N> e;
    if (live) env_load<N>(p, b, t, e);
    load_trig_to_lds(T);
    const int wave_b0 = b0 + 4 * w;
    const int nvalid = p.B - wave_b0 < 4 ? p.B - wave_b0 : 4;
    const bool wave_valid = nvalid > 0;                      // wave-uniform
    WaveTile &tile = tiles[w];
    const EmitPlan<N> plan = make_emit_plan<N>(p, lane, wave_valid ? nvalid : 1);
    constexpr bool PIPE = N <= 4;

    // ---- policy: weight fragments and biases of this wavefront's column tile, once (k_policy)
    const int crow = (lane >> 4) * 4, ccol = lane & 15, col = 16 * w + ccol;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#if CS_POLICY_F16
    const unsigned ulane = lane;
    const BFrag b1 = load_bfrag(pio.w, HOFF_W1, w, ulane);
    BFrag bg[6][2];
#pragma unroll
    for (int g = 0; g < 3; g++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bg[2 * g][ks] = load_bfrag(pio.w, HOFF_WIH, (w + 4 * g) * 2 + ks, ulane);
            bg[2 * g + 1][ks] = load_bfrag(pio.w, HOFF_WHH, (w + 4 * g) * 2 + ks, ulane);
        }
    BFrag b2[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) b2[ks] = load_bfrag(pio.w, HOFF_W2, w * 2 + ks, ulane);
    BFrag b3[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) b3[ks] = load_bfrag(pio.w, HOFF_W3, ks, ulane);
    constexpr int PO_B1 = HOFF_B1, PO_BIH = HOFF_BIH, PO_BHH = HOFF_BHH, PO_B2 = HOFF_B2, PO_B3 = HOFF_B3;
#else
    float b1[4], bg[6][16], b2[16], b3f[4];
    {
        const unsigned ulane = lane;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) b1[kk] = (pio.w + OFF_W1 + (w * (KIN_MAX / 4) + kk) * FR)[ulane];
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                bg[2 * g][kk] = (pio.w + OFF_WIH + ((w + 4 * g) * 16 + kk) * FR)[ulane];
                bg[2 * g + 1][kk] = (pio.w + OFF_WHH + ((w + 4 * g) * 16 + kk) * FR)[ulane];
            }
#pragma unroll
        for (int kk = 0; kk < 16; kk++) b2[kk] = (pio.w + OFF_W2 + (w * 16 + kk) * FR)[ulane];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) b3f[kk] = (pio.w + OFF_W3 + (4 * w + kk) * FR)[ulane];
    }
    constexpr int PO_B1 = OFF_B1, PO_BIH = OFF_BIH, PO_BHH = OFF_BHH, PO_B2 = OFF_B2, PO_B3 = OFF_B3;
#endif
    const float bias1 = pio.w[PO_B1 + col], bias2 = pio.w[PO_B2 + col];
    const float bir = pio.w[PO_BIH + col], biz = pio.w[PO_BIH + 64 + col], bin = pio.w[PO_BIH + 128 + col];
    const float bhr = pio.w[PO_BHH + col], bhz = pio.w[PO_BHH + 64 + col], bhn = pio.w[PO_BHH + 128 + col];
#if CS_POLICY_F16
    const float b_r = bir + bhr, b_z = biz + bhz;   // the r and z gates run as one chain over [x | h] with one bias (gru_cell)
#endif
    if (threadIdx.x < 16) s_b3[threadIdx.x] = pio.w[PO_B3 + threadIdx.x];
    if (threadIdx.x < BLOCK / G)
        s_eps[threadIdx.x] = (pio.eps_dev && b0 + (int)threadIdx.x < p.B) ? pio.eps_dev[b0 + threadIdx.x] : pio.epsilon;
    // hidden state and last actions of the block's rows -> LDS
    const int srow = threadIdx.x >> 4, kcol = threadIdx.x & 15;
#pragma unroll
    for (int m = 0; m < N; m++) {
        const int r = 16 * m + srow;
        const size_t grow = (size_t)b0 * N + (r < rows_valid ? r : 0);
        const float4 hv = *reinterpret_cast<const float4 *>(pio.hidden + grow * H + 4 * kcol);
        *reinterpret_cast<float4 *>(s_h + r * LDW + 4 * kcol) = hv;
#if CS_POLICY_F16
        split_store(hs_base, hs_base + ROWS * HST, r * HST + 4 * kcol + 0, hv.x);   // hs[0]: what step 0 reads
        split_store(hs_base, hs_base + ROWS * HST, r * HST + 4 * kcol + 1, hv.y);
        split_store(hs_base, hs_base + ROWS * HST, r * HST + 4 * kcol + 2, hv.z);
        split_store(hs_base, hs_base + ROWS * HST, r * HST + 4 * kcol + 3, hv.w);
#endif
    }
    for (int r = threadIdx.x; r < ROWS; r += BLOCK) s_act[r] = r < rows_valid ? (int)pio.last[(size_t)b0 * N + r] : -1;
    // the current observation of every env goes into its wavefront's tile (what get_obs would return now)
    if (live) env_trig<N>(T, e);
    emit_deposit<N>(p, tile, t, grp, live, e, 0, false);
    // Draws: teams of up to 3 read them from the env's hit tape like the open-loop kernels (rows topped up here, once per launch; an env
    // that outlives its row falls back to twisting on demand inside detect_pass_tape); larger teams have no registers left for the
    // ten tape words and twist on demand throughout.
    constexpr bool USE_TAPE = N <= 3 && CS_RP_TAPE;
    __shared__ unsigned rowbufs[USE_TAPE ? BLOCK / 64 : 1][USE_TAPE ? MT_N : 1];
    MtWin win = {0u, 0u};
    unsigned tape[TAPE_DW];
    bool tape_ok = false;
    if (USE_TAPE) {
        if (live) tape_ok = tape_load(p, b, e, tape);
        if (wave_valid) group_wave_advance<N>(p, wave_b0, nvalid, lane, io.min_ahead, rowbufs[USE_TAPE ? w : 0], e, tape, tape_ok);
    } else if (live) {
        win = mt_prefetch(p.mt + (size_t)b * MT_STRIDE, e.mt_pos, t);
    }

    const int in_dim = 4 + NA + N;
#if CS_POLICY_F16
    // the env lane of agent t of env (w, grp) keeps row r's four observation columns of x current (tile.row is what emit_deposit left)
    auto put_obs_columns = [&]() __attribute__((always_inline)) {
        if (live && t < N) {
            const int r = (4 * w + grp) * N + t;
#pragma unroll
            for (int k = 0; k < 4; k++) split_store(x_hi, x_lo, r * HXS + k, tile.row[grp][4 * t + k]);
        }
    };
    {   // x once: agent-id one-hot, zero padding, the last action on entry; the observation columns as after every step
        __syncthreads();   // s_act and the tiles are complete
#pragma unroll
        for (int m = 0; m < N; m++) {
            const int r = 16 * m + srow, el = r / N, ag = r - el * N;
            float v = 0.0f;
            if (kcol >= 4 && kcol < 4 + NA) v = (kcol - 4 == s_act[r]) ? 1.0f : 0.0f;
            else if (kcol >= 4 + NA && kcol < in_dim) v = (kcol - 4 - NA == ag) ? 1.0f : 0.0f;
            split_store(x_hi, x_lo, r * HXS + kcol, r < rows_valid ? v : 0.0f);
            split_store(x_hi, x_lo, r * HXS + kcol + 16, 0.0f);
        }
        __syncthreads();   // (the observation columns below overwrite the zeros of columns 0..3)
        put_obs_columns();
    }
#endif
    for (int s = 0; s < io.T; s++) {
        LANE_STAMP(6);
        __syncthreads();   // x is complete (observation after the previous step, last action); the previous s_q has been consumed
        LANE_STAMP(7);
#if CS_POLICY_F16
        _Float16 *hc_hi = hs_base + (size_t)(s & 1) * 2 * ROWS * HST, *hc_lo = hc_hi + ROWS * HST;          // hidden state in
        _Float16 *hn_hi = hs_base + (size_t)((s + 1) & 1) * 2 * ROWS * HST, *hn_lo = hn_hi + ROWS * HST;    // hidden state out
        float *s_q = reinterpret_cast<float *>(hc_hi);   // [ROWS][17] floats (68 B per row <= a plane's 144 B), written after the GRU
#pragma unroll
        for (int m = 0; m < N; m++) {   // h1 = relu(W1 x + b1), columns 16w..16w+15 of every row tile (the bias enters the accumulator)
            f32x4 hi = splat4(bias1), lo = zero;
            h8 ah, al;
            load_afrag<HXS>(x_hi, x_lo, 16 * m, 0, lane, ah, al);
            mfma_split(ah, al, b1, hi, lo);
#pragma unroll
            for (int r = 0; r < 4; r++)
                split_store(b_hi, b_lo, (16 * m + crow + r) * HST + col, fmaxf(split_sum(hi[r], lo[r]), 0.0f));
        }
        LANE_STAMP(8);
        __syncthreads();
        LANE_STAMP(9);
#pragma unroll
        for (int m = 0; m < N; m++) {   // GRUCell of row tile m (gru_products / gru_cell: the very code k_policy_h runs)
            h8 xh[2], xl[2], hh[2], hl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                load_afrag(b_hi, b_lo, 16 * m, ks, lane, xh[ks], xl[ks]);
                load_afrag(hc_hi, hc_lo, 16 * m, ks, lane, hh[ks], hl[ks]);
            }
            GruAcc acc;
            gru_products(xh, xl, hh, hl, bg, b_r, b_z, bin, bhn, acc);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = 16 * m + crow + r;
                const float hnew = gru_cell(acc, r, s_h[o * LDW + col]);
                s_h[o * LDW + col] = hnew;                          // (this thread's own element)
                split_store(hn_hi, hn_lo, o * HST + col, hnew);     // h' for fc2 and for the next step's GRU
            }
        }
        LANE_STAMP(10);
        __syncthreads();
        LANE_STAMP(11);
#pragma unroll
        for (int m = 0; m < N; m++) {   // f = relu(W2 h' + b2)
            f32x4 hi = splat4(bias2), lo = zero;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                h8 ah, al;
                load_afrag(hn_hi, hn_lo, 16 * m, ks, lane, ah, al);
                mfma_split(ah, al, b2[ks], hi, lo);
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
                split_store(b_hi, b_lo, (16 * m + crow + r) * HST + col, fmaxf(split_sum(hi[r], lo[r]), 0.0f));
        }
        LANE_STAMP(12);
        __syncthreads();   // f complete
        LANE_STAMP(13);
        // q = W3 f + b3 and the choice of row tile m, by wavefront m % 4 alone (as in k_policy_h: no K split, no exchange of
        // partial sums, no barrier between the product and the selection)
#pragma unroll
        for (int m = 0; m < N; m++) {
            if ((m & 3) != w) continue;   // wave-uniform
            f32x4 hi = splat4(s_b3[ccol]), lo = zero;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                h8 ah, al;
                load_afrag(b_hi, b_lo, 16 * m, ks, lane, ah, al);
                mfma_split(ah, al, b3[ks], hi, lo);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) s_q[(16 * m + crow + r) * 17 + ccol] = split_sum(hi[r], lo[r]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 16) {   // argmax / epsilon-greedy, one lane per row
                const int r = 16 * m + lane;
                auto qf = [&](int a) { return s_q[r * 17 + a]; };
                const unsigned long long grow = pio.row0 + (unsigned long long)(b0 * N + r);
                const int er = r / N;   // the row's env within the block
                const double eps = s_eps[er < BLOCK / G ? er : 0];
                const int act = select_action(qf, NA, pio.select, (float)eps, pio.seed, pio.step0 + (unsigned)s, grow);
                s_act[r] = act;
#pragma unroll
                for (int a = 0; a < NA; a++) split_store(x_hi, x_lo, r * HXS + 4 + a, (r < rows_valid && a == act) ? 1.0f : 0.0f);
                if (r < rows_valid) {
                    pio.actions[((size_t)s * p.B + b0) * N + r] = act;
                    if (pio.trace && r == er * N) pio.trace[(size_t)s * p.B + b0 + er] = eps;
                }
            }
        }
        LANE_STAMP(14);
        __syncthreads();   // s_act is complete
        LANE_STAMP(15);
#else
        // (fp32 matrix path: the loop-top barrier above also covers the tiles / s_act of the previous step)
        // ---- x = obs(4) | one_hot(last action) | one_hot(agent id) per row (agent.py:41-52), one column per thread
#pragma unroll
        for (int m = 0; m < N; m++) {
            const int r = 16 * m + srow, el = r / N, ag = r - el * N;
            float v = 0.0f;
            if (kcol < 4) v = tiles[el >> 2].row[el & 3][4 * ag + kcol];
            else if (kcol < 4 + NA) v = (kcol - 4 == s_act[r]) ? 1.0f : 0.0f;
            else if (kcol < in_dim) v = (kcol - 4 - NA == ag) ? 1.0f : 0.0f;
            s_a[r * LDW + kcol] = r < rows_valid ? v : 0.0f;
        }
        __syncthreads();
        {   // h1 = relu(W1 x + b1), columns 16w..16w+15 of every row tile
            f32x4 acc[N];
#pragma unroll
            for (int m = 0; m < N; m++) acc[m] = zero;
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int m = 0; m < N; m++)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_a[(16 * m + (lane & 15)) * LDW + 4 * kk + (lane >> 4)],
                                                                  b1[kk], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < N; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) s_b[(16 * m + crow + r) * LDW + col] = fmaxf(acc[m][r] + bias1, 0.0f);
        }
        __syncthreads();
        {   // GRUCell: per row tile the six chains in k_policy's order
            f32x4 hnew[N];
#pragma unroll
            for (int m = 0; m < N; m++) {
                f32x4 ir = zero, iz = zero, in_ = zero, hr = zero, hz = zero, hn_ = zero;
#pragma unroll
                for (int kk = 0; kk < 16; kk++) {
                    const float ax = s_b[(16 * m + (lane & 15)) * LDW + 4 * kk + (lane >> 4)];
                    const float ah = s_h[(16 * m + (lane & 15)) * LDW + 4 * kk + (lane >> 4)];
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
                    hnew[m][r] = (1.0f - zg) * ng + zg * s_h[(16 * m + crow + r) * LDW + col];
                    s_a[(16 * m + crow + r) * LDW + col] = hnew[m][r];
                }
            }
            __syncthreads();   // every wavefront has finished reading s_h
#pragma unroll
            for (int m = 0; m < N; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) s_h[(16 * m + crow + r) * LDW + col] = hnew[m][r];
        }
        {   // f = relu(W2 h' + b2)
            f32x4 acc[N];
#pragma unroll
            for (int m = 0; m < N; m++) acc[m] = zero;
#pragma unroll
            for (int kk = 0; kk < 16; kk++)
#pragma unroll
                for (int m = 0; m < N; m++)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_a[(16 * m + (lane & 15)) * LDW + 4 * kk + (lane >> 4)],
                                                                  b2[kk], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < N; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) s_b[(16 * m + crow + r) * LDW + col] = fmaxf(acc[m][r] + bias2, 0.0f);
        }
        __syncthreads();   // f complete; s_a (h') no longer needed: its space now takes the partial q
        {
            f32x4 acc[N];
#pragma unroll
            for (int m = 0; m < N; m++) acc[m] = zero;
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int m = 0; m < N; m++)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        s_b[(16 * m + (lane & 15)) * LDW + 16 * w + 4 * kk + (lane >> 4)], b3f[kk], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < N; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) s_q[w * (ROWS * 17) + (16 * m + crow + r) * 17 + ccol] = acc[m][r];
        }
        __syncthreads();
        for (int r = threadIdx.x; r < ROWS; r += BLOCK) {   // argmax / epsilon-greedy, one thread per row
            auto qf = [&](int a) {
                const int o = r * 17 + a;
                return ((s_q[o] + s_q[ROWS * 17 + o]) + (s_q[2 * ROWS * 17 + o] + s_q[3 * ROWS * 17 + o])) + s_b3[a];
            };
            const unsigned long long grow = pio.row0 + (unsigned long long)(b0 * N + r);
            const int er = r / N;   // the row's env within the block
            const double eps = s_eps[er];
            const int act = select_action(qf, NA, pio.select, (float)eps, pio.seed, pio.step0 + (unsigned)s, grow);
            s_act[r] = act;
            if (r < rows_valid) {
                pio.actions[((size_t)s * p.B + b0) * N + r] = act;
                if (pio.trace && r == er * N) pio.trace[(size_t)s * p.B + b0 + er] = eps;
            }
        }
        __syncthreads();
#endif
        // ---- env.step with the chosen actions
        int act[N];
        const int el = 4 * w + grp;
#pragma unroll
        for (int i = 0; i < N; i++) act[i] = s_act[el * N + i];
        // will this env execute the step?  (step_once: an env terminated on entry is reset first under CS_AUTO_RESET, left alone
        // under CS_FREEZE_DONE): only executed steps anneal (the reference's episode loop has ended for a finished env)
        const bool executed = live && !((e.target_find >= p.n_targets || e.time_step >= p.time_limit) &&
                                        !(io.flags & CS_AUTO_RESET) && (io.flags & CS_FREEZE_DONE));
        if (wave_valid)
            step_once<N, 0>(p, T, io, tile, b, lane, (size_t)s * p.B + wave_b0, plan, live, act, win, s + 1 < io.T,
                            PIPE && s > 0, (size_t)(s - 1) * p.B + wave_b0, PIPE, e, tape, USE_TAPE, tape_ok);
        if (pio.per_step && pio.eps_dev && executed && t == 0) {   // epsilon = epsilon - anneal if epsilon > min else epsilon
            const double v = s_eps[el];
            s_eps[el] = v > pio.min_eps ? v - pio.anneal : v;
        }
#if CS_POLICY_F16
        if (wave_valid) put_obs_columns();   // the next forward's observation columns (emit_deposit has left them in the tile)
#endif
    }
    if (PIPE && wave_valid) {  // rows of the last step
        FlushRegs<N> fr;
        emit_flush_load<N>(tile, plan, fr);
        emit_flush_store<N>(p, io, plan, fr, (size_t)(io.T - 1) * p.B + wave_b0);
    }
    if (live) {
        env_store<N>(p, b, t, e, false);
        if (USE_TAPE && tape_ok) group_tape_store<N>(p, b, t, e, tape);
    }
    __syncthreads();
    if (pio.eps_dev && threadIdx.x < BLOCK / G && b0 + (int)threadIdx.x < p.B) pio.eps_dev[b0 + threadIdx.x] = s_eps[threadIdx.x];
#pragma unroll
    for (int m = 0; m < N; m++) {
        const int r = 16 * m + srow;
        if (r < rows_valid)
            *reinterpret_cast<float4 *>(pio.hidden + ((size_t)b0 * N + r) * H + 4 * kcol) =
                *reinterpret_cast<const float4 *>(s_h + r * LDW + 4 * kcol);
    }
}

// =========================================================================================================
// Lane-per-env path (flight_easy): one environment per LANE, 64 per wavefront.
//
// The 16-lane-group kernels above minimise the latency of one step when the batch is small (every SIMD gets a
// wave even at B = 4096) but replicate the kinematics 16 times.  For larger batches this path does each env's
// arithmetic exactly once and is built to keep TWO wavefronts per SIMD resident (<= 256 VGPRs, 2 x 4 staging
// tiles in LDS) so that one wavefront's memory waits hide behind the other's arithmetic:
//   * the agents live in the lane's registers; the targets do NOT: their normalised fp32 coordinates sit in the
//     lane's row of the staging tile anyway (get_state emits them every step), and the n*m sensor tests are
//     decided from those in fp32 whenever the fp32 distance is clear of the threshold by more than its error
//     bound -- the few pairs that are not (~2e-6 of them) re-read the fp64 target and run the reference's exact
//     comparison, so the outcome is the exact one in every case;
//   * MT19937: the state is regenerated AHEAD of consumption, 192 words of one env at a time by the whole
//     wavefront (three coalesced 256-byte loads and one store per 64 words instead of per-lane gathers),
//     `cs_layout.ahead_off` counting the words that are twisted but not yet consumed; a draw is then two loaded
//     words and a temper, and the 32 words a step may need are requested at the top of the step;
//   * the in-range pairs form a per-lane bitmask consumed in agent-major order (bit 16*i + j), get_state rows
//     leave through the per-wave LDS tile as one contiguous block;
//   * resets (data-dependent length) are done wave-cooperatively, four envs at a time, by the four 16-lane groups
//     of the wavefront running the group code above.
// Results are bit-identical to the group kernels (same per-env arithmetic, same MT19937 word order);
// tests/test_gpu_parity.py runs both.
// =========================================================================================================
#ifndef CS_LANE_REFRESH_MAX_N
#define CS_LANE_REFRESH_MAX_N 5   /* measured at B = 262144: 4 agents 29.9 -> 34-38 %, 5 agents 23.5 -> 28 % */
#endif
#ifndef CS_LANE_FROM_LARGE_TEAMS
#define CS_LANE_FROM_LARGE_TEAMS 1048576   /* ... for teams of 5 and more agents (lane_from) */
#endif
#ifndef CS_LANE_FROM
#define CS_LANE_FROM 131072     /* default kernel of cs_rollout from this many envs: one env per lane (65536: octet 7.6e9
                                   against lane 7.1e9 env-steps/s at 3 agents, 5.0e9 against 4.8e9 at 5; 262144: 8.0 / 10.4) */
#endif
#ifndef CS_LANEV_DEFAULT
#define CS_LANEV_DEFAULT 1      /* the lane-per-env kernel of teams of up to 5 is k_rollout_lanev (rollout_lanev.h) */
#endif
#ifndef CS_ODE_UPTO
#define CS_ODE_UPTO 8192        /* ... up to this many envs with the third (emitting) wavefront: four 3-wavefront workgroups per CU (32 KB of LDS each since E refreshes the rows: 10240 envs would need a fifth and run 3.7e9 against the pair variant's 5.0e9) x 256 CUs x 8 envs */
#endif
#ifndef CS_OD_UPTO
#define CS_OD_UPTO 16384        /* cs_rollout up to this many envs: the octet pair kernel */
#endif
#ifndef CS_OCT_FROM
#define CS_OCT_FROM 16384       /* cs_rollout above this many envs (and below CS_LANE_FROM): one env per 8 lanes, one wavefront */
#endif
constexpr int LANE_REFILL = 192;   // words twisted per refill (<= 227: independent of each other)
constexpr int LANE_REFILL_MAX = 192;
#ifndef CS_LANE_CHUNK
#define CS_LANE_CHUNK 64
#endif
constexpr int LANE_CHUNK = CS_LANE_CHUNK;     // steps per launch of the lane kernel: cs_rollout twists every row ahead in between

template <int N>
struct EnvL {
    double ax[N], ay[N], yaw[N], cs[N], sn[N];
    unsigned found, newly, newly_reset;
    int target_find, flags, time_step, total_reward, mt_pos, episodes, curr_reward, ahead;
    unsigned long long words;
};

template <int N>
__device__ __forceinline__ void envl_zero(EnvL<N> &e) {
#pragma unroll
    for (int i = 0; i < N; i++) e.ax[i] = e.ay[i] = e.yaw[i] = e.cs[i] = e.sn[i] = 0.0;
    e.found = e.newly = e.newly_reset = 0u;
    e.target_find = e.flags = e.time_step = e.total_reward = e.mt_pos = e.episodes = e.curr_reward = 0;
    e.ahead = 1 << 20;  // a lane without an env never asks for a refill
    e.words = 0ull;
}

// hdr / agents of env b into the lane's registers, its targets (normalised, fp32) into the lane's tile row
template <int N>
__device__ __forceinline__ void envl_load(const DevParams &p, int b, const double *T, float *row, EnvL<N> &e) {
    const int4 *h4 = reinterpret_cast<const int4 *>(p.hdr + (size_t)b * CS_H_WORDS);
    int4 h0 = h4[0], h1 = h4[1], h2 = h4[2];
    e.found = (unsigned)h0.x;
    e.newly = (unsigned)h0.y;
    e.target_find = h0.z;
    e.flags = h0.w;
    e.time_step = h1.x;
    e.total_reward = h1.y;
    e.mt_pos = h1.z;
    e.episodes = h1.w;
    e.words = (unsigned long long)(unsigned)h2.x | ((unsigned long long)(unsigned)h2.y << 32);
    e.curr_reward = h2.z;
    e.newly_reset = (unsigned)h2.w;
    e.ahead = p.ahead[b];
    const double4 *a4 = reinterpret_cast<const double4 *>(p.agent + (size_t)b * CS_MAX_AGENTS * 4);
#pragma unroll
    for (int i = 0; i < N; i++) {
        double4 a = a4[i];
        e.ax[i] = a.x;
        e.ay[i] = a.y;
        e.yaw[i] = a.z;
    }
    const double2 *t2 = reinterpret_cast<const double2 *>(p.tgt + (size_t)b * G * 2);
#pragma unroll
    for (int j = 0; j < CS_MAX_TARGETS; j++) {
        if (j < p.n_targets) {
            const double2 tt = t2[j];
            row[4 * N + 3 * j + 0] = (float)((tt.x - p.mid) * p.inv_half);   // what get_state emits (norm_target)
            row[4 * N + 3 * j + 1] = (float)((tt.y - p.mid) * p.inv_half);
            row[4 * N + 3 * j + 2] = ((e.found >> j) & 1u) ? 1.0f : 0.0f;
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) trig_heading(T, e.yaw[i], e.sn[i], e.cs[i]);
}

template <int N>
__device__ __forceinline__ void envl_store(const DevParams &p, int b, const EnvL<N> &e) {
    int4 *h4 = reinterpret_cast<int4 *>(p.hdr + (size_t)b * CS_H_WORDS);
    h4[0] = make_int4((int)e.found, (int)e.newly, e.target_find, e.flags);
    h4[1] = make_int4(e.time_step, e.total_reward, e.mt_pos, e.episodes);
    h4[2] = make_int4((int)(unsigned)(e.words & 0xffffffffull), (int)(unsigned)(e.words >> 32), e.curr_reward,
                      (int)e.newly_reset);
    p.ahead[b] = e.ahead;
    double4 *a4 = reinterpret_cast<double4 *>(p.agent + (size_t)b * CS_MAX_AGENTS * 4);
#pragma unroll
    for (int i = 0; i < N; i++) a4[i] = make_double4(e.ax[i], e.ay[i], e.yaw[i], 0.0);
}

// Kinematics of one lane's env: same contract as kinematics<> above, organised for 64 DIFFERENT envs per
// wavefront.  The repulsion of agent i (flight_env_easy.py:293-301) is a loop over the neighbours that ARE within
// force_dist, in ascending j like the reference's, instead of n-1 predicated copies of the two fp64 divisions:
// with 64 envs per wavefront some lane has a close pair almost every step, so every predicated copy would run.
template <int N>
__device__ __forceinline__ void kinematics_lane(const DevParams &p, const double *T, const int (&act)[N], EnvL<N> &e) {
    const double PI = 3.141592653589793, TWO_PI = 2.0 * 3.141592653589793, THREE_PI = 3.0 * 3.141592653589793;
    const double DYAW = 3.141592653589793 / 18.0;
    double yw[N], s1[N], c1[N], yr[N], s2[N], c2[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        double yaw = e.yaw[i];
        yaw = act[i] == 1 ? yaw + DYAW : (act[i] == 2 ? yaw + -DYAW : yaw);  // dyaw = [0, pi/18, -pi/18][act]
        yaw = yaw > TWO_PI ? yaw - TWO_PI : (yaw < 0.0 ? yaw + TWO_PI : yaw);
        yw[i] = yaw;
        yr[i] = (yaw <= PI) ? PI - yaw : THREE_PI - yaw;
        // (evaluating the 2n headings branch-free in one basic block so that their chains interleave was
        // measured: 39.1 -> 39.1 % at 2^18 envs, 41.7 -> 42.2 % at 2^20, for 32 more VGPRs: not kept here)
        trig_heading(T, yaw, s1[i], c1[i]);
        trig_heading(T, yr[i], s2[i], c2[i]);
    }
    unsigned out = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double x0 = e.ax[i], y0 = e.ay[i];
        unsigned pend = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            const double xa = e.ax[j], ya = e.ay[j];  // already moved if j < i (quirk Q7)
            const double d2 = (xa - x0) * (xa - x0) + (ya - y0) * (ya - y0);
            pend |= (d2 < p.force_d2 && (xa != x0 || ya != y0)) ? (1u << j) : 0u;
        }
        double fx = 0.0, fy = 0.0;
        while (pend) {
            const int j = __ffs((int)pend) - 1;
            pend &= pend - 1;
            double xa = 0.0, ya = 0.0;
#pragma unroll
            for (int q = 0; q < N; q++) {
                xa = q == j ? e.ax[q] : xa;
                ya = q == j ? e.ay[q] : ya;
            }
            const double den = (x0 - xa) * (x0 - xa) + (y0 - ya) * (y0 - ya);
            fx += p.force_k * (x0 - xa) / den;
            fy += p.force_k * (y0 - ya) / den;
        }
        const double x = (x0 + p.velocity * c1[i]) + fx;
        const double y = (y0 + p.velocity * s1[i]) + fy;
        const bool hit = (x < 0.0) | (x > p.L) | (y < 0.0) | (y > p.L);    // flight_env_easy.py:278
        e.ax[i] = hit ? fmin(fmax(x, 0.0), p.L) : x;
        e.ay[i] = hit ? fmin(fmax(y, 0.0), p.L) : y;
        e.yaw[i] = hit ? yr[i] : yw[i];
        e.cs[i] = hit ? c2[i] : c1[i];
        e.sn[i] = hit ? s2[i] : s1[i];
        out |= hit ? (1u << i) : 0u;
    }
    e.flags = (e.flags & ~0xff00) | (int)(out << 8);
}

// Fallback of the lane kernel (rare once cs_rollout's pre-pass has run): for every lane whose bit is set in `need`,
// the whole wavefront twists LANE_REFILL more words of that lane's env (when there is room) and rebuilds the env's hit
// tape from its cursor -- in the state blob and, through the ballots, in the lane's registers.
template <int N>
__device__ __forceinline__ void lane_rebuild(const DevParams &p, int b0, int lane, unsigned long long need, EnvL<N> &e,
                                             unsigned (&tape)[TAPE_DW]) {
    while (need) {
        const int src = __ffsll((long long)need) - 1;
        need &= need - 1;
        const int pos = __shfl(e.mt_pos, src);
        int a = __shfl(e.ahead, src);
        const unsigned wlo = (unsigned)__shfl((int)(unsigned)(e.words & 0xffffffffull), src);
        const unsigned whi = (unsigned)__shfl((int)(unsigned)(e.words >> 32), src);
        unsigned *m = p.mt + (size_t)(b0 + src) * MT_STRIDE;
        if (a <= MT_N - LANE_REFILL) {   // wave-uniform
            const int g = wrap624(pos + a);
            unsigned nw[3];
            int idx[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {   // word j needs stored words j, j+1, j+397: none written by this batch (192 <= 227)
                const int j = wrap624(g + 64 * c + lane);
                idx[c] = j;
                nw[c] = mt_mix(m[j], m[wrap624(j + 1)], m[wrap624(j + MT_M)]);
            }
#pragma unroll
            for (int c = 0; c < 3; c++) mt_store(m, idx[c], nw[c]);
            a += LANE_REFILL;
        }
        unsigned *tp = p.tape + (size_t)(b0 + src) * TAPE_STRIDE;
#pragma unroll
        for (int it = 0; it < TAPE_DW / 2; it++) {
            const int r = 64 * it + lane;   // draw slot from the cursor: words pos + 2r, pos + 2r + 1 (pos is even)
            bool hit = false;
            if (2 * r < a) {
                const U2 w = *reinterpret_cast<const U2 *>(m + wrap624(pos + 2 * r));
                hit = draw_hits(p, w.x, w.y);
            }
            const unsigned long long bm = __ballot(hit);
            if (lane == src) {
                tape[2 * it] = (unsigned)(bm & 0xffffffffull);
                tape[2 * it + 1] = (unsigned)(bm >> 32);
            }
            if (lane == 0) *reinterpret_cast<U2 *>(tp + 2 * it) = U2{(unsigned)(bm & 0xffffffffull), (unsigned)(bm >> 32)};
        }
        if (lane == 0) {
            *reinterpret_cast<U2 *>(tp + 10) = U2{wlo, whi};
            *reinterpret_cast<U2 *>(tp + 12) = U2{(unsigned)(p.detect_K & 0xffffffffull), (unsigned)(p.detect_K >> 32)};
        }
        if (lane == src) e.ahead = a;
    }
    drain_vmem();   // rare path: joins the steady-state path with nothing of its own in flight
}

// In-loop refresh of the lane kernel (teams of up to 3): instead of a separate pre-pass over every row, each wavefront
// tops up ONE of its 64 envs per step -- the one running lowest on twisted words: the env's row is requested at the end
// of a step (ten coalesced dwords per lane, held in registers), and after the next step's kinematics the wavefront
// copies it to LDS, twists everything that is not yet twisted (row_twist_ahead: new words go to the state blob) and
// rebuilds the env's hit tape straight into its lane's registers (ballots).  Each env comes round about every 64 steps,
// having consumed ~400 words: the MT19937 traffic (2.5 KB read + ~1.6 KB written per refresh) is spread under the
// arithmetic of the whole rollout, and no lane waits for words.
template <int N>
__device__ __forceinline__ void lane_advance_finish(const DevParams &p, int b0, int lane, int src, const RowRegs &rr,
                                                    unsigned *rowbuf, EnvL<N> &e, unsigned (&tape)[TAPE_DW]) {
    row_to_lds(rr, rowbuf, lane);
    const int pos = __shfl(e.mt_pos, src);
    const int a = __shfl(e.ahead, src);
    row_twist_ahead(rowbuf, p.mt + (size_t)(b0 + src) * MT_STRIDE, pos, a < 0 ? 0 : a, lane);
#pragma unroll
    for (int it = 0; it < TAPE_DW / 2; it++) {
        const unsigned long long bm = row_slot_hits(p, rowbuf, pos, it, lane);
        if (lane == src) {
            tape[2 * it] = (unsigned)(bm & 0xffffffffull);
            tape[2 * it + 1] = (unsigned)(bm >> 32);
        }
    }
    if (lane == src) e.ahead = MT_N;
}

// the same, start to finish, for every lane in `need` (kernel entry, or a lane that could not wait for its turn)
template <int N>
__device__ __forceinline__ void lane_advance_now(const DevParams &p, int b0, int lane, unsigned long long need, unsigned *rowbuf,
                                                 EnvL<N> &e, unsigned (&tape)[TAPE_DW]) {
    while (need) {
        const int src = __ffsll((long long)need) - 1;
        need &= need - 1;
        RowRegs rr;
        row_load(p.mt + (size_t)(b0 + src) * MT_STRIDE, lane, rr);
        lane_advance_finish<N>(p, b0, lane, src, rr, rowbuf, e, tape);
    }
    drain_vmem();   // rare path: joins the steady-state path with nothing of its own in flight
}

// Ordering inside one step (gfx9 has ONE in-order counter for vector loads and stores: waiting for a load also waits
// for every store issued before it): the only loads of the steady-state loop -- the next step's actions -- are requested
// before the step's output stores, and the number of stores between any load and its use is a compile-time constant, so
// no wait of the loop ever needs a store to have been acknowledged by the memory system.  Rare paths (reset, tape
// rebuild) end with nothing of their own in flight.
//
// VEC (every wavefront of the launch is full and every step's block of rows is 16-byte aligned; the host splits a batch
// into a VEC launch and, for the last < 64 envs or an unaligned tensor, a plain one): the 64 get_state rows of step s
// leave the tile as float4 chunks DURING step s + 1 -- a third after the kinematics, a third after the sensor tests, a
// third after the draws -- so the write stream of a wavefront is spread over its arithmetic instead of arriving as
// one burst per step.
template <int N, bool VEC>
__global__ __launch_bounds__(BLOCK, 2) void k_rollout_lane(DevParams p, StepIO io) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *T = reinterpret_cast<double *>(smem);                                   // trig table (2072 B)
    const int W = 4 * N + 3 * p.n_targets;
    float *tiles = reinterpret_cast<float *>(smem + ((TRIG_ROWS * TRIG_COLS * 8 + 15) / 16) * 16);
    int lane = threadIdx.x & 63;   // (made opaque once per step, see the loop)
    const int wave = threadIdx.x >> 6;
    float *tile = tiles + (size_t)wave * 64 * W;
    float *row = tile + (size_t)lane * W;   // W is odd for m = 15: conflict-free column accesses
    const int b = io.env0 + blockIdx.x * BLOCK + threadIdx.x;
    const int b0 = b - lane;  // first env of this wavefront
    const int b_end = io.env0 + io.env_n;
    const bool live = b < b_end;
    __shared__ double rtab[4 * G];   // the reset's target tables (load_reset_tab)
    if (wave == 0) load_reset_tab(rtab, lane);
    load_trig_to_lds(T);
    if (b0 >= b_end) return;  // whole wavefront out of range
    const int t16 = lane & (G - 1), gshift = lane & ~(G - 1), grp = lane >> 4;
    const unsigned tmask = p.n_targets >= 16 ? 0xffffu : ((1u << p.n_targets) - 1u);
    constexpr int LOW = 2 * N * CS_MAX_TARGETS;   // words one step can consume: every lane enters a step with that many twisted
    constexpr bool REFRESH = N <= CS_LANE_REFRESH_MAX_N;   // in-loop refresh (above); larger teams rely on cs_rollout's pre-pass
    unsigned *rowbuf = reinterpret_cast<unsigned *>(tiles + (size_t)(BLOCK / 64) * 64 * W) + wave * MT_N;
    RowRegs rr;
    int cand = -1;                                // env (lane) whose row is in flight in `rr`
    EnvL<N> e;
    int act[N];
    unsigned tape[TAPE_DW];
    const size_t arow = live ? (size_t)b : 0;
    bool tape_ok = true;
    if (live) {
        envl_load<N>(p, b, T, row, e);
        tape_ok = tape_load(p, b, e, tape);
    } else {
        envl_zero<N>(e);
#pragma unroll
        for (int k = 0; k < TAPE_DW; k++) tape[k] = 0u;
    }
    if (REFRESH) {   // (an advance also rebuilds a tape that does not match the cursor or the detection threshold)
        const unsigned long long low = __ballot(live && (!tape_ok || e.ahead < LOW));
        if (low) lane_advance_now<N>(p, b0, lane, low, rowbuf, e, tape);
    } else {
        while (const unsigned long long low = __ballot(live && (!tape_ok || e.ahead < LOW))) {
            lane_rebuild<N>(p, b0, lane, low, e, tape);
            tape_ok = true;
        }
    }
    load_actions<N>(io, arow, act);
    const int rows_valid = b_end - b0 < 64 ? b_end - b0 : 64;
    constexpr int W_MAX = 4 * N + 3 * CS_MAX_TARGETS;
    constexpr int Q = (16 * W_MAX + 63) / 64;   // float4 chunks per lane of the largest tile
    // float4 chunks [q0, q1) of the tile -> rows of step `step`.  Chunk k = min(lane + 64 q, last): surplus lanes repeat
    // the last chunk (same value, same address), so every lane stores every time.
    auto copy_chunks = [&](int q0, int q1, size_t step) __attribute__((always_inline)) {
        const float4 *src4 = reinterpret_cast<const float4 *>(tile);
        float4 *dst4 = reinterpret_cast<float4 *>(io.state + (step * p.B + b0) * W);
        const int last = 16 * W - 1;
        int l0 = lane;
        asm volatile("" : "+v"(l0));   // the address pairs are recomputed at every use (hoisted out of the loop they spill)
#pragma unroll
        for (int q = q0; q < q1; q++) {
            const int k = l0 + 64 * q < last ? l0 + 64 * q : last;
            const float4 v = src4[k];
            const v4f nv = {v.x, v.y, v.z, v.w};   // write-once stream: non-temporal (+6 % on the whole kernel)
            __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst4 + k));
        }
    };
    bool flushed = true;   // VEC: the tile holds no step that still has to be written out
    for (int s = 0; s < io.T; s++) {
        asm volatile("" : "+v"(lane));   // lane predicates are recomputed per step instead of being held (and spilled) as SGPR pairs
        const size_t slot = (size_t)s * p.B + arow;
        LANE_STAMP(0);
        REAL_STAMP(8);
        bool done = live && (e.target_find >= p.n_targets || e.time_step >= p.time_limit);
        // ---- auto-reset: the four 16-lane groups of the wavefront each take one resetting env per round.  The env's
        //      cursor goes to its group by shuffle, the new targets come back through the lane's tile row (fp64 copies
        //      go to the state blob without anybody waiting for them), the counters by shuffle: the only memory round
        //      trip of a reset is the MT19937 words of its polar-gaussian attempts.
        const unsigned long long need = __ballot(done && (io.flags & CS_AUTO_RESET));
        if (need) {
            if (VEC && !flushed) copy_chunks(0, Q, (size_t)(s - 1));   // the resets rewrite rows of the tile
            flushed = true;
            if (REFRESH && cand >= 0 && ((need >> cand) & 1ull)) cand = -1;   // its cursor moves: the row in flight is void
            const bool mine = (need >> lane) & 1ull;
            const int my_rank = __popcll(need & ((1ull << lane) - 1ull));
            unsigned long long pend = need;
            for (int round = 0; pend; round++) {
                unsigned long long m = pend;
                for (int q = 0; q < grp; q++) m &= m ? m - 1 : 0ull;   // this group's env: the grp-th pending one
                const int src = m ? __ffsll((long long)m) - 1 : -1;
                for (int q = 0; q < 4; q++) pend &= pend ? pend - 1 : 0ull;
                const int sl = src >= 0 ? src : lane;
                Env<N> g;
                g.mt_pos = __shfl(e.mt_pos, sl);
                g.ahead = __shfl(e.ahead, sl);
                g.episodes = __shfl(e.episodes, sl);
                g.words = (unsigned long long)(unsigned)__shfl((int)(unsigned)(e.words & 0xffffffffull), sl) |
                          ((unsigned long long)(unsigned)__shfl((int)(unsigned)(e.words >> 32), sl) << 32);
                g.newly_reset = 0u;
                g.curr_reward = 0;
                g.tx = g.ty = 0.0;
                if (src >= 0) {
                    const int br = b0 + src;
                    const DevParams &cp = cold_params();
                    // (with 64 envs per wavefront there is a reset in nearly every step: through env_reset alone they cost 16 %
                    // of the kernel at 2^18 envs; env_reset_fast: the lean path for the usual case)
                    env_reset_fast<N, false>(cp, T, rtab, br, t16, gshift, g);
                    reinterpret_cast<double2 *>(cp.tgt + (size_t)br * G * 2)[t16] = make_double2(g.tx, g.ty);
                    if (t16 < p.n_targets) {
                        float *rs = tile + (size_t)src * W + 4 * N + 3 * t16;
                        rs[0] = g.ntx;
                        rs[1] = g.nty;
                        rs[2] = ((g.found >> t16) & 1u) ? 1.0f : 0.0f;
                    }
                }
                // the q-th pending env of this round was reset by group q: its (group-uniform) counters come back
                const int q = my_rank - 4 * round;
                const bool got = mine && q >= 0 && q < 4;
                const int leader = got ? 16 * q : lane;
                const int r_pos = __shfl(g.mt_pos, leader), r_ahead = __shfl(g.ahead, leader);
                const int r_epi = __shfl(g.episodes, leader), r_tf = __shfl(g.target_find, leader);
                const int r_flags = __shfl(g.flags, leader), r_cr = __shfl(g.curr_reward, leader);
                const int r_found = __shfl((int)g.found, leader), r_newly = __shfl((int)g.newly, leader);
                const int r_wlo = __shfl((int)(unsigned)(g.words & 0xffffffffull), leader);
                const int r_whi = __shfl((int)(unsigned)(g.words >> 32), leader);
                if (got) {
                    const unsigned long long w_new = (unsigned long long)(unsigned)r_wlo | ((unsigned long long)(unsigned)r_whi << 32);
                    // the reset consumed (w_new - words) stream words, twisted ones first: their draw slots leave the tape
                    const unsigned long long used = w_new - e.words;
                    tape_shift<8>(tape, used < 2ull * 319ull ? (int)(used >> 1) : 319);
                    e.mt_pos = r_pos;
                    e.ahead = r_ahead;
                    e.episodes = r_epi;
                    e.target_find = r_tf;
                    e.flags = r_flags;
                    e.curr_reward = r_cr;
                    e.found = (unsigned)r_found;
                    e.newly = (unsigned)r_newly;
                    e.words = w_new;
                    e.time_step = 0;
                    e.total_reward = 0;
                    {   // start poses: the host's table; every agent starts with the same heading: one evaluation
                        const StartTab<N> st = start_tab<N>();
                        double s0, c0;
                        trig_heading(T, st.yaw, s0, c0);
#pragma unroll
                        for (int i = 0; i < N; i++) {
                            e.ax[i] = st.x[i];
                            e.ay[i] = st.y[i];
                            e.yaw[i] = st.yaw;
                            e.sn[i] = s0;
                            e.cs[i] = c0;
                        }
                    }
                    done = false;
                }
            }
            // a reset that ran past the twisted words leaves its lane without a tape for this step: rebuild
            if (REFRESH) {
                const unsigned long long low = __ballot(e.ahead < LOW);
                if (low) {
                    if (cand >= 0 && ((low >> cand) & 1ull)) cand = -1;
                    lane_advance_now<N>(p, b0, lane, low, rowbuf, e, tape);
                }
            } else {
                while (const unsigned long long low = __ballot(e.ahead < LOW)) lane_rebuild<N>(p, b0, lane, low, e, tape);
            }
            drain_vmem();
        }
        // rows still to be written out: step s - 1's; after a flush (or at s = 0) the same chunks go to step s's own
        // slot instead, which this wavefront overwrites with the real rows one step later
        const size_t cstep = (size_t)(flushed ? s : s - 1);
        LANE_STAMP(1);
        int reward = 0;
        bool term = true;
        const bool stepping = live && !(done && (io.flags & CS_FREEZE_DONE));
        e.flags &= ~(FLAG_DIRTY | FLAG_RESET_PASS);
        if (stepping) kinematics_lane<N>(p, T, act, e);
        if (VEC) copy_chunks(0, Q / 3, cstep);
        if (REFRESH && cand >= 0) {   // wave-uniform: the row requested a step ago has long arrived
            lane_advance_finish<N>(p, b0, lane, cand, rr, rowbuf, e, tape);
            cand = -1;
        }
        LANE_STAMP(2);
        float4 f[N];
#pragma unroll
        for (int i = 0; i < N; i++)
            f[i] = make_float4((float)((e.ax[i] - p.mid) * p.inv_half), (float)((e.ay[i] - p.mid) * p.inv_half),
                               (float)e.cs[i], (float)e.sn[i]);
        // ---- sensor tests (flight_env_easy.py:237): fp32 pre-filter on the normalised coordinates, exact fp64
        //      comparison for the pairs it cannot decide; bit 16*i + j = (agent i, target j) in range
        unsigned long long lo = 0, hi = 0;  // agents 0..3 / 4..7
        if (stepping) {
            float ntx[CS_MAX_TARGETS], nty[CS_MAX_TARGETS];
#pragma unroll
            for (int j = 0; j < CS_MAX_TARGETS; j++) {
                ntx[j] = j < p.n_targets ? row[4 * N + 3 * j + 0] : 0.0f;
                nty[j] = j < p.n_targets ? row[4 * N + 3 * j + 1] : 0.0f;
            }
            const float thr_lo = p.thr32 - p.eps32, thr_hi = p.thr32 + p.eps32;
#pragma unroll
            for (int i = 0; i < N; i++) {
                // sign bits of d2 - thr_lo / d2 - thr_hi, target 15 first, funnel-shifted into the masks (one
                // v_alignbit each): bit j of `sure` = (d2 < thr - eps), of `maybe` = (d2 < thr + eps)
                unsigned sure = 0, maybe = 0;
#pragma unroll
                for (int j = CS_MAX_TARGETS - 1; j >= 0; j--) {
                    const float dx = ntx[j] - f[i].x, dy = nty[j] - f[i].y;
                    const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                    sure = __builtin_amdgcn_alignbit(sure, __float_as_uint(d2 - thr_lo), 31);
                    maybe = __builtin_amdgcn_alignbit(maybe, __float_as_uint(d2 - thr_hi), 31);
                }
                unsigned m = sure & tmask;
                unsigned fz = maybe & ~sure & tmask;
                while (fz) {  // (t_x-x)**2 + (t_y-y)**2 <= view_range**2 on the fp64 values
                    const int j = __ffs((int)fz) - 1;
                    fz &= fz - 1;
                    const double2 tt = reinterpret_cast<const double2 *>(p.tgt + (size_t)b * G * 2)[j];
                    const double ddx = tt.x - e.ax[i], ddy = tt.y - e.ay[i];
                    m |= (ddx * ddx + ddy * ddy <= p.view_r2 ? 1u : 0u) << j;
                }
                if (i < 4) lo |= (unsigned long long)m << (16 * i);
                else hi |= (unsigned long long)m << (16 * (i - 4));
            }
        }
        if (VEC) copy_chunks(Q / 3, 2 * Q / 3, cstep);
        LANE_STAMP(3);
        // ---- one np.random.rand() per in-range pair, found or not (quirk Q4), in agent-major order: the r-th set bit
        //      of (lo, hi) takes draw slot r of the tape
        const int total = __popcll(lo) + (N > 4 ? __popcll(hi) : 0);
        unsigned hitmask = 0;
        {
            unsigned w16 = tape[0];
            for (int r0 = 0; __ballot(r0 < total); r0 += 16) {
                const int take = total - r0;   // <= 0: nothing left for this lane
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k < take) {
                        int bit;
                        if (N <= 4 || lo) {
                            bit = __ffsll((long long)lo) - 1;
                            lo &= lo - 1;
                        } else {
                            bit = __ffsll((long long)hi) - 1;
                            hi &= hi - 1;
                        }
                        hitmask |= ((w16 >> k) & 1u) << (bit & 15);
                    }
                }
                // slots r0 + 16 ..: (r0 is wave-uniform, so the tape dword is picked with uniform selects)
                const int nx = r0 + 16;
                unsigned nxt = 0;
#pragma unroll
                for (int d = 0; d < (N * CS_MAX_TARGETS + 31) / 32; d++) nxt = (nx >> 5) == d ? tape[d] : nxt;
                w16 = nxt >> (nx & 31);
            }
            e.mt_pos = wrap624(e.mt_pos + 2 * total);
            e.words += (unsigned long long)(2 * total);
            e.ahead -= 2 * total;
            tape_shift<(N * CS_MAX_TARGETS) / 32 < 1 ? 1 : (N * CS_MAX_TARGETS) / 32>(tape, total);
        }
        if (VEC) copy_chunks(2 * Q / 3, Q, cstep);
        LANE_STAMP(4);
        if (stepping) {
            const unsigned newly = hitmask & ~e.found;
            const int cnt = __popc(newly);
            int r = -1;     // MOVE_COST
            r += 10 * cnt;  // FIND_ONE_TGT
            e.found |= newly;
            e.newly = newly;
            e.target_find += cnt;
            if (cnt > 0 && e.target_find == p.n_targets && !(e.flags & FLAG_WIN)) {
                r += 100;  // FIND_ALL_TGT
                e.flags |= FLAG_WIN;
            }
            r -= __popc(((unsigned)e.flags >> 8) & 0xffu);  // OUT_PUNISH
            e.curr_reward = r;
            e.flags |= FLAG_DIRTY;
            reward = r;
            e.total_reward += reward;
            e.time_step += 1;
            term = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
            if (newly) {
#pragma unroll
                for (int j = 0; j < CS_MAX_TARGETS; j++)
                    if ((newly >> j) & 1u) row[4 * N + 3 * j + 2] = 1.0f;
            }
        }
        if (live && (io.state || io.obs)) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                row[4 * i + 0] = f[i].x;
                row[4 * i + 1] = f[i].y;
                row[4 * i + 2] = f[i].z;
                row[4 * i + 3] = f[i].w;
            }
        }
        LANE_STAMP(5);
        // ---- what the next step waits for, requested BEFORE this step's stores: the row of the env to refresh next (or,
        //      rarely, an immediate advance / tape rebuild), the next actions
        if (REFRESH) {
            const unsigned long long low = __ballot(e.ahead < LOW);
            if (low) lane_advance_now<N>(p, b0, lane, low, rowbuf, e, tape);
            const unsigned long long urgent = __ballot(e.ahead < 192), normal = __ballot(e.ahead < 352);
            cand = urgent ? __ffsll((long long)urgent) - 1 : (normal ? __ffsll((long long)normal) - 1 : -1);
            if (cand >= 0) row_load(p.mt + (size_t)(b0 + cand) * MT_STRIDE, lane, rr);
        } else {
            while (const unsigned long long low = __ballot(e.ahead < LOW)) lane_rebuild<N>(p, b0, lane, low, e, tape);
        }
        load_actions<N>(io, (size_t)(s + 1 < io.T ? s + 1 : s) * p.B + arow, act);
        LANE_STAMP(6);
        // ---- this step's outputs
        if (live) {
            io.reward[slot] = (float)reward;
            io.terminated[slot] = term ? 1 : 0;
            io.win[slot] = (e.flags & FLAG_WIN) ? 1 : 0;
        }
        if (io.obs) {
            // get_obs: the wavefront's 64 N float4 are one contiguous block of the table; stored from the lanes that own the envs
            // they would be N stores of 64 pieces at a stride of 16 N bytes each (partial sectors, which non-temporal stores do not
            // let the L2 merge).  The agents' floats are in the tile already (the get_state rows): chunk k = (env k / N, agent k % N)
            // is gathered from there and the block leaves as N coalesced 1 KB stores.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v4f *o = reinterpret_cast<v4f *>(io.obs) + ((size_t)s * p.B + b0) * N;
            v4f ov[N];
#pragma unroll
            for (int q = 0; q < N; q++) {
                const int k = lane + 64 * q, r = k / N, i = k - r * N;
                const float *src = tile + (size_t)r * W + 4 * i;
                ov[q] = v4f{src[0], src[1], src[2], src[3]};
            }
#pragma unroll
            for (int q = 0; q < N; q++)
                if (lane + 64 * q < rows_valid * N) __builtin_nontemporal_store(ov[q], o + lane + 64 * q);
        }
        if (!VEC && io.state) {   // plain launch: the wave's rows (contiguous in get_state's [B][W] layout) leave now
            float *dst = io.state + ((size_t)s * p.B + b0) * W;
            for (int k = lane; k < rows_valid * W; k += 64) dst[k] = tile[k];
        }
        flushed = false;
        LANE_STAMP(7);
    }
    if (VEC) copy_chunks(0, Q, (size_t)(io.T - 1));
    if (live) {
        envl_store<N>(p, b, e);
        if (REFRESH) {   // the tape lives in registers here: leave it, rebased to the cursor, for the next launch
            U4 *tp = reinterpret_cast<U4 *>(p.tape + (size_t)b * TAPE_STRIDE);
            tp[0] = U4{tape[0], tape[1], tape[2], tape[3]};
            tp[1] = U4{tape[4], tape[5], tape[6], tape[7]};
            tp[2] = U4{tape[8], tape[9], (unsigned)(e.words & 0xffffffffull), (unsigned)(e.words >> 32)};
            tp[3] = U4{(unsigned)(p.detect_K & 0xffffffffull), (unsigned)(p.detect_K >> 32), 0u, 0u};
        }
    }
}


// =========================================================================================================
// Octet path (flight_easy): one environment per EIGHT lanes, 8 per wavefront -- the rollout kernel between the
// 16-lane pair kernel (B <= 4096) and the HBM regime.
//
// The 16-lane kernels replicate all n agents in every lane (5 doubles per agent and lane: 252 VGPRs at 5 agents, two
// wavefronts per SIMD), so from 8192 envs up a batch no longer fits the chip in one resident round and a 100-step
// launch runs its rounds one after the other (profiles/r02_batch_sweep.md: 2x per step from 8192 to 16384 envs).
// Here nothing about an env is replicated except its header:
//   * lane t < n of the octet OWNS agent t (n <= 8 = lanes): its position, heading and the two correctly rounded
//     trig evaluations of a step live in that lane only; the team's positions meet in LDS (OctShared.pos) for the
//     proximity test and for the sensor tests;
//   * lane t owns targets t and t + 8 (<= 16 targets): 2n sensor tests per lane, the in-range mask of an agent is two
//     ballots, a prefix popcount of the octet's 16 bits gives every in-range pair its draw slot in the reference's
//     agent-major order, exactly as in the 16-lane kernels; draws are bits of the env's hit tape;
//   * kinematics: every agent is first moved as if the repulsion were zero and every ordered pair (i, j) is tested the
//     way the reference would test it (agent i's pre-move position against j's already moved position if j < i);
//     an octet with a pair in range (3-14 % of env-steps) loads the team into registers and runs the reference's
//     sequential loop (quirk Q7) with the repulsion as a loop over the neighbours that ARE in range;
//   * the get_state rows of the wavefront's 8 envs sit in a persistent LDS tile (targets' normalised coordinates are
//     written once per episode, found flags when they change, the agents' four floats every step) and leave as
//     float4 chunks, non-temporal; the next step's actions are requested before the step's stores (one in-order
//     counter for loads and stores: the wait for the actions then never waits for a store);
//   * resets run wave-cooperatively on the 16-lane reset code above (four resetting envs per round, one per 16-lane
//     group), results handed back through LDS / shuffles; MT19937 rows are topped up in place, whole wavefront on one
//     row, when an env is about to run out of twisted words.
// ~45 persistent VGPRs per lane instead of ~130: four and more wavefronts per SIMD, i.e. 32768+ envs in one resident
// round, and per-env arithmetic that is exactly the 16-lane kernels' (same functions / same expression order), so
// the results are bit-identical (tests/test_gpu_parity.py compares every kernel with the CPU restatement of the reference and with every other kernel).
// =========================================================================================================
#ifndef CS_OCT_WAVES
#define CS_OCT_WAVES 3                     /* wavefronts per SIMD the register budget must allow (168 VGPRs): measured 2 / 3 / 4,
                                              3 agents 16384 envs 3.00 / 3.10 / 3.25 us per step, 32768: 6.04 / 5.62 / 5.58;
                                              5 agents 16384: 4.51 / 4.63 / 4.87, 32768: 8.34 / 7.67 / 7.37 (at 4 the cold paths spill) */
#endif
constexpr int OG = 8;                      // lanes per env
constexpr int OCT_ENVS = 64 / OG;          // envs per wavefront
constexpr int OCT_BLOCK = 256;             // 4 wavefronts = 32 envs
constexpr int OCT_PAD = CS_MAX_AGENTS + 1; // row of 8 double2 padded to 144 bytes: the 8 octets' rows fall in distinct banks

struct __attribute__((aligned(16))) OctShared {
    double2 pos[OCT_ENVS][OCT_PAD];        // current (x, y) of agent j of octet o
    double2 tgt[OCT_ENVS][CS_MAX_TARGETS]; // reset: the new targets on their way from the 16-lane group to the octet
    float tile[OCT_ENVS * TILE_W];         // get_state rows of the 8 envs, stride W = 4n + 3m floats
    float reward[OCT_ENVS];
    int term[OCT_ENVS], win[OCT_ENVS];
    unsigned rowbuf[MT_N];                 // one MT19937 row (top-ups)
};

// Lanes per env of the "octet" kernels.  LG = 8 is the octet: 8 envs per wavefront, lane t owns agent t and targets t, t + 8.
// LG = 5 (round 5, teams of exactly 5 with at most 15 targets, k_rollout_od only; built with -DCS_OD_PENT=1 -- measured slower than the
// octet at the batches the pair kernels serve, see CS_OD_PENT): the same roles, protocol and arithmetic with FIVE
// lanes per env -- three envs per 16-lane DPP row (lanes 0-4, 5-9, 10-14; lane 15 of every row holds nothing), TWELVE envs per
// wavefront, lane t owns agent t and targets t, t + 5, t + 10.  Nothing about the octet is left idle by a team of 5 then: the
// kinematics wavefront does the work of twelve envs in the instructions it spent on eight, the detection pass tests three targets per
// lane instead of two for half again as many envs.  A lane that holds nothing reports t = 16 (no agent, no target below 16, never
// "lane 0 of its env"), aliases the last env of its row for reads and is never `live`.
template <int LG>
struct OctLay;
template <>
struct OctLay<8> {
    static constexpr int ENVS = 8, TPL = 2;   // envs per wavefront, targets per lane
    static constexpr unsigned SLICE = 0xffu;
    static __device__ __forceinline__ bool valid(int) { return true; }
    static __device__ __forceinline__ int env(int lane) { return lane >> 3; }
    static __device__ __forceinline__ int t(int lane) { return lane & 7; }
    static __device__ __forceinline__ int first(int lane) { return lane & ~7; }          // first lane of the lane's env = shift of its ballot slice
    static __device__ __forceinline__ int first_of(int o) { return 8 * o; }
    static __device__ __forceinline__ int env_of_first(int f) { return f >> 3; }
    static constexpr unsigned long long lanes_t(int I) { return 0x0101010101010101ull << I; }   // the lanes with t == I
};
template <>
struct OctLay<5> {
    static constexpr int ENVS = 12, TPL = 3;
    static constexpr unsigned SLICE = 0x1fu;
    static __device__ __forceinline__ int grp(int lane) { return ((lane & 15) * 13) >> 6; }   // 0, 1, 2; 3 for lane 15 of a row
    static __device__ __forceinline__ bool valid(int lane) { return (lane & 15) != 15; }
    static __device__ __forceinline__ int env(int lane) { const int g = grp(lane); return 3 * (lane >> 4) + (g < 3 ? g : 2); }
    static __device__ __forceinline__ int t(int lane) { const int g = grp(lane); return g < 3 ? (lane & 15) - 5 * g : 16; }
    static __device__ __forceinline__ int first(int lane) { const int g = grp(lane); return (lane & ~15) + 5 * (g < 3 ? g : 2); }
    static __device__ __forceinline__ int first_of(int o) { return 16 * (o / 3) + 5 * (o % 3); }
    static __device__ __forceinline__ int env_of_first(int f) { return 3 * (f >> 4) + (((f & 15) * 13) >> 6); }
    static constexpr unsigned long long lanes_t(int I) { return 0x0421042104210421ull << I; }
};

template <int N, int LG = OG>
struct EnvO {
    double x, y, yaw, cs, sn;              // this lane's agent (lanes t < N)
    double tx[OctLay<LG>::TPL], ty[OctLay<LG>::TPL];   // targets t + LG k
    unsigned found, newly, newly_reset;    // octet-uniform from here on
    int target_find, flags, time_step, total_reward, mt_pos, episodes, curr_reward, ahead;
    unsigned long long words;
};

// The env's slice of a wavefront ballot (bit k = lane first + k)
template <int LG = OG>
__device__ __forceinline__ unsigned oct_slice(unsigned long long ballot, int sh8) { return (unsigned)(ballot >> sh8) & OctLay<LG>::SLICE; }

// trig_heading for TWO headings at once (a step's new heading and its wall reflection).  Same arithmetic per heading, value for
// value; the difference is control flow: trig_heading ends in a branch for off-grid headings (only reachable by editing the raw
// state), which splits the two evaluations into separate basic blocks that the compiler schedules one after the other --
// two dependent chains of ~25 fp64 operations in series.  Here both on-grid evaluations sit in one block (the chains
// interleave) and ONE rarely-taken branch afterwards redoes whichever heading was off the grid.
__device__ __forceinline__ void trig_heading_pair(const double *T, double ya, double yb, double &sa, double &ca, double &sb,
                                                  double &cb) {
    double dh[2], s[2], c[2];
    const double *rr[2];
    const double y[2] = {ya, yb};
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int k = (int)(y[q] * 5.729577951308232 + 0.5);  // 18/pi
        k = k < 0 ? 0 : (k > 36 ? 36 : k);
        const double *r = T + k * TRIG_COLS;
        const double t = y[q] - r[0];  // exact (Sterbenz) for headings on the pi/18 grid
        const double d = t - r[1];
        const double bb = d - t;
        const double err = (t - (d - bb)) + ((-r[1]) - bb);  // TwoSum tail
        const double dl = err - r[2];
        s[q] = r[3] + ((r[4] + d * (r[5] - 0.5 * d * r[3])) + dl * r[5]);
        c[q] = r[5] + ((r[6] - d * (r[3] + 0.5 * d * r[5])) - dl * r[3]);
        dh[q] = d;
        rr[q] = r;
    }
    if (__builtin_expect((fabs(dh[0]) > 1e-6) | (fabs(dh[1]) > 1e-6), 0)) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (fabs(dh[q]) > 1e-6) {   // off-grid heading: the series of trig_heading, same operations
                const double d = dh[q], d2 = d * d;
                const double sd = d * (1.0 + d2 * (-1.0 / 6 + d2 * (1.0 / 120 + d2 * (-1.0 / 5040 + d2 * (1.0 / 362880)))));
                const double cd = 1.0 + d2 * (-0.5 + d2 * (1.0 / 24 + d2 * (-1.0 / 720 + d2 * (1.0 / 40320 + d2 * (-1.0 / 3628800)))));
                s[q] = rr[q][3] * cd + rr[q][5] * sd;
                c[q] = rr[q][5] * cd - rr[q][3] * sd;
            }
        }
    }
    sa = s[0];
    ca = c[0];
    sb = s[1];
    cb = c[1];
}

// 64-bit DPP move: lane L of every 16-lane row receives the value of lane L - k (row_shr:k) / L + k (row_shl:k)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    // bound_ctrl: a lane whose source lies outside its 16-lane row receives 0 -- what the zero "old" operand gave before, without the
    // two v_mov 0 per moved double that operand cost (16 VALU instructions per repulsion stage at 5 agents)
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u & 0xffffffffull), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
// lane I of every octet receives the value of lane J of the same octet (octets are aligned halves of the 16-lane DPP rows)
template <int I, int J>
__device__ __forceinline__ double oct_from(double v) {
    static_assert(I != J && I >= 0 && J >= 0 && I < OG && J < OG, "lanes of one env (which never straddles a 16-lane row)");
    return dpp_f64<(I > J) ? (0x110 | (I - J)) : (0x100 | (J - I))>(v);   // row_shr : row_shl
}
// fx, fy in lane I = sum over the neighbours J != I, ASCENDING J like the reference's loop (flight_env_easy.py:296-300), of
// the contributions (tx, ty) lane J computed.  Contributions of neighbours out of range are +0.0, which never changes a
// partial sum (no partial sum is ever -0.0: they start from +0.0, and +0.0 + -0.0 = +0.0 = x - x), so padding with them is exact.
template <int N, int I, int J = 0>
struct OctForceSum {
    static __device__ __forceinline__ void run(double tx, double ty, double &fx, double &fy) {
        if constexpr (J < N) {
            if constexpr (J != I) {
                fx += oct_from<I, J>(tx);
                fy += oct_from<I, J>(ty);
            }
            OctForceSum<N, I, J + 1>::run(tx, ty, fx, fy);
        }
    }
};

struct OctKin {   // one lane's agent during the kinematics of a step
    double cx, cy;       // current position: the new one once the agent's own stage has run (quirk Q7)
    double c1, s1;       // cos / sin of the new heading
    double xf, yf;       // the move with zero repulsion, (x + v*cos) + 0.0, wall rule applied: what most stages commit
    bool hitf;
    bool hit;            // wall flag of the position in (cx, cy) once the own stage has run
};

// (qx, qy) = (nx / den, ny / den), each quotient the correctly rounded IEEE one -- bit for bit what `/` gives -- with ONE reciprocal
// for both (flight_env_easy.py:299-300 divides the two components of a repulsion term by the same squared distance).  The
// compiler expands an fp64 division into v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal, the quotient and its
// fused residual correction (v_div_fmas) and v_div_fixup: ~16 instructions, of which the reciprocal part depends on the
// denominator alone.  For operands whose exponents are far from the ends of the range (here: squared distances below 9, terms
// below 3) v_div_scale scales nothing and v_div_fixup changes nothing, so the sequence below IS that expansion with the
// reciprocal shared; anything else (never seen: a squared distance below 1e-30) takes the plain divisions.
#ifndef CS_SHARED_RCP_DIV
#define CS_SHARED_RCP_DIV 1
#endif
__device__ __forceinline__ void div2_same_denominator(double nx, double ny, double den, double &qx, double &qy) {
#if CS_SHARED_RCP_DIV
    // the guard: den within 2^-100 .. 2^100, each numerator zero or within that range (NaN and infinities fail the <=).  The lower
    // bounds of the numerators are tested on their binary exponents (v_frexp_exp_i32_f64 gives 0 for a zero, so a zero passes): eight
    // instructions where the six range comparisons of round 4 took eighteen
    const double LO = 0x1p-100, HI = 0x1p100;
    const int e_lo = min(__builtin_amdgcn_frexp_exp(nx), __builtin_amdgcn_frexp_exp(ny));
    const bool plain = (den >= LO) & (den <= HI) & (fabs(nx) <= HI) & (fabs(ny) <= HI) & (e_lo >= -99);
    if (__builtin_expect(__ballot(!plain) == 0ull, 1)) {
        double r = __builtin_amdgcn_rcp(den);
        r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
        const double mx = nx * r, my = ny * r;
        qx = __builtin_fma(__builtin_fma(-den, mx, nx), r, mx);
        qy = __builtin_fma(__builtin_fma(-den, my, ny), r, my);
        return;
    }
#endif
    qx = nx / den;
    qy = ny / den;
}

// Stage I of the reference's sequential loop over agents (flight_env_easy.py:260-290, quirk Q7), for all 8 envs of the
// wavefront at once: every OTHER agent J tests itself against agent I's pre-move position -- its own position being the
// already-moved one if J < I -- and, if it is within force_dist, computes its term of I's repulsion (:293-301); the terms
// meet in lane I (ordered DPP sum); lane I moves its agent, applies the wall rule and becomes "already moved" for the
// later stages.  The two fp64 divisions run only if SOME env of the wavefront has such a neighbour in this stage.
template <int N, int I, bool SHARED_DIV, int LG = OG>
struct OctStage {
    static __device__ __forceinline__ void run(const DevParams &p, const double2 (&pre)[N], int t, bool act_lane,
                                               unsigned long long act_mask, OctKin &k) {
        if constexpr (I < N) {
            const double2 pi = pre[I];   // agent I's position BEFORE its move (read from the team's LDS row ahead of the trig evaluation)
            const double xi = pi.x, yi = pi.y;
            const double dx = k.cx - xi, dy = k.cy - yi;
            const double d2 = dx * dx + dy * dy;
            const bool c_lt = d2 < p.force_d2, c_nx = k.cx != xi, c_ny = k.cy != yi;
            const bool inr = act_lane & (t != I) & c_lt & (c_nx | c_ny);
            // "some lane is in range", from the three comparisons' own lane masks (a ballot of a bare comparison IS its result register;
            // a ballot of the combined predicate costs a select and a compare to rebuild that mask) and the step's mask of agent lanes
            const unsigned long long not_i = ~OctLay<LG>::lanes_t(I);
            if (__ballot(c_lt) & (__ballot(c_nx) | __ballot(c_ny)) & act_mask & not_i) {   // wave-uniform
                // x_a - x = -(x - x_a) exactly (a difference and its mirror round alike; a zero difference comes out as -0.0 here where
                // the reference has +0.0: its square is +0.0 all the same, and its term, -0.0, leaves every sum it is added to as it
                // was -- the sums start from +0.0).  So the squared distance of the term IS the one the range test computed, and the
                // mirrored differences cost a sign bit in the multiplications instead of two subtractions, two products and a sum.
                const double ex = -dx, ey = -dy;
                const double den = d2;
                double qx, qy;   // force_k*(x-x_a)/den: product first, then the division
                if constexpr (SHARED_DIV) {
                    div2_same_denominator(p.force_k * ex, p.force_k * ey, inr ? den : 1.0, qx, qy);
                } else {
                    qx = p.force_k * ex / den;
                    qy = p.force_k * ey / den;
                }
                const double tx = inr ? qx : 0.0;
                const double ty = inr ? qy : 0.0;
                double fx = 0.0, fy = 0.0;
                OctForceSum<N, I>::run(tx, ty, fx, fy);
                const double x = (k.cx + p.velocity * k.c1) + fx;   // lane I: (x + v*cos) + f_x on its pre-move position
                const double y = (k.cy + p.velocity * k.s1) + fy;
                const bool h = (x < 0.0) | (x > p.L) | (y < 0.0) | (y > p.L);    // flight_env_easy.py:278
                if (t == I) {
                    k.cx = h ? fmin(fmax(x, 0.0), p.L) : x;
                    k.cy = h ? fmin(fmax(y, 0.0), p.L) : y;
                    k.hit = h;
                }
            } else if (t == I) {   // no neighbour in range anywhere: f = 0, the move is (x + v*cos) + 0.0
                k.cx = k.xf;
                k.cy = k.yf;
                k.hit = k.hitf;
            }
            OctStage<N, I + 1, SHARED_DIV, LG>::run(p, pre, t, act_lane, act_mask, k);
        }
    }
};

// Kinematics of one step for the octet's env (flight_env_easy.py:255-301); `act` = this lane's agent's action.
// Returns the octet's out_flag bits.  Lanes t >= N hold no agent and take no part in any decision.
// SHARED_DIV: the two components of a repulsion term share one reciprocal (div2_same_denominator: same quotients, ~19 instructions
// fewer per stage that runs).  Measured (tools/gpu_r4_f.sh, 3 agents): the one-wavefront octet kernel at 32768 envs +8 %, the pair
// kernel at 8192 / 16384 envs +2 %, but the c2 pair (4096 envs, K alone on its SIMD and bound by its dependent chain) -1.9 %: the
// range check in front of the shared sequence lengthens the chain.  So: the one-wavefront kernel only.
template <int N, bool SHARED_DIV = false, int LG = OG, int AP = OCT_PAD>
__device__ __forceinline__ unsigned oct_kinematics(const DevParams &p, const double *T, const double2 (*pos)[AP], int o, int t,
                                                   int sh8, bool stepping, int act, EnvO<N, LG> &e, int tl_step = -1) {
    static_assert(LG == OG || (N == LG), "the 5-lane packing is for teams of exactly 5");
    const double PI = 3.141592653589793, TWO_PI = 2.0 * 3.141592653589793, THREE_PI = 3.0 * 3.141592653589793;
    const double DYAW = 3.141592653589793 / 18.0;
    const bool upd = (t < N) & stepping;
    // the team's pre-move positions: every stage tests against one of them, and the LDS row does not change before the stages are
    // through -- all N reads are issued here, ahead of the trig evaluation, instead of one exposed LDS round trip per stage
    double2 pre[N];
#pragma unroll
    for (int I = 0; I < N; I++) pre[I] = pos[o][I];
    double yaw = e.yaw;
    yaw = act == 1 ? yaw + DYAW : (act == 2 ? yaw + -DYAW : yaw);  // dyaw = [0, pi/18, -pi/18][act]
    yaw = yaw > TWO_PI ? yaw - TWO_PI : (yaw < 0.0 ? yaw + TWO_PI : yaw);
    const double yw = yaw, yr = (yaw <= PI) ? PI - yaw : THREE_PI - yaw;
    double s1, c1, s2, c2;
    KIN_STAMP(3);
#ifndef CS_OCT_TRIG_SPLIT
#define CS_OCT_TRIG_SPLIT 1   /* teams of up to 4: the wall reflection's sin / cos come from the idle lane four places up */
#endif
    if constexpr (CS_OCT_TRIG_SPLIT && N <= 4) {
        // Lanes 4..7 of an octet hold no agent.  Lane t + 4 evaluates agent t's REFLECTED heading while lane t evaluates the new
        // one: one correctly rounded evaluation per lane instead of two interleaved ones -- the chain is as long, but a lone
        // wavefront is bound by instruction issue (one per ~4.5 cycles), and the pair is ~100 instructions (same values: the
        // pair IS two single evaluations).  Octets are aligned halves of the 16-lane DPP rows: row_shr:4 / row_shl:4 stay
        // inside the octet for the lanes that use the result.
        const double yr_up = dpp_f64<0x114>(yr);          // lane L receives lane L - 4's reflected heading
        double sm, cm;
        trig_heading(T, t >= 4 ? yr_up : yw, sm, cm);
        s1 = sm;
        c1 = cm;
        s2 = dpp_f64<0x104>(sm);                          // lane L receives lane L + 4's result
        c2 = dpp_f64<0x104>(cm);
    } else {
        trig_heading_pair(T, yw, yr, s1, c1, s2, c2);
    }
    KIN_STAMP(4);
    // the move every agent makes unless a neighbour is within force_dist: (x + v*cos) + 0.0 -- the "+ 0.0" so that even
    // signed zeros agree with the reference's `x += force[0]`
    const double xt = (e.x + p.velocity * c1) + 0.0, yt = (e.y + p.velocity * s1) + 0.0;
    const bool hitf = (xt < 0.0) | (xt > p.L) | (yt < 0.0) | (yt > p.L);    // flight_env_easy.py:278
    OctKin k{e.x, e.y, c1, s1, hitf ? fmin(fmax(xt, 0.0), p.L) : xt, hitf ? fmin(fmax(yt, 0.0), p.L) : yt, hitf, false};
    // Would the reference find ANY neighbour within force_dist in this step, in any env of the wavefront?  Every ordered
    // pair (I, this lane's agent) is tested the way stage I would test it if no force had been applied before it: this
    // agent's position is its zero-repulsion move if it precedes I (quirk Q7), else its old one.  If no pair is in range
    // the reference's loop adds f = 0 everywhere and every agent's move IS the zero-repulsion one -- one ballot instead of
    // one per stage (3 agents: ~3 wavefront-steps in 4); otherwise the stages run, exactly.
    // (measured, pair kernel: 3 agents -2 % per step, 5 agents +5 %: with 40 agents per wavefront some pair is nearly always
    // in range and the pre-test is pure overhead -- so only small teams take it)
#ifndef CS_OCT_FASTPATH_MAX_N
#define CS_OCT_FASTPATH_MAX_N 3
#endif
    constexpr bool FASTPATH = N <= CS_OCT_FASTPATH_MAX_N;
    bool any_pair = !FASTPATH;
#pragma unroll
    for (int I = 0; I < (FASTPATH ? N : 0); I++) {
        const double2 pi = pre[I];
        const double qx = t < I ? k.xf : e.x, qy = t < I ? k.yf : e.y;
        const double dx = qx - pi.x, dy = qy - pi.y;
        any_pair = any_pair | ((t != I) & (dx * dx + dy * dy < p.force_d2) & ((qx != pi.x) | (qy != pi.y)));
    }
    const unsigned long long upd_mask = __ballot(upd);
    if (FASTPATH ? __ballot(any_pair & upd) != 0ull : upd_mask != 0ull) {
        OctStage<N, 0, SHARED_DIV, LG>::run(p, pre, t, upd, upd_mask, k);
    } else {
        k.cx = k.xf;
        k.cy = k.yf;
        k.hit = k.hitf;
    }
    KIN_STAMP(5);
    e.x = upd ? k.cx : e.x;
    e.y = upd ? k.cy : e.y;
    e.yaw = upd ? (k.hit ? yr : yw) : e.yaw;
    e.cs = upd ? (k.hit ? c2 : c1) : e.cs;
    e.sn = upd ? (k.hit ? s2 : s1) : e.sn;
    return oct_slice<LG>(__ballot(k.hit & upd), sh8);
}

// Detection pass + reward (flight_env_easy.py:223-253) for the octet's env on the positions in sh.pos; draws from the hit
// tape, which the caller guarantees to cover a step's worst case.  Returns curr_reward.
// LAZY (k_rollout_od's step): the tape is NOT shifted by the step's draws.  `tcur` (< 32 on entry and on return) is the bit of
// tape[0] at which the env's cursor stands: the pass reads its slots from a 64-bit window taken at that bit (two v_alignbit), adds
// its draws to tcur and lets whole dwords fall out of the tape only when tcur passes 32 -- a wave-uniform test, true in a minority of
// steps, in front of the ten selects.  The shift of EVERY step it replaces was ten v_alignbit and ten selects per 32 possible draws
// (teams of 5: 33 VALU instructions per step).  tape_canon() restores the canonical form (cursor at bit 0 of tape[0]), which every
// other user of the tape expects.
__device__ __forceinline__ void tape_canon(unsigned (&t)[TAPE_DW], int &tcur) {
#pragma unroll
    for (int k = 0; k < TAPE_DW; k++) t[k] = __builtin_amdgcn_alignbit(k + 1 < TAPE_DW ? t[k + 1] : 0u, t[k], (unsigned)tcur);
    tcur = 0;
}
template <int N, int LG, int AP, bool LAZY>
__device__ __forceinline__ int oct_detect_impl(const DevParams &p, const double2 (*pos)[AP], int o, int t, int sh8, bool stepping,
                                               EnvO<N, LG> &e, unsigned (&tape)[TAPE_DW], int &tcur) {
    constexpr int MAXDW = (N * CS_MAX_TARGETS) / 32 < 1 ? 1 : (N * CS_MAX_TARGETS) / 32;   // draws of one pass, in dwords
    constexpr int TPL = OctLay<LG>::TPL;   // this lane's targets: t, t + LG, ...
    bool has[TPL], inr[N][TPL];
    unsigned below[TPL];
    unsigned long long hasm[TPL];
    int rank[N][TPL];
    int base = 0;
    // (a ballot of a bare comparison is the comparison's own result register; the lanes that hold a target of a stepping env are
    // the same for every agent: their mask is taken once and applied on the scalar side)
#pragma unroll
    for (int k = 0; k < TPL; k++) {
        has[k] = stepping & (t + LG * k < p.n_targets);
        below[k] = (1u << (t + LG * k)) - 1u;
        hasm[k] = __ballot(has[k]);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double2 a = pos[o][i];
        unsigned gm = 0u;
#pragma unroll
        for (int k = 0; k < TPL; k++) {
            const double dx = e.tx[k] - a.x, dy = e.ty[k] - a.y;
            const bool c = dx * dx + dy * dy <= p.view_r2;   // (t_x-x)**2 + (t_y-y)**2 <= view_range**2
            inr[i][k] = has[k] & c;
            gm |= oct_slice<LG>(__ballot(c) & hasm[k], sh8) << (LG * k);
        }
#pragma unroll
        for (int k = 0; k < TPL; k++) rank[i][k] = base + __popc(gm & below[k]);   // agent-major order of the reference's double loop
        base += __popc(gm);
    }
    // draw slot r = bit r of the tape: one 64-bit shift (teams of up to 4 never reach slot 64; up to 8: slot 127)
    unsigned w4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w4[k] = LAZY ? __builtin_amdgcn_alignbit(tape[k + 1], tape[k], (unsigned)tcur) : tape[k];
    const unsigned long long t64a = (unsigned long long)w4[0] | ((unsigned long long)w4[1] << 32);
    const unsigned long long t64b = (unsigned long long)w4[2] | ((unsigned long long)w4[3] << 32);
    auto slot = [&](int r) __attribute__((always_inline)) {
        if (N * CS_MAX_TARGETS <= 64) return (bool)((t64a >> r) & 1ull);
        return (bool)(((r >= 64 ? t64b : t64a) >> (r & 63)) & 1ull);
    };
    bool hit[TPL];
#pragma unroll
    for (int k = 0; k < TPL; k++) hit[k] = false;
    // teams of 5 and more can draw past slot 63 -- an env with more than 64 (agent, target) pairs in range in ONE step, which no
    // run has ever shown -- so the common case reads every slot from the first 64-bit window (no per-slot window select) and a
    // wave-uniform test sends the other one through the general form
    if (N * CS_MAX_TARGETS <= 64 || __builtin_expect(__ballot(base > 64) == 0ull, 1)) {
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int k = 0; k < TPL; k++) hit[k] = hit[k] | (inr[i][k] & (bool)((t64a >> rank[i][k]) & 1ull));
    } else {
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int k = 0; k < TPL; k++) hit[k] = hit[k] | (inr[i][k] & slot(rank[i][k]));
    }
    e.mt_pos = wrap624(e.mt_pos + 2 * base);
    e.words += (unsigned long long)(2 * base);
    e.ahead -= 2 * base;
    if constexpr (LAZY) {
        tcur += base;   // < 32 + N * CS_MAX_TARGETS: at most MAXDW + 1 whole dwords
#pragma unroll
        for (int r = 0; r <= MAXDW; r++) {
            const bool out = tcur >= 32;
            if (__ballot(out) == 0ull) break;   // wave-uniform
#pragma unroll
            for (int k = 0; k < TAPE_DW; k++) tape[k] = out ? (k + 1 < TAPE_DW ? tape[k + 1] : 0u) : tape[k];
            tcur -= out ? 32 : 0;
        }
    } else {
        tape_shift<MAXDW>(tape, base);
    }
    // flight_env_easy.py:238-247
    unsigned newly = 0u;
#pragma unroll
    for (int k = 0; k < TPL; k++) {
        const bool nw = hit[k] & !((e.found >> (t + LG * k)) & 1u);
        newly |= oct_slice<LG>(__ballot(nw), sh8) << (LG * k);
    }
    int r = 0;
    if (stepping) {
        const int cnt = __popc(newly);
        r = -1 + 10 * cnt;   // MOVE_COST, FIND_ONE_TGT
        e.found |= newly;
        e.newly = newly;
        e.target_find += cnt;
        if (cnt > 0 && e.target_find == p.n_targets && !(e.flags & FLAG_WIN)) {
            r += 100;  // FIND_ALL_TGT
            e.flags |= FLAG_WIN;
        }
        r -= __popc(((unsigned)e.flags >> 8) & 0xffu);  // OUT_PUNISH per agent with out_flag set
        e.curr_reward = r;
        e.flags |= FLAG_DIRTY;
    }
    return r;
}
template <int N, int LG = OG, int AP = OCT_PAD>
__device__ __forceinline__ int oct_detect(const DevParams &p, const double2 (*pos)[AP], int o, int t, int sh8, bool stepping,
                                          EnvO<N, LG> &e, unsigned (&tape)[TAPE_DW]) {
    int zero = 0;
    return oct_detect_impl<N, LG, AP, false>(p, pos, o, t, sh8, stepping, e, tape, zero);
}

// The wavefront tops up the MT19937 rows of those of its 8 envs that have fewer than `min_ahead` twisted words left or no
// matching tape -- whole wavefront on one row at a time, like group_wave_advance -- and hands the new tape to the env's
// octet by ballot.
// DRAIN: end with nothing of its own in flight (callers whose steady-state loop waits for loads, see drain_vmem).
template <int N, bool DRAIN = true, int LG = OG>
__device__ __forceinline__ void oct_wave_advance(const DevParams &p, int wave_b0, int nvalid, int lane, int min_ahead,
                                                 unsigned *rowbuf, EnvO<N, LG> &e, unsigned (&tape)[TAPE_DW], bool &tape_ok) {
    const int o = OctLay<LG>::valid(lane) ? OctLay<LG>::env(lane) : -1;
#pragma unroll 1
    for (int g = 0; g < OctLay<LG>::ENVS; g++) {
        const int pos = __shfl(e.mt_pos, OctLay<LG>::first_of(g)), a = __shfl(e.ahead, OctLay<LG>::first_of(g));
        const int ok = __shfl(tape_ok ? 1 : 0, OctLay<LG>::first_of(g));
        if (g >= nvalid || (ok && a >= min_ahead)) continue;   // wave-uniform
        unsigned *m = p.mt + (size_t)(wave_b0 + g) * MT_STRIDE;
        RowRegs rr;
        row_load(m, lane, rr);
        row_to_lds(rr, rowbuf, lane);
        row_twist_ahead(rowbuf, m, pos, a < 0 ? 0 : a, lane);
#pragma unroll
        for (int it = 0; it < TAPE_DW / 2; it++) {
            const unsigned long long bm = row_slot_hits(p, rowbuf, pos, it, lane);
            if (o == g) {
                tape[2 * it] = (unsigned)(bm & 0xffffffffull);
                tape[2 * it + 1] = (unsigned)(bm >> 32);
            }
        }
        if (o == g) {
            e.ahead = MT_N;
            tape_ok = true;
        }
    }
    if (DRAIN) drain_vmem();
}

// VEC: every wavefront of the launch is full and every step's block of get_state rows is 16-byte aligned (the host splits a
// batch into a VEC launch and a plain one for the last < 8 envs).  EMIT: obs and state are both written -- then every
// store of a step is unconditional, the number of stores between the action prefetch and its use is a compile-time constant
// and the wait for the actions never waits for a store (with the stores behind `if (io.obs)` the compiler has to assume
// the shortest path and waits for the first stores of the step to be acknowledged: +0.4 us per step).

// Second half of an ASYNCHRONOUS row refresh (octet pair kernel, D): the row of env `g` of the wavefront was requested a
// step ago straight into `rowbuf` (global_load_lds) and has arrived (the caller waited for it); it is twisted ahead of the
// env's cursor in LDS, the new words go back to the state blob, and the env's octet receives its new hit tape.  Same work
// as oct_wave_advance for one env, minus the wait for the row.
template <int N, int LG = OG>
__device__ __forceinline__ void oct_advance_finish(const DevParams &p, int wave_b0, int g, int lane, unsigned *rowbuf,
                                                   EnvO<N, LG> &e, unsigned (&tape)[TAPE_DW], bool &tape_ok) {
    const int o = OctLay<LG>::valid(lane) ? OctLay<LG>::env(lane) : -1;
    const int pos = __shfl(e.mt_pos, OctLay<LG>::first_of(g)), a = __shfl(e.ahead, OctLay<LG>::first_of(g));
    row_twist_ahead(rowbuf, p.mt + (size_t)(wave_b0 + g) * MT_STRIDE, pos, a < 0 ? 0 : a, lane);
#pragma unroll
    for (int it = 0; it < TAPE_DW / 2; it++) {
        const unsigned long long bm = row_slot_hits(p, rowbuf, pos, it, lane);
        if (o == g) {
            tape[2 * it] = (unsigned)(bm & 0xffffffffull);
            tape[2 * it + 1] = (unsigned)(bm >> 32);
        }
    }
    if (o == g) {
        e.ahead = MT_N;
        tape_ok = true;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Auto-reset of the octet kernels (flight_env_easy.py:79-182).  With random or trained policies an episode of the shipped
// configuration ends after ~50 steps, so a wavefront of 8 envs resets one of them every ~6 steps: not a rare path.  The
// first version ran the 16-lane reset_targets() on kernel parameters read through cold_params(): a generic pointer, so
// every field was a flat load followed by a full wait -- some 25 dependent memory round trips per reset, ~8000 cycles, three
// steps' worth.  Here
//  * the scalars come from the kernarg segment through a CONSTANT-address-space pointer: scalar loads, one wait for all;
//  * the target tables (a*cx, a*cy, 2*a*dx, 2*a*dy per target) sit in LDS since the prologue (`rtab`, 4 x 16 doubles);
//  * every attempt batch reads twisted words only: an env with fewer than 64 left is topped up BEFORE its batch (whole
//    wavefront on the row, as everywhere), so there is no twist-on-the-fly path, no write-back of stream words and no
//    spilled predicates of one; an env whose first 16 attempts did not yield enough accepted pairs (~1 %) simply stays
//    pending for another round, its partial placement in LDS;
//  * start poses are a table the host filled (DevParams::start_x / start_y), not N divisions.
// ---------------------------------------------------------------------------------------------------------
// Target placement for the envs in `need` (bit 8 o = env o of the wavefront resets): new targets into the octet's e.tx /
// e.ty, the state blob and (normalised, found = 0) the env's get_state row in `tile`; each env's stream cursor, word count,
// twisted-ahead count and hit tape advance by what the reference's sequential algorithm consumes.  One env per 16-lane
// group and round (lane = polar attempt; `slots`: four rows of 16 positions, the hand-over from group to octet, free between
// rounds); pre(w) may hand a group the four stream words of its FIRST batch (fetched ahead of time); before_tile() runs
// before the first write to `tile`.
template <int N, bool DRAIN, int LG = OG, class BeforeTile, class Pre>
__device__ __forceinline__ void oct_place_targets(const DevParams &cp, int wave_b0, int nvalid, int lane, bool live,
                                                  unsigned long long need, const double *rtab, double2 (*slots)[G], float *tile,
                                                  int W, unsigned *rowbuf, EnvO<N, LG> &e, unsigned (&tape)[TAPE_DW], bool &tape_ok,
                                                  BeforeTile before_tile, Pre pre) {
    using Lay = OctLay<LG>;
    const CS_AS4 DevParams *q = cold_params4();
    const int t16 = lane & (G - 1), gshift16 = lane & ~(G - 1), grp = lane >> 4, sh8 = Lay::first(lane), t = Lay::t(lane);
    const int n_targets = q->n_targets, target_mode = q->target_mode;
    const unsigned deter_mask = q->deter_mask;
    const double mid = q->mid, inv_half = q->inv_half, L = q->L;
    const CS_AS1 unsigned *mt = (const CS_AS1 unsigned *)q->mt;
    CS_AS1 double *tgt = (CS_AS1 double *)q->tgt;
    const unsigned tmask = n_targets >= 32 ? ~0u : ((1u << n_targets) - 1u);
    const unsigned fmask = target_mode == 0 ? ~deter_mask & tmask : 0u;   // jittered targets (flight_env_easy.py:95-113)
    const int need_total = __popc(fmask);
    const bool jit = (fmask >> t16) & 1u;
    const int my_rank = __popc(fmask & ((1u << t16) - 1u));   // which accepted attempt is this target's
    unsigned long long pend = need;
    int taken_env = 0;   // octet-uniform: accepted attempts of this env so far
    bool first = true;
    while (pend) {   // wave-uniform
        const bool pending = Lay::valid(lane) && ((pend >> sh8) & 1ull);
        if (__ballot(live && pending && e.ahead < 4 * G))
            oct_wave_advance<N, DRAIN, LG>(cp, wave_b0, nvalid, lane, 4 * G, rowbuf, e, tape, tape_ok);
        unsigned long long m = pend;
        for (int k = 0; k < grp; k++) m &= m ? m - 1 : 0ull;   // this group's env: the grp-th pending one
        const int src = m ? __ffsll((long long)m) - 1 : -1;
        const int sl = src >= 0 ? src : lane;
        const int g_pos = __shfl(e.mt_pos, sl), g_taken = __shfl(taken_env, sl);
        // an env back for another batch (~1 %) brings its partial placement along: target j sits in lane j % LG of its env, slot j / LG
        double px = 0.0, py = 0.0;
        if (__ballot(src >= 0 && g_taken > 0)) {   // wave-uniform
            const int from = src >= 0 ? src + (t16 % LG) : lane;
#pragma unroll
            for (int k = 0; k < Lay::TPL; k++) {
                const double xk = __shfl(e.tx[k], from), yk = __shfl(e.ty[k], from);
                if (t16 / LG == k) {
                    px = xk;
                    py = yk;
                }
            }
        }
        int words = 0, taken_new = 0;
        bool fin = false;
        if (src >= 0) {
            const int br = wave_b0 + Lay::env_of_first(src);
            double mx = rtab[t16], my = rtab[G + t16];   // a*cx, a*cy of target t16 (flight_env_easy.py:95-113)
            fin = true;
            if (target_mode != 0 || need_total > 0) {
                unsigned w[4];
                if (!(first && pre(w))) {
                    const CS_AS1 unsigned *row = mt + (size_t)br * MT_STRIDE + wrap624(g_pos + 4 * t16);
#pragma unroll
                    for (int k = 0; k < 4; k++) w[k] = row[k];   // (words 0..31 are mirrored behind the row: no wrap inside a lane's four)
                }
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = mt_temper(w[k]);
                // numpy random_sample: 53-bit double from two words
                const double u1 = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
                const double u2 = ((double)(w[2] >> 5) * 67108864.0 + (double)(w[3] >> 6)) / 9007199254740992.0;
                if (target_mode == 0) {
                    // np.random.randn is the legacy polar method: attempts (x1, x2) until 0 < r2 < 1; the pair's SECOND value
                    // f*x2 is returned first, f*x1 is cached for the next call -- the j-th accepted attempt serves the j-th
                    // jittered target (see reset_targets)
                    if (g_taken > 0) {
                        mx = px;
                        my = py;
                    }
                    const double x1 = 2.0 * u1 - 1.0, x2 = 2.0 * u2 - 1.0;
                    const double r2 = x1 * x1 + x2 * x2;
                    const bool accept = !(r2 >= 1.0 || r2 == 0.0);
                    const double f = sqrt(-2.0 * log(accept ? r2 : 0.5) / (accept ? r2 : 0.5));
                    const double g1 = f * x2, g2 = f * x1;
                    const unsigned amask = (unsigned)((__ballot(accept) >> gshift16) & 0xffffull);
                    const int have = __popc(amask);
                    const int want = need_total - g_taken;
                    const int k = my_rank - g_taken;   // my index within this batch's accepts
                    const int sel = kth_set_bit16(amask, (k >= 0 && k < 16) ? k : 0);
                    const double s1 = __shfl(g1, sel & 15, G), s2 = __shfl(g2, sel & 15, G);
                    if (jit && k >= 0 && k < have && k < want) {
                        mx += rtab[2 * G + t16] * (s1 - 0.5);  // dx*2*(randn-0.5)
                        my += rtab[3 * G + t16] * (s2 - 0.5);
                    }
                    // words consumed: up to and including the attempt that supplied the last needed pair, else the batch
                    const int last = have >= want ? kth_set_bit16(amask, want - 1) : 15;
                    words = 4 * (last + 1);
                    taken_new = g_taken + (have < want ? have : want);
                    fin = taken_new >= need_total;
                } else {   // x, y = map_size*np.random.rand() per target, flight_env_easy.py:122-127
                    mx = L * u1;
                    my = L * u2;
                    words = 4 * n_targets;
                }
            }
            slots[grp][t16] = make_double2(mx, my);
            if (fin) {
                typedef double v2d __attribute__((ext_vector_type(2)));
                reinterpret_cast<CS_AS1 v2d *>(tgt + (size_t)br * G * 2)[t16] = v2d{mx, my};
                before_tile();   // (the octet pair's emitting wavefront may still be reading the old rows)
                if (t16 < n_targets) {
                    float *rs = tile + Lay::env_of_first(src) * W + 4 * N + 3 * t16;
                    rs[0] = (float)((mx - mid) * inv_half);   // norm_target
                    rs[1] = (float)((my - mid) * inv_half);
                    rs[2] = 0.0f;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the k-th pending env was served by group k: its octet takes the placement (final or partial) and the stream position
        const int rank = __popcll(pend & ((1ull << sh8) - 1ull));
        const bool got = pending && rank < 4;
        const int leader = got ? G * rank : lane;
        const int r_words = __shfl(words, leader), r_taken = __shfl(taken_new, leader), r_fin = __shfl(fin ? 1 : 0, leader);
        if (got) {
#pragma unroll
            for (int k = 0; k < Lay::TPL; k++) {
                const double2 tk = slots[rank][(t + LG * k) & (G - 1)];   // (t + LG k < 16 for every lane that is `pending`)
                e.tx[k] = tk.x;
                e.ty[k] = tk.y;
            }
            tape_shift<1>(tape, r_words >> 1);   // (<= 32 draw slots leave the tape)
            e.mt_pos = wrap624(e.mt_pos + r_words);
            e.words += (unsigned long long)r_words;
            e.ahead -= r_words;
            taken_env = r_taken;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the slots are free again (next round; a top-up may reuse their memory)
        pend &= ~__ballot(got && r_fin != 0 && t == 0);
        first = false;
    }
}

template <int N, bool VEC, bool EMIT>
__global__ __launch_bounds__(OCT_BLOCK, CS_OCT_WAVES) void k_rollout_oct(DevParams p, StepIO io) {
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    __shared__ OctShared shared[OCT_BLOCK / 64];
    __shared__ double rtab[4 * G];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int o = lane >> 3, sh8 = lane & ~(OG - 1);
    int t = lane & (OG - 1);   // (not const: made opaque once per step, see the loop)
    const int wave_b0 = io.env0 + (blockIdx.x * (OCT_BLOCK / 64) + wave) * OCT_ENVS;
    const int b_end = io.env0 + io.env_n;
    const int b = wave_b0 + o;
    const bool live = VEC || b < b_end;   // a VEC launch has only full wavefronts (the early return below takes the empty ones)
    const int nvalid = b_end - wave_b0 < OCT_ENVS ? b_end - wave_b0 : OCT_ENVS;   // <= 0: a wavefront without envs
    const int W = 4 * N + 3 * p.n_targets;
    bool ag = t < N;
    const bool auto_reset = io.flags & CS_AUTO_RESET, freeze = io.flags & CS_FREEZE_DONE;
    OctShared &sh = shared[wave];
    EnvO<N> e;
    // ---- everything the first step waits for is requested before the barrier that publishes the trig table
    const size_t bl = live ? (size_t)b : (size_t)io.env0;
    {
        const int4 *h4 = reinterpret_cast<const int4 *>(p.hdr + bl * CS_H_WORDS);
        const int4 h0 = h4[0], h1 = h4[1], h2 = h4[2];
        e.found = (unsigned)h0.x;
        e.newly = (unsigned)h0.y;
        e.target_find = h0.z;
        e.flags = h0.w;
        e.time_step = h1.x;
        e.total_reward = h1.y;
        e.mt_pos = h1.z;
        e.episodes = h1.w;
        e.words = (unsigned long long)(unsigned)h2.x | ((unsigned long long)(unsigned)h2.y << 32);
        e.curr_reward = h2.z;
        e.newly_reset = (unsigned)h2.w;
        e.ahead = p.ahead[bl];
        const double4 a = reinterpret_cast<const double4 *>(p.agent + bl * CS_MAX_AGENTS * 4)[t];
        e.x = a.x;
        e.y = a.y;
        e.yaw = a.z;
        const double2 *t2 = reinterpret_cast<const double2 *>(p.tgt + bl * G * 2);
        const double2 ta = t2[t], tb = t2[t + OG];
        e.tx[0] = ta.x;
        e.ty[0] = ta.y;
        e.tx[1] = tb.x;
        e.ty[1] = tb.y;
    }
    const TapeRaw traw = tape_fetch(p, (int)bl);
    const int aidx = ag ? t : N - 1;   // lanes without an agent repeat the last agent's (valid) address
    const int astride = (io.flags & CS_ACTIONS_I64) ? 2 : 1;
    const int *ap = reinterpret_cast<const int *>(io.actions) + (bl * N + aidx) * astride;   // this lane's action of step 0
    const size_t astep = (size_t)p.B * N * astride;
    int act = ap[0];
    if (io.T > 1) ap += astep;
    int act_next = ap[0];   // one step ahead of its use
    if (io.T > 2) ap += astep;   // -> step 2 (or the last step: short launches re-read it, the value is never used)
    if (wave == 0) load_reset_tab(rtab, lane);
    load_trig_to_lds(T);
    if (nvalid <= 0) return;   // wave-uniform
    if (!live) {   // a lane without an env never steps, resets or asks for a top-up
        e.target_find = 0;
        e.time_step = 0;
        e.ahead = 1 << 20;
    }
    unsigned tape[TAPE_DW];
    bool tape_ok = tape_finish(p, traw, e, tape) || !live;
    int tcur = 0;   // the step's detection pass leaves the tape unshifted (oct_detect_impl, LAZY): canonical again wherever else it is used
    trig_heading(T, e.yaw, e.sn, e.cs);   // what a frozen env keeps emitting
    // ---- persistent rows: agents' floats, targets' normalised coordinates and found flags (get_state, :190-216)
    float *row = sh.tile + o * W;
    auto put_agent = [&]() __attribute__((always_inline)) {
        if (ag) {
            row[4 * t + 0] = (float)((e.x - p.mid) * p.inv_half);
            row[4 * t + 1] = (float)((e.y - p.mid) * p.inv_half);
            row[4 * t + 2] = (float)e.cs;
            row[4 * t + 3] = (float)e.sn;
        }
    };
    auto put_found = [&]() __attribute__((always_inline)) {
        if (t < p.n_targets) row[4 * N + 3 * t + 2] = ((e.found >> t) & 1u) ? 1.0f : 0.0f;
        if (t + OG < p.n_targets) row[4 * N + 3 * (t + OG) + 2] = ((e.found >> (t + OG)) & 1u) ? 1.0f : 0.0f;
    };
    if (t < p.n_targets) {
        row[4 * N + 3 * t + 0] = (float)((e.tx[0] - p.mid) * p.inv_half);   // norm_target
        row[4 * N + 3 * t + 1] = (float)((e.ty[0] - p.mid) * p.inv_half);
    }
    if (t + OG < p.n_targets) {
        row[4 * N + 3 * (t + OG) + 0] = (float)((e.tx[1] - p.mid) * p.inv_half);
        row[4 * N + 3 * (t + OG) + 1] = (float)((e.ty[1] - p.mid) * p.inv_half);
    }
    put_found();
    put_agent();
    sh.pos[o][t] = make_double2(e.x, e.y);
    constexpr int LOW = 2 * N * CS_MAX_TARGETS;   // words one step can consume
    oct_wave_advance<N>(p, wave_b0, nvalid, lane, io.min_ahead > LOW ? io.min_ahead : LOW, sh.rowbuf, e, tape, tape_ok);
    // ---- write-out plan (loop invariant)
    const int rows_valid = nvalid;
    constexpr int W_MAX = 4 * N + 3 * CS_MAX_TARGETS;
    constexpr int Q = (OCT_ENVS * W_MAX / 4 + 63) / 64;   // float4 chunks per lane of the largest tile
    const int ol = lane < rows_valid * N ? lane : rows_valid * N - 1;
    const int orow = ol / N, oag = ol - orow * N;
    const int obs_lds = orow * W + 4 * oag;
    const int rtw = (lane & 7) < rows_valid ? (lane & 7) : rows_valid - 1;
    const int t16 = lane & (G - 1), gshift16 = lane & ~(G - 1), grp = lane >> 4;
    float *p_rew = io.reward + wave_b0 + rtw;
    uint8_t *p_term = io.terminated + wave_b0 + rtw, *p_win = io.win + wave_b0 + rtw;
    v4f *p_obs = reinterpret_cast<v4f *>(io.obs + (size_t)wave_b0 * N * 4) + ol;
    v4f *p_st = reinterpret_cast<v4f *>(io.state + (size_t)wave_b0 * W);   // VEC: the wavefront's block of rows, as float4 chunks
    int chunk[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) chunk[q] = lane + 64 * q < OCT_ENVS * W / 4 - 1 ? lane + 64 * q : OCT_ENVS * W / 4 - 1;

    for (int s = 0; s < io.T; s++) {
        OCT_STAMP(0);
        REAL_STAMP(8);
        // Lane predicates that never change (t < N, t != I, t < n_targets ...) are cheaper to recompute -- one v_cmp -- than to
        // keep: hoisted out of the loop each is an SGPR pair, ~30 SGPRs in all, which the scalar file does not have left
        // (they came back as v_readlane pairs at every use).  Making t opaque once per step keeps the compares in the loop.
        asm volatile("" : "+v"(t));
        ag = t < N;
        // the actions of step s + 2, requested a whole step before their use and BEFORE this step's stores: the wait for them
        // never waits for a store (one in-order counter for loads and stores)
        const int act_after = ap[0];
        if (s + 3 < io.T) ap += astep;
        bool done = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
        e.flags &= ~(FLAG_DIRTY | FLAG_RESET_PASS);
        // ---- auto-reset (flight_env_easy.py:79-182).  Target placement -- the polar-gaussian attempts, 16 at a time --
        //      runs on the 16-lane code above (reset_targets), one resetting env per 16-lane group and round; the new
        //      targets come back through LDS, the stream position by shuffle; the agents' start poses and the reset-time
        //      detection pass (quirk Q3) are the octet's own.
        const unsigned long long need = __ballot(live && done && auto_reset && t == 0);   // bit 8 o'
        if (__builtin_expect(need != 0ull, 0)) {   // one wave-step in ~6 with the shipped configuration (see oct_place_targets)
            const DevParams &cp = cold_params();
            const bool mine = (need >> sh8) & 1ull;
            const StartTab<N> st = start_tab<N>();
            tape_canon(tape, tcur);
            oct_place_targets<N, true>(cp, wave_b0, nvalid, lane, live, need, rtab, sh.tgt, sh.tile, W, sh.rowbuf, e, tape, tape_ok,
                                       []() {}, [](unsigned (&)[4]) { return false; });
            if (mine) {
                e.episodes += 1;
                e.found = 0;
                e.newly = 0;
                e.target_find = 0;
                e.time_step = 0;
                e.total_reward = 0;
                e.flags = 0;
                start_pick<N>(st, ag ? t : 0, e.x, e.y);
                e.yaw = st.yaw;
                trig_heading(T, e.yaw, e.sn, e.cs);
                sh.pos[o][t] = make_double2(e.x, e.y);
            }
            drain_vmem();
            // reset-time detection pass (quirk Q3: its reward is discarded) of the envs just reset, from the tape -- topped up
            // first where the attempts ran past the twisted words
            if (__ballot(live && e.ahead < LOW)) oct_wave_advance<N>(cp, wave_b0, nvalid, lane, LOW, sh.rowbuf, e, tape, tape_ok);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (agent_mode 0 with the shipped target file never has a target within view of a start pose: the pass -- whose
            // reward is discarded anyway -- is then three assignments; the test costs a third of the pass it usually saves)
            bool near = false;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const double sx = st.x[i], sy = st.y[i];
                const double ax0 = e.tx[0] - sx, ay0 = e.ty[0] - sy, ax1 = e.tx[1] - sx, ay1 = e.ty[1] - sy;
                near = near | ((t < cp.n_targets) & (ax0 * ax0 + ay0 * ay0 <= cp.view_r2)) |
                       ((t + OG < cp.n_targets) & (ax1 * ax1 + ay1 * ay1 <= cp.view_r2));
            }
            if (__ballot(mine && near)) {
                oct_detect<N>(p, sh.pos, o, t, sh8, mine, e, tape);
                put_found();
            } else if (mine) {   // what the pass does when no pair is in range: no draw, reward -1
                e.newly = 0u;
                e.curr_reward = -1;
                e.flags |= FLAG_DIRTY;
            }
            done = done && !mine;
            if (__ballot(live && e.ahead < LOW)) oct_wave_advance<N>(cp, wave_b0, nvalid, lane, LOW, sh.rowbuf, e, tape, tape_ok);
        }
        const bool stepping = live && !(done && freeze);
        OCT_STAMP(1);
        // ---- kinematics -> positions, obs floats, out flags
        const unsigned out = oct_kinematics<N, CS_SHARED_RCP_DIV != 0>(p, T, sh.pos, o, t, sh8, stepping, act, e);
        OCT_STAMP(2);
        if (stepping) e.flags = (e.flags & ~0xff00) | (int)(out << 8);
        sh.pos[o][t] = make_double2(e.x, e.y);
        put_agent();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- detection, reward, termination
        OCT_STAMP(3);
        const int reward = oct_detect_impl<N, OG, OCT_PAD, true>(p, sh.pos, o, t, sh8, stepping, e, tape, tcur);
        OCT_STAMP(4);
        bool term = true;
        if (stepping) {
            e.total_reward += reward;
            e.time_step += 1;
            term = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
        }
        if (__ballot(stepping && e.newly != 0u)) put_found();   // wave-uniform: some env found a target in this step
        if (t == 0) {
            sh.reward[o] = (float)reward;
            sh.term[o] = term ? 1 : 0;
            sh.win[o] = (e.flags & FLAG_WIN) ? 1 : 0;
        }
        OCT_STAMP(5);
        // ---- a row that is about to run out of twisted words is topped up in place (about one wave-step in 10)
        if (__builtin_expect(__ballot(live && e.ahead < LOW) != 0ull, 0)) {
            tape_canon(tape, tcur);
            oct_wave_advance<N>(cold_params(), wave_b0, nvalid, lane, LOW, sh.rowbuf, e, tape, tape_ok);
        }
        act = act_next;
        act_next = act_after;
        OCT_STAMP(6);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- this step's outputs: the wavefront's 8 envs are contiguous in every output tensor; every lane keeps running
        //      pointers (one 64-bit add per tensor and step instead of rebuilding them from kernel arguments)
        //      (all LDS reads first, then the stores: one LDS round trip instead of one per store)
        const float o_rew = sh.reward[rtw];   // duplicates write the same value
        const int o_term = sh.term[rtw], o_win = sh.win[rtw];
        v4f o_obs = {0.f, 0.f, 0.f, 0.f}, o_st[Q];
        if (EMIT || io.obs) {
            const float *src = sh.tile + obs_lds;
            o_obs = v4f{src[0], src[1], src[2], src[3]};
        }
        if (VEC && (EMIT || io.state)) {
            const float4 *src4 = reinterpret_cast<const float4 *>(sh.tile);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const float4 v = src4[chunk[q]];
                o_st[q] = v4f{v.x, v.y, v.z, v.w};
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        *p_rew = o_rew;
        *p_term = (uint8_t)o_term;
        *p_win = (uint8_t)o_win;
        p_rew += p.B;
        p_term += p.B;
        p_win += p.B;
        if (EMIT || io.obs) {   // one float4 per (env, agent)
            __builtin_nontemporal_store(o_obs, p_obs);
            p_obs += (size_t)p.B * N;
        }
        if (EMIT || io.state) {
            if (VEC) {   // full wavefront, 16-byte aligned block of rows: float4 chunks; surplus lanes repeat the last chunk
#pragma unroll
                for (int q = 0; q < Q; q++) __builtin_nontemporal_store(o_st[q], p_st + chunk[q]);
                p_st += (size_t)p.B * W / 4;
            } else {
                float *dst = io.state + ((size_t)s * p.B + wave_b0) * W;
                for (int k = lane; k < rows_valid * W; k += 64) dst[k] = sh.tile[k];
            }
        }
        OCT_STAMP(7);
    }
    tape_canon(tape, tcur);
    if (live) {
        const DevParams &cp = cold_params();
        if (t == 0) {
            int4 *h4 = reinterpret_cast<int4 *>(cp.hdr + (size_t)b * CS_H_WORDS);
            h4[0] = make_int4((int)e.found, (int)e.newly, e.target_find, e.flags);
            h4[1] = make_int4(e.time_step, e.total_reward, e.mt_pos, e.episodes);
            h4[2] = make_int4((int)(unsigned)(e.words & 0xffffffffull), (int)(unsigned)(e.words >> 32), e.curr_reward,
                              (int)e.newly_reset);
            cp.ahead[b] = e.ahead;
        }
        if (ag) reinterpret_cast<double4 *>(cp.agent + (size_t)b * CS_MAX_AGENTS * 4)[t] = make_double4(e.x, e.y, e.yaw, 0.0);
        if (tape_ok) {
            U4 *tp = reinterpret_cast<U4 *>(cp.tape + (size_t)b * TAPE_STRIDE);
            if (t == 0) tp[0] = U4{tape[0], tape[1], tape[2], tape[3]};
            if (t == 1) tp[1] = U4{tape[4], tape[5], tape[6], tape[7]};
            if (t == 2) tp[2] = U4{tape[8], tape[9], (unsigned)(e.words & 0xffffffffull), (unsigned)(e.words >> 32)};
            if (t == 3) tp[3] = U4{(unsigned)(cp.detect_K & 0xffffffffull), (unsigned)(cp.detect_K >> 32), 0u, 0u};
        }
    }
}

// =========================================================================================================
// Octet pair (flight_easy): the octet layout split by ROLE -- per 8 envs a kinematics wavefront K, a detection
// wavefront D and (up to 8192 envs) an emitting wavefront E, one such team per workgroup, no barrier in the loops.
//
// In the octet kernel one wavefront walks the whole dependent chain of a step -- kinematics (~1800 cycles for 3 agents),
// then detection + reward + rows (~1500) -- and at the batch sizes where every SIMD holds at most one or two wavefronts
// nothing fills its stalls.  As in k_rollout_duo, the kinematics of step s + 1 need nothing from the detection pass of
// step s (the actions are an open-loop table; the only coupling is a termination K cannot predict from the step counter:
// an env finding its last target), so K runs AHEAD and leaves each step's positions in a ring of OD_RING LDS slots; D
// consumes them.  A batch gets twice the wavefronts -- 4096 envs fill all 1024 SIMDs (the octet kernel: half of them) -- and
// the two halves of a step overlap.  The pair synchronises through two LDS counters, not through workgroup barriers: K
// may produce step j once D has finished step j - OD_RING, D may consume step s once K has produced it.  With a barrier per
// step (round 2's pair kernel, and the first version of this one) every rare event on either side -- a reset, an MT19937
// row top-up: 1.5-3 us each -- stops BOTH wavefronts, and each step pays the barrier's own latency on top of max(K, D);
// with counters K simply runs up to OD_RING - 1 steps ahead, D (the longer half) never waits, and its events cost only D's
// own time.  Roles:
//   K  lane t owns agent t: trig, the repulsion stages (OctStage), wall rule; keeps the team's current positions in its
//      own LDS array (kpos), publishes (x, y, yaw, cos, sin, out flags) per step; predicts resets / freezes from the step
//      counter.  When D reports a termination K could not predict (a win at step s), K restores that env from ring slot s
//      and REDOES every step it has already produced past s, for that env only (D holds slot s and waits meanwhile).
//   D  lane t owns targets t, t + 8, the env's header and its hit tape: sensor tests on the ring's positions, draws,
//      reward, termination; resets (oct_place_targets, reset-time pass) and row top-ups; without E also the persistent
//      get_state rows and every output store.
//   E  (template flag E3) owns the get_state tile and writes reward, terminated, win, obs, state of each step from K's
//      ring slot and the record D leaves per step (OdOut): a quarter of D's plain step, which D -- the role that also carries
//      every event -- no longer has to do.
// Arithmetic per env is the octet kernel's (same functions), so results are bit-identical.
// =========================================================================================================
#ifndef CS_OD_WAVES
#define CS_OD_WAVES 4
#endif
#ifndef CS_OD_RING
#define CS_OD_RING 4
#endif
#ifndef CS_OD_RING_E3
#define CS_OD_RING_E3 8   /* ring depth of the three-wavefront variant (measured at c2: 2 -> 2.74e9, 4 -> 3.07e9, 8 -> 3.15e9) */
#endif
// The 16-lanes-per-env ROLLOUT kernels of rounds 1-2 (k_rollout "solo", k_rollout_duo) are selected by no dispatch row any more
// (DESIGN.md section 4); they stay in the source behind this switch for cross-kernel comparisons (CS_KERNEL_SOLO / CS_KERNEL_DUO then
// work again) and cost 16 kernel instantiations of compile time.  Without them CS_KERNEL_GROUP rollouts are T launches of k_step.
#ifndef CS_LEGACY_KERNELS
#define CS_LEGACY_KERNELS 0
#endif
constexpr int OD_BLOCK = 128;
// Teams of 5: the 5-lanes-per-env packing of the pair kernels (OctLay<5>: twelve envs per workgroup).  1: cs_rollout's pair kernels take
// it for teams of exactly 5; 0 (default): the octet layout only, and k_rollout_od5 is not even instantiated.
// Round 5's experiment for the 5-agent configurations, bit-identical to the step kernel (tests/test_gpu_jitter.py keeps a build of it
// in the suite) and SLOWER where it was meant to pay (one box, us per step, packing / octet): pair kernel at 8192 envs 2.50 (8184 envs:
// no tail) / 2.26, at 16384 envs 3.58 / 3.40; three-wavefront variant at 8192 envs 2.28 / 1.90; first ahead at 32768 envs (7.42 / 8.27),
// which the lane kernels serve.  Why: (1) it executes 14.5 % fewer VALU instructions, not the third the idle lanes suggested -- what a
// lane does per TARGET (three per lane instead of two) is most of the step and does not shrink, only the per-agent kinematics do
// (SQ_INSTS_VALU at 32760 envs: 2.74e8 against 3.20e8 per 100 steps); (2) 8192 envs are 682.7 workgroups of twelve on 256 CUs: two CUs
// in three run three workgroups, the rest two, and the step takes what the fuller ones take -- as many wavefronts per SIMD as the
// octet's four workgroups per CU, each wavefront a quarter longer.  DESIGN.md section 9.
#ifndef CS_OD_PENT
#define CS_OD_PENT 0
#endif
// Teams from this size on divide the two components of a repulsion term with ONE reciprocal in K (div2_same_denominator: the same
// quotients bit for bit).  Small teams keep the plain divisions: K is alone on its SIMD there and the range check in front of the
// shared sequence lengthens its chain (c2: -1.9 %, round 4); large teams run four wavefronts per SIMD at the VALU issue limit,
// where only the instruction count matters.
#ifndef CS_OD_SHARED_DIV_FROM_N
#define CS_OD_SHARED_DIV_FROM_N 99
#endif
// steps K may be ahead of D (power of two).  The pair variant serves up to 16384 envs with eight workgroups per CU: 20 KB of LDS each,
// four slots.  The three-wavefront variant stops at 8192 envs = four workgroups per CU, so its ring can be eight deep (30 KB + E's row buffer):
// K absorbs more of D's events before it has to wait for a slot.
constexpr int od_ring(bool e3) { return e3 ? CS_OD_RING_E3 : CS_OD_RING; }
static_assert((od_ring(false) & (od_ring(false) - 1)) == 0 && od_ring(false) >= 2, "ring depth");
static_assert((od_ring(true) & (od_ring(true) - 1)) == 0 && od_ring(true) >= 2, "ring depth");

template <int ENVS, int AP>
struct __attribute__((aligned(16))) OdRingT {   // what K hands to D for one step (ENVS envs per workgroup: 8, or 12 in the 5-lane packing)
    double2 pos[ENVS][AP];                      // AP columns per env: the agents + one of padding (bank spread)
    double yaw[ENVS][AP - 1];
    float2 cssn[ENVS][AP - 1];
    unsigned out[ENVS];
    unsigned pad[ENVS];
};
template <int RING, int ENVS = OCT_ENVS, int AP = OCT_PAD, int TW = TILE_W>
struct __attribute__((aligned(16))) OdSharedT {
    OdRingT<ENVS, AP> ring[RING];
    double2 kpos[ENVS][AP];        // K: the team's current positions (the "old" ones of its next step)
    double2 dpos[ENVS][AP];        // D: start poses for the reset-time detection pass
    float tile[ENVS * TW];
    float reward[ENVS];
    int term[ENVS], win[ENVS];
    // pair synchronisation (LDS words, written by one side, polled by the other; the LDS serves a workgroup's accesses in
    // order, so data written before a counter is visible to whoever has seen the counter)
    int k_steps;                             // K: steps produced so far (slot s is valid once k_steps > s)
    int d_steps;                             // D: steps finished so far (slot s may be overwritten once d_steps > s)
    int fix_req, fix_ack;                    // D -> K: "step fix_req - 1 ended an episode you could not predict" / K -> D: redone
    unsigned fix_mask;                       // ... for the envs in this mask (bit o)
    int e_steps;                             // E (three-wavefront variant): steps written out so far
    // (d_steps, fix_req and e_steps within a few dwords of each other: K reads its two words with ONE ds_read2_b32)
    union {   // never live together: a requested row is consumed at the top of a step, before any reset of that step
        unsigned rowbuf[MT_N + 16];          // one MT19937 row (+ the 16 words lanes 48..63 of the tenth dword column land on)
        double2 tgt[4][CS_MAX_TARGETS];      // D: reset hand-over (16-lane group -> octet), free between rounds
    };
    unsigned prebuf[4 * 64];                 // [q][lane]: the first attempt batch of the resets due at the next step
    double rtab[4 * G];                      // the reset's target tables (load_reset_tab)
};

// CS_OD_E_REFRESH (three-wavefront variant): E, which has most of a step to spare, does the MT19937 row refreshes instead of D.
// D posts (env, cursor, twisted words ahead) and goes on drawing from the env's old tape, which covers the words still ahead; E loads
// the row, twists it ahead of THAT cursor (the words it writes lie behind the cursor D reads from, in ring order), computes the
// 320-slot hit tape and posts it; D adopts it at a step boundary, shifted by the slots it consumed meanwhile.  One request at a time;
// anything that needs the row itself (a reset, an on-the-spot top-up) first waits for the outstanding one.
#ifndef CS_OD_E_REFRESH
#define CS_OD_E_REFRESH 1   /* measured: c2 3.17 -> 3.30e9 at 100 steps per launch, 1.75 -> 1.79e9 at 20; c5's 8192-env shard 3.77 -> 4.00e9 */
#endif
template <bool ON>
struct __attribute__((aligned(16))) OdRefreshT {
    int rf_req, rf_done;                     // D -> E: sequence number of the latest request / E -> D: ... of the latest one served
    int rf_env, rf_pos, rf_ahead;            // the request: octet, cursor, twisted words ahead of it
    int d_done;                              // D -> E: no further requests (E's exit condition)
    int pad[2];
    unsigned rf_tape[TAPE_DW + 2];           // E -> D: hit bits of the 312 slots from rf_pos
    unsigned erow[ON ? MT_N + 16 : 4];       // E's row buffer
};

// The pair's counters are plain LDS words written and polled with hand-placed ds instructions.  The LDS serves one
// wavefront's accesses in order, so slot data written before a counter is visible to whoever has seen the counter; nothing
// else is needed -- and anything else costs: a workgroup-scope release fence, and even a relaxed workgroup-scope atomic store,
// make the compiler wait for every GLOBAL operation in flight first (`s_waitcnt vmcnt(0)` before the ds_write: D's six output
// stores of the step, K's action prefetch), i.e. one memory round trip per step on both sides.
__device__ __forceinline__ unsigned lds_offset_of(const void *w) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)w;
}
__device__ __forceinline__ void lds_post(int *w, int v) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(lds_offset_of(w)), "v"(v) : "memory");
}
__device__ __forceinline__ int lds_peek(const int *w) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_offset_of(w)) : "memory");
    return v;
}
// two words OFF0 and OFF1 dwords behind `base` in one LDS round trip
template <int OFF0, int OFF1>
__device__ __forceinline__ int2 lds_peek2(const int *base) {
    static_assert(OFF0 >= 0 && OFF0 < 256 && OFF1 >= 0 && OFF1 < 256, "ds_read2_b32 offsets are 8-bit dword counts");
    typedef int v2i __attribute__((ext_vector_type(2)));
    v2i v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(v) : "v"(lds_offset_of(base)), "n"(OFF0), "n"(OFF1) : "memory");
    return make_int2(v.x, v.y);
}

// The same read in two halves: issued here, waited for (lds_peek2_wait) where the words are needed -- K reads its flow-control words for
// the NEXT loop head in the middle of a step, so the LDS round trip runs beside the step's publication instead of in front of the next
// step.  (The compiler does not know the asm is an LDS read; its own lgkmcnt waits can only become longer by one outstanding read it
// does not count, never shorter: LDS operations return in order.)
template <int OFF0, int OFF1>
__device__ __forceinline__ int2 lds_peek2_issue(const int *base) {
    static_assert(OFF0 >= 0 && OFF0 < 256 && OFF1 >= 0 && OFF1 < 256, "ds_read2_b32 offsets are 8-bit dword counts");
    typedef int v2i __attribute__((ext_vector_type(2)));
    v2i v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(v) : "v"(lds_offset_of(base)), "n"(OFF0), "n"(OFF1) : "memory");
    return make_int2(v.x, v.y);
}
__device__ __forceinline__ void lds_peek2_wait(int2 &v) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v.x), "+v"(v.y) : : "memory");
}

#ifndef CS_OD_COLD_PARAMS
#define CS_OD_COLD_PARAMS 1
#endif
#if CS_OD_COLD_PARAMS
#define OD_COLD() cold_params()
#else
#define OD_COLD() p
#endif
// E3: a THIRD wavefront per 8 envs, E, owns the get_state tile and writes every output (reward, terminated, win, obs, state) --
// a quarter of D's plain step.  D, which also carries every reset and row top-up, is the pair's slower half (K alone sustains
// ~3500 cycles per step, D ~2650 + ~1450 of events); without the emission it has the slack to absorb its events.  D hands each
// step's reward / terminated / win / found mask to E through a ring of OdOut records; E reads the agents' floats from K's
// ring slot.  Three wavefronts of 128 VGPRs and 32 KB of LDS: four workgroups per CU, so this variant serves batches up to 8192 envs.
template <int ENVS>
struct __attribute__((aligned(16))) OdOutT {
    float reward[ENVS];
    int term[ENVS], win[ENVS];
    unsigned found[ENVS];
};
// -DCS_JITTER (test builds only, tests/test_gpu_jitter.py): a pseudo-random pause of 0..1800 cycles -- up to two thirds of a step --
// in each role at every hand-shake of the pair's protocol (before a counter is read, before it is posted, around a fix request and
// its acknowledgement).  The K / D / E hand-shakes rest on LDS ordering with hand-placed ds instructions and no fence; the parity
// suite exercises the protocol's LOGIC (mispredictions every other step) but at the kernel's natural timing only.  With the pauses
// every interleaving of the three wavefronts that the counters allow actually happens; results must not move by a bit.
#ifdef CS_JITTER
#define OD_JITTER(salt) do { \
        unsigned jh_ = (unsigned)blockIdx.x * 2654435761u ^ (unsigned)(s + 1) * 40503u ^ (unsigned)(salt) * 2246822519u ^ (unsigned)role * 3266489917u; \
        jh_ ^= jh_ >> 15; jh_ *= 2246822519u; jh_ ^= jh_ >> 13; \
        for (unsigned jq_ = __builtin_amdgcn_readfirstlane(jh_ & 7u); jq_ > 0u; jq_--) __builtin_amdgcn_s_sleep(4); \
    } while (0)
#else
#define OD_JITTER(salt) do {} while (0)
#endif
// (the 5-lane packing is compiled for THREE wavefronts per SIMD -- 168 VGPRs: a third target per lane is 15 more ranks and masks, and
// its workgroups of 12 envs are fewer: 16384 envs are six two-wavefront workgroups per CU, 8192 three three-wavefront ones.  The
// compiler derives a kernel's occupancy from its LDS use and would otherwise give these kernels the registers of two wavefronts.)
template <int N, bool VEC, bool EMIT, bool E3, int LG>
__device__ __forceinline__ void rollout_od_body(const DevParams &p, const StepIO &io);
template <int N, bool VEC, bool EMIT, bool E3>
__global__ __launch_bounds__(E3 ? OD_BLOCK + 64 : OD_BLOCK, CS_OD_WAVES) void k_rollout_od(DevParams p, StepIO io) {
    rollout_od_body<N, VEC, EMIT, E3, OG>(p, io);
}
template <int N, bool E3>
__global__ __launch_bounds__(E3 ? OD_BLOCK + 64 : OD_BLOCK, 3) void k_rollout_od5(DevParams p, StepIO io) {
    rollout_od_body<N, true, true, E3, 5>(p, io);
}
template <int N, bool VEC, bool EMIT, bool E3, int LG>
__device__ __forceinline__ void rollout_od_body(const DevParams &p, const StepIO &io) {
    static_assert(!E3 || (VEC && EMIT), "the emitting wavefront has the full-wavefront, obs + state stores only");
    static_assert(LG == OG || (LG == 5 && N == 5 && VEC && EMIT), "the 5-lane packing: teams of 5, full wavefronts, obs + state written");
    using Lay = OctLay<LG>;
    constexpr int ENVS = Lay::ENVS;   // envs per workgroup (= per wavefront of each role)
    constexpr int AP = LG == OG ? OCT_PAD : N + 1;                                  // columns of the per-agent LDS rows
    constexpr int TW = LG == OG ? TILE_W : 4 * N + 3 * (CS_MAX_TARGETS - 1);       // widest get_state row (the packing: <= 15 targets)
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    constexpr int OD_RING = od_ring(E3);
    using OdShared = OdSharedT<OD_RING, ENVS, AP, TW>;
    using OdRing = OdRingT<ENVS, AP>;
    using OdOut = OdOutT<ENVS>;
    __shared__ OdShared sh;
    __shared__ OdOut outs[E3 ? OD_RING : 1];
    constexpr bool EREF = E3 && (CS_OD_E_REFRESH != 0);
    __shared__ OdRefreshT<EREF> rf;
    int &e_steps = sh.e_steps;
    const int lane = threadIdx.x & 63;
    // Which wavefront of the workgroup plays which role decides who shares a SIMD: at 4096 envs a CU holds two workgroups,
    // six wavefronts on four SIMDs, handed out in order -- wavefront 0 of one workgroup lands beside wavefront 1 of the other,
    // wavefront 1 beside wavefront 2.  With E (busy a third of the time) in the middle, K and D -- the two full-time
    // wavefronts -- only ever share with an E.  Measured (us per step, 100-step launches, 3 agents x 4096 envs): K,E,D 1.62;
    // D,E,K 1.62; E,K,D 1.72; K,D,E 1.81; E,D,K 1.74; D,K,E 1.74.
#ifndef CS_ODE_ROLES
#define CS_ODE_ROLES 0x120   /* nibble w = role of wavefront w of the workgroup (0: K, 1: D, 2: E) */
#endif
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = E3 ? (CS_ODE_ROLES >> (4 * wv)) & 15 : wv;   // 0: K, 1: D, 2: E
    const bool is_k = role == 0;
    SPIN_DECL;
    const int o = Lay::env(lane), sh8 = Lay::first(lane);
    int t = Lay::t(lane);   // (made opaque once per step: lane predicates are recomputed, not held in SGPR pairs)
    const int tc = t < AP ? t : AP - 1;   // column of the per-agent LDS rows (a lane that holds nothing: the padding column)
    const int wave_b0 = io.env0 + blockIdx.x * ENVS;
    const int b_end = io.env0 + io.env_n;
    const int b = wave_b0 + o;
    const bool live = (VEC || b < b_end) && Lay::valid(lane);
    if (role < 2) BLK_STAMP(is_k ? 0 : 4);
    const int nvalid = b_end - wave_b0 < ENVS ? b_end - wave_b0 : ENVS;   // >= 1: the grid covers env_n exactly
    const int W = 4 * N + 3 * p.n_targets;
    bool ag = t < N;
    const bool auto_reset = io.flags & CS_AUTO_RESET, freeze = io.flags & CS_FREEZE_DONE;
    const size_t bl = live ? (size_t)b : (size_t)io.env0;
    EnvO<N, LG> e;
    {
        const int4 *h4 = reinterpret_cast<const int4 *>(p.hdr + bl * CS_H_WORDS);
        const int4 h0 = h4[0], h1 = h4[1], h2 = h4[2];
        e.found = (unsigned)h0.x;
        e.newly = (unsigned)h0.y;
        e.target_find = h0.z;
        e.flags = h0.w;
        e.time_step = h1.x;
        e.total_reward = h1.y;
        e.mt_pos = h1.z;
        e.episodes = h1.w;
        e.words = (unsigned long long)(unsigned)h2.x | ((unsigned long long)(unsigned)h2.y << 32);
        e.curr_reward = h2.z;
        e.newly_reset = (unsigned)h2.w;
    }
    if (!live) {   // a lane without an env never steps, resets or asks for a top-up
        e.target_find = 0;
        e.time_step = 0;
    }

    if (is_k) {
        // ------------------------------------------------------------------------------------------ K: kinematics
        // with the emitting wavefront K bounds the pipeline: it wins the issue arbitration against whoever shares its SIMD
        // (an E of the neighbouring workgroup at 4096 envs, two or three other wavefronts at 8192: -2 % / -4 % per step)
#ifndef CS_ODE_KPRIO
#define CS_ODE_KPRIO 3
#endif
        if (E3) __builtin_amdgcn_s_setprio(CS_ODE_KPRIO);
        {
            const double4 a = reinterpret_cast<const double4 *>(p.agent + bl * CS_MAX_AGENTS * 4)[t < CS_MAX_AGENTS ? t : 0];
            e.x = a.x;
            e.y = a.y;
            e.yaw = a.z;
        }
        const int aidx = ag ? t : N - 1;   // lanes without an agent repeat the last agent's (valid) address
        const int astride = (io.flags & CS_ACTIONS_I64) ? 2 : 1;
        const int *ap = reinterpret_cast<const int *>(io.actions) + (bl * N + aidx) * astride;   // this lane's action of step 0
        const size_t astep = (size_t)p.B * N * astride;
        int act = ap[0];
        if (io.T > 1) ap += astep;
        int act_next = ap[0];   // one step ahead of its use
        if (io.T > 2) ap += astep;
        load_trig_to_lds(T);
        trig_heading(T, e.yaw, e.sn, e.cs);   // what a frozen env keeps emitting
        sh.kpos[o][tc] = make_double2(e.x, e.y);
        bool k_done = live && (e.target_find >= p.n_targets || e.time_step >= p.time_limit);   // exact at launch
        int k_time = e.time_step;
        unsigned k_out = ((unsigned)e.flags >> 8) & 0xffu;
        auto peek = [](const int *w) __attribute__((always_inline)) { return lds_peek(w); };
        auto post = [&](int *w, int v) __attribute__((always_inline)) {   // (lds_post above: LDS-only ordering)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) lds_post(w, v);
        };
        int fix_seen = 0;
        // the state after step `sp` from the state after step sp - 1, for the octets in `sel`, into ring slot sp % OD_RING
        auto produce = [&](int sp, int a, bool sel, auto &&between) __attribute__((always_inline)) {
            const bool rs = sel && live && k_done && auto_reset;   // predicted reset (flight_env_easy.py:139-180: start poses)
            if (__ballot(rs)) {
                if (rs) {
                    const StartTab<N> st = start_tab<N>();
                    start_pick<N>(st, ag ? t : 0, e.x, e.y);
                    e.yaw = st.yaw;
                    sh.kpos[o][tc] = make_double2(e.x, e.y);
                    k_out = 0u;
                    k_time = 0;
                    k_done = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            const bool stepping = sel && live && !(k_done && freeze);
#ifdef CS_OD_ABL_NOKIN   /* experiment: what D alone sustains */
            const unsigned out = 0u;
#else
            const unsigned out = oct_kinematics<N, (N >= CS_OD_SHARED_DIV_FROM_N), LG, AP>(p, T, sh.kpos, o, t, sh8, stepping, a, e, sp);
#endif
            KIN_STAMP_SP(6);
            between();   // (the main loop issues its next flow-control read here)
            if (stepping) {
                k_out = out;
                k_time += 1;
                k_done = k_time >= p.time_limit;   // a win is D's knowledge: see the fix-up below
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // every lane has read the old positions
            OdRing &r = sh.ring[sp & (OD_RING - 1)];
            if (sel && Lay::valid(lane)) {
                const double2 xy = make_double2(e.x, e.y);
                sh.kpos[o][t] = xy;
                r.pos[o][t] = xy;
                r.yaw[o][t] = e.yaw;
                r.cssn[o][t] = make_float2((float)e.cs, (float)e.sn);
                if (t == 0) r.out[o] = k_out;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        BLK_STAMP(1);
        const int *const abase = reinterpret_cast<const int *>(io.actions) + (bl * N + aidx) * astride;
        // D reported a termination K could not predict (an env found its last target at step fs before the time limit):
        // restore the env as it was after step fs from the ring (D holds that slot), mark it done -- the next produce then
        // resets or freezes it like a predicted termination -- and redo the steps already produced past fs, for it alone
        auto handle_fix = [&](int produced) __attribute__((always_inline)) {
            const int req = peek(&sh.fix_req);
            if (__builtin_expect(req != fix_seen, 0)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int fs = req - 1;
                const bool mine = live && ((sh.fix_mask >> o) & 1u);
                if (mine) {
                    const OdRing &r = sh.ring[fs & (OD_RING - 1)];
                    const double2 xy = r.pos[o][t];
                    e.x = xy.x;
                    e.y = xy.y;
                    e.yaw = r.yaw[o][t];
                    trig_heading(T, e.yaw, e.sn, e.cs);
                    sh.kpos[o][tc] = xy;
                    k_out = r.out[o];
                    k_done = true;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int j = fs + 1; j < produced; j++) produce(j, abase[(size_t)j * astep], mine, []() {});
                fix_seen = req;
                post(&sh.fix_ack, req);
            }
        };
#ifndef CS_OD_EARLY_PEEK
#define CS_OD_EARLY_PEEK 0
#endif
        constexpr int OFF_FIX = (int)(offsetof(OdShared, fix_req) - offsetof(OdShared, d_steps)) / 4;
        constexpr int OFF_E = (int)(offsetof(OdShared, e_steps) - offsetof(OdShared, d_steps)) / 4;
        int2 pv = make_int2(0, 0);
        for (int s = 0; s < io.T; s++) {   // (D zeroed the counters before the barrier that published the trig table)
            asm volatile("" : "+v"(t));
            ag = t < N;
            DUO_STAMP(0);
            OD_JITTER(1);
            const int act_after = ap[0];
            if (s + 3 < io.T) ap += astep;
            // flow control: slot s % OD_RING is free once D has finished step s - OD_RING (E3: ... once E has written step
            // s - OD_RING out; E never passes D).  The progress word and D's fix request come in ONE LDS round trip, and the
            // common case -- slot free, nothing to fix -- touches none of the fix-up code (whose state updates otherwise cost a
            // row of register copies at every pass through the loop head).
            // (the words were requested in the middle of the previous step -- CS_OD_EARLY_PEEK -- and may be that old: both only ever
            // grow, so an old progress word can only make K look again below, and an old fix_req only delays the fix by a step.  The
            // slot rule holds as before: K overwrites slot s after ONE read that showed d_steps (e_steps) > s - RING, and that read
            // also returned every fix_req posted before that progress word)
            if (!CS_OD_EARLY_PEEK || s == 0) pv = E3 ? lds_peek2<OFF_E, OFF_FIX>(&sh.d_steps) : lds_peek2<0, OFF_FIX>(&sh.d_steps);
            if (__builtin_expect(pv.x <= s - OD_RING || pv.y != fix_seen, 0)) {
                for (;;) {
                    // progress word FIRST, fix request second: the request that belongs to a progress value was posted before it,
                    // so a fix read issued after the progress read cannot miss it (the other order could see an old fix_req and a
                    // new progress word and overwrite the very slot the fix restores from)
                    const int prog = peek(E3 ? &e_steps : &sh.d_steps);
                    handle_fix(s);
                    if (prog > s - OD_RING) break;
                    SPIN_TICK;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            DUO_STAMP(2);
            produce(s, act, true, [&]() __attribute__((always_inline)) {
                if (CS_OD_EARLY_PEEK) pv = E3 ? lds_peek2_issue<OFF_E, OFF_FIX>(&sh.d_steps) : lds_peek2_issue<0, OFF_FIX>(&sh.d_steps);
            });
            DUO_STAMP(1);
            OD_JITTER(2);
            post(&sh.k_steps, s + 1);
            // the words requested in the middle of this step arrived long ago: the wait is free here, and it sits INSIDE the iteration
            // that issued the read -- between the two asm statements the compiler believes the registers already hold the words, so
            // nothing but straight-line code may lie there (a copy at the loop's back edge, say, would copy them too early)
            if (CS_OD_EARLY_PEEK) lds_peek2_wait(pv);
            act = act_next;
            act_next = act_after;
        }
        BLK_STAMP(2);
        SPIN_STORE(0);
        // D may still report an unpredicted termination of a step K has long left behind: stay until it has judged step T - 2
        // (the last one whose successor exists)
        while (peek(&sh.d_steps) < io.T - 1) {
            handle_fix(io.T);
            __builtin_amdgcn_s_sleep(4);
        }
        handle_fix(io.T);
        if (live && ag)   // agents are K's part of the state
            reinterpret_cast<double4 *>(OD_COLD().agent + (size_t)b * CS_MAX_AGENTS * 4)[t] = make_double4(e.x, e.y, e.yaw, 0.0);
        BLK_STAMP(3);
        return;
    }

    if (E3 && role == 2) {
        // ------------------------------------------------------------------------------------------ E: emission
        const double2 *t2 = reinterpret_cast<const double2 *>(p.tgt + bl * G * 2);
        double2 tk[Lay::TPL];
#pragma unroll
        for (int k = 0; k < Lay::TPL; k++) tk[k] = t2[(t + LG * k) & (G - 1)];
        load_trig_to_lds(T);   // (K's table: E only joins the barrier; D zeroed the counters before it)
        float *row = sh.tile + o * W;
        // persistent rows: targets' normalised coordinates (rewritten by D when an env resets) and found flags (get_state, :190-216)
#pragma unroll
        for (int k = 0; k < Lay::TPL; k++) {
            if (t + LG * k < p.n_targets) {
                row[4 * N + 3 * (t + LG * k) + 0] = (float)((tk[k].x - p.mid) * p.inv_half);   // norm_target
                row[4 * N + 3 * (t + LG * k) + 1] = (float)((tk[k].y - p.mid) * p.inv_half);
            }
        }
        auto peek = [](const int *w) __attribute__((always_inline)) { return lds_peek(w); };
        constexpr int W_MAX = 4 * N + 3 * CS_MAX_TARGETS;
        constexpr int Q = (ENVS * W_MAX / 4 + 63) / 64;   // float4 chunks per lane of the largest tile
        const int ol = lane < ENVS * N ? lane : ENVS * N - 1;
        const int orow = ol / N, oag = ol - orow * N;
        const int obs_lds = orow * W + 4 * oag;
        const int rtw = lane < ENVS ? lane : ENVS - 1;   // (duplicates write the same value)
        float *p_rew = io.reward + wave_b0 + rtw;
        uint8_t *p_term = io.terminated + wave_b0 + rtw, *p_win = io.win + wave_b0 + rtw;
        v4f *p_obs = reinterpret_cast<v4f *>(io.obs + (size_t)wave_b0 * N * 4) + ol;
        v4f *p_st = reinterpret_cast<v4f *>(io.state + (size_t)wave_b0 * W);
        int chunk[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) chunk[q] = lane + 64 * q < ENVS * W / 4 - 1 ? lane + 64 * q : ENVS * W / 4 - 1;
        int rf_served = 0;
        auto rf_serve = [&]() __attribute__((always_inline)) {   // EREF: a row refresh for D, if one is asked for
            const int seq = peek(&rf.rf_req);
            if (__builtin_expect(seq == rf_served, 1)) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int g = __builtin_amdgcn_readfirstlane(rf.rf_env), pos = __builtin_amdgcn_readfirstlane(rf.rf_pos);
            const int a = __builtin_amdgcn_readfirstlane(rf.rf_ahead);
            unsigned *m = OD_COLD().mt + (size_t)(wave_b0 + g) * MT_STRIDE;
            RowRegs rr;
            row_load(m, lane, rr);
            row_to_lds(rr, rf.erow, lane);
            row_twist_ahead(rf.erow, m, pos, a < 0 ? 0 : a, lane);
#pragma unroll
            for (int it = 0; it < TAPE_DW / 2; it++) {
                const unsigned long long bm = row_slot_hits(OD_COLD(), rf.erow, pos, it, lane);
                if (lane == 0) {
                    rf.rf_tape[2 * it] = (unsigned)(bm & 0xffffffffull);
                    rf.rf_tape[2 * it + 1] = (unsigned)(bm >> 32);
                }
            }
            drain_vmem();   // the new words are in memory before D learns of them (its resets read stream words from there)
            {
                const int s = seq;   // (the jitter hash's step)
                (void)s;
                OD_JITTER(11);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) lds_post(&rf.rf_done, seq);
            rf_served = seq;
        };
        for (int s = 0; s < io.T; s++) {
            asm volatile("" : "+v"(t));
            ag = t < N;
            OD_JITTER(3);
            if (EREF) rf_serve();
            while (peek(&sh.d_steps) <= s) {   // D has judged step s: its record and K's slot are final
                if (EREF) rf_serve();           // (D may be waiting for the refresh before it can finish the step)
                SPIN_TICK;
                __builtin_amdgcn_s_sleep(1);
            }
            OD_JITTER(4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const OdRing &r = sh.ring[s & (OD_RING - 1)];
            const OdOut &d = outs[s & (OD_RING - 1)];
            if (ag) {
                const double2 xy = r.pos[o][t];
                const float2 cs = r.cssn[o][t];
                row[4 * t + 0] = (float)((xy.x - p.mid) * p.inv_half);
                row[4 * t + 1] = (float)((xy.y - p.mid) * p.inv_half);
                row[4 * t + 2] = cs.x;
                row[4 * t + 3] = cs.y;
            }
            const unsigned found = d.found[o];
#pragma unroll
            for (int k = 0; k < Lay::TPL; k++)
                if (t + LG * k < p.n_targets) row[4 * N + 3 * (t + LG * k) + 2] = ((found >> (t + LG * k)) & 1u) ? 1.0f : 0.0f;
            const float o_rew = d.reward[rtw];
            const int o_term = d.term[rtw], o_win = d.win[rtw];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v4f o_obs, o_st[Q];
            {
                const float *src = sh.tile + obs_lds;
                o_obs = v4f{src[0], src[1], src[2], src[3]};
                const float4 *src4 = reinterpret_cast<const float4 *>(sh.tile);
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const float4 v = src4[chunk[q]];
                    o_st[q] = v4f{v.x, v.y, v.z, v.w};
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            *p_rew = o_rew;
            *p_term = (uint8_t)o_term;
            *p_win = (uint8_t)o_win;
            p_rew += p.B;
            p_term += p.B;
            p_win += p.B;
            __builtin_nontemporal_store(o_obs, p_obs);
            p_obs += (size_t)p.B * N;
#pragma unroll
            for (int q = 0; q < Q; q++) __builtin_nontemporal_store(o_st[q], p_st + chunk[q]);
            p_st += (size_t)p.B * W / 4;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the tile reads above are complete: their values are in registers)
            OD_JITTER(5);
            if (lane == 0) lds_post(&e_steps, s + 1);
        }
        if (EREF) {   // D may still ask until its loop has ended (it waits for every answer before it says so)
            for (;;) {
                rf_serve();
                if (peek(&rf.d_done)) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        SPIN_STORE(2);
        return;
    }

    // ---------------------------------------------------------------------------------------------- D: detection
    // The pair variant (two K and two D wavefronts per SIMD at 16384 envs): D ahead of K in the issue arbitration.  One box, two passes,
    // us per step at 16384 envs, priority 0 / 1 / 2 / 3: 5 agents 3.29-3.31 / 3.18 / 3.18-3.19 / 3.15-3.22, 3 agents 2.35-2.36 / 2.29-2.31 /
    // 2.30-2.34 / 2.28-2.31 (8192 envs, 3: 2.17 -> 2.08 / 1.64 -> 1.55); K at 3 instead: 3.25 / 2.38, and slower at 8192 and 32768 envs.
#ifndef CS_OD_DPRIO
#define CS_OD_DPRIO 1
#endif
    if (!E3) __builtin_amdgcn_s_setprio(CS_OD_DPRIO);
#ifndef CS_ODE_DPRIO
#define CS_ODE_DPRIO 2   /* three-wavefront variant: K (3) > D (2) > E (0) where wavefronts share a SIMD: -3 % per step at 8192 envs */
#endif
    if (E3) __builtin_amdgcn_s_setprio(CS_ODE_DPRIO);
    e.ahead = live ? p.ahead[bl] : (1 << 20);
    {
        const double2 *t2 = reinterpret_cast<const double2 *>(p.tgt + bl * G * 2);
#pragma unroll
        for (int k = 0; k < Lay::TPL; k++) {
            const double2 tk = t2[(t + LG * k) & (G - 1)];
            e.tx[k] = tk.x;
            e.ty[k] = tk.y;
        }
    }
    const TapeRaw traw = tape_fetch(p, (int)bl);
    if (lane == 0) {   // the pair's counters: zero before the barrier below lets K start
        sh.k_steps = 0;
        sh.d_steps = 0;
        sh.fix_req = 0;
        sh.fix_ack = 0;
        sh.fix_mask = 0u;
        e_steps = 0;
        rf.rf_req = 0;
        rf.rf_done = 0;
        rf.d_done = 0;
    }
    load_reset_tab(sh.rtab, lane);
    load_trig_to_lds(T);   // (K's table; D only joins its barrier -- after which K produces ahead, up to OD_RING steps)
    unsigned tape[TAPE_DW];
    bool tape_ok = tape_finish(p, traw, e, tape) || !live;
#ifndef CS_OD_LAZY_TAPE
#define CS_OD_LAZY_TAPE 1   /* the step's detection pass leaves the tape unshifted (oct_detect_impl, LAZY); 0: shifted every step */
#endif
    int tcur = 0;   // the cursor's bit within tape[0]; 0 = canonical, which everything but the step's own pass expects
    auto canon = [&]() __attribute__((always_inline)) { if (CS_OD_LAZY_TAPE) tape_canon(tape, tcur); };
    float *row = sh.tile + o * W;
    auto put_found = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < Lay::TPL; k++)
            if (t + LG * k < p.n_targets) row[4 * N + 3 * (t + LG * k) + 2] = ((e.found >> (t + LG * k)) & 1u) ? 1.0f : 0.0f;
    };
    if (!E3) {   // (E3: the tile is E's)
#pragma unroll
        for (int k = 0; k < Lay::TPL; k++) {
            if (t + LG * k < p.n_targets) {
                row[4 * N + 3 * (t + LG * k) + 0] = (float)((e.tx[k] - p.mid) * p.inv_half);   // norm_target
                row[4 * N + 3 * (t + LG * k) + 1] = (float)((e.ty[k] - p.mid) * p.inv_half);
            }
        }
        put_found();
    }
    auto peek = [](const int *w) __attribute__((always_inline)) { return lds_peek(w); };
    auto post = [&](int *w, int v) __attribute__((always_inline)) {   // (lds_post above: LDS-only ordering)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane == 0) lds_post(w, v);
    };
    constexpr int LOW = 2 * N * CS_MAX_TARGETS;   // words one step can consume
    // Rare events stall the whole pair (K waits at the barrier), and what they cost is mostly ONE dependent round trip to
    // memory: the MT19937 row of a top-up, the stream words of a reset's attempt batch.  Both are known a step ahead -- an
    // env running low on twisted words; an env whose step just terminated -- so D requests them at the end of that step
    // straight into LDS (global_load_lds: asynchronous, no registers) and uses them at the top of the next one.  An env
    // that cannot wait (several running low at once, a reset that consumed its words) is topped up on the spot.
    // D's steady-state loop waits for no load, so none of these paths needs to end drained (-4 % per step at 4096 envs).
#ifndef CS_OD_ASYNC
#define CS_OD_ASYNC 1   /* rows and reset words are fetched a step ahead, straight into LDS (global_load_lds: no registers; the
                           first version held the row in ten VGPRs across the step and was slower: spills in the hot path) */
#endif
#ifndef CS_OD_DRAIN
#define CS_OD_DRAIN 0
#endif
#ifndef CS_OD_REQ_SLACK
#define CS_OD_REQ_SLACK 64   /* words above one step's worst case at which an env's row is requested (an env that falls below LOW
                                before its turn is topped up on the spot).  The first setting, max(LOW, 96), refreshed a 5-agent row
                                with 300 of its 624 words still unused: every refresh costs the same ~4000 cycles whatever it twists */
#endif
    constexpr int REQ = CS_OD_ASYNC ? LOW + CS_OD_REQ_SLACK : 0;
    // The requests of a step are issued BEFORE its output stores, and loads / stores retire in order: waiting until no more
    // than the step's own stores are in flight is waiting for the requests -- without also sitting out the stores, which were
    // issued a few hundred cycles ago and take a memory round trip (measured: a plain vmcnt(0) here cost ~1000 cycles per event).
    // What this rests on, and what keeps it true:
    //  * gfx9 returns vector-memory loads AND stores through one in-order counter (vmcnt): "at most k outstanding" means everything
    //    issued before the last k operations has completed;
    //  * the stores after the requests are EXACTLY the STEP_STORES below, each one instruction, none conditional: EMIT && VEC is a
    //    compile-time property of the kernel (reward, terminated, win: three scalar stores; obs: one 16-byte store; state: Q 16-byte
    //    stores, Q being the very constant the store loop below runs over).  Every other variant -- stores behind `if (io.obs)`, the
    //    scalar tail loop, E3 -- takes drain_vmem();
    //  * -DCS_OD_SAFE_WAIT turns the counted wait into a full drain and -DCS_OD_ASYNC=0 removes the requests altogether: both builds
    //    must reproduce the shipped one bit for bit (tests/test_gpu_jitter.py builds and compares them).
    constexpr int W_MAX = 4 * N + 3 * CS_MAX_TARGETS;
    constexpr int Q = (ENVS * W_MAX / 4 + 63) / 64;   // float4 chunks per lane of the largest tile = state stores per step
    constexpr int STEP_STORES = 3 + 1 + Q;                // reward, terminated, win | obs | state
    static_assert(Q >= 1 && STEP_STORES == 4 + Q, "STEP_STORES counts the stores of the VEC && EMIT step: keep it next to them");
    auto wait_for_requests = [&]() __attribute__((always_inline)) {
#ifdef CS_OD_SAFE_WAIT
        drain_vmem();
#else
        if (!E3 && EMIT && VEC && STEP_STORES <= 15) __builtin_amdgcn_s_waitcnt(0x0F70 | STEP_STORES);   // vmcnt(STEP_STORES)
        else drain_vmem();   // (E3: D stores nothing per step)
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    int cand = -1;                       // env (octet) of the wavefront whose row is on its way into sh.rowbuf
    int ack_wait = 0;                    // fix request of the previous step that K has yet to acknowledge (0: none)
    // EREF: the refresh E is working on
    int rf_seq = 0, rf_pending = -1;     // sequence number of the latest request; octet it is for (-1: none outstanding)
    unsigned long long rf_words0 = 0ull; // this lane's env's word count when the request was posted
    auto rf_poll = [&](bool wait) __attribute__((always_inline)) {   // adopt E's answer (wait: stay until it is there)
        if (rf_pending < 0) return;
        if (wait) {
            {
                const int s = rf_seq;   // (the jitter hash's step)
                (void)s;
                OD_JITTER(12);
            }
            while (peek(&rf.rf_done) != rf_seq) __builtin_amdgcn_s_sleep(1);
        } else if (peek(&rf.rf_done) != rf_seq) {
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        canon();
        unsigned nt[TAPE_DW];
#pragma unroll
        for (int k = 0; k < TAPE_DW; k++) nt[k] = rf.rf_tape[k];
        const int c = (int)((e.words - rf_words0) >> 1);   // draw slots this lane's env has consumed since the request
        tape_shift<8>(nt, c);
        if (o == rf_pending) {
#pragma unroll
            for (int k = 0; k < TAPE_DW; k++) tape[k] = nt[k];
            e.ahead = MT_N - 2 * c;
            tape_ok = true;
        }
        rf_pending = -1;
    };
    unsigned long long pre_need = 0ull;  // the reset mask sh.prebuf was filled for
    unsigned pre_valid = 0u;             // bit g: 16-lane group g's attempt batch is (on its way) in sh.prebuf
    oct_wave_advance<N, CS_OD_DRAIN != 0, LG>(p, wave_b0, nvalid, lane, io.min_ahead > LOW ? io.min_ahead : LOW, sh.rowbuf, e, tape, tape_ok);   // while K produces step 0
    // ---- write-out plan (loop invariant)
    const int rows_valid = nvalid;
    const int ol = lane < rows_valid * N ? lane : rows_valid * N - 1;
    const int orow = ol / N, oag = ol - orow * N;
    const int obs_lds = orow * W + 4 * oag;
    const int rtw = lane < rows_valid ? lane : rows_valid - 1;   // (duplicates write the same value)
    const int t16 = lane & (G - 1), gshift16 = lane & ~(G - 1), grp = lane >> 4;
    float *p_rew = io.reward + wave_b0 + rtw;
    uint8_t *p_term = io.terminated + wave_b0 + rtw, *p_win = io.win + wave_b0 + rtw;
    v4f *p_obs = reinterpret_cast<v4f *>(io.obs + (size_t)wave_b0 * N * 4) + ol;
    v4f *p_st = reinterpret_cast<v4f *>(io.state + (size_t)wave_b0 * W);
    int chunk[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) chunk[q] = lane + 64 * q < ENVS * W / 4 - 1 ? lane + 64 * q : ENVS * W / 4 - 1;
    BLK_STAMP(5);
    for (int s = 0; s < io.T; s++) {
        asm volatile("" : "+v"(t));
        ag = t < N;
        DUO_STAMP(8);
        if (__builtin_expect(cand >= 0, 0)) {   // wave-uniform: the row requested a step ago is in sh.rowbuf
            wait_for_requests();
            canon();
            oct_advance_finish<N, LG>(OD_COLD(), wave_b0, cand, lane, sh.rowbuf, e, tape, tape_ok);
            cand = -1;
        }
        if (EREF) rf_poll(false);
        if (__builtin_expect(__ballot(live && e.ahead < LOW) != 0ull, 0)) {   // could not wait for its turn
            if (EREF) rf_poll(true);   // (E may be at this very row; and its answer may be all that was needed)
            canon();
            if (!EREF || __ballot(live && e.ahead < LOW) != 0ull)
                oct_wave_advance<N, CS_OD_DRAIN != 0, LG>(OD_COLD(), wave_b0, nvalid, lane, LOW, sh.rowbuf, e, tape, tape_ok);
        }
        bool done = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
        e.flags &= ~(FLAG_DIRTY | FLAG_RESET_PASS);
        // ---- auto-reset: target placement on the 16-lane code (one resetting env per 16-lane group and round), then the
        //      reset-time detection pass (quirk Q3) on the start poses
        const unsigned long long need = __ballot(live && done && auto_reset && t == 0);   // bit 8 o'
        if (__builtin_expect(need != 0ull, 0)) {
            DUO_STAMP(13);
            if (EREF) rf_poll(true);   // a reset tops rows up on the spot and reads stream words: not beside E's refresh
            canon();
            const DevParams &cp = OD_COLD();
            const bool mine = Lay::valid(lane) && ((need >> sh8) & 1ull);
            const StartTab<N> st = start_tab<N>();
            // round 0's attempt batches were requested when the envs' steps terminated (same mask -> same groups)
            // (E3: the tile still holds the rows of step s - 1 until E has written them out: the new targets wait for that)
            oct_place_targets<N, CS_OD_DRAIN != 0, LG>(cp, wave_b0, nvalid, lane, live, need, sh.rtab, sh.tgt, sh.tile, W, sh.rowbuf, e, tape, tape_ok,
                                                   [&]() __attribute__((always_inline)) {
                                                       if (E3) {
                                                           while (lds_peek(&e_steps) < s) __builtin_amdgcn_s_sleep(1);
                                                           __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                                                       }
                                                   },
                                                   [&](unsigned (&w)[4]) __attribute__((always_inline)) {
                                                       const bool ok = CS_OD_ASYNC && need == pre_need && ((pre_valid >> grp) & 1u);
                                                       if (ok) {
                                                           wait_for_requests();
#pragma unroll
                                                           for (int k = 0; k < 4; k++) w[k] = sh.prebuf[64 * k + lane];
                                                       }
                                                       return ok;
                                                   });
            if (mine) {
                e.episodes += 1;
                e.found = 0;
                e.newly = 0;
                e.target_find = 0;
                e.time_step = 0;
                e.total_reward = 0;
                e.flags = 0;
                double sx, sy;
                start_pick<N>(st, ag ? t : 0, sx, sy);
                sh.dpos[o][tc] = make_double2(sx, sy);
            }
            DUO_STAMP(14);
            if (__ballot(live && e.ahead < LOW)) oct_wave_advance<N, CS_OD_DRAIN != 0, LG>(cp, wave_b0, nvalid, lane, LOW, sh.rowbuf, e, tape, tape_ok);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            DUO_STAMP(15);
            // (agent_mode 0 with the shipped target file never has a target within view of a start pose: the pass -- whose
            // reward is discarded anyway -- is then three assignments; the test costs a third of the pass it usually saves)
            bool near = false;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const double sx = st.x[i], sy = st.y[i];
#pragma unroll
                for (int k = 0; k < Lay::TPL; k++) {
                    const double axk = e.tx[k] - sx, ayk = e.ty[k] - sy;
                    near = near | ((t + LG * k < cp.n_targets) & (axk * axk + ayk * ayk <= cp.view_r2));
                }
            }
            if (__ballot(mine && near)) {
                oct_detect<N, LG, AP>(p, sh.dpos, o, t, sh8, mine, e, tape);
                if (!E3) put_found();
            } else if (mine) {   // what the pass does when no pair is in range: no draw, reward -1
                e.newly = 0u;
                e.curr_reward = -1;
                e.flags |= FLAG_DIRTY;
            }
            done = done && !mine;
            if (__ballot(live && e.ahead < LOW)) oct_wave_advance<N, CS_OD_DRAIN != 0, LG>(cp, wave_b0, nvalid, lane, LOW, sh.rowbuf, e, tape, tape_ok);
        }
        const bool stepping = live && !(done && freeze);
        DUO_STAMP(9);
        OD_JITTER(6);
        // ---- K's step s (normally produced long ago): out flags, the agents' four floats (get_obs / get_state), positions
        if (__builtin_expect(ack_wait != 0, 0)) {   // ... redone for the envs of the previous step's fix request (see below)
            OD_JITTER(10);
            while (peek(&sh.fix_ack) != ack_wait) __builtin_amdgcn_s_sleep(1);
            ack_wait = 0;
        }
        while (peek(&sh.k_steps) <= s) { SPIN_TICK; __builtin_amdgcn_s_sleep(1); }
        OD_JITTER(7);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef CS_OD_ABL_NODET   /* experiment: what K alone sustains */
        post(&sh.d_steps, s + 1);
        continue;
#endif
        const OdRing &r = sh.ring[s & (OD_RING - 1)];
        if (live) e.flags = (e.flags & ~0xff00) | (int)(r.out[o] << 8);
        if (!E3 && ag) {
            const double2 xy = r.pos[o][t];
            const float2 cs = r.cssn[o][t];
            row[4 * t + 0] = (float)((xy.x - p.mid) * p.inv_half);
            row[4 * t + 1] = (float)((xy.y - p.mid) * p.inv_half);
            row[4 * t + 2] = cs.x;
            row[4 * t + 3] = cs.y;
        }
        const int reward = oct_detect_impl<N, LG, AP, CS_OD_LAZY_TAPE != 0>(p, r.pos, o, t, sh8, stepping, e, tape, tcur);
        DUO_STAMP(10);
        bool term = true, mispredicted = false;
        if (stepping) {
            e.total_reward += reward;
            e.time_step += 1;
            term = e.target_find >= p.n_targets || e.time_step >= p.time_limit;
            mispredicted = (auto_reset || freeze) && term && e.time_step < p.time_limit;   // K steps on unless the counter says otherwise
        }
        const unsigned long long mb = s + 1 < io.T ? __ballot(mispredicted && t == 0) : 0ull;
        if (__builtin_expect(mb != 0ull, 0)) {   // K has stepped these envs on as if nothing had happened: have it redo them
            unsigned m8 = 0;
#pragma unroll
            for (int q = 0; q < ENVS; q++) m8 |= (unsigned)((mb >> Lay::first_of(q)) & 1ull) << q;
            if (lane == 0) sh.fix_mask = m8;
            OD_JITTER(8);
            post(&sh.fix_req, s + 1);
        }
        if (E3) {   // this step's record for E (published with d_steps below)
            if (t == 0) {
                OdOut &d = outs[s & (OD_RING - 1)];
                d.reward[o] = (float)reward;
                d.term[o] = term ? 1 : 0;
                d.win[o] = (e.flags & FLAG_WIN) ? 1 : 0;
                d.found[o] = e.found;
            }
        } else {
            if (__ballot(stepping && e.newly != 0u)) put_found();   // wave-uniform: some env found a target in this step
            if (t == 0) {
                sh.reward[o] = (float)reward;
                sh.term[o] = term ? 1 : 0;
                sh.win[o] = (e.flags & FLAG_WIN) ? 1 : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- this step's outputs (all LDS reads first, then the stores)
        float o_rew = 0.f;
        int o_term = 0, o_win = 0;
        if (!E3) {
            o_rew = sh.reward[rtw];   // duplicates write the same value
            o_term = sh.term[rtw];
            o_win = sh.win[rtw];
        }
        v4f o_obs = {0.f, 0.f, 0.f, 0.f}, o_st[Q];
        if (!E3 && (EMIT || io.obs)) {
            const float *src = sh.tile + obs_lds;
            o_obs = v4f{src[0], src[1], src[2], src[3]};
        }
        if (!E3 && VEC && (EMIT || io.state)) {
            const float4 *src4 = reinterpret_cast<const float4 *>(sh.tile);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const float4 v = src4[chunk[q]];
                o_st[q] = v4f{v.x, v.y, v.z, v.w};
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (CS_OD_ASYNC && s + 1 < io.T) {   // requests for the next step, before this step's stores
            // (a) the row of the env running lowest on twisted words, if any is below REQ: ten dword columns -> sh.rowbuf
            const unsigned long long lowb = (EREF && rf_pending >= 0) ? 0ull : __ballot(live && e.ahead < REQ && t == 0);
            cand = lowb ? __builtin_amdgcn_readfirstlane(Lay::env_of_first(__ffsll((long long)lowb) - 1)) : -1;
            if (EREF && cand >= 0) {   // E's job: post the request, go on with the old tape
                if (o == cand && t == 0) {
                    rf.rf_env = cand;
                    rf.rf_pos = e.mt_pos;
                    rf.rf_ahead = e.ahead;
                }
                rf_words0 = e.words;
                rf_pending = cand;
                rf_seq += 1;
                OD_JITTER(13);
                post(&rf.rf_req, rf_seq);
                cand = -1;
            }
            if (__builtin_expect(cand >= 0, 0)) {
                const unsigned *m = OD_COLD().mt + (size_t)(wave_b0 + cand) * MT_STRIDE;
#pragma unroll
                for (int i = 0; i < 10; i++)   // (the tenth column reaches words 576..639: inside the row's 672, mirror included)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(m + lane + 64 * i),
                                                     (__attribute__((address_space(3))) void *)(sh.rowbuf + 64 * i), 4, 0, 0);
            }
            // (b) the first attempt batch of every env whose step just terminated: it resets at the top of the next step
            const unsigned long long nn = __ballot(live && stepping && term && auto_reset && t == 0);
            pre_need = nn;
            pre_valid = 0u;
            if (__builtin_expect(nn != 0ull, 0)) {
                unsigned long long mm = nn;
                for (int q = 0; q < grp; q++) mm &= mm ? mm - 1 : 0ull;   // this 16-lane group's env in round 0 (as in the reset)
                const int src = mm ? __ffsll((long long)mm) - 1 : -1;
                const int sl = src >= 0 ? src : lane;
                const int ppos = __shfl(e.mt_pos, sl), pah = __shfl(e.ahead, sl);
                const bool okg = src >= 0 && pah >= 4 * G;   // its words are twisted already: their stored values are final
                if (okg) {
                    const unsigned *m = OD_COLD().mt + (size_t)(wave_b0 + Lay::env_of_first(src)) * MT_STRIDE;
                    const int i0 = wrap624(ppos + 4 * t16);
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(m + wrap624(i0 + q)),
                                                         (__attribute__((address_space(3))) void *)(sh.prebuf + 64 * q), 4, 0, 0);
                }
                const unsigned long long vb = __ballot(okg);
                pre_valid = (unsigned)((vb >> 0) & 1ull) | ((unsigned)((vb >> 16) & 1ull) << 1) | ((unsigned)((vb >> 32) & 1ull) << 2) |
                            ((unsigned)((vb >> 48) & 1ull) << 3);
            }
        }
        if (!E3) {
            *p_rew = o_rew;
            *p_term = (uint8_t)o_term;
            *p_win = (uint8_t)o_win;
            p_rew += p.B;
            p_term += p.B;
            p_win += p.B;
        }
        if (!E3 && (EMIT || io.obs)) {   // one float4 per (env, agent)
            __builtin_nontemporal_store(o_obs, p_obs);
            p_obs += (size_t)p.B * N;
        }
        if (!E3 && (EMIT || io.state)) {
            if (VEC) {
#pragma unroll
                for (int q = 0; q < Q; q++) __builtin_nontemporal_store(o_st[q], p_st + chunk[q]);
                p_st += (size_t)p.B * W / 4;
            } else {
                float *dst = io.state + ((size_t)s * p.B + wave_b0) * W;
                for (int k = lane; k < rows_valid * W; k += 64) dst[k] = sh.tile[k];
            }
        }
        DUO_STAMP(11);
        // A fix request is NOT waited for here.  Slot s is safe without it: K may overwrite slot s only when it produces step
        // s + RING, which it does after a loop head at which it has seen d_steps (E3: e_steps) > s -- a read that also returns this
        // fix_req, posted earlier through the same in-order LDS queue -- and a seen request is handled before the next produce.
        // What D must wait for is the REDONE slot s + 1 (K had produced it long ago; k_steps says nothing about the redo): that
        // wait sits in front of the next step's read of the ring, AFTER that step's reset work -- so the reset of the env that just
        // won (target placement, 7-9 k cycles) runs beside K's redo (2-3 produce calls, 4.5-10 k) instead of after it.
        if (__builtin_expect(mb != 0ull, 0)) ack_wait = s + 1;
        OD_JITTER(9);
        post(&sh.d_steps, s + 1);
        DUO_STAMP(12);
    }
    BLK_STAMP(6);
    SPIN_STORE(1);
    if (EREF) {
        rf_poll(true);
        post(&rf.d_done, 1);
    }
    canon();
    if (live) {   // header, cursor and tape are D's part of the state; targets were stored at each reset
        const DevParams &cp = OD_COLD();
        if (t == 0) {
            int4 *h4 = reinterpret_cast<int4 *>(cp.hdr + (size_t)b * CS_H_WORDS);
            h4[0] = make_int4((int)e.found, (int)e.newly, e.target_find, e.flags);
            h4[1] = make_int4(e.time_step, e.total_reward, e.mt_pos, e.episodes);
            h4[2] = make_int4((int)(unsigned)(e.words & 0xffffffffull), (int)(unsigned)(e.words >> 32), e.curr_reward,
                              (int)e.newly_reset);
            cp.ahead[b] = e.ahead;
        }
        if (tape_ok) {
            U4 *tp = reinterpret_cast<U4 *>(cp.tape + (size_t)b * TAPE_STRIDE);
            if (t == 0) tp[0] = U4{tape[0], tape[1], tape[2], tape[3]};
            if (t == 1) tp[1] = U4{tape[4], tape[5], tape[6], tape[7]};
            if (t == 2) tp[2] = U4{tape[8], tape[9], (unsigned)(e.words & 0xffffffffull), (unsigned)(e.words >> 32)};
            if (t == 3) tp[3] = U4{(unsigned)(cp.detect_K & 0xffffffffull), (unsigned)(cp.detect_K >> 32), 0u, 0u};
        }
    }
    BLK_STAMP(7);
}
#include "rollout_lanev.h"   // k_rollout_lanev: the lane-per-env kernel built for three to four wavefronts per SIMD

template <int N>
__global__ __launch_bounds__(BLOCK) void k_reset(DevParams p, const uint8_t *mask, int init, float *obs, float *state) {
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    load_trig_to_lds(T);
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    const int b = gid / G, t = gid % G;
    if (b >= p.B) return;
    const int gshift = (int)(threadIdx.x & 63) & ~15;
    Env<N> e;
    env_load<N>(p, b, t, e);
    if (!mask || mask[b]) {
        env_reset<N>(p, T, b, t, gshift, init, e);
        env_store<N>(p, b, t, e, true);
    } else {
        env_trig<N>(T, e);
        if (e.flags & (FLAG_DIRTY | FLAG_RESET_PASS)) {
            e.flags &= ~(FLAG_DIRTY | FLAG_RESET_PASS);
            env_store<N>(p, b, t, e, false);
        }
    }
    if (p.variant == 1) job_store<N>(p, 0, b, t, e);   // the sweep that follows a reset reads record 0
    const size_t obs_w = (size_t)N * p.obs_row_w;
    const size_t st_w = (size_t)(4 * N + 3 * p.n_targets);
    emit<N>(p, t, e, obs ? obs + (size_t)b * obs_w : nullptr, state ? state + (size_t)b * st_w : nullptr);
}

template <int N>
__global__ __launch_bounds__(BLOCK) void k_emit(DevParams p, float *obs, float *state) {
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    load_trig_to_lds(T);
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    const int b = gid / G, t = gid % G;
    if (b >= p.B) return;
    Env<N> e;
    env_load<N>(p, b, t, e);
    env_trig<N>(T, e);
    const size_t obs_w = (size_t)N * p.obs_row_w;
    const size_t st_w = (size_t)(4 * N + 3 * p.n_targets);
    emit<N>(p, t, e, obs ? obs + (size_t)b * obs_w : nullptr, state ? state + (size_t)b * st_w : nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// flight: probability-map update (flight_env.py:275-303) fused with the map part of get_obs (:223-230).
// One workgroup per env streams the 10 KB map once: float4 per lane, update the cells whose corners fall in a
// sensor disc (only when the env ran a detection pass since the last call), write the map back only where it
// changed, and write the n copies that get_obs emits.  An env that was auto-reset inside k_step carries two
// pending passes (reset-time pass at the start positions, then the step's pass); both are applied, in order,
// in the same sweep.
//
// Corner test `(x-ax)**2 + (y-ay)**2 < view_range**2` (strict, flight_env.py:300): decided in fp32 when the
// fp32 distance is clear of the threshold by more than its error bound, in exact fp64 otherwise.
// ---------------------------------------------------------------------------------------------------------
// Per-pass data of one env in LDS.
struct MapPassLds {
    unsigned long long rowbits[CS_MAX_MAP + 2];  // bit Y of rowbits[X]: lattice point (X, Y) strictly inside a disc
    int cells[CS_MAX_TARGETS];                   // flat cell index of each newly found target, -1 otherwise
    int any_found;
};

// The reference's corner test, exactly: (x-ax)**2 + (y-ay)**2 < view_range**2 (flight_env.py:299-300), with dx2 =
// (x-ax)*(x-ax) hoisted (same fp64 value).
__device__ __forceinline__ bool corner_exact(double dx2, int yi, double ay, double r2) {
    const double dy = (double)yi - ay;
    return dx2 + dy * dy < r2;
}

// Lattice bitmap of one pass, built by ONE wavefront: lane X owns lattice row X (0..map_size).  For a fixed row
// the exact predicate is monotone in |Y - ay| (fp64 rounding is monotone), so each agent covers a contiguous
// interval [lo, hi] of columns: an fp32 sqrt gives the estimate (error << 1) and the three lattice columns around
// each end are settled with the exact fp64 comparison.
template <int N>
__device__ __forceinline__ void build_rowbits(const DevParams &p, const double (&ax)[N], const double (&ay)[N], int lane,
                                              unsigned long long *rowbits) {
    unsigned long long bits = 0;
    const int X = lane;
    if (X <= p.map_size) {
#pragma unroll
        for (int a = 0; a < N; a++) {
            const double dxd = (double)X - ax[a];
            const double dx2 = dxd * dxd;
            const float w2 = (float)(p.view_r2 - dx2);
            if (w2 > -0.01f) {
                const float w = sqrtf(fmaxf(w2, 0.0f));
                const float ayf = (float)ay[a];
                const int y0 = (int)ceilf(ayf - w), y1 = (int)floorf(ayf + w);
                // first column of [y0-1, y0+1] and last column of [y1-1, y1+1] that pass the exact test
                const bool l0 = corner_exact(dx2, y0 - 1, ay[a], p.view_r2), l1 = corner_exact(dx2, y0, ay[a], p.view_r2),
                           l2 = corner_exact(dx2, y0 + 1, ay[a], p.view_r2);
                const bool h0 = corner_exact(dx2, y1 + 1, ay[a], p.view_r2), h1 = corner_exact(dx2, y1, ay[a], p.view_r2),
                           h2 = corner_exact(dx2, y1 - 1, ay[a], p.view_r2);
                int lo = l0 ? y0 - 1 : (l1 ? y0 : y0 + 1);
                int hi = h0 ? y1 + 1 : (h1 ? y1 : y1 - 1);
                const bool any = (l0 | l1 | l2) & (h0 | h1 | h2);
                lo = lo < 0 ? 0 : lo;
                hi = hi > p.map_size ? p.map_size : hi;
                if (any && lo <= hi) {
                    const unsigned long long upto_hi = hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull);
                    bits |= upto_hi & ~((1ull << lo) - 1ull);
                }
            }
        }
        rowbits[X] = bits;
    }
}

// flight: probability-map update (flight_env.py:275-303) fused with the map part of get_obs (:223-230).
// Each workgroup streams its share of one env's 10 KB map once: float4 per lane, update the cells with a corner
// in a sensor disc (only when the env ran a detection pass in the preceding k_step / k_reset), write the map back
// only where it changed, and write the n copies that get_obs emits (write-once stream: non-temporal stores).
// An env that was auto-reset inside k_step carries two pending passes (reset-time pass at the start positions,
// then the step's pass); both are applied, in order, in the same sweep.
//
// Launch: grid (B, ceil(chunks / MAP_BLOCK)), MAP_BLOCK threads: several small workgroups per env so that a CU
// holds many of them and one workgroup's load latency overlaps another's arithmetic and stores.  The pending-
// update flags are written only by k_step / k_reset (set or cleared on every launch), never here, so the
// workgroups of one env need no ordering; `apply` = 0 makes this a pure get_obs sweep (cs_emit).
#ifndef CS_MAP_BLOCK
#define CS_MAP_BLOCK 256
#endif
#ifndef CS_MAP_NT
#define CS_MAP_NT 1
#endif
#ifndef CS_MAP_ILP
#define CS_MAP_ILP 1
#endif
constexpr int MAP_BLOCK = CS_MAP_BLOCK;
constexpr int MAP_ILP = CS_MAP_ILP;   // float4 chunks per thread, all loaded before the first is processed

// The pending pass(es) applied to float4 chunk c of an env's map: true if a cell changed (flight_env.py:275-303).
__device__ __forceinline__ bool map_update_chunk(const DevParams &p, const MapPassLds *s_pass, bool dirty, bool reset_pass, int c,
                                                 float4 &v) {
    const float qf = (float)p.q;
    const float inv_map = 1.0f / (float)p.map_size;
    float pv[4] = {v.x, v.y, v.z, v.w};
    const int cell0 = 4 * c;
    const int ci = (int)(((float)cell0 + 0.5f) * inv_map);  // exact for cell0 < 4096
    const int cj0 = cell0 - ci * p.map_size;
    unsigned any = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (k == 0 ? !reset_pass : !dirty) continue;
        const MapPassLds &m = s_pass[k];
        const unsigned long long r0 = m.rowbits[ci], r1 = m.rowbits[ci + 1];
        const unsigned long long r2 = m.rowbits[ci + 2 <= CS_MAX_MAP + 1 ? ci + 2 : CS_MAX_MAP + 1];
        unsigned cnts = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool wrap = cj0 + q >= p.map_size;  // chunk straddles two rows when map_size % 4 != 0
            const int yi = wrap ? cj0 + q - p.map_size : cj0 + q;
            const unsigned long long ra = wrap ? r1 : r0, rb = wrap ? r2 : r1;
            const int cnt = __popc((unsigned)((ra >> yi) & 3ull)) + __popc((unsigned)((rb >> yi) & 3ull));
            cnts |= (unsigned)cnt << (4 * q);
        }
        if (cnts == 0) continue;   // no corner of these four cells in view (4 of 5 chunks): nothing to update
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int cnt = (int)((cnts >> (4 * q)) & 0xfu);
            // percent*(1-detect_prob)*p / ((1-detect_prob)*p + (1-p)), flight_env.py:292
            const float upd = ((float)cnt * 0.25f) * qf * pv[q] / (qf * pv[q] + (1.0f - pv[q]));
            pv[q] = cnt ? upd : pv[q];
        }
        if (m.any_found && cnts) {  // a newly found target's cell, if in view, is set to 1 (:288-289)
            for (int j = 0; j < p.n_targets; j++) {
                const int d = m.cells[j] - cell0;
#pragma unroll
                for (int q = 0; q < 4; q++) pv[q] = (d == q && ((cnts >> (4 * q)) & 0xfu)) ? 1.0f : pv[q];
            }
        }
        any |= cnts;
    }
    if (any) v = make_float4(pv[0], pv[1], pv[2], pv[3]);
    return any != 0;
}

// One wavefront's share of a pass: lattice bitmap + cells of the newly found targets (lane < 16) into `pass`.
template <int N>
__device__ __forceinline__ void map_build_pass(const DevParams &p, int k, const double (&jx)[N], const double (&jy)[N],
                                               unsigned newly, int cell, int lane, MapPassLds &pass) {
    double ax[N], ay[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (k == 1) {
            ax[i] = jx[i];
            ay[i] = jy[i];
        } else {
            const double s = N != 1 ? (double)(i * p.map_size) / (double)(N - 1) : p.L / 2.0;  // flight_env.py:148-187
            switch (p.agent_mode) {
            case 0: ax[i] = s; ay[i] = 0.0; break;
            case 1: ax[i] = s; ay[i] = p.L / 2.0; break;
            case 2: ax[i] = 0.0; ay[i] = s; break;
            default: ax[i] = p.L; ay[i] = s; break;
            }
        }
    }
    build_rowbits<N>(p, ax, ay, lane, pass.rowbits);
    if (lane == 63) pass.rowbits[CS_MAX_MAP + 1] = 0;  // row map_size + 1 is read by wrapping chunks
    if (lane < CS_MAX_TARGETS) {  // cells of the newly found targets
        pass.cells[lane] = ((newly >> lane) & 1u) ? cell : -1;
        if (lane == 0) pass.any_found = newly != 0;
    }
}

template <int N, int ILP, int BLK = MAP_BLOCK>
__device__ __forceinline__ void map_sweep(const DevParams &p, MapPassLds *s_pass, float *obs, int apply, int parity, int b,
                                          int yblk) {
    const MapJob *job = job_ptr(p, parity, b);
    const int flags = apply ? job->flags : 0;
    const bool dirty = flags & FLAG_DIRTY;
    const bool reset_pass = flags & FLAG_RESET_PASS;
    if (!dirty && !reset_pass && !obs) return;
    float4 *m4 = reinterpret_cast<float4 *>(p.prob + (size_t)b * p.cells);
    const int nchunks = p.cells / 4;
    // the map loads do not depend on anything below: issue them first
    const int c_first = yblk * ILP * BLK + threadIdx.x;
    float4 v_in[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) {
        v_in[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_first + k * BLK < nchunks) v_in[k] = m4[c_first + k * BLK];
    }

    if (dirty || reset_pass) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        // wave 0 builds the step's pass and wave 1 the reset-time pass (a one-wave workgroup builds both in turn)
        for (int k = 1; k >= 0; k--) {
            if (wave != (BLK >= 128 ? 1 - k : 0) || !(k == 1 ? dirty : reset_pass)) continue;
            double jx[N], jy[N];
#pragma unroll
            for (int i = 0; i < N; i++) {
                const double2 a = *reinterpret_cast<const double2 *>(job->axy[i]);
                jx[i] = a.x;
                jy[i] = a.y;
            }
            map_build_pass<N>(p, k, jx, jy, k == 0 ? job->newly_reset : job->newly, job->cell[lane & (CS_MAX_TARGETS - 1)], lane,
                              s_pass[k]);
        }
    }
    __syncthreads();  // uniform: dirty / reset_pass are per-workgroup values
    const size_t row_w = (size_t)p.cells + 4;
#pragma unroll
    for (int kc = 0; kc < ILP; kc++) {
        const int c = c_first + kc * BLK;
        if (c >= nchunks) break;
        float4 v = v_in[kc];
        if ((dirty || reset_pass) && map_update_chunk(p, s_pass, dirty, reset_pass, c, v)) m4[c] = v;
        if (obs) {
            const v4f nv = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int a = 0; a < N; a++) {  // write-once stream: keep it out of the caches
#if CS_MAP_NT
                __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(obs + ((size_t)b * N + a) * row_w) + c);
#else
                reinterpret_cast<v4f *>(obs + ((size_t)b * N + a) * row_w)[c] = nv;
#endif
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(MAP_BLOCK) void k_map(DevParams p, float *obs, int apply, int parity) {
    __shared__ MapPassLds s_pass[2];  // [0] reset-time pass at the start positions, [1] the step's pass
    map_sweep<N, MAP_ILP>(p, s_pass, obs, apply, parity, blockIdx.x, blockIdx.y);
}

// The update alone (no observation rows wanted): without the n output copies to hide it, the sweep is bound by the
// per-workgroup prologue (job record -> lattice bitmap -> barrier), so one workgroup per env does the whole map.
#ifndef CS_MAP_UPD_BLOCK
#define CS_MAP_UPD_BLOCK 256
#endif
constexpr int MAP_UPD_BLOCK = CS_MAP_UPD_BLOCK;
constexpr int MAP_UPD_ILP = (CS_MAX_MAP * CS_MAX_MAP / 4 + MAP_UPD_BLOCK - 1) / MAP_UPD_BLOCK;
template <int N>
__global__ __launch_bounds__(MAP_UPD_BLOCK) void k_map_update(DevParams p, int parity) {
    __shared__ MapPassLds s_pass[2];
    map_sweep<N, MAP_UPD_ILP, MAP_UPD_BLOCK>(p, s_pass, nullptr, 1, parity, blockIdx.x, 0);
}

// flight rollouts: the map sweep of step t and the kinematics / detection of step t + 1 in ONE launch.  The two do not
// depend on each other (the sweep reads step t's MapJob record, the step writes the other one), the sweep is bandwidth
// bound and the step latency bound, so the step's workgroups (lowest indices: dispatched first) ride inside the sweep's
// shadow instead of costing a serial ~10 us of their own.  The step's registers cap the occupancy at four workgroups
// per CU (at the price of a 12-byte spill in the step role), so each sweep thread keeps PIPE_ILP float4 loads in flight
// (measured: the sweep alone loses nothing at that occupancy, profiles/r02_flight_pipe.md).
#ifndef CS_PIPE_ILP
#define CS_PIPE_ILP 3
#endif
#ifndef CS_PIPE_WAVES
#define CS_PIPE_WAVES 4   // wavefronts per SIMD the register budget must allow (<= 128 VGPRs): four workgroups per CU
#endif
// Larger teams get a larger register budget instead of spills: at four wavefronts per SIMD (128 VGPRs) the step role of teams of 4..8
// spilled 105..473 VGPRs; with three (168) teams of 4 and 5 spill nothing, with two (256) neither do teams of 6..8.  Measured, flight
// B = 8192, us per step of cs_rollout: 5 agents 107.0 -> 100.2, 8 agents 179.7 -> 169.8 (three) -> 160.3 (two).
constexpr int pipe_waves(int n) { return n <= 3 ? CS_PIPE_WAVES : (n <= 5 ? 3 : 2); }
constexpr int PIPE_ILP = CS_PIPE_ILP;
template <int N>
__global__ __launch_bounds__(BLOCK, pipe_waves(N)) void k_flight_pipe(DevParams p, StepIO io, float *map_obs, int map_parity,
                                                                      int nstep, int stride, int ysplit) {
    static_assert(BLOCK == MAP_BLOCK, "one workgroup shape for both roles");
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    __shared__ WaveTile tiles[BLOCK / 64];
    __shared__ MapPassLds s_pass[2];
    // every stride-th workgroup steps 16 envs, the others sweep: spread out, the (long-lived) step workgroups never hold
    // more than a small share of a CU's slots
    const int blk = blockIdx.x;
    const int q = blk / stride, r = blk - q * stride;
    if (r == 0 && q < nstep) {
        step_block<N, 1, false, false>(p, io, T, tiles, q);
    } else {
        const int before = q + 1 < nstep ? q + 1 : nstep;   // step workgroups with a lower index
        const int m = blk - before;
        map_sweep<N, PIPE_ILP>(p, s_pass, map_obs, 1, map_parity, m / ysplit, m % ysplit);
    }
}

// ---------------------------------------------------------------------------------------------------------
__global__ void k_seed(DevParams p, const uint32_t *seeds) {
    // np.random.seed(s): init_genrand; the circular form starts at cursor 0 over the seed array
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    unsigned *mt = p.mt + (size_t)b * MT_STRIDE;
    unsigned x = seeds[b];
    mt[0] = x;
    for (int i = 1; i < MT_N; i++) {
        x = 1812433253u * (x ^ (x >> 30)) + (unsigned)i;
        mt[i] = x;
    }
    for (int i = 0; i < MT_PAD; i++) mt[MT_N + i] = mt[i];
    int *hdr = p.hdr + (size_t)b * CS_H_WORDS;
    hdr[CS_H_MT_POS] = 0;
    hdr[CS_H_WORDS_LO] = 0;
    hdr[CS_H_WORDS_HI] = 0;
    p.ahead[b] = 0;
}

// MT19937 pre-pass of the lane-per-env rollout: one wavefront per env twists the WHOLE row ahead of the cursor
// (ahead -> 624) for every env that has fewer than `min_ahead` twisted words left, and writes the env's HIT TAPE: the
// detection pass only ever asks of a draw whether `rand() <= detect_prob`, so the 312 draws of a row boil down to 312
// bits, which the lane kernel keeps in ten registers -- its loop loads no MT19937 word and tempers nothing (resets, which
// need the uniforms themselves, read the twisted words).  Row in, row out, fully coalesced:
// 2.5 KB read + the regenerated words written, against 3 loads + 1 store per 64 words for the in-kernel refill, and the
// rollout's steady-state loop then never waits for a refill (which stays as the fallback for envs that draw more than
// a row's worth inside one chunk).  Super-batches of 192 words: word j needs stored words j, j+1, j+397, none of which
// another word of the same super-batch writes (192 <= 227); within a wavefront LDS operations complete in order.
__global__ __launch_bounds__(256) void k_mt_advance(DevParams p, int min_ahead) {
    __shared__ unsigned rows[4][MT_N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= p.B) return;
    int a = p.ahead[b];
    if (a >= min_ahead || a >= MT_N) return;   // wave-uniform
    unsigned *m = p.mt + (size_t)b * MT_STRIDE;
    unsigned *row = rows[wave];
    const int pos = p.hdr[(size_t)b * CS_H_WORDS + CS_H_MT_POS];
    for (int i = lane; i < MT_N; i += 64) row[i] = m[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    row_twist_ahead(row, m, pos, a, lane);
    // hit tape: bit r = "the draw made of stream words 2r, 2r + 1 from the cursor hits" for all 312 slots of the row
    unsigned *tp = p.tape + (size_t)b * TAPE_STRIDE;
#pragma unroll
    for (int it = 0; it < TAPE_DW / 2; it++) {
        const unsigned long long bm = row_slot_hits(p, row, pos, it, lane);
        if (lane == 0) *reinterpret_cast<U2 *>(tp + 2 * it) = U2{(unsigned)(bm & 0xffffffffull), (unsigned)(bm >> 32)};
    }
    if (lane == 0) {
        const int *h = p.hdr + (size_t)b * CS_H_WORDS;
        *reinterpret_cast<U2 *>(tp + 10) = U2{(unsigned)h[CS_H_WORDS_LO], (unsigned)h[CS_H_WORDS_HI]};
        *reinterpret_cast<U2 *>(tp + 12) = U2{(unsigned)(p.detect_K & 0xffffffffull), (unsigned)(p.detect_K >> 32)};
        p.ahead[b] = MT_N;
    }
}

// One wavefront per env: the env's row in canonical form (MT_CANON words twisted ahead of the cursor), state untouched.
__global__ __launch_bounds__(64) void k_mt_canonical(DevParams p, unsigned *out) {
    __shared__ unsigned row[MT_N];
    const int b = blockIdx.x, lane = threadIdx.x;
    const unsigned *m = p.mt + (size_t)b * MT_STRIDE;
    for (int i = lane; i < MT_N; i += 64) row[i] = m[i];
    __syncthreads();
    const int pos = p.hdr[(size_t)b * CS_H_WORDS + CS_H_MT_POS];
    int a = p.ahead[b];
    while (a < MT_CANON) {  // block-uniform; 64 <= 227 words per round are independent of each other
        const int r = MT_CANON - a < 64 ? MT_CANON - a : 64;
        const int j = wrap624(wrap624(pos + a) + lane);
        unsigned nw = 0;
        if (lane < r) nw = mt_mix(row[j], row[wrap624(j + 1)], row[wrap624(j + MT_M)]);
        __syncthreads();
        if (lane < r) row[j] = nw;
        __syncthreads();
        a += r;
    }
    unsigned *o = out + (size_t)b * MT_STRIDE;
    for (int i = lane; i < MT_STRIDE; i += 64) o[i] = i < MT_N ? row[i] : (i - MT_N < MT_PAD ? row[i - MT_N] : 0u);
}

__global__ void k_fill_prob(float *prob, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) reinterpret_cast<float4 *>(prob)[i] = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
}

// One step of the exploration schedule for per-step callers (cs_epsilon_step): the envs the next cs_step(flags) will execute
// anneal, the others keep their value; `trace` receives what this step's selection used.
__global__ void k_eps_step(DevParams p, int flags, double *eps, double anneal, double min_eps, double *trace) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    const int *h = p.hdr + (size_t)b * CS_H_WORDS;
    const bool done = h[CS_H_TARGET_FIND] >= p.n_targets || h[CS_H_TIME_STEP] >= p.time_limit;
    const bool executed = !(done && !(flags & CS_AUTO_RESET) && (flags & CS_FREEZE_DONE));
    const double v = eps[b];
    if (trace) trace[b] = v;
    if (executed) eps[b] = v > min_eps ? v - anneal : v;   // common/rollout.py:75-76
}

// CS_CHECK_ACTIONS: every action of a call must index dyaw = [0, pi/18, -pi/18] (flight_env_easy.py:259-262).  The offender with the
// LOWEST flat index -- the one the reference's sequential loops would raise on -- is reported through one host-mapped 64-bit word
// (atomic minimum over all offenders; ~0 = none).
__global__ void k_check_actions(const void *actions, unsigned long long count, int i64, int n_actions, unsigned long long *first_bad) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const long long v = i64 ? static_cast<const long long *>(actions)[i] : (long long)static_cast<const int *>(actions)[i];
        if (v < 0 || v >= n_actions) {
            atomicMin_system(first_bad, i);
            return;   // this thread's later indices are all higher
        }
    }
}

__global__ void k_metrics(DevParams p, double *out4) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (b < p.B) {
        const int *hdr = p.hdr + (size_t)b * CS_H_WORDS;
        v0 = (double)hdr[CS_H_TOTAL_REWARD];
        v1 = (hdr[CS_H_FLAGS] & FLAG_WIN) ? 1.0 : 0.0;
        v2 = (double)hdr[CS_H_TARGET_FIND];
        v3 = 1.0;
    }
    for (int off = 32; off > 0; off >>= 1) {  // wave64 shuffle reduction (sums of small integers: exact)
        v0 += __shfl_down(v0, off, 64);
        v1 += __shfl_down(v1, off, 64);
        v2 += __shfl_down(v2, off, 64);
        v3 += __shfl_down(v3, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out4 + 0, v0);
        atomicAdd(out4 + 1, v1);
        atomicAdd(out4 + 2, v2);
        atomicAdd(out4 + 3, v3);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int check_config(const cs_config *c) {
    if (!c) return fail(CS_E_CONFIG, "null config");
    if (c->variant != 0 && c->variant != 1) return fail(CS_E_CONFIG, "variant must be 0 (flight_easy) or 1 (flight)");
    if (c->n_agents < 1 || c->n_agents > CS_MAX_AGENTS) return fail(CS_E_CONFIG, "n_agents must be 1..8");
    if (c->n_targets < 1 || c->n_targets > CS_MAX_TARGETS) return fail(CS_E_CONFIG, "n_targets must be 1..16");
    if (c->map_size < 2 || c->map_size > CS_MAX_MAP) return fail(CS_E_CONFIG, "map_size must be 2..64");
    if (c->variant == 1 && (c->map_size * c->map_size) % 4 != 0) return fail(CS_E_CONFIG, "flight needs map_size^2 % 4 == 0");
    if (c->variant == 1 && c->map_size > 62) return fail(CS_E_CONFIG, "flight needs map_size <= 62 (one u64 per lattice row)");
    if (c->agent_mode < 0 || c->agent_mode > 3) return fail(CS_E_CONFIG, "No such agent mode");
    if (c->target_mode < 0 || c->target_mode > 1) return fail(CS_E_CONFIG, "No such target mode");
    if (c->time_limit < 1) return fail(CS_E_CONFIG, "time_limit must be positive");
    if (!(c->detect_prob >= 0.0 && c->detect_prob <= 1.0)) return fail(CS_E_CONFIG, "detect_prob must be in [0,1]");
    if (c->batch < 1 || c->batch > (1ll << 27)) return fail(CS_E_CONFIG, "batch must be 1..2^27");
    return CS_OK;
}

int make_params(const cs_config *c, void *state, DevParams *p) {
    int rc = check_config(c);
    if (rc) return rc;
    if (!state) return fail(CS_E_ARG, "null state");
    cs_layout lay;
    cs_state_layout(c, &lay);
    memset(p, 0, sizeof(*p));
    p->B = (int)c->batch;
    p->n_targets = c->n_targets;
    p->map_size = c->map_size;
    p->cells = c->map_size * c->map_size;
    p->time_limit = c->time_limit;
    p->agent_mode = c->agent_mode;
    p->target_mode = c->target_mode;
    p->variant = c->variant;
    p->velocity = c->velocity;
    p->force_k = c->safe_dist * c->force_factor * c->velocity;  // safe_dist*POTENTIAL_FORCE_FACTOR*velocity, left to right
    p->force_d2 = c->force_dist * c->force_dist;
    p->view_r2 = (double)(c->view_range * c->view_range);
    p->L = (double)c->map_size;
    p->mid = 0.5 * p->L;
    p->inv_half = 1.0 / (p->L / 2.0);
    p->q = 1.0 - c->detect_prob;
    p->detect_K = c->detect_prob >= 1.0 ? (1ull << 53) : (unsigned long long)floor(c->detect_prob * 9007199254740992.0);
    const double a = (double)c->map_size / 10.0;  // a = self.map_size/10
    for (int j = 0; j < c->n_targets; j++) {
        p->tx0[j] = a * c->cx[j];
        p->ty0[j] = a * c->cy[j];
        p->jx2[j] = (a * c->dx[j]) * 2.0;
        p->jy2[j] = (a * c->dy[j]) * 2.0;
        if (c->deter[j]) p->deter_mask |= 1u << j;
    }
    char *base = (char *)state;
    p->tgt = (double *)(base + lay.tgt_off);
    p->agent = (double *)(base + lay.agent_off);
    p->hdr = (int *)(base + lay.hdr_off);
    p->mt = (unsigned *)(base + lay.mt_off);
    p->ahead = (int *)(base + lay.ahead_off);
    p->tape = (unsigned *)(base + lay.tape_off);
    // lane kernel's fp32 pre-filter of `d2 <= view_range**2` in get_state's normalised coordinates: |fp32 d2 - exact| <=
    // 4.5e-7 sqrt(thr) + 1.2e-7 thr near the threshold; pairs inside +-eps take the fp64 test
    {
        const double thr = p->view_r2 * p->inv_half * p->inv_half;
        p->thr32 = (float)thr;
        p->eps32 = (float)(1e-6 + 4e-6 * thr);
    }
    for (int i = 0; i < c->n_agents; i++) {   // flight_env_easy.py:139-180, the arithmetic of start_pose() (IEEE: same bits)
        const double sp = c->n_agents != 1 ? (double)(i * c->map_size) / (double)(c->n_agents - 1) : p->L / 2.0;
        switch (c->agent_mode) {
        case 0: p->start_x[i] = sp; p->start_y[i] = 0.0; p->start_yaw = 3.141592653589793 / 2.0; break;
        case 1: p->start_x[i] = sp; p->start_y[i] = p->L / 2.0; p->start_yaw = 3.141592653589793 / 2.0; break;
        case 2: p->start_x[i] = 0.0; p->start_y[i] = sp; p->start_yaw = 0.0; break;
        default: p->start_x[i] = p->L; p->start_y[i] = sp; p->start_yaw = 3.141592653589793; break;
        }
    }
    p->prob = (float *)(base + lay.prob_off);
    p->job = base + lay.job_off;
    p->obs_row_w = c->variant == 1 ? p->cells + 4 : 4;
    p->obs_feat_off = c->variant == 1 ? p->cells : 0;
    return CS_OK;
}

// CS_CHECK_ACTIONS (debug aid; -DCS_CHECK_ACTIONS_ALWAYS turns it on for every call of a debug build): the actions of a cs_step /
// cs_rollout call are validated on the device BEFORE anything is stepped; the call then returns CS_E_ARG with the reference's
// IndexError wording and the env state untouched.  The kernels themselves treat any value other than 1 / 2 as 0 (no bounds
// check in the hot loops); the reference raises at dyaw[act] (flight_env_easy.py:262).  Costs a stream synchronisation: not for
// stream capture, not for the production loop.
int check_actions(const void *actions_dev, size_t count, int flags, int n_agents, size_t B, hipStream_t s) {
    static std::mutex mu;
    static unsigned long long *report[64] = {};   // one host-mapped word per device, visible to every device (portable)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    // a captured call cannot synchronise, and asking about a stream while ANOTHER one is in a global-mode capture is an error that
    // can invalidate that capture: in both cases the call goes unchecked
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return CS_OK;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(CS_E_LAUNCH, "CS_CHECK_ACTIONS: no current device");
    std::lock_guard<std::mutex> lock(mu);
    if (!report[dev]) {
        if (hipHostMalloc((void **)&report[dev], sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
            report[dev] = nullptr;
            (void)hipGetLastError();
            return fail(CS_E_LAUNCH, "CS_CHECK_ACTIONS: cannot allocate the report word");
        }
    }
    volatile unsigned long long *rep = report[dev];
    *rep = ~0ull;
    unsigned long long *report_dev = nullptr;
    if (hipHostGetDevicePointer((void **)&report_dev, report[dev], 0) != hipSuccess) return fail(CS_E_LAUNCH, "CS_CHECK_ACTIONS: no device view of the report word");
    const unsigned blocks = (unsigned)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_check_actions, dim3(blocks ? blocks : 1), dim3(256), 0, s, actions_dev, (unsigned long long)count,
                       (flags & CS_ACTIONS_I64) ? 1 : 0, 3, report_dev);
    if (hipStreamSynchronize(s) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "CS_CHECK_ACTIONS: %s", hipGetErrorString(hipGetLastError()));
        return CS_E_LAUNCH;
    }
    const unsigned long long i = *rep;
    if (i != ~0ull) {
        long long v = 0;
        int v32 = 0;
        const bool wide = (flags & CS_ACTIONS_I64) != 0;
        const hipError_t e = wide ? hipMemcpy(&v, static_cast<const long long *>(actions_dev) + i, sizeof(v), hipMemcpyDeviceToHost)
                                  : hipMemcpy(&v32, static_cast<const int *>(actions_dev) + i, sizeof(v32), hipMemcpyDeviceToHost);
        if (e != hipSuccess) (void)hipGetLastError();
        if (!wide) v = v32;
        const unsigned long long per_step = (unsigned long long)B * (unsigned long long)n_agents;
        snprintf(g_err, sizeof(g_err), "list index out of range: action %lld of step %llu, env %llu, agent %llu is not in 0..2 "
                 "(dyaw[act], flight_env_easy.py:262; the batched path takes no negative indices)", v,
                 i / per_step, (i % per_step) / (unsigned long long)n_agents, i % (unsigned long long)n_agents);
        return CS_E_ARG;
    }
    return CS_OK;
}
#ifdef CS_CHECK_ACTIONS_ALWAYS
constexpr int CHECK_ACTIONS_FORCED = CS_CHECK_ACTIONS;
#else
constexpr int CHECK_ACTIONS_FORCED = 0;
#endif

int launched(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return CS_E_LAUNCH;
    }
    return CS_OK;
}

inline dim3 map_grid(const DevParams &p) {
    return dim3((unsigned)p.B, (unsigned)((p.cells / 4 + MAP_ILP * MAP_BLOCK - 1) / (MAP_ILP * MAP_BLOCK)));
}

// Lane-per-env launch(es) of one chunk: a VEC launch over the full wavefronts when every step's block of get_state
// rows is 16-byte aligned, a plain launch for the remaining < 64 envs (or for everything otherwise).
template <int N>
void launch_lane(const cs_config *cfg, const DevParams &p, StepIO io, size_t smem, hipStream_t s) {
    const size_t W = 4 * (size_t)cfg->n_agents + 3 * (size_t)cfg->n_targets;
    const bool aligned = io.state && (reinterpret_cast<size_t>(io.state) & 15) == 0 && ((size_t)p.B * W) % 4 == 0;
    const int full = aligned ? (p.B / 64) * 64 : 0;
    if (full > 0) {
        io.env0 = 0;
        io.env_n = full;
        hipLaunchKernelGGL((k_rollout_lane<N, true>), dim3((unsigned)((full + BLOCK - 1) / BLOCK)), dim3(BLOCK), smem, s, p, io);
    }
    if (p.B - full > 0) {
        io.env0 = full;
        io.env_n = p.B - full;
        hipLaunchKernelGGL((k_rollout_lane<N, false>), dim3((unsigned)((p.B - full + BLOCK - 1) / BLOCK)), dim3(BLOCK), smem, s, p, io);
    }
}
inline size_t lane_smem(const cs_config *c) {
    const size_t W = 4 * (size_t)c->n_agents + 3 * (size_t)c->n_targets;
    return ((TRIG_ROWS * TRIG_COLS * 8 + 15) / 16) * 16 + (BLOCK / 64) * 64 * W * sizeof(float) +
           (BLOCK / 64) * MT_N * sizeof(unsigned)   // + one MT19937 row per wavefront (in-loop refresh)
#ifdef CS_LANE_PADLDS   /* experiment: extra LDS per workgroup, i.e. fewer wavefronts per SIMD (occupancy sensitivity) */
           + CS_LANE_PADLDS
#endif
        ;
}
#ifndef CS_LV_W_FROM
#define CS_LV_W_FROM 524288   /* envs from which teams of up to 3 take the three-wavefronts-per-SIMD build of k_rollout_lanev (round 5, A/B on one box, two passes, % of the HBM roofline: 2^18 envs 44.6 / 44.0 two / three wavefronts, 2^19 46.5 / 49.9, 2^20 46.8 / 51.3, 2^22 50.2 / 54.5) */
#endif
// k_rollout_lanev launch(es): a VEC launch over the full wavefronts when obs and state are both written and every step's
// block of get_state rows is 16-byte aligned, a plain launch for the remaining < 64 envs (or for everything otherwise).
template <int N>
void launch_lanev(const cs_config *cfg, const DevParams &p, StepIO io, hipStream_t s) {
    const size_t W = 4 * (size_t)cfg->n_agents + 3 * (size_t)cfg->n_targets;
    const size_t smem = LV_HEAD_BYTES + (LV_BLOCK / 64) * lv_wave_bytes((int)W, cfg->n_agents);
    const bool aligned = io.state && io.obs && (reinterpret_cast<size_t>(io.state) & 15) == 0 &&
                         (reinterpret_cast<size_t>(io.obs) & 15) == 0 && ((size_t)p.B * W) % 4 == 0;
    const int full = aligned ? (p.B / 64) * 64 : 0;
    if (full > 0) {
        io.env0 = 0;
        io.env_n = full;
        if constexpr (N <= 3) {
            if (full >= CS_LV_W_FROM)   // three wavefronts per SIMD (see k_rollout_lanev)
                hipLaunchKernelGGL((k_rollout_lanev<N, true, 3>), dim3((unsigned)((full + LV_BLOCK - 1) / LV_BLOCK)), dim3(LV_BLOCK), smem, s, p, io);
            else
                hipLaunchKernelGGL((k_rollout_lanev<N, true>), dim3((unsigned)((full + LV_BLOCK - 1) / LV_BLOCK)), dim3(LV_BLOCK), smem, s, p, io);
        } else {
            hipLaunchKernelGGL((k_rollout_lanev<N, true>), dim3((unsigned)((full + LV_BLOCK - 1) / LV_BLOCK)), dim3(LV_BLOCK), smem, s, p, io);
        }
    }
    if (p.B - full > 0) {
        io.env0 = full;
        io.env_n = p.B - full;
        hipLaunchKernelGGL((k_rollout_lanev<N, false>), dim3((unsigned)((p.B - full + LV_BLOCK - 1) / LV_BLOCK)), dim3(LV_BLOCK), smem, s, p, io);
    }
}
// Which lane-per-env kernel: k_rollout_lanev for teams of up to 5 (its in-loop MT19937 refresh tops up one env per wavefront
// and step, which covers the draw rate of those teams), k_rollout_lane (+ the k_mt_advance pre-pass) for larger ones.
inline bool use_lanev(const cs_config *c, int flags) {
    if (c->n_agents > CS_LANE_REFRESH_MAX_N) return false;
    if (flags & CS_KERNEL_LANEV) return true;
    if (flags & CS_KERNEL_LANE) return false;
    return CS_LANEV_DEFAULT != 0;
}
// Octet launch(es): a VEC launch over the full wavefronts (8 envs each) when every step's block of get_state rows is
// 16-byte aligned, a plain launch for the remaining < 8 envs (or for everything otherwise).
template <int N>
void launch_oct(const cs_config *cfg, const DevParams &p, StepIO io, hipStream_t s) {
    const size_t W = 4 * (size_t)cfg->n_agents + 3 * (size_t)cfg->n_targets;
    const bool aligned = !io.state || ((reinterpret_cast<size_t>(io.state) & 15) == 0 && ((size_t)p.B * W) % 4 == 0);
    const int full = aligned ? (p.B / OCT_ENVS) * OCT_ENVS : 0;
    constexpr int EPB = (OCT_BLOCK / 64) * OCT_ENVS;   // envs per workgroup
    io.min_ahead = 2 * cfg->n_agents * CS_MAX_TARGETS;  // rows are topped up in place whenever one runs low
    if (full > 0) {
        io.env0 = 0;
        io.env_n = full;
        const dim3 grid((unsigned)((full + EPB - 1) / EPB));
        if (io.obs && io.state) hipLaunchKernelGGL((k_rollout_oct<N, true, true>), grid, dim3(OCT_BLOCK), 0, s, p, io);
        else hipLaunchKernelGGL((k_rollout_oct<N, true, false>), grid, dim3(OCT_BLOCK), 0, s, p, io);
    }
    if (p.B - full > 0) {   // the tail (or an unaligned output tensor): plain stores, runtime checks
        io.env0 = full;
        io.env_n = p.B - full;
        hipLaunchKernelGGL((k_rollout_oct<N, false, false>), dim3((unsigned)((p.B - full + EPB - 1) / EPB)), dim3(OCT_BLOCK), 0, s, p, io);
    }
}
// A second stream per device for launches that must run BESIDE the caller's (launch_od: the tail of the 5-lane packing), with the two
// events of the fork / join.  Created on first use, kept for the life of the process.  The events are shared by every caller on the
// device: a fork .. join sequence is enqueued under side_order() so that two host threads cannot interleave their records and waits.
inline std::mutex &side_order() {
    static std::mutex mu;
    return mu;
}
struct SideStream {
    hipStream_t stream;
    hipEvent_t fork, join;
};
inline SideStream *side_stream() {
    static std::mutex mu;
    static SideStream per_dev[64];
    static bool made[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!made[dev]) {
        SideStream sd{};
        if (hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        per_dev[dev] = sd;
        made[dev] = true;
    }
    return &per_dev[dev];
}
// Octet-pair launch(es): like launch_oct, one workgroup (K + D wavefront) per 8 envs.
template <int N>
void launch_od(const cs_config *cfg, const DevParams &p, StepIO io, hipStream_t s) {
    const size_t W = 4 * (size_t)cfg->n_agents + 3 * (size_t)cfg->n_targets;
    const bool aligned = !io.state || ((reinterpret_cast<size_t>(io.state) & 15) == 0 && ((size_t)p.B * W) % 4 == 0);
    // three wavefronts per 8 envs (K, D and the emitting E) while five such workgroups per CU hold the batch in one round
    const bool e3 = (io.flags & CS_KERNEL_ODE) || (!(io.flags & CS_KERNEL_OD) && p.B <= CS_ODE_UPTO);
    io.min_ahead = 2 * cfg->n_agents * CS_MAX_TARGETS;  // rows are topped up in place whenever one runs low
    // teams of exactly 5 with at most 15 targets, obs and state both written: FIVE lanes per env, twelve envs per workgroup (OctLay<5>)
    if constexpr (N == 5 && CS_OD_PENT != 0) {
        if (aligned && io.obs && io.state && cfg->n_targets <= 15 && p.B >= 12) {
            const int full5 = (p.B / 12) * 12;
            io.env0 = 0;
            io.env_n = full5;
            const dim3 grid((unsigned)(full5 / 12));
            // The last < 12 envs go through the octet kernels -- on a SIDE stream, beside the main launch: one workgroup running T steps
            // takes as long as the whole grid (a step is latency-, not throughput-bound), so queued behind the main launch it would
            // double the call.  fork: side waits for everything queued on s so far; join: s waits for the side launches.
            const int tail = p.B - full5;
            SideStream *sd = tail > 0 ? side_stream() : nullptr;
            std::unique_lock<std::mutex> order(side_order(), std::defer_lock);
            if (tail > 0 && sd) {
                order.lock();   // (until the join below has been enqueued)
                (void)hipEventRecord(sd->fork, s);
                (void)hipStreamWaitEvent(sd->stream, sd->fork, 0);
                StepIO it = io;
                const int t8 = tail >= OCT_ENVS ? OCT_ENVS : 0;   // a full octet: the same variant as the main launch
                if (t8) {
                    it.env0 = full5;
                    it.env_n = t8;
                    if (e3) hipLaunchKernelGGL((k_rollout_od<N, true, true, true>), dim3(1), dim3(OD_BLOCK + 64), 0, sd->stream, p, it);
                    else hipLaunchKernelGGL((k_rollout_od<N, true, true, false>), dim3(1), dim3(OD_BLOCK), 0, sd->stream, p, it);
                }
                if (tail - t8 > 0) {
                    it.env0 = full5 + t8;
                    it.env_n = tail - t8;
                    hipLaunchKernelGGL((k_rollout_od<N, false, false, false>), dim3(1), dim3(OD_BLOCK), 0, sd->stream, p, it);
                }
                (void)hipEventRecord(sd->join, sd->stream);
            }
            if (e3) hipLaunchKernelGGL((k_rollout_od5<N, true>), grid, dim3(OD_BLOCK + 64), 0, s, p, io);
            else hipLaunchKernelGGL((k_rollout_od5<N, false>), grid, dim3(OD_BLOCK), 0, s, p, io);
            if (tail > 0 && sd) {
                (void)hipStreamWaitEvent(s, sd->join, 0);
            } else if (tail > 0) {   // no side stream to be had: behind the main launch
                io.env0 = full5;
                io.env_n = tail;
                hipLaunchKernelGGL((k_rollout_od<N, false, false, false>), dim3((unsigned)((tail + OCT_ENVS - 1) / OCT_ENVS)), dim3(OD_BLOCK), 0, s, p, io);
            }
            return;
        }
    }
    const int full = aligned ? (p.B / OCT_ENVS) * OCT_ENVS : 0;
    if (full > 0) {
        io.env0 = 0;
        io.env_n = full;
        const dim3 grid((unsigned)(full / OCT_ENVS));
        if (io.obs && io.state && e3) hipLaunchKernelGGL((k_rollout_od<N, true, true, true>), grid, dim3(OD_BLOCK + 64), 0, s, p, io);
        else if (io.obs && io.state) hipLaunchKernelGGL((k_rollout_od<N, true, true, false>), grid, dim3(OD_BLOCK), 0, s, p, io);
        else hipLaunchKernelGGL((k_rollout_od<N, true, false, false>), grid, dim3(OD_BLOCK), 0, s, p, io);
    }
    if (p.B - full > 0) {   // the tail (or an unaligned output tensor): plain stores, runtime checks
        io.env0 = full;
        io.env_n = p.B - full;
        hipLaunchKernelGGL((k_rollout_od<N, false, false, false>), dim3((unsigned)((p.B - full + OCT_ENVS - 1) / OCT_ENVS)), dim3(OD_BLOCK), 0, s, p, io);
    }
}
// cs_rollout: the first-generation lane kernel's lower bound, by bench.py's protocol (round 3): 3 agents 65536 envs octet 7.8e9 against
// lane 7.3e9, 98304 8.0 / 8.2, 131072 8.2 / 10.4; 5 agents (the 250-VGPR lane variant) 262144 octet 5.6e9 against lane 4.9e9,
// 524288 5.7 / 5.0, 2^20 5.8 / 6.4.
// k_rollout_lanev (teams of up to 5) takes over from the octet kernel at 65536 envs -- one wavefront per SIMD -- (tools/gpu_r4_d.sh:
// 3 agents 32768 envs octet 6.0e9 against lanev 4.4e9, 65536: 7.0 / 8.1, 131072: 7.8 / 11.2; 5 agents 32768: 4.2 / 3.0,
// 65536: 4.8 / 5.7, 131072: 5.3 / 7.8)
#ifndef CS_LANEV_FROM
#define CS_LANEV_FROM 65536
#endif
inline long long lane_from(const cs_config *c) {
    if (CS_LANEV_DEFAULT && c->n_agents <= CS_LANE_REFRESH_MAX_N) return CS_LANEV_FROM;
    return c->n_agents <= 4 ? CS_LANE_FROM : CS_LANE_FROM_LARGE_TEAMS;
}
// Kernel choice for flight_easy: one env per 16-lane group (lowest latency, fills the chip from B = 4096) or one
// env per lane (no replicated arithmetic; wins once the batch gives every SIMD a wavefront anyway).
inline bool use_lane_kernel(const cs_config *c, int flags, bool rollout) {
    if (flags & (CS_KERNEL_LANE | CS_KERNEL_LANEV)) return true;
    if (flags & (CS_KERNEL_GROUP | CS_KERNEL_SOLO | CS_KERNEL_DUO)) return false;   // a forced 16-lane kernel is never replaced by another one
    if (rollout && (flags & (CS_KERNEL_OCT | CS_KERNEL_OD | CS_KERNEL_ODE))) return false;
    // single steps have no octet variant: the lane kernel takes over from the 16-lane step kernel at 32768 envs as before
    return c->batch >= (rollout ? lane_from(c) : 32768);
}
// cs_rollout: the octet kernel (one env per 8 lanes) between the pair kernel's range and the lane kernel's
inline bool use_oct_kernel(const cs_config *c, int flags) {
    if (flags & CS_KERNEL_OCT) return true;
    if (flags & (CS_KERNEL_GROUP | CS_KERNEL_LANE | CS_KERNEL_LANEV | CS_KERNEL_SOLO | CS_KERNEL_DUO | CS_KERNEL_OD | CS_KERNEL_ODE)) return false;
    return c->batch > CS_OCT_FROM && c->batch < lane_from(c);
}
// cs_rollout: the octet PAIR kernel (kinematics wavefront + detection wavefront per 8 envs)
inline bool use_od_kernel(const cs_config *c, int flags) {
    if (flags & (CS_KERNEL_OD | CS_KERNEL_ODE)) return true;
    if (flags & (CS_KERNEL_GROUP | CS_KERNEL_LANE | CS_KERNEL_LANEV | CS_KERNEL_SOLO | CS_KERNEL_DUO | CS_KERNEL_OCT)) return false;
    return c->batch <= CS_OD_UPTO;
}

// 16-lanes-per-env rollout: the kinematics / detection wavefront pair pays while its two wavefronts per four envs still
// find a SIMD each (1024 SIMDs x 2 wave slots at these register counts); teams of 7 and 8 spill in the pair.  Measured
// crossover at 4096 envs (profiles/r02_batch_sweep.md; 3 agents at 4608 envs: pair 3.40 us per step, one-wavefront kernel
// 3.18).
inline bool duo_pays(const cs_config *c) { return c->n_agents <= 6 && c->batch <= 4096; }

// Rows with at least this many twisted words ahead are left alone by the pre-pass of a T-step rollout: enough for the
// typical draw rate (two words per draw, a few draws per step) with a step's worst case in reserve.  Short rollouts
// then advance each row only every few calls instead of touching all of them every time.
inline int prepass_min_ahead(const cs_config *c, int T) {
    const long long want = 2ll * c->n_agents * CS_MAX_TARGETS + 64 + 4ll * c->n_agents * T;
    return (int)(want < MT_N - 64 ? want : MT_N - 64);
}

inline unsigned env_blocks(const DevParams &p) { return (unsigned)(((size_t)p.B * G + BLOCK - 1) / BLOCK); }

#ifdef CS_ONLY_N   // experiments only: instantiate one team size (fast compiles)
#define CS_DISPATCH_N(n, CALL) { constexpr int N = CS_ONLY_N; CALL; }
#else
#define CS_DISPATCH_N(n, CALL)                                   \
    switch (n) {                                                 \
    case 1: { constexpr int N = 1; CALL; } break;                \
    case 2: { constexpr int N = 2; CALL; } break;                \
    case 3: { constexpr int N = 3; CALL; } break;                \
    case 4: { constexpr int N = 4; CALL; } break;                \
    case 5: { constexpr int N = 5; CALL; } break;                \
    case 6: { constexpr int N = 6; CALL; } break;                \
    case 7: { constexpr int N = 7; CALL; } break;                \
    default: { constexpr int N = 8; CALL; } break;               \
    }
#endif

}  // namespace

extern "C" {

int cs_abi_version(void) { return CS_ABI_VERSION; }
#ifndef CS_SOURCE_HASH
#define CS_SOURCE_HASH ""
#endif
const char *cs_source_hash(void) { return CS_SOURCE_HASH; }
int cs_has_legacy_kernels(void) { return CS_LEGACY_KERNELS; }
const char *cs_last_error(void) { return g_err; }

int cs_state_layout(const cs_config *cfg, cs_layout *out) {
    int rc = check_config(cfg);
    if (rc) return rc;
    if (!out) return fail(CS_E_ARG, "null layout");
    const size_t B = (size_t)cfg->batch;
    size_t off = 0;
    out->tgt_off = off;
    off = align_up(off + B * G * 2 * sizeof(double), 256);
    out->agent_off = off;
    off = align_up(off + B * CS_MAX_AGENTS * 4 * sizeof(double), 256);
    out->hdr_off = off;
    off = align_up(off + B * CS_H_WORDS * sizeof(int32_t), 256);
    out->mt_off = off;
    off = align_up(off + B * MT_STRIDE * sizeof(uint32_t), 256);
    out->ahead_off = off;
    off = align_up(off + B * sizeof(int32_t), 256);
    out->tape_off = off;
    off = align_up(off + B * TAPE_STRIDE * sizeof(uint32_t), 256);
    out->prob_off = off;
    if (cfg->variant == 1) off = align_up(off + B * (size_t)cfg->map_size * cfg->map_size * sizeof(float), 256);
    out->job_off = off;
    if (cfg->variant == 1) off += 2 * B * CS_JOB_BYTES;
    out->total_bytes = off;
    return CS_OK;
}

int cs_init(const cs_config *cfg, void *state_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    cs_layout lay;
    cs_state_layout(cfg, &lay);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(state_dev, 0, lay.total_bytes, s) != hipSuccess) return fail(CS_E_LAUNCH, "hipMemsetAsync failed");
    if (cfg->variant == 1) {
        size_t n4 = (size_t)p.B * p.cells / 4;
        unsigned blocks = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(k_fill_prob, dim3(blocks), dim3(256), 0, s, p.prob, n4);
    }
    return launched("cs_init");
}

int cs_seed(const cs_config *cfg, void *state_dev, const uint32_t *seeds_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    if (!seeds_dev) return fail(CS_E_ARG, "null seeds");
    hipLaunchKernelGGL(k_seed, dim3((p.B + 63) / 64), dim3(64), 0, (hipStream_t)stream, p, seeds_dev);
    return launched("cs_seed");
}

int cs_reset(const cs_config *cfg, void *state_dev, const uint8_t *mask_dev, int init, float *obs_dev,
             float *state_out_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    CS_DISPATCH_N(cfg->n_agents,
                  hipLaunchKernelGGL(k_reset<N>, dim3(env_blocks(p)), dim3(BLOCK), 0, s, p, mask_dev, init, obs_dev,
                                     state_out_dev));
    if (cfg->variant == 1) {
        CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_map<N>, map_grid(p), dim3(MAP_BLOCK), 0, s, p, obs_dev, 1, 0));
    }
    return launched("cs_reset");
}

int cs_step(const cs_config *cfg, void *state_dev, const void *actions_dev, int flags, float *reward_dev,
            uint8_t *terminated_dev, uint8_t *win_dev, float *obs_dev, float *state_out_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    if (!actions_dev || !reward_dev || !terminated_dev || !win_dev) return fail(CS_E_ARG, "null step buffer");
    hipStream_t s = (hipStream_t)stream;
    if ((flags | CHECK_ACTIONS_FORCED) & CS_CHECK_ACTIONS) {
        rc = check_actions(actions_dev, (size_t)p.B * cfg->n_agents, flags, cfg->n_agents, (size_t)p.B, s);
        if (rc) return rc;
    }
    StepIO io{actions_dev, reward_dev, terminated_dev, win_dev, obs_dev, state_out_dev, flags, 1};
    // (the lane-per-env kernels store an (env, agent) observation as one 16-byte piece of a coalesced block: same rule as cs_rollout)
    if (cfg->variant == 0 && use_lane_kernel(cfg, flags, false) && obs_dev && (reinterpret_cast<size_t>(obs_dev) & 15) != 0)
        return fail(CS_E_ARG, "obs_dev must be 16-byte aligned");
    if (cfg->variant == 0 && use_lane_kernel(cfg, flags, false)) {
        if (use_lanev(cfg, flags)) {
            CS_DISPATCH_N(cfg->n_agents, launch_lanev<N>(cfg, p, io, s));
        } else {
            CS_DISPATCH_N(cfg->n_agents, launch_lane<N>(cfg, p, io, lane_smem(cfg), s));
        }
    } else if (cfg->variant == 0) {
        CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL((k_step<N, 0>), dim3(env_blocks(p)), dim3(BLOCK), 0, s, p, io));
    } else {
        CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL((k_step<N, 1>), dim3(env_blocks(p)), dim3(BLOCK), 0, s, p, io));
        if (obs_dev) {
            CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_map<N>, map_grid(p), dim3(MAP_BLOCK), 0, s, p, obs_dev, 1, 0));
        } else {
            CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_map_update<N>, dim3((unsigned)p.B), dim3(MAP_UPD_BLOCK), 0, s, p, 0));
        }
    }
    return launched("cs_step");
}

int cs_rollout(const cs_config *cfg, void *state_dev, const void *actions_dev, int T, int flags, float *reward_dev,
               uint8_t *terminated_dev, uint8_t *win_dev, float *obs_dev, float *state_out_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    if (T < 1) return fail(CS_E_ARG, "T must be >= 1");
    if (!actions_dev || !reward_dev || !terminated_dev || !win_dev) return fail(CS_E_ARG, "null rollout buffer");
    // the octet and lane kernels write an (env, agent) observation as ONE 16-byte store in every variant (the state table has a scalar
    // fallback, the observations do not): a caller's slice at a 4- or 8-byte offset is refused, not stored to with misaligned dwordx4
    if (cfg->variant == 0 && obs_dev && (reinterpret_cast<size_t>(obs_dev) & 15) != 0)
        return fail(CS_E_ARG, "obs_dev must be 16-byte aligned");
    if ((flags | CHECK_ACTIONS_FORCED) & CS_CHECK_ACTIONS) {
        rc = check_actions(actions_dev, (size_t)T * p.B * cfg->n_agents, flags, cfg->n_agents, (size_t)p.B, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (cfg->variant == 1) {
        // flight: k_step for step 0, then T - 1 launches of k_flight_pipe (the map sweep of step t beside the kinematics /
        // detection of step t + 1), then k_map for the last step's sweep -- enqueued back to back by this one call, each
        // writing its own [t] slice of the outputs
        hipStream_t s = (hipStream_t)stream;
        const size_t n = (size_t)cfg->n_agents, W = 4 * n + 3 * (size_t)cfg->n_targets, B = (size_t)p.B;
        const size_t obs_w = n * ((size_t)p.cells + 4), act_w = n * ((flags & CS_ACTIONS_I64) ? 8 : 4);
        auto step_io = [&](int t) {
            StepIO it{(const char *)actions_dev + (size_t)t * B * act_w, reward_dev + (size_t)t * B,
                      terminated_dev + (size_t)t * B, win_dev + (size_t)t * B,
                      obs_dev ? obs_dev + (size_t)t * B * obs_w : nullptr,
                      state_out_dev ? state_out_dev + (size_t)t * B * W : nullptr, flags, 1};
            it.job_parity = t & 1;
            return it;
        };
        // step 0, then T - 1 launches that sweep step t's map beside step t + 1, then the last sweep
        const int nstep = (int)env_blocks(p);
        const int ysplit = (p.cells / 4 + PIPE_ILP * MAP_BLOCK - 1) / (PIPE_ILP * MAP_BLOCK);
        // step workgroups spread over the first quarter of the grid
        const int total = nstep + p.B * ysplit;
        const int stride = total / 4 / nstep > 1 ? total / 4 / nstep : 1;
        CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL((k_step<N, 1>), dim3(nstep), dim3(BLOCK), 0, s, p, step_io(0)));
        for (int t = 0; t + 1 < T; t++) {
            const StepIO nx = step_io(t + 1);
            float *map_obs = obs_dev ? obs_dev + (size_t)t * B * obs_w : nullptr;
            CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_flight_pipe<N>, dim3((unsigned)total), dim3(BLOCK), 0, s, p, nx,
                                                            map_obs, t & 1, nstep, stride, ysplit));
        }
        CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_map<N>, map_grid(p), dim3(MAP_BLOCK), 0, s, p,
                                                        obs_dev ? obs_dev + (size_t)(T - 1) * B * obs_w : nullptr, 1, (T - 1) & 1));
        return launched("cs_rollout");
    }
    StepIO io{actions_dev, reward_dev, terminated_dev, win_dev, obs_dev, state_out_dev, flags, T};
    if (use_od_kernel(cfg, flags)) {
        CS_DISPATCH_N(cfg->n_agents, launch_od<N>(cfg, p, io, (hipStream_t)stream));
    } else if (use_oct_kernel(cfg, flags)) {
        CS_DISPATCH_N(cfg->n_agents, launch_oct<N>(cfg, p, io, (hipStream_t)stream));
    } else if (use_lane_kernel(cfg, flags, true) && use_lanev(cfg, flags)) {
        CS_DISPATCH_N(cfg->n_agents, launch_lanev<N>(cfg, p, io, (hipStream_t)stream));   // one launch: rows are refreshed inside the kernel
    } else if (use_lane_kernel(cfg, flags, true)) {
        // LANE_CHUNK steps per launch; before each chunk every env's MT19937 row is twisted fully ahead of its cursor by
        // a coalesced pre-pass, so the rollout loop itself (almost) never has to stop for a refill
        hipStream_t s = (hipStream_t)stream;
        const size_t n = (size_t)cfg->n_agents, W = 4 * n + 3 * (size_t)cfg->n_targets, B = (size_t)p.B;
        const size_t act_w = n * ((flags & CS_ACTIONS_I64) ? 8 : 4);
        // (teams of up to 3 refresh their rows inside the kernel, one env per wavefront and step: one launch, no pre-pass)
        const int chunk = cfg->n_agents <= CS_LANE_REFRESH_MAX_N ? T : LANE_CHUNK;
        for (int t0 = 0; t0 < T; t0 += chunk) {
            const int tc = T - t0 < chunk ? T - t0 : chunk;
            if (tc >= 8 && cfg->n_agents > CS_LANE_REFRESH_MAX_N)
                hipLaunchKernelGGL(k_mt_advance, dim3((unsigned)((p.B + 3) / 4)), dim3(256), 0, s, p, prepass_min_ahead(cfg, tc));
            StepIO it{(const char *)actions_dev + (size_t)t0 * B * act_w, reward_dev + (size_t)t0 * B,
                      terminated_dev + (size_t)t0 * B, win_dev + (size_t)t0 * B,
                      obs_dev ? obs_dev + (size_t)t0 * B * n * 4 : nullptr,
                      state_out_dev ? state_out_dev + (size_t)t0 * B * W : nullptr, flags, tc};
            CS_DISPATCH_N(cfg->n_agents, launch_lane<N>(cfg, p, it, lane_smem(cfg), s));
        }
#if CS_LEGACY_KERNELS
    } else if ((flags & CS_KERNEL_SOLO) || ((flags & CS_KERNEL_DUO) == 0 && !duo_pays(cfg))) {
        io.min_ahead = prepass_min_ahead(cfg, T);   // rows are topped up in the kernels' prologue: no pre-pass launch
        CS_DISPATCH_N(cfg->n_agents,
                      hipLaunchKernelGGL(k_rollout<N>, dim3(env_blocks(p)), dim3(BLOCK), 0, (hipStream_t)stream, p, io));
    } else {
        // the pair tops rows up in place whenever one runs low (D's loop), so the prologue only has to cover one step
        io.min_ahead = 2 * cfg->n_agents * CS_MAX_TARGETS;
        CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_rollout_duo<N>, dim3((unsigned)((p.B + DUO_ENVS - 1) / DUO_ENVS)), dim3(DUO_BLOCK),
                                                        0, (hipStream_t)stream, p, io));
    }
#else
    } else if (flags & (CS_KERNEL_SOLO | CS_KERNEL_DUO)) {
        return fail(CS_E_CONFIG, "k_rollout / k_rollout_duo (the 16-lanes-per-env rollout kernels of rounds 1-2) are not in this build: "
                                 "compile with -DCS_LEGACY_KERNELS=1");
    } else {
        // CS_KERNEL_GROUP without the round-2 rollout kernels: T launches of the 16-lane step kernel, each on its own [t] slice
        hipStream_t s = (hipStream_t)stream;
        const size_t n = (size_t)cfg->n_agents, W = 4 * n + 3 * (size_t)cfg->n_targets, B = (size_t)p.B;
        const size_t act_w = n * ((flags & CS_ACTIONS_I64) ? 8 : 4);
        for (int t = 0; t < T; t++) {
            StepIO it{(const char *)actions_dev + (size_t)t * B * act_w, reward_dev + (size_t)t * B, terminated_dev + (size_t)t * B,
                      win_dev + (size_t)t * B, obs_dev ? obs_dev + (size_t)t * B * n * 4 : nullptr,
                      state_out_dev ? state_out_dev + (size_t)t * B * W : nullptr, flags, 1};
            CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL((k_step<N, 0>), dim3(env_blocks(p)), dim3(BLOCK), 0, s, p, it));
        }
    }
#endif
    return launched("cs_rollout");
}

int cs_rollout_policy(const cs_config *cfg, void *state_dev, const float *packed_dev, float *hidden_dev,
                      const int64_t *last_dev, int T, int flags, const cs_epsilon *eps, uint64_t seed, uint32_t step0,
                      uint64_t row0, int select, int64_t *actions_dev, float *reward_dev, uint8_t *terminated_dev,
                      uint8_t *win_dev, float *obs_dev, float *state_out_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    if (cfg->variant != 0) return fail(CS_E_CONFIG, "cs_rollout_policy: flight_easy only");
    const cs_epsilon greedy = {0.0, 0.0, 0.0, 0, 0, nullptr, nullptr};
    if (!eps) eps = &greedy;
    if (eps->trace_dev && !eps->eps_dev) return fail(CS_E_ARG, "cs_epsilon: trace_dev needs eps_dev");
    if (cfg->n_agents > 5) return fail(CS_E_CONFIG, "cs_rollout_policy: at most 5 agents");
    if (T < 1) return fail(CS_E_ARG, "T must be >= 1");
    if (!packed_dev || !hidden_dev || !last_dev || !actions_dev || !reward_dev || !terminated_dev || !win_dev)
        return fail(CS_E_ARG, "null rollout buffer");
    StepIO io{nullptr, reward_dev, terminated_dev, win_dev, obs_dev, state_out_dev, flags, T};
    io.min_ahead = prepass_min_ahead(cfg, T);   // rows with fewer twisted words are topped up in the kernel's prologue
    PolicyIO pio{packed_dev, hidden_dev, last_dev, actions_dev, eps->epsilon, eps->anneal, eps->min_epsilon, eps->per_step,
                 eps->eps_dev, eps->trace_dev, seed, step0, row0, select};
#if CS_POLICY_F16
#define CS_RP_LDS(NN) ((size_t)16 * (NN) * (2 * HXS * 2 + 6 * HST * 2 + LDW * 4))   /* x, b, hs[2] plane pairs of halves + s_h fp32, per row */
#else
#define CS_RP_LDS(NN) ((size_t)3 * 16 * (NN) * LDW * 4)
#endif
    const size_t lds = CS_RP_LDS(cfg->n_agents);
#define CS_LAUNCH_RP(NN)                                                                                               \
    case NN: {                                                                                                         \
        static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_rollout_policy<NN>),           \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,                     \
                                                       (int)CS_RP_LDS(NN)) == hipSuccess;                              \
        if (!lds_ok) return fail(CS_E_LAUNCH, "cs_rollout_policy: cannot reserve the LDS tile");                       \
        hipLaunchKernelGGL(k_rollout_policy<NN>, dim3(env_blocks(p)), dim3(BLOCK), lds, (hipStream_t)stream, p, io, pio); \
    } break;
    switch (cfg->n_agents) {
        CS_LAUNCH_RP(1) CS_LAUNCH_RP(2) CS_LAUNCH_RP(3) CS_LAUNCH_RP(4) CS_LAUNCH_RP(5)
    }
#undef CS_LAUNCH_RP
    return launched("cs_rollout_policy");
}

int cs_rollout_policy_flight(const cs_config *cfg, void *state_dev, const float *packed_dev, const float *conv1_w_dev,
                             const float *conv1_b_dev, const float *conv2_w_dev, const float *conv2_b_dev,
                             const float *lin_w_dev, const float *lin_b_dev, float *hidden_dev, const int64_t *last_dev,
                             float *scratch_dev, int T, int flags, const cs_epsilon *eps, uint64_t seed, uint32_t step0, uint64_t row0,
                             int select, int64_t *actions_dev, float *reward_dev, uint8_t *terminated_dev, uint8_t *win_dev,
                             float *obs_dev, float *state_out_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    const cs_epsilon greedy = {0.0, 0.0, 0.0, 0, 0, nullptr, nullptr};
    if (!eps) eps = &greedy;
    if (eps->trace_dev && !eps->eps_dev) return fail(CS_E_ARG, "cs_epsilon: trace_dev needs eps_dev");
    if (cfg->variant != 1) return fail(CS_E_CONFIG, "cs_rollout_policy_flight: flight only");
    if (cfg->map_size != 50) return fail(CS_E_CONFIG, "cs_rollout_policy_flight: the conv front end is built for map_size 50");
    if (T < 1) return fail(CS_E_ARG, "T must be >= 1");
    if (!packed_dev || !conv1_w_dev || !conv1_b_dev || !conv2_w_dev || !conv2_b_dev || !lin_w_dev || !lin_b_dev ||
        !hidden_dev || !last_dev || !scratch_dev || !actions_dev || !reward_dev || !terminated_dev || !win_dev)
        return fail(CS_E_ARG, "null rollout buffer");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)cfg->n_agents, B = (size_t)p.B, W = 4 * n + 3 * (size_t)cfg->n_targets;
    const size_t obs_w = n * ((size_t)p.cells + 4);
    constexpr int NA = 3;   // the env has three actions (flight_env.py:32)
    float *feat = scratch_dev, *tails = scratch_dev + B * 16;
    // the agents' own 4 floats, compact ([B][n][4]): what the first network call reads and, when no observation rows
    // are wanted, where every step leaves them
    DevParams pc = p;
    pc.obs_row_w = 4;
    pc.obs_feat_off = 0;
    CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_emit<N>, dim3(env_blocks(p)), dim3(BLOCK), 0, s, pc, tails, nullptr));
    for (int t = 0; t < T; t++) {
        // the conv front end reads each env's map where it lives: one read per env, no observation copy needed
        if (cs_policy_conv_features(conv1_w_dev, conv1_b_dev, conv2_w_dev, conv2_b_dev, lin_w_dev, lin_b_dev, p.prob,
                                    (int64_t)p.cells, p.B, feat, stream) != CS_OK)
            return fail(CS_E_LAUNCH, cs_policy_last_error());
        const bool prev_rows = obs_dev && t > 0;   // own floats of step t - 1: in its observation rows, or compact
        const float *own = prev_rows ? obs_dev + (size_t)(t - 1) * B * obs_w : tails;
        int64_t *act = actions_dev + (size_t)t * B * n;
        if (cs_policy_forward(packed_dev, own, prev_rows ? p.cells + 4 : 4, prev_rows ? p.cells : 0,
                              t == 0 ? last_dev : act - B * n, feat, cfg->n_agents, hidden_dev, nullptr, act, (int)(B * n),
                              cfg->n_agents, NA, (float)eps->epsilon, eps->eps_dev, seed, step0 + (uint32_t)t, row0, select, stream) != CS_OK)
            return fail(CS_E_LAUNCH, cs_policy_last_error());
        if (eps->eps_dev && (eps->per_step || eps->trace_dev))   // the schedule's step, before the env step it belongs to
            hipLaunchKernelGGL(k_eps_step, dim3((unsigned)((p.B + 255) / 256)), dim3(256), 0, s, p, flags, eps->eps_dev,
                               eps->per_step ? eps->anneal : 0.0, eps->per_step ? eps->min_epsilon : 1.0e300,
                               eps->trace_dev ? eps->trace_dev + (size_t)t * B : nullptr);
        StepIO it{act, reward_dev + (size_t)t * B, terminated_dev + (size_t)t * B, win_dev + (size_t)t * B,
                  obs_dev ? obs_dev + (size_t)t * B * obs_w : tails,
                  state_out_dev ? state_out_dev + (size_t)t * B * W : nullptr, flags | CS_ACTIONS_I64, 1};
        CS_DISPATCH_N(cfg->n_agents,
                      hipLaunchKernelGGL((k_step<N, 1>), dim3(env_blocks(p)), dim3(BLOCK), 0, s, obs_dev ? p : pc, it));
        if (obs_dev) {
            CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_map<N>, map_grid(p), dim3(MAP_BLOCK), 0, s, p,
                                                            obs_dev + (size_t)t * B * obs_w, 1, 0));
        } else {   // the update alone: fusing it into the conv kernel was measured slower
            CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_map_update<N>, dim3((unsigned)p.B), dim3(MAP_UPD_BLOCK), 0, s, p, 0));
        }
    }
    return launched("cs_rollout_policy_flight");
}

int cs_epsilon_step(const cs_config *cfg, void *state_dev, int flags, double *eps_dev, double anneal, double min_epsilon,
                    double *trace_row_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    if (!eps_dev) return fail(CS_E_ARG, "null epsilon buffer");
    hipLaunchKernelGGL(k_eps_step, dim3((unsigned)((p.B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, flags, eps_dev, anneal,
                       min_epsilon, trace_row_dev);
    return launched("cs_epsilon_step");
}

int cs_mt_advance(const cs_config *cfg, void *state_dev, int min_ahead, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mt_advance, dim3((unsigned)((p.B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, min_ahead);
    return launched("cs_mt_advance");
}

int cs_mt_canonical(const cs_config *cfg, void *state_dev, uint32_t *rows_out_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    if (!rows_out_dev) return fail(CS_E_ARG, "null output rows");
    hipLaunchKernelGGL(k_mt_canonical, dim3((unsigned)p.B), dim3(64), 0, (hipStream_t)stream, p, rows_out_dev);
    return launched("cs_mt_canonical");
}

int cs_emit(const cs_config *cfg, void *state_dev, float *obs_dev, float *state_out_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    CS_DISPATCH_N(cfg->n_agents,
                  hipLaunchKernelGGL(k_emit<N>, dim3(env_blocks(p)), dim3(BLOCK), 0, s, p, obs_dev, state_out_dev));
    if (cfg->variant == 1 && obs_dev) {
        CS_DISPATCH_N(cfg->n_agents, hipLaunchKernelGGL(k_map<N>, map_grid(p), dim3(MAP_BLOCK), 0, s, p, obs_dev, 0, 0));
    }
    return launched("cs_emit");
}

int cs_metrics(const cs_config *cfg, void *state_dev, double *out4_dev, void *stream) {
    DevParams p;
    int rc = make_params(cfg, state_dev, &p);
    if (rc) return rc;
    if (!out4_dev) return fail(CS_E_ARG, "null metrics buffer");
    hipLaunchKernelGGL(k_metrics, dim3((p.B + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, out4_dev);
    return launched("cs_metrics");
}

#ifdef CS_REGION_COUNTS
// measurement builds only (tools/spill_exec.py): the region counters of k_rollout_lanev (rollout_lanev.h: LV_COUNT)
int cs_debug_region_counts(unsigned long long *out16_host, int reset) {
    if (out16_host && hipMemcpyFromSymbol(out16_host, HIP_SYMBOL(g_region), 16 * sizeof(unsigned long long)) != hipSuccess)
        return fail(CS_E_LAUNCH, "cs_debug_region_counts: hipMemcpyFromSymbol");
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_region), z, sizeof(z)) != hipSuccess) return fail(CS_E_LAUNCH, "cs_debug_region_counts: reset");
    }
    return CS_OK;
}
#endif
#ifdef CS_TIMELINE
int cs_debug_read_spin(unsigned *host1024x4) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host1024x4, HIP_SYMBOL(g_spin), sizeof(unsigned) * 1024 * 4);
}
int cs_debug_read_blk(unsigned long long *host1024x8) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host1024x8, HIP_SYMBOL(g_blk), sizeof(unsigned long long) * 1024 * 8);
}
int cs_debug_read_stamps(unsigned long long *host64x16) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host64x16, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 16);
}
#endif

}  // extern "C"
