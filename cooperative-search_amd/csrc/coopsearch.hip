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
static_assert(MT_STRIDE >= 640, "row_load reads ten dword columns of 64 lanes: words 0..639 of the row must exist");
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
// LEAN (the lane kernel's in-loop refresh, which counts no stores): the mirror of words 0..31 is stored only by the lanes that hold one --
// mt_store's unconditional second store doubles the twist's store instructions for the sake of callers whose waits count them.
template <bool LEAN = false>
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
                if (LEAN) {
                    m[idx[c]] = nw[c];
                    if (idx[c] < MT_PAD) m[MT_N + idx[c]] = nw[c];
                } else {
                    mt_store(m, idx[c], nw[c]);
                }
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
    for (int i = 0; i < 10; i++) r.w[i] = m[lane + 64 * i];   // (unconditional: the tenth column reaches words 576..639, inside the row's 672)
}

// row registers -> LDS (`row`: 624 words owned by this wavefront)
__device__ __forceinline__ void row_to_lds(const RowRegs &r, unsigned *row, int lane) {
#pragma unroll
    for (int i = 0; i < 10; i++)
        if (lane + 64 * i < MT_N) row[lane + 64 * i] = r.w[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// All 312 hit bits of a fully twisted row in LDS: bm[it] = hit bits of draw slots 64 it .. 64 it + 63 from the cursor (slot r = words
// pos + 2r, pos + 2r + 1; pos is even, so the pair never straddles the end of the row), arranged for a wavefront that has nothing
// else to overlap the latency with (round 6; the lane kernel refreshes a row every wavefront-step at 5 agents, 2 000 cycles of it in
// this loop: profiles/r06_lanev5_timeline.log): the five word pairs are loaded first (the first version, one 64-slot group per call,
// left an LDS write of the caller between consecutive reads, i.e. five round trips in series), the first word decides in straight-line code, and the
// second word -- needed only when the first one's 27 bits equal the threshold's, 2^-27 of the draws -- sits behind ONE wave-uniform
// test instead of a divergent branch per slot.
__device__ __forceinline__ void row_hits_all(const DevParams &p, const unsigned *row, int pos, int lane, unsigned long long (&bm)[TAPE_DW / 2]) {
    constexpr int NIT = TAPE_DW / 2;
    unsigned wa[NIT], wb[NIT];
    bool valid[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int r = 64 * it + lane;
        valid[it] = 2 * r < MT_N;
        const int i0 = wrap624(pos + (valid[it] ? 2 * r : 0));   // (even: the pair is 8-byte aligned and never straddles the end)
        const U2 w = *reinterpret_cast<const U2 *>(row + i0);
        wa[it] = w.x;
        wb[it] = w.y;
    }
    const unsigned khi = (unsigned)(p.detect_K >> 26);
    bool lt[NIT], eq_any = false;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const unsigned hi = mt_temper(wa[it]) >> 5;   // u = (a >> 5) * 2^26 + (b >> 6) <= K: decided by the first word unless hi == khi
        lt[it] = hi < khi;
        eq_any = eq_any | (valid[it] & (hi == khi));
    }
    if (__builtin_expect(__ballot(eq_any) != 0ull, 0)) {   // wave-uniform, 2^-27 per draw: the exact comparison for everyone
#pragma unroll
        for (int it = 0; it < NIT; it++) lt[it] = draw_hits(p, wa[it], wb[it]);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) bm[it] = __ballot(valid[it] & lt[it]);
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
        unsigned long long bm[TAPE_DW / 2];
        row_hits_all(p, rowbuf, pos, lane, bm);
#pragma unroll
        for (int it = 0; it < TAPE_DW / 2; it++) {
            if (grp == g) {
                tape[2 * it] = (unsigned)(bm[it] & 0xffffffffull);
                tape[2 * it + 1] = (unsigned)(bm[it] >> 32);
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
#define OCT_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && s < 64) g_stamps[s][k] = __builtin_readcyclecounter(); } while (0)
#define KIN_STAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && tl_step >= 0 && tl_step < 64) g_stamps[tl_step][k] = __builtin_readcyclecounter(); } while (0)
#define KIN_STAMP_SP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && sp >= 0 && sp < 64) g_stamps[sp][k] = __builtin_readcyclecounter(); } while (0)
#else
#define KIN_STAMP(k) do {} while (0)
#define KIN_STAMP_SP(k) do {} while (0)
#define OCT_STAMP(k) do {} while (0)
#define DUO_STAMP(k) do {} while (0)
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

#include "rollout_policy.h"

#include "rollout_lane.h"

#include "rollout_oct.h"

#include "rollout_od.h"
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

#include "flight_map.h"

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
    unsigned long long bm[TAPE_DW / 2];
    row_hits_all(p, row, pos, lane, bm);
#pragma unroll
    for (int it = 0; it < TAPE_DW / 2; it++)
        if (lane == 0) *reinterpret_cast<U2 *>(tp + 2 * it) = U2{(unsigned)(bm[it] & 0xffffffffull), (unsigned)(bm[it] >> 32)};
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
// Octet-pair launch(es): like launch_oct, one workgroup (K + D wavefront) per 8 envs.
template <int N>
void launch_od(const cs_config *cfg, const DevParams &p, StepIO io, hipStream_t s) {
    const size_t W = 4 * (size_t)cfg->n_agents + 3 * (size_t)cfg->n_targets;
    const bool aligned = !io.state || ((reinterpret_cast<size_t>(io.state) & 15) == 0 && ((size_t)p.B * W) % 4 == 0);
    // three wavefronts per 8 envs (K, D and the emitting E) while five such workgroups per CU hold the batch in one round
    const bool e3 = (io.flags & CS_KERNEL_ODE) || (!(io.flags & CS_KERNEL_OD) && p.B <= CS_ODE_UPTO);
    io.min_ahead = 2 * cfg->n_agents * CS_MAX_TARGETS;  // rows are topped up in place whenever one runs low
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
int cs_has_legacy_kernels(void) { return 0; }   // (ABI 7 entry point; the kernels it asked about were removed in round 6)
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
    } else if (flags & (CS_KERNEL_SOLO | CS_KERNEL_DUO)) {
        return fail(CS_E_CONFIG, "k_rollout / k_rollout_duo (the 16-lanes-per-env rollout kernels of rounds 1-2) were removed in round 6: "
                                 "CS_KERNEL_GROUP runs a rollout as T launches of the 16-lane step kernel");
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
