// cooperative-search_amd/csrc/rollout_od.h -- k_rollout_od: the octet layout split by role -- kinematics wavefront K, detection wavefront D and (up to 8192 envs) emitting wavefront E per 8 envs (the default rollout kernel up to 16384 envs: BASELINE configs 2, 3 and 5's shards).
// Included by coopsearch.hip inside its anonymous namespace, after rollout_oct.h (EnvO, oct_kinematics, oct_detect_impl, oct_place_targets).  Not a translation unit of its own.

// =========================================================================================================
// Octet pair (flight_easy): the octet layout split by ROLE -- per 8 envs a kinematics wavefront K, a detection
// wavefront D and (up to 8192 envs) an emitting wavefront E, one such team per workgroup, no barrier in the loops.
//
// In the octet kernel one wavefront walks the whole dependent chain of a step -- kinematics (~1800 cycles for 3 agents),
// then detection + reward + rows (~1500) -- and at the batch sizes where every SIMD holds at most one or two wavefronts
// nothing fills its stalls.  As in round 2's wavefront pair, the kinematics of step s + 1 need nothing from the detection pass of
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
constexpr int OD_BLOCK = 128;
// K divides the two components of a repulsion term with ONE reciprocal (div2_same_denominator: the same quotients bit for bit) where it
// measured faster.  Small teams keep the plain divisions: K is alone on its SIMD there and the range check in front of the shared
// sequence lengthens its chain (c2: -1.9 %, round 4).  The PAIR variant at 5 agents: no difference (round 5: 3.432 -> 3.438 us per step
// at 16384 envs; round 6: 3.173 / 3.204 -> 3.203 / 3.194).  The THREE-WAVEFRONT variant at 5 agents: 1.962 -> 1.935 (round 5), 1.975 /
// 1.981 -> 1.909 / 1.941 us per step at 8192 envs (round 6, profiles/r06_knob_sweep.log): on from teams of CS_ODE_SHARED_DIV_FROM_N.
#ifndef CS_OD_SHARED_DIV_FROM_N
#define CS_OD_SHARED_DIV_FROM_N 99   /* pair variant */
#endif
#ifndef CS_ODE_SHARED_DIV_FROM_N
#define CS_ODE_SHARED_DIV_FROM_N 5   /* three-wavefront variant */
#endif
// steps K may be ahead of D (power of two).  The pair variant serves up to 16384 envs with eight workgroups per CU: 20 KB of LDS each,
// four slots.  The three-wavefront variant stops at 8192 envs = four workgroups per CU, so its ring can be eight deep (30 KB + E's row buffer):
// K absorbs more of D's events before it has to wait for a slot.
constexpr int od_ring(bool e3) { return e3 ? CS_OD_RING_E3 : CS_OD_RING; }
static_assert((od_ring(false) & (od_ring(false) - 1)) == 0 && od_ring(false) >= 2, "ring depth");
static_assert((od_ring(true) & (od_ring(true) - 1)) == 0 && od_ring(true) >= 2, "ring depth");

template <int ENVS, int AP>
struct __attribute__((aligned(16))) OdRingT {   // what K hands to D for one step (ENVS envs per workgroup: 8)
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
    union {   // never live together: the start poses are consumed inside a step's reset block, the floats written after it
        double2 dpos[ENVS][AP];    // D: start poses for the reset-time detection pass
        float2 dnp[ENVS][AP];      // D: the team's normalised fp32 positions of the step (sensor pre-filter, CS_OD_PREFILTER)
    };
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
template <int N, bool VEC, bool EMIT, bool E3, int LG>
__device__ __forceinline__ void rollout_od_body(const DevParams &p, const StepIO &io);
template <int N, bool VEC, bool EMIT, bool E3>
__global__ __launch_bounds__(E3 ? OD_BLOCK + 64 : OD_BLOCK, CS_OD_WAVES) void k_rollout_od(DevParams p, StepIO io) {
    rollout_od_body<N, VEC, EMIT, E3, OG>(p, io);
}
template <int N, bool VEC, bool EMIT, bool E3, int LG>
__device__ __forceinline__ void rollout_od_body(const DevParams &p, const StepIO &io) {
    static_assert(!E3 || (VEC && EMIT), "the emitting wavefront has the full-wavefront, obs + state stores only");
    using Lay = OctLay<LG>;
    constexpr int ENVS = Lay::ENVS;   // envs per workgroup (= per wavefront of each role)
    constexpr int AP = OCT_PAD;   // columns of the per-agent LDS rows
    constexpr int TW = TILE_W;    // widest get_state row
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
            const unsigned out = oct_kinematics<N, (N >= (E3 ? CS_ODE_SHARED_DIV_FROM_N : CS_OD_SHARED_DIV_FROM_N)), LG, AP>(p, T, sh.kpos, o, t, sh8, stepping, a, e, sp);
#endif
            KIN_STAMP_SP(6);
            between();
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
        // (Rounds 5 / 6 tried requesting these words in the middle of the previous step so that the LDS round trip runs beside the publication:
        // round 5 as one ds_read2_b32 split over two asm statements -- a load in flight the register allocator knew nothing of, ADVICE r5 --
        // and round 6 as compiler-visible volatile loads: no gain, c2 1.18 -> 1.20-1.21 us per step, profiles/r06_experiments.md H.  Gone.)
        constexpr int OFF_FIX = (int)(offsetof(OdShared, fix_req) - offsetof(OdShared, d_steps)) / 4;
        constexpr int OFF_E = (int)(offsetof(OdShared, e_steps) - offsetof(OdShared, d_steps)) / 4;
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
            // (K overwrites slot s after ONE read that showed d_steps (e_steps) > s - RING, and that read also returned every fix_req
            // posted before that progress word)
            const int2 pv = E3 ? lds_peek2<OFF_E, OFF_FIX>(&sh.d_steps) : lds_peek2<0, OFF_FIX>(&sh.d_steps);
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
            produce(s, act, true, []() {});
            DUO_STAMP(1);
            OD_JITTER(2);
            post(&sh.k_steps, s + 1);
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
            unsigned long long bms[TAPE_DW / 2];
            row_hits_all(OD_COLD(), rf.erow, pos, lane, bms);
#pragma unroll
            for (int it = 0; it < TAPE_DW / 2; it++) {
                const unsigned long long bm = bms[it];
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
        oct_norm_targets<N, LG>(p, t, e);
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
    int k_seen = 0;                      // K's progress word as D last read it
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
        // (K's progress word only grows, and a slot K published stays published -- a redo rewrites it behind fix_ack, waited for above:
        // the word is re-read only when the last value D saw does not cover step s.  With K a few steps ahead, as it is whenever D is the
        // longer role, D's steady-state step has no LDS round trip in front of the ring read: CS_OD_KSEEN, round 6)
#ifndef CS_OD_KSEEN
#define CS_OD_KSEEN 1
#endif
        if (!CS_OD_KSEEN || k_seen <= s) {
            while ((k_seen = peek(&sh.k_steps)) <= s) { SPIN_TICK; __builtin_amdgcn_s_sleep(1); }
        }
        OD_JITTER(7);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef CS_OD_ABL_NODET   /* experiment: what K alone sustains */
        post(&sh.d_steps, s + 1);
        continue;
#endif
        const OdRing &r = sh.ring[s & (OD_RING - 1)];
        if (live) e.flags = (e.flags & ~0xff00) | (int)(r.out[o] << 8);
        // D decides the sensor tests in packed fp32 first (oct_detect_impl, PRE) -- in the PAIR variant only.  Measured (round 6, one box,
        // two passes, us per step, fp64 only -> pre-filter): pair at 16384 envs 5 agents 3.206 / 3.222 -> 3.189 / 3.210, 3 agents 2.306 /
        // 2.306 -> 2.264 / 2.250 (four wavefronts per SIMD, VALU-issue-bound: 20 instructions fewer per D wavefront-step at 5 agents);
        // three-wavefront variant 8192 envs 5 agents 1.956 / 1.944 -> 2.055 / 2.022, c2 (4096 envs, 3 agents) 1.179 / 1.197 -> 1.302 /
        // 1.281: there D sits on the pipeline's critical path and the floats' way through LDS (write, wave barrier, read: D does not own
        // the tile there) lengthens its chain by ~290 cycles per step.  -DCS_OD_PREFILTER=0 / 2: nowhere / in both variants.
#ifndef CS_OD_PREFILTER
#define CS_OD_PREFILTER 1
#endif
        constexpr bool PRE = CS_OD_PREFILTER == 2 || (CS_OD_PREFILTER == 1 && !E3);
        if ((!E3 || PRE) && ag) {   // the agents' normalised floats: get_state's (the tile is D's without E) and the pre-filter's
            const double2 xy = r.pos[o][t];
            const float nx = (float)((xy.x - p.mid) * p.inv_half), ny = (float)((xy.y - p.mid) * p.inv_half);
            if (!E3) {
                const float2 cs = r.cssn[o][t];
                row[4 * t + 0] = nx;
                row[4 * t + 1] = ny;
                row[4 * t + 2] = cs.x;
                row[4 * t + 3] = cs.y;
            } else {
                sh.dnp[o][t] = make_float2(nx, ny);
            }
        }
        if (PRE) {   // every lane reads the whole team's floats: written by other lanes of this wavefront
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const int reward = oct_detect_impl<N, LG, AP, CS_OD_LAZY_TAPE != 0, PRE>(p, r.pos, o, t, sh8, stepping, e, tape, tcur,
                                                                                 E3 ? &sh.dnp[o][0].x : row, E3 ? 2 : 4);
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
