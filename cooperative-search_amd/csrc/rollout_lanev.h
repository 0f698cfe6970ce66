// cooperative-search_amd/csrc/rollout_lanev.h -- k_rollout_lanev: the lane-per-env rollout kernel, second generation.
// Included by coopsearch.hip inside its anonymous namespace (after the octet helpers: trig_heading_pair, reset_batch_twisted).
//
// k_rollout_lane (one env per lane, 64 per wavefront) does each env's arithmetic exactly once -- 30 VALU instructions per
// env-step against the octet kernels' 60 -- but sits at 210-250 VGPRs and 16.6 KB of LDS per wavefront (the 64 get_state rows
// of a step staged as one tile), i.e. TWO wavefronts per SIMD, and the SQ counters show each of them waiting 43 % of its
// cycles: the VALU pipe is idle half the time (profiles/r03_lane3_pmc.json).  This kernel keeps the layout and the per-env
// arithmetic (same functions / same expression order: bit-identical results) and was built for three to four wavefronts per
// SIMD.  What it bought is something else (profiles/r04_lanev.md): occupancy turned out NOT to be the lever -- k_rollout_lane<5>
// runs as fast with one wavefront per SIMD as with two, and this kernel capped at 168 VGPRs spills and loses -- but the
// restructured step issues fewer instructions and waits less, at the same two wavefronts per SIMD (5 agents, 2^18 envs: 42.4 ->
// 34.1 us per step).  The design points:
//   * kinematics agent by agent, the reference's own order (flight_env_easy.py:255-301): the two headings of ONE agent
//     (new heading, wall reflection) are evaluated together (trig_heading_pair), then its repulsion, move and wall rule --
//     the 6 n doubles of all headings of a step are never live together (60 VGPRs at 5 agents); the instruction-level
//     parallelism that gave up is supplied by the extra wavefronts;
//   * the targets' normalised fp32 coordinates live in the lane's REGISTERS (2 x 16 floats), not in an LDS tile: the
//     register file of a SIMD holds 512 KB per CU against 160 KB of LDS, so at this occupancy registers are the cheaper
//     place, and the n*m sensor tests read no LDS at all;
//   * get_state rows leave through a HALF-wavefront staging piece (32 rows x W floats, 8.3 KB at 5 agents): the lanes of
//     one half deposit their rows (agents' floats, targets' floats, found flags -- all from registers), all 64 lanes
//     copy the piece out as float4 chunks, non-temporal; then the other half.  The wavefront's MT19937 row buffer of the
//     in-loop refresh ALIASES the piece (the row is twisted between two write-outs);
//   * the refresh row is requested after the kinematics and consumed after the draws of the SAME step (ten dwords per lane
//     live across the sensor tests only, where the register pressure is lowest), not held across a step; the hit tapes sit
//     in LDS with a per-lane cursor instead of ten barrel-shifted registers, and a step walks the MISSES among its draws
//     (0.3 per env-step) instead of the in-range pairs;
//   * resets take the lean path only (one attempt batch from twisted words, LDS tables, scalar kernarg loads, host start
//     poses); everything unusual -- batch not sufficient, words not twisted, a target within view of a start pose (the
//     reset-time detection pass then draws, quirk Q3) -- goes through the generic 16-lane code on a temporary Env<N>.
// Output order inside a step: the next step's actions are requested BEFORE the step's stores and every store of the VEC
// variant is unconditional, so the wait for the actions never waits for a store (one in-order counter for loads and stores).
#pragma once

// Wavefronts per SIMD the register budget must allow.  Measured (profiles/r04_lanev.md): the kernel wants ~210 VGPRs at 3 agents
// and ~240 at 5; capped at 168 (three wavefronts) the 3-agent variant spills 23 registers and runs 40.6 % of the HBM roofline at
// 2^18 envs against 44.0 % uncapped (4096 wavefronts = 1.33 resident rounds instead of 2 full ones) but 49.3 % against 46.0 % at
// 2^20; the 5-agent variant spills 670 and halves.  So: two.
#ifndef CS_LV_WAVES_SMALL
#define CS_LV_WAVES_SMALL 2   /* teams of up to 3 */
#endif
#ifndef CS_LV_WAVES_MID
#define CS_LV_WAVES_MID 2     /* teams of 4 and 5 */
#endif
#ifndef CS_LV_WAVES_LARGE
#define CS_LV_WAVES_LARGE 2   /* ... teams of 6 to 8 */
#endif
// In-loop refresh: a wavefront tops up one env per step, and takes any env with fewer than this many twisted words left (the one
// running lowest first).  A refresh reads and rewrites the whole 2.6 KB row whatever it twists, so the threshold sets the MT19937
// traffic: at 352 some env always qualifies and every env comes round every 64 steps (42 B read + 39 B written per env-step,
// profiles/r05_lanev_traffic.md); lower thresholds refresh by need.
#ifndef CS_LV_NORMAL
#define CS_LV_NORMAL 0   /* NORMAL = LOW + 128.  Measured A/B on one box, two passes (round 5): 3 agents 352 -> 224: 2^18 envs 21.7 -> 21.3 us
                            per step, 65536 envs 8.3 -> 8.05; 5 agents 352 -> 288: no difference */
#endif
#ifndef CS_LV_BLOCK
#define CS_LV_BLOCK 256   /* threads per workgroup of k_rollout_lanev (>= 128: load_trig_to_lds) */
#endif
constexpr int LV_BLOCK = CS_LV_BLOCK;
constexpr int lv_waves(int n) { return n <= 3 ? CS_LV_WAVES_SMALL : (n <= 5 ? CS_LV_WAVES_MID : CS_LV_WAVES_LARGE); }
constexpr int LV_PIECE = 32;           // get_state rows per staging piece: half a wavefront
constexpr int LV_SLOT_FLOATS = 4 * G * 2;   // reset hand-over: four rows of 16 (ntx, nty) pairs
constexpr int LV_TAPE_ROWS = TAPE_DW + 3;   // hit tapes of the wavefront's 64 envs in LDS, [dword][lane]; three rows of zeros behind
                                            // them so that a window of up to 96 + 31 bits never reads past the end

// -DCS_REGION_COUNTS (measurement builds only, tools/spill_exec.py): how often each conditional region of the step loop runs, per
// wavefront-step -- the weights that turn the static spill report (where the v_readlane / v_writelane instructions ARE) into executed
// instructions.  Read back through cs_debug_region_counts().
#ifdef CS_REGION_COUNTS
__device__ unsigned long long g_region[16];
#define LV_COUNT(k) do { if ((int)(__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&g_region[k], 1ull); } while (0)
#else
#define LV_COUNT(k) do {} while (0)
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
template <int N>
struct EnvV {
    double ax[N], ay[N], yaw[N];
    float csf[N], snf[N];                          // cos / sin of the CURRENT yaw as get_obs emits them
    // targets' normalised coordinates as get_state emits them (norm_target), two targets per register pair: (x of 2k, x of 2k + 1),
    // (y of 2k, y of 2k + 1) -- the sensor tests run on the packed fp32 pipe, two targets per instruction
    v2f tnx[CS_MAX_TARGETS / 2], tny[CS_MAX_TARGETS / 2];
    unsigned found, newly;
    int target_find, flags, time_step, total_reward, mt_pos, episodes, curr_reward, ahead;
    unsigned long long words;
};

// bytes of LDS per wavefront: the staging piece (aliased by the MT19937 row of the in-loop refresh and by the wavefront's block
// of observations on its way out) + the reset hand-over + the hit tapes
__host__ __device__ inline size_t lv_wave_bytes(int W, int n_agents) {
    size_t piece = (size_t)LV_PIECE * W * sizeof(float), row = (size_t)MT_N * sizeof(unsigned);
    size_t obs = (size_t)64 * n_agents * 4 * sizeof(float);
    size_t u = piece > row ? piece : row;
    u = u > obs ? u : obs;
    return (u + 15) / 16 * 16 + LV_SLOT_FLOATS * sizeof(float) + LV_TAPE_ROWS * 64 * sizeof(unsigned);
}
constexpr size_t LV_HEAD_BYTES = ((TRIG_ROWS * TRIG_COLS * 8 + 15) / 16) * 16 + 4 * G * sizeof(double);   // trig table | reset tables

// Kinematics of one lane's env, agent by agent: same arithmetic per agent as kinematics_lane / kinematics (value for value:
// trig_heading_pair is trig_heading twice; the repulsion is the loop over the neighbours that ARE in range, ascending j).
template <int N>
__device__ __forceinline__ void kinematics_v(const DevParams &p, const double *T, const int (&act)[N], EnvV<N> &e) {
    const double PI = 3.141592653589793, TWO_PI = 2.0 * 3.141592653589793, THREE_PI = 3.0 * 3.141592653589793;
    const double DYAW = 3.141592653589793 / 18.0;
    unsigned out = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double yaw = e.yaw[i];
        yaw = act[i] == 1 ? yaw + DYAW : (act[i] == 2 ? yaw + -DYAW : yaw);  // dyaw = [0, pi/18, -pi/18][act]
        yaw = yaw > TWO_PI ? yaw - TWO_PI : (yaw < 0.0 ? yaw + TWO_PI : yaw);
        const double yw = yaw, yr = (yaw <= PI) ? PI - yaw : THREE_PI - yaw;
        double s1, c1, s2, c2;
        trig_heading_pair(T, yw, yr, s1, c1, s2, c2);
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
        while (pend) {   // flight_env_easy.py:293-301, the neighbours within force_dist in ascending order
            LV_COUNT(11);
            const int j = __ffs((int)pend) - 1;
            pend &= pend - 1;
            double xa = 0.0, ya = 0.0;
#pragma unroll
            for (int q = 0; q < N; q++) {
                xa = q == j ? e.ax[q] : xa;
                ya = q == j ? e.ay[q] : ya;
            }
            const double den = (x0 - xa) * (x0 - xa) + (y0 - ya) * (y0 - ya);
            double qx, qy;
            div2_same_denominator(p.force_k * (x0 - xa), p.force_k * (y0 - ya), den, qx, qy);   // one reciprocal for both quotients
            fx += qx;
            fy += qy;
        }
        const double x = (x0 + p.velocity * c1) + fx;
        const double y = (y0 + p.velocity * s1) + fy;
        const bool hit = (x < 0.0) | (x > p.L) | (y < 0.0) | (y > p.L);    // flight_env_easy.py:278
        e.ax[i] = hit ? fmin(fmax(x, 0.0), p.L) : x;
        e.ay[i] = hit ? fmin(fmax(y, 0.0), p.L) : y;
        e.yaw[i] = hit ? yr : yw;
        e.csf[i] = (float)(hit ? c2 : c1);
        e.snf[i] = (float)(hit ? s2 : s1);
        out |= hit ? (1u << i) : 0u;
    }
    e.flags = (e.flags & ~0xff00) | (int)(out << 8);
}

// lane_advance_finish / lane_advance_now of k_rollout_lane for any per-lane env type with mt_pos / ahead
// The hit tapes of the wavefront's envs sit in LDS (`tl`: [LV_TAPE_ROWS][64] dwords, lane l's tape in column l) with a
// per-lane cursor `tpos` (draw slots consumed since the tape was written) instead of ten registers that are barrel-shifted
// after every step: a step reads the few dwords at its cursor (one bank per lane: conflict-free) and adds to tpos.
template <class EnvT>
__device__ __forceinline__ void lv_advance_finish(const DevParams &p, int b0, int lane, int src, const RowRegs &rr, unsigned *rowbuf,
                                                  EnvT &e, unsigned *tl, int &tpos, int s = 64) {
    (void)s;   // (timeline builds, tools/lanev_timeline.py: the step the stamps belong to)
    LANE_STAMP(9);
    row_to_lds(rr, rowbuf, lane);
    const int pos = __shfl(e.mt_pos, src);
    const int a = __shfl(e.ahead, src);
    LANE_STAMP(10);
    row_twist_ahead<true>(rowbuf, p.mt + (size_t)(b0 + src) * MT_STRIDE, pos, a < 0 ? 0 : a, lane);
    LANE_STAMP(11);
    unsigned long long bms[TAPE_DW / 2];
    row_hits_all(p, rowbuf, pos, lane, bms);
#pragma unroll
    for (int it = 0; it < TAPE_DW / 2; it++) {
        const unsigned long long bm = bms[it];
        if (lane == src) {
            tl[(2 * it) * 64 + lane] = (unsigned)(bm & 0xffffffffull);
            tl[(2 * it + 1) * 64 + lane] = (unsigned)(bm >> 32);
        }
    }
    if (lane == src) {
        e.ahead = MT_N;
        tpos = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the buffer is free again (it aliases the staging piece)
    LANE_STAMP(12);
}
template <class EnvT>
__device__ __forceinline__ void lv_advance_now(const DevParams &p, int b0, int lane, unsigned long long need, unsigned *rowbuf, EnvT &e,
                                               unsigned *tl, int &tpos) {
    LV_COUNT(6);
    while (need) {
        LV_COUNT(7);
        const int src = __ffsll((long long)need) - 1;
        need &= need - 1;
        RowRegs rr;
        row_load(p.mt + (size_t)(b0 + src) * MT_STRIDE, lane, rr);
        lv_advance_finish(p, b0, lane, src, rr, rowbuf, e, tl, tpos);
    }
    drain_vmem();   // rare path: joins the steady-state path with nothing of its own in flight
}

// WV: wavefronts per SIMD the register budget allows.  lv_waves(N) = 2 everywhere; teams of up to 3 also exist with 3 (168 VGPRs, a handful
// of spilled registers): slower while a batch is one or two resident rounds (2^18 envs: 4096 wavefronts = 1.33 rounds of 3072), faster once
// the rounds stop mattering -- launch_lanev takes it from CS_LV_W_FROM envs (measured in DESIGN.md section 9).
template <int N, bool VEC, int WV = lv_waves(N)>
__global__ __launch_bounds__(LV_BLOCK, WV) void k_rollout_lanev(DevParams p, StepIO io) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *T = reinterpret_cast<double *>(smem);                                   // trig table (2072 B)
    double *rtab = reinterpret_cast<double *>(smem + LV_HEAD_BYTES - 4 * G * sizeof(double));   // the reset's target tables
    const int W = 4 * N + 3 * p.n_targets;
    int lane = threadIdx.x & 63;   // (made opaque once per step, see the loop)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *wbase = smem + LV_HEAD_BYTES + (size_t)wave * lv_wave_bytes(W, N);
    float *piece = reinterpret_cast<float *>(wbase);                // [LV_PIECE][W] get_state rows of half the wavefront ...
    unsigned *rowbuf = reinterpret_cast<unsigned *>(wbase);         // ... or one MT19937 row (in-loop refresh): never live together
    unsigned *tl = reinterpret_cast<unsigned *>(wbase + lv_wave_bytes(W, N) - LV_TAPE_ROWS * 64 * sizeof(unsigned));   // hit tapes, [dword][lane]
    float2 *slots = reinterpret_cast<float2 *>(reinterpret_cast<char *>(tl) - LV_SLOT_FLOATS * sizeof(float));      // [4][G] reset hand-over
    const int b = io.env0 + blockIdx.x * LV_BLOCK + threadIdx.x;
    const int b0 = b - lane;  // first env of this wavefront
    const int b_end = io.env0 + io.env_n;
    const bool live = VEC || b < b_end;   // a VEC launch has only full wavefronts
    if (wave == 0) load_reset_tab(rtab, lane);
    load_trig_to_lds(T);
    if (b0 >= b_end) return;  // whole wavefront out of range
    const int t16 = lane & (G - 1), gshift = lane & ~(G - 1), grp = lane >> 4;
    const unsigned tmask = p.n_targets >= 16 ? 0xffffu : ((1u << p.n_targets) - 1u);
    constexpr int LOW = 2 * N * CS_MAX_TARGETS;   // words one step can consume: every lane enters a step with that many twisted
    const bool auto_reset = io.flags & CS_AUTO_RESET, freeze = io.flags & CS_FREEZE_DONE;
    EnvV<N> e;
    int tpos = 0;   // draw slots consumed since this lane's tape (in `tl`) was written
    const size_t arow = live ? (size_t)b : (size_t)io.env0;
    bool tape_ok = true;
    {   // hdr / agents / targets of env b -> registers
        const int4 *h4 = reinterpret_cast<const int4 *>(p.hdr + arow * CS_H_WORDS);
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
        e.ahead = p.ahead[arow];
        const double4 *a4 = reinterpret_cast<const double4 *>(p.agent + arow * CS_MAX_AGENTS * 4);
#pragma unroll
        for (int i = 0; i < N; i++) {
            const double4 a = a4[i];
            e.ax[i] = a.x;
            e.ay[i] = a.y;
            e.yaw[i] = a.z;
        }
        const double2 *t2 = reinterpret_cast<const double2 *>(p.tgt + arow * G * 2);
#pragma unroll
        for (int j = 0; j < CS_MAX_TARGETS; j++) {
            const double2 tt = t2[j];   // (rows are 16 targets wide: entries past n_targets are zero and never used)
            e.tnx[j >> 1][j & 1] = (float)((tt.x - p.mid) * p.inv_half);   // what get_state emits (norm_target)
            e.tny[j >> 1][j & 1] = (float)((tt.y - p.mid) * p.inv_half);
        }
#pragma unroll
        for (int i = 0; i < N; i++) {
            double s0, c0;
            trig_heading(T, e.yaw[i], s0, c0);
            e.csf[i] = (float)c0;
            e.snf[i] = (float)s0;
        }
        {
            unsigned tape[TAPE_DW];
            tape_ok = tape_load(p, (int)arow, e, tape);   // (aligned to the cursor)
#pragma unroll
            for (int k = 0; k < LV_TAPE_ROWS; k++) tl[k * 64 + lane] = k < TAPE_DW ? tape[k] : 0u;
        }
        if (!live) {   // a lane without an env never steps, resets or asks for a refill
            e.target_find = 0;
            e.time_step = 0;
            e.ahead = 1 << 20;
            tape_ok = true;
        }
    }
    {   // (an advance also rebuilds a tape that does not match the cursor or the detection threshold)
        const unsigned long long low = __ballot(live && (!tape_ok || e.ahead < LOW));
        if (low) lv_advance_now(p, b0, lane, low, rowbuf, e, tl, tpos);
    }
    int act[N], act_next[N];
    load_actions<N>(io, arow, act_next);
    const int rows_valid = b_end - b0 < 64 ? b_end - b0 : 64;
    constexpr int W_MAX = 4 * N + 3 * CS_MAX_TARGETS;
    constexpr int QP = (LV_PIECE * W_MAX / 4 + 63) / 64;   // float4 chunks per lane of the largest piece
    for (int s = 0; s < io.T; s++) {
        asm volatile("" : "+v"(lane));   // lane predicates are recomputed per step instead of being held (and spilled) as SGPR pairs
#pragma unroll
        for (int i = 0; i < N; i++) act[i] = act_next[i];
        const size_t slot = (size_t)s * p.B + arow;
        LANE_STAMP(0);
        LV_COUNT(0);
        REAL_STAMP(8);
        bool done = live && (e.target_find >= p.n_targets || e.time_step >= p.time_limit);
        // ---- auto-reset (flight_env_easy.py:79-182): the four 16-lane groups of the wavefront each take one resetting env per
        //      round (lane = polar attempt / target); the new targets come back through LDS, the counters by shuffle
        const unsigned long long need = __ballot(done && auto_reset);
        if (__builtin_expect(need != 0ull, 0)) {
            LV_COUNT(1);
            const DevParams &cp = cold_params();
            const CS_AS4 DevParams *q4 = cold_params4();
            const bool mine = (need >> lane) & 1ull;
            const int my_rank = __popcll(need & ((1ull << lane) - 1ull));
            const int nt = q4->n_targets, tm = q4->target_mode;
            const unsigned tmk = nt >= 32 ? ~0u : ((1u << nt) - 1u);
            const unsigned fm = tm == 0 ? ~q4->deter_mask & tmk : 0u;
            unsigned long long pend = need;
            for (int round = 0; pend; round++) {
                LV_COUNT(2);
                unsigned long long m = pend;
                for (int q = 0; q < grp; q++) m &= m ? m - 1 : 0ull;   // this group's env: the grp-th pending one
                const int src = m ? __ffsll((long long)m) - 1 : -1;
                for (int q = 0; q < 4; q++) pend &= pend ? pend - 1 : 0ull;
                const int sl = src >= 0 ? src : lane;
                int g_pos = __shfl(e.mt_pos, sl), g_ahead = __shfl(e.ahead, sl);
                int g_words = 0;                                 // stream words this reset consumed
                unsigned g_found = 0u, g_newly = 0u;
                int g_tf = 0, g_flags = FLAG_DIRTY, g_cr = -1;   // the reset-time pass with no pair in range: no draw, reward -1
                if (src >= 0) {
                    const int br = b0 + src;
                    double mx = 0.0, my = 0.0;
                    bool lean = g_ahead >= 4 * G;   // group-uniform: the first attempt batch lies within the twisted words
                    if (lean) {
                        const CS_AS1 unsigned *wrow = (const CS_AS1 unsigned *)q4->mt + (size_t)br * MT_STRIDE + wrap624(g_pos + 4 * t16);
                        unsigned w4[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) w4[k] = wrow[k];   // (words 0..31 are mirrored behind the row)
                        int words;
                        lean = reset_batch_twisted(w4, rtab[t16], rtab[G + t16], rtab[2 * G + t16], rtab[3 * G + t16], fm, nt, tm, q4->L,
                                                   t16, gshift, mx, my, words);
                        if (lean) {
                            g_pos = wrap624(g_pos + words);
                            g_ahead -= words;
                            g_words = words;
                        }
                    }
                    if (!lean) {   // several batches, or words twisted on the fly: the generic placement from the untouched cursor
                        LV_COUNT(3);
                        unsigned long long wt = 0ull;
                        reset_targets(cp, cp.mt + (size_t)br * MT_STRIDE, t16, gshift, g_pos, wt, g_ahead, mx, my);
                        g_words = (int)wt;
                    }
                    // reset-time detection pass (quirk Q3; its reward is discarded): draws only if a target landed within view
                    // of a start pose (never for agent_mode 0 with the shipped target file)
                    const StartTab<N> st = start_tab<N>();
                    const double vr2 = q4->view_r2;
                    bool near = false;
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        const double ddx = mx - st.x[i], ddy = my - st.y[i];
                        near = near | ((t16 < nt) & (ddx * ddx + ddy * ddy <= vr2));
                    }
                    if (__builtin_expect(((__ballot(near) >> gshift) & 0xffffull) != 0ull, 0)) {   // group-uniform
                        LV_COUNT(4);
                        Env<N> g;
#pragma unroll
                        for (int i = 0; i < N; i++) {
                            g.ax[i] = st.x[i];
                            g.ay[i] = st.y[i];
                            g.yaw[i] = st.yaw;
                            g.cs[i] = g.sn[i] = 0.0;
                        }
                        g.tx = mx;
                        g.ty = my;
                        g.ntx = g.nty = 0.0f;
                        g.found = g.newly = g.newly_reset = 0u;
                        g.target_find = 0;
                        g.flags = 0;
                        g.time_step = 0;
                        g.total_reward = 0;
                        g.curr_reward = 0;
                        g.episodes = 0;
                        g.mt_pos = g_pos;
                        g.ahead = g_ahead;
                        g.words = (unsigned long long)g_words;
                        detect_pass<N>(cp, br, t16, gshift, g, mt_prefetch(cp.mt + (size_t)br * MT_STRIDE, g.mt_pos, t16));
                        g_pos = g.mt_pos;
                        g_ahead = g.ahead;
                        g_words = (int)g.words;
                        g_found = g.found;
                        g_newly = g.newly;
                        g_tf = g.target_find;
                        g_flags = g.flags;
                        g_cr = g.curr_reward;
                    }
                    reinterpret_cast<double2 *>(cp.tgt + (size_t)br * G * 2)[t16] = make_double2(mx, my);
                    slots[grp * G + t16] = make_float2((float)((mx - q4->mid) * q4->inv_half), (float)((my - q4->mid) * q4->inv_half));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // the q-th pending env of this round was reset by group q: its (group-uniform) results come back
                const int q = my_rank - 4 * round;
                const bool got = mine && q >= 0 && q < 4;
                const int leader = got ? 16 * q : lane;
                const int r_pos = __shfl(g_pos, leader), r_ahead = __shfl(g_ahead, leader), r_words = __shfl(g_words, leader);
                const int r_found = __shfl((int)g_found, leader), r_newly = __shfl((int)g_newly, leader);
                const int r_tf = __shfl(g_tf, leader), r_flags = __shfl(g_flags, leader), r_cr = __shfl(g_cr, leader);
                if (got) {
#pragma unroll
                    for (int j = 0; j < CS_MAX_TARGETS; j++) {
                        const float2 v = slots[q * G + j];
                        e.tnx[j >> 1][j & 1] = v.x;
                        e.tny[j >> 1][j & 1] = v.y;
                    }
                    // the reset consumed r_words stream words, twisted ones first: their draw slots leave the tape
                    tpos += r_words >> 1;
                    tpos = tpos < 319 ? tpos : 319;   // (a tape run past its end: the env is below LOW and gets a new one below)
                    e.mt_pos = r_pos;
                    e.ahead = r_ahead;
                    e.words += (unsigned long long)r_words;
                    e.episodes += 1;
                    e.found = (unsigned)r_found;
                    e.newly = (unsigned)r_newly;
                    e.target_find = r_tf;
                    e.flags = r_flags;
                    e.curr_reward = r_cr;
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
                            e.snf[i] = (float)s0;
                            e.csf[i] = (float)c0;
                        }
                    }
                    done = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();   // the slots are free again
            }
            // a reset that ran past the twisted words leaves its lane without a tape for this step: rebuild
            const unsigned long long low = __ballot(e.ahead < LOW);
            if (low) lv_advance_now(p, b0, lane, low, rowbuf, e, tl, tpos);
            drain_vmem();
        }
        int reward = 0;
        bool term = true;
        const bool stepping = live && !(done && freeze);
        LANE_STAMP(1);
        e.flags &= ~(FLAG_DIRTY | FLAG_RESET_PASS);
        // ---- in-loop refresh, first half: the row of the env running lowest on twisted words is requested after the kinematics
        //      and twisted after the draws of this step (each env comes round about every 64 steps)
#ifndef CS_LV_REFRESH_EARLY
#define CS_LV_REFRESH_EARLY 0   /* request the refresh row BEFORE the kinematics instead of after them: measured on one box, two
                                   runs each (tools/gpu_r4_g.sh): 5 agents no difference, 3 agents -2..-3 % */
#endif
        RowRegs rr;
        int cand;
        auto request_row = [&]() __attribute__((always_inline)) {
            constexpr int URGENT = LOW + 64, NORMAL = CS_LV_NORMAL > LOW + 128 ? CS_LV_NORMAL : LOW + 128;
            const unsigned long long urgent = __ballot(e.ahead < URGENT), normal = __ballot(e.ahead < NORMAL);
            cand = urgent ? __ffsll((long long)urgent) - 1 : (normal ? __ffsll((long long)normal) - 1 : -1);
            if (cand >= 0) row_load(p.mt + (size_t)(b0 + cand) * MT_STRIDE, lane, rr);
        };
        if (CS_LV_REFRESH_EARLY) request_row();
        if (stepping) kinematics_v<N>(p, T, act, e);
        if (!CS_LV_REFRESH_EARLY) request_row();
        LANE_STAMP(2);
        // ---- the agents' floats (get_obs / get_state)
        float fx[N], fy[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            fx[i] = (float)((e.ax[i] - p.mid) * p.inv_half);
            fy[i] = (float)((e.ay[i] - p.mid) * p.inv_half);
        }
        // ---- sensor tests (flight_env_easy.py:237): fp32 pre-filter on the normalised coordinates, exact fp64
        //      comparison for the pairs it cannot decide; bit 16*i + j = (agent i, target j) in range
        unsigned long long lo = 0, hi = 0;  // agents 0..3 / 4..7
        if (stepping) {
            const float thr_lo = p.thr32 - p.eps32, thr_hi = p.thr32 + p.eps32;
#pragma unroll
            for (int i = 0; i < N; i++) {
                // sign bits of d2 - thr_lo / d2 - thr_hi, target 15 first, funnel-shifted into the masks (one
                // v_alignbit each): bit j of `sure` = (d2 < thr - eps), of `maybe` = (d2 < thr + eps)
                unsigned sure = 0, maybe = 0;
#pragma unroll
                for (int k = CS_MAX_TARGETS / 2 - 1; k >= 0; k--) {   // targets 2k + 1, 2k: one packed instruction per operation
                    const v2f dx = e.tnx[k] - v2f{fx[i], fx[i]}, dy = e.tny[k] - v2f{fy[i], fy[i]};
                    const v2f d2 = __builtin_elementwise_fma(dx, dx, dy * dy);   // per element fma(dx, dx, dy * dy), as before
                    const v2f a = d2 - v2f{thr_lo, thr_lo}, c = d2 - v2f{thr_hi, thr_hi};
                    sure = __builtin_amdgcn_alignbit(sure, __float_as_uint(a[1]), 31);
                    sure = __builtin_amdgcn_alignbit(sure, __float_as_uint(a[0]), 31);
                    maybe = __builtin_amdgcn_alignbit(maybe, __float_as_uint(c[1]), 31);
                    maybe = __builtin_amdgcn_alignbit(maybe, __float_as_uint(c[0]), 31);
                }
                unsigned m = sure & tmask;
                unsigned fz = maybe & ~sure & tmask;
                while (fz) {  // (t_x-x)**2 + (t_y-y)**2 <= view_range**2 on the fp64 values
                    LV_COUNT(10);
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
        LANE_STAMP(3);
        // ---- one np.random.rand() per in-range pair, found or not (quirk Q4), in agent-major order: the r-th set bit of
        //      (lo, hi) takes draw slot r of the tape.  A draw hits with probability 0.9, so instead of walking the pairs the
        //      step walks the MISSES: the zero bits among the tape's next `total` slots (0.3 per env-step, at most two or
        //      three in a wavefront); a miss at slot r clears the r-th pair, every other in-range pair detects its target.
        const int total = __popcll(lo) + (N > 4 ? __popcll(hi) : 0);
        unsigned hitmask;
        {
            constexpr int NW = (N * CS_MAX_TARGETS + 31) / 32;   // dwords of the widest window
            unsigned z[NW];
            {
                const int idx = tpos >> 5;
                const unsigned sh = (unsigned)tpos & 31u;
                unsigned d[NW + 1];
#pragma unroll
                for (int k = 0; k <= NW; k++) d[k] = tl[(idx + k) * 64 + lane];
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    const unsigned w = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh);   // slots 32 k .. 32 k + 31 from the cursor
                    const int nb = total - 32 * k;
                    const unsigned in = nb >= 32 ? ~0u : (nb > 0 ? (1u << nb) - 1u : 0u);
                    z[k] = ~w & in;
                }
            }
            unsigned any = z[0];
#pragma unroll
            for (int k = 1; k < NW; k++) any |= z[k];
            unsigned long long mlo = 0ull, mhi = 0ull;   // the pairs that missed
            if (__builtin_expect(__ballot(any != 0u) != 0ull, 0)) {
                LV_COUNT(8);
                // in-range pairs before agent i's (agent-major order): base[i]
                int base[N + 1];
                base[0] = 0;
#pragma unroll
                for (int i = 0; i < N; i++)
                    base[i + 1] = base[i] + __popc((unsigned)((i < 4 ? lo >> (16 * i) : hi >> (16 * (i - 4))) & 0xffffull));
                while (__ballot(any != 0u)) {   // wave-uniform
                    LV_COUNT(9);
                    if (any != 0u) {
                        int r = 0;
                        bool got = false;
#pragma unroll
                        for (int k = 0; k < NW; k++) {
                            const bool here = !got && z[k] != 0u;
                            r = here ? 32 * k + __ffs((int)z[k]) - 1 : r;
                            z[k] = here ? z[k] & (z[k] - 1u) : z[k];
                            got = got || here;
                        }
                        int i = 0, bs = 0;
#pragma unroll
                        for (int k = 1; k < N; k++) {
                            const bool past = r >= base[k];
                            i = past ? k : i;
                            bs = past ? base[k] : bs;
                        }
                        const int fsh = 16 * (i & 3);   // agent i's 16-bit field of lo (agents 0..3) / hi (4..7)
                        const unsigned mi = (unsigned)(((i < 4 ? lo : hi) >> fsh) & 0xffffull);
                        const int bit = kth_set_bit16(mi, r - bs) & 15;
                        const unsigned long long bm = 1ull << (fsh + bit);
                        mlo |= i < 4 ? bm : 0ull;
                        mhi |= i < 4 ? 0ull : bm;
                        any = z[0];
#pragma unroll
                        for (int k = 1; k < NW; k++) any |= z[k];
                    }
                }
            }
            const unsigned long long hl = lo & ~mlo, hh = hi & ~mhi;
            hitmask = (unsigned)((hl | (hl >> 16) | (hl >> 32) | (hl >> 48)) & 0xffffull);
            if (N > 4) hitmask |= (unsigned)((hh | (hh >> 16) | (hh >> 32) | (hh >> 48)) & 0xffffull);
            e.mt_pos = wrap624(e.mt_pos + 2 * total);
            e.words += (unsigned long long)(2 * total);
            e.ahead -= 2 * total;
            tpos += total;
        }
        LANE_STAMP(4);
        if (stepping) {   // flight_env_easy.py:238-253
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
        }
        LANE_STAMP(5);
        // ---- in-loop refresh, second half (wave-uniform): the row has long arrived; its new tape goes to the env's lane
        if (cand >= 0) {
            LV_COUNT(5);
            lv_advance_finish(p, b0, lane, cand, rr, rowbuf, e, tl, tpos, s);
        }
        LANE_STAMP(13);
        {   // a lane that cannot wait for its turn (several running low at once): on the spot
            const unsigned long long low = __ballot(e.ahead < LOW);
            if (__builtin_expect(low != 0ull, 0)) lv_advance_now(p, b0, lane, low, rowbuf, e, tl, tpos);
        }
        // ---- the next step's actions, requested BEFORE this step's stores
        load_actions<N>(io, (size_t)(s + 1 < io.T ? s + 1 : s) * p.B + arow, act_next);
        LANE_STAMP(6);
        // ---- this step's outputs
        if (live) {
            io.reward[slot] = (float)reward;
            io.terminated[slot] = term ? 1 : 0;
            io.win[slot] = (e.flags & FLAG_WIN) ? 1 : 0;
        }
        if (VEC || io.obs) {
            // get_obs (flight_env_easy.py:218-221): one float4 per (env, agent).  Stored straight from the lanes, the N stores of a
            // step each scatter 64 16-byte pieces at a stride of 16 N bytes -- partial sectors that non-temporal stores do not let
            // the L2 merge (WRITE_SIZE +115 B per env-step at 5 agents, profiles/r04_lanev5_pmc.json).  The wavefront's block of
            // 64 N float4 is contiguous in the table: it goes through the staging piece and leaves as N fully coalesced 1 KB stores.
            v4f *ob = reinterpret_cast<v4f *>(piece);
            if (live) {
#pragma unroll
                for (int i = 0; i < N; i++) ob[lane * N + i] = v4f{fx[i], fy[i], e.csf[i], e.snf[i]};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v4f *o = reinterpret_cast<v4f *>(io.obs) + ((size_t)s * p.B + b0) * N;
            v4f ov[N];
#pragma unroll
            for (int q = 0; q < N; q++) ov[q] = ob[lane + 64 * q];
#pragma unroll
            for (int q = 0; q < N; q++)
                if (VEC || lane + 64 * q < rows_valid * N) __builtin_nontemporal_store(ov[q], o + lane + 64 * q);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // every lane has read its chunks: the get_state rows may overwrite the piece
        }
        if (VEC || io.state) {   // get_state rows (flight_env_easy.py:190-216), half a wavefront at a time through the staging piece
#pragma unroll
            for (int h = 0; h < 2; h++) {   // (unrolled: every store of the step in one straight line)
                // (the target count as an opaque scalar: left to itself the compiler hoists the sixteen `j < n_targets` tests out of the
                // step loop as sixteen 64-bit lane masks -- 32 SGPRs of a kernel at its SGPR ceiling, i.e. ~60 v_readlane reloads per
                // wavefront-step in this block alone, tools/spill_exec.py; recomputed here they are one s_cmp each)
                int nt = p.n_targets;
                asm volatile("" : "+s"(nt));
                if ((lane >> 5) == h && live) {
                    float *row = piece + (size_t)(lane & (LV_PIECE - 1)) * W;
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        row[4 * i + 0] = fx[i];
                        row[4 * i + 1] = fy[i];
                        row[4 * i + 2] = e.csf[i];
                        row[4 * i + 3] = e.snf[i];
                    }
#pragma unroll
                    for (int j = 0; j < CS_MAX_TARGETS; j++) {
                        if (j < nt) {
                            row[4 * N + 3 * j + 0] = e.tnx[j >> 1][j & 1];
                            row[4 * N + 3 * j + 1] = e.tny[j >> 1][j & 1];
                            row[4 * N + 3 * j + 2] = ((e.found >> j) & 1u) ? 1.0f : 0.0f;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float *dst = io.state + ((size_t)s * p.B + b0 + h * LV_PIECE) * W;
                if (VEC) {   // full wavefront, 16-byte aligned block of rows: float4 chunks; surplus lanes repeat the last chunk
                    const float4 *src4 = reinterpret_cast<const float4 *>(piece);
                    float4 *dst4 = reinterpret_cast<float4 *>(dst);
                    const int last = LV_PIECE * W / 4 - 1;
                    constexpr int GRP = 3;   // chunks in flight per lane: LDS reads first, then their stores (12 registers, not 4 QP)
#pragma unroll
                    for (int q0 = 0; q0 < QP; q0 += GRP) {
                        v4f v[GRP];
                        int k[GRP];
#pragma unroll
                        for (int q = 0; q < GRP; q++) {
                            if (q0 + q < QP) {
                                k[q] = lane + 64 * (q0 + q) < last ? lane + 64 * (q0 + q) : last;
                                const float4 x = src4[k[q]];
                                v[q] = v4f{x.x, x.y, x.z, x.w};
                            }
                        }
#pragma unroll
                        for (int q = 0; q < GRP; q++)
                            if (q0 + q < QP) __builtin_nontemporal_store(v[q], reinterpret_cast<v4f *>(dst4 + k[q]));
                        asm volatile("" ::: "memory");
                    }
                } else {
                    const int rows = rows_valid - h * LV_PIECE < LV_PIECE ? rows_valid - h * LV_PIECE : LV_PIECE;   // may be <= 0
                    for (int k = lane; k < rows * W; k += 64) dst[k] = piece[k];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();   // every lane has read the piece: the other half may overwrite it
            }
        }
        LANE_STAMP(7);
    }
    if (live) {
        // (the state blob's tables through cold_params(): read from the kernarg segment here, so that their six pointers are not
        // held -- i.e. spilled and reloaded -- across the step loop of a kernel at its SGPR ceiling)
        const DevParams &cp = cold_params();
        int4 *h4 = reinterpret_cast<int4 *>(cp.hdr + (size_t)b * CS_H_WORDS);
        h4[0] = make_int4((int)e.found, (int)e.newly, e.target_find, e.flags);
        h4[1] = make_int4(e.time_step, e.total_reward, e.mt_pos, e.episodes);
        int *h2 = cp.hdr + (size_t)b * CS_H_WORDS + 8;   // (word 11, CS_H_NEWLY_RESET, is flight's: left as it is)
        *reinterpret_cast<int2 *>(h2) = make_int2((int)(unsigned)(e.words & 0xffffffffull), (int)(unsigned)(e.words >> 32));
        h2[2] = e.curr_reward;
        cp.ahead[b] = e.ahead;
        double4 *a4 = reinterpret_cast<double4 *>(cp.agent + (size_t)b * CS_MAX_AGENTS * 4);
#pragma unroll
        for (int i = 0; i < N; i++) a4[i] = make_double4(e.ax[i], e.ay[i], e.yaw[i], 0.0);
        // the tape goes back to the state blob rebased to the cursor, for the next launch
        unsigned tape[TAPE_DW];
#pragma unroll
        for (int k = 0; k < TAPE_DW; k++) tape[k] = tl[k * 64 + lane];
        tape_shift<8>(tape, tpos);
        U4 *tp = reinterpret_cast<U4 *>(cp.tape + (size_t)b * TAPE_STRIDE);
        tp[0] = U4{tape[0], tape[1], tape[2], tape[3]};
        tp[1] = U4{tape[4], tape[5], tape[6], tape[7]};
        tp[2] = U4{tape[8], tape[9], (unsigned)(e.words & 0xffffffffull), (unsigned)(e.words >> 32)};
        tp[3] = U4{(unsigned)(cp.detect_K & 0xffffffffull), (unsigned)(cp.detect_K >> 32), 0u, 0u};
    }
}
