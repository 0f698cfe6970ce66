// cooperative-search_amd/csrc/rollout_oct.h -- the octet layout (one env per 8 lanes): its kinematics stages, detection pass, resets, row top-ups, and k_rollout_oct, the one-wavefront kernel (16384 < envs < 65536).
// Included by coopsearch.hip inside its anonymous namespace, after rollout_lane.h (lane_from and the reset helpers it shares).  Not a translation unit of its own.

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
#ifndef CS_OCT_PREFILTER
#define CS_OCT_PREFILTER 0   /* k_rollout_oct's step: packed-fp32 pre-filter of the sensor tests (oct_detect_impl, PRE); 0: fp64 only.  Measured
                                (round 6, 32768 envs, two passes, us per step off -> on): 5 agents 6.383 / 6.339 -> 6.539 / 6.571, 3 agents 4.158 /
                                4.100 -> 4.292 / 4.280 -- the one-wavefront kernel is at its 168-VGPR cap and the four extra registers cost
                                more (22 -> 26 spilled VGPRs at 5 agents) than the eight instructions per agent save */
#endif
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

// Lanes per env of the "octet" kernels: LG = 8, 8 envs per wavefront, lane t owns agent t and targets t, t + 8.  (The layout is a
// trait so that the roles, protocol and arithmetic do not spell out the 8; round 5 measured a 5-lanes-per-env packing for teams of 5
// through it -- slower at every batch the pair kernels serve, profiles/r05_experiments.md K / L -- and round 6 removed it.)
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
template <int N, int LG = OG>
struct EnvO {
    double x, y, yaw, cs, sn;              // this lane's agent (lanes t < N)
    double tx[OctLay<LG>::TPL], ty[OctLay<LG>::TPL];   // targets t + LG k
    float tnx[OctLay<LG>::TPL], tny[OctLay<LG>::TPL];  // ... as get_state emits them ((x - mid) * inv_half in fp32): the sensor pre-filter's operands
    unsigned found, newly, newly_reset;    // octet-uniform from here on
    int target_find, flags, time_step, total_reward, mt_pos, episodes, curr_reward, ahead;
    unsigned long long words;
};

// tnx / tny from tx / ty (after every assignment of the targets: prologue, reset).  A slot without a target gets a coordinate far outside
// the map: never in range, never near the threshold, no NaN whatever the padding of the target array holds.
template <int N, int LG>
__device__ __forceinline__ void oct_norm_targets(double mid, double inv_half, int n_targets, int t, EnvO<N, LG> &e) {
#pragma unroll
    for (int k = 0; k < OctLay<LG>::TPL; k++) {
        const bool exists = t + LG * k < n_targets;
        e.tnx[k] = exists ? (float)((e.tx[k] - mid) * inv_half) : 1.0e3f;   // norm_target, the value get_state emits
        e.tny[k] = exists ? (float)((e.ty[k] - mid) * inv_half) : 1.0e3f;
    }
}
template <int N, int LG>
__device__ __forceinline__ void oct_norm_targets(const DevParams &p, int t, EnvO<N, LG> &e) {
    oct_norm_targets<N, LG>(p.mid, p.inv_half, p.n_targets, t, e);
}

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
            const bool c_lt = d2 < p.force_d2;
            // "some lane is within force_dist", from the comparison's own lane mask (a ballot of a bare comparison IS its result register)
            // and the step's mask of agent lanes.  The coincidence test of flight_env_easy.py:298 (an agent on top of agent I exerts no
            // force) is made inside: it can only matter where c_lt holds, and two agents clamped into one corner are rare enough that a
            // stage entered for them alone costs less than two comparisons in every stage of every step (round 6)
            const unsigned long long not_i = ~OctLay<LG>::lanes_t(I);
            if (__ballot(c_lt) & act_mask & not_i) {   // wave-uniform
                const bool c_nx = k.cx != xi, c_ny = k.cy != yi;
                const bool inr = act_lane & (t != I) & c_lt & (c_nx | c_ny);
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
    static_assert(LG == OG, "one layout: the octet");
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
// PRE (round 6; the step's pass of k_rollout_od and k_rollout_oct): the n x 2 sensor tests of a lane are decided on the packed-fp32
// pipe first, as in the lane kernels: the lane's two targets as ONE pair of floats in normalised coordinates
// (e.tnx / e.tny), agent i's normalised position from `nrow[nstride * i]` (the floats get_state emits; LDS), d2 = fma(dx, dx, dy * dy)
// per element -- two packed subtractions, one packed product, one packed fma and one packed subtraction of the threshold per AGENT
// instead of six fp64 operations per PAIR -- and the fp32 verdict `d2 < thr` stands unless some pair of the wavefront lies within
// eps32 = 1e-6 + 4e-6 thr of the threshold (at least 4x the error bound of d2; one running minimum of |d2 - thr| per lane, one
// wave-uniform test), in which case the whole pass is redone with the reference's fp64 comparison (about one wavefront-step in 2000
// at the shipped configuration).  The outcome is the exact one in every case.
template <int N, int LG, int AP, bool LAZY, bool PRE = false>
__device__ __forceinline__ int oct_detect_impl(const DevParams &p, const double2 (*pos)[AP], int o, int t, int sh8, bool stepping,
                                               EnvO<N, LG> &e, unsigned (&tape)[TAPE_DW], int &tcur, const float *nrow = nullptr,
                                               int nstride = 0) {
    constexpr int MAXDW = (N * CS_MAX_TARGETS) / 32 < 1 ? 1 : (N * CS_MAX_TARGETS) / 32;   // draws of one pass, in dwords
    constexpr int TPL = OctLay<LG>::TPL;   // this lane's targets: t, t + LG, ...
    static_assert(!PRE || TPL == 2, "the packed pre-filter holds a lane's two targets in one register pair");
    bool has[TPL], inr[N][TPL];
    unsigned below[TPL];
    int rank[N][TPL];
    int base = 0;
#pragma unroll
    for (int k = 0; k < TPL; k++) {
        has[k] = stepping & (t + LG * k < p.n_targets);
        below[k] = (1u << (t + LG * k)) - 1u;
    }
    auto exact = [&]() __attribute__((always_inline)) {   // (t_x-x)**2 + (t_y-y)**2 <= view_range**2 on the fp64 values
#pragma unroll
        for (int i = 0; i < N; i++) {
            const double2 a = pos[o][i];
#pragma unroll
            for (int k = 0; k < TPL; k++) {
                const double dx = e.tx[k] - a.x, dy = e.ty[k] - a.y;
                inr[i][k] = has[k] & (dx * dx + dy * dy <= p.view_r2);
            }
        }
    };
    if constexpr (PRE) {
        typedef float pk2 __attribute__((ext_vector_type(2)));
        const pk2 tnx = {e.tnx[0], e.tnx[1]}, tny = {e.tny[0], e.tny[1]};
        const float thr = p.thr32;
        float margin = 3.0e38f;   // min over this lane's pairs of |d2 - thr|
#pragma unroll
        for (int i = 0; i < N; i++) {
            const float ax = nrow[nstride * i], ay = nrow[nstride * i + 1];
            const pk2 dx = tnx - pk2{ax, ax}, dy = tny - pk2{ay, ay};
            const pk2 d2 = __builtin_elementwise_fma(dx, dx, dy * dy);
            const pk2 u = d2 - pk2{thr, thr};
            inr[i][0] = has[0] & (u[0] < 0.0f);
            inr[i][1] = has[1] & (u[1] < 0.0f);
            margin = __builtin_fminf(margin, __builtin_fminf(__builtin_fabsf(u[0]), __builtin_fabsf(u[1])));
        }
#ifndef CS_PREFILTER_EPS_SCALE
#define CS_PREFILTER_EPS_SCALE 1.0f   /* test builds widen the band (tests/test_gpu_jitter.py: prewide_n5) so that both paths run often */
#endif
        if (__builtin_expect(__ballot(margin <= p.eps32 * CS_PREFILTER_EPS_SCALE) != 0ull, 0)) exact();   // wave-uniform
    } else {
        exact();
    }
    // (the ballot of `has & c` is the comparison's own result register and one scalar AND)
#pragma unroll
    for (int i = 0; i < N; i++) {
        unsigned gm = 0u;
#pragma unroll
        for (int k = 0; k < TPL; k++) gm |= oct_slice<LG>(__ballot(inr[i][k]), sh8) << (LG * k);
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
        unsigned long long bms[TAPE_DW / 2];
        row_hits_all(p, rowbuf, pos, lane, bms);
#pragma unroll
        for (int it = 0; it < TAPE_DW / 2; it++) {
            const unsigned long long bm = bms[it];
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
    unsigned long long bms[TAPE_DW / 2];
    row_hits_all(p, rowbuf, pos, lane, bms);
#pragma unroll
    for (int it = 0; it < TAPE_DW / 2; it++) {
        const unsigned long long bm = bms[it];
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
            oct_norm_targets<N, LG>(mid, inv_half, n_targets, t, e);
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
        oct_norm_targets<N, OG>(p, t, e);
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
        const int reward = oct_detect_impl<N, OG, OCT_PAD, true, CS_OCT_PREFILTER != 0>(p, sh.pos, o, t, sh8, stepping, e, tape, tcur, row, 4);
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
