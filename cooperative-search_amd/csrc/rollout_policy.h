// cooperative-search_amd/csrc/rollout_policy.h -- k_rollout_policy: the fused closed loop (agent network forward -> env.step, T steps per launch).
// Included by coopsearch.hip inside its anonymous namespace, after the 16-lane group code and policy_dev.h.  Not a translation unit of its own.

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
