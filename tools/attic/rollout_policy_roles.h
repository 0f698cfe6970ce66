// cooperative-search_amd/csrc/rollout_policy_roles.h -- k_rollout_policy_roles: the fused closed loop (T x (agent network forward
// -> env.step), common/rollout.py:43-76) split by ROLE.  Included by coopsearch.hip inside its anonymous namespace (after
// k_rollout_policy, whose helper types it uses).  Split-fp16 matrix path only (policy_dev.h).
//
// k_rollout_policy keeps a block's 16 envs and their network rows on one CU and walks the chain network -> actions -> env.step ->
// observation strictly in series: 8 us per step, in which the matrix pipe works for 1 us and the env arithmetic for 2.5; the rest
// is seven workgroup barriers and the LDS round trips between them, with ONE wavefront per SIMD (444 registers: the weights and
// the env state together) and nothing to fill its stalls.  Here the two halves get their own wavefronts and half the envs each:
//     wavefronts 0..3  NET   the network (weights resident: 128 VGPRs), one 8-env GROUP at a time;
//     wavefronts 4..7  ENV   env.step of their 4 envs each (the 16-lane group code of k_rollout_policy, unchanged);
//                            wavefronts 4, 5 are group 0, wavefronts 6, 7 group 1;
// and the groups alternate: while NET computes the actions of group 0's step s, ENV steps group 1 with the actions of ITS step
// s (computed the phase before), then the other way round.  Both roles fit 256 registers, so a CU holds all eight wavefronts --
// two per SIMD, one of each role -- and one role's latencies hide under the other's work.  No workgroup barrier in the loop:
//     net_done[g]   steps for which group g's actions are in s_act[g]            (NET wavefront 0 -> ENV of group g)
//     env_done[j]   steps ENV wavefront j has executed (its tile holds the obs)   (ENV j -> NET)
//     nsync[k]      the six rendezvous of the four NET wavefronts inside a phase (LDS counters, ds_add + poll)
// -- plain LDS words written and polled with hand-placed ds instructions: the LDS serves a wavefront's accesses in order, so data
// written before a counter is visible to whoever has seen the counter (the protocol of k_rollout_od).
// Per-row arithmetic is k_policy_h's / k_rollout_policy's (same functions, same order: a row's dot products do not depend on which
// tile it sits in), the env arithmetic is step_once: results are bit-identical to T pairs of cs_policy_forward + cs_step.
#pragma once

#if CS_POLICY_F16

constexpr int RP_BLOCK = 512;
constexpr int RP_GROUP_ENVS = 8;            // envs per group: two ENV wavefronts of 4 envs

struct RpSync {
    int nsync[8];       // NET-internal rendezvous counters (monotonic: 4 per pass)
    int net_done[2];
    int env_done[4];
    int pad[2];
};

__device__ __forceinline__ void rp_post(int *w, int v, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) lds_post(w, v);
}
__device__ __forceinline__ void rp_wait_ge(const int *w, int v) {
    while (lds_peek(w) < v) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// rendezvous of the four NET wavefronts: the `pass`-th time this counter is used (pass = 0, 1, ...)
__device__ __forceinline__ void rp_net_sync(int *ctr, int pass, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(lds_offset_of(ctr)), "v"(1) : "memory");
    rp_wait_ge(ctr, 4 * (pass + 1));
}

template <int N>
__global__ __launch_bounds__(RP_BLOCK) void k_rollout_policy_roles(DevParams p, StepIO io, PolicyIO pio) {
    constexpr int NA = 3;                                   // the env has three actions (flight_env_easy.py:32)
    constexpr int GR = RP_GROUP_ENVS * N;                   // network rows of a group
    constexpr int TG = (GR + 15) / 16, ROWS_G = 16 * TG;    // row tiles of a group
    __shared__ double T[TRIG_ROWS * TRIG_COLS];
    __shared__ WaveTile tiles[4];                           // one per ENV wavefront
    __shared__ int s_act[2][ROWS_G];                        // last / chosen action per row of each group
    __shared__ float s_b3[16];
    __shared__ double s_eps[BLOCK / G];                     // the block's 16 envs' epsilon (cs_epsilon)
    __shared__ RpSync sy;
    extern __shared__ __attribute__((aligned(16))) float pol_lds[];
    // scratch of the group NET is working on: a | b as (hi, lo) plane pairs [ROWS_G][HST]; per group: hs planes and s_h (fp32)
    _Float16 *a_hi = reinterpret_cast<_Float16 *>(pol_lds), *a_lo = a_hi + ROWS_G * HST;
    _Float16 *b_hi = a_lo + ROWS_G * HST, *b_lo = b_hi + ROWS_G * HST;
    _Float16 *hs_base = b_lo + ROWS_G * HST;                // [2 groups][2 planes][ROWS_G][HST]
    float *sh_base = reinterpret_cast<float *>(hs_base + 2 * 2 * ROWS_G * HST);   // [2 groups][ROWS_G][LDW]
    float *s_q = pol_lds;                                   // [4][ROWS_G * 17] floats: aliases the a planes (272 <= 288 B per row)

    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool is_net = wv < 4;
    const int b0 = blockIdx.x * (BLOCK / G);                // first env of the block (16 envs)
    if (threadIdx.x == 0) {
        for (int k = 0; k < 8; k++) sy.nsync[k] = 0;
        sy.net_done[0] = sy.net_done[1] = 0;
        for (int k = 0; k < 4; k++) sy.env_done[k] = 0;
    }
    if (threadIdx.x < 16) s_b3[threadIdx.x] = pio.w[HOFF_B3 + threadIdx.x];
    if (threadIdx.x < BLOCK / G)
        s_eps[threadIdx.x] = (pio.eps_dev && b0 + (int)threadIdx.x < p.B) ? pio.eps_dev[b0 + threadIdx.x] : pio.epsilon;
    const int in_dim = 4 + NA + N;

    if (!is_net) {
        // ------------------------------------------------------------------------------------------------ ENV role
        const int j = wv - 4, g = j >> 1;                   // ENV wavefront j: envs b0 + 4j .. + 3, group j / 2
        const int lt = threadIdx.x - 256;
        const int b = b0 + lt / G, t = lt % G, grp = lane >> 4;
        const bool live = b < p.B;
        Env<N> e;
        if (live) env_load<N>(p, b, t, e);
        {   // the trig table: this role's 256 threads load it (load_trig_to_lds uses threadIdx.x / blockDim.x of a whole block)
            constexpr int NT = TRIG_ROWS * TRIG_COLS;
            const double *src = &g_trig[0][0];
            for (int i = lt; i < NT; i += 256) T[i] = src[i];
        }
        __syncthreads();   // (1) table, sync words, s_eps
        const int wave_b0 = b0 + 4 * j;
        const int nvalid = p.B - wave_b0 < 4 ? p.B - wave_b0 : 4;
        const bool wave_valid = nvalid > 0;
        WaveTile &tile = tiles[j];
        const EmitPlan<N> plan = make_emit_plan<N>(p, lane, wave_valid ? nvalid : 1);
        constexpr bool PIPE = N <= 4;
        if (live) env_trig<N>(T, e);
        emit_deposit<N>(p, tile, t, grp, live, e, 0, false);   // the current observation (what get_obs would return now)
        MtWin win = {0u, 0u};
        if (live) win = mt_prefetch(p.mt + (size_t)b * MT_STRIDE, e.mt_pos, t);
        unsigned no_tape[TAPE_DW];   // the closed loop twists its words on demand
        __syncthreads();   // (2) every tile holds its initial observation; NET's hidden state / last actions are staged
        const int el = 4 * (j & 1) + grp;                   // env within the group
        for (int s = 0; s < io.T; s++) {
            rp_wait_ge(&sy.net_done[g], s + 1);             // the actions of this group's step s are in s_act[g]
            int act[N];
#pragma unroll
            for (int i = 0; i < N; i++) act[i] = s_act[g][el * N + i];
            // will this env execute the step?  (step_once: an env terminated on entry is reset first under CS_AUTO_RESET, left
            // alone under CS_FREEZE_DONE): only executed steps anneal epsilon (rollout.py:75-76)
            const bool executed = live && !((e.target_find >= p.n_targets || e.time_step >= p.time_limit) &&
                                            !(io.flags & CS_AUTO_RESET) && (io.flags & CS_FREEZE_DONE));
            if (wave_valid)
                step_once<N, 0>(p, T, io, tile, b, lane, (size_t)s * p.B + wave_b0, plan, live, act, win, s + 1 < io.T,
                                PIPE && s > 0, (size_t)(s - 1) * p.B + wave_b0, PIPE, e, no_tape, false, false);
            if (pio.per_step && pio.eps_dev && executed && t == 0) {
                const double v = s_eps[4 * j + grp];
                s_eps[4 * j + grp] = v > pio.min_eps ? v - pio.anneal : v;
            }
            rp_post(&sy.env_done[j], s + 1, lane);          // the tile holds the observation after step s (and s_eps is current)
        }
        if (PIPE && wave_valid) {  // rows of the last step
            FlushRegs<N> fr;
            emit_flush_load<N>(tile, plan, fr);
            emit_flush_store<N>(p, io, plan, fr, (size_t)(io.T - 1) * p.B + wave_b0);
        }
        if (live) env_store<N>(p, b, t, e, false);
        __syncthreads();   // (3) everybody is done: NET writes hidden states and epsilon back
        return;
    }

    // ---------------------------------------------------------------------------------------------------- NET role
    const int w = wv;
    const int crow = (lane >> 4) * 4, ccol = lane & 15, col = 16 * w + ccol;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const unsigned ulane = lane;
    const BFrag b1 = load_bfrag(pio.w, HOFF_W1, w, ulane);
    BFrag bg[6][2];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bg[2 * q][ks] = load_bfrag(pio.w, HOFF_WIH, (w + 4 * q) * 2 + ks, ulane);
            bg[2 * q + 1][ks] = load_bfrag(pio.w, HOFF_WHH, (w + 4 * q) * 2 + ks, ulane);
        }
    BFrag b2[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) b2[ks] = load_bfrag(pio.w, HOFF_W2, w * 2 + ks, ulane);
    const BFrag b3 = load_bfrag(pio.w, HOFF_W3, w, ulane);
    const float bias1 = pio.w[HOFF_B1 + col], bias2 = pio.w[HOFF_B2 + col];
    const float bir = pio.w[HOFF_BIH + col], biz = pio.w[HOFF_BIH + 64 + col], bin = pio.w[HOFF_BIH + 128 + col];
    const float bhr = pio.w[HOFF_BHH + col], bhz = pio.w[HOFF_BHH + 64 + col], bhn = pio.w[HOFF_BHH + 128 + col];
    const int srow = threadIdx.x >> 4, kcol = threadIdx.x & 15;   // staging: 16 threads per row
    // hidden state and last actions of both groups' rows -> LDS
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int gb0 = b0 + RP_GROUP_ENVS * g;
        const int rows_valid = (p.B - gb0 < RP_GROUP_ENVS ? (p.B - gb0 > 0 ? p.B - gb0 : 0) : RP_GROUP_ENVS) * N;
        _Float16 *hs_hi = hs_base + (size_t)g * 2 * ROWS_G * HST, *hs_lo = hs_hi + ROWS_G * HST;
        float *s_h = sh_base + (size_t)g * ROWS_G * LDW;
#pragma unroll
        for (int m = 0; m < TG; m++) {
            const int r = 16 * m + srow;
            const size_t grow = (size_t)gb0 * N + (r < rows_valid ? r : 0);
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rows_valid > 0) hv = *reinterpret_cast<const float4 *>(pio.hidden + grow * H + 4 * kcol);
            *reinterpret_cast<float4 *>(s_h + r * LDW + 4 * kcol) = hv;
            split_store(hs_hi, hs_lo, r * HST + 4 * kcol + 0, hv.x);
            split_store(hs_hi, hs_lo, r * HST + 4 * kcol + 1, hv.y);
            split_store(hs_hi, hs_lo, r * HST + 4 * kcol + 2, hv.z);
            split_store(hs_hi, hs_lo, r * HST + 4 * kcol + 3, hv.w);
        }
        for (int r = threadIdx.x; r < ROWS_G; r += 256) s_act[g][r] = r < rows_valid ? (int)pio.last[(size_t)gb0 * N + r] : -1;
    }
    __syncthreads();   // (1)
    __syncthreads();   // (2) the tiles hold the initial observations
    for (int ph = 0; ph < 2 * io.T; ph++) {
        const int g = ph & 1, s = ph >> 1;                  // NET(group g, step s); pass index of every rendezvous counter = ph
        const int gb0 = b0 + RP_GROUP_ENVS * g;
        const int rows_valid = (p.B - gb0 < RP_GROUP_ENVS ? (p.B - gb0 > 0 ? p.B - gb0 : 0) : RP_GROUP_ENVS) * N;
        _Float16 *hs_hi = hs_base + (size_t)g * 2 * ROWS_G * HST, *hs_lo = hs_hi + ROWS_G * HST;
        float *s_h = sh_base + (size_t)g * ROWS_G * LDW;
        rp_wait_ge(&sy.env_done[2 * g], s);                 // the group's tiles hold the observation after its step s - 1
        rp_wait_ge(&sy.env_done[2 * g + 1], s);
        // ---- x = obs(4) | one_hot(last action) | one_hot(agent id) per row (agent.py:41-52), zero up to column 32
#pragma unroll
        for (int m = 0; m < TG; m++) {
            const int r = 16 * m + srow, el = r / N, ag = r - el * N;   // el: env within the group (0..7; padding rows beyond)
            float v = 0.0f;
            if (r < GR) {
                if (kcol < 4) v = tiles[2 * g + (el >> 2)].row[el & 3][4 * ag + kcol];
                else if (kcol < 4 + NA) v = (kcol - 4 == s_act[g][r]) ? 1.0f : 0.0f;
                else if (kcol < in_dim) v = (kcol - 4 - NA == ag) ? 1.0f : 0.0f;
            }
            split_store(a_hi, a_lo, r * HST + kcol, r < rows_valid ? v : 0.0f);
            split_store(a_hi, a_lo, r * HST + kcol + 16, 0.0f);
        }
        rp_net_sync(&sy.nsync[0], ph, lane);
#pragma unroll
        for (int m = 0; m < TG; m++) {   // h1 = relu(W1 x + b1), columns 16w..16w+15 of every row tile
            f32x4 hi = zero, lo = zero;
            h8 ah, al;
            load_afrag(a_hi, a_lo, 16 * m, 0, lane, ah, al);
            mfma_split(ah, al, b1, hi, lo);
#pragma unroll
            for (int r = 0; r < 4; r++)
                split_store(b_hi, b_lo, (16 * m + crow + r) * HST + col, fmaxf(split_sum(hi[r], lo[r]) + bias1, 0.0f));
        }
        rp_net_sync(&sy.nsync[1], ph, lane);
        {   // GRUCell: per row tile the six chains in k_policy_h's order
            f32x4 hnew[TG];
#pragma unroll
            for (int m = 0; m < TG; m++) {
                f32x4 hi[6], lo[6];
#pragma unroll
                for (int c = 0; c < 6; c++) hi[c] = lo[c] = zero;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    h8 xh, xl, hh, hl;
                    load_afrag(b_hi, b_lo, 16 * m, ks, lane, xh, xl);
                    load_afrag(hs_hi, hs_lo, 16 * m, ks, lane, hh, hl);
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        mfma_split(xh, xl, bg[2 * q][ks], hi[2 * q], lo[2 * q]);
                        mfma_split(hh, hl, bg[2 * q + 1][ks], hi[2 * q + 1], lo[2 * q + 1]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float ir = split_sum(hi[0][r], lo[0][r]), hr = split_sum(hi[1][r], lo[1][r]);
                    const float iz = split_sum(hi[2][r], lo[2][r]), hz = split_sum(hi[3][r], lo[3][r]);
                    const float in_ = split_sum(hi[4][r], lo[4][r]), hn_ = split_sum(hi[5][r], lo[5][r]);
                    const float rg = sigmoidf_((ir + bir) + (hr + bhr));
                    const float zg = sigmoidf_((iz + biz) + (hz + bhz));
                    const float ng = tanhf_((in_ + bin) + rg * (hn_ + bhn));
                    hnew[m][r] = (1.0f - zg) * ng + zg * s_h[(16 * m + crow + r) * LDW + col];
                    split_store(a_hi, a_lo, (16 * m + crow + r) * HST + col, hnew[m][r]);
                }
            }
            rp_net_sync(&sy.nsync[2], ph, lane);   // every NET wavefront has finished reading s_h / hs
#pragma unroll
            for (int m = 0; m < TG; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    s_h[(16 * m + crow + r) * LDW + col] = hnew[m][r];
                    split_store(hs_hi, hs_lo, (16 * m + crow + r) * HST + col, hnew[m][r]);
                }
        }
#pragma unroll
        for (int m = 0; m < TG; m++) {   // f = relu(W2 h' + b2)
            f32x4 hi = zero, lo = zero;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                h8 ah, al;
                load_afrag(a_hi, a_lo, 16 * m, ks, lane, ah, al);
                mfma_split(ah, al, b2[ks], hi, lo);
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
                split_store(b_hi, b_lo, (16 * m + crow + r) * HST + col, fmaxf(split_sum(hi[r], lo[r]) + bias2, 0.0f));
        }
        rp_net_sync(&sy.nsync[3], ph, lane);   // f complete; the a planes (h') are no longer needed: their space takes the partial q
#pragma unroll
        for (int m = 0; m < TG; m++) {   // partial q over this wavefront's 16 of the 64 k
            f32x4 hi = zero, lo = zero;
            h8 ah = {0, 0, 0, 0, 0, 0, 0, 0}, al = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((lane >> 4) < 2) {
                const int idx = (16 * m + (lane & 15)) * HST + 16 * w + 8 * (lane >> 4);
                ah = *reinterpret_cast<const h8 *>(b_hi + idx);
                al = *reinterpret_cast<const h8 *>(b_lo + idx);
            }
            mfma_split(ah, al, b3, hi, lo);
#pragma unroll
            for (int r = 0; r < 4; r++) s_q[w * (ROWS_G * 17) + (16 * m + crow + r) * 17 + ccol] = split_sum(hi[r], lo[r]);
        }
        rp_net_sync(&sy.nsync[4], ph, lane);
        if (w == 0) {   // argmax / epsilon-greedy, one thread per row (GR <= 64: one wavefront)
            const int r = lane;
            if (r < GR) {
                auto qf = [&](int a) {
                    const int o = r * 17 + a;
                    return ((s_q[o] + s_q[ROWS_G * 17 + o]) + (s_q[2 * ROWS_G * 17 + o] + s_q[3 * ROWS_G * 17 + o])) + s_b3[a];
                };
                const unsigned long long grow = pio.row0 + (unsigned long long)(gb0 * N + r);
                const int er = r / N;   // the row's env within the group
                const double eps = s_eps[RP_GROUP_ENVS * g + er];
                const int act = select_action(qf, NA, pio.select, (float)eps, pio.seed, pio.step0 + (unsigned)s, grow);
                s_act[g][r] = act;
                if (r < rows_valid) {
                    pio.actions[((size_t)s * p.B + gb0) * N + r] = act;
                    if (pio.trace && r == er * N) pio.trace[(size_t)s * p.B + gb0 + er] = eps;
                }
            }
            rp_post(&sy.net_done[g], s + 1, lane);
        }
        rp_net_sync(&sy.nsync[5], ph, lane);   // the partial q have been consumed: the a planes may be rewritten
    }
    __syncthreads();   // (3) every ENV wavefront has finished its last step
    if (pio.eps_dev && threadIdx.x < BLOCK / G && b0 + (int)threadIdx.x < p.B) pio.eps_dev[b0 + threadIdx.x] = s_eps[threadIdx.x];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int gb0 = b0 + RP_GROUP_ENVS * g;
        const int rows_valid = (p.B - gb0 < RP_GROUP_ENVS ? (p.B - gb0 > 0 ? p.B - gb0 : 0) : RP_GROUP_ENVS) * N;
        const float *s_h = sh_base + (size_t)g * ROWS_G * LDW;
#pragma unroll
        for (int m = 0; m < TG; m++) {
            const int r = 16 * m + srow;
            if (r < rows_valid)
                *reinterpret_cast<float4 *>(pio.hidden + ((size_t)gb0 * N + r) * H + 4 * kcol) =
                    *reinterpret_cast<const float4 *>(s_h + r * LDW + 4 * kcol);
        }
    }
}

// dynamic LDS of k_rollout_policy_roles<N>: a, b scratch planes + per group hs planes and s_h
template <int N>
constexpr size_t rp_roles_lds() {
    constexpr int ROWS_G = 16 * ((RP_GROUP_ENVS * N + 15) / 16);
    return (size_t)ROWS_G * (4 * HST * 2 + 2 * (2 * HST * 2 + LDW * 4));
}

#endif   // CS_POLICY_F16
