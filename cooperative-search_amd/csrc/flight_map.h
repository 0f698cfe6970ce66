// cooperative-search_amd/csrc/flight_map.h -- flight variant: k_map / k_map_update (probability-map update fused with the map part of get_obs) and k_flight_pipe (the sweep of step t beside the step kernel of step t + 1).
// Included by coopsearch.hip inside its anonymous namespace, after k_reset / k_emit (the flight step kernel is k_step<N, 1>).  Not a translation unit of its own.

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
