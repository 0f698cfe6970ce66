// cooperative-search_amd/csrc/episodes.hip -- episode batch assembly for the collector / replay buffer (gfx950).
//
// "Next" rows f1 / f2 of SURVEY.md section 8(f).  The reference builds, per episode, eleven [T, ...] arrays with its
// padding rules (common/rollout.py:66-76 and 105-132: steps after termination are zero rows with padded = 1 and
// terminated = 1) and copies them into the replay ring (common/replay_buffer.py:41-61).  The batched collector's
// kernels leave step-major tables ([T+1][B][...] obs / state, [T][B][...] actions / reward / terminated); this file
// turns them into the episode-major, masked, float32 arrays in ONE pass, writing either a fresh [B][T][...] batch or
// straight into the ring slots of a DeviceReplayBuffer.  Pure data movement: HBM-bound.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "coopsearch.h"

namespace {

struct EpisodeParams {
    int B, T, n, A, obs_w, state_w;
    const float *o_tab, *s_tab, *r_tab;
    const int64_t *u_tab;
    const uint8_t *term_tab;
    const int64_t *slot;   // destination episode slot of env b (null: b)
    cs_episode_out out;
};

// a step is real if its env had not terminated before it (the batched env freezes finished envs, so `terminated`
// stays 1 once set): real(t, b) = t == 0 || !terminated[t-1][b]
__device__ __forceinline__ bool step_is_real(const EpisodeParams &p, int t, int b) {
    return t == 0 || p.term_tab[(size_t)(t - 1) * p.B + b] == 0;
}

// rows: out[slot][t][:] = real ? tab[t + shift][b][:] : 0 for the wide keys (o, o_next, s, s_next).
// grid (B, T chunks); every thread moves VEC floats at a time along the row.
template <int VEC>
__global__ __launch_bounds__(256) void k_episode_rows(EpisodeParams p, int t_per_block) {
    const int b = blockIdx.x;
    const size_t slot = p.slot ? (size_t)p.slot[b] : (size_t)b;
    const int t0 = blockIdx.y * t_per_block, t1 = min(p.T, t0 + t_per_block);
    const int ow = p.n * p.obs_w, sw = p.state_w;
    using V = typename std::conditional<VEC == 4, float4, float>::type;
    const int nt = t1 - t0;
    const V zero = {};
    // (t, column) flattened over the block's threads: a 12-float obs row alone would keep 3 threads busy
    for (int i = threadIdx.x; i < nt * (ow / VEC); i += blockDim.x) {
        const int t = t0 + i / (ow / VEC), c = i % (ow / VEC);
        const bool real = step_is_real(p, t, b);
        const V *src0 = reinterpret_cast<const V *>(p.o_tab + ((size_t)t * p.B + b) * ow);
        const V *src1 = reinterpret_cast<const V *>(p.o_tab + ((size_t)(t + 1) * p.B + b) * ow);
        V v0 = zero, v1 = zero;   // (a select between a loaded value and a constant, not between two addresses)
        if (real) {
            v0 = src0[c];
            v1 = src1[c];
        }
        reinterpret_cast<V *>(p.out.o + (slot * p.T + t) * ow)[c] = v0;
        reinterpret_cast<V *>(p.out.o_next + (slot * p.T + t) * ow)[c] = v1;
    }
    // state rows: wavefront = time step, lane = column (no per-element division)
    for (int tt = threadIdx.x >> 6; tt < nt; tt += 4) {
        const int t = t0 + tt;
        const bool real = step_is_real(p, t, b);
        const float *src0 = p.s_tab + ((size_t)t * p.B + b) * sw, *src1 = p.s_tab + ((size_t)(t + 1) * p.B + b) * sw;
        float *d0 = p.out.s + (slot * p.T + t) * sw, *d1 = p.out.s_next + (slot * p.T + t) * sw;
        for (int c = threadIdx.x & 63; c < sw; c += 64) {
            d0[c] = real ? src0[c] : 0.0f;
            d1[c] = real ? src1[c] : 0.0f;
        }
    }
}

// the narrow keys, one thread per (b, t): u, r, avail_u, avail_u_next, u_onehot, padded, terminated
__global__ __launch_bounds__(256) void k_episode_small(EpisodeParams p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)p.B * p.T) return;
    const int b = (int)(i / p.T), t = (int)(i % p.T);
    const size_t slot = p.slot ? (size_t)p.slot[b] : (size_t)b, e = slot * p.T + t;
    const bool real = step_is_real(p, t, b);
    const float rf = real ? 1.0f : 0.0f;
    p.out.r[e] = real ? p.r_tab[(size_t)t * p.B + b] : 0.0f;
    p.out.padded[e] = 1.0f - rf;
    p.out.terminated[e] = real ? (p.term_tab[(size_t)t * p.B + b] ? 1.0f : 0.0f) : 1.0f;
    for (int a = 0; a < p.n; a++) {
        const int act = (int)p.u_tab[((size_t)t * p.B + b) * p.n + a];
        p.out.u[e * p.n + a] = real ? (float)act : 0.0f;
        for (int k = 0; k < p.A; k++) {
            const size_t j = (e * p.n + a) * p.A + k;
            p.out.avail_u[j] = rf;        // every action is always available (flight_env_easy.py:184-188)
            p.out.avail_u_next[j] = rf;
            p.out.u_onehot[j] = (real && k == act) ? 1.0f : 0.0f;
        }
    }
}

thread_local char g_eerr[160] = "";

}  // namespace

extern "C" {

int cs_store_episodes(int B, int T, int n_agents, int n_actions, int obs_w, int state_w, const float *o_tab_dev,
                      const float *s_tab_dev, const int64_t *u_tab_dev, const float *r_tab_dev, const uint8_t *term_tab_dev,
                      const int64_t *slot_dev, const cs_episode_out *out, void *stream) {
    if (B < 1 || T < 1 || n_agents < 1 || n_actions < 1 || obs_w < 1 || state_w < 1 || !o_tab_dev || !s_tab_dev || !u_tab_dev ||
        !r_tab_dev || !term_tab_dev || !out || !out->o || !out->u || !out->s || !out->r || !out->o_next || !out->s_next ||
        !out->avail_u || !out->avail_u_next || !out->u_onehot || !out->padded || !out->terminated) {
        snprintf(g_eerr, sizeof(g_eerr), "cs_store_episodes: bad argument");
        return CS_E_ARG;
    }
    EpisodeParams p{B, T, n_agents, n_actions, obs_w, state_w, o_tab_dev, s_tab_dev, r_tab_dev, u_tab_dev, term_tab_dev,
                    slot_dev, *out};
    hipStream_t s = (hipStream_t)stream;
    // enough blocks to fill the device even for small B: split T when B alone gives fewer than ~2048 blocks
    int chunks = 1;
    while (B * chunks < 2048 && chunks < T) chunks *= 2;
    const int t_per_block = (T + chunks - 1) / chunks;
    const dim3 grid(B, (T + t_per_block - 1) / t_per_block);
    const bool vec4 = (n_agents * obs_w) % 4 == 0 && ((uintptr_t)o_tab_dev % 16 == 0) && ((uintptr_t)out->o % 16 == 0) &&
                      ((uintptr_t)out->o_next % 16 == 0);
    if (vec4)
        hipLaunchKernelGGL(k_episode_rows<4>, grid, dim3(256), 0, s, p, t_per_block);
    else
        hipLaunchKernelGGL(k_episode_rows<1>, grid, dim3(256), 0, s, p, t_per_block);
    hipLaunchKernelGGL(k_episode_small, dim3((unsigned)(((size_t)B * T + 255) / 256)), dim3(256), 0, s, p);
    if (hipGetLastError() != hipSuccess) {
        snprintf(g_eerr, sizeof(g_eerr), "cs_store_episodes: kernel launch failed");
        return CS_E_LAUNCH;
    }
    return CS_OK;
}

const char *cs_episodes_last_error(void) { return g_eerr; }

}  // extern "C"
