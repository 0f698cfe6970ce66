// examples/closed_loop_demo.cpp -- the caller-side entry points of include/coopsearch.h from plain C++/HIP
// (no Python, no torch): pack a (random) agent network, run whole episodes of network-in-the-loop collection as ONE
// launch (cs_rollout_policy), and assemble the reference's 11-key episode batch (cs_store_episodes).
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/closed_loop_demo.cpp -Lcooperative-search_amd/csrc \
//         -lcoopsearch_hip -Wl,-rpath,$PWD/cooperative-search_amd/csrc -o closed_loop_demo && ./closed_loop_demo
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "coopsearch.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CS_OK_(x) do { int r_ = (x); if (r_ != CS_OK) { fprintf(stderr, "%s: %d %s %s\n", #x, r_, cs_last_error(), cs_policy_last_error()); return 1; } } while (0)

template <typename T>
static T *dev_alloc(size_t count) {
    void *p = nullptr;
    if (hipMalloc(&p, count * sizeof(T)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); }
    return static_cast<T *>(p);
}

int main() {
    const int B = 4096, n = 3, m = 15, T = 200, A = 3, W = 4 * n + 3 * m, in_dim = 4 + A + n;
    const double cx[15] = {5, 2, 7.5, 2.8, 6.9, 5.5, 5.3, 1.8, 3, 4.5, 6.3, 8, 0.9, 9.4, 4.2};
    const double cy[15] = {9.1, 7.5, 7, 8, 8.5, 8, 6.6, 6.8, 5.7, 5, 5.7, 6.7, 8.7, 9, 9.3};
    const double dx[15] = {0.2, 0.3, 0.3, 0.27, 0.25, 0.25, 0.1, 0.28, 0.18, 0.23, 0.31, 0.29, 0.15, 0.21, 0.34};
    const double dy[15] = {0.2, 0.3, 0.26, 0.27, 0.25, 0.25, 0.12, 0.28, 0.18, 0.25, 0.30, 0.28, 0.16, 0.21, 0.33};
    const char *deter = "ftfftfttfftfftf";
    cs_config cfg = {};
    cfg.variant = 0; cfg.n_agents = n; cfg.n_targets = m; cfg.map_size = 50; cfg.view_range = 7; cfg.time_limit = T;
    cfg.velocity = 1; cfg.safe_dist = 1; cfg.detect_prob = 0.9; cfg.force_dist = 3; cfg.force_factor = 0.8;
    for (int j = 0; j < m; j++) { cfg.cx[j] = cx[j]; cfg.cy[j] = cy[j]; cfg.dx[j] = dx[j]; cfg.dy[j] = dy[j]; cfg.deter[j] = deter[j] == 't'; }
    cfg.batch = B;

    // a random network in torch layout (network/base_net.py parameter shapes), packed on the host
    std::mt19937 gen(7);
    std::uniform_real_distribution<float> uni(-0.5f, 0.5f);
    auto rnd = [&](size_t k) { std::vector<float> v(k); for (auto &x : v) x = uni(gen); return v; };
    auto fc1_w = rnd(64 * in_dim), fc1_b = rnd(64), w_ih = rnd(192 * 64), b_ih = rnd(192), w_hh = rnd(192 * 64), b_hh = rnd(192),
         fc2a_w = rnd(64 * 64), fc2a_b = rnd(64), fc2b_w = rnd(A * 64), fc2b_b = rnd(A);
    std::vector<float> packed(cs_policy_packed_floats());
    CS_OK_(cs_policy_pack(fc1_w.data(), fc1_b.data(), w_ih.data(), b_ih.data(), w_hh.data(), b_hh.data(), fc2a_w.data(),
                          fc2a_b.data(), fc2b_w.data(), fc2b_b.data(), in_dim, A, packed.data()));

    cs_layout lay;
    CS_OK_(cs_state_layout(&cfg, &lay));
    void *state = dev_alloc<char>(lay.total_bytes);
    uint32_t *seeds = dev_alloc<uint32_t>(B);
    float *packed_dev = dev_alloc<float>(packed.size()), *hidden = dev_alloc<float>((size_t)B * n * 64);
    int64_t *last = dev_alloc<int64_t>((size_t)B * n), *u_tab = dev_alloc<int64_t>((size_t)T * B * n);
    float *r_tab = dev_alloc<float>((size_t)T * B), *o_tab = dev_alloc<float>((size_t)(T + 1) * B * n * 4),
          *s_tab = dev_alloc<float>((size_t)(T + 1) * B * W);
    uint8_t *term_tab = dev_alloc<uint8_t>((size_t)T * B), *win_tab = dev_alloc<uint8_t>((size_t)T * B);
    double *metrics = dev_alloc<double>(4);
    cs_episode_out ep;   // the reference's episode batch, [B][T][...] float32
    ep.o = dev_alloc<float>((size_t)B * T * n * 4);        ep.o_next = dev_alloc<float>((size_t)B * T * n * 4);
    ep.s = dev_alloc<float>((size_t)B * T * W);            ep.s_next = dev_alloc<float>((size_t)B * T * W);
    ep.u = dev_alloc<float>((size_t)B * T * n);            ep.r = dev_alloc<float>((size_t)B * T);
    ep.avail_u = dev_alloc<float>((size_t)B * T * n * A);  ep.avail_u_next = dev_alloc<float>((size_t)B * T * n * A);
    ep.u_onehot = dev_alloc<float>((size_t)B * T * n * A); ep.padded = dev_alloc<float>((size_t)B * T);
    ep.terminated = dev_alloc<float>((size_t)B * T);

    std::vector<uint32_t> h_seeds(B);
    for (int b = 0; b < B; b++) h_seeds[b] = 20240000u + b;
    HIP_OK(hipMemcpy(seeds, h_seeds.data(), B * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(packed_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    CS_OK_(cs_init(&cfg, state, stream));
    CS_OK_(cs_seed(&cfg, state, seeds, stream));
    double secs = 0;
    // the reference's exploration schedule (common/arguments.py:78-82: epsilon 1 -> 0.05 over 10000 steps, scale 'step'), one
    // epsilon per env, annealed on the device after every executed step and carried from batch to batch in eps_env
    double *eps_env = dev_alloc<double>(B);
    {
        std::vector<double> one(B, 1.0);
        HIP_OK(hipMemcpy(eps_env, one.data(), B * sizeof(double), hipMemcpyHostToDevice));
    }
    const cs_epsilon sched = {1.0, (1.0 - 0.05) / 10000, 0.05, /*per_step=*/1, 0, eps_env, nullptr};
    for (int round = 0; round < 3; round++) {   // three batches of 4096 episodes, epsilon-greedy on the annealing schedule
        HIP_OK(hipMemsetAsync(hidden, 0, (size_t)B * n * 64 * sizeof(float), stream));      // init_hidden
        HIP_OK(hipMemsetAsync(last, 0xff, (size_t)B * n * sizeof(int64_t), stream));        // -1: no last action
        HIP_OK(hipMemsetAsync(metrics, 0, 4 * sizeof(double), stream));
        HIP_OK(hipStreamSynchronize(stream));
        const auto t0 = std::chrono::steady_clock::now();
        CS_OK_(cs_reset(&cfg, state, nullptr, 0, o_tab, s_tab, stream));                    // o[0], s[0]
        CS_OK_(cs_rollout_policy(&cfg, state, packed_dev, hidden, last, T, CS_FREEZE_DONE, &sched, 99, (uint32_t)(round * T), /*row0=*/0, /*select=*/0,
                                 u_tab, r_tab, term_tab, win_tab, o_tab + (size_t)B * n * 4, s_tab + (size_t)B * W, stream));
        CS_OK_(cs_store_episodes(B, T, n, A, 4, W, o_tab, s_tab, u_tab, r_tab, term_tab, nullptr, &ep, stream));
        CS_OK_(cs_metrics(&cfg, state, metrics, stream));
        HIP_OK(hipStreamSynchronize(stream));
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    double h_m[4];
    HIP_OK(hipMemcpy(h_m, metrics, sizeof(h_m), hipMemcpyDeviceToHost));
    std::vector<float> padded((size_t)B * T), onehot((size_t)B * T * n * A);
    HIP_OK(hipMemcpy(padded.data(), ep.padded, padded.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(onehot.data(), ep.u_onehot, onehot.size() * sizeof(float), hipMemcpyDeviceToHost));
    double real = 0, hot = 0;
    for (float v : padded) real += 1.0 - v;
    for (float v : onehot) hot += v;
    double h_eps[2];
    HIP_OK(hipMemcpy(h_eps, eps_env, sizeof(h_eps), hipMemcpyDeviceToHost));
    printf("epsilon of env 0 / env 1 after three episodes: %.6f / %.6f (1 - executed steps x 9.5e-5)\n", h_eps[0], h_eps[1]);
    printf("episodes %.0f  mean episode_reward %.2f  win rate %.4f  mean targets_find %.2f of %d\n", h_m[3], h_m[0] / h_m[3],
           h_m[1] / h_m[3], h_m[2] / h_m[3], m);
    printf("real steps %.0f  one-hot sum %.0f (= real steps x agents: %s)  batch time %.2f ms  %.3g env-step slots/s\n", real, hot,
           hot == real * n ? "yes" : "NO", secs * 1e3, (double)B * T / secs);
    return hot == real * n ? 0 : 2;
}
