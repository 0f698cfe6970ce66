// examples/c_api_demo.cpp -- driving libcoopsearch_hip.so from a plain C++/HIP host program (no Python, no torch).
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/c_api_demo.cpp -Lcooperative-search_amd/csrc -lcoopsearch_hip \
//         -Wl,-rpath,$PWD/cooperative-search_amd/csrc -o c_api_demo && ./c_api_demo
//
// 4096 flight_easy environments (3 agents, 15 targets from the shipped target table), one episode of 200 steps with
// host-generated uniform actions, then the evaluation metrics of runner.py:86-96 from cs_metrics.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "coopsearch.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CS_OK_(x) do { int r_ = (x); if (r_ != CS_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, cs_last_error()); return 1; } } while (0)

int main() {
    const int B = 4096, n = 3, m = 15, T = 200;
    // flight_targets.txt as main.py:19-32 parses it
    const double cx[15] = {5, 2, 7.5, 2.8, 6.9, 5.5, 5.3, 1.8, 3, 4.5, 6.3, 8, 0.9, 9.4, 4.2};
    const double cy[15] = {9.1, 7.5, 7, 8, 8.5, 8, 6.6, 6.8, 5.7, 5, 5.7, 6.7, 8.7, 9, 9.3};
    const double dx[15] = {0.2, 0.3, 0.3, 0.27, 0.25, 0.25, 0.1, 0.28, 0.18, 0.23, 0.31, 0.29, 0.15, 0.21, 0.34};
    const double dy[15] = {0.2, 0.3, 0.26, 0.27, 0.25, 0.25, 0.12, 0.28, 0.18, 0.25, 0.30, 0.28, 0.16, 0.21, 0.33};
    const char *deter = "ftfftfttfftfftf";
    cs_config cfg = {};
    cfg.variant = 0; cfg.n_agents = n; cfg.n_targets = m; cfg.map_size = 50; cfg.view_range = 7; cfg.time_limit = T;
    cfg.agent_mode = 0; cfg.target_mode = 0;
    cfg.velocity = 1; cfg.safe_dist = 1; cfg.detect_prob = 0.9; cfg.force_dist = 3; cfg.force_factor = 0.8;
    for (int j = 0; j < m; j++) { cfg.cx[j] = cx[j]; cfg.cy[j] = cy[j]; cfg.dx[j] = dx[j]; cfg.dy[j] = dy[j]; cfg.deter[j] = deter[j] == 't'; }
    cfg.batch = B;

    cs_layout lay;
    CS_OK_(cs_state_layout(&cfg, &lay));
    void *state; uint32_t *seeds; int32_t *actions; float *reward, *obs, *st; uint8_t *term, *win; double *metrics;
    HIP_OK(hipMalloc(&state, lay.total_bytes));
    HIP_OK(hipMalloc(&seeds, B * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&actions, (size_t)T * B * n * sizeof(int32_t)));
    HIP_OK(hipMalloc(&reward, (size_t)T * B * sizeof(float)));
    HIP_OK(hipMalloc(&term, (size_t)T * B));
    HIP_OK(hipMalloc(&win, (size_t)T * B));
    HIP_OK(hipMalloc(&obs, (size_t)B * n * 4 * sizeof(float)));
    HIP_OK(hipMalloc(&st, (size_t)B * (4 * n + 3 * m) * sizeof(float)));
    HIP_OK(hipMalloc(&metrics, 4 * sizeof(double)));
    std::vector<uint32_t> h_seeds(B);
    for (int b = 0; b < B; b++) h_seeds[b] = 20240000u + b;
    std::vector<int32_t> h_act((size_t)T * B * n);
    srand(1);
    for (auto &a : h_act) a = rand() % 3;
    HIP_OK(hipMemcpy(seeds, h_seeds.data(), B * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(actions, h_act.data(), h_act.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(metrics, 0, 4 * sizeof(double)));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    CS_OK_(cs_init(&cfg, state, stream));
    CS_OK_(cs_seed(&cfg, state, seeds, stream));
    CS_OK_(cs_reset(&cfg, state, nullptr, 1, obs, st, stream));                       // env.reset(init=True)
    // one launch per step (what a policy in the loop would do) ...
    for (int t = 0; t < 20; t++)
        CS_OK_(cs_step(&cfg, state, actions + (size_t)t * B * n, CS_FREEZE_DONE, reward, term, win, obs, st, stream));
    // ... or many steps per launch for an open-loop action table
    CS_OK_(cs_rollout(&cfg, state, actions + (size_t)20 * B * n, T - 20, CS_FREEZE_DONE, reward, term, win, nullptr, nullptr, stream));
    CS_OK_(cs_metrics(&cfg, state, metrics, stream));
    HIP_OK(hipStreamSynchronize(stream));
    double h_m[4];
    HIP_OK(hipMemcpy(h_m, metrics, sizeof(h_m), hipMemcpyDeviceToHost));
    printf("episodes %.0f  mean episode_reward %.2f  win rate %.4f  mean targets_find %.2f of %d\n", h_m[3], h_m[0] / h_m[3],
           h_m[1] / h_m[3], h_m[2] / h_m[3], m);
    return 0;
}
