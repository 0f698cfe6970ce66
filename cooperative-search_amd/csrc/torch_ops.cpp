// cooperative-search_amd/csrc/torch_ops.cpp -- thin PyTorch-ROCm op layer over the C ABI of include/coopsearch.h.
//
// "C++/HIP extension ... over caller-owned contiguous device tensors (no hidden allocation on
// the step path; stream = current torch HIP stream; errors -> TORCH_CHECK -> Python RuntimeError carrying the
// reference's message strings)".  Every op below checks device / dtype / contiguity / element counts of its tensors in
// C++, takes the stream from torch, and forwards to the cs_* entry point of libcoopsearch_hip.so -- no kernel lives
// here.  The environment constants travel as the bytes of a `cs_config` in a CPU uint8 tensor (built once by the
// host side, cooperative-search_amd/env.py), so the struct has exactly one definition: the header.
//
// Registered as torch.ops.coopsearch.* (torch.ops.load_library on the in-tree coopsearch_torch.so).  The ctypes
// binding (cooperative-search_amd/_lib.py) stays as the torch-free route to the same C ABI.
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>
#include <torch/types.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "coopsearch.h"

namespace {

using at::Tensor;

const cs_config &config_of(const Tensor &cfg) {
    TORCH_CHECK(cfg.device().is_cpu() && cfg.scalar_type() == at::kByte && cfg.is_contiguous() &&
                    cfg.numel() == (int64_t)sizeof(cs_config),
                "coopsearch: cfg must be a contiguous CPU uint8 tensor of sizeof(cs_config) = ", sizeof(cs_config), " bytes");
    return *reinterpret_cast<const cs_config *>(cfg.data_ptr());
}

void check_dev(const Tensor &t, const char *name, at::ScalarType dt, int64_t numel, const Tensor &state) {
    TORCH_CHECK(t.is_cuda(), "coopsearch: ", name, " must be a GPU tensor");
    TORCH_CHECK(t.device() == state.device(), "coopsearch: ", name, " is on ", t.device(), ", the env state on ", state.device());
    TORCH_CHECK(t.scalar_type() == dt, "coopsearch: ", name, " must be ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), "coopsearch: ", name, " must be contiguous");
    TORCH_CHECK(t.numel() == numel, "coopsearch: ", name, " must have ", numel, " elements, got ", t.numel());
}

struct Shapes {
    int64_t B, n, m, obs_w, state_w;
};

Shapes shapes_of(const cs_config &c) {
    const int64_t cells = (int64_t)c.map_size * c.map_size;
    return {c.batch, c.n_agents, c.n_targets, c.variant == 1 ? cells + 4 : 4, 4 * (int64_t)c.n_agents + 3 * (int64_t)c.n_targets};
}

void check_state(const cs_config &c, const Tensor &state) {
    cs_layout lay;
    TORCH_CHECK(cs_state_layout(&c, &lay) == CS_OK, cs_last_error());
    TORCH_CHECK(state.is_cuda() && state.scalar_type() == at::kByte && state.is_contiguous() &&
                    state.numel() >= (int64_t)lay.total_bytes,
                "coopsearch: state must be a contiguous GPU uint8 tensor of at least ", lay.total_bytes, " bytes");
}

// The cs_* entry points launch on whatever device is current in the process: make the tensors' device current for the
// duration of the call (the temporary lives to the end of the full expression `ok(cs_...(..., stream_of(t)))`) and hand
// over torch's current stream ON THAT DEVICE -- an env on cuda:1 works while cuda:0 is current.
struct StreamOn {
    c10::hip::HIPGuardMasqueradingAsCUDA guard;   // ROCm tensors report DeviceType::CUDA: the plain HIPGuard refuses them
    void *stream;
    explicit StreamOn(const Tensor &t)
        : guard(t.device()), stream(c10::hip::getCurrentHIPStream(t.device().index()).stream()) {}
    operator void *() const { return stream; }
};
StreamOn stream_of(const Tensor &t) { return StreamOn(t); }

void ok(int rc) {
    // an action outside 0..2 is the reference's IndexError (dyaw[act], flight_env_easy.py:262): CS_CHECK_ACTIONS reports it with that wording
    if (rc == CS_E_ARG && strncmp(cs_last_error(), "list index out of range", 23) == 0) TORCH_CHECK_INDEX(false, cs_last_error());
    TORCH_CHECK(rc == CS_OK, cs_last_error());
}

template <class T>
T *opt_ptr(const c10::optional<Tensor> &t) {
    return t.has_value() && t->defined() ? reinterpret_cast<T *>(t->data_ptr()) : nullptr;
}

void check_outputs(const cs_config &c, const Tensor &state, int64_t T, const c10::optional<Tensor> &obs,
                   const c10::optional<Tensor> &state_out) {
    const Shapes s = shapes_of(c);
    if (obs.has_value() && obs->defined()) check_dev(*obs, "obs", at::kFloat, T * s.B * s.n * s.obs_w, state);
    if (state_out.has_value() && state_out->defined()) check_dev(*state_out, "state_out", at::kFloat, T * s.B * s.state_w, state);
}

int64_t state_bytes(const Tensor &cfg) {
    cs_layout lay;
    ok(cs_state_layout(&config_of(cfg), &lay));
    return (int64_t)lay.total_bytes;
}

void env_init(const Tensor &cfg, Tensor state) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    ok(cs_init(&c, state.data_ptr(), stream_of(state)));
}

void env_seed(const Tensor &cfg, Tensor state, const Tensor &seeds) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    check_dev(seeds, "seeds", at::kInt, c.batch, state);   // uint32 values in an int32 tensor
    ok(cs_seed(&c, state.data_ptr(), reinterpret_cast<const uint32_t *>(seeds.data_ptr()), stream_of(state)));
}

// env.reset(init) -- flight_env_easy.py:79-182, flight_env.py:83-191
void env_reset(const Tensor &cfg, Tensor state, const c10::optional<Tensor> &mask, bool init, c10::optional<Tensor> obs,
               c10::optional<Tensor> state_out) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    if (mask.has_value() && mask->defined()) check_dev(*mask, "mask", at::kByte, c.batch, state);
    check_outputs(c, state, 1, obs, state_out);
    ok(cs_reset(&c, state.data_ptr(), opt_ptr<const uint8_t>(mask), init ? 1 : 0, opt_ptr<float>(obs), opt_ptr<float>(state_out),
                stream_of(state)));
}

// CS_CHECK_ACTIONS through the op layer: a caller that has decided says so -- CS_CHECK_ACTIONS in `flags` = on, OP_NO_CHECK_ACTIONS
// (a bit of this layer, stripped before the C ABI sees the flags) = off; with neither, the default applies: on for batches of up to
// 64 envs (debugging sessions; a stream synchronisation costs nothing there), COOPSEARCH_CHECK_ACTIONS=0 / 1 turns the DEFAULT off /
// on for every batch.  (BatchedFlightEnv always decides: its check_actions argument is what runs.)  Under stream capture the check
// is skipped silently -- a captured call cannot synchronise -- see CS_CHECK_ACTIONS in include/coopsearch.h.
constexpr int64_t OP_NO_CHECK_ACTIONS = int64_t(1) << 30;
bool check_actions_default(int64_t batch) {
    static const int mode = [] {
        const char *e = getenv("COOPSEARCH_CHECK_ACTIONS");
        return !e || !*e ? -1 : (e[0] == '0' ? 0 : 1);
    }();
    return mode < 0 ? batch <= 64 : mode != 0;
}

int action_flags(const Tensor &actions, int64_t flags, int64_t batch) {
    TORCH_CHECK(actions.scalar_type() == at::kInt || actions.scalar_type() == at::kLong,
                "coopsearch: actions must be int32 or int64, got ", actions.scalar_type());
    const bool decided = (flags & (CS_CHECK_ACTIONS | OP_NO_CHECK_ACTIONS)) != 0;
    const bool check = decided ? (flags & OP_NO_CHECK_ACTIONS) == 0 : check_actions_default(batch);
    return (int)(flags & ~(int64_t)(CS_ACTIONS_I64 | CS_CHECK_ACTIONS | OP_NO_CHECK_ACTIONS)) |
           (actions.scalar_type() == at::kLong ? CS_ACTIONS_I64 : 0) | (check ? CS_CHECK_ACTIONS : 0);
}

// env.step(act_list) -- flight_env_easy.py:303-314, flight_env.py:357-368
void env_step(const Tensor &cfg, Tensor state, const Tensor &actions, int64_t flags, Tensor reward, Tensor terminated, Tensor win,
              c10::optional<Tensor> obs, c10::optional<Tensor> state_out) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    const Shapes s = shapes_of(c);
    TORCH_CHECK(actions.dim() >= 1 && actions.size(-1) == s.n, "Act num mismatch agent");   // flight_env_easy.py:256-257
    const int f = action_flags(actions, flags, s.B);
    check_dev(actions, "actions", actions.scalar_type(), s.B * s.n, state);
    check_dev(reward, "reward", at::kFloat, s.B, state);
    check_dev(terminated, "terminated", at::kByte, s.B, state);
    check_dev(win, "win", at::kByte, s.B, state);
    check_outputs(c, state, 1, obs, state_out);
    ok(cs_step(&c, state.data_ptr(), actions.data_ptr(), f, reward.data_ptr<float>(), terminated.data_ptr<uint8_t>(),
               win.data_ptr<uint8_t>(), opt_ptr<float>(obs), opt_ptr<float>(state_out), stream_of(state)));
}

// T consecutive env.step calls from one call (cs_rollout)
void env_rollout(const Tensor &cfg, Tensor state, const Tensor &actions, int64_t flags, Tensor reward, Tensor terminated,
                 Tensor win, c10::optional<Tensor> obs, c10::optional<Tensor> state_out) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    const Shapes s = shapes_of(c);
    TORCH_CHECK(actions.dim() == 3 && actions.size(1) == s.B && actions.size(2) == s.n,
                "coopsearch: rollout actions must be [T, ", s.B, ", ", s.n, "]");
    const int64_t T = actions.size(0);
    TORCH_CHECK(T >= 1, "coopsearch: T must be >= 1");
    const int f = action_flags(actions, flags, s.B);
    check_dev(actions, "actions", actions.scalar_type(), T * s.B * s.n, state);
    check_dev(reward, "reward", at::kFloat, T * s.B, state);
    check_dev(terminated, "terminated", at::kByte, T * s.B, state);
    check_dev(win, "win", at::kByte, T * s.B, state);
    check_outputs(c, state, T, obs, state_out);
    ok(cs_rollout(&c, state.data_ptr(), actions.data_ptr(), (int)T, f, reward.data_ptr<float>(), terminated.data_ptr<uint8_t>(),
                  win.data_ptr<uint8_t>(), opt_ptr<float>(obs), opt_ptr<float>(state_out), stream_of(state)));
}

void env_emit(const Tensor &cfg, Tensor state, c10::optional<Tensor> obs, c10::optional<Tensor> state_out) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    check_outputs(c, state, 1, obs, state_out);
    ok(cs_emit(&c, state.data_ptr(), opt_ptr<float>(obs), opt_ptr<float>(state_out), stream_of(state)));
}

// per-device partial sums of the evaluation metrics (runner.py:86-96); out4 += [sum reward, sum win, sum found, count]
void env_metrics(const Tensor &cfg, Tensor state, Tensor out4) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    check_dev(out4, "out4", at::kDouble, 4, state);
    ok(cs_metrics(&c, state.data_ptr(), out4.data_ptr<double>(), stream_of(state)));
}

void mt_advance(const Tensor &cfg, Tensor state, int64_t min_ahead) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    ok(cs_mt_advance(&c, state.data_ptr(), (int)min_ahead, stream_of(state)));
}

void mt_canonical(const Tensor &cfg, Tensor state, Tensor rows_out) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    check_dev(rows_out, "rows_out", at::kInt, c.batch * CS_MT_STRIDE, state);
    ok(cs_mt_canonical(&c, state.data_ptr(), reinterpret_cast<uint32_t *>(rows_out.data_ptr()), stream_of(state)));
}

// ---- caller-side rows: agent network forward, fused closed loop, episode assembly ------------

void check_f32(const Tensor &t, const char *name, int64_t numel, const Tensor &like) {
    TORCH_CHECK(t.is_cuda() && t.device() == like.device(), "coopsearch: ", name, " must be on ", like.device());
    TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), "coopsearch: ", name, " must be contiguous float32");
    TORCH_CHECK(numel < 0 || t.numel() == numel, "coopsearch: ", name, " must have ", numel, " elements, got ", t.numel());
}

int64_t policy_packed_floats() { return (int64_t)cs_policy_packed_floats(); }

// Agents.choose_action (agent/agent.py:33-97) for rows = B * n_agents rows in one launch (cs_policy_forward)
void policy_forward(const Tensor &packed, const Tensor &obs, int64_t obs_stride, int64_t obs_offset,
                    const c10::optional<Tensor> &last, const c10::optional<Tensor> &feat, int64_t rows_per_feat, Tensor hidden,
                    c10::optional<Tensor> q, Tensor actions, int64_t rows, int64_t n_agents, int64_t n_actions, double epsilon,
                    const c10::optional<Tensor> &eps_env, int64_t seed, int64_t step, int64_t row0, int64_t select) {
    check_f32(packed, "packed", (int64_t)cs_policy_packed_floats(), packed);
    check_f32(obs, "obs", -1, packed);
    TORCH_CHECK(rows >= 1 && obs_stride >= 4 && obs_offset >= 0 && obs.numel() >= (rows - 1) * obs_stride + obs_offset + 4,
                "coopsearch: obs does not hold ", rows, " rows of stride ", obs_stride);
    check_f32(hidden, "hidden", rows * 64, packed);
    if (q.has_value() && q->defined()) check_f32(*q, "q", rows * n_actions, packed);
    if (feat.has_value() && feat->defined()) {
        TORCH_CHECK(rows_per_feat >= 1, "coopsearch: rows_per_feat must be >= 1");
        check_f32(*feat, "feat", ((rows + rows_per_feat - 1) / rows_per_feat) * 16, packed);
    }
    if (last.has_value() && last->defined()) check_dev(*last, "last", at::kLong, rows, packed);
    check_dev(actions, "actions", at::kLong, rows, packed);
    if (eps_env.has_value() && eps_env->defined()) {
        TORCH_CHECK(n_agents >= 1 && rows % n_agents == 0, "coopsearch: eps_env needs rows = envs * n_agents");
        check_dev(*eps_env, "eps_env", at::kDouble, rows / n_agents, packed);
    }
    const int rc = cs_policy_forward(packed.data_ptr<float>(), obs.data_ptr<float>(), (int)obs_stride, (int)obs_offset,
                                     opt_ptr<const int64_t>(last), opt_ptr<const float>(feat), (int)rows_per_feat,
                                     hidden.data_ptr<float>(), opt_ptr<float>(q), actions.data_ptr<int64_t>(), (int)rows,
                                     (int)n_agents, (int)n_actions, (float)epsilon, opt_ptr<const double>(eps_env), (uint64_t)seed,
                                     (uint32_t)step, (uint64_t)row0, (int)select, stream_of(packed));
    TORCH_CHECK(rc == CS_OK, cs_policy_last_error());
}

// flight: conv front end of the agent network (network/base_net.py:9-18) on n_maps probability maps
void policy_conv_features(const Tensor &c1w, const Tensor &c1b, const Tensor &c2w, const Tensor &c2b, const Tensor &lw,
                          const Tensor &lb, const Tensor &maps, int64_t map_stride, int64_t n_maps, Tensor feat) {
    check_f32(maps, "maps", -1, maps);
    TORCH_CHECK(n_maps >= 1 && maps.numel() >= (n_maps - 1) * map_stride + 2500, "coopsearch: maps does not hold ", n_maps, " maps");
    check_f32(c1w, "conv1.weight", 4 * 16, maps);
    check_f32(c1b, "conv1.bias", 4, maps);
    check_f32(c2w, "conv2.weight", 4 * 9, maps);
    check_f32(c2b, "conv2.bias", 1, maps);
    check_f32(lw, "linear.weight", 16 * 576, maps);
    check_f32(lb, "linear.bias", 16, maps);
    check_f32(feat, "feat", n_maps * 16, maps);
    const int rc = cs_policy_conv_features(c1w.data_ptr<float>(), c1b.data_ptr<float>(), c2w.data_ptr<float>(), c2b.data_ptr<float>(),
                                           lw.data_ptr<float>(), lb.data_ptr<float>(), maps.data_ptr<float>(), map_stride, (int)n_maps,
                                           feat.data_ptr<float>(), stream_of(maps));
    TORCH_CHECK(rc == CS_OK, cs_policy_last_error());
}

// the exploration schedule of common/rollout.py:35-41,75-76,133-135 as a cs_epsilon: `eps_env` (float64 [B], in / out) carries
// every env's own epsilon across calls; `eps_trace` (float64 [T, B], out) records what each step's selection used
static cs_epsilon schedule_of(double epsilon, c10::optional<Tensor> &eps_env, double anneal, double min_epsilon, bool per_step,
                              c10::optional<Tensor> &eps_trace, int64_t T, int64_t B, const Tensor &like) {
    cs_epsilon e{epsilon, anneal, min_epsilon, per_step ? 1 : 0, 0, nullptr, nullptr};
    if (eps_env.has_value() && eps_env->defined()) {
        check_dev(*eps_env, "eps_env", at::kDouble, B, like);
        e.eps_dev = eps_env->data_ptr<double>();
    }
    if (eps_trace.has_value() && eps_trace->defined()) {
        TORCH_CHECK(e.eps_dev != nullptr, "coopsearch: eps_trace needs eps_env");
        check_dev(*eps_trace, "eps_trace", at::kDouble, T * B, like);
        e.trace_dev = eps_trace->data_ptr<double>();
    }
    return e;
}

// cs_epsilon_step: one step of the schedule for callers that drive policy_forward / env_step themselves
void epsilon_step(const Tensor &cfg, const Tensor &state, int64_t flags, Tensor eps_env, double anneal, double min_epsilon,
                  c10::optional<Tensor> trace_row) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    const Shapes s = shapes_of(c);
    check_dev(eps_env, "eps_env", at::kDouble, s.B, state);
    if (trace_row.has_value() && trace_row->defined()) check_dev(*trace_row, "trace_row", at::kDouble, s.B, state);
    ok(cs_epsilon_step(&c, state.data_ptr(), (int)flags, eps_env.data_ptr<double>(), anneal, min_epsilon, opt_ptr<double>(trace_row),
                       stream_of(state)));
}

// T x (cs_policy_forward -> cs_step) in one launch (cs_rollout_policy; flight_easy, n_agents <= 5)
void rollout_policy(const Tensor &cfg, Tensor state, const Tensor &packed, Tensor hidden, const Tensor &last, int64_t T, int64_t flags,
                    double epsilon, c10::optional<Tensor> eps_env, double anneal, double min_epsilon, bool per_step,
                    c10::optional<Tensor> eps_trace, int64_t seed, int64_t step0, int64_t row0, int64_t select, Tensor actions,
                    Tensor reward, Tensor terminated, Tensor win, c10::optional<Tensor> obs, c10::optional<Tensor> state_out) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    const Shapes s = shapes_of(c);
    TORCH_CHECK(T >= 1, "coopsearch: T must be >= 1");
    check_f32(packed, "packed", (int64_t)cs_policy_packed_floats(), state);
    check_f32(hidden, "hidden", s.B * s.n * 64, state);
    check_dev(last, "last", at::kLong, s.B * s.n, state);
    check_dev(actions, "actions", at::kLong, T * s.B * s.n, state);
    check_dev(reward, "reward", at::kFloat, T * s.B, state);
    check_dev(terminated, "terminated", at::kByte, T * s.B, state);
    check_dev(win, "win", at::kByte, T * s.B, state);
    check_outputs(c, state, T, obs, state_out);
    const cs_epsilon sched = schedule_of(epsilon, eps_env, anneal, min_epsilon, per_step, eps_trace, T, s.B, state);
    ok(cs_rollout_policy(&c, state.data_ptr(), packed.data_ptr<float>(), hidden.data_ptr<float>(), last.data_ptr<int64_t>(), (int)T,
                         (int)flags, &sched, (uint64_t)seed, (uint32_t)step0, (uint64_t)row0, (int)select,
                         actions.data_ptr<int64_t>(), reward.data_ptr<float>(), terminated.data_ptr<uint8_t>(),
                         win.data_ptr<uint8_t>(), opt_ptr<float>(obs), opt_ptr<float>(state_out), stream_of(state)));
}

// flight: T x (conv features of the env's map -> policy forward -> env step) enqueued by one call (cs_rollout_policy_flight)
void rollout_policy_flight(const Tensor &cfg, Tensor state, const Tensor &packed, const Tensor &c1w, const Tensor &c1b,
                           const Tensor &c2w, const Tensor &c2b, const Tensor &lw, const Tensor &lb, Tensor hidden, const Tensor &last,
                           Tensor scratch, int64_t T, int64_t flags, double epsilon, c10::optional<Tensor> eps_env, double anneal,
                           double min_epsilon, bool per_step, c10::optional<Tensor> eps_trace, int64_t seed, int64_t step0,
                           int64_t row0, int64_t select, Tensor actions, Tensor reward, Tensor terminated, Tensor win,
                           c10::optional<Tensor> obs, c10::optional<Tensor> state_out) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    const Shapes s = shapes_of(c);
    TORCH_CHECK(T >= 1, "coopsearch: T must be >= 1");
    check_f32(packed, "packed", (int64_t)cs_policy_packed_floats(), state);
    check_f32(c1w, "conv1.weight", 4 * 16, state);
    check_f32(c1b, "conv1.bias", 4, state);
    check_f32(c2w, "conv2.weight", 4 * 9, state);
    check_f32(c2b, "conv2.bias", 1, state);
    check_f32(lw, "linear.weight", 16 * 576, state);
    check_f32(lb, "linear.bias", 16, state);
    check_f32(hidden, "hidden", s.B * s.n * 64, state);
    check_dev(last, "last", at::kLong, s.B * s.n, state);
    check_f32(scratch, "scratch", s.B * (16 + 4 * s.n), state);
    check_dev(actions, "actions", at::kLong, T * s.B * s.n, state);
    check_dev(reward, "reward", at::kFloat, T * s.B, state);
    check_dev(terminated, "terminated", at::kByte, T * s.B, state);
    check_dev(win, "win", at::kByte, T * s.B, state);
    check_outputs(c, state, T, obs, state_out);
    const cs_epsilon sched = schedule_of(epsilon, eps_env, anneal, min_epsilon, per_step, eps_trace, T, s.B, state);
    ok(cs_rollout_policy_flight(&c, state.data_ptr(), packed.data_ptr<float>(), c1w.data_ptr<float>(), c1b.data_ptr<float>(),
                                c2w.data_ptr<float>(), c2b.data_ptr<float>(), lw.data_ptr<float>(), lb.data_ptr<float>(),
                                hidden.data_ptr<float>(), last.data_ptr<int64_t>(), scratch.data_ptr<float>(), (int)T, (int)flags,
                                &sched, (uint64_t)seed, (uint32_t)step0, (uint64_t)row0, (int)select,
                                actions.data_ptr<int64_t>(), reward.data_ptr<float>(), terminated.data_ptr<uint8_t>(),
                                win.data_ptr<uint8_t>(), opt_ptr<float>(obs), opt_ptr<float>(state_out), stream_of(state)));
}

// common/rollout.py:66-76,105-132 + replay_buffer.py:41-61: step-major tables -> the 11-key episode batch (cs_store_episodes);
// `outs` in the order o, u, s, r, o_next, s_next, avail_u, avail_u_next, u_onehot, padded, terminated
void store_episodes(const Tensor &o_tab, const Tensor &s_tab, const Tensor &u_tab, const Tensor &r_tab, const Tensor &term_tab,
                    const c10::optional<Tensor> &slots, int64_t n_actions, std::vector<Tensor> outs) {
    TORCH_CHECK(u_tab.dim() == 3, "coopsearch: u_tab must be [T, B, n]");
    const int64_t T = u_tab.size(0), B = u_tab.size(1), n = u_tab.size(2);
    TORCH_CHECK(o_tab.dim() == 4 && s_tab.dim() == 3, "coopsearch: o_tab must be [T+1, B, n, w], s_tab [T+1, B, S]");
    const int64_t w = o_tab.size(3), S = s_tab.size(2), A = n_actions;
    check_f32(o_tab, "o_tab", (T + 1) * B * n * w, o_tab);
    check_f32(s_tab, "s_tab", (T + 1) * B * S, o_tab);
    check_dev(u_tab, "u_tab", at::kLong, T * B * n, o_tab);
    check_f32(r_tab, "r_tab", T * B, o_tab);
    TORCH_CHECK(term_tab.scalar_type() == at::kByte || term_tab.scalar_type() == at::kBool, "coopsearch: term_tab must be uint8 / bool");
    check_dev(term_tab, "term_tab", term_tab.scalar_type(), T * B, o_tab);
    if (slots.has_value() && slots->defined()) check_dev(*slots, "slots", at::kLong, B, o_tab);
    TORCH_CHECK(outs.size() == 11, "coopsearch: store_episodes takes the 11 destination tensors");
    const int64_t per_slot[11] = {T * n * w, T * n, T * S, T, T * n * w, T * S, T * n * A, T * n * A, T * n * A, T, T};
    float *ptr[11];
    for (int k = 0; k < 11; k++) {
        check_f32(outs[k], "episode destination", -1, o_tab);
        TORCH_CHECK(outs[k].dim() >= 2 && outs[k].size(1) == T && outs[k].numel() == outs[k].size(0) * per_slot[k] &&
                        outs[k].size(0) >= ((slots.has_value() && slots->defined()) ? 1 : B),
                    "coopsearch: episode destination ", k, " must be [slots, T, ...] of the episode shape");
        ptr[k] = outs[k].data_ptr<float>();
    }
    const cs_episode_out eo{ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5], ptr[6], ptr[7], ptr[8], ptr[9], ptr[10]};
    const int rc = cs_store_episodes((int)B, (int)T, (int)n, (int)A, (int)w, (int)S, o_tab.data_ptr<float>(), s_tab.data_ptr<float>(),
                                     u_tab.data_ptr<int64_t>(), r_tab.data_ptr<float>(),
                                     reinterpret_cast<const uint8_t *>(term_tab.data_ptr()), opt_ptr<const int64_t>(slots), &eo,
                                     stream_of(o_tab));
    TORCH_CHECK(rc == CS_OK, cs_episodes_last_error());
}

int64_t abi_version() { return cs_abi_version(); }

}  // namespace

TORCH_LIBRARY(coopsearch, m) {
    m.def("abi_version() -> int", &abi_version);
    m.def("state_bytes(Tensor cfg) -> int", &state_bytes);
    m.def("env_init(Tensor cfg, Tensor(a!) state) -> ()", &env_init);
    m.def("env_seed(Tensor cfg, Tensor(a!) state, Tensor seeds) -> ()", &env_seed);
    m.def("env_reset(Tensor cfg, Tensor(a!) state, Tensor? mask, bool init, Tensor(b!)? obs, Tensor(c!)? state_out) -> ()", &env_reset);
    m.def("env_step(Tensor cfg, Tensor(a!) state, Tensor actions, int flags, Tensor(b!) reward, Tensor(c!) terminated, "
          "Tensor(d!) win, Tensor(e!)? obs, Tensor(f!)? state_out) -> ()", &env_step);
    m.def("env_rollout(Tensor cfg, Tensor(a!) state, Tensor actions, int flags, Tensor(b!) reward, Tensor(c!) terminated, "
          "Tensor(d!) win, Tensor(e!)? obs, Tensor(f!)? state_out) -> ()", &env_rollout);
    m.def("env_emit(Tensor cfg, Tensor(a!) state, Tensor(b!)? obs, Tensor(c!)? state_out) -> ()", &env_emit);
    m.def("env_metrics(Tensor cfg, Tensor(a!) state, Tensor(b!) out4) -> ()", &env_metrics);
    m.def("mt_advance(Tensor cfg, Tensor(a!) state, int min_ahead) -> ()", &mt_advance);
    m.def("mt_canonical(Tensor cfg, Tensor state, Tensor(a!) rows_out) -> ()", &mt_canonical);
    m.def("policy_packed_floats() -> int", &policy_packed_floats);
    m.def("policy_forward(Tensor packed, Tensor obs, int obs_stride, int obs_offset, Tensor? last, Tensor? feat, int rows_per_feat, "
          "Tensor(a!) hidden, Tensor(b!)? q, Tensor(c!) actions, int rows, int n_agents, int n_actions, float epsilon, Tensor? eps_env, "
          "int seed, int step, int row0, int select) -> ()", &policy_forward);
    m.def("epsilon_step(Tensor cfg, Tensor state, int flags, Tensor(a!) eps_env, float anneal, float min_epsilon, "
          "Tensor(b!)? trace_row) -> ()", &epsilon_step);
    m.def("policy_conv_features(Tensor conv1_w, Tensor conv1_b, Tensor conv2_w, Tensor conv2_b, Tensor lin_w, Tensor lin_b, "
          "Tensor maps, int map_stride, int n_maps, Tensor(a!) feat) -> ()", &policy_conv_features);
    m.def("rollout_policy(Tensor cfg, Tensor(a!) state, Tensor packed, Tensor(b!) hidden, Tensor last, int T, int flags, "
          "float epsilon, Tensor(j!)? eps_env, float anneal, float min_epsilon, bool per_step, Tensor(k!)? eps_trace, int seed, "
          "int step0, int row0, int select, Tensor(c!) actions, Tensor(d!) reward, Tensor(e!) terminated, "
          "Tensor(f!) win, Tensor(g!)? obs, Tensor(h!)? state_out) -> ()", &rollout_policy);
    m.def("rollout_policy_flight(Tensor cfg, Tensor(a!) state, Tensor packed, Tensor c1w, Tensor c1b, Tensor c2w, Tensor c2b, "
          "Tensor lw, Tensor lb, Tensor(b!) hidden, Tensor last, Tensor(i!) scratch, int T, int flags, float epsilon, "
          "Tensor(j!)? eps_env, float anneal, float min_epsilon, bool per_step, Tensor(k!)? eps_trace, int seed, "
          "int step0, int row0, int select, Tensor(c!) actions, Tensor(d!) reward, Tensor(e!) terminated, Tensor(f!) win, "
          "Tensor(g!)? obs, Tensor(h!)? state_out) -> ()", &rollout_policy_flight);
    m.def("store_episodes(Tensor o_tab, Tensor s_tab, Tensor u_tab, Tensor r_tab, Tensor term_tab, Tensor? slots, int n_actions, "
          "Tensor(a!)[] outs) -> ()", &store_episodes);
}
