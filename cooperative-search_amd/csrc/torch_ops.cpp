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
#include <c10/hip/HIPStream.h>
#include <torch/library.h>
#include <torch/types.h>

#include <cstring>

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

void *stream_of(const Tensor &state) { return c10::hip::getCurrentHIPStream(state.device().index()).stream(); }

void ok(int rc) { TORCH_CHECK(rc == CS_OK, cs_last_error()); }

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

int action_flags(const Tensor &actions, int64_t flags) {
    TORCH_CHECK(actions.scalar_type() == at::kInt || actions.scalar_type() == at::kLong,
                "coopsearch: actions must be int32 or int64, got ", actions.scalar_type());
    return (int)(flags & ~(int64_t)CS_ACTIONS_I64) | (actions.scalar_type() == at::kLong ? CS_ACTIONS_I64 : 0);
}

// env.step(act_list) -- flight_env_easy.py:303-314, flight_env.py:357-368
void env_step(const Tensor &cfg, Tensor state, const Tensor &actions, int64_t flags, Tensor reward, Tensor terminated, Tensor win,
              c10::optional<Tensor> obs, c10::optional<Tensor> state_out) {
    const cs_config &c = config_of(cfg);
    check_state(c, state);
    const Shapes s = shapes_of(c);
    TORCH_CHECK(actions.dim() >= 1 && actions.size(-1) == s.n, "Act num mismatch agent");   // flight_env_easy.py:256-257
    const int f = action_flags(actions, flags);
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
    const int f = action_flags(actions, flags);
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
}
