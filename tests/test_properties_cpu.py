"""Property-based CPU tests (hypothesis): invariants of the oracle under arbitrary seeds / action sequences, the replay
ring against a brute-force model, and the sharding helper."""
import types

import numpy as np
from hypothesis import given, settings, strategies as st

import cooperative_search_amd as cs
from cooperative_search_amd import dist as csd
from oracle import oracle as orc


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**32 - 1), n=st.integers(1, 8), agent_mode=st.integers(0, 3), target_mode=st.integers(0, 1),
       variant=st.sampled_from(["flight_easy", "flight"]), data=st.data())
def test_oracle_invariants(seed, n, agent_mode, target_mode, variant, data):
    steps = 40 if variant == "flight" else 120
    env = orc.OracleEnv(orc.make_config(variant=variant, n_agents=n, agent_mode=agent_mode, target_mode=target_mode), seed=seed)
    env.reset(init=True)
    acts = data.draw(st.lists(st.lists(st.integers(0, 2), min_size=n, max_size=n), min_size=steps, max_size=steps))
    last_find, total, words = env.target_find, 0, env.words_consumed()
    assert words % 2 == 0
    for t, a in enumerate(acts):
        r, term, win = env.step(a)
        c = env.counters()
        pos, yaw, out = env.agents()
        # integer team reward: -1 move cost, +10 per new target, +100 on completion, -1 per agent on a wall
        new = c["target_find"] - last_find
        assert new >= 0 and r == -1 + 10 * new + (100 if (win and c["target_find"] == 15 and new > 0) else 0) - int(out.sum())
        assert (pos >= 0).all() and (pos <= 50).all() and (yaw >= 0).all() and (yaw <= 2 * np.pi + 1e-12).all()
        assert term == (c["target_find"] >= 15 or c["time_step"] >= 200) and c["time_step"] == t + 1
        assert (env.words_consumed() - words) % 2 == 0       # every consumer takes an even number of MT words
        words, last_find, total = env.words_consumed(), c["target_find"], total + r
        assert c["total_reward"] == total
        s = env.get_state()
        assert s.shape == (4 * n + 45,) and set(np.unique(s[4 * n + 2::3])) <= {0.0, 1.0}
    if variant == "flight":
        m = env.prob_map()
        assert (m >= 0).all() and (m <= 1).all()
        assert np.array_equal(env.get_obs()[0, :2500], m.reshape(-1))


@settings(max_examples=60, deadline=None)
@given(size=st.integers(1, 40), incs=st.lists(st.integers(1, 40), min_size=1, max_size=30))
def test_replay_ring_matches_brute_force_model(size, incs):
    args = types.SimpleNamespace(n_actions=3, n_agents=2, state_shape=5, obs_shape=4, episode_limit=2, conv=False, map_size=50)
    rb = cs.DeviceReplayBuffer(args, size, device="cpu")
    cur, filled = 0, 0
    for inc in incs:
        inc = min(inc, size)
        idx = rb._get_storage_idx(inc)
        # reference rule, spelled out (replay_buffer.py:84-101)
        if cur + inc <= size:
            want, cur = list(range(cur, cur + inc)), cur + inc
        elif cur < size:
            over = inc - (size - cur)
            want, cur = list(range(cur, size)) + list(range(over)), over
        else:
            want, cur = list(range(inc)), inc
        filled = min(size, filled + inc)
        assert list(idx) == want and (rb.current_idx, rb.current_size) == (cur, filled)
        k = min(3, filled)
        ref_latest = list(range(cur - k, cur)) if cur >= k else list(range(filled - (k - cur), filled)) + list(range(cur))
        assert rb.latest_indices(k) == ref_latest


@settings(deadline=None)
@given(B=st.integers(1, 10**6), world=st.integers(1, 64))
def test_shard_partition(B, world):
    spans = [csd.shard(B, r, world) for r in range(world)]
    assert spans[0][0] == 0 and sum(c for _, c in spans) == B
    assert all(o0 + c0 == o1 for (o0, c0), (o1, _) in zip(spans, spans[1:]))
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    off, cnt = spans[world // 2]
    if cnt:
        seeds = csd.seeds_for(off, cnt)
        assert seeds[0] == (20240000 + off) % 2**32 and len(seeds) == cnt


def test_oracle_rollout_region_equals_stepwise():
    """orc_batch_rollout (one OpenMP region per T steps, env-major: bench.py's cpu_baseline leg) writes exactly what T
    orc_batch_step calls write, whatever the thread count."""
    import numpy as np
    from oracle import oracle as orc
    for variant, n, B, T in (("flight_easy", 3, 96, 60), ("flight", 3, 6, 12)):
        cfg = orc.make_config(variant=variant, n_agents=n, time_limit=25)
        seeds = (np.arange(B) + 5).astype(np.uint32)
        acts = np.random.RandomState(0).randint(0, 3, size=(T, B, n)).astype(np.int32)
        a, b = orc.OracleBatch(cfg, B, seeds), orc.OracleBatch(cfg, B, seeds)
        a.reset(init=True)
        b.reset(init=True)
        out = b.rollout(acts, auto_reset=True, freeze_done=False, threads=3)
        for t in range(T):
            r, te, w = a.step(acts[t], auto_reset=True, freeze_done=False, threads=2)
            assert np.array_equal(r, out["reward"][t]) and np.array_equal(te, out["terminated"][t])
            assert np.array_equal(w, out["win"][t])
            assert np.array_equal(a.state, out["state"][t]) and np.array_equal(a.obs, out["obs"][t])
        # walking the table twice in one region == two calls
        c = orc.OracleBatch(cfg, B, seeds)
        c.reset(init=True)
        c.rollout(acts, auto_reset=True, freeze_done=False, threads=1)
        last = c.rollout(acts, auto_reset=True, freeze_done=False, threads=1)
        d = orc.OracleBatch(cfg, B, seeds)
        d.reset(init=True)
        twice = d.rollout(acts, auto_reset=True, freeze_done=False, threads=2, repeat=2)
        for k in ("reward", "state", "obs"):
            assert np.array_equal(last[k], twice[k]), k
