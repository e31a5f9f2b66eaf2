"""Device-resident self-play (azg_selfplay_*): invariants on the oracle (CPU) and bit-exact HIP-vs-oracle rows (GPU)."""
import numpy as np
import pytest

import oracle_lib as O
from alphazero_gym_amd import _capi

CASES = {
    "pendulum": dict(kw=dict(env_id=2, mode=1, n_sims=30, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=11, tree_id_base=5),
                     desc=(3, [64, 64], 2, "elu"), max_len=4, det=False),
    "cartpole_sampled": dict(kw=dict(env_id=0, mode=0, n_sims=24, c_uct=20.0, gamma=0.97, num_actions=2, seed=12),
                             desc=(4, [64, 64], 2, "relu"), max_len=6, det=False),
    "cartpole_det": dict(kw=dict(env_id=0, mode=0, n_sims=24, c_uct=20.0, gamma=0.97, num_actions=2, seed=13, v_target="on_policy"),
                         desc=(4, [128, 128], 2, "relu"), max_len=7, det=True),
}


def play(engine_cls, case, n_trees=21, steps=9):
    c = CASES[case]
    e = engine_cls(n_trees=n_trees, **c["kw"])
    in_dim, hidden, nd, act = c["desc"]
    e.set_weights(_capi.make_desc(in_dim, hidden, nd, act), O.make_weights(3, in_dim, hidden, nd, scale=2.0))
    e.selfplay_begin(c["max_len"], c["det"], capacity_steps=steps)
    for _ in range(steps):
        e.selfplay_step()
    rows = e.selfplay_rows(clear=True)
    stats = e.selfplay_stats()
    assert e.selfplay_rows().shape[0] == 0   # cleared
    e.close()
    return rows, stats


@pytest.mark.parametrize("case", sorted(CASES))
def test_selfplay_invariants_on_oracle(case):
    rows, (fsum, fcnt, state) = play(O.OracleEngine, case)
    c = CASES[case]
    n_sims = c["kw"]["n_sims"]
    K = 6 if c["kw"]["mode"] == 1 else 2    # ceil(sqrt(30)) root children / two CartPole actions
    so = 3 if c["kw"]["mode"] == 1 else 4
    assert rows.shape == (9 * 21, so + 3 * K + 1)
    counts = rows[:, so + K:so + 2 * K]
    np.testing.assert_array_equal(counts.sum(1), np.full(len(rows), float(n_sims)))
    if c["kw"]["mode"] == 1:
        np.testing.assert_allclose(np.hypot(rows[:, 0], rows[:, 1]), 1.0, atol=1e-6)
        assert (fcnt == 9 // c["max_len"]).all() and (fsum < 0).all()       # Pendulum never terminates: episodes end by length
    else:
        assert fcnt.sum() > 0 and (fsum[fcnt > 0] / fcnt[fcnt > 0] >= 1).all()
    assert np.isfinite(state).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_selfplay_hip_matches_oracle_bit_for_bit(case):
    from alphazero_gym_amd import _native
    a_rows, a_stats = play(_native.HipEngine, case)
    b_rows, b_stats = play(O.OracleEngine, case)
    np.testing.assert_array_equal(a_rows.view(np.uint32), b_rows.view(np.uint32))
    for x, y in zip(a_stats, b_stats):
        np.testing.assert_array_equal(x, y)
