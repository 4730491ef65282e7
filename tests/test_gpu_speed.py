"""GPU parity of the S-T speed DP (SURVEY.md section 8 row a-ST; reference planner/speed_planning_test.py:38-305).

Bars: generate_st_graph bit-exact; edge costs 1e-12 relative against oracle/st_speed.py (exact_*; the only
non-correctly-rounded operation is pow in the 0.5..1.5 m band) and against the reference's golden vectors;
forward tables of the reference's own speed_DP: cost 1e-12, node index-exact, s_dot bit-exact."""
import numpy as np
import pytest

from tests.conftest import assert_rel, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pl():
    from emplanner_carla_amd.api import Planner
    return Planner(0)


def _params(**kw):
    from emplanner_carla_amd.api import speed_dp_params
    return speed_dp_params(**kw)


def test_st_graph_bit_exact_vs_reference(pl):
    g = load_golden("speed.npz")
    out = pl.st_graph(*g["graph_in"])
    for i in range(4):
        np.testing.assert_array_equal(out[i], g["graph_out"][i])
    out = pl.st_graph(*[a[None] for a in g["graph_mid_in"]])
    for i in range(4):
        np.testing.assert_array_equal(out[i][0], g["graph_mid_out"][i])


def test_start_condition_vs_reference(pl):
    """calc_speed_planning_start_condition (:23-35): the batched entry point and the drop-in function against the imported
    reference's outputs.  The projection of a velocity ACROSS the heading cancels to ~1e-16 in the reference, so the bar
    is relative to the operands (|v|), which is what 1e-6 of a sum of two products can mean."""
    g = load_golden("speed.npz")
    x, want = g["start_in"], g["start_out"]
    s1, s2 = pl.speed_start_condition(*[np.ascontiguousarray(x[:, i]) for i in range(5)])
    assert_rel(s1, want[:, 0], 1e-13, scale=float(np.abs(x[:, :2]).max()))
    assert_rel(s2, want[:, 1], 1e-13, scale=float(np.abs(x[:, 2:4]).max()))
    assert s1[0] == 7.5 and s2[0] == 0.3 and s1[1] == 0.0 and s2[1] == 0.0
    from emplanner_carla_amd.planner import speed_planning_test as sp
    for k in (0, 2, 3, 17):
        a, b = sp.calc_speed_planning_start_condition(*x[k])
        assert a == s1[k] and b == s2[k]
    e1, e2 = pl.speed_start_condition(*[np.zeros(0)] * 5)
    assert e1.shape == (0,) and e2.shape == (0,)


def test_collision_cost_vs_reference(pl):
    g = load_golden("speed.npz")
    got = pl.st_collision_cost(10000000, g["coll_d"])
    assert_rel(got, g["coll_cost"], 1e-13, scale=1.0)
    assert got[1] == 0.0 and got[2] == 0.0 and got[0] == 1e7


def test_edge_costs_vs_reference_golden(pl):
    g = load_golden("speed.npz")
    from emplanner_carla_amd.api import st_grid
    s_list, t_list = st_grid()
    for n, b in enumerate(g["edge_sets"]):
        sets = [g["graph_out"][i, b][None] for i in range(4)]
        e = g["obs_edges"][n]
        edges = np.column_stack((e[:, 0], e[:, 1], np.zeros(len(e)), e[:, 2], e[:, 3]))[None]
        _, obs = pl.st_edge_costs(_params(), edges, *sets)
        assert_rel(obs[0], g["obs_cost"][n], 1e-12, scale=1.0)
        rc = g["dp_idx"][n]
        tab = g["dp_s_dot_table"]
        origin = rc[:, 0] == 0
        edges = np.column_stack((np.where(origin, 0.0, s_list[39 - rc[:, 0]]), np.where(origin, 0.0, t_list[rc[:, 1]]),
                                 np.where(origin, 7.5, tab[rc[:, 0], rc[:, 1]]), s_list[39 - rc[:, 2]],
                                 t_list[rc[:, 3]]))[None]
        tot, _ = pl.st_edge_costs(_params(), edges, *sets)
        assert_rel(tot[0], g["dp_cost"][n], 1e-12, scale=1.0)


def test_forward_tables_vs_reference_golden(pl):
    g = load_golden("speed.npz")
    from oracle import st_speed
    for n in range(len(g["tables_in"])):
        sets = g["tables_in"][n][:64].reshape(4, 1, 16)
        v0 = g["tables_in"][n][64:65]
        kw = dict(zip(("reference_speed", "w_cost_ref_speed", "w_cost_accel", "w_cost_obs"), g["tables_kw"][n]))
        res = pl.speed_dp(_params(**kw), sets[0], sets[1], sets[2], sets[3], v0)
        cost, s_dot, node = g["tables_out"][n]
        assert_rel(res.cost[0], cost, 1e-12, scale=1.0, what=f"cost table {n}")
        np.testing.assert_array_equal(res.node[0], node.astype(np.int32))
        np.testing.assert_array_equal(res.s_dot[0], s_dot)
        r, c = st_speed.terminal_node(cost)
        assert tuple(res.end_node[0]) == (r, c)
        ss, tt = st_speed.backtrack(node, r, c)
        np.testing.assert_array_equal(res.speed_s[0], ss)
        np.testing.assert_array_equal(res.speed_t[0], tt)


def test_speed_dp_batch_vs_exact_oracle(pl):
    """96 scenes with up to 16 obstacle slots through generate_st_graph and the sweep."""
    from emplanner_carla_amd import scenes as S
    from oracle import st_speed
    o = S.make_dynamic_batch(range(200, 296))
    sets = pl.st_graph(*o[:4])
    ex_sets = st_speed.exact_generate_st_graph(*o[:4])
    for i in range(4):
        np.testing.assert_array_equal(sets[i], ex_sets[i])
    res = pl.speed_dp(_params(), *sets, o[4])
    ex = st_speed.exact_speed_dp(*ex_sets, o[4])
    assert_rel(res.cost, ex["cost"], 1e-12, scale=1.0)
    same = res.node == ex["node"]
    # a predecessor may differ only where two candidates tie to within pow()'s last-bit noise
    assert same.mean() > 0.9999, f"node mismatch fraction {1 - same.mean():.2e}"
    if same.all():
        np.testing.assert_array_equal(res.s_dot, ex["s_dot"])
        np.testing.assert_array_equal(res.end_node, ex["end"])
        np.testing.assert_array_equal(res.speed_s, ex["speed_s"])
        np.testing.assert_array_equal(res.speed_t, ex["speed_t"])
    # structure: one node per column up to the terminal column, t samples in order
    for b in range(len(o[4])):
        c = int(res.end_node[b, 1])
        assert np.isfinite(res.speed_s[b, :c + 1]).all() and np.isnan(res.speed_s[b, c + 1:]).all()
        np.testing.assert_array_equal(res.speed_t[b, :c + 1], st_speed.grid()[1][:c + 1])


def test_speed_dp_large_batch_takes_the_heaviest_first_path(pl):
    """Beyond 512 scenes the launcher sorts the scenes by obstacle count (heaviest blocks first) and block i takes scene
    order[i]: every scene's result must equal what it gets in a small batch (no ordering), bit for bit, and the
    exact oracle's on a sample."""
    from emplanner_carla_amd import scenes as S
    from oracle import st_speed
    B = 1536
    o = S.make_dynamic_batch(range(5000, 5000 + B))
    sets = pl.st_graph(*o[:4])
    big = pl.speed_dp(_params(), *sets, o[4])
    for lo in range(0, B, 384):
        sl = slice(lo, lo + 384)
        small = pl.speed_dp(_params(), *[a[sl] for a in sets], o[4][sl])
        for name in ("cost", "s_dot", "node", "end_node", "speed_s", "speed_t"):
            np.testing.assert_array_equal(getattr(big, name)[sl], getattr(small, name), err_msg=f"{name}, scenes {lo}..")
    pick = np.arange(0, B, 48)
    ex = st_speed.exact_speed_dp(*[a[pick] for a in sets], o[4][pick])
    assert_rel(big.cost[pick], ex["cost"], 1e-12, scale=1.0)
    np.testing.assert_array_equal(big.node[pick], ex["node"])
    np.testing.assert_array_equal(big.end_node[pick], ex["end"])


def test_speed_dp_refuses_a_negative_obstacle_weight(pl):
    """w_cost_obs ** (1.5 - d) is complex for a negative base and the reference fails on its next comparison
    (speed_planning_test.py:281); the library refuses the call."""
    nan16 = np.full((1, 16), np.nan)
    with pytest.raises(Exception):
        pl.speed_dp(_params(w_cost_obs=-1.0), nan16, nan16, nan16, nan16, np.zeros(1))
    res = pl.speed_dp(_params(w_cost_obs=0.0), np.full((1, 16), 10.0), np.full((1, 16), 20.0), np.full((1, 16), 1.0),
                      np.full((1, 16), 5.0), np.array([5.0]))        # 0 ** y = 0: a legal, obstacle-blind run
    from oracle import st_speed
    ex = st_speed.exact_speed_dp(np.full((1, 16), 10.0), np.full((1, 16), 20.0), np.full((1, 16), 1.0), np.full((1, 16), 5.0),
                                 np.array([5.0]), w_cost_obs=0.0)
    np.testing.assert_array_equal(res.cost, ex["cost"])
    np.testing.assert_array_equal(res.node, ex["node"])


def test_speed_dp_device_pointers_and_no_tables(pl):
    import torch
    from emplanner_carla_amd import scenes as S
    o = S.make_dynamic_batch(range(300, 364))
    sets = pl.st_graph(*o[:4])
    host = pl.speed_dp(_params(), *sets, o[4])
    dev_in = [torch.from_numpy(a).cuda() for a in (*sets, o[4])]
    dev = pl.speed_dp(_params(), *dev_in, tables=False)
    pl.synchronize()
    assert dev.cost is None and dev.node is None
    np.testing.assert_array_equal(dev.end_node.cpu().numpy(), host.end_node)
    np.testing.assert_array_equal(dev.speed_s.cpu().numpy(), host.speed_s)
    np.testing.assert_array_equal(dev.speed_t.cpu().numpy(), host.speed_t)


def test_speed_dp_edge_cases(pl):
    """No obstacles at all; all slots NaN; start speed 0; many slots (64)."""
    from oracle import st_speed
    nan16 = np.full((3, 16), np.nan)
    v0 = np.array([0.0, 7.0, 30.0])
    res = pl.speed_dp(_params(), nan16, nan16, nan16, nan16, v0)
    ex = st_speed.exact_speed_dp(nan16, nan16, nan16, nan16, v0)
    np.testing.assert_array_equal(res.cost, ex["cost"])          # no pow involved: bit-exact
    np.testing.assert_array_equal(res.node, ex["node"])
    np.testing.assert_array_equal(res.speed_s, ex["speed_s"])
    rng = np.random.default_rng(5)
    s_in = rng.uniform(5, 50, (2, 64))
    s_out = s_in + rng.uniform(0, 20, (2, 64))
    t_in = rng.uniform(0, 6, (2, 64))
    t_out = t_in + rng.uniform(1, 6, (2, 64))
    s_in[:, ::3] = np.nan
    v0 = np.array([3.0, 12.0])
    res = pl.speed_dp(_params(), s_in, s_out, t_in, t_out, v0)
    ex = st_speed.exact_speed_dp(s_in, s_out, t_in, t_out, v0)
    assert_rel(res.cost, ex["cost"], 1e-12, scale=1.0)
    assert (res.node == ex["node"]).mean() > 0.999
    with pytest.raises(Exception):
        pl.speed_dp(_params(), np.zeros((1, 65)), np.zeros((1, 65)), np.zeros((1, 65)), np.zeros((1, 65)), np.zeros(1))


def test_dropin_speed_module(pl):
    from emplanner_carla_amd.planner import speed_planning_test as sp
    g = load_golden("speed.npz")
    out = sp.generate_st_graph(*g["graph_in"][:, 5])
    for i in range(4):
        np.testing.assert_array_equal(out[i], g["graph_out"][i, 5])
    s_list, t_list = g["s_list"], g["t_list"]
    assert sp.CalcSTCoordinate(0, 3, s_list, t_list) == (54.5, 2.0)
    with pytest.raises(IndexError):
        sp.CalcSTCoordinate(2.0, 3, s_list, t_list)
    assert sp.CalcCollisionCost(10000000, 0.2) == 1e7
    sets = [g["graph_out"][i, 5] for i in range(4)]
    e = g["obs_edges"][1][3]
    assert abs(sp.CalcObsCost(*e, *sets, 10000000) - g["obs_cost"][1][3]) <= 1e-12 * max(1.0, g["obs_cost"][1][3])
    rc = g["dp_idx"][1][9]
    got = sp.CalcDpCost(int(rc[0]), int(rc[1]), int(rc[2]), int(rc[3]), *sets, 4000, 50, 100, 10000000, 7.5, s_list,
                        t_list, g["dp_s_dot_table"])
    assert abs(got - g["dp_cost"][1][9]) <= 1e-12 * g["dp_cost"][1][9]
    n = 2
    tin = g["tables_in"][n]
    sets = tin[:64].reshape(4, 16)
    ss, tt = sp.speed_DP(sets[0], sets[1], sets[2], sets[3], tin[64])
    from oracle import st_speed
    cost, _, node = g["tables_out"][n]
    r, c = st_speed.terminal_node(cost)
    es, et = st_speed.backtrack(node, r, c)
    np.testing.assert_array_equal(ss, es)
    np.testing.assert_array_equal(tt, et)
    if c != 0:
        with pytest.raises(IndexError):
            sp.speed_DP(sets[0], sets[1], sets[2], sets[3], tin[64], reference_behaviour=True)
