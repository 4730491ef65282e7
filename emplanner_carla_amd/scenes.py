"""Deterministic synthetic scenes (reference line + ego + obstacles) for tests and bench.

The reference has no data files: its inputs come from a live CARLA server
(reference test_9.py:380-392 builds the request tuple).  One *scene* here is the content of
that tuple for one planning cycle, synthesised from ``numpy.random.default_rng(seed)``
with ``seed`` = scene index, so every test, the golden-vector generator and ``bench.py``
see the same inputs.  Pure numpy; no GPU code.

Geometry (adapted): the reference line is a circular arc of
``n_ref`` points spaced ``ref_ds`` metres apart with analytic heading and curvature
(what reference planning_utils.py:185-228 would approximate); the ego sits near point
``origin_index``; the planning start ("predicted location", reference test_9.py:390) is
``start_ahead`` metres further on; obstacles are placed in Frenet space relative to the
ego's projection and mapped to x/y through the arc's normal.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass(frozen=True)
class LatticeConfig:
    """Arguments of reference ``DP_algorithm`` (path_planning.py:276-279) plus scene shape."""

    name: str
    row: int
    col: int
    sample_s: float
    sample_l: float
    sampling_res: float
    n_obs: int
    n_ref: int = 61
    ref_ds: float = 2.0
    w_collision_cost: float = 1e12
    w_smooth_cost: tuple = (300.0, 1000.0, 5000.0)
    w_reference_cost: float = 20.0
    obs_length: float = 5.0  # reference test_9.py:192
    obs_width: float = 5.0


#: BASELINE.json configs[0]: plumbing case, no obstacles -> DP bypass branch (path_planning.py:362-363)
CFG1 = LatticeConfig("cfg1_20x5_0obs", row=5, col=20, sample_s=5.0, sample_l=1.0, sampling_res=2, n_obs=0)
#: BASELINE.json configs[1..3]: the metric's lattice (col=40 stations x row=9 lateral samples, 8 obstacles)
CFG2 = LatticeConfig("cfg2_40x9_8obs", row=9, col=40, sample_s=2.5, sample_l=1.5, sampling_res=2, n_obs=8)
#: the reference's own defaults (path_planning.py:277-279) with 3 obstacles, as in the survey probe
CFG_DEFAULT = LatticeConfig("default_6x12_3obs", row=12, col=6, sample_s=15.0, sample_l=1.5, sampling_res=2,
                            n_obs=3)
#: BASELINE.json configs[4] lattice (stress; judged against the exact restatement)
CFG5 = LatticeConfig("cfg5_120x21_16obs", row=21, col=120, sample_s=1.0, sample_l=0.6, sampling_res=1,
                     n_obs=16, n_ref=71)

CONFIGS = {c.name: c for c in (CFG1, CFG2, CFG_DEFAULT, CFG5)}


@dataclass
class Scene:
    """One planning request (one unit of work)."""

    seed: int
    ref: np.ndarray            # (P, 4)  x, y, theta, kappa of the local reference line
    origin_xy: np.ndarray      # (2,)    vehicle location (origin of the s axis)
    start_xy: np.ndarray       # (2,)    predicted location = planning start
    start_v: np.ndarray        # (2,)    velocity at the planning start
    start_a: np.ndarray        # (2,)    acceleration at the planning start
    obs_xy: np.ndarray         # (n_obs, 2) static obstacle positions
    # the same scene expressed directly in Frenet space (inputs of DP_algorithm); these are the
    # *generator's* values, close to but not identical with what the projection functions return
    sl_obs_s: np.ndarray = field(default=None)
    sl_obs_l: np.ndarray = field(default=None)
    sl_start: np.ndarray = field(default=None)  # (4,) s, l, dl, ddl


#: default arc radii: gentle curvature.  The reference projects every obstacle on the tangent line at the FIRST
#: obstacle's match point (quirk of planning_utils.py:413), so its lateral error grows as (delta s)^2 / (2 R);
#: R >= 1500 m keeps that below ~1.7 m over 70 m
GENTLE_ARCS = (1500.0, 6000.0)
#: SURVEY.md section 8(d)'s own radii: the quirk above bends the S-L picture by up to 16 m over 70 m at R = 150 m
SURVEY_ARCS = (150.0, 1000.0)


def _arc(rng, n_ref, ref_ds, radius_range=GENTLE_ARCS):
    radius = rng.uniform(radius_range[0], radius_range[1])
    sign = 1.0 if rng.random() < 0.5 else -1.0
    kappa = sign / radius
    phi0 = rng.uniform(-np.pi, np.pi)
    x0, y0 = rng.uniform(-200.0, 200.0, size=2)
    arc_s = np.arange(n_ref) * ref_ds
    theta = phi0 + kappa * arc_s
    # exact arc: integrate (cos, sin)(phi0 + kappa*s)
    x = x0 + (np.sin(theta) - np.sin(phi0)) / kappa
    y = y0 - (np.cos(theta) - np.cos(phi0)) / kappa
    # keep theta in (-pi, pi] like arctan2 would (reference planning_utils.py:219)
    theta = np.arctan2(np.sin(theta), np.cos(theta))
    ref = np.stack([x, y, theta, np.full(n_ref, kappa)], axis=1)
    return ref, (x0, y0, phi0, kappa)


def _arc_point(arc, s, l):
    x0, y0, phi0, kappa = arc
    th = phi0 + kappa * s
    x = x0 + (np.sin(th) - np.sin(phi0)) / kappa
    y = y0 - (np.cos(th) - np.cos(phi0)) / kappa
    return np.array([x - l * np.sin(th), y + l * np.cos(th)]), th


#: obstacle layouts: "corridor" (default, below), "survey" (SURVEY.md section 8d), "worst" (every obstacle within reach
#: of the same stretch of lattice)
SCENE_DISTS = ("corridor", "survey", "worst")


def _survey_obstacles(rng, n, horizon):
    """``s_k = 12 + 11 k + U(-2, 2)`` metres ahead of the planning start (k = 0..7, all <= 91 m
    on the 100 m horizon; scaled with the horizon for other lattices), ``l_k = +-U(2.5, 5.0)`` with the sign alternating
    with probability 0.7."""
    scale = horizon / 100.0
    k = np.arange(n)
    obs_s = (12.0 + 11.0 * k * (8.0 / max(n, 1)) + rng.uniform(-2.0, 2.0, n)) * scale
    sign = np.empty(n)
    cur = 1.0 if rng.random() < 0.5 else -1.0
    for j in range(n):
        if j and rng.random() < 0.7:
            cur = -cur
        sign[j] = cur
    return obs_s, sign * rng.uniform(2.5, 5.0, n)


def _worst_obstacles(rng, n, horizon):
    """Every obstacle within reach of the same lattice columns: 1.5 m apart from 30 % of the horizon on, alternating
    sides just outside the hard radius of the centre row, so that the centre line stays drivable while every edge of
    some fifteen metres of lattice scans all of them."""
    obs_s = 0.3 * horizon + 1.5 * np.arange(n) + rng.uniform(-0.2, 0.2, n)
    side = np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
    return obs_s, side * rng.uniform(5.2, 5.8, n)


def make_scene(seed: int, cfg: LatticeConfig = CFG2, origin_index: int = 5, start_ahead: float = 2.0,
               blocked_fraction: float = 0.1, dist: str = "corridor", radius_range=GENTLE_ARCS) -> Scene:
    """Scene ``seed``.  ``radius_range``: the arc radius is drawn from U(radius_range) (first draw of the scene's
    generator, so the default leaves every existing scene bit-identical); ``SURVEY_ARCS`` is SURVEY.md 8(d)'s range.

    Obstacle layout: the reference's quirked smoothness cost makes lateral moves beyond
    s ~ 28 m dearer than a collision (5000 * sum(dddl_quirk^2) ~ 1.5e6 * s^4 for one 1.5 m row
    step on the 2.5 m lattice), so a drivable scene is one where a *corridor* - one lattice
    row kept from ~25 m to the horizon - stays clear.  We draw the corridor row first and put
    every obstacle at least the hard radius (4 m, path_planning.py:588) plus the projection
    drift away from it, on a random side; corridors off the centre line force an early swerve.
    ``blocked_fraction`` of the scenes get a three-abreast wall instead (DP "can't find a
    feasible path" branch, path_planning.py:351-352).
    """
    rng = np.random.default_rng(seed)
    ref, arc = _arc(rng, cfg.n_ref, cfg.ref_ds, radius_range)
    radius = 1.0 / abs(arc[3])
    s_origin = origin_index * cfg.ref_ds
    origin_xy, _ = _arc_point(arc, s_origin, rng.normal(0.0, 0.2))

    start_l = rng.uniform(-0.5, 0.5)
    start_dl = rng.uniform(-0.05, 0.05)
    start_ddl = rng.uniform(-0.01, 0.01)
    start_xy, th = _arc_point(arc, s_origin + start_ahead, start_l)
    speed = rng.uniform(5.0, 12.0)
    heading = th + np.arctan(start_dl)
    start_v = speed * np.array([np.cos(heading), np.sin(heading)])
    start_a = rng.uniform(-0.5, 0.5, size=2)

    horizon = cfg.col * cfg.sample_s
    n = cfg.n_obs
    if n:
        # longitudinal: first obstacle ~10 % into the horizon, last one before ~70 % of it - the
        # reference's bound builder indexes argmin+2 unchecked (path_planning.py:240-241,265-272)
        # and the QP pins the end state to (0, 0, 0), so the last stations must stay unbounded
        pitch = 0.6 * horizon / n
        obs_s = start_ahead + 0.08 * horizon + pitch * np.arange(n) + rng.uniform(-0.18, 0.18, n) * pitch
        half = cfg.row // 2
        reach = min(2, half)
        corridor = float(rng.integers(-reach, reach + 1)) * cfg.sample_l if rng.random() < 0.6 else 0.0
        drift = (obs_s - obs_s[0]) ** 2 / (2.0 * radius) + 0.3
        side = np.where(rng.random(n) < 0.5, 1.0, -1.0)
        obs_l = corridor + side * (4.3 + drift + rng.uniform(0.0, 3.0, n))
        if n >= 4 and rng.random() < 0.3:
            # a dodge: the SECOND obstacle sits IN the corridor, far enough out (s ~ 17 m) for the
            # QP's fixed start state to reach the bound and close enough that swerving around it
            # and back is cheaper than the collision weight.  It is offset to the +l side so the
            # cheaper swerve passes on the -l side: the reference's bound builder applies upper
            # bounds in place but lower bounds four stations late once the path is decimated
            # (argmin+2 at path_planning.py:240 plus back_index at :131), and late lower bounds
            # collide with the next obstacle's upper bound.  Its neighbours move off-road.
            corridor = max(corridor, -1.0 * cfg.sample_l)
            obs_l = corridor + side * (4.3 + drift + rng.uniform(0.0, 3.0, n))
            obs_l[1] = corridor + rng.uniform(0.55, 1.3)
            for k in (0, 2):
                obs_l[k] = corridor + side[k] * (8.0 + drift[k] + rng.uniform(0.0, 2.0))
        if dist == "survey":
            rel_s, obs_l = _survey_obstacles(rng, n, horizon)
            obs_s = start_ahead + rel_s
        elif dist == "worst":
            rel_s, obs_l = _worst_obstacles(rng, n, horizon)
            obs_s = start_ahead + rel_s
        elif dist != "corridor":
            raise ValueError(f"unknown scene distribution {dist!r}: one of {SCENE_DISTS}")
        if n >= 3 and dist != "worst" and rng.random() < blocked_fraction:
            # a wall: three obstacles abreast cover every lattice row within the 4 m hard radius
            k = int(rng.integers(0, n - 2))
            obs_s[k:k + 3] = obs_s[k + 1]
            obs_l[k:k + 3] = np.array([-4.0, 0.0, 4.0]) * (cfg.sample_l * half / 6.0)
        obs_xy = np.stack([_arc_point(arc, s_origin + s, l)[0] for s, l in zip(obs_s, obs_l)])
    else:
        obs_s = np.zeros(0)
        obs_l = np.zeros(0)
        obs_xy = np.zeros((0, 2))
    return Scene(seed=seed, ref=ref, origin_xy=origin_xy, start_xy=start_xy, start_v=start_v,
                 start_a=start_a, obs_xy=obs_xy, sl_obs_s=obs_s, sl_obs_l=obs_l,
                 sl_start=np.array([start_ahead, start_l, start_dl, start_ddl]))


@dataclass
class SceneBatch:
    """Structure-of-arrays view of B scenes: the host-side layout the C-ABI consumes."""

    cfg: LatticeConfig
    seeds: np.ndarray          # (B,)
    ref: np.ndarray            # (B, P, 4)
    origin_xy: np.ndarray      # (B, 2)
    start_xy: np.ndarray       # (B, 2)
    start_v: np.ndarray        # (B, 2)
    start_a: np.ndarray        # (B, 2)
    obs_xy: np.ndarray         # (B, max_obs, 2)
    n_obs: np.ndarray          # (B,) int32
    sl_obs_s: np.ndarray       # (B, max_obs)
    sl_obs_l: np.ndarray       # (B, max_obs)
    sl_start: np.ndarray       # (B, 4)

    def __len__(self):
        return len(self.seeds)


#: planning start of the benchmark batch, metres ahead of the ego: NOT a multiple of ``ref_ds``.  With 2.0 (the
#: fixtures of rounds 1-4) three scenes in four put the start exactly on a reference-line node, where
#: ``s_map[idx + 1] < s`` (reference path_planning.py:63) is decided by the last bit of a dot product
BENCH_START_AHEAD = 2.7


def survey_geometry_kwargs(seed: int) -> dict:
    """Scene options of the "tight" fixture and of bench.py's ``survey_leg``: SURVEY.md 8(d)'s arc radii for every
    scene; even seeds keep the corridor layout (mostly plannable), odd seeds use the survey's own slalom (the
    reference refuses nearly all of them: status paths); every other PAIR of seeds starts off the nodes."""
    return dict(radius_range=SURVEY_ARCS, dist="survey" if seed % 2 else "corridor",
                start_ahead=BENCH_START_AHEAD if (seed // 2) % 2 else 2.0)


def make_batch(seeds, cfg: LatticeConfig = CFG2, per_seed=None, **kw) -> SceneBatch:
    """``per_seed``: optional ``seed -> dict`` of ``make_scene`` options applied on top of ``kw``
    (e.g. ``survey_geometry_kwargs``)."""
    seeds = np.asarray(list(seeds), dtype=np.int64)
    scenes = [make_scene(int(s), cfg, **{**kw, **(per_seed(int(s)) if per_seed else {})}) for s in seeds]
    B = len(scenes)
    mo = max(cfg.n_obs, 1)
    obs_xy = np.zeros((B, mo, 2))
    sl_s = np.zeros((B, mo))
    sl_l = np.zeros((B, mo))
    n_obs = np.zeros(B, dtype=np.int32)
    for i, sc in enumerate(scenes):
        k = len(sc.obs_xy)
        n_obs[i] = k
        obs_xy[i, :k] = sc.obs_xy
        sl_s[i, :k] = sc.sl_obs_s
        sl_l[i, :k] = sc.sl_obs_l
    return SceneBatch(
        cfg=cfg, seeds=seeds,
        ref=np.stack([sc.ref for sc in scenes]),
        origin_xy=np.stack([sc.origin_xy for sc in scenes]),
        start_xy=np.stack([sc.start_xy for sc in scenes]),
        start_v=np.stack([sc.start_v for sc in scenes]),
        start_a=np.stack([sc.start_a for sc in scenes]),
        obs_xy=obs_xy, n_obs=n_obs, sl_obs_s=sl_s, sl_obs_l=sl_l,
        sl_start=np.stack([sc.sl_start for sc in scenes]))


# --------------------------------------------------------------------------------------
# S-T speed DP inputs (BASELINE config 5; SURVEY.md section 8d "cfg5")
# --------------------------------------------------------------------------------------
def make_dynamic_obstacles(seed: int, n_slots: int = 16, n_present: int | None = None):
    """Dynamic obstacles in Frenet space for reference ``generate_st_graph``
    (speed_planning_test.py:38): ``s ~ U(5,50)``, ``l ~ U(-6,6)``, ``s_dot ~ U(0,6)``,
    ``|l_dot| ~ U(0.5,2)`` heading for the lane centre with probability 0.8; one obstacle in ten
    drifts slower than the 0.3 m/s cut (:53) and is ignored by the reference.  Slots past
    ``n_present`` are NaN (the reference stops scanning at the first NaN s, :51).  Returns
    ``(obs_s, obs_l, obs_s_dot, obs_l_dot, plan_start_s_dot)``."""
    rng = np.random.default_rng(1_000_003 + seed)
    k = int(rng.integers(0, n_slots + 1)) if n_present is None else int(n_present)
    s = rng.uniform(5.0, 50.0, n_slots)
    l = rng.uniform(-6.0, 6.0, n_slots)
    s_dot = rng.uniform(0.0, 6.0, n_slots)
    speed = rng.uniform(0.5, 2.0, n_slots)
    toward = rng.uniform(size=n_slots) < 0.8
    l_dot = np.where(toward, -np.sign(l), np.sign(l)) * speed
    slow = rng.uniform(size=n_slots) < 0.1
    l_dot = np.where(slow, rng.uniform(-0.29, 0.29, n_slots), l_dot)
    for a in (s, l, s_dot, l_dot):
        a[k:] = np.nan
    return s, l, s_dot, l_dot, float(rng.uniform(0.0, 15.0))


def make_dynamic_batch(seeds, n_slots: int = 16, n_present: int | None = None):
    """Stack ``make_dynamic_obstacles`` over seeds -> four [B, n_slots] arrays and start speeds [B]."""
    rows = [make_dynamic_obstacles(int(sd), n_slots, n_present) for sd in seeds]
    return tuple(np.stack([r[i] for r in rows]) for i in range(4)) + (np.array([r[4] for r in rows]),)
