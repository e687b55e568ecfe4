"""TEST INFRASTRUCTURE: float64 oracle environments (oracle/gym_restated.FixedWingOracle) run in a process pool, one worker per
host core, for the long / large parity tests -- the frozen benched kernels through their real 2 000-step time limit, sampled env
ids of a 65 536-env batch in the staggered steady state (tests/test_gpu_oracle_coverage.py, tests/test_emu_fuzz.py).

An oracle env is a pure function of (configuration, seed, GLOBAL env id, reset schedule, action sequence): the trace of env
`e` of a batch can therefore be computed without the other envs, in any process, in any order.  A trace follows the VecEnv
semantics of the product (auto-reset on done, terminal observation, metrics at done; fixed_wing.py:338-437, :287-336 and the
SubprocVecEnv worker loop of examples/train_rl_controller.py:223)."""
import copy
import os

import numpy as np


def usable_cpus():
    """Worker count: the affinity mask capped by the container's CPU quota (the GPU box lists 256 threads and grants 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, n)


def trace_envs(job):
    """Worker: job = dict(config, config_kw, sim_config_kw, seed, env_ids [n], actions float32 [T, n, 3],
    resets {step t: [positions in env_ids] masked-reset BEFORE step t}, curriculum {step t: level} applied before step t,
    anchors {step t: (y [n, 18], dryden [n, 8], gust [n, 6])} = the PRODUCT's simulator state after step t, which the oracle
    simulators of the envs that did not end an episode in that step take over).  Returns the per-step arrays of these envs.

    Anchors: a fixed-wing aircraft under full-scale random commands is a chaotic system once it tumbles -- a float32 and a
    float64 trajectory from the same state separate by e-folds per second (emulator, C3 preset: 63 of 64 envs stay within 4e-3
    for 2 000 steps, the 64th leaves at step 1 434) -- so a run through a 2 000-step time limit re-bases the ORACLE's ODE state
    on the product's every few hundred steps.  What stays independent: every step between two anchors (the one-step 1e-5
    parity tests are the bar on the integration itself), the whole gym half (targets, goal windows, reward, observation
    history, metrics, episode ends, reset draws), which is never re-based."""
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(k, "1")
    from oracle.gym_restated import FixedWingOracle, PhiloxStream
    ids = [int(e) for e in job["env_ids"]]
    acts = np.asarray(job["actions"], dtype=np.float32)
    T, n = acts.shape[0], len(ids)
    envs = []
    for e in ids:
        o = FixedWingOracle(copy.deepcopy(job["config"]), config_kw=copy.deepcopy(job.get("config_kw")),
                            sim_config_kw=copy.deepcopy(job.get("sim_config_kw")))
        o.seed(job["seed"])
        o.rng = PhiloxStream(job["seed"], e)
        o.simulator.env_id = e
        envs.append(o)
    level0 = job.get("curriculum", {}).get(-1)
    if level0 is not None:
        for o in envs:
            o.set_curriculum_level(level0)
    first = [np.asarray(o.reset(), dtype=np.float64) for o in envs]
    D = first[0].size
    k0 = int(job.get("keep_from", 0))   # outputs are kept from this step on (keys and rows count from it)
    out = {"env_ids": ids, "reset_obs": np.stack(first).reshape(n, D),
           "obs": np.zeros((T - k0, n, D)), "reward": np.zeros((T - k0, n)), "done": np.zeros((T - k0, n), dtype=bool),
           "target": np.zeros((T - k0, n, 3)), "term": {}, "term_obs": {}, "metrics": {}, "masked_reset_obs": {}}
    resets = {int(t): list(v) for t, v in job.get("resets", {}).items()}
    for tt in range(T):
        t = tt - k0
        lvl = job.get("curriculum", {}).get(tt)
        if lvl is not None:
            for o in envs:
                o.set_curriculum_level(lvl)
        for j in resets.get(tt, ()):
            out["masked_reset_obs"][(tt, j)] = np.asarray(envs[j].reset(), dtype=np.float64).reshape(-1)
        ended = []
        for j, o in enumerate(envs):
            ob, r, d, info = o.step(acts[tt, j].astype(np.float64))
            if d:
                ended.append(j)
            if t < 0:
                if d:
                    o.reset()
                continue
            out["reward"][t, j], out["done"][t, j] = r, d
            tg = info["target"]
            out["target"][t, j] = [tg[k] for k in tg]   # (config order = FixedWingVecEnv.target_names)
            if d:
                out["term"][(t, j)] = info["termination"]
                out["term_obs"][(t, j)] = np.asarray(ob, dtype=np.float64).reshape(-1)
                out["metrics"][(t, j)] = {k: v for k, v in info.items() if isinstance(v, dict) and k != "target"}
                ob = o.reset()
            out["obs"][t, j] = np.asarray(ob, dtype=np.float64).reshape(-1)
        anc = job.get("anchors", {}).get(tt)
        if anc is not None:
            y, dry, gust = anc
            for j, o in enumerate(envs):
                if j not in ended:
                    sim = o.simulator
                    sim._y = np.asarray(y[j], dtype=np.float64).reshape(1, -1).copy()
                    if sim.turbulence:
                        sim._dry_x = np.asarray(dry[j], dtype=np.float64).reshape(1, -1).copy()
                        sim._gust_now = np.asarray(gust[j], dtype=np.float64).reshape(1, -1).copy()
    return out


def run_traces(config, env_ids, actions, seed, config_kw=None, sim_config_kw=None, resets=None, curriculum=None, workers=None,
               anchors=None, keep_from=0):
    """Traces of `env_ids` (global ids) under actions [T, len(env_ids), 3], split over a spawn pool.  `resets`:
    {step: [positions in env_ids]}.  Returns one merged dict (arrays in the order of env_ids; event dicts keyed (t, position))."""
    import multiprocessing as mp
    env_ids = [int(e) for e in env_ids]
    n = len(env_ids)
    workers = max(1, min(workers or usable_cpus(), n))
    bounds = np.linspace(0, n, workers + 1).astype(int)
    jobs = []
    for w in range(workers):
        lo, hi = int(bounds[w]), int(bounds[w + 1])
        if hi <= lo:
            continue
        sub = {}
        for t, pos in (resets or {}).items():
            loc = [p - lo for p in pos if lo <= p < hi]
            if loc:
                sub[int(t)] = loc
        jobs.append((lo, {"config": config, "config_kw": config_kw, "sim_config_kw": sim_config_kw, "seed": seed,
                          "env_ids": env_ids[lo:hi], "actions": np.ascontiguousarray(actions[:, lo:hi]), "resets": sub,
                          "curriculum": curriculum or {}, "keep_from": int(keep_from),
                          "anchors": {int(t): tuple(np.ascontiguousarray(a[lo:hi]) for a in v) for t, v in (anchors or {}).items()}}))
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"
    try:
        if len(jobs) == 1:
            parts = [trace_envs(jobs[0][1])]
        else:
            with mp.get_context("spawn").Pool(len(jobs)) as pool:
                parts = pool.map(trace_envs, [j for _, j in jobs])
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out = {"env_ids": env_ids}
    for key in ("reset_obs",):
        out[key] = np.concatenate([p[key] for p in parts], axis=0)
    for key in ("obs", "reward", "done", "target"):
        out[key] = np.concatenate([p[key] for p in parts], axis=1)
    for key in ("term", "term_obs", "metrics", "masked_reset_obs"):
        out[key] = {}
        for (lo, _), p in zip(jobs, parts):
            for (t, j), v in p[key].items():
                out[key][(t, j + lo)] = v
    return out


# ----------------------------------------------------------------------------------------------------------------------
# comparison of a recorded product run with the traces
# ----------------------------------------------------------------------------------------------------------------------
def sim_rows(vec, positions=None):
    """(y [n, 18], dryden [n, 8], gust [n, 6]) float64 host copies of the simulator rows of the envs at `positions`."""
    import parity
    g0 = vec.layout.sim >> 2
    rows = vec.state[g0:g0 + 8]
    if positions is not None:
        rows = rows[:, positions if hasattr(rows, "cpu") is False else vec._mem.torch.as_tensor(np.asarray(positions), device=rows.device)]
    w = parity._np(rows).astype(np.float64).transpose(1, 0, 2).reshape(-1, 32)
    return w[:, :18].copy(), w[:, 18:26].copy(), w[:, 26:32].copy()


def record_run(vec, actions, resets=None, positions=None, anchor_every=0):
    """Drives `vec` (as_numpy=True) through actions [T, N, 3] with masked resets {step: [env indices]} BEFORE that step and
    records, for the env indices in `positions` (default all), what trace_envs records for the oracle."""
    import parity
    T, N = actions.shape[0], vec.num_envs
    pos = list(range(N)) if positions is None else [int(p) for p in positions]
    D = vec.obs_dim
    rec = {"reset_obs": parity._np(vec.reset()).reshape(N, D)[pos].astype(np.float64),
           "obs": np.zeros((T, len(pos), D)), "reward": np.zeros((T, len(pos))), "done": np.zeros((T, len(pos)), dtype=bool),
           "target": np.zeros((T, len(pos), 3)), "term": {}, "term_obs": {}, "metrics": {}, "masked_reset_obs": {}, "anchors": {}}
    where = {p: j for j, p in enumerate(pos)}
    for t in range(T):
        if resets and t in resets:
            ob = parity._np(vec.reset(indices=np.asarray(resets[t]))).reshape(N, D)
            for p in resets[t]:
                if int(p) in where:
                    rec["masked_reset_obs"][(t, where[int(p)])] = ob[int(p)].astype(np.float64)
        obs, rew, done, infos = vec.step(actions[t])
        obs, rew, done = parity._np(obs).reshape(N, D), parity._np(rew), parity._np(done).astype(bool)
        rec["obs"][t], rec["reward"][t], rec["done"][t] = obs[pos], rew[pos], done[pos]
        for j, p in enumerate(pos):
            if done[p] or len(pos) <= 512:
                info = infos[p]
                rec["target"][t, j] = [info["target"][k] for k in vec.target_names]
                if done[p]:
                    rec["term"][(t, j)] = info["termination"]
                    rec["term_obs"][(t, j)] = np.asarray(info["terminal_observation"], dtype=np.float64).reshape(-1)
                    rec["metrics"][(t, j)] = {m["name"]: info[m["name"]] for m in vec.cfg.get("metrics", [])}
        if anchor_every and t % anchor_every == anchor_every - 1:
            rec["anchors"][t] = sim_rows(vec, pos)
    return rec


def compare(rec, trace, rtol, atol, metric_rtol=5e-3, what="", check_target=True, max_report=6, max_borderline=2):
    """Everything step()/reset() return, recorded product run against the oracle traces; raises parity.Mismatch listing the
    first deviations.  Returns the worst deviations seen."""
    import math
    import parity
    problems = []

    def note(msg):
        if len(problems) < max_report:
            problems.append(msg)

    worst = {"obs": 0.0, "reward": 0.0, "target": 0.0, "term_obs": 0.0}
    T, n = rec["done"].shape
    ids = trace.get("env_ids", list(range(n)))

    def dev(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        bad_nan = np.isnan(a) != np.isnan(b)
        err = np.abs(np.where(np.isnan(a) | np.isnan(b), 0.0, a - b))
        tol = atol + rtol * np.abs(np.where(np.isnan(b), 0.0, b))
        return err, (err > tol) | bad_nan

    err, bad = dev(rec["reset_obs"], trace["reset_obs"])
    worst["obs"] = max(worst["obs"], float(err.max()))
    for j in np.nonzero(bad.any(axis=1))[0]:
        note("reset obs of env {}: worst |d| {:.3e}".format(ids[j], err[j].max()))
    # everything after the first differing episode end of an env is a different episode: compare up to it.  A differing end whose
    # termination (on the side that ended) is a SIMULATOR VARIABLE is a constraint comparison within float32 rounding of the limit
    # (the state agrees to ~1e-6 right up to it, checked below): expected once per ~10^5 failure ends, tolerated `max_borderline`
    # times per comparison and reported; a differing time-limit / success end is never tolerated
    first_bad = np.full(n, T, dtype=int)
    dd = rec["done"] != trace["done"]
    borderline = []
    for j in np.nonzero(dd.any(axis=0))[0]:
        t = int(np.argmax(dd[:, j]))
        first_bad[j] = t
        name = rec["term"].get((t, j)) if rec["done"][t, j] else trace["term"].get((t, j))
        if name not in (None, "steps", "success", "nan") and len(borderline) < max_borderline:
            borderline.append((t, ids[j], name))
        else:
            note("done differs first at step {} env {} (got {}, want {}; terminations {} / {})".format(
                t, ids[j], rec["done"][t, j], trace["done"][t, j], rec["term"].get((t, j)), trace["term"].get((t, j))))
    worst["borderline_constraint_trips"] = borderline
    valid = np.arange(T)[:, None] < first_bad[None, :]
    for key in ("obs", "reward") + (("target",) if check_target else ()):
        err, bad = dev(rec[key], trace[key])
        v = valid if err.ndim == 2 else valid[:, :, None]
        bad = bad & v
        worst[key] = float(np.where(v, err, 0.0).max()) if err.size else 0.0
        if bad.any():
            idx = np.argwhere(bad)[0]
            t, j = int(idx[0]), int(idx[1])
            g_, w_ = np.asarray(rec[key][t, j], dtype=np.float64).ravel(), np.asarray(trace[key][t, j], dtype=np.float64).ravel()
            k = int(np.nanargmax(np.abs(g_ - w_)))
            note("{} differs first at step {} env {}: entry {} got {!r} want {!r} ({} entries beyond tolerance in that record; "
                 "{} records in all)".format(key, t, ids[j], k, float(g_[k]), float(w_[k]),
                                             int(np.asarray(bad[t, j]).sum()), int(np.asarray(bad).reshape(T, n, -1).any(axis=2).sum())))
    for (t, j), name in sorted(trace["term"].items()):
        if t >= first_bad[j]:
            continue
        if rec["term"].get((t, j)) != name:
            note("step {} env {}: termination {} vs {}".format(t, ids[j], rec["term"].get((t, j)), name))
            continue
        err, bad = dev(rec["term_obs"][(t, j)], trace["term_obs"][(t, j)])
        worst["term_obs"] = max(worst["term_obs"], float(err.max()))
        if bad.any():
            k = int(np.argmax(err))
            note("step {} env {} ({}): terminal observation entry {}: got {!r} want {!r}".format(
                t, ids[j], name, k, float(rec["term_obs"][(t, j)][k]), float(trace["term_obs"][(t, j)][k])))
        for mname, exp in trace["metrics"][(t, j)].items():
            got = rec["metrics"].get((t, j), {}).get(mname)
            if got is None:
                continue
            if set(got.keys()) != set(exp.keys()):
                note("step {} env {} metric {} keys {} vs {}".format(t, ids[j], mname, sorted(got), sorted(exp)))
                continue
            for k in got:
                g, e_ = float(got[k]), float(exp[k])
                # |e0| < 0.01 (the reference itself declares avg_error undefined then, fixed_wing.py:1151): the rise-time thresholds
                # 0.9 |e0| and 0.1 |e0| and the overshoot ratio are comparisons with a number of the size of the float32 rounding
                # of the state they come from -- a crossing exists in one precision and not in the other.  Not compared.
                small_e0 = mname in ("rise_time", "overshoot") and k in trace["metrics"][(t, j)].get("avg_error", {}) and \
                    math.isnan(float(trace["metrics"][(t, j)]["avg_error"][k]))
                if small_e0:
                    continue
                if mname in ("rise_time", "settling_time"):   # integer step indices: a crossing may move by a step in fp32
                    slack = 2.0 if mname == "rise_time" else 1.0     # (rise time = the difference of two crossings)
                    if (math.isnan(g) != math.isnan(e_)) or (not math.isnan(g) and abs(g - e_) > slack):
                        note("step {} env {} metric {}[{}]: {} vs {}   (all metrics of that end: got {} | want {})".format(
                            t, ids[j], mname, k, g, e_, {a: {b: round(float(c), 5) for b, c in v.items()} for a, v in rec["metrics"][(t, j)].items()},
                            {a: {b: round(float(c), 5) for b, c in v.items()} for a, v in trace["metrics"][(t, j)].items()}))
                    continue
                # overshoot / avg_error are ratios to the initial error e0: their condition number grows with the ratio itself
                # (a ratio of 900 means |e0| is a thousandth of the excursion, known to a few fp32 ulps of the state it comes from)
                if mname == "success_time_frac" and not (math.isnan(g) or math.isnan(e_)) and abs(g - e_) <= 0.05:
                    continue   # (a goal flag |e| <= bound within float32 rounding of the bound flips: 1 of 46 steps = 0.022)
                cond = max(1.0, abs(e_) / 5.0) if mname in ("overshoot", "avg_error") and not math.isnan(e_) else 1.0
                if (math.isnan(g) != math.isnan(e_)) or (not math.isnan(g) and abs(g - e_) > max(atol, 1e-3) + metric_rtol * cond * abs(e_)):
                    note("step {} env {} metric {}[{}]: {} vs {}".format(t, ids[j], mname, k, g, e_))
    for key, exp in trace["masked_reset_obs"].items():
        got = rec["masked_reset_obs"].get(key)
        if got is None:
            note("masked reset {} not recorded".format(key))
            continue
        err, bad = dev(got, exp)
        worst["obs"] = max(worst["obs"], float(err.max()))
        if bad.any():
            note("masked reset at step {} env {}: worst |d| {:.3e}".format(key[0], ids[key[1]], err.max()))
    worst["episodes"] = int(trace["done"].sum())
    terms = {}
    for v in trace["term"].values():
        terms[v] = terms.get(v, 0) + 1
    worst["terminations"] = terms
    if problems:
        raise parity.Mismatch("{}: {} deviation(s), first ones:\n  ".format(what, len(problems)) + "\n  ".join(problems) +
                              "\n  worst so far {}".format(worst))
    return worst
