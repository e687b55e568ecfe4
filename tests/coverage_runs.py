"""TEST INFRASTRUCTURE: the two long-form oracle comparisons of round 6, written once for both back ends (the HIP library on a
GPU box, the host-emulation build in the CPU suite):

  * through_time_limit -- a batch through its configuration's REAL time limit (the frozen presets run 2 000-step episodes),
    every output of step()/reset() against pooled oracles (tests/oracle_pool.py);
  * steady_state_sampled -- a large batch brought to the steady state of a long run exactly as bench.py does it (stagger_ages:
    a random 1/parts of the envs reset every steps_max/parts steps), then a window of steps in which every launch mixes ending,
    failing, early-episode and drawing lanes; all outputs of ALL envs are kept on the device, a sample of env ids is chosen
    AFTERWARDS -- every lane whose last step failed on its time-limit step, failure ends, time-limit ends, their wave
    neighbours, uniform picks -- and compared with oracles created by global env id under the same reset schedule."""
import copy

import numpy as np

import oracle_pool as op
import parity


def jumpy_actions(seed, steps, n, scale=1.3, p_jump=0.3):
    rng = np.random.default_rng(seed)
    a = np.zeros((steps, n, 3), dtype=np.float32)
    cur = rng.uniform(-1, 1, size=(n, 3))
    for t in range(steps):
        jump = rng.uniform(size=(n, 1)) < p_jump
        cur = np.where(jump, np.clip(cur + rng.normal(0, 0.4, size=(n, 3)), -scale, scale), cur)
        a[t] = cur
    return a


def through_time_limit(vec, cfg, ckw, skw, seed, steps, anchor_every=250, rtol=4e-3, atol=4e-3, actions=None, workers=None,
                       what=""):
    """`vec`: as_numpy=True, freshly constructed with `seed`.  Returns compare()'s summary."""
    n = vec.num_envs
    acts = jumpy_actions(5, steps, n) if actions is None else actions
    rec = op.record_run(vec, acts, anchor_every=anchor_every)
    tr = op.run_traces(copy.deepcopy(cfg), list(range(vec.env_id_base, vec.env_id_base + n)), acts, seed, config_kw=ckw,
                       sim_config_kw=skw, anchors=rec["anchors"], workers=workers)
    return op.compare(rec, tr, rtol, atol, what=what)


# ----------------------------------------------------------------------------------------------------------------------
def _take(buf, w, pos):
    """buf[w][pos] as a float64 / native host array (torch or numpy buffers)."""
    x = buf[w]
    if hasattr(x, "cpu"):
        import torch
        return x[torch.as_tensor(np.asarray(pos), device=x.device)].cpu().numpy()
    return np.asarray(x[np.asarray(pos)])


def steady_state_sampled(vec, cfg, ckw, skw, seed, window=300, parts=None, sample=256, pool_size=8, anchor_every=10,
                         rtol=4e-3, atol=4e-3, workers=None, what="", metrics_every=100, select_seed=0, first_pick=None):
    """`vec`: device tensors (as_numpy=False) or the emulation backend, auto_reset=True, freshly constructed with `seed`.
    parts = 0: no run-in -- the window starts at the reset of a fresh VecEnv, all episodes in LOCK-STEP (the regime of a real
    run's time-limit ends: whole cohorts of lanes end in one launch, next to lanes that failed earlier and are at other ages).
    Anchors every 10 steps (oracle_pool.trace_envs): full-scale independent random commands on 65 536 aircraft find the
    tumbling / diving ones (Va 43 m/s on its way to the airspeed constraint), whose float32 and float64 trajectories separate by
    1e-2 within fifty steps -- and the lagged rows of an observation carry the drift of before an anchor for eight more steps."""
    from gym_fixed_wing import _native as nat
    N, D, m = vec.num_envs, vec.obs_dim, vec._mem
    steps_max = int(vec.cfg["steps_max"])
    parts = steps_max if parts is None else int(parts)
    per = max(1, steps_max // max(parts, 1))
    rng = np.random.default_rng(1234)
    pool_h = [rng.uniform(-1, 1, (N, 3)).astype(np.float32) for _ in range(pool_size)]
    pool_d = [m.from_host(p) for p in pool_h]
    perm = np.random.RandomState(4321).permutation(N)
    base = vec.env_id_base

    anchors_all = {}

    def rows_all():
        g0 = vec.layout.sim >> 2
        r = vec.state[g0:g0 + 8]
        return r.clone() if hasattr(r, "clone") else r.copy()

    # ---- the run-in: bench.py's stagger_ages
    reset_obs0 = vec.reset()
    reset_obs0 = reset_obs0.reshape(N, D).clone() if hasattr(reset_obs0, "clone") else np.array(reset_obs0).reshape(N, D)
    g = 0
    resets_at = {}
    for k in range(parts):   # (parts = 0: none)
        idx = np.sort(perm[k::parts])
        resets_at[g] = idx
        vec.reset(indices=idx)
        for _ in range(per):
            vec.step_device(pool_d[g % pool_size], want_obs=False)
            if anchor_every and g % anchor_every == anchor_every - 1:
                anchors_all[g] = rows_all()
            g += 1
    if g % 2:
        vec.step_device(pool_d[g % pool_size], want_obs=False)
        g += 1
    g0 = g
    # ---- the window: everything every env returns, kept on the device (65 536 envs x 300 steps x 60 floats = 4.7 GB, twice)
    W = int(window)
    obs_b, tobs_b = m.zeros((W, N, D)), m.zeros((W, N, D))
    rew_b, done_b, term_b = m.zeros((W, N)), m.zeros((W, N), "u8"), m.zeros((W, N), "u8")
    metr = {}
    for w in range(W):
        o, r, d = vec.step_device(pool_d[g % pool_size], want_obs=True)
        obs_b[w] = o.reshape(N, D)
        tobs_b[w] = vec._term_obs
        rew_b[w], done_b[w], term_b[w] = r, d, vec._term
        if anchor_every and g % anchor_every == anchor_every - 1:
            anchors_all[g] = rows_all()
        g += 1
        if metrics_every and (w % metrics_every == metrics_every - 1 or w == W - 1):
            mt = vec.metrics()
            metr[w] = mt.clone() if hasattr(mt, "clone") else np.array(mt)
    m.sync()
    done_h = parity._np(done_b).astype(bool)
    term_h = parity._np(term_b)
    # ---- which envs to check: chosen from what happened
    first_reset = np.zeros(N, dtype=np.int64)
    for gs, idx in resets_at.items():
        first_reset[idx] = gs
    limit_w = first_reset + steps_max - 1 - g0          # window step on which an env reset at first_reset runs out of time
    fail_ends = done_h & (term_h >= nat.TERM_VAR0)
    steps_ends = done_h & (term_h == nat.TERM_STEPS)
    ws = np.arange(W)[:, None]
    fail_on_limit = np.nonzero((fail_ends & (ws == limit_w[None, :])).any(axis=0))[0]
    sel_rng = np.random.default_rng(select_seed)
    chosen = []

    def add(cands, k):
        k = min(k, sample - len(chosen))
        if k <= 0:
            return
        cands = list(dict.fromkeys(int(c) for c in cands if int(c) not in set(chosen)))   # (no duplicates: an env checked twice would get its masked reset once)
        if len(cands) > k:
            cands = list(sel_rng.choice(cands, size=k, replace=False))
        chosen.extend(int(c) for c in cands)

    picked_first = 0
    if first_pick is not None:   # (a test's own class of lanes: first_pick(fail_ends [W, N], steps_ends [W, N], g0) -> env ids)
        add(first_pick(fail_ends, steps_ends, g0), sample // 4)
        picked_first = len(chosen)
    add(fail_on_limit, sample // 4)
    add(np.nonzero(fail_ends.any(axis=0))[0], sample // 4)
    add(np.nonzero(steps_ends.any(axis=0))[0], sample // 4)
    neigh = [e ^ 1 for e in chosen if (e ^ 1) < N] + [min(N - 1, (e & ~63) + int(sel_rng.integers(64))) for e in chosen]
    add(neigh, sample // 8)
    add(sel_rng.choice(N, size=min(N, 4 * sample), replace=False), sample - len(chosen))
    pos = np.array(sorted(chosen[:sample]))
    assert len(set(pos.tolist())) == len(pos)
    # ---- the oracles: the same life by global env id
    T = g
    acts = np.stack([pool_h[t % pool_size][pos] for t in range(T)])
    where = {int(p): j for j, p in enumerate(pos)}
    resets = {}
    for gs, idx in resets_at.items():
        loc = [where[int(e)] for e in idx if int(e) in where]
        if loc:
            resets[gs] = loc
    anchors = {}
    for gs, rows in anchors_all.items():
        w_ = parity._np(rows[:, pos] if not hasattr(rows, "cpu") else rows[:, __import__("torch").as_tensor(pos, device=rows.device)])
        w_ = w_.astype(np.float64).transpose(1, 0, 2).reshape(len(pos), 32)
        anchors[gs] = (w_[:, :18].copy(), w_[:, 18:26].copy(), w_[:, 26:32].copy())
    tr = op.run_traces(copy.deepcopy(cfg), [base + int(p) for p in pos], acts, seed, config_kw=ckw, sim_config_kw=skw,
                       resets=resets, anchors=anchors, workers=workers, keep_from=g0)
    # ---- the product's record of the same envs over the window
    rec = {"reset_obs": parity._np(reset_obs0[pos] if not hasattr(reset_obs0, "cpu") else reset_obs0[__import__("torch").as_tensor(pos, device=reset_obs0.device)]).astype(np.float64),
           "obs": np.stack([_take(obs_b, w, pos) for w in range(W)]).astype(np.float64),
           "reward": np.stack([_take(rew_b, w, pos) for w in range(W)]).astype(np.float64), "done": done_h[:, pos],
           "target": np.zeros((W, len(pos), 3)), "term": {}, "term_obs": {}, "metrics": {}, "masked_reset_obs": tr["masked_reset_obs"]}
    ends_per_env = done_h[:, pos].sum(axis=0)
    for w, j in zip(*np.nonzero(done_h[:, pos])):
        w, j = int(w), int(j)
        rec["term"][(w, j)] = nat.term_name(term_h[w, pos[j]])
        rec["term_obs"][(w, j)] = _take(tobs_b, w, [pos[j]])[0].astype(np.float64)
        # metrics: collected every `metrics_every` steps; the column of an env holds its LAST finished episode
        later = [x for x in sorted(metr) if x >= w]
        if later and not done_h[w + 1:later[0] + 1, pos[j]].any():
            col = parity._np(metr[later[0]])[:, pos[j]]
            rec["metrics"][(w, j)] = vec.metrics_dict(col)
    res = op.compare(rec, tr, rtol, atol, what=what, check_target=False)
    res.update({"sampled": len(pos), "ends_in_window": int(done_h.sum()), "failure_ends": int(fail_ends.sum()),
                "time_limit_ends": int(steps_ends.sum()), "failed_on_the_limit_step": int(len(fail_on_limit)),
                "failed_on_the_limit_step_checked": int(sum(1 for e in fail_on_limit if int(e) in where)),
                "sampled_ends": int(ends_per_env.sum()), "first_pick_checked": picked_first})
    return res
