"""Scenario-replay evaluation on the vectorised env -- the protocol of the reference's
examples/evaluate_controller.py:44-169 (evaluate_model_on_set), with every scenario of the test set running in its own
env slot at the same time instead of being queued onto a handful of sub-process envs.

A scenario is {"state": {21 floats}, "target": {roll, pitch, Va}} (the record format of get_initial_state,
fixed_wing.py:848-862).  Overrides applied exactly as the reference does (:68-78): steps_max 1500, on_success "done",
streak 100 @ fraction 1, bounds 5deg/5deg/2 m/s, PID => scale_space False, turbulence via sim_config_kw.
Returns the reference's result layout: {metric: {state: [per scenario]}, "rewards": [per scenario [per step]]}, all in
SCENARIO order (the reference stores the metric lists in completion order, evaluate_controller.py:127-137)."""
import copy

import numpy as np

from .pid import BatchedPID
from .vec_env import FixedWingVecEnv

METRICS = ("success", "control_variation", "rise_time", "overshoot", "settling_time")


def evaluation_overrides(use_pid, config_kw=None):
    kw = {} if config_kw is None else copy.deepcopy(config_kw)
    kw.update({"steps_max": 1500, "target": {"on_success": "done", "success_streak_fraction": 1,
                                             "success_streak_req": 100,
                                             "states": {0: {"bound": 5}, 1: {"bound": 5}, 2: {"bound": 2}}}})
    if use_pid:
        kw["action"] = {"scale_space": False}
    return kw


def evaluate_on_set(scenarios, config_path=None, policy=None, config_kw=None, turbulence_intensity="none", device=0,
                    seed=0, metrics=METRICS, pid_gains=None, first_step_policy=None, **vec_kw):
    """policy: None => the PID baseline; otherwise a callable obs[N, ...] (device tensor) -> actions[N, 3].

    first_step_policy: the callable that produces the FIRST action of every episode (default: `policy`).  The reference's
    evaluation wraps the envs in VecNormalize but takes the observation of a scenario's reset straight from
    env_method("reset", ...) (evaluate_controller.py:118), which bypasses the wrapper: the model's first action of every
    episode is computed from the UN-normalised observation, all later ones from normalised ones (:139, :153).  The published
    RL results carry that step (second-step reward error against eval_res_RL_MLP_none.npy 0.023 -> 0.0006 with it, control
    variation 0.29 -> 0.36 against the published 0.41); pass the policy WITHOUT its observation normalisation here to fly the
    reference's protocol to the letter."""
    import torch
    use_pid = policy is None
    n = len(scenarios)
    kw = evaluation_overrides(use_pid, config_kw)
    sim_kw = {"turbulence": turbulence_intensity != "none", "turbulence_intensity": turbulence_intensity}
    vec = FixedWingVecEnv(config_path, num_envs=n, device=device, config_kw=kw, sim_config_kw=sim_kw, auto_reset=False,
                          seed=seed, **vec_kw)
    names = vec.target_names
    states = {k: np.array([s["state"][k] for s in scenarios], dtype=np.float32) for k in scenarios[0]["state"]}
    targets = {k: np.array([s["target"][k] for s in scenarios], dtype=np.float32) for k in names}
    obs = vec.reset(states=states, targets=targets)
    as_t = (lambda x: x) if isinstance(obs, torch.Tensor) else (lambda x: torch.as_tensor(np.asarray(x)))
    obs = as_t(obs)
    dev = obs.device
    if use_pid:
        obs_names = [v["name"] for v in vec.cfg["observation"]["states"]]
        try:
            i_phi, i_theta, i_va = obs_names.index("roll"), obs_names.index("pitch"), obs_names.index("Va")
            i_om = [obs_names.index("omega_p"), obs_names.index("omega_q"), obs_names.index("omega_r")]
        except ValueError:
            raise ValueError("When using PID roll, pitch, Va, omega_p, omega_q, omega_r must be part of the "
                             "observation vector.")
        pid = BatchedPID(n, dt=vec.dt, device=dev)
        for k, v in (pid_gains or {}).items():
            setattr(pid, k, v)
        pid.set_reference(*(torch.as_tensor(targets[k], device=dev) for k in names))
    active = torch.ones(n, dtype=torch.bool, device=dev)
    rewards = [[] for _ in range(n)]
    res = {m: {} for m in metrics}
    finished = {}
    for t_step in range(vec.cfg["steps_max"] + 1):
        if use_pid:
            row = obs.reshape(n, -1)
            act = pid.get_action(row[:, i_phi], row[:, i_theta], row[:, i_va], row[:, i_om])
        else:
            act = as_t((first_step_policy if (t_step == 0 and first_step_policy is not None) else policy)(obs)).to(dev)
        act = torch.where(active[:, None], act.float(), torch.zeros_like(act, dtype=torch.float32))
        obs, rew, done, infos = vec.step(act if isinstance(vec._obs, torch.Tensor) else act.cpu().numpy())
        obs, rew, done = as_t(obs), as_t(rew), as_t(done).bool()
        if use_pid:  # the reference refreshes the PID reference from info["target"] (evaluate_controller.py:146-149)
            tg = as_t(vec._target)
            pid.set_reference(tg[:, names.index("roll")], tg[:, names.index("pitch")], tg[:, names.index("Va")])
        r_host, a_host = rew.cpu().numpy(), active.cpu().numpy()
        for i in np.nonzero(a_host)[0]:
            rewards[i].append(float(r_host[i]))
        newly = (done.to(dev) & active).cpu().numpy()
        for i in np.nonzero(newly)[0]:
            finished[int(i)] = dict(infos[int(i)])
        active = active & ~done.to(dev)
        if not bool(active.any()):
            break
    for m in metrics:
        for i in range(n):
            val = finished[i][m] if i in finished else {}
            for state, v in val.items():
                res[m].setdefault(state, [None] * n)[i] = v
    res["rewards"] = rewards
    res["termination"] = [finished[i].get("termination") if i in finished else None for i in range(n)]
    vec.close()
    return res


def summarize(res, dt=0.01):
    """The numbers of the reference's results table (examples/README.md:33-47; print_results,
    evaluate_controller.py:32-41): success rates in %, times in seconds, overshoot in %, metrics other than success
    averaged over the successful episodes."""
    ok = np.array([bool(v) for v in res["success"]["all"]])
    out = {"success_%": {k: 100.0 * np.mean([bool(x) for x in v]) for k, v in res["success"].items()}}
    for m, scale in (("rise_time", dt), ("settling_time", dt), ("overshoot", 100.0), ("control_variation", 1.0)):
        if m in res:
            out[m] = {k: float(np.nanmean([x * scale if (o and x is not None) else np.nan
                                           for x, o in zip(v, ok)])) for k, v in res[m].items()}
    return out
