"""Shared parity harness: drives a FixedWingVecEnv (real HIP library on the GPU box, or the host-emulation build in a
GPU-less container) next to N float64 oracle environments on identical seeds/inputs and compares everything the
reference's step()/reset() return."""
import copy
import math

import numpy as np

from oracle import physics as ph
from oracle.gym_restated import FixedWingOracle, PhiloxStream


def make_oracles(config, n, seed, config_kw=None, sim_config_kw=None, env_id_base=0):
    envs = []
    for i in range(n):
        o = FixedWingOracle(copy.deepcopy(config), config_kw=copy.deepcopy(config_kw),
                            sim_config_kw=copy.deepcopy(sim_config_kw))
        o.seed(seed)
        o.rng = PhiloxStream(seed, env_id_base + i)
        o.simulator.env_id = env_id_base + i
        envs.append(o)
    return envs


def _np(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
    except ImportError:
        pass
    return np.asarray(x)


class Mismatch(AssertionError):
    pass


def close(a, b, rtol, atol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        raise Mismatch("{}: shape {} vs {}".format(what, a.shape, b.shape))
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        raise Mismatch("{}: NaN pattern differs\n got {}\n want {}".format(what, a, b))
    err = np.abs(np.where(na, 0, a - b))
    tol = atol + rtol * np.abs(np.where(nb, 0, b))
    if np.any(err > tol):
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise Mismatch("{}: max violation at {}: got {!r} want {!r} (err {:.3e}, tol {:.3e})".format(
            what, i, a[i], b[i], err[i], tol[i]))
    return float(err.max()) if err.size else 0.0


def check_derived_views(vec, oracles, what, atol=2e-3):
    """The host views of the derived variables (field('roll') ..., written by store_sim with derived_views=True -- what PID
    / host controllers and the single-env facade read) against the oracle envs' current simulator state."""
    for name in ("roll", "pitch", "Va", "alpha"):
        got = _np(vec.field(name)).astype(np.float64)
        want = np.array([o.simulator.state[name].value for o in oracles], dtype=np.float64)
        d = got - want
        if name == "roll":
            d = (d + np.pi) % (2 * np.pi) - np.pi
        if np.abs(d).max() > atol:
            i = int(np.argmax(np.abs(d)))
            raise Mismatch("{}: derived view {} of env {}: got {!r} want {!r}".format(what, name, i, got[i], want[i]))


def run_gym_parity(vec, oracles, n_steps, action_fn, rtol=2e-3, atol=2e-3, check_metrics=True, reset_kw=None,
                   metric_rtol=None, log=None, check_views=False):
    """Free-running comparison over n_steps.  `action_fn(t) -> float32 [N,3]`.  Oracle envs are reset by hand when
    done (the VecEnv auto-resets).  Returns summary statistics."""
    N = vec.num_envs
    names = vec.target_names
    reset_kw = reset_kw or {}
    obs = _np(vec.reset(**({} if not reset_kw else reset_kw["vec"])))
    want = np.stack([o.reset(**({} if not reset_kw else reset_kw["oracle"][i])) for i, o in enumerate(oracles)])
    worst = {"obs": 0.0, "reward": 0.0, "target": 0.0}
    worst["obs"] = max(worst["obs"], close(obs.reshape(N, -1), want.reshape(N, -1), rtol, atol, "reset obs"))
    episodes = 0
    terms = {}
    for t in range(n_steps):
        a = np.asarray(action_fn(t), dtype=np.float32)
        obs, rew, done, infos = vec.step(a)
        obs, rew, done = _np(obs), _np(rew), _np(done).astype(bool)
        w_obs, w_rew, w_done, w_info = [], [], [], []
        for i, o in enumerate(oracles):
            ob, r, d, info = o.step(a[i].astype(np.float64))
            w_obs.append(ob), w_rew.append(r), w_done.append(d), w_info.append(info)
        w_done = np.array(w_done)
        if not np.array_equal(done, w_done):
            bad = np.nonzero(done != w_done)[0]
            raise Mismatch("step {}: done differs at envs {} (got {}, want {}; oracle info {})".format(
                t, bad[:8], done[bad[:8]], w_done[bad[:8]], [w_info[b].get("termination") for b in bad[:8]]))
        worst["reward"] = max(worst["reward"], close(rew, np.array(w_rew, dtype=np.float64), rtol, atol, "step {} reward".format(t)))
        tgt = np.array([[infos[i]["target"][n] for n in names] for i in range(N)]) if N <= 512 else None
        # info["target"] is the target at the END of the step (fixed_wing.py:435), i.e. for an env that ends its episode
        # the target BEFORE the VecEnv's auto-reset samples a new one
        w_tgt = np.array([[w_info[i]["target"][n] for n in names] for i in range(N)])
        for i in np.nonzero(done)[0]:
            info = infos[int(i)]
            episodes += 1
            terms[info["termination"]] = terms.get(info["termination"], 0) + 1
            if info["termination"] != w_info[i]["termination"]:
                raise Mismatch("step {} env {}: termination {} vs {}".format(t, i, info["termination"], w_info[i]["termination"]))
            if vec.auto_reset:
                close(info["terminal_observation"].reshape(-1), np.asarray(w_obs[i]).reshape(-1), rtol, atol,
                      "step {} env {} terminal obs".format(t, i))
            if check_metrics:
                for m in vec.cfg.get("metrics", []):
                    got, exp = info[m["name"]], w_info[i][m["name"]]
                    if set(got.keys()) != set(exp.keys()):
                        raise Mismatch("metric {} keys {} vs {}".format(m["name"], got.keys(), exp.keys()))
                    for k in got:
                        mr = metric_rtol if metric_rtol is not None else max(rtol, 5e-3)
                        if m["name"] in ("rise_time", "settling_time"):
                            # integer step indices: a threshold crossing may move by a step under fp32 rounding
                            g, e_ = float(got[k]), float(exp[k])
                            if (math.isnan(g) != math.isnan(e_)) or (not math.isnan(g) and abs(g - e_) > 1.0):
                                raise Mismatch("step {} env {} metric {}[{}]: {} vs {}".format(t, i, m["name"], k, g, e_))
                        else:
                            close(float(got[k]), float(exp[k]), mr, max(atol, 1e-3), "step {} env {} metric {}[{}]".format(t, i, m["name"], k))
            if vec.auto_reset:
                w_obs[i] = oracles[i].reset()
        if tgt is not None:
            worst["target"] = max(worst["target"], close(tgt, w_tgt, rtol, atol, "step {} target".format(t)))
        worst["obs"] = max(worst["obs"], close(obs.reshape(N, -1), np.stack(w_obs).reshape(N, -1), rtol, atol, "step {} obs".format(t)))
        if check_views and (done.any() or t % 16 == 0):   # (after an auto-reset: the NEW episode's initial state)
            check_derived_views(vec, oracles, "step {}".format(t), max(atol, 2e-3))
        if log is not None and t % 20 == 0:
            log("step {} worst {}".format(t, worst))
    worst["episodes"] = episodes
    worst["terminations"] = terms
    return worst


# ----------------------------------------------------------------------------------------------------------------------
# single-step physics parity on arbitrary batch sizes (the 1e-5 bar on the 13 rigid-body states)
# ----------------------------------------------------------------------------------------------------------------------
STATE_SCALE = np.array([1, 1, 1, 1, 1, 1, 1, 100, 100, 100, 20, 5, 5, 0.5, 0.5, 1, 3.5, 3.5])


def oracle_spec_from_env_config(ec):
    """SimSpec of the oracle for the same simulator configuration as a product EnvConfig (constraints after the gym
    config's overrides)."""
    sim_cfg = copy.deepcopy(ec.sim_cfg)
    sim_cfg["turbulence"] = ec.turbulence
    sim_cfg["turbulence_intensity"] = ec.turbulence_intensity
    sim_cfg["turbulence_output"] = ec.turbulence_output
    spec = ph.SimSpec(sim_cfg, ec.params)
    for name, var in ec.state.items():
        i = ph.VAR_ID[name]
        for prop, arr in (("constraint_min", spec.con_min), ("constraint_max", spec.con_max),
                          ("value_min", spec.val_min), ("value_max", spec.val_max)):
            v = getattr(var, prop)
            arr[i] = np.nan if v is None else v
    return spec


def physics_state(vec):
    """(y[N,18], wind[N,3], dryden[N,8]) float64 copies of the device state."""
    W = words(vec)
    s0, c0 = vec.layout.sim, vec.layout.cold
    y = W[s0:s0 + 18].T.astype(np.float64)
    wind = W[c0:c0 + 3].T.astype(np.float64)
    dry = W[s0 + 18:s0 + 26].T.astype(np.float64) if vec.env_config.turbulence else np.zeros((y.shape[0], 8))
    return y, wind, dry


def device_gust(vec, spec, dry):
    """The gust sample [N,6] the NEXT device step will use: kept in the simulator rows (increment turbulence) or C x."""
    if not vec.env_config.turbulence:
        return np.zeros((dry.shape[0], 6))
    if vec.env_config.turbulence_output == "increment":
        W = words(vec)
        s0 = vec.layout.sim
        return W[s0 + 26:s0 + 32].T.astype(np.float64)
    return ph.dryden_output(spec, dry)


def words(vec):
    """Host copy of the arena as [word][env] (the device layout is 16-byte groups [word >> 2][env][word & 3])."""
    S = _np(vec.state)
    return np.ascontiguousarray(S.transpose(0, 2, 1).reshape(S.shape[0] * 4, S.shape[1]))


def scaled_actions(vec, raw):
    """What the device feeds the simulator for raw actions (fixed_wing.py:349-354,439-459), float64."""
    ec = vec.env_config
    raw = np.asarray(raw, dtype=np.float64)
    if not ec.scale_actions:
        return raw
    lo, hi = ec.cfg["action"]["scale_low"], ec.cfg["action"]["scale_high"]
    return (ec.action_scale_to_high - ec.action_scale_to_low) * (np.clip(raw, lo, hi) - lo) / (hi - lo) + ec.action_scale_to_low


# ---- simulator["model"]: per-env force / moment constants in the arena (section L.aero, order of FWG_AERO_LIST in
# csrc/fwgym_dev.h) against the constants derived from an oracle env's sampled parameter table
AERO_NAMES = ("half_rho_S mg inv_mass inv_Jy G1 G2 G3 G4 G5 G6 G7 G8 M Ma0 CL0 CLa cLq CLde CDp kInd CDb1 CDb2 cDq CDde "
              "Cm0 Cma cmq Cmde Cmfp chord span CY0 CYb cYp cYr CYda Cl0 Clb clp clr Clda Cn0 Cnb cnp cnr Cnda kprop kmotor ktp").split()


def aero_from_params(P, rho, g, inertia=None):
    # inertia: the parameter FILE's (Jx, Jy, Jz, Jxz) -- PyFly's inertia matrix and gammas are built once (oracle/physics.py SimSpec)
    Jx, Jy, Jz, Jxz = inertia if inertia is not None else (P["Jx"], P["Jy"], P["Jz"], P["Jxz"])
    G = Jx * Jz - Jxz * Jxz
    b, c = P["b"], P["c"]
    v = {"half_rho_S": 0.5 * rho * P["S_wing"], "mg": P["mass"] * g, "inv_mass": 1 / P["mass"], "inv_Jy": 1 / Jy,
         "G1": Jxz * (Jx - Jy + Jz) / G, "G2": (Jz * (Jz - Jy) + Jxz * Jxz) / G, "G3": Jz / G, "G4": Jxz / G,
         "G5": (Jz - Jx) / Jy, "G6": Jxz / Jy, "G7": ((Jx - Jy) * Jx + Jxz * Jxz) / G, "G8": Jx / G,
         "M": P["M"], "Ma0": P["M"] * P["a_0"], "CL0": P["C_L_0"], "CLa": P["C_L_alpha"], "cLq": P["C_L_q"] * c,
         "CLde": P["C_L_delta_e"], "CDp": P["C_D_p"], "kInd": 1 / (np.pi * P["e"] * P["ar"]), "CDb1": P["C_D_beta1"],
         "CDb2": P["C_D_beta2"], "cDq": P["C_D_q"] * c, "CDde": P["C_D_delta_e"], "Cm0": P["C_m_0"], "Cma": P["C_m_alpha"],
         "cmq": P["C_m_q"] * b, "Cmde": P["C_m_delta_e"], "Cmfp": P["C_m_fp"], "chord": c, "span": b, "CY0": P["C_Y_0"],
         "CYb": P["C_Y_beta"], "cYp": P["C_Y_p"] * b, "cYr": P["C_Y_r"] * b, "CYda": P["C_Y_delta_a"], "Cl0": P["C_l_0"],
         "Clb": P["C_l_beta"], "clp": P["C_l_p"] * b, "clr": P["C_l_r"] * b, "Clda": P["C_l_delta_a"], "Cn0": P["C_n_0"],
         "Cnb": P["C_n_beta"], "cnp": P["C_n_p"] * b, "cnr": P["C_n_r"] * b, "Cnda": P["C_n_delta_a"],
         "kprop": 0.5 * rho * P["S_prop"] * P["C_prop"], "kmotor": P["k_motor"], "ktp": P["k_T_P"] * P["k_Omega"] ** 2}
    return np.array([v[n] for n in AERO_NAMES], dtype=np.float64)


def device_aero(vec):
    """[N, 49] host copy of the envs' current force / moment constants."""
    L = vec.layout
    return np.stack([_np(vec.word(L.aero + i)) for i in range(len(AERO_NAMES))], axis=1).astype(np.float64)


def check_parameter_api(vec, oracles, when):
    """get_simulator_parameters (fixed_wing.py:872-888): the device keeps the sampled values in float32, so the raw values
    agree to 1e-6 relative and the normalised ones to that resolution divided by the spread."""
    raw_d, raw_o = vec.get_simulator_parameters(False), np.array([o.get_simulator_parameters(False) for o in oracles])
    close(raw_d, raw_o, 1e-6, 1e-12, "get_simulator_parameters(False) " + when)
    nrm_d, nrm_o = vec.get_simulator_parameters(True), np.array([o.get_simulator_parameters(True) for o in oracles])
    assert nrm_d.shape == nrm_o.shape, (nrm_d.shape, nrm_o.shape)
    model = vec.cfg["simulator"]["model"]
    j = 0
    for pa in model["parameters"]:
        orig = float(vec.env_config.params[pa["name"]])
        var = pa.get("var", model["var"])
        if model.get("var_type", "relative") == "relative":
            if orig == 0:
                continue
            var = var * orig
        close(nrm_d[:, j], nrm_o[:, j], 0.0, 2e-6 * max(abs(orig), 1e-30) / abs(var) + 1e-9, "normalised {} {}".format(pa["name"], when))
        j += 1
    assert j == nrm_d.shape[1]


def check_model_randomisation(vec, oracles, steps, action_fn):
    """Resets, then steps through at least one auto-reset: after every (re)start the constants in the arena equal those of
    the oracle env's sampled table (1e-5: float32 sampling and derivation vs float64), and they change between episodes."""
    vec.reset()
    for o in oracles:
        o.reset()
    rho, g = oracles[0].simulator.rho, oracles[0].simulator.g
    first = device_aero(vec)
    want = np.stack([aero_from_params(o.simulator.params, rho, g, o.simulator._file_inertia) for o in oracles])
    close(first, want, 1e-5, 1e-7, "per-env constants after reset")
    check_parameter_api(vec, oracles, "after reset")
    assert np.abs(first[0] - first[1]).max() > 0, "two envs drew the same aircraft"
    changed = 0
    for t in range(steps):
        a = action_fn(t)
        out = vec.step(a)
        done = _np(out[2]).astype(bool)
        for i, o in enumerate(oracles):
            _, _, d, _ = o.step(np.asarray(a[i], dtype=np.float64))
            assert bool(d) == bool(done[i]), (t, i)
            if d:
                o.reset()
        if done.any():
            now = device_aero(vec)
            want = np.stack([aero_from_params(o.simulator.params, rho, g, o.simulator._file_inertia) for o in oracles])
            close(now, want, 1e-5, 1e-7, "per-env constants after the auto-reset at step {}".format(t))
            check_parameter_api(vec, oracles, "after the auto-reset at step {}".format(t))
            changed += int((np.abs(now - first).max(axis=1) > 0)[done].sum())
            first = now
    assert changed > 0
    return changed
