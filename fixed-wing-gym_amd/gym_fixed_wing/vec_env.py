"""FixedWingVecEnv -- N fixed-wing environments stepped by ONE fused HIP kernel launch on one MI355X.

This is the batched replacement for what the reference runs as `SubprocVecEnv([make_env(...) for i in range(n)])`
(examples/train_rl_controller.py:223, examples/evaluate_controller.py:80): every env is still the reference's
FixedWingAircraft (same config schema, same step/reset semantics), but all of them live in one SoA state arena in HBM
and one `fwg_step` call advances them all.  The class honours the stable-baselines VecEnv contract the example scripts
rely on (reset/step/step_async/step_wait/seed/get_attr/set_attr/env_method/close, auto-reset with
info["terminal_observation"], per-episode metric entries in info at done).
"""
import ctypes
import math
import os
import warnings

import numpy as np

from . import _native as nat
from .config import EnvConfig
from .spaces import Box


class _TorchBackend(object):
    """Device memory and streams come from PyTorch-ROCm (plumbing only)."""

    def __init__(self, device):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise nat.NativeError("no ROCm device visible to PyTorch: FixedWingVecEnv needs an MI355X (no CPU fallback)")
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self.index)   # explicit index: torch.device("cuda") != torch.device("cuda", 0)
        self._dt = {"f32": torch.float32, "u8": torch.uint8, "i32": torch.int32}

    def zeros(self, shape, kind="f32"):
        return self.torch.zeros(shape, dtype=self._dt[kind], device=self.device)

    def full(self, shape, value, kind="f32"):
        return self.torch.full(shape, value, dtype=self._dt[kind], device=self.device)

    def ptr(self, t):
        return ctypes.c_void_p(t.data_ptr())

    def stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def sync(self):
        self.torch.cuda.current_stream(self.device).synchronize()

    def to_host(self, t):
        return t.detach().cpu().numpy()

    def from_host(self, a, kind="f32"):
        return self.torch.as_tensor(np.ascontiguousarray(a), dtype=self._dt[kind]).to(self.device)

    def as_device(self, x, kind="f32"):
        if isinstance(x, self.torch.Tensor):
            if x.device != self.device or x.dtype != self._dt[kind] or not x.is_contiguous():
                x = x.to(device=self.device, dtype=self._dt[kind]).contiguous()
            return x
        return self.from_host(np.asarray(x, dtype=np.float32), kind)

    def view_i32(self, t):
        return t.view(self.torch.int32)


class CaptureToken(int):
    """What capture_begin returns: the step parity of the capture (its int value, fwg_replay_check's argument) + the global step
    it started at and whether zero-copy observation windows were handed out while it ran (replay_check's phase rule)."""

    def __new__(cls, parity, gstep):
        obj = int.__new__(cls, parity)
        obj.gstep, obj.uses_views = int(gstep), False
        return obj


class LazyInfos(object):
    """The `infos` list of VecEnv.step_wait: dicts are materialised on first access (65 536 Python dicts per step would
    cost more than the simulation).  infos[i] has "target" always (fixed_wing.py:435) and, for finished episodes,
    "termination", the configured metrics (fixed_wing.py:419-421) and "terminal_observation"."""

    def __init__(self, env, done, term, target, metrics, term_obs):
        self._env, self._done, self._term, self._target, self._metrics, self._term_obs = env, done, term, target, metrics, term_obs
        self._cache = {}

    def __len__(self):
        return self._env.num_envs

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if i not in self._cache:
            self._cache[i] = self._env._build_info(i, self._done, self._term, self._target, self._metrics, self._term_obs)
        return self._cache[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class FixedWingVecEnv(object):
    def __init__(self, config_path=None, num_envs=1, device=0, sim_config_path=None, sim_parameter_path=None,
                 config_kw=None, sim_config_kw=None, auto_reset=True, as_numpy=False, env_id_base=0, seed=0,
                 derived_views=True, specialize=None, obs_log_rows=None, obs_layout=None, _backend=None, _lib_path=None):
        self.env_config = EnvConfig(config_path, sim_config_path, sim_parameter_path, config_kw, sim_config_kw)
        self.cfg = self.env_config.cfg
        self.num_envs = int(num_envs)
        self.env_id_base = int(env_id_base)   # global id of env 0 (RNG streams are keyed by global ids)
        self.as_numpy = as_numpy
        self.auto_reset = auto_reset
        # derived_views=False drops the per-step write of roll/pitch/yaw/Va/alpha/beta into the arena (32 B/env-step);
        # they are only needed by host-side views (field("roll"), the single-env class), never by step() itself
        self.derived_views = bool(derived_views)
        self.training = True
        self._lib = nat.load_library(_lib_path)
        self._mem = _backend if _backend is not None else _TorchBackend(device)
        ec = self.env_config
        self.observation_space = Box(low=ec.obs_low, high=ec.obs_high, dtype=np.float32)
        self.action_space = Box(low=ec.action_space_low, high=ec.action_space_high, dtype=np.float32)
        self.obs_shape = tuple(ec.obs_shape)
        self.obs_dim = int(np.prod(self.obs_shape))
        self.target_names = list(ec.target_names)
        self.dt = ec.dt

        # obs_log_rows = L > 0 (lagged observations, no observation noise): the observation history is kept once, as a
        # row log, and the observation handed out is a zero-copy strided view of it (include/fwgym.h "Row-log
        # observations"): same values, 432 B/env-step less traffic at C3.  .contiguous() gives the dense batch.
        # None (default) = row log wherever it applies (lagged rows, no observation noise, records of whole 16-byte
        # groups), dense batch otherwise; 0 = always the dense batch.
        # obs_layout = "dense" / "row_log" / "auto" spells the same choice by name.  Take "dense" for a consumer that reads the
        # whole batch on every step of a REPLAYED GRAPH (a torch policy): the zero-copy window of the row log is a host-side
        # view that goes stale under replay, so graph mode hands out a gathered copy instead (fwg_obs_gather after every step:
        # 21 us per C3 step against 14 us with the dense batch, MI355X, 65 536 envs, steady state).  The HIP rollout head
        # (DeviceActor / FusedRollout) reads the log in place and wants the default.
        if obs_layout is not None:
            if obs_layout not in ("auto", "dense", "row_log"):
                raise ValueError("obs_layout must be 'auto', 'dense' or 'row_log', not {!r}".format(obs_layout))
            if obs_log_rows is not None:
                raise ValueError("give obs_layout or obs_log_rows, not both")
            if obs_layout == "dense":
                obs_log_rows = 0
            elif obs_layout == "row_log":
                if not self._row_log_applies():
                    raise ValueError("obs_layout='row_log' needs lagged observation rows, no observation noise and records "
                                     "of whole 16-byte groups")
                from . import presets as _presets
                obs_log_rows = _presets.OBS_LOG_ROWS
        if obs_log_rows is None:
            from . import presets as _presets
            obs_log_rows = _presets.OBS_LOG_ROWS if self._row_log_applies() else 0
        self.obs_log_rows = int(obs_log_rows)
        self._c = ec.compile(auto_reset=auto_reset, store_derived=self.derived_views, obs_log_rows=self.obs_log_rows)
        # Three tiers (include/fwgym.h fwg_spec_index): a configuration that IS one of the build-time presets runs its frozen
        # kernel; one that differs from a preset in VALUES only (scalings, normalisation, constraints, aircraft constants, noise,
        # the time limit ...) runs that preset's SHAPE instance, ~1.1x the frozen kernel's step time, nothing to compile; anything
        # else -- another observation layout, another set of reward factors -- runs the GENERIC kernel, which interprets the
        # configuration (scalar loads, LDS tables, ~4 KB of scratch per lane: ~35x per step at 65 536 envs) unless a specialised
        # copy of the library is compiled for it (jit.py: hipcc, one to two minutes once per distinct configuration, cached by
        # content hash).  Default: compile whenever hipcc is on the machine and only the generic kernel is left; specialize=True /
        # FWGYM_JIT=1 compile for a shape-instance configuration too (the last 10 %); specialize=False / FWGYM_JIT=0 never
        # compile.  The generic kernel is announced either way: nobody should find out from a profile.
        # (Batches below 1 024 envs -- the single-env class, unit tests -- are bound by launch and host overhead whatever the
        # kernel: they keep the generic kernel silently unless asked.)
        big = self.num_envs >= 1024
        insist = specialize is True
        if specialize is None:
            env_jit = os.environ.get("FWGYM_JIT")
            if env_jit is not None:
                specialize = insist = env_jit == "1"
            else:
                from . import jit as _jit
                specialize = big and _jit.hipcc_path() is not None
        if _lib_path is None and (specialize or big) and not self._preset_matches(shape_ok=not insist):
            path = None
            if specialize:
                from . import jit
                if not jit.is_cached(ec, auto_reset=auto_reset, store_derived=self.derived_views, base_lib=self._lib,
                                     obs_log_rows=self.obs_log_rows):
                    warnings.warn("FixedWingVecEnv: this configuration is not one of the build-time presets; compiling a "
                                  "specialised kernel for it with hipcc (one to two minutes, cached afterwards; "
                                  "specialize=False or FWGYM_JIT=0 run the ~35x slower generic kernel instead)", RuntimeWarning, stacklevel=2)
                path = jit.specialised_library(ec, auto_reset=auto_reset, store_derived=self.derived_views, base_lib=self._lib,
                                               obs_log_rows=self.obs_log_rows)
                if path is not None:
                    self._lib = nat.load_library(path)
            if path is None:
                warnings.warn("FixedWingVecEnv: this configuration is not one of the build-time presets and no specialised "
                              "kernel is available ({}): it runs the GENERIC kernel, about 35x slower per step than a specialised "
                              "one at 65 536 envs".format("specialisation switched off" if not specialize else "hipcc missing or the compile failed"),
                              RuntimeWarning, stacklevel=2)
        self.layout = nat.Layout()
        nat.check(self._lib, self._lib.fwg_get_layout(ctypes.byref(self._c), ctypes.byref(self.layout)))
        N, m = self.num_envs, self._mem
        # state arena: 16-byte groups [rows/4][env][4] (word w of env e = state[w >> 2, e, w & 3]); zero = "never reset"
        self.state = m.zeros((self.layout.rows // 4, N, 4))
        self._obs = m.zeros((N, self.obs_dim))
        self._obs_dense = self._obs        # dense [N][obs_dim] buffer (row-log mode: target of fwg_obs_gather)
        self._graph_mode = False
        self._obs_buf = self._obs          # what fwg_reset / fwg_step write: the dense batch, or the row log
        if self.obs_log_rows:
            n_log = int(self._lib.fwg_obs_log_floats(ctypes.byref(self._c), N))
            self._obs_buf = m.zeros((n_log // (N * self._c.n_obs), N, self._c.n_obs))   # [obs_step * L][N][n_obs]
        self._rew = m.zeros((N,))
        self._done = m.zeros((N,), "u8")
        self._term = m.zeros((N,), "u8")
        self._term_obs = m.zeros((N, self.obs_dim))
        self._metrics = m.full((nat.N_METRICS, N), math.nan)
        self._target = m.zeros((N, len(self.target_names)))
        self._handle = ctypes.c_void_p()
        nat.check(self._lib, self._lib.fwg_create(ctypes.byref(self._c), N, getattr(m, "index", 0), m.ptr(self.state),
                                                  int(env_id_base), ctypes.byref(self._handle)))
        self.check_actions = False
        self._pending = None
        self.seed(seed)

    def _row_log_applies(self):
        ob = self.cfg["observation"]
        noise = ob.get("noise", None)
        noisy = noise is not None and (noise.get("var", 0) != 0 or noise.get("mean", 0) != 0)
        return int(ob.get("length", 1)) > 1 and not noisy and len(ob["states"]) % 4 == 0

    # ------------------------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.fwg_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seed(self, seed=None):
        """FixedWingAircraft.seed (fixed_wing.py:214-222): env i of a VecEnv gets stream (seed, env_id_base + i)."""
        seed = 0 if seed is None else int(seed)
        self._seed = seed
        nat.check(self._lib, self._lib.fwg_seed(self._handle, ctypes.c_uint64(seed & (2 ** 64 - 1))))
        return [seed + i for i in range(min(self.num_envs, 1))] if self.num_envs == 1 else [seed]

    def set_curriculum_level(self, level):
        """fixed_wing.py:224-285 for every env (the reference broadcasts it with env_method,
        examples/train_rl_controller.py:84)."""
        self.env_config.set_curriculum_level(level)
        self._upload()

    def _preset_matches(self, shape_ok=True):
        """Does this library hold a kernel instance for the configuration: a frozen one, or (shape_ok) a shape instance?"""
        import ctypes
        inst = int(self._lib.fwg_config_instance(ctypes.byref(self._c)))
        if inst < 0:
            nat.check(self._lib, inst)
        return inst < (nat.INSTANCE_SHAPE if not shape_ok else nat.INSTANCE_GENERIC)

    def set_graph_mode(self, enable=True, obs="gather"):
        """Keeps the global step counter on the device so that a captured sequence of step launches (hipGraph /
        torch.cuda.CUDAGraph) can be replayed; see fwg_set_graph_mode in include/fwgym.h for the rules.

        Row-log envs, `obs`: what step()/step_device() hand out while the mode is on.
        "gather" (default): a dense copy gathered on the device after every step (fwg_obs_gather reads the position on the
        device: right under replay for ANY chunk length; one more launch and ~700 B/env-step more traffic per step).
        "view" (opt-in): the zero-copy window of the log, as without graphs.  The window's position is a function of the global
        step with period `obs_window_period` (obs_step x log depth: 64 steps for the default log of a 5 x 12 observation at
        step 2), so the views handed out during a CAPTURE stay right under replay exactly when every replay starts a whole
        number of periods after the capture did -- capture chunks whose length is a multiple of the period (the default log
        depth is chosen so that 64, 128, 256 are): capture_begin(n_steps) refuses other lengths, replay_check(token) other
        phases.  The HIP rollout head reads the log in place (fwg_actor_set_obs_log) and needs neither."""
        if obs not in ("view", "gather"):
            raise ValueError("obs must be 'view' or 'gather'")
        nat.check(self._lib, self._lib.fwg_set_graph_mode(self._handle, int(bool(enable)), self._mem.stream()))
        self._graph_mode = bool(enable)
        self._graph_obs = obs
        self._capturing, self._cap_token = False, None
        self._refresh_obs_view()

    @property
    def obs_window_period(self):
        """Row-log envs: the number of steps after which the observation window returns to the same planes of the log."""
        if not self.obs_log_rows:
            return 0
        return int(self._c.obs_step) * (self.obs_log_rows - (int(self._c.obs_length) - 1))

    def capture_begin(self, n_steps=None):
        """Brackets the step calls issued under stream capture.  Returns the capture's TOKEN -- an int, the step parity the graph
        may be replayed at, that also carries the step the capture started at and whether zero-copy observation windows were
        handed out during it: keep it with the graph and pass it to replay_check (one token per graph; several graphs of one env
        may be alive).  `n_steps` (the number of steps about to be captured), when given in "view" mode, is checked against the
        window period here instead of at the second replay."""
        if n_steps is not None and self.obs_log_rows and getattr(self, "_graph_obs", "gather") == "view" and \
                int(n_steps) % self.obs_window_period:
            raise ValueError("set_graph_mode(obs='view'): a captured chunk must be a multiple of the observation window period "
                             "({} steps), not {} steps -- or use obs='gather' / obs_layout='dense'".format(self.obs_window_period, n_steps))
        nat.check(self._lib, self._lib.fwg_capture_begin(self._handle))
        self._capturing = True
        self._cap_token = CaptureToken(int(self._lib.fwg_capture_parity(self._handle)), self.global_step)
        return self._cap_token

    def capture_end(self):
        nat.check(self._lib, self._lib.fwg_capture_end(self._handle))
        self._capturing = False
        return self._cap_token

    @property
    def global_step(self):
        return int(self._lib.fwg_global_step(self._handle))

    def replay_check(self, token):
        """Raises unless the graph whose capture returned `token` (capture_begin) may be replayed now."""
        nat.check(self._lib, self._lib.fwg_replay_check(self._handle, int(token)))
        if self.obs_log_rows and getattr(token, "uses_views", False):
            off = (self.global_step - token.gstep) % self.obs_window_period
            if off != 0:
                raise RuntimeError("the captured steps handed out zero-copy observation windows of the row log: a replay must start a "
                                   "whole number of window periods ({} steps) after the capture did, this one starts {} steps past one.  "
                                   "Capture chunks that are a multiple of the period, or use set_graph_mode(True, obs='gather') / "
                                   "obs_layout='dense'".format(self.obs_window_period, off))

    def note_replayed_steps(self, n_steps):
        nat.check(self._lib, self._lib.fwg_note_replayed_steps(self._handle, int(n_steps)))
        # row-log mode: the window moved with the replayed steps (view mode: recomputed from the host's count of them)
        self._refresh_obs_view(want_obs=getattr(self, "_graph_obs", "gather") == "view")

    @property
    def spec_index(self):
        """Index of the build-time specialised kernel this env runs, -1 = generic kernel."""
        return int(self._lib.fwg_spec_index(self._handle))

    def _upload(self):
        self._c = self.env_config.compile(auto_reset=self.auto_reset, store_derived=self.derived_views,
                                          obs_log_rows=self.obs_log_rows)
        nat.check(self._lib, self._lib.fwg_update_config(self._handle, ctypes.byref(self._c)))

    def set_simulator_attr(self, key, value):
        """setattr(simulator, key, val) (fixed_wing.py:570), e.g. turbulence_intensity."""
        if key == "turbulence_intensity":
            self.env_config.turbulence_intensity = value
        elif key == "turbulence":
            if bool(value) != self.env_config.turbulence:
                raise ValueError("turbulence on/off changes the state layout: construct the env with sim_config_kw")
        else:
            raise NotImplementedError("simulator attribute {}".format(key))
        self._upload()

    # ------------------------------------------------------------------------------------------------------------------
    def _out(self, t, shape=None):
        if shape is not None:
            t = t.reshape(shape)
        return self._mem.to_host(t) if self.as_numpy else t

    def reset(self, indices=None, states=None, targets=None):
        """Resets all envs (or those in `indices`).  `states`: dict var name -> value(s) as in the reference's
        reset(state=...) (fixed_wing.py:287,308); `targets`: dict target name -> value(s) (fixed_wing.py:311-315).
        Returns the observations of ALL envs (rows of untouched envs are their latest observations)."""
        N, m = self.num_envs, self._mem
        mask_t = None
        idx = None
        if indices is not None:
            idx = np.atleast_1d(np.asarray(indices, dtype=np.int64))
            mask = np.zeros(N, dtype=np.uint8)
            mask[idx] = 1
            mask_t = m.from_host(mask, "u8")
        n_sel = N if idx is None else len(idx)

        def expand(rows, names, values):
            arr = np.full((rows, N), np.nan, dtype=np.float32)
            for name, val in values.items():
                if val is None:
                    continue
                r = names.index(name)
                val = np.asarray(val, dtype=np.float32)
                if idx is None:
                    arr[r, :] = val
                else:
                    arr[r, idx] = val
            return m.from_host(arr)

        st = expand(nat.N_RESET_VARS, nat.VARS[:nat.N_RESET_VARS], states) if states else None
        tg = expand(len(self.target_names), self.target_names, targets) if targets else None
        null = ctypes.c_void_p()
        nat.check(self._lib, self._lib.fwg_reset(self._handle, m.ptr(mask_t) if mask_t is not None else null,
                                                 m.ptr(st) if st is not None else null,
                                                 m.ptr(tg) if tg is not None else null, m.ptr(self._obs_buf), m.stream()))
        self._refresh_obs_view()
        if st is not None or tg is not None or mask_t is not None:
            m.sync()  # the temporaries above must outlive the launch
        del n_sel
        return self._out(self._obs, (N,) + self.obs_shape)

    def obs_dense(self, out=None):
        """Dense [N, obs_dim] copy of the current observation.  Row-log mode: gathered on the device from the window of
        the last completed step (fwg_obs_gather; the position is read on the device in graph mode, so the call may be
        captured and replayed)."""
        if not self.obs_log_rows:
            return self._obs
        out = self._obs_dense if out is None else out
        nat.check(self._lib, self._lib.fwg_obs_gather(self._handle, self._mem.ptr(self._obs_buf), self._mem.ptr(out),
                                                      self._mem.stream()))
        return out

    def _refresh_obs_view(self, want_obs=True):
        """Row-log mode: the current observation is the window [plane, plane + length) of the log, env-major view
        [N, length, n_obs] (zero-copy; valid until the next step/reset).  In graph mode a host-computed window would go
        stale under replay, so the observation handed out is the dense copy gathered on the device instead."""
        if not self.obs_log_rows:
            return
        if self._graph_mode and getattr(self, "_graph_obs", "gather") == "gather":
            if want_obs:
                self._obs = self.obs_dense()
            return
        if self._graph_mode and not want_obs:
            return
        if self._graph_mode and getattr(self, "_capturing", False) and self._cap_token is not None:
            self._cap_token.uses_views = True   # (replay_check: the replays must keep the capture's phase of the window period)
        plane = ctypes.c_int64()
        nat.check(self._lib, self._lib.fwg_obs_window(self._handle, ctypes.byref(plane)))
        win = self._obs_buf[plane.value:plane.value + self._c.obs_length]
        self._obs = win.permute(1, 0, 2) if hasattr(win, "permute") else win.transpose(1, 0, 2)

    def step_async(self, actions):
        m = self._mem
        act = m.as_device(actions).reshape(self.num_envs, 3)
        if self.check_actions:
            nat.check(self._lib, self._lib.fwg_check_actions(self._handle, m.ptr(act), m.stream()))
        nat.check(self._lib, self._lib.fwg_step(self._handle, m.ptr(act), m.ptr(self._obs_buf), m.ptr(self._rew), m.ptr(self._done),
                                                m.ptr(self._term), m.ptr(self._term_obs), m.ptr(self._metrics),
                                                m.ptr(self._target), m.stream()))
        self._refresh_obs_view()
        self._pending = act  # keeps the action buffer alive until the kernel has consumed it

    def step_wait(self):
        N = self.num_envs
        infos = LazyInfos(self, self._done, self._term, self._target, self._metrics, self._term_obs)
        return (self._out(self._obs, (N,) + self.obs_shape), self._out(self._rew), self._out(self._done), infos)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def step_device(self, actions, want_obs=True):
        """Fast path for on-device rollouts: no info objects; returns the (obs, reward, done) device tensors.
        `actions`: float32, contiguous, [N, 3], on this env's device.  want_obs=False (row-log mode): the caller reads
        the observation out of the log itself (the HIP rollout head does, fwg_actor_set_obs_log), so no view/gather."""
        m = self._mem
        actions = self._checked_actions(actions)
        nat.check(self._lib, self._lib.fwg_step(self._handle, m.ptr(actions), m.ptr(self._obs_buf), m.ptr(self._rew), m.ptr(self._done),
                                                m.ptr(self._term), m.ptr(self._term_obs), m.ptr(self._metrics),
                                                ctypes.c_void_p(), m.stream()))   # targets stay in the state arena
        if want_obs:
            self._refresh_obs_view()
        return self._obs, self._rew, self._done

    def _checked_actions(self, actions):
        """step_device hands the raw pointer to the kernel: anything but a float32 contiguous [N, 3] batch on this device
        would be read as garbage, so it is rejected here (cheap attribute checks, no synchronisation)."""
        shape = tuple(getattr(actions, "shape", ()))
        n = 1
        for d in shape:
            n *= int(d)
        if n != self.num_envs * 3:
            raise ValueError("step_device: actions must hold {} x 3 values, got shape {}".format(self.num_envs, shape))
        if hasattr(actions, "is_contiguous"):   # torch tensor
            import torch
            if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self._mem.device:
                raise ValueError("step_device: actions must be a contiguous float32 tensor on {} (got {} {} contiguous={})"
                                 .format(self._mem.device, actions.dtype, actions.device, actions.is_contiguous()))
        elif isinstance(actions, np.ndarray):
            if actions.dtype != np.float32 or not actions.flags["C_CONTIGUOUS"]:
                raise ValueError("step_device: actions must be a C-contiguous float32 array")
        return actions

    # ------------------------------------------------------------------------------------------------------------------
    def finish_episodes(self):
        """Turns the finished-episode records of the envs that ended since the last call into the metrics block (and the
        success sums): fwg_finish_episodes, one small launch, stream-ordered.  The step itself only parks the accumulators."""
        nat.check(self._lib, self._lib.fwg_finish_episodes(self._handle, self._mem.ptr(self._metrics), self._mem.stream()))

    def metrics(self):
        """Device tensor [N_METRICS, N] of the episodic metrics, valid for every env that has reported done: collects the
        finished-episode records first (fwg_step itself only parks them, so reading `_metrics` right after a step would show
        the previous episode's column or NaN).  Stream-ordered, no synchronisation."""
        self.finish_episodes()
        return self._metrics

    def _host(self, name, t):
        cache = self.__dict__.setdefault("_host_cache", {})
        key = (name, int(self._lib.fwg_global_step(self._handle)))
        if cache.get("key") != key:
            cache.clear()
            cache["key"] = key
        if name not in cache:
            if name == "metrics":
                self.finish_episodes()
            cache[name] = self._mem.to_host(t)
        return cache[name]

    def _build_info(self, i, done, term, target, metrics, term_obs):
        info = {}
        d = self._host("done", done)
        tg = self._host("target", target)
        if d[i]:
            info["termination"] = nat.term_name(self._host("term", term)[i])
            mt = self._host("metrics", metrics)[:, i]
            info.update(self.metrics_dict(mt))
            if self.auto_reset:
                info["terminal_observation"] = self._host("term_obs", term_obs)[i].reshape(self.obs_shape)
                info["TimeLimit.truncated"] = info["termination"] == "steps"
        info["target"] = {n: float(tg[i, k]) for k, n in enumerate(self.target_names)}
        return info

    def metrics_dict(self, mt):
        """Column of the metrics block -> the info entries of the reference (get_metric, fixed_wing.py:1095-1162)."""
        names = self.target_names
        with_all = names + ["all"]
        has_bound = [t.get("bound", None) is not None for t in self.cfg["target"]["states"]]
        out = {}
        for metric in self.cfg.get("metrics", []):
            n = metric["name"]
            if n == "rise_time":
                out[n] = {s: float(mt[nat.M_RISE_TIME + k]) for k, s in enumerate(names)}
            elif n == "settling_time":
                out[n] = {s: float(mt[nat.M_SETTLING_TIME + (k if s != "all" else 3)]) for k, s in enumerate(with_all)
                          if s == "all" or has_bound[k]}
            elif n == "success":
                out[n] = {s: bool(mt[nat.M_SUCCESS + (k if s != "all" else 3)] == 1.0) for k, s in enumerate(with_all)
                          if s == "all" or has_bound[k]}
            elif n == "success_time_frac":
                out[n] = {s: float(mt[nat.M_SUCCESS_TIME_FRAC + (k if s != "all" else 3)]) for k, s in enumerate(with_all)
                          if s == "all" or has_bound[k]}
            elif n == "overshoot":
                out[n] = {s: float(mt[nat.M_OVERSHOOT + k]) for k, s in enumerate(names)}
            elif n == "total_error":
                out[n] = {s: float(mt[nat.M_TOTAL_ERROR + k]) for k, s in enumerate(names)}
            elif n == "avg_error":
                out[n] = {s: float(mt[nat.M_AVG_ERROR + k]) for k, s in enumerate(names)}
            elif n == "end_error":
                out[n] = {s: float(mt[nat.M_END_ERROR + k]) for k, s in enumerate(names)}
            elif n == "control_variation":
                out[n] = {"all": float(mt[nat.M_CONTROL_VARIATION])}
            else:
                out[n] = {}
        return out

    # ------------------------------------------------------------------------------------------------------------------
    def word(self, w):
        """Device view [N] of word w of every env (strided view into the grouped arena, no copy)."""
        return self.state[w >> 2, :, w & 3]

    def field(self, name):
        """Device view [N] of one simulator variable or bookkeeping field of the state arena (no copy)."""
        L = self.layout
        sim = {"e0": 0, "e1": 1, "e2": 2, "e3": 3, "omega_p": 4, "omega_q": 5, "omega_r": 6, "position_n": 7,
               "position_e": 8, "position_d": 9, "velocity_u": 10, "velocity_v": 11, "velocity_w": 12,
               "elevon_right": 13, "elevon_left": 14, "throttle": 15, "elevon_right_dot": 16, "elevon_left_dot": 17}
        derived = {"roll": 0, "pitch": 1, "yaw": 2, "Va": 3, "alpha": 4, "beta": 5}
        if name in sim:
            return self.word(L.sim + sim[name])
        if name in derived:
            if not self.derived_views:
                raise KeyError("{} needs FixedWingVecEnv(derived_views=True)".format(name))
            return self.word(L.derived + derived[name])
        if name in ("wind_n", "wind_e", "wind_d"):
            return self.word(L.cold + ("wind_n", "wind_e", "wind_d").index(name))
        if name == "elevator":
            return 0.5 * (self.word(L.sim + 13) + self.word(L.sim + 14))
        if name == "aileron":
            return 0.5 * (self.word(L.sim + 14) - self.word(L.sim + 13))
        if name.startswith("target_"):
            return self.word(L.gym + self.target_names.index(name[7:]))
        iw = self._mem.view_i32(self.state)
        if name == "steps_count":
            return iw[(L.gym + 3) >> 2, :, (L.gym + 3) & 3] & 0xFFFF
        if name == "steps_for_target":
            return (iw[(L.gym + 3) >> 2, :, (L.gym + 3) & 3] >> 16) & 0xFFFF
        if name == "flags":
            return iw[(L.gym + 4) >> 2, :, (L.gym + 4) & 3]
        if name == "episode":
            return iw[(L.cold + 3) >> 2, :, (L.cold + 3) & 3]
        raise KeyError(name)

    def get_state(self, names):
        """Host copy {name: array[N]} of simulator variables (render, tests, checkpoints)."""
        return {n: np.array(self._mem.to_host(self.field(n))) for n in names}

    def get_simulator_parameters(self, normalize=True):
        """fixed_wing.py:872-888 for every env: [N, n] host array of the aircraft parameters simulator["model"] sampled for
        the current episodes, in list order (parameters whose original value is 0 are never sampled and, for relative
        spreads, left out as in the reference); normalised as (value - original) / var with the reference's signed var."""
        model = self.cfg["simulator"].get("model", None)
        if model is None:
            return np.zeros((self.num_envs, 0), dtype=np.float64)
        L, cols, j = self.layout, [], 0
        for pa in model["parameters"]:
            orig_file = float(self.env_config.params[pa["name"]])
            sampled = orig_file != 0
            val = np.array(self._mem.to_host(self.word(L.model_raw + j)), dtype=np.float64) if sampled \
                else np.full(self.num_envs, orig_file)
            j += int(sampled)
            if normalize:   # (the reference records `original` in the config at the first reset: the parameter file's value)
                var = pa.get("var", model["var"])
                original_value = pa.get("original", orig_file)
                if model.get("var_type", "relative") == "relative":
                    if original_value == 0:
                        continue
                    var = var * original_value
                val = (val - original_value) / var
            cols.append(val)
        return np.stack(cols, axis=1) if cols else np.zeros((self.num_envs, 0), dtype=np.float64)

    def reduce_success(self):
        """Local sums over the episodes finished since the last call (see fwg_reduce_success)."""
        out = (ctypes.c_float * nat.N_REDUCE)()
        nat.check(self._lib, self._lib.fwg_reduce_success(self._handle, out, self._mem.stream()))
        return np.array(out[:], dtype=np.float64)

    def reduce_success_device(self, out=None):
        """The same sums as a device tensor [16], stream-ordered, no synchronisation (feed it to the all-gather)."""
        out = self._mem.zeros((nat.N_REDUCE,)) if out is None else out
        nat.check(self._lib, self._lib.fwg_reduce_success_device(self._handle, self._mem.ptr(out), self._mem.stream()))
        return out

    # stable-baselines VecEnv surface used by the example scripts ---------------------------------------------------
    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        if attr_name == "simulator":
            return [self.env_config] * n
        return [getattr(self, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        if method_name == "set_curriculum_level":
            self.set_curriculum_level(*args, **kwargs)
            return [None] * self.num_envs
        if method_name == "reset":
            state, target = kwargs.pop("state", None), kwargs.pop("target", None)
            obs = self.reset(indices=indices, states=state, targets=target)
            idx = range(self.num_envs) if indices is None else np.atleast_1d(indices)
            return [obs[i] for i in idx]
        if method_name == "render":
            return [None]
        raise NotImplementedError(method_name)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs
