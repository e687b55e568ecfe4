"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- float64 NumPy restatement of the flight-simulator half of the hot path.

The reference delegates this arithmetic to the un-vendored third-party package ``pyfly-fixed-wing==0.1.2``
(reference setup.py:27; imported at gym_fixed_wing/fixed_wing.py:3, constructed :46, stepped :358, reset :308).
Its source is absent from /root/reference, so this file restates the *published* model it implements
(Beard & McLain small-UAV 6-DOF quaternion equations, Gryte et al. 2018 Skywalker X8 aerodynamics with the
flat-plate stall blend, MIL-F-8785C Dryden turbulence) behind the interface the reference's call sites need
(SURVEY.md App. B.1).  PARITY UNPINNED versus real PyFly 0.1.2: the reference holds no tests or golden vectors at this
boundary.  What the reference does ship about it -- the converged-airspeed lines hard-coded in its Va target, the per-step
rewards of its PID and MLP evaluations, the jitter statistics of its turbulence evaluations -- is gated two-sidedly by
tests/test_simulator_pins.py (CPU, this file) and tests/test_evaluate.py (GPU): DESIGN.md section 2 A-D.
What IS pinned exactly: the HIP kernels must reproduce THIS file (same fixed-step scheme) to 1e-5 relative in fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Everything is vectorised over a leading env axis N (N=1 when used as the PyFly stand-in under the verbatim
reference gym code) and computed in float64.

State vector y[N,18]: quaternion e0..e3 | omega p,q,r | position n,e,d | body velocity u,v,w |
                      elevon_right, elevon_left, throttle values | elevon_right, elevon_left rates.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------------------------------
# variable table shared by oracle, host and kernels (index = "var id"); the first 21 are the keys of the reference's
# test-set "state" records, i.e. the kwargs accepted by reset(state=...) (fixed_wing.py:287,308)
# ----------------------------------------------------------------------------------------------------------------------
VARS = ["roll", "pitch", "yaw", "omega_p", "omega_q", "omega_r", "position_n", "position_e", "position_d",
        "velocity_u", "velocity_v", "velocity_w", "Va", "alpha", "beta", "elevator", "aileron", "throttle",
        "wind_n", "wind_e", "wind_d", "elevon_right", "elevon_left"]
VAR_ID = {n: i for i, n in enumerate(VARS)}
N_VARS = len(VARS)
TERM_NAN = 250  # failure code used when a state becomes non-finite

IQ, IW, IP, IV, IA, IAD = slice(0, 4), slice(4, 7), slice(7, 10), slice(10, 13), slice(13, 16), slice(16, 18)
NY = 18

AERO_KEYS = ["mass", "Jx", "Jy", "Jz", "Jxz", "S_wing", "b", "c", "S_prop", "C_prop", "k_motor", "k_T_P", "k_Omega",
             "e", "ar", "M", "a_0", "C_L_0", "C_L_alpha", "C_L_q", "C_L_delta_e", "C_D_p", "C_D_beta1", "C_D_beta2",
             "C_D_q", "C_D_delta_e", "C_m_0", "C_m_alpha", "C_m_q", "C_m_delta_e", "C_m_fp", "C_Y_0", "C_Y_beta",
             "C_Y_p", "C_Y_r", "C_Y_delta_a", "C_Y_delta_r", "C_l_0", "C_l_beta", "C_l_p", "C_l_r", "C_l_delta_a",
             "C_l_delta_r", "C_n_0", "C_n_beta", "C_n_p", "C_n_r", "C_n_delta_a", "C_n_delta_r"]


class SimSpec:
    """Constants of one simulator instance (float64)."""

    def __init__(self, sim_cfg, params, dryden_span=None, inertia=None):
        # dryden_span: wingspan the turbulence filters are built with (the parameter FILE's value when simulator.model
        # re-samples the table: the filters are set up once, csrc lower_config / config.dryden_matrices)
        # inertia: (Jx, Jy, Jz, Jxz) the inertia matrix and its gammas are built with -- PyFly 0.1.2 builds self.I and
        # self.gammas once in __init__ from the parameter FILE, so a later write of Jx.. into simulator.params
        # (fixed_wing.py:532-559) changes what get_simulator_parameters returns but not the rotational dynamics
        self.dryden_span = None if dryden_span is None else float(dryden_span)
        self.inertia = None if inertia is None else tuple(float(x) for x in inertia)
        self.dt = float(sim_cfg["dt"])
        self.rho = float(sim_cfg["rho"])
        self.g = float(sim_cfg["g"])
        integ = sim_cfg.get("integrator", {})
        assert integ.get("method", "rk4") == "rk4"
        self.nsub = int(integ.get("substeps", 1))
        self.act_micro = int(integ.get("actuator_microsteps", 16))
        assert self.act_micro % (2 * self.nsub) == 0, "actuator_microsteps must be a multiple of 2*substeps"
        self.params = {k: float(params[k]) for k in AERO_KEYS}
        self.turbulence = bool(sim_cfg.get("turbulence", False))
        self.turbulence_intensity = sim_cfg.get("turbulence_intensity", "light")
        # what enters the airspeed / body rates: "increment" = the first difference of the filter outputs (what PyFly 0.1.2's
        # published traces show, see dryden_gust_after_advance), "filter" = the MIL-F-8785C outputs themselves
        self.turbulence_output = sim_cfg.get("turbulence_output", "increment")
        assert self.turbulence_output in ("increment", "filter")
        self.turb_h = float(sim_cfg.get("turbulence_nominal_altitude", 100.0))
        self.turb_va = float(sim_cfg.get("turbulence_nominal_airspeed", 25.0))
        # per-variable limits, radians; NaN = absent
        self.con_min = np.full(N_VARS, np.nan)
        self.con_max = np.full(N_VARS, np.nan)
        self.val_min = np.full(N_VARS, np.nan)
        self.val_max = np.full(N_VARS, np.nan)
        self.init_min = np.full(N_VARS, np.nan)
        self.init_max = np.full(N_VARS, np.nan)
        self.wrap = np.zeros(N_VARS, dtype=bool)
        self.act = {}
        for st in sim_cfg["states"]:
            i = VAR_ID[st["name"]]
            conv = np.radians if st.get("unit", "") in ("degrees", "degrees/s") else (lambda x: x)
            for prop, arr in (("constraint_min", self.con_min), ("constraint_max", self.con_max),
                              ("value_min", self.val_min), ("value_max", self.val_max),
                              ("init_min", self.init_min), ("init_max", self.init_max)):
                if st.get(prop, None) is not None:
                    arr[i] = conv(float(st[prop]))
            self.wrap[i] = bool(st.get("wrap", False))
            if "order" in st:
                a = {"order": int(st["order"])}
                if a["order"] == 1:
                    a["tau"] = float(st["tau"])
                else:
                    a["omega_0"] = float(st["omega_0"])
                    a["zeta"] = float(st["zeta"])
                a["dot_max"] = conv(float(st["dot_max"])) if st.get("dot_max", None) is not None else np.inf
                self.act[st["name"]] = a
        assert sim_cfg["actuation"]["dynamics"] == ["elevon_right", "elevon_left", "throttle"]
        self.inputs = list(sim_cfg["actuation"]["inputs"])
        self._dryden = None

    # -- derived constants ------------------------------------------------------------------------------------------
    def inertia_values(self):
        p = self.params
        return self.inertia if self.inertia is not None else (p["Jx"], p["Jy"], p["Jz"], p["Jxz"])

    def gammas(self):
        Jx, Jy, Jz, Jxz = self.inertia_values()
        G = Jx * Jz - Jxz ** 2
        return (Jxz * (Jx - Jy + Jz) / G, (Jz * (Jz - Jy) + Jxz ** 2) / G, Jz / G, Jxz / G, (Jz - Jx) / Jy, Jxz / Jy,
                ((Jx - Jy) * Jx + Jxz ** 2) / G, Jx / G)

    def dryden(self):
        if self._dryden is None:
            self._dryden = dryden_discretise(self.params["b"] if self.dryden_span is None else self.dryden_span, self.dt, self.turb_h, self.turb_va,
                                             self.turbulence_intensity)
        return self._dryden


def _lim(x, lo, hi):
    """np.clip that treats NaN limits as absent."""
    if not np.isnan(lo):
        x = np.maximum(x, lo)
    if not np.isnan(hi):
        x = np.minimum(x, hi)
    return x


# ----------------------------------------------------------------------------------------------------------------------
# actuation (SURVEY.md App. B.2 "Actuation"): elevator/aileron <-> elevons, command constraints
# ----------------------------------------------------------------------------------------------------------------------
def constrain_commands(spec, cmd):
    """cmd[N,3] in the order of spec.inputs (elevator, aileron, throttle) -> (constrained inputs [N,3] as stored in
    the simulator's "command" history (fixed_wing.py:828,1110), dynamics set-points [N,3] = elevon_right, elevon_left,
    throttle)."""
    cmd = np.asarray(cmd, dtype=np.float64)
    e, a, t = cmd[:, 0], cmd[:, 1], cmd[:, 2]
    ie, ia, it = VAR_ID["elevator"], VAR_ID["aileron"], VAR_ID["throttle"]
    e = _lim(e, spec.val_min[ie], spec.val_max[ie])
    a = _lim(a, spec.val_min[ia], spec.val_max[ia])
    t = _lim(t, spec.val_min[it], spec.val_max[it])
    er = _lim(e - a, spec.val_min[VAR_ID["elevon_right"]], spec.val_max[VAR_ID["elevon_right"]])
    el = _lim(e + a, spec.val_min[VAR_ID["elevon_left"]], spec.val_max[VAR_ID["elevon_left"]])
    e_c, a_c = 0.5 * (er + el), 0.5 * (el - er)
    return np.stack([e_c, a_c, t], axis=1), np.stack([er, el, t], axis=1)


def elevons_from_inputs(e, a):
    return e - a, e + a  # right, left


# ----------------------------------------------------------------------------------------------------------------------
# kinematics helpers
# ----------------------------------------------------------------------------------------------------------------------
def quat_from_euler(roll, pitch, yaw):
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    return np.stack([cy * cp * cr + sy * sp * sr, cy * cp * sr - sy * sp * cr,
                     cy * sp * cr + sy * cp * sr, sy * cp * cr - cy * sp * sr], axis=-1)


def euler_from_quat(q):
    e0, e1, e2, e3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    roll = np.arctan2(2 * (e0 * e1 + e2 * e3), e0 ** 2 + e3 ** 2 - e1 ** 2 - e2 ** 2)
    pitch = np.arcsin(np.clip(2 * (e0 * e2 - e1 * e3), -1.0, 1.0))
    yaw = np.arctan2(2 * (e0 * e3 + e1 * e2), e0 ** 2 + e1 ** 2 - e2 ** 2 - e3 ** 2)
    return roll, pitch, yaw


def rot_body_to_ned(q):
    """Rows of R (v_ned = R v_body) for unit quaternion q[N,4]; returns 9 arrays."""
    e0, e1, e2, e3 = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r00 = e0 * e0 + e1 * e1 - e2 * e2 - e3 * e3
    r01 = 2 * (e1 * e2 - e0 * e3)
    r02 = 2 * (e1 * e3 + e0 * e2)
    r10 = 2 * (e1 * e2 + e0 * e3)
    r11 = e0 * e0 - e1 * e1 + e2 * e2 - e3 * e3
    r12 = 2 * (e2 * e3 - e0 * e1)
    r20 = 2 * (e1 * e3 - e0 * e2)
    r21 = 2 * (e2 * e3 + e0 * e1)
    r22 = e0 * e0 - e1 * e1 - e2 * e2 + e3 * e3
    return r00, r01, r02, r10, r11, r12, r20, r21, r22


def airspeed_factors(q, vel, wind_ned, gust_lin):
    """Va, alpha, beta (+ the body-frame airspeed vector) -- SURVEY.md App. B.2 "Airspeed" (confirmed against the
    shipped test set: Va=|v_a|, alpha=atan2(w_a,u_a), beta=asin(v_a/Va))."""
    r00, r01, r02, r10, r11, r12, r20, r21, r22 = rot_body_to_ned(q)
    wn, we, wd = wind_ned[:, 0], wind_ned[:, 1], wind_ned[:, 2]
    # body <- ned is R^T
    ua = vel[:, 0] - (r00 * wn + r10 * we + r20 * wd) - gust_lin[:, 0]
    va = vel[:, 1] - (r01 * wn + r11 * we + r21 * wd) - gust_lin[:, 1]
    wa = vel[:, 2] - (r02 * wn + r12 * we + r22 * wd) - gust_lin[:, 2]
    Va = np.sqrt(ua * ua + va * va + wa * wa)
    alpha = np.arctan2(wa, ua)
    with np.errstate(invalid="ignore", divide="ignore"):
        beta = np.arcsin(np.clip(np.where(Va > 0, va / np.where(Va > 0, Va, 1.0), 0.0), -1.0, 1.0))
    return Va, alpha, beta, ua, va, wa


# ----------------------------------------------------------------------------------------------------------------------
# right-hand side
# ----------------------------------------------------------------------------------------------------------------------
_STAGE_CHECK = ["omega_p", "omega_q", "omega_r", "position_n", "position_e", "position_d",
                "velocity_u", "velocity_v", "velocity_w"]


def _check(spec, fail, name, x):
    i = VAR_ID[name]
    bad = np.zeros(x.shape, dtype=bool)
    if not np.isnan(spec.con_min[i]):
        bad |= x < spec.con_min[i]
    if not np.isnan(spec.con_max[i]):
        bad |= x > spec.con_max[i]
    return np.where((fail == 0) & bad, i + 1, fail)


def rhs(spec, yb, act, wind_ned, gust, fail):
    """d/dt of the 13 rigid-body states yb[N,13] (quaternion, omega, position, body velocity) for actuator deflections
    act[N,3] = (elevon_right, elevon_left, throttle) at the stage time; gust[N,6] = body-frame Dryden sample
    (u,v,w linear, p,q,r angular) held over the env step.
    `fail` is the sticky per-env failure code (0 = ok, var_id+1 = first violated constraint)."""
    P = spec.params
    q4, om, vel = yb[:, IQ], yb[:, IW], yb[:, IV]
    p, q, r = om[:, 0], om[:, 1], om[:, 2]
    u, v, w = vel[:, 0], vel[:, 1], vel[:, 2]

    # constraint checks on the stage state (PyFly applies Variable constraints whenever states are set from the
    # ODE solution, also at intermediate stages -- SURVEY.md App. B.2 "Constraints")
    stage_vals = {"omega_p": p, "omega_q": q, "omega_r": r, "position_n": yb[:, 7], "position_e": yb[:, 8],
                  "position_d": yb[:, 9], "velocity_u": u, "velocity_v": v, "velocity_w": w}
    for name in _STAGE_CHECK:
        fail = _check(spec, fail, name, stage_vals[name])

    er, el, th = act[:, 0], act[:, 1], act[:, 2]
    elevator, aileron = 0.5 * (er + el), 0.5 * (el - er)
    rudder = 0.0

    Va, alpha, beta, ua, va, wa = airspeed_factors(q4, vel, wind_ned, gust[:, 0:3])
    fail = _check(spec, fail, "Va", Va)
    fail = _check(spec, fail, "alpha", alpha)
    fail = _check(spec, fail, "beta", beta)
    Va = _lim(Va, spec.val_min[VAR_ID["Va"]], spec.val_max[VAR_ID["Va"]])

    pa, qa, ra = p - gust[:, 3], q - gust[:, 4], r - gust[:, 5]

    pre = 0.5 * spec.rho * Va * Va * P["S_wing"]
    e0, e1, e2, e3 = q4[:, 0], q4[:, 1], q4[:, 2], q4[:, 3]
    mg = P["mass"] * spec.g
    fgx = mg * 2 * (e1 * e3 - e2 * e0)
    fgy = mg * 2 * (e2 * e3 + e1 * e0)
    fgz = mg * (e3 * e3 + e0 * e0 - e1 * e1 - e2 * e2)

    # stall blend, overflow-safe form of sigma = (1+ea+eb)/((1+ea)(1+eb)), ea=exp(-M(a-a0)), eb=exp(M(a+a0)):
    # 1-sigma = 1/((1+exp(M(a-a0)))(1+exp(-M(a+a0))))
    M, a0 = P["M"], P["a_0"]
    with np.errstate(over="ignore"):
        oms = 1.0 / ((1.0 + np.exp(M * (alpha - a0))) * (1.0 + np.exp(-M * (alpha + a0))))
    sig = 1.0 - oms
    rxz = np.sqrt(ua * ua + wa * wa)
    with np.errstate(invalid="ignore", divide="ignore"):
        inv_rxz = np.where(rxz > 0, 1.0 / np.where(rxz > 0, rxz, 1.0), 0.0)
        inv_Va_air = np.where(rxz > 0, 1.0 / np.sqrt(ua * ua + va * va + wa * wa), 0.0)
    ca = np.where(rxz > 0, ua * inv_rxz, 1.0)
    sa = wa * inv_rxz
    sb = va * inv_Va_air
    cb = np.where(rxz > 0, rxz * inv_Va_air, 1.0)
    sgn = np.sign(alpha)

    CL_lin = P["C_L_0"] + P["C_L_alpha"] * alpha
    CL = oms * CL_lin + sig * (2.0 * sgn * sa * sa * ca)
    with np.errstate(invalid="ignore", divide="ignore"):
        inv2Va = np.where(Va > 0, 0.5 / np.where(Va > 0, Va, 1.0), 0.0)
    f_lift = pre * (CL + P["C_L_q"] * P["c"] * inv2Va * qa + P["C_L_delta_e"] * elevator)
    CD = P["C_D_p"] + oms * CL_lin * CL_lin / (np.pi * P["e"] * P["ar"]) + sig * (2.0 * sgn * sa * sa * sa)
    CDb = P["C_D_beta1"] * beta + P["C_D_beta2"] * beta * beta
    f_drag = pre * (CD + CDb + P["C_D_q"] * P["c"] * inv2Va * qa + P["C_D_delta_e"] * elevator * elevator)
    Cm = oms * (P["C_m_0"] + P["C_m_alpha"] * alpha) + sig * (P["C_m_fp"] * sgn * sa * sa)
    # NB: the pitch-damping term is scaled with the span b (as in the model the reference was trained on)
    m_ = pre * P["c"] * (Cm + P["C_m_q"] * P["b"] * inv2Va * qa + P["C_m_delta_e"] * elevator)
    bv = P["b"] * inv2Va
    f_y = pre * (P["C_Y_0"] + P["C_Y_beta"] * beta + P["C_Y_p"] * bv * pa + P["C_Y_r"] * bv * ra
                 + P["C_Y_delta_a"] * aileron + P["C_Y_delta_r"] * rudder)
    l_ = pre * P["b"] * (P["C_l_0"] + P["C_l_beta"] * beta + P["C_l_p"] * bv * pa + P["C_l_r"] * bv * ra
                         + P["C_l_delta_a"] * aileron + P["C_l_delta_r"] * rudder)
    n_ = pre * P["b"] * (P["C_n_0"] + P["C_n_beta"] * beta + P["C_n_p"] * bv * pa + P["C_n_r"] * bv * ra
                         + P["C_n_delta_a"] * aileron + P["C_n_delta_r"] * rudder)

    # wind -> body: rotation of (-D, Y, -L) through (alpha, beta)
    fx_a = ca * cb * (-f_drag) - ca * sb * f_y - sa * (-f_lift)
    fy_a = sb * (-f_drag) + cb * f_y
    fz_a = sa * cb * (-f_drag) - sa * sb * f_y + ca * (-f_lift)

    Vd = Va + th * (P["k_motor"] - Va)
    f_prop = 0.5 * spec.rho * P["S_prop"] * P["C_prop"] * Vd * (Vd - Va)
    tau_prop = -P["k_T_P"] * (P["k_Omega"] * th) ** 2

    fx, fy, fz = f_prop + fgx + fx_a, fgy + fy_a, fgz + fz_a
    l_ = l_ + tau_prop

    dy = np.empty_like(yb)
    # quaternion kinematics
    dy[:, 0] = 0.5 * (-p * e1 - q * e2 - r * e3)
    dy[:, 1] = 0.5 * (p * e0 + r * e2 - q * e3)
    dy[:, 2] = 0.5 * (q * e0 - r * e1 + p * e3)
    dy[:, 3] = 0.5 * (r * e0 + q * e1 - p * e2)
    G1, G2, G3, G4, G5, G6, G7, G8 = spec.gammas()
    dy[:, 4] = G1 * p * q - G2 * q * r + G3 * l_ + G4 * n_
    dy[:, 5] = G5 * p * r - G6 * (p * p - r * r) + m_ / spec.inertia_values()[1]
    dy[:, 6] = G7 * p * q - G1 * q * r + G4 * l_ + G8 * n_
    r00, r01, r02, r10, r11, r12, r20, r21, r22 = rot_body_to_ned(q4)
    dy[:, 7] = r00 * u + r01 * v + r02 * w
    dy[:, 8] = r10 * u + r11 * v + r12 * w
    dy[:, 9] = r20 * u + r21 * v + r22 * w
    im = 1.0 / P["mass"]
    dy[:, 10] = r * v - q * w + fx * im
    dy[:, 11] = p * w - r * u + fy * im
    dy[:, 12] = q * u - p * v + fz * im
    return dy, fail


def actuator_transition(spec, hh):
    """Exact discretisation of the actuator models over one micro-step hh: per elevon the 2x2 matrix exponential of
    x' = [[0, 1], [-w0^2, -2 zeta w0]] x with x = (value - command, rate); throttle exp(-hh/tau)."""
    from scipy.linalg import expm
    phis = []
    for name in ("elevon_right", "elevon_left"):
        a = spec.act[name]
        A = np.array([[0.0, 1.0], [-a["omega_0"] ** 2, -2.0 * a["zeta"] * a["omega_0"]]])
        phis.append(expm(A * hh))
    return phis, float(np.exp(-hh / spec.act["throttle"]["tau"]))


def advance_actuators(spec, a, setpoint, hh, trans):
    """a[N,5] = (elevon_right, elevon_left, throttle, elevon_right_rate, elevon_left_rate) advanced by hh with the
    command held: exact linear response, then the rate limit (on the rate AND on the travel over hh) and the value
    limits."""
    phis, ethr = trans
    out = a.copy()
    for k, name in enumerate(("elevon_right", "elevon_left")):
        P_ = phis[k]
        x0, x1 = a[:, k] - setpoint[:, k], a[:, 3 + k]
        v = setpoint[:, k] + P_[0, 0] * x0 + P_[0, 1] * x1
        d = P_[1, 0] * x0 + P_[1, 1] * x1
        dm = spec.act[name]["dot_max"]
        if np.isfinite(dm):
            d = np.clip(d, -dm, dm)
            v = np.clip(v, a[:, k] - dm * hh, a[:, k] + dm * hh)
        out[:, k] = _lim(v, spec.val_min[VAR_ID[name]], spec.val_max[VAR_ID[name]])
        out[:, 3 + k] = d
    th = setpoint[:, 2] + ethr * (a[:, 2] - setpoint[:, 2])
    out[:, 2] = _lim(th, spec.val_min[VAR_ID["throttle"]], spec.val_max[VAR_ID["throttle"]])
    return out


def sanitize_actuators(spec, y):
    y = y.copy()
    for col, name in ((13, "elevon_right"), (14, "elevon_left"), (15, "throttle")):
        y[:, col] = _lim(y[:, col], spec.val_min[VAR_ID[name]], spec.val_max[VAR_ID[name]])
    y[:, 16] = np.clip(y[:, 16], -spec.act["elevon_right"]["dot_max"], spec.act["elevon_right"]["dot_max"])
    y[:, 17] = np.clip(y[:, 17], -spec.act["elevon_left"]["dot_max"], spec.act["elevon_left"]["dot_max"])
    return y


def derive(spec, y, wind_ned, gust):
    """Euler angles, Va/alpha/beta, elevator/aileron for a committed state."""
    roll, pitch, yaw = euler_from_quat(y[:, IQ])
    Va, alpha, beta, _, _, _ = airspeed_factors(y[:, IQ], y[:, IV], wind_ned, gust[:, 0:3])
    elevator, aileron = 0.5 * (y[:, 13] + y[:, 14]), 0.5 * (y[:, 14] - y[:, 13])
    return {"roll": roll, "pitch": pitch, "yaw": yaw, "Va": Va, "alpha": alpha, "beta": beta,
            "elevator": elevator, "aileron": aileron, "throttle": y[:, 15]}


def sim_step(spec, y, cmd_inputs, wind_ned, gust):
    """One env step (dt) of the simulator: the restatement of PyFly.step called at fixed_wing.py:358.

    Returns (y_new, ok[N], fail_code[N], commands_constrained[N,3], derived dict).  Where ok is False y_new keeps the
    last valid state (SURVEY.md section 5 'Failure detection').
    Integration scheme (identical in the HIP kernels): the actuators (stiff: fastest elevon pole ~ -310 1/s, and
    piecewise-linear because of the rate limit) are advanced EXACTLY over `spec.act_micro` equal micro-steps of their
    linear dynamics, with the rate/value limits applied after every micro-step; the 13 rigid-body states take
    `spec.nsub` classical RK4 steps of h = dt/nsub whose stages see the actuator deflections at t, t+h/2, t+h/2, t+h
    (micro-step boundaries).  The quaternion is re-normalised once per env step.
    """
    N = y.shape[0]
    cmd_c, setpoint = constrain_commands(spec, cmd_inputs)
    fail = np.zeros(N, dtype=np.int64)
    h = spec.dt / spec.nsub
    hm = spec.dt / spec.act_micro
    per_half = spec.act_micro // (2 * spec.nsub)
    trans = actuator_transition(spec, hm)
    yb = y[:, 0:13].copy()
    a = sanitize_actuators(spec, y)[:, 13:18]
    for _ in range(spec.nsub):
        a_half = a
        for _m in range(per_half):
            a_half = advance_actuators(spec, a_half, setpoint, hm, trans)
        a_full = a_half
        for _m in range(per_half):
            a_full = advance_actuators(spec, a_full, setpoint, hm, trans)
        k1, fail = rhs(spec, yb, a[:, 0:3], wind_ned, gust, fail)
        k2, fail = rhs(spec, yb + 0.5 * h * k1, a_half[:, 0:3], wind_ned, gust, fail)
        k3, fail = rhs(spec, yb + 0.5 * h * k2, a_half[:, 0:3], wind_ned, gust, fail)
        k4, fail = rhs(spec, yb + h * k3, a_full[:, 0:3], wind_ned, gust, fail)
        yb = yb + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        a = a_full
    yy = np.concatenate([yb, a], axis=1)
    qn = np.sqrt(np.sum(yy[:, IQ] ** 2, axis=1, keepdims=True))
    yy[:, IQ] = yy[:, IQ] / qn
    # end-of-step checks: rigid-body constraints, Euler-angle constraints, airspeed factors
    vals = {"omega_p": yy[:, 4], "omega_q": yy[:, 5], "omega_r": yy[:, 6], "position_n": yy[:, 7],
            "position_e": yy[:, 8], "position_d": yy[:, 9], "velocity_u": yy[:, 10], "velocity_v": yy[:, 11],
            "velocity_w": yy[:, 12]}
    for name in _STAGE_CHECK:
        fail = _check(spec, fail, name, vals[name])
    d = derive(spec, yy, wind_ned, gust)
    for name in ("roll", "pitch", "yaw", "Va", "alpha", "beta"):
        fail = _check(spec, fail, name, d[name])
    finite = np.all(np.isfinite(yy), axis=1)
    fail = np.where((fail == 0) & ~finite, TERM_NAN + 1, fail)
    ok = fail == 0
    y_new = np.where(ok[:, None], yy, y)
    if not np.all(ok):
        d_old = derive(spec, y, wind_ned, gust)
        d = {k: np.where(ok, d[k], d_old[k]) for k in d}
    return y_new, ok, fail - 1, cmd_c, d


def initial_state(spec, values):
    """values: dict var name -> array[N] for the 21 reset variables (Va/alpha/beta entries ignored: derived).
    Returns y[N,18], wind_ned[N,3]."""
    N = len(np.atleast_1d(values["roll"]))
    y = np.zeros((N, NY))
    y[:, IQ] = quat_from_euler(np.asarray(values["roll"], float), np.asarray(values["pitch"], float),
                               np.asarray(values["yaw"], float))
    for j, n in enumerate(["omega_p", "omega_q", "omega_r", "position_n", "position_e", "position_d",
                           "velocity_u", "velocity_v", "velocity_w"]):
        y[:, 4 + j] = values[n]
    e = _lim(np.asarray(values["elevator"], float), spec.val_min[VAR_ID["elevator"]], spec.val_max[VAR_ID["elevator"]])
    a = _lim(np.asarray(values["aileron"], float), spec.val_min[VAR_ID["aileron"]], spec.val_max[VAR_ID["aileron"]])
    y[:, 13], y[:, 14] = e - a, e + a
    y[:, 15] = _lim(np.asarray(values["throttle"], float), spec.val_min[VAR_ID["throttle"]],
                    spec.val_max[VAR_ID["throttle"]])
    wind = np.stack([np.asarray(values["wind_n"], float) * np.ones(N), np.asarray(values["wind_e"], float) * np.ones(N),
                     np.asarray(values["wind_d"], float) * np.ones(N)], axis=1)
    return y, wind


# ----------------------------------------------------------------------------------------------------------------------
# Dryden turbulence (MIL-F-8785C low-altitude model at fixed nominal altitude/airspeed) -- SURVEY.md App. B.2 "Dryden"
# ----------------------------------------------------------------------------------------------------------------------
N_DRY = 8  # joint realisation: u(1) | v,r(3) | w,q(3) | p(1)


def dryden_continuous(b, h=100.0, Va=25.0, intensity="light"):
    """Continuous-time joint state-space (A[8,8], B[8,4], C[6,8]) with inputs (n_u, n_v, n_w, n_p) and outputs
    (u_g, v_g, w_g [m/s], p_g, q_g, r_g [rad/s])."""
    m2f = 3.281
    f2m = 1.0 / m2f
    kn2ms = 0.5144
    h, Va, b = h * m2f, Va * m2f, b * m2f
    W20 = {"light": 15.0, "moderate": 30.0, "severe": 45.0}[intensity] * kn2ms * m2f
    sw = 0.1 * W20
    su = sw / (0.177 + 0.000823 * h) ** 0.4
    sv = su
    Lu = h / (0.177 + 0.000823 * h) ** 1.2
    Lv, Lw = Lu, h
    Ku = su * np.sqrt(2 * Lu / (np.pi * Va))
    Kv = sv * np.sqrt(Lv / (np.pi * Va))
    Kw = sw * np.sqrt(Lw / (np.pi * Va))
    Tu, Tv1, Tv2, Tw1, Tw2 = Lu / Va, np.sqrt(3.0) * Lv / Va, Lv / Va, np.sqrt(3.0) * Lw / Va, Lw / Va
    Kp = sw * np.sqrt(0.8 / Va) * (np.pi / (4 * b)) ** (1.0 / 6.0) / Lw ** (1.0 / 3.0)
    Kq = Kr = 1.0 / Va
    Tp = 4 * b / (np.pi * Va)
    Tq, Tr = Tp, 3 * b / (np.pi * Va)
    A = np.zeros((8, 8))
    B = np.zeros((8, 4))
    C = np.zeros((6, 8))
    # u: Ku/(Tu s+1)
    A[0, 0] = -1 / Tu
    B[0, 0] = 1 / Tu
    C[0, 0] = f2m * Ku

    def second(i0, nz, K, T1, T2, Ta, Ka, lin_row, ang_row, sign):
        # z1' = (n - z1)/T2 ; z2' = (z1 - z2)/T2 ; f = K (T1 (z1 - z2)/T2 + z2) ; z3' = (f - z3)/Ta ;
        # ang = sign*Ka*(f - z3)/Ta
        A[i0, i0] = -1 / T2
        B[i0, nz] = 1 / T2
        A[i0 + 1, i0] = 1 / T2
        A[i0 + 1, i0 + 1] = -1 / T2
        f = np.zeros(8)
        f[i0] = K * T1 / T2
        f[i0 + 1] = K * (1 - T1 / T2)
        A[i0 + 2, :] = f / Ta
        A[i0 + 2, i0 + 2] -= 1 / Ta
        C[lin_row, :] = f2m * f
        C[ang_row, :] = sign * Ka * f / Ta
        C[ang_row, i0 + 2] -= sign * Ka / Ta

    second(1, 1, Kv, Tv1, Tv2, Tr, Kr, 1, 5, +1.0)   # v and r
    second(4, 2, Kw, Tw1, Tw2, Tq, Kq, 2, 4, -1.0)   # w and q
    A[7, 7] = -1 / Tp
    B[7, 3] = 1 / Tp
    C[3, 7] = Kp
    return A, B, C


def dryden_discretise(b, dt, h, Va, intensity):
    """Exact zero-order-hold discretisation; the white-noise input is scaled by sqrt(pi/dt) (folded into Bd)."""
    from scipy.linalg import expm
    A, B, C = dryden_continuous(b, h, Va, intensity)
    Maug = np.zeros((12, 12))
    Maug[:8, :8] = A
    Maug[:8, 8:] = B
    E = expm(Maug * dt)
    return E[:8, :8], E[:8, 8:] * np.sqrt(np.pi / dt), C


def dryden_advance(spec, x, normals):
    Ad, Bd, _ = spec.dryden()
    return x @ Ad.T + normals @ Bd.T


def dryden_output(spec, x):
    return x @ spec.dryden()[2].T


def dryden_gust_after_advance(spec, x_old, x_new):
    """The gust sample (u, v, w, p, q, r) of the NEXT env step, given the filter state before and after this step's advance.

    "filter": C x_new, the MIL-F-8785C signal.  "increment" (default): C (x_new - x_old), its first difference -- the only
    reading of the turbulence samples consistent with the data the reference ships about PyFly 0.1.2: in all six published
    turbulence evaluations with a deterministic or a learnt controller (examples/evaluations/eval_res_{PID,RL_MLP}_{light,
    moderate,severe}.npy) the airspeed term of the per-step reward carries WHITE noise (lag-1 autocorrelation of the
    increments -0.47, against +0.44 for a linearly interpolated and -0.05 for a held Dryden signal) of standard deviation
    0.043 / 0.086 / 0.135 m/s = (K_u / T_u) dt sqrt(pi / dt), i.e. exactly one step's forced response of the longitudinal
    filter, and next to no random-walk component (<= 0.015 m/s per step at light; the specification signal would give 0.033
    to 0.047).  tools/turbulence_scan.py, tests/test_simulator_pins.py::test_turbulence_jitter_fingerprint.  The sample used
    during the first step after a reset is zero in both forms (the filters start from rest)."""
    C = spec.dryden()[2]
    if spec.turbulence_output == "increment":
        return (x_new - x_old) @ C.T
    return x_new @ C.T


# ----------------------------------------------------------------------------------------------------------------------
# Philox4x32-10 counter-based RNG (Salmon et al., Random123) -- the device RNG; restated so that noise streams are
# bit-identical between oracle and kernels.
# ----------------------------------------------------------------------------------------------------------------------
_M0, _M1, _W0, _W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(ctr, key):
    """ctr[N,4] uint32, key[N,2] uint32 -> [N,4] uint32."""
    c = [np.asarray(ctr[:, i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[:, 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[:, 1], dtype=np.uint32).copy()
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _M0 * c[0].astype(np.uint64)
        p1 = _M1 * c[2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + _W0).astype(np.uint32)
        k1 = (k1 + _W1).astype(np.uint32)
    return np.stack(c, axis=1)


def u01(x):
    """uint32 -> uniform in (0,1), exactly representable in fp32: ((x>>8)+0.5)*2^-24."""
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) + 0.5) * (1.0 / 16777216.0)


def box_muller(bits):
    """bits[N,4] uint32 -> normals[N,4]."""
    u = u01(bits)
    r0 = np.sqrt(-2.0 * np.log(u[:, 0]))
    r1 = np.sqrt(-2.0 * np.log(u[:, 2]))
    t0, t1 = 2 * np.pi * u[:, 1], 2 * np.pi * u[:, 3]
    return np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], axis=1)


# stream ids (ctr[3]) shared with the kernels
STREAM_TURB, STREAM_RESET_STATE, STREAM_RESET_TARGET, STREAM_OBS_NOISE, STREAM_INIT_NOISE = 1, 2, 3, 4, 5
STREAM_MODEL = 7   # simulator.model draws (6: the rollout head's policy noise)
STREAM_REWARD_SCALE = 8   # reward.randomize_scaling draws
STREAM_SIM_KEY = 9        # simulator.<key> draws (turbulence, turbulence_intensity)


def rng_bits(seed, env_ids, counter, stream, sub=0):
    """Philox call convention: key=(seed_lo, seed_hi), ctr=(env_id, counter, sub, stream)."""
    env_ids = np.asarray(env_ids, dtype=np.uint32)
    n = env_ids.shape[0]
    ctr = np.stack([env_ids, np.broadcast_to(np.asarray(counter, dtype=np.uint32), (n,)),
                    np.broadcast_to(np.asarray(sub, dtype=np.uint32), (n,)),
                    np.full(n, stream, dtype=np.uint32)], axis=1)
    key = np.stack([np.full(n, seed & 0xFFFFFFFF, dtype=np.uint32),
                    np.full(n, (seed >> 32) & 0xFFFFFFFF, dtype=np.uint32)], axis=1)
    return philox4x32(ctr, key)
