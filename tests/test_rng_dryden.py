"""Known-answer / closed-form checks of the two stochastic pieces the HIP path and the oracle share, so that a common
error cannot hide behind HIP == oracle parity:

  * Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): the Random123 known-answer
    vectors, for oracle/physics.py here and for the device implementation (csrc/fwgym_dev.h) on the GPU;
  * Dryden turbulence (MIL-F-8785C low-altitude model, the model PyFly's dryden.py implements -- SURVEY.md App. B.2):
    stationary standard deviations and the longitudinal power spectral density of the discretised filter against the
    military-specification formulas for light / moderate / severe (W20 = 15 / 30 / 45 kt at h = 100 m, Va = 25 m/s)."""
import json
import os

import numpy as np
import pytest

from oracle import physics as ph

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key, expected)
PHILOX_KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_oracle_philox_known_answers():
    ctr = np.array([k[0] for k in PHILOX_KAT], dtype=np.uint32)
    key = np.array([k[1] for k in PHILOX_KAT], dtype=np.uint32)
    want = np.array([k[2] for k in PHILOX_KAT], dtype=np.uint32)
    np.testing.assert_array_equal(ph.philox4x32(ctr, key), want)


@pytest.mark.gpu
def test_device_philox_known_answers_and_matches_oracle():
    import ctypes
    import torch
    from gym_fixed_wing import _native as nat
    lib = nat.load_library()
    rng = np.random.default_rng(0)
    extra = rng.integers(0, 2 ** 32, size=(4096, 6), dtype=np.uint64).astype(np.uint32)
    kat = np.array([list(k[0]) + list(k[1]) for k in PHILOX_KAT], dtype=np.uint32)
    inp = np.concatenate([kat, extra])
    d_in = torch.as_tensor(inp.view(np.int32)).cuda()
    d_out = torch.zeros((inp.shape[0], 4), dtype=torch.int32, device="cuda")
    nat.check(lib, lib.fwg_selftest_philox(ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(d_out.data_ptr()), inp.shape[0],
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    got = d_out.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got[:3], np.array([k[2] for k in PHILOX_KAT], dtype=np.uint32))
    np.testing.assert_array_equal(got, ph.philox4x32(inp[:, :4], inp[:, 4:]))


# ---- MIL-F-8785C low-altitude Dryden model (h < 1000 ft): intensities and scale lengths (feet, feet/s) ---------------
def mil_f_8785c(intensity, h_m=100.0):
    h = h_m * 3.281
    w20 = {"light": 15.0, "moderate": 30.0, "severe": 45.0}[intensity] * 0.5144 * 3.281   # knots -> ft/s
    sw = 0.1 * w20
    su = sw / (0.177 + 0.000823 * h) ** 0.4
    Lw = h
    Lu = h / (0.177 + 0.000823 * h) ** 1.2
    return {"sigma": (su / 3.281, su / 3.281, sw / 3.281), "L": (Lu / 3.281, Lu / 3.281, Lw / 3.281)}   # metres


def _lyapunov_output_sigma(A, B, C):
    from scipy.linalg import solve_discrete_lyapunov
    P = solve_discrete_lyapunov(A, B @ B.T)
    return np.sqrt(np.diag(C @ P @ C.T))


@pytest.mark.parametrize("intensity", ["light", "moderate", "severe"])
def test_dryden_stationary_sigmas_match_mil_f_8785c(intensity):
    """Exact stationary output covariance of the DISCRETE filter (Lyapunov equation; unit-variance normals in, as the
    kernels feed it) against sigma_u = sigma_v = sigma_w / (0.177 + 0.000823 h)^0.4, sigma_w = 0.1 W20."""
    A, B, C = ph.dryden_discretise(2.1, 0.01, 100.0, 25.0, intensity)
    got = _lyapunov_output_sigma(A, B, C)
    want = mil_f_8785c(intensity)["sigma"]
    np.testing.assert_allclose(got[:3], want, rtol=0.01)   # zero-order hold at dt = 0.01 s: < 1 %
    assert np.all(got[3:] > 0)                             # angular channels present (no closed-form variance: the MIL
    # p/q/r spectra are defined through the spatial derivative of the linear gusts; see the cross-checks below)
    # q_g = d(w_g)/dx / V filtered: the filter's q channel must correlate with w, r with v (signs as in the model)
    from scipy.linalg import solve_discrete_lyapunov
    P = solve_discrete_lyapunov(A, B @ B.T)
    cov = C @ P @ C.T
    assert abs(cov[0, 1]) < 1e-9 and abs(cov[0, 2]) < 1e-9 and abs(cov[1, 2]) < 1e-9   # u, v, w independent


def test_dryden_longitudinal_psd_matches_the_specification():
    """Welch PSD of a 2^20-sample oracle series (u channel, moderate) against
    Phi_u(omega) = sigma_u^2 (2 L_u / (pi V)) / (1 + (L_u omega / V)^2) (one-sided in rad/s)."""
    from scipy.signal import welch
    V, dt, n = 25.0, 0.01, 1 << 20
    A, B, C = ph.dryden_discretise(2.1, dt, 100.0, V, "moderate")
    rng = np.random.default_rng(1)
    x = np.zeros(8)
    a00, b00, c00 = A[0, 0], B[0, 0], C[0, 0]   # the u channel is a scalar first-order block
    noise = rng.standard_normal(n)
    u = np.empty(n)
    xs = 0.0
    for i in range(n):
        u[i] = c00 * xs
        xs = a00 * xs + b00 * noise[i]
    f, pxx = welch(u, fs=1.0 / dt, nperseg=1 << 14)           # one-sided PSD per Hz
    m = mil_f_8785c("moderate")
    su, Lu = m["sigma"][0], m["L"][0]
    w = 2 * np.pi * f
    spec = su ** 2 * (2 * Lu / (np.pi * V)) / (1 + (Lu * w / V) ** 2) * 2 * np.pi   # per rad/s -> per Hz
    sel = (f > 0.02) & (f < 5.0)                                # well below Nyquist (50 Hz), above the window resolution
    ratio = pxx[sel] / spec[sel]
    assert abs(np.median(ratio) - 1.0) < 0.05 and np.all(np.abs(np.log(ratio)) < 0.5)
    np.testing.assert_allclose(np.std(u), su, rtol=0.03)


@pytest.mark.gpu
@pytest.mark.parametrize("intensity", ["light", "moderate", "severe"])
def test_device_turbulence_variances_at_full_size(intensity):
    """65 536 envs x 2 000 steps on the GPU: per-channel variance of the gust the kernels produce (C x of the Dryden
    state in the arena) within 2 % of the discrete filter's stationary value, which the test above ties to MIL-F-8785C."""
    import torch
    import configs
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    n = 65536
    vec = FixedWingVecEnv(configs.reference_like("cnn"), num_envs=n, device=0, seed=3, derived_views=False,
                          config_kw={"steps_max": 5000}, sim_config_kw={"turbulence": True, "turbulence_intensity": intensity})
    vec.reset()
    A, B, C = ph.dryden_discretise(2.1, 0.01, 100.0, 25.0, intensity)
    want = _lyapunov_output_sigma(A, B, C) ** 2
    Ct = torch.as_tensor(C, dtype=torch.float32, device="cuda")
    act = torch.zeros((n, 3), device="cuda")
    act[:, 2] = 0.4
    s1 = torch.zeros(6, dtype=torch.float64, device="cuda")
    s2 = torch.zeros(6, dtype=torch.float64, device="cuda")
    i1 = torch.zeros(6, dtype=torch.float64, device="cuda")
    i2 = torch.zeros(6, dtype=torch.float64, device="cuda")
    assert vec.env_config.turbulence_output == "increment"   # the default: the gust sample kept in the simulator rows
    cnt = 0
    L = vec.layout
    for t in range(2000):
        vec.step_device(act, want_obs=False)
        if t >= 400 and t % 8 == 0:     # past the filter's transient (L_u / V ~ 21 s would need more: start from the
            x = torch.stack([vec.word(L.sim + 18 + k) for k in range(8)], dim=1)   # stationary part measured below)
            g = (x @ Ct.T).double()
            s1 += g.sum(dim=0); s2 += (g * g).sum(dim=0); cnt += n
            inc = torch.stack([vec.word(L.sim + 26 + k) for k in range(6)], dim=1).double()   # the next step's gust sample
            i1 += inc.sum(dim=0); i2 += (inc * inc).sum(dim=0)
    var = (s2 / cnt - (s1 / cnt) ** 2).cpu().numpy()
    # the filter starts from x = 0 at reset: after k steps the state covariance is P_k = sum_{j<k} A^j B B^T A^jT; use the
    # exact finite-time value averaged over the sampled steps (the slow u/v channels have not reached P_inf after 20 s)
    P = np.zeros((8, 8)); acc = np.zeros((8, 8)); acc_inc = np.zeros((8, 8)); m = 0
    BB = B @ B.T
    AI = A - np.eye(8)
    for k in range(1, 2001):
        Pn = A @ P @ A.T + BB
        if k - 1 >= 400 and (k - 1) % 8 == 0:
            acc += Pn; m += 1
            acc_inc += AI @ P @ AI.T + BB      # covariance of x_k - x_(k-1) = (A - I) x_(k-1) + B n
        P = Pn
    want_t = np.diag(C @ (acc / m) @ C.T)
    np.testing.assert_allclose(var, want_t, rtol=0.02)
    # the gust the simulator USES (turbulence_output "increment"): the first difference of those outputs, white to within
    # the filters' mean reversion; its longitudinal component is the (K_u / T_u) dt sqrt(pi / dt) of the published traces
    var_inc = (i2 / cnt - (i1 / cnt) ** 2).cpu().numpy()
    np.testing.assert_allclose(var_inc, np.diag(C @ (acc_inc / m) @ C.T), rtol=0.02)
    w20 = {"light": 15.0, "moderate": 30.0, "severe": 45.0}[intensity]
    assert abs(np.sqrt(var_inc[0]) - 0.0031 * w20) < 0.05 * 0.0031 * w20
    assert np.all(want_t[:3] <= want[:3] * 1.0001)
    vec.close()
