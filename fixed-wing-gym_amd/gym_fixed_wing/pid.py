"""Batched baseline PID attitude/airspeed controller (device tensors).

The reference evaluates a PID baseline with `pyfly.pid_controller.PIDController`
(examples/evaluate_controller.py:6,82-84,120-124,143-150; fixed_wing.py:1281-1300), which is part of the absent PyFly
package; this is the same control law for N aircraft at once (gains as recalled in SURVEY.md App. B.2: roll PD on
(phi, p), pitch PID on (theta, q), PI airspeed -> throttle; outputs clipped to the actuator ranges)."""
import math


class BatchedPID(object):
    def __init__(self, n, dt=0.01, device=None):
        import torch
        self.torch = torch
        self.n, self.dt, self.device = n, dt, device
        self.k_p_V, self.k_i_V = 0.5, 0.1
        self.k_p_phi, self.k_i_phi, self.k_d_phi = 1.0, 0.0, 0.5
        self.k_p_theta, self.k_i_theta, self.k_d_theta = -4.0, -0.75, -0.1
        self.delta_a_min, self.delta_a_max = math.radians(-30), math.radians(30)
        self.delta_e_min, self.delta_e_max = math.radians(-30), math.radians(35)
        z = lambda: torch.zeros(n, dtype=torch.float32, device=device)
        self.phi_r, self.theta_r, self.va_r = z(), z(), z()
        self.int_va, self.int_roll, self.int_pitch = z(), z(), z()

    def reset(self, mask=None):
        for t in (self.int_va, self.int_roll, self.int_pitch):
            if mask is None:
                t.zero_()
            else:
                t[mask] = 0

    def set_reference(self, phi, theta, va):
        self.phi_r, self.theta_r, self.va_r = phi, theta, va

    def get_action(self, phi, theta, va, omega):
        torch = self.torch
        e_va, e_phi, e_theta = va - self.va_r, phi - self.phi_r, theta - self.theta_r
        p = omega[:, 0]
        q = omega[:, 1] * torch.cos(phi) - omega[:, 2] * torch.sin(phi)
        delta_a = -self.k_p_phi * e_phi - self.k_i_phi * self.int_roll - self.k_d_phi * p
        delta_e = -self.k_p_theta * e_theta - self.k_i_theta * self.int_pitch - self.k_d_theta * q
        delta_t = -self.k_p_V * e_va - self.k_i_V * self.int_va
        self.int_va = self.int_va + self.dt * e_va
        self.int_roll = self.int_roll + self.dt * e_phi
        self.int_pitch = self.int_pitch + self.dt * e_theta
        return torch.stack([delta_e.clamp(self.delta_e_min, self.delta_e_max),
                            delta_a.clamp(self.delta_a_min, self.delta_a_max), delta_t.clamp(0.0, 1.0)], dim=1)
