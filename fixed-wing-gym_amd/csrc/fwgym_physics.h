// fwgym_physics.h -- 6-DOF rigid-body + actuator right-hand side, RK4 sub-stepping, Dryden filter (device, fp32).
// Specification: oracle/physics.py (float64).  One lane = one aircraft; every value below lives in VGPRs, the constants
// of DevCfg are wave-uniform scalar loads.
#pragma once
#include <type_traits>
#include "fwgym_dev.h"

#define NY 18  // e0 e1 e2 e3 | p q r | pn pe pd | u v w | elevon_r elevon_l throttle | elevon_r_dot elevon_l_dot

struct Air { float Va, alpha, beta, ua, va, wa; };

// v_ned = R v_body
struct Rot { float r00, r01, r02, r10, r11, r12, r20, r21, r22; };
__device__ __forceinline__ Rot rot_from_quat(float e0, float e1, float e2, float e3) {
    Rot R;
    const float e00 = e0 * e0, e11 = e1 * e1, e22 = e2 * e2, e33 = e3 * e3;
    R.r00 = e00 + e11 - e22 - e33; R.r01 = 2.f * (e1 * e2 - e0 * e3); R.r02 = 2.f * (e1 * e3 + e0 * e2);
    R.r10 = 2.f * (e1 * e2 + e0 * e3); R.r11 = e00 - e11 + e22 - e33; R.r12 = 2.f * (e2 * e3 - e0 * e1);
    R.r20 = 2.f * (e1 * e3 - e0 * e2); R.r21 = 2.f * (e2 * e3 + e0 * e1); R.r22 = e00 - e11 - e22 + e33;
    return R;
}

template <bool TURB>
__device__ __forceinline__ Air airspeed(const Rot& R, float u, float v, float w, const float* wind, const float* gust) {
    Air a;
    a.ua = u - (R.r00 * wind[0] + R.r10 * wind[1] + R.r20 * wind[2]);
    a.va = v - (R.r01 * wind[0] + R.r11 * wind[1] + R.r21 * wind[2]);
    a.wa = w - (R.r02 * wind[0] + R.r12 * wind[1] + R.r22 * wind[2]);
    if (TURB) { a.ua -= gust[0]; a.va -= gust[1]; a.wa -= gust[2]; }
    const float xz2 = a.ua * a.ua + a.wa * a.wa;
    const float v2 = fmaxf(xz2 + a.va * a.va, 1e-30f);
    a.Va = v2 * frsq(v2);
    const f2 ab = fast_atan2x2(mk2(a.wa, a.va), mk2(a.ua, xz2 * frsq(fmaxf(xz2, 1e-30f))));   // beta = asin(va / Va)
    a.alpha = ab[0];
    a.beta = ab[1];
    return a;
}

__device__ __forceinline__ void check_var(const DevCfg& c, int& fail, int var, float x) {
    if (c.con_mask & (1u << var)) {
        const bool bad = (x < V(c).con_min[var]) | (x > V(c).con_max[var]);
        fail = (fail == 0 && bad) ? var + 1 : fail;
    }
}

// Constraint checks of one env step: 4 right-hand sides + the end of the step, each over the variables of c.con_mask in a
// fixed order; the FIRST violation names the termination (oracle/physics.py _check).
//  * ImmediateChecks: the sticky code is updated at every check (3 vector + 1 scalar instruction per variable and check point).
//  * DeferredChecks (specialised kernels, one RK4 step per env step): the checked values are KEPT (registers, no instructions)
//    and looked at once after the last stage -- a violation anywhere is detected with one 3-way max / min per pair of values
//    (|x| against the bound where the interval is symmetric) and a compare per variable; only a wave in which some lane did
//    violate a constraint then names it, in the reference order, from the kept values.  The physics wave's instruction stream
//    IS the length of an env step: this takes ~60 instructions off it.
struct ImmediateChecks {
    int fail = 0;
    __device__ __forceinline__ void put(const DevCfg& c, int var, float x) { check_var(c, fail, var, x); }
    __device__ __forceinline__ void next() {}
    __device__ __forceinline__ void last() {}
    __device__ __forceinline__ int result(const DevCfg&) const { return fail; }
};
#define FWG_CHK_POINTS 5           /* 4 stages + the end of the step */
#define FWG_CHK_VARS 15            /* variables 0 .. FWG_V_BETA can carry a constraint */
struct DeferredChecks {
    float v[FWG_CHK_POINTS][FWG_CHK_VARS];
    int cp = 0;
    __device__ __forceinline__ void put(const DevCfg& c, int var, float x) {
        if ((c.con_mask & (1u << var)) && var < FWG_CHK_VARS) {
#pragma unroll
            for (int k = 0; k < FWG_CHK_POINTS; ++k) if (k == cp) v[k][var] = x;   // (cp is a compile-time constant after unrolling)
        }
    }
    __device__ __forceinline__ void next() { ++cp; }
    __device__ __forceinline__ void last() { cp = FWG_CHK_POINTS - 1; }   // the end-of-step checks
    // n_pts = check points recorded for variable groups that exist only at the end of the step (roll, pitch, yaw): they are
    // put() at the last point only, the earlier slots of those variables are never read
    __device__ __forceinline__ int result(const DevCfg& c) const {
        bool any = false;
#pragma unroll
        for (int var = 0; var < FWG_CHK_VARS; ++var) {
            if (!(c.con_mask & (1u << var))) continue;
            const int first = var < FWG_V_OMEGA_P ? FWG_CHK_POINTS - 1 : 0;   // Euler angles: end of the step only
            if (V(c).con_min[var] == -V(c).con_max[var]) {
                float m = fabsf(v[first][var]);
#pragma unroll
                for (int k = first + 1; k < FWG_CHK_POINTS; ++k) m = fmaxf(m, fabsf(v[k][var]));
                any = any || (m > V(c).con_max[var]);
            } else {
                float hi = v[first][var], lo = v[first][var];
#pragma unroll
                for (int k = first + 1; k < FWG_CHK_POINTS; ++k) { hi = fmaxf(hi, v[k][var]); lo = fminf(lo, v[k][var]); }
                any = any || (hi > V(c).con_max[var]) || (lo < V(c).con_min[var]);
            }
        }
        int fail = 0;
        if (__ballot(any) != 0ull) {   // (rare) some lane of the wave tripped: name it in the reference order
#pragma unroll
            for (int k = 0; k < FWG_CHK_POINTS; ++k) {
#pragma unroll
                for (int var = FWG_V_OMEGA_P; var < FWG_V_OMEGA_P + 9; ++var) check_var(c, fail, var, v[k][var]);
                if (k == FWG_CHK_POINTS - 1) {
                    check_var(c, fail, FWG_V_ROLL, v[k][FWG_V_ROLL]); check_var(c, fail, FWG_V_PITCH, v[k][FWG_V_PITCH]);
                    check_var(c, fail, FWG_V_YAW, v[k][FWG_V_YAW]);
                }
                check_var(c, fail, FWG_V_VA, v[k][FWG_V_VA]); check_var(c, fail, FWG_V_ALPHA, v[k][FWG_V_ALPHA]);
                check_var(c, fail, FWG_V_BETA, v[k][FWG_V_BETA]);
            }
        }
        return fail;
    }
};

#define NB 13  // rigid-body part of the state vector (quaternion, omega, position, body velocity)

// d/dt of the 13 rigid-body states for the actuator deflections act = (elevon_right, elevon_left, throttle) at the
// stage time (oracle/physics.py rhs)
// `a` = the force / moment constants: the DevCfg itself (one aircraft for all envs; compile-time constants in the specialised
// kernels) or this lane's own set (Aero, simulator.model)
// CK: what becomes of the values the constraints are checked on at this evaluation -- checked on the spot (ImmediateChecks:
// the sticky failure code, first violation wins) or recorded for one ordered evaluation after the last stage (DeferredChecks).
// ACT: where the actuator deflections of the stage come from -- act(er, el, th) is called as LATE as the arithmetic allows:
// everything that does not depend on the deflections (rotation, airspeed and incidence, the stall blend, the coefficient
// parts in alpha / beta / body rates, kinematics, gravity) comes first, ~200 of the ~300 instructions of an evaluation.  In
// k_step2 the deflections at t + h/2 and t + h are a message from the partner wave (PartnerActuators): the second stage asks for
// it here, a thousand ticks after the first stage ended, instead of waiting between the stages.
// (two functions with plain values between them -- rhs_free: everything that does not depend on the deflections; rhs_act: the
// rest.  A functor handed into one function kept the caller's actuator arrays in scratch memory.)
struct RhsFree { float pre, Va, lift0, drag0, m0, fy0, l0, n0, ca, sa, sb, cb, gx, gy, gz, w4, w5, w6, c10, c11, c12; };
template <bool TURB, class AP, class CK>
__device__ __forceinline__ RhsFree rhs_free(const DevCfg& c, const AP& a_, const float (&y)[NB], const float (&wind)[3],
                                            const float (&gust)[6], float (&dy)[NB], CK& ck) {
    const float e0 = y[0], e1 = y[1], e2 = y[2], e3 = y[3];
    const float p = y[4], q = y[5], r = y[6];
    const float u = y[10], v = y[11], w = y[12];
    if (c.con_mask & 0xFF8u) {  // omega_p .. velocity_w
#pragma unroll
        for (int k = 0; k < 9; ++k) ck.put(c, FWG_V_OMEGA_P + k, y[4 + k]);
    }

    const Rot R = rot_from_quat(e0, e1, e2, e3);
    const Air a = airspeed<TURB>(R, u, v, w, wind, gust);
    if (c.con_mask & 0x7000u) {
        ck.put(c, FWG_V_VA, a.Va);
        ck.put(c, FWG_V_ALPHA, a.alpha);
        ck.put(c, FWG_V_BETA, a.beta);
    }
    ck.next();
    const float Va = fclampf(a.Va, V(c).val_min[FWG_V_VA], V(c).val_max[FWG_V_VA]);
    float pa = p, qa = q, ra = r;
    if (TURB) { pa -= gust[3]; qa -= gust[4]; ra -= gust[5]; }

    // direction cosines of the airspeed vector (no trigonometry needed)
    const float v2 = fmaxf(a.ua * a.ua + a.va * a.va + a.wa * a.wa, 1e-30f);
    const float xz2 = fmaxf(a.ua * a.ua + a.wa * a.wa, 1e-30f);
    const float rv = frsq(v2), rxz = frsq(xz2);
    const float ca = a.ua * rxz, sa = a.wa * rxz, sb = a.va * rv, cb = xz2 * rxz * rv;

    const float pre = a_.half_rho_S * Va * Va;
    const float i2v = 0.5f * frcp(Va);
    // stall blend: 1 - sigma = 1/((1+exp(M(a-a0)))(1+exp(-M(a+a0))))  (overflow-safe form of the published sigma)
    const float ex1 = __expf(a_.M * a.alpha - a_.Ma0), ex2 = __expf(-a_.M * a.alpha - a_.Ma0);
    const float oms = frcp((1.f + ex1) * (1.f + ex2));
    const float sig = 1.f - oms;
    const float sgn = fsignf(a.alpha);
    const float sa2 = sa * sa;
    const float CLlin = a_.CL0 + a_.CLa * a.alpha;
    const float CL = oms * CLlin + sig * (2.f * sgn * sa2 * ca);
    const float lift0 = CL + a_.cLq * i2v * qa;
    const float CD = a_.CDp + oms * CLlin * CLlin * a_.kInd + sig * (2.f * sgn * sa2 * sa);
    const float CDb = (a_.CDb1 + a_.CDb2 * a.beta) * a.beta;
    const float drag0 = CD + CDb + a_.cDq * i2v * qa;
    const float Cm = oms * (a_.Cm0 + a_.Cma * a.alpha) + sig * (a_.Cmfp * sgn * sa2);
    const float m0 = Cm + a_.cmq * i2v * qa;
    const float fy0 = a_.CY0 + a_.CYb * a.beta + i2v * (a_.cYp * pa + a_.cYr * ra);
    const float l0 = a_.Cl0 + a_.Clb * a.beta + i2v * (a_.clp * pa + a_.clr * ra);
    const float n0 = a_.Cn0 + a_.Cnb * a.beta + i2v * (a_.cnp * pa + a_.cnr * ra);
    // kinematics, gravity and the gyroscopic terms
    dy[0] = 0.5f * (-p * e1 - q * e2 - r * e3);
    dy[1] = 0.5f * (p * e0 + r * e2 - q * e3);
    dy[2] = 0.5f * (q * e0 - r * e1 + p * e3);
    dy[3] = 0.5f * (r * e0 + q * e1 - p * e2);
    dy[7] = R.r00 * u + R.r01 * v + R.r02 * w;
    dy[8] = R.r10 * u + R.r11 * v + R.r12 * w;
    dy[9] = R.r20 * u + R.r21 * v + R.r22 * w;
    const float gx = a_.mg * 2.f * (e1 * e3 - e2 * e0), gy = a_.mg * 2.f * (e2 * e3 + e1 * e0);
    const float gz = a_.mg * (e3 * e3 + e0 * e0 - e1 * e1 - e2 * e2);
    const float w4 = a_.G1 * p * q - a_.G2 * q * r, w5 = a_.G5 * p * r - a_.G6 * (p * p - r * r), w6 = a_.G7 * p * q - a_.G1 * q * r;
    const float c10 = r * v - q * w, c11 = p * w - r * u, c12 = q * u - p * v;
    return RhsFree{pre, Va, lift0, drag0, m0, fy0, l0, n0, ca, sa, sb, cb, gx, gy, gz, w4, w5, w6, c10, c11, c12};
}
template <class AP>
__device__ __forceinline__ void rhs_act(const AP& a_, const RhsFree& f, float er, float el, float th, float (&dy)[NB]) {
    const float pre = f.pre, Va = f.Va, lift0 = f.lift0, drag0 = f.drag0, m0 = f.m0, fy0 = f.fy0, l0 = f.l0, n0 = f.n0;
    const float ca = f.ca, sa = f.sa, sb = f.sb, cb = f.cb, gx = f.gx, gy = f.gy, gz = f.gz;
    const float w4 = f.w4, w5 = f.w5, w6 = f.w6, c10 = f.c10, c11 = f.c11, c12 = f.c12;
    const float elev = 0.5f * (er + el), ail = 0.5f * (el - er);
    const float lift = pre * (lift0 + a_.CLde * elev);
    const float drag = pre * (drag0 + a_.CDde * elev * elev);
    const float m_ = pre * a_.chord * (m0 + a_.Cmde * elev);
    const float fy_s = pre * (fy0 + a_.CYda * ail);
    const float l_ = pre * a_.span * (l0 + a_.Clda * ail) - a_.ktp * th * th;
    const float n_ = pre * a_.span * (n0 + a_.Cnda * ail);

    // wind axes -> body axes
    const float fxa = -ca * cb * drag - ca * sb * fy_s + sa * lift;
    const float fya = -sb * drag + cb * fy_s;
    const float fza = -sa * cb * drag - sa * sb * fy_s - ca * lift;
    const float Vd = Va + th * (a_.kmotor - Va);
    const float fprop = a_.kprop * Vd * (Vd - Va);
    const float fx = fprop + gx + fxa;
    const float fy = gy + fya;
    const float fz = gz + fza;

    dy[4] = w4 + a_.G3 * l_ + a_.G4 * n_;
    dy[5] = w5 + m_ * a_.inv_Jy;
    dy[6] = w6 + a_.G4 * l_ + a_.G8 * n_;
    dy[10] = c10 + fx * a_.inv_mass;
    dy[11] = c11 + fy * a_.inv_mass;
    dy[12] = c12 + fz * a_.inv_mass;
}

// one actuator micro-step with the command held: exact linear response (2x2 transition per elevon, exponential for
// the throttle), then the rate limit -- on the rate and on the travel over the micro-step -- and the value limits
// (oracle/physics.py advance_actuators).  a = (elevon_r, elevon_l, throttle, elevon_r_rate, elevon_l_rate)
__device__ __forceinline__ void advance_actuators(const DevCfg& c, float (&a)[5], const float (&sp)[3]) {
#if !defined(FWG_EMU) && !defined(FWG_SCALAR_ACTUATORS)
    // the two elevons side by side: the 2x2 transitions and the travel window as packed f32 operations
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 spv = {sp[0], sp[1]}, val = {a[0], a[1]}, rate = {a[3], a[4]};
    const f2 x0 = val - spv;
    const f2 p0 = {V(c).act_phi[0][0], V(c).act_phi[1][0]}, p1 = {V(c).act_phi[0][1], V(c).act_phi[1][1]};
    const f2 p2 = {V(c).act_phi[0][2], V(c).act_phi[1][2]}, p3 = {V(c).act_phi[0][3], V(c).act_phi[1][3]};
    const f2 v = spv + p0 * x0 + p1 * rate;
    const f2 d = p2 * x0 + p3 * rate;
    const f2 trav = {V(c).act_travel[0], V(c).act_travel[1]};
    const f2 lo = val - trav, hi = val + trav;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        a[3 + k] = fclampf(d[k], -V(c).dot_max[k], V(c).dot_max[k]);
        a[k] = fclampf(fclampf(v[k], lo[k], hi[k]), V(c).val_min[FWG_V_ELEVON_RIGHT + k], V(c).val_max[FWG_V_ELEVON_RIGHT + k]);
    }
#else
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float x0 = a[k] - sp[k], x1 = a[3 + k];
        float v = sp[k] + V(c).act_phi[k][0] * x0 + V(c).act_phi[k][1] * x1;
        float d = V(c).act_phi[k][2] * x0 + V(c).act_phi[k][3] * x1;
        d = fclampf(d, -V(c).dot_max[k], V(c).dot_max[k]);
        v = fclampf(v, a[k] - V(c).act_travel[k], a[k] + V(c).act_travel[k]);
        a[k] = fclampf(v, V(c).val_min[FWG_V_ELEVON_RIGHT + k], V(c).val_max[FWG_V_ELEVON_RIGHT + k]);
        a[3 + k] = d;
    }
#endif
    a[2] = fclampf(sp[2] + V(c).act_ethr * (a[2] - sp[2]), V(c).val_min[FWG_V_THROTTLE], V(c).val_max[FWG_V_THROTTLE]);
}

__device__ __forceinline__ void sanitize_actuators(const DevCfg& c, float (&a)[5]) {
    a[0] = fclampf(a[0], V(c).val_min[FWG_V_ELEVON_RIGHT], V(c).val_max[FWG_V_ELEVON_RIGHT]);
    a[1] = fclampf(a[1], V(c).val_min[FWG_V_ELEVON_LEFT], V(c).val_max[FWG_V_ELEVON_LEFT]);
    a[2] = fclampf(a[2], V(c).val_min[FWG_V_THROTTLE], V(c).val_max[FWG_V_THROTTLE]);
    a[3] = fclampf(a[3], -V(c).dot_max[0], V(c).dot_max[0]);
    a[4] = fclampf(a[4], -V(c).dot_max[1], V(c).dot_max[1]);
}

// elevator/aileron/throttle commands -> constrained inputs (the "command" history of the reference,
// fixed_wing.py:828,1110) and the elevon/throttle set-points of the actuator dynamics
__device__ __forceinline__ void constrain_commands(const DevCfg& c, const float (&cmd)[3], float (&cmd_c)[3], float (&sp)[3]) {
    const float e = fclampf(cmd[0], V(c).val_min[FWG_V_ELEVATOR], V(c).val_max[FWG_V_ELEVATOR]);
    const float a = fclampf(cmd[1], V(c).val_min[FWG_V_AILERON], V(c).val_max[FWG_V_AILERON]);
    const float t = fclampf(cmd[2], V(c).val_min[FWG_V_THROTTLE], V(c).val_max[FWG_V_THROTTLE]);
    const float er = fclampf(e - a, V(c).val_min[FWG_V_ELEVON_RIGHT], V(c).val_max[FWG_V_ELEVON_RIGHT]);
    const float el = fclampf(e + a, V(c).val_min[FWG_V_ELEVON_LEFT], V(c).val_max[FWG_V_ELEVON_LEFT]);
    cmd_c[0] = 0.5f * (er + el); cmd_c[1] = 0.5f * (el - er); cmd_c[2] = t;
    sp[0] = er; sp[1] = el; sp[2] = t;
}

// the actuator states at t + h/2 and t + h for one RK4 step per env step (what sim_step computes itself otherwise)
__device__ __forceinline__ void actuators_over_step(const DevCfg& c, const float (&a0)[5], const float (&sp)[3], float (&a_half)[5],
                                                    float (&a_full)[5]) {
    float a[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = a0[i];
    sanitize_actuators(c, a);
#pragma unroll
    for (int i = 0; i < 5; ++i) a_half[i] = a[i];
    _Pragma("unroll") for (int m = 0; m < c.act_per_half; ++m) advance_actuators(c, a_half, sp);
#pragma unroll
    for (int i = 0; i < 5; ++i) a_full[i] = a_half[i];
    _Pragma("unroll") for (int m = 0; m < c.act_per_half; ++m) advance_actuators(c, a_full, sp);
}

struct Derived { float roll, pitch, yaw, Va, alpha, beta; };

template <bool TURB>
__device__ __forceinline__ Derived derive(const float (&y)[NY], const float (&wind)[3], const float (&gust)[6]);

// RK4 stage loop: unrolled by two where the checks are immediate (measured -0.5...-0.7 us per step on every workload against
// the rolled loop; the stage selectors fold), fully where they are deferred (every stage keeps its checked values in registers
// of its own)
#ifndef FWG_MICRO_UNROLL
#define FWG_MICRO_PRAGMA "unroll"
#elif FWG_MICRO_UNROLL == 1
#define FWG_MICRO_PRAGMA "unroll 1"
#elif FWG_MICRO_UNROLL == 2
#define FWG_MICRO_PRAGMA "unroll 2"
#else
#define FWG_MICRO_PRAGMA "unroll 4"
#endif
#ifndef FWG_STAGE_UNROLL
#define FWG_STAGE_UNROLL 2
#endif

// the arguments of the three Euler-angle arctangents (derive): roll = atan2(e[0], e[1]), pitch = asin(e[2]), yaw = atan2(e[3], e[4])
__device__ __forceinline__ void euler_args(const float (&y)[NY], float (&e)[5]) {
    const float e0 = y[0], e1 = y[1], e2 = y[2], e3 = y[3];
    e[0] = 2.f * (e0 * e1 + e2 * e3); e[1] = e0 * e0 + e3 * e3 - e1 * e1 - e2 * e2;
    e[2] = fclampf(2.f * (e0 * e2 - e1 * e3), -1.f, 1.f);
    e[3] = 2.f * (e0 * e3 + e1 * e2); e[4] = e0 * e0 + e1 * e1 - e2 * e2 - e3 * e3;
}
__device__ __forceinline__ void euler_from_args(const float (&e)[5], Derived& d) {
    d.roll = fast_atan2(e[0], e[1]);
    d.pitch = fast_atan2(e[2], __builtin_amdgcn_sqrtf(fmaxf(1.f - e[2] * e[2], 0.f)));  // = asin
    d.yaw = fast_atan2(e[3], e[4]);
}

template <bool TURB>
__device__ __forceinline__ Derived derive(const float (&y)[NY], const float (&wind)[3], const float (&gust)[6]) {
    Derived d;
    float ea[5];
    euler_args(y, ea);
    euler_from_args(ea, d);
    const Rot R = rot_from_quat(y[0], y[1], y[2], y[3]);
    const Air a = airspeed<TURB>(R, y[10], y[11], y[12], wind, gust);
    d.Va = a.Va; d.alpha = a.alpha; d.beta = a.beta;
    return d;
}

// One env step (dt) -- the scheme of oracle/physics.py sim_step: actuators advanced exactly over c.act_micro
// micro-steps, rigid body by c.nsub classical RK4 steps whose stages see the actuator deflections at t, t+h/2,
// t+h/2, t+h.  On success y holds the new state, otherwise y is untouched.  Returns the failure code (0 = ok,
// var+1 = violated constraint, FWG_TERM_NAN - FWG_TERM_VAR0 + 1 = non-finite).
// `ext` (k_step2): the actuator deflections at t + h/2 and t + h come from the partner wave, which advances the actuators
// while this wave evaluates the first stage (they depend on the commands and the actuator states only); fetch() waits for
// them.  Only for one RK4 step per env step (c.nsub == 1).
struct NoExtActuators {
    static constexpr bool enabled = false;
    const float* unused = nullptr;
    __device__ __forceinline__ void fetch_half(float (&)[5]) const {}
    __device__ __forceinline__ void fetch_full(float (&)[5]) const {}
};
// `hook(st)` runs after stage st of the first sub-step: a place for work that only needs to be STARTED while the integration
// runs (k_step2: cache prefetches for the tail work, decided from a word its partner wave has written by then)
struct NoStageHook {
    const void* unused = nullptr;
    template <class T> __device__ __forceinline__ NoStageHook(const T&) {}
    __device__ __forceinline__ NoStageHook() {}
    __device__ __forceinline__ void operator()(int) const {}
};
// `hand` (k_step2): the new state leaves for the partner wave AS SOON AS it exists -- state(): the candidate state and the
// arguments of its Euler-angle arctangents, right after the RK4 update and the quaternion normalisation (the partner turns them
// into roll / pitch / yaw while this wave computes airspeed, angle of attack and sideslip and evaluates the checks) -- and
// result(): Va, alpha, beta and the failure code.  With euler_here() false the Euler angles are NOT computed here (d.roll /
// pitch / yaw stay 0): only when no constraint is set on them.
struct NoHandOff {
    static constexpr bool enabled = false;
    __device__ __forceinline__ void state(const float (&)[NY], const float (&)[5]) const {}
    __device__ __forceinline__ void result(float, float, float, int) const {}
};
template <bool TURB, class EXT = NoExtActuators, class AP = DevCfg, class HOOK = NoStageHook, class HAND = NoHandOff, bool DEFER = false>
__device__ __forceinline__ int sim_step(const DevCfg& c, const AP& aero, float (&y)[NY], const float (&sp)[3], const float (&wind)[3],
                                        const float (&gust)[6], Derived& d, const EXT& ext = EXT(), const HOOK& hook = HOOK(),
                                        const HAND& hand = HAND()) {
    const bool use_ext = EXT::enabled && c.nsub == 1;
    float yb[NB], a[5];
#pragma unroll
    for (int i = 0; i < NB; ++i) yb[i] = y[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = y[NB + i];
    sanitize_actuators(c, a);
    typename std::conditional<DEFER, DeferredChecks, ImmediateChecks>::type ck;
    for (int s = 0; s < (DEFER ? 1 : c.nsub); ++s) {
        float a_half[5], a_full[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) a_half[i] = a[i];
        if (!use_ext) {
#ifdef FWG_ABL_ACT1
            advance_actuators(c, a_half, sp);
#else
            _Pragma(FWG_MICRO_PRAGMA) for (int m = 0; m < c.act_per_half; ++m) advance_actuators(c, a_half, sp);
#endif
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) a_full[i] = a_half[i];
        if (!use_ext) {
#ifdef FWG_ABL_ACT1
            advance_actuators(c, a_full, sp);
#else
            _Pragma(FWG_MICRO_PRAGMA) for (int m = 0; m < c.act_per_half; ++m) advance_actuators(c, a_full, sp);
#endif
        }
        float acc[NB], ys[NB], k[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) { acc[i] = 0.f; ys[i] = yb[i]; }
        auto stage = [&](int st) {
            // (a team's second / fourth stage: the partner's message is taken where the evaluation first needs the deflections)
            RhsFree f = rhs_free<TURB>(c, aero, ys, wind, gust, k, ck);
            if (use_ext && (st == 1 || st == 3)) {   // (everything the evaluation can do without the deflections is done before the wait)
                fwg_pin(f.pre, f.Va, f.lift0, f.drag0, f.m0, f.fy0, f.l0, f.n0, f.ca, f.sa, f.sb, f.cb, f.gx, f.gy, f.gz, f.w4, f.w5, f.w6, f.c10, f.c11, f.c12);
                fwg_pin(k[0], k[1], k[2], k[3], k[7], k[8], k[9]);
                if (st == 1) ext.fetch_half(a_half); else ext.fetch_full(a_full);
            }
            const float er = (st == 0) ? a[0] : ((st == 3) ? a_full[0] : a_half[0]);
            const float el = (st == 0) ? a[1] : ((st == 3) ? a_full[1] : a_half[1]);
            const float th = (st == 0) ? a[2] : ((st == 3) ? a_full[2] : a_half[2]);
            rhs_act(aero, f, er, el, th, k);
            if (s == 0) hook(st);
            const float bw = (st == 0 || st == 3) ? V(c).h_sixth : 2.f * V(c).h_sixth;
            const float aw = (st == 2) ? V(c).h : V(c).half_h;
#pragma unroll
            for (int i = 0; i + 1 < NB; i += 2) {   // the update on pairs (NB is odd: the last state alone)
                const f2 kk = mk2(k[i], k[i + 1]);
                const f2 an = mk2(acc[i], acc[i + 1]) + splat2(bw) * kk, yn = mk2(yb[i], yb[i + 1]) + splat2(aw) * kk;
                acc[i] = an[0]; acc[i + 1] = an[1]; ys[i] = yn[0]; ys[i + 1] = yn[1];
            }
            acc[NB - 1] += bw * k[NB - 1]; ys[NB - 1] = yb[NB - 1] + aw * k[NB - 1];
        };
        if (DEFER || HAND::enabled) {   // (the team's kernels: straight-line stages -- the rolled loop selected the stage's actuator
            // state through a scratch-memory pointer)
#ifdef FWG_ABL_RK1
            stage(0);
#else
            stage(0); stage(1); stage(2); stage(3);
#endif
        } else {
#pragma unroll FWG_STAGE_UNROLL
#ifdef FWG_ABL_RK1
            for (int st = 0; st < 1; ++st) stage(st);
#else
            for (int st = 0; st < 4; ++st) stage(st);
#endif
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) yb[i] += acc[i];
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = a_full[i];
    }
    float yy[NY];
    const float rn = frsq(yb[0] * yb[0] + yb[1] * yb[1] + yb[2] * yb[2] + yb[3] * yb[3]);
#pragma unroll
    for (int i = 0; i < NB; ++i) yy[i] = (i < 4) ? yb[i] * rn : yb[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) yy[NB + i] = a[i];
    float ea[5];
    euler_args(yy, ea);
    hand.state(yy, ea);
    ck.last();
    if (c.con_mask & 0xFF8u) {
#pragma unroll
        for (int k = 0; k < 9; ++k) ck.put(c, FWG_V_OMEGA_P + k, yy[4 + k]);
    }
    Derived dn;
    dn.roll = 0.f; dn.pitch = 0.f; dn.yaw = 0.f;
    const bool euler_here = !HAND::enabled || (c.con_mask & 0x7u) != 0u;
    if (euler_here) euler_from_args(ea, dn);
    {
        const Rot R = rot_from_quat(yy[0], yy[1], yy[2], yy[3]);
        const Air a_ = airspeed<TURB>(R, yy[10], yy[11], yy[12], wind, gust);
        dn.Va = a_.Va; dn.alpha = a_.alpha; dn.beta = a_.beta;
    }
    if (c.con_mask & 0x7007u) {
        if (c.con_mask & 0x7u) { ck.put(c, FWG_V_ROLL, dn.roll); ck.put(c, FWG_V_PITCH, dn.pitch); ck.put(c, FWG_V_YAW, dn.yaw); }
        ck.put(c, FWG_V_VA, dn.Va); ck.put(c, FWG_V_ALPHA, dn.alpha); ck.put(c, FWG_V_BETA, dn.beta);
    }
    int fail = ck.result(c);
    // non-finite state: x * 0 is 0 for every finite x and NaN otherwise (pairs: one packed multiply-add per two states)
    f2 z = mk2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NY; i += 2) z = mk2(yy[i], yy[i + 1]) * splat2(0.f) + z;
    const bool finite = (z[0] + z[1]) == 0.f;
    if (fail == 0 && !finite) fail = FWG_TERM_NAN - FWG_TERM_VAR0 + 1;
    hand.result(dn.Va, dn.alpha, dn.beta, fail);
    if (fail == 0) {
#pragma unroll
        for (int i = 0; i < NY; ++i) y[i] = yy[i];
        d = dn;
    }
    return fail;
}

// Dryden: gust sample of the current step = C x ; advance x' = A x + B n with 4 standard normals.  The joint filter
// is block structured -- states u(0) | v,r(1..3) | w,q(4..6) | p(7), one noise channel per block -- so only the
// non-zero blocks of the dense matrices are evaluated (config.py dryden_matrices / oracle physics.dryden_continuous).
__device__ __forceinline__ void dryden_output(const DevCfg& c, const float (&x)[FWG_N_DRYDEN], float (&gust)[6]) {
    const float* C = V(c).dryC;
    gust[0] = C[0 * 8 + 0] * x[0];
    gust[1] = C[1 * 8 + 1] * x[1] + C[1 * 8 + 2] * x[2] + C[1 * 8 + 3] * x[3];
    gust[2] = C[2 * 8 + 4] * x[4] + C[2 * 8 + 5] * x[5] + C[2 * 8 + 6] * x[6];
    gust[3] = C[3 * 8 + 7] * x[7];
    gust[4] = C[4 * 8 + 4] * x[4] + C[4 * 8 + 5] * x[5] + C[4 * 8 + 6] * x[6];
    gust[5] = C[5 * 8 + 1] * x[1] + C[5 * 8 + 2] * x[2] + C[5 * 8 + 3] * x[3];
}
__device__ __forceinline__ void dryden_advance(const DevCfg& c, float (&x)[FWG_N_DRYDEN], const float (&n)[4]) {
    const float* A = V(c).dryA;
    const float* B = V(c).dryB;
    float xn[FWG_N_DRYDEN];
    xn[0] = A[0] * x[0] + B[0 * 4 + 0] * n[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        xn[i] = A[i * 8 + 1] * x[1] + A[i * 8 + 2] * x[2] + A[i * 8 + 3] * x[3] + B[i * 4 + 1] * n[1];
#pragma unroll
    for (int i = 4; i < 7; ++i)
        xn[i] = A[i * 8 + 4] * x[4] + A[i * 8 + 5] * x[5] + A[i * 8 + 6] * x[6] + B[i * 4 + 2] * n[2];
    xn[7] = A[7 * 8 + 7] * x[7] + B[7 * 4 + 3] * n[3];
#pragma unroll
    for (int i = 0; i < FWG_N_DRYDEN; ++i) x[i] = xn[i];
}
// the gust sample of the NEXT env step after the filter advanced from x_old to x_new (oracle/physics.py
// dryden_gust_after_advance): the first difference of the outputs (increment turbulence) or the outputs themselves
__device__ __forceinline__ void dryden_next_gust(const DevCfg& c, const float (&x_old)[FWG_N_DRYDEN], const float (&x_new)[FWG_N_DRYDEN],
                                                 float (&gust)[6]) {
    float d[FWG_N_DRYDEN];
#pragma unroll
    for (int i = 0; i < FWG_N_DRYDEN; ++i) d[i] = c.turb_increment ? x_new[i] - x_old[i] : x_new[i];
    dryden_output(c, d, gust);
}
// Box-Muller with the hardware transcendental units: v_log_f32, v_sqrt_f32 and v_sin/v_cos_f32 (which take their
// argument in revolutions, so sin(2 pi u) is a single instruction)
__device__ __forceinline__ void box_muller(const u4& b, float (&n)[4]) {
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(b.x)));  // -2 ln u = -2 ln2 log2 u
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(b.z)));
    const float t0 = u01(b.y), t1 = u01(b.w);
    n[0] = r0 * __builtin_amdgcn_cosf(t0); n[1] = r0 * __builtin_amdgcn_sinf(t0);
    n[2] = r1 * __builtin_amdgcn_cosf(t1); n[3] = r1 * __builtin_amdgcn_sinf(t1);
}
