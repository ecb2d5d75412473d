"""TEST INFRASTRUCTURE (with the rest of oracle/): the numpy "ANYmal-C-like" quadruped model.

It is the CHECKER of the device dynamics-terms producer (hoqp_wbc_dynamics_batched) and the input
generator of the tests; the product package only keeps the state / command samplers
(quadruped_locomotion_amd/state_sampler.py), which are re-exported here.

Synthetic "ANYmal-C-like" quadruped: batched dynamics terms for the HO-QP solve.

The reference pulls M, h, J_c, J_c_dot*u, J_b and foot positions from a Drake
MultibodyPlant built from Drake's ANYmal-C URDF (``src/dynamics/dynamics.cpp:25-29,
75-228``).  Neither Drake nor that URDF exists in this image, so this module is the
build's OWN, documented, approximate model (parity unpinned for the model itself;
the solve is checked on identical inputs, so model fidelity does not enter parity).

Conventions follow the reference / Drake:
  q = [quat wxyz (4), base position (3), joints (12)]        (anymal_plugin.cpp:266-272)
  u = [omega_W (3), v_W (3), joint velocities (12)]          (anymal_plugin.cpp:285-290)
  joint j = kind * 4 + leg, kind in (HAA, HFE, KFE), leg in (LF, RF, LH, RH)
                                                           (anymal_constants.hpp:28-52)
  M u_dot + h = S^T tau + J_c^T lambda

Model numbers (approximate, chosen here -- total mass 50 kg):
  base   26 kg, box inertia diag(0.30, 1.20, 1.30) kg m^2
  hip     2 kg, thigh 3 kg, shank 1 kg (incl. foot), small box inertias
  HAA at (+-0.30, +-0.10, 0) in base; HFE at HAA + (0, +-0.07, 0);
  KFE at HFE + (0, +-0.03, -0.30); foot at KFE + (0, 0, -0.32).

`include_gravity=True` (default) adds the gravity term to h.  NOTE: the reference's
``GetBiasVector`` uses Drake's ``CalcBiasTerm`` (dynamics.cpp:82-87), which is C(q,v)v
WITHOUT gravity; pass ``include_gravity=False`` to mimic that.
"""
from __future__ import annotations

import numpy as np

from quadruped_locomotion_amd.state_sampler import INITIAL_JOINT_CONFIG, sample_states, standing_state  # noqa: F401

NU = 18
NJ = 12
GRAV = 9.81

_SX = np.array([+1.0, +1.0, -1.0, -1.0])  # LF RF LH RH : front/hind
_SY = np.array([+1.0, -1.0, +1.0, -1.0])  # left/right

BASE_MASS = 26.0
BASE_INERTIA = np.diag([0.30, 1.20, 1.30])
HIP_MASS, THIGH_MASS, SHANK_MASS = 2.0, 3.0, 1.0
HIP_INERTIA = np.diag([0.004, 0.004, 0.003])
THIGH_INERTIA = np.diag([0.030, 0.030, 0.004])
SHANK_INERTIA = np.diag([0.012, 0.012, 0.001])
HAA_OFF = np.array([0.30, 0.10, 0.0])
HFE_OFF = np.array([0.0, 0.07, 0.0])
KFE_OFF = np.array([0.0, 0.03, -0.30])
FOOT_OFF = np.array([0.0, 0.0, -0.32])
HIP_COM = np.array([0.0, 0.04, 0.0])
THIGH_COM = np.array([0.0, 0.02, -0.12])
SHANK_COM = np.array([0.0, 0.0, -0.14])



def _quat_to_rot(qw):
    w, x, y, z = (qw[..., i] for i in range(4))
    R = np.empty(qw.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - z * w)
    R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w)
    R[..., 2, 1] = 2 * (y * z + x * w)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _rot_x(a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3))
    R[..., 0, 0] = 1
    R[..., 1, 1] = c
    R[..., 1, 2] = -s
    R[..., 2, 1] = s
    R[..., 2, 2] = c
    return R


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3))
    R[..., 0, 0] = c
    R[..., 0, 2] = s
    R[..., 1, 1] = 1
    R[..., 2, 0] = -s
    R[..., 2, 2] = c
    return R


def _cross(a, b):
    return np.cross(a, b)


def _skew(v):
    S = np.zeros(v.shape[:-1] + (3, 3))
    S[..., 0, 1] = -v[..., 2]
    S[..., 0, 2] = v[..., 1]
    S[..., 1, 0] = v[..., 2]
    S[..., 1, 2] = -v[..., 0]
    S[..., 2, 0] = -v[..., 1]
    S[..., 2, 1] = v[..., 0]
    return S


def dynamics_terms(q, u, include_gravity=True):
    """Batched M (B,18,18), h (B,18), Jfoot (B,4,3,18), Jdotu (B,4,3), footpos (B,4,3), Jb (B,6,18).

    Plays the role of ``HoQpController::UpdateModelDynamics`` (ho_qp_controller.cpp:89-114).
    """
    q = np.atleast_2d(np.asarray(q, dtype=np.float64))
    u = np.atleast_2d(np.asarray(u, dtype=np.float64))
    B = q.shape[0]
    quat = q[:, 0:4] / np.linalg.norm(q[:, 0:4], axis=1, keepdims=True)
    Rb = _quat_to_rot(quat)
    pb = q[:, 4:7]
    qj = q[:, 7:19]
    om_b = u[:, 0:3]
    v_b = u[:, 3:6]
    qdj = u[:, 6:18]

    M = np.zeros((B, NU, NU))
    h = np.zeros((B, NU))
    Jfoot = np.zeros((B, 4, 3, NU))
    Jdotu = np.zeros((B, 4, 3))
    footpos = np.zeros((B, 4, 3))
    gvec = np.array([0.0, 0.0, -GRAV])

    def add_body(mass, I_body, R, com, Jv, Jw, omega, a_com_bias, alpha_bias):
        Iw = R @ I_body @ np.swapaxes(R, -1, -2)
        nonlocal M, h
        M += mass * np.einsum("bki,bkj->bij", Jv, Jv) + np.einsum("bki,bkl,blj->bij", Jw, Iw, Jw)
        wrench_f = mass * a_com_bias
        if include_gravity:
            wrench_f = wrench_f - mass * gvec
        Iw_om = np.einsum("bij,bj->bi", Iw, omega)
        wrench_t = np.einsum("bij,bj->bi", Iw, alpha_bias) + _cross(omega, Iw_om)
        h += np.einsum("bki,bk->bi", Jv, wrench_f) + np.einsum("bki,bk->bi", Jw, wrench_t)

    def point_jac(r, anc):
        """Translational Jacobian of world point r attached after ancestor joints `anc`
        (list of (col, axis, origin))."""
        Jv = np.zeros((B, 3, NU))
        Jv[:, :, 0:3] = -_skew(r - pb)
        Jv[:, :, 3:6] = np.eye(3)
        for col, ax, org in anc:
            Jv[:, :, col] = _cross(ax, r - org)
        return Jv

    def ang_jac(anc):
        Jw = np.zeros((B, 3, NU))
        Jw[:, :, 0:3] = np.eye(3)
        for col, ax, _ in anc:
            Jw[:, :, col] = ax
        return Jw

    # base body (COM at base origin)
    zero3 = np.zeros((B, 3))
    add_body(BASE_MASS, BASE_INERTIA, Rb, pb, point_jac(pb, []), ang_jac([]), om_b, zero3, zero3)

    for leg in range(4):
        sx, sy = _SX[leg], _SY[leg]
        sgn = np.array([sx, sy, 1.0])
        sgn_y = np.array([1.0, sy, 1.0])
        c_haa, c_hfe, c_kfe = 6 + leg, 6 + 4 + leg, 6 + 8 + leg
        q1, q2, q3 = qj[:, leg], qj[:, 4 + leg], qj[:, 8 + leg]
        d1, d2, d3 = qdj[:, leg], qdj[:, 4 + leg], qdj[:, 8 + leg]

        # --- kinematics (world frame) ---
        o1 = pb + np.einsum("bij,j->bi", Rb, HAA_OFF * sgn)
        a1 = Rb[:, :, 0]
        R1 = Rb @ _rot_x(q1)
        o2 = o1 + np.einsum("bij,j->bi", R1, HFE_OFF * sgn_y)
        a2 = R1[:, :, 1]
        R2 = R1 @ _rot_y(q2)
        o3 = o2 + np.einsum("bij,j->bi", R2, KFE_OFF * sgn_y)
        a3 = R2[:, :, 1]
        R3 = R2 @ _rot_y(q3)
        pf = o3 + np.einsum("bij,j->bi", R3, FOOT_OFF)
        c1 = o1 + np.einsum("bij,j->bi", R1, HIP_COM * sgn_y)
        c2 = o2 + np.einsum("bij,j->bi", R2, THIGH_COM * sgn_y)
        c3 = o3 + np.einsum("bij,j->bi", R3, SHANK_COM)

        # --- velocities / bias accelerations (u_dot = 0) ---
        w0 = om_b
        al0 = zero3
        acc_o1 = _cross(w0, _cross(w0, o1 - pb))  # base origin has zero bias acceleration
        w1 = w0 + a1 * d1[:, None]
        al1 = al0 + _cross(w0, a1) * d1[:, None]
        acc_o2 = acc_o1 + _cross(al1, o2 - o1) + _cross(w1, _cross(w1, o2 - o1))
        w2 = w1 + a2 * d2[:, None]
        al2 = al1 + _cross(w1, a2) * d2[:, None]
        acc_o3 = acc_o2 + _cross(al2, o3 - o2) + _cross(w2, _cross(w2, o3 - o2))
        w3 = w2 + a3 * d3[:, None]
        al3 = al2 + _cross(w2, a3) * d3[:, None]

        def pt_acc(acc_o, al, w, r, o):
            return acc_o + _cross(al, r - o) + _cross(w, _cross(w, r - o))

        anc1 = [(c_haa, a1, o1)]
        anc2 = anc1 + [(c_hfe, a2, o2)]
        anc3 = anc2 + [(c_kfe, a3, o3)]
        add_body(HIP_MASS, HIP_INERTIA, R1, c1, point_jac(c1, anc1), ang_jac(anc1), w1,
                 pt_acc(acc_o1, al1, w1, c1, o1), al1)
        add_body(THIGH_MASS, THIGH_INERTIA, R2, c2, point_jac(c2, anc2), ang_jac(anc2), w2,
                 pt_acc(acc_o2, al2, w2, c2, o2), al2)
        add_body(SHANK_MASS, SHANK_INERTIA, R3, c3, point_jac(c3, anc3), ang_jac(anc3), w3,
                 pt_acc(acc_o3, al3, w3, c3, o3), al3)

        Jfoot[:, leg] = point_jac(pf, anc3)
        Jdotu[:, leg] = pt_acc(acc_o3, al3, w3, pf, o3)
        footpos[:, leg] = pf

    M = 0.5 * (M + np.swapaxes(M, 1, 2))
    Jb = np.zeros((B, 6, NU))
    Jb[:, :, 0:6] = np.eye(6)  # spatial velocity of the base origin in W is u[0:6] itself
    return M, h, Jfoot, Jdotu, footpos, Jb


def total_mass():
    return BASE_MASS + 4 * (HIP_MASS + THIGH_MASS + SHANK_MASS)


def make_wbc_inputs(state, include_gravity=True):
    """State dict -> every array the WBC entry point takes (all float64, C-contiguous)."""
    M, h, Jfoot, Jdotu, footpos, Jb = dynamics_terms(state["q"], state["u"], include_gravity)
    leg_cmd = state["leg_cmd"].copy()
    if np.any(state["contact"] == 0):
        leg_cmd[:, 0:12] += footpos.reshape(-1, 12)  # swing target = current foot + offset
    c = np.ascontiguousarray
    return dict(M=c(M), h=c(h), Jfoot=c(Jfoot), Jdotu=c(Jdotu), footpos=c(footpos), Jb=c(Jb),
                q=c(state["q"]), u=c(state["u"]), base_cmd=c(state["base_cmd"]), leg_cmd=c(leg_cmd),
                contact=c(state["contact"].astype(np.int32)))
