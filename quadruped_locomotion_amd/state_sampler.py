"""Synthetic state and command samplers for the BASELINE configurations: input
GENERATORS of the benchmark and the tests -- states (q, u), base / leg commands and contact patterns, no
dynamics.  The dynamics terms of a state come from the device producer (hoqp_wbc_dynamics_batched) on the
product path; the numpy model that CHECKS that producer lives with the test infrastructure, outside this package.

Conventions follow the reference: q = [quat wxyz (4), base position (3), joints (12)] (anymal_plugin.cpp:266-272),
u = [omega_W (3), v_W (3), joint velocities (12)] (:285-290), joint j = kind * 4 + leg (anymal_constants.hpp:28-52).
"""
from __future__ import annotations

import numpy as np

# kInitialJointConfig, anymal_constants.hpp:48-52
INITIAL_JOINT_CONFIG = np.array(
    [0, 0, 0, 0, 0.5, 0.5, -0.5, -0.5, -0.9, -0.9, 0.9, 0.9], dtype=np.float64
)


def _euler_to_quat(roll, pitch, yaw):
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    return np.stack(
        [cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy,
         cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy], axis=-1)


def standing_state():
    """C1: identity orientation, height 0.5, kInitialJointConfig, u = 0, zero commands, mask 1111."""
    q = np.zeros((1, 19))
    q[0, 0] = 1.0
    q[0, 6] = 0.5
    q[0, 7:] = INITIAL_JOINT_CONFIG
    u = np.zeros((1, 18))
    return dict(q=q, u=u, base_cmd=np.zeros((1, 9)), leg_cmd=np.zeros((1, 36)),
                contact=np.ones((1, 4), dtype=np.int32))


def sample_states(n, seed, contact="stance"):
    """Randomised states: C2 'stance' (mask 1111), C3 'trot' (25% LF+RH, 25% RF+LH, 50% all),
    C5 'three' (one leg swinging, uniform)."""
    rng = np.random.default_rng(seed)
    q = np.zeros((n, 19))
    q[:, 0:4] = _euler_to_quat(rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n),
                               rng.uniform(-np.pi, np.pi, n))
    q[:, 4:6] = 0.0
    q[:, 6] = rng.uniform(0.4, 0.6, n)
    q[:, 7:] = INITIAL_JOINT_CONFIG + rng.uniform(-0.3, 0.3, (n, 12))
    u = rng.normal(0.0, 0.3, (n, 18))
    base_cmd = rng.normal(0.0, 0.5, (n, 9))
    contact_arr = np.ones((n, 4), dtype=np.int32)
    if contact == "trot":
        r = rng.uniform(size=n)
        contact_arr[r < 0.25] = np.array([1, 0, 0, 1], dtype=np.int32)
        contact_arr[(r >= 0.25) & (r < 0.5)] = np.array([0, 1, 1, 0], dtype=np.int32)
    elif contact == "three":
        sw = rng.integers(0, 4, n)
        contact_arr[np.arange(n), sw] = 0
    elif contact != "stance":
        raise ValueError(contact)
    leg_cmd = np.zeros((n, 36))
    leg_cmd[:, 0:12] = rng.uniform(-0.05, 0.05, (n, 12))  # offset from current foot pos, fixed up below
    leg_cmd[:, 12:36] = rng.normal(0.0, 0.5, (n, 24))
    return dict(q=q, u=u, base_cmd=base_cmd, leg_cmd=leg_cmd, contact=contact_arr)
