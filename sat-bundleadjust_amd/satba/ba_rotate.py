"""
Rotation conventions of the bundle-adjustment path.

The camera rotation is parametrised by Euler angles (roll a, pitch b, yaw g) with
R = Rz(g) @ Ry(b) @ Rx(a)  (ref:bundle_adjust/ba_rotate.py:85-94); `ba_core.rotate_euler`
applies the same three rotations point-wise (ref:bundle_adjust/ba_core.py:36-56).
Only the Euler <-> matrix pair is on the hot path (used by ba_params); the quaternion
helpers exist for the round-trip identities the reference tests (ref:tests/test_functions.py:44-63).
"""
import numpy as np


def euler_angles_to_R(roll, pitch, yaw):
    """R = Rz(yaw) Ry(pitch) Rx(roll)  (ref:bundle_adjust/ba_rotate.py:85-94)."""
    ca, sa = np.cos(roll), np.sin(roll)
    cb, sb = np.cos(pitch), np.sin(pitch)
    cg, sg = np.cos(yaw), np.sin(yaw)
    # closed form of the triple product
    return np.array(
        [
            [cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa],
            [sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa],
            [-sb, cb * sa, cb * ca],
        ],
        dtype=np.float64,
    )


def euler_angles_from_R(R):
    """Inverse of euler_angles_to_R, with the gimbal-lock branch (ref:bundle_adjust/ba_rotate.py:67-82)."""
    sy = np.hypot(R[0, 0], R[1, 0])
    if sy >= 1e-6:
        roll = np.arctan2(R[2, 1], R[2, 2])
        yaw = np.arctan2(R[1, 0], R[0, 0])
    else:
        roll = np.arctan2(-R[1, 2], R[1, 1])
        yaw = 0
    pitch = np.arctan2(-R[2, 0], sy)
    return roll, pitch, yaw


def euler_to_quaternion(roll, pitch, yaw):
    """Unit quaternion (w, x, y, z) of Rz(yaw) Ry(pitch) Rx(roll) (ref:bundle_adjust/ba_rotate.py:12-24)."""
    cr, sr = np.cos(0.5 * roll), np.sin(0.5 * roll)
    cp, sp = np.cos(0.5 * pitch), np.sin(0.5 * pitch)
    cy, sy = np.cos(0.5 * yaw), np.sin(0.5 * yaw)
    q0 = cr * cp * cy + sr * sp * sy
    q1 = sr * cp * cy - cr * sp * sy
    q2 = cr * sp * cy + sr * cp * sy
    q3 = cr * cp * sy - sr * sp * cy
    return q0, q1, q2, q3


def quaternion_to_R(q0, q1, q2, q3):
    """Rotation matrix of a unit quaternion (ref:bundle_adjust/ba_rotate.py:27-42)."""
    return np.array(
        [
            [1 - 2 * (q2 * q2 + q3 * q3), 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2)],
            [2 * (q1 * q2 + q0 * q3), 1 - 2 * (q1 * q1 + q3 * q3), 2 * (q2 * q3 - q0 * q1)],
            [2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 1 - 2 * (q1 * q1 + q2 * q2)],
        ]
    )


def quaternion_to_euler(q0, q1, q2, q3):
    return euler_angles_from_R(quaternion_to_R(q0, q1, q2, q3))


def R_to_quaternion(R):
    return euler_to_quaternion(*euler_angles_from_R(R))
