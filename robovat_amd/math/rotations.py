"""Rotation conversions with RoboVat's conventions: quaternions are xyzw,
Euler angles are static-xyz ("sxyz": roll about x, then pitch about y, then yaw
about z, all in the fixed frame).

Own numpy implementation of the subset of ``third_party/transformations.py``
(reference lines 1034-1359) that the hot path uses; pinned by the known-answer
vectors in ``tests/golden/math_golden.json`` generated from the reference.
Unlike the reference's ``quaternion_from_matrix3`` (trace branch only, SURVEY.md
Appendix B-3) the matrix->quaternion conversion here is Shepperd's method and is
valid for every rotation.
"""
import numpy as np

_EPS = np.finfo(np.float64).eps * 4.0


def quaternion_from_euler(roll, pitch, yaw):
    ci, si = np.cos(roll * 0.5), np.sin(roll * 0.5)
    cj, sj = np.cos(pitch * 0.5), np.sin(pitch * 0.5)
    ck, sk = np.cos(yaw * 0.5), np.sin(yaw * 0.5)
    return np.array([si * cj * ck - ci * sj * sk,
                     ci * sj * ck + si * cj * sk,
                     ci * cj * sk - si * sj * ck,
                     ci * cj * ck + si * sj * sk], dtype=np.float64)


def matrix3_from_quaternion(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.dot(q, q)
    if n < _EPS:
        return np.identity(3)
    x, y, z, w = q * np.sqrt(2.0 / n)
    return np.array([
        [1.0 - y * y - z * z, x * y - z * w, x * z + y * w],
        [x * y + z * w, 1.0 - x * x - z * z, y * z - x * w],
        [x * z - y * w, y * z + x * w, 1.0 - x * x - y * y]], dtype=np.float64)


def matrix3_from_euler(roll, pitch, yaw):
    return matrix3_from_quaternion(quaternion_from_euler(roll, pitch, yaw))


def euler_from_matrix3(m):
    """Static-xyz Euler angles of a rotation matrix."""
    m = np.asarray(m, dtype=np.float64)[:3, :3]
    cy = np.sqrt(m[0, 0] * m[0, 0] + m[1, 0] * m[1, 0])
    if cy > _EPS:
        roll = np.arctan2(m[2, 1], m[2, 2])
        pitch = np.arctan2(-m[2, 0], cy)
        yaw = np.arctan2(m[1, 0], m[0, 0])
    else:
        roll = np.arctan2(-m[1, 2], m[1, 1])
        pitch = np.arctan2(-m[2, 0], cy)
        yaw = 0.0
    return np.array([roll, pitch, yaw], dtype=np.float64)


def euler_from_quaternion(q):
    return euler_from_matrix3(matrix3_from_quaternion(q))


def quaternion_from_matrix3(m):
    """Shepperd's method (robust near 180 degrees)."""
    m = np.asarray(m, dtype=np.float64)[:3, :3]
    t = np.trace(m)
    if t > 0.0:
        s = np.sqrt(t + 1.0) * 2.0
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2.0
        q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2.0
        q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2.0
        q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    return np.array(q, dtype=np.float64)


def quaternion_multiply(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], dtype=np.float64)


def quaternion_angle(a, b):
    """Geodesic angle (rad) between unit quaternions ``a`` and ``b`` (arrays [..., 4], xyzw),
    accurate for tiny angles: 2 atan2(|vec(a b*)|, |w(a b*)|)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    ax, ay, az, aw = (a[..., i] for i in range(4))
    bx, by, bz, bw = (-b[..., 0], -b[..., 1], -b[..., 2], b[..., 3])
    x = aw * bx + ax * bw + ay * bz - az * by
    y = aw * by - ax * bz + ay * bw + az * bx
    z = aw * bz + ax * by - ay * bx + az * bw
    w = aw * bw - ax * bx - ay * by - az * bz
    return 2.0 * np.arctan2(np.sqrt(x * x + y * y + z * z), np.abs(w))
