"""Scene description: convex-decomposed shape templates and the Sawyer-like arm.

Replaces the URDF / OBJ ingest of the reference
(``robovat/simulation/physics/bullet_physics.py:143-186``,
``tools/convert_obj_to_urdf.py:211-344``): the reference's ``assets/``
directory is not distributed with its source (README.md:48-59), so the shapes
here are BUILD-CHOSEN procedural bodies plus V-HACD fixtures generated with the
reference's own ``bin/vhacd`` (``tests/golden/gen_vhacd_fixtures.py``).

Every template is expressed in its centre-of-mass / principal-axes frame at
unit scale, which is what the device kernels assume (diagonal inertia).
"""
import json
import os

import numpy as np

from robovat_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE_PATH = os.path.join(_HERE, 'assets', 'vhacd_hulls.json')


# --------------------------------------------------------------- mass props
def _hull_triangles(verts):
    from scipy.spatial import ConvexHull
    hull = ConvexHull(verts)
    centre = verts.mean(axis=0)
    tris = []
    for simplex, eq in zip(hull.simplices, hull.equations):
        a, b, c = verts[simplex]
        n = np.cross(b - a, c - a)
        if np.dot(n, eq[:3]) < 0:
            b, c = c, b
        tris.append((a, b, c))
    return tris, centre


def _mass_properties(hulls):
    """Volume, centre of mass and inertia tensor (unit density) of a union of
    convex hulls (overlaps ignored), by signed tetrahedra."""
    vol = 0.0
    com = np.zeros(3)
    # second moments about the origin
    C = np.zeros((3, 3))
    canon = np.array([[2, 1, 1], [1, 2, 1], [1, 1, 2]]) / 120.0
    for verts in hulls:
        tris, _ = _hull_triangles(np.asarray(verts, dtype=np.float64))
        for a, b, c in tris:
            A = np.stack([a, b, c], axis=1)
            det = np.linalg.det(A)
            vol += det / 6.0
            com += det / 24.0 * (a + b + c)
            C += det * A @ canon @ A.T
    com /= vol
    C -= vol * np.outer(com, com)
    inertia = np.trace(C) * np.eye(3) - C
    return vol, com, inertia


def hull_planes(verts, tol=1e-6):
    """Face planes (n, d) with n.x <= d of a convex hull, coplanar triangles merged."""
    from scipy.spatial import ConvexHull
    hull = ConvexHull(verts)
    planes = []
    for eq in hull.equations:              # n.x + off <= 0 inside
        n, d = eq[:3], -eq[3]
        if not any(np.dot(n, q[:3]) > 1.0 - tol and abs(d - q[3]) < tol for q in planes):
            planes.append(np.array([n[0], n[1], n[2], d]))
    return np.array(planes)


def make_shape(hulls):
    """Build an ``rv_shape`` from a list of [n, 3] vertex arrays."""
    hulls = [np.asarray(h, dtype=np.float64) for h in hulls]
    assert 1 <= len(hulls) <= abi.RV_MAXH
    vol, com, inertia = _mass_properties(hulls)
    off = inertia - np.diag(np.diag(inertia))
    if np.abs(off).max() < 1e-6 * np.trace(inertia):
        # already principal (symmetric primitives): keep the authoring axes so
        # that repeated eigenvalues cannot rotate the template arbitrarily
        evals, evecs = np.diag(inertia).copy(), np.eye(3)
    else:
        evals, evecs = np.linalg.eigh(inertia)
        if np.linalg.det(evecs) < 0:
            evecs[:, 2] *= -1.0
    shape = abi.rv_shape()
    shape.n_hulls = len(hulls)
    radius = 0.0
    for h, verts in enumerate(hulls):
        assert 4 <= len(verts) <= abi.RV_MAXV, len(verts)
        local = (verts - com) @ evecs
        shape.n_verts[h] = len(local)
        for i, v in enumerate(local):
            for k in range(3):
                shape.verts[h][i][k] = float(v[k])
        radius = max(radius, float(np.linalg.norm(local, axis=1).max()))
        planes = hull_planes(local)
        assert len(planes) <= abi.RV_MAXP, len(planes)
        shape.n_planes[h] = len(planes)
        for i, pl in enumerate(planes):
            for k in range(4):
                shape.planes[h][i][k] = float(pl[k])
    for k in range(3):
        shape.inertia_k[k] = float(evals[k] / vol)
    shape.radius = radius
    return shape


def shape_to_arrays(shape):
    """List of [n, 3] float32 arrays (one per hull) of an ``rv_shape``."""
    out = []
    for h in range(shape.n_hulls):
        n = shape.n_verts[h]
        out.append(np.array([[shape.verts[h][i][k] for k in range(3)]
                             for i in range(n)], dtype=np.float32))
    return out


# ------------------------------------------------------- procedural shapes
def box_hull(hx, hy, hz, centre=(0, 0, 0)):
    c = np.asarray(centre, dtype=np.float64)
    return np.array([[sx * hx, sy * hy, sz * hz]
                     for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)],
                    dtype=np.float64) + c


def cylinder_hull(radius, half_height, n=8):
    """Cylinder as a 2n-vertex prism (n=8 gives the 16-vertex hull of
    BASELINE.md config 2)."""
    ang = np.arange(n) * 2 * np.pi / n
    ring = np.stack([radius * np.cos(ang), radius * np.sin(ang)], axis=1)
    top = np.concatenate([ring, np.full((n, 1), half_height)], axis=1)
    bot = np.concatenate([ring, np.full((n, 1), -half_height)], axis=1)
    return np.concatenate([top, bot], axis=0)


def random_hull(rng, n=12, extent=(0.04, 0.03, 0.03)):
    """Random convex hull with at most ``n`` vertices."""
    from scipy.spatial import ConvexHull
    pts = rng.normal(size=(n, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    pts *= np.asarray(extent)
    hull = ConvexHull(pts)
    return pts[hull.vertices]


def load_vhacd_fixtures(path=FIXTURE_PATH):
    """Concave movables as V-HACD hull lists: name -> list of [n,3] arrays."""
    with open(path) as f:
        data = json.load(f)
    return {k: [np.asarray(h, dtype=np.float64) for h in v]
            for k, v in data.items()}


CONVEX_SET = ('box', 'cylinder16', 'hull_a', 'hull_b')


def default_shape_hulls():
    """name -> hull list for the default template library.

    Index order is part of the scene contract: 0..3 are the four convex
    templates of BASELINE.md config 2, the rest are the V-HACD concave
    fixtures of config 3."""
    rng = np.random.RandomState(7)
    shapes = [
        ('box', [box_hull(0.035, 0.03, 0.03)]),
        ('cylinder16', [cylinder_hull(0.03, 0.03, 8)]),
        ('hull_a', [random_hull(rng, 14, (0.045, 0.035, 0.03))]),
        ('hull_b', [random_hull(rng, 16, (0.035, 0.04, 0.035))]),
    ]
    if os.path.exists(FIXTURE_PATH):
        for name, hulls in sorted(load_vhacd_fixtures().items()):
            shapes.append((name, hulls))
    # graspables of BASELINE config 4: narrower than the open gripper (3.8 cm between the pads)
    shapes += [
        ('grasp_cube', [box_hull(0.014, 0.014, 0.014)]),
        ('grasp_bar', [box_hull(0.012, 0.035, 0.015)]),
        ('grasp_cyl', [cylinder_hull(0.014, 0.02, 8)]),
    ]
    # the wall of ArmEnv._reset_scene (arm_env.py:94-99; SIM.WALL.PATH is not part of the reference tree: BUILD-CHOSEN
    # slab, 4 cm thick, 1.3 m wide, 0.8 m high) -- loaded as a static body when SIM.WALL.USE is set
    shapes += [('wall', [box_hull(0.02, 0.65, 0.4)])]
    return shapes


# -------------------------------------------------------------------- arm
def _rpy_quat(r, p, y):
    """Fixed-axis (static xyz) roll/pitch/yaw -> xyzw quaternion."""
    ci, si = np.cos(r / 2), np.sin(r / 2)
    cj, sj = np.cos(p / 2), np.sin(p / 2)
    ck, sk = np.cos(y / 2), np.sin(y / 2)
    return np.array([si * cj * ck - ci * sj * sk,
                     ci * sj * ck + si * cj * sk,
                     ci * cj * sk - si * sj * ck,
                     ci * cj * ck + si * sj * sk])


# Sawyer-like 7-DoF chain.  RECALLED from the public `sawyer_description`
#  and therefore shipped as the build's own
# "sawyer-like" arm: joint origins parent->child (xyz; rpy), revolute about z.
SAWYER_JOINT_ORIGINS = [
    ((0.0, 0.0, 0.08), (0.0, 0.0, 0.0)),                      # right_j0
    ((0.081, 0.05, 0.237), (-np.pi / 2, np.pi / 2, 0.0)),     # right_j1
    ((0.0, -0.14, 0.1425), (np.pi / 2, 0.0, 0.0)),            # right_j2
    ((0.0, -0.042, 0.26), (-np.pi / 2, 0.0, 0.0)),            # right_j3
    ((0.0, -0.125, -0.1265), (np.pi / 2, 0.0, 0.0)),          # right_j4
    ((0.0, 0.031, 0.275), (-np.pi / 2, 0.0, 0.0)),            # right_j5
    ((0.0, -0.11, 0.1053), (-np.pi / 2, -0.17453, np.pi)),    # right_j6
    ((0.0, 0.0, 0.0245), (0.0, 0.0, np.pi / 2)),              # right_hand (fixed)
]
SAWYER_LIMITS = [(-3.0503, 3.0503), (-3.8095, 2.2736), (-3.0426, 3.0426),
                 (-3.0439, 3.0439), (-2.9761, 2.9761), (-2.9761, 2.9761),
                 (-4.7124, 4.7124)]
SAWYER_MAX_VELOCITY = [1.74, 1.328, 1.957, 1.957, 3.485, 3.485, 4.545]
SAWYER_MAX_ACCEL = [8.0, 8.0, 10.0, 10.0, 15.0, 15.0, 20.0]
SAWYER_EFFORT = [80.0, 80.0, 40.0, 40.0, 9.0, 9.0, 9.0]     # N m
FINGER_EFFORT = 20.0                                         # N
# <inertial> of right_l0 .. right_l6 (mass kg, centre of mass in the link frame m, principal moments kg m^2;
# RECALLED from sawyer_description like the chain above), and the hand with the
# electric gripper lumped on the hand frame.  Read by PHYSICS.LIMB_DYNAMICS only
SAWYER_INERTIAL = [
    (5.3213, (0.024366, 0.010969, 0.14363), (0.053314, 0.057902, 0.023659)),
    (4.505, (-0.0030849, -0.026811, 0.092521), (0.022398, 0.014613, 0.017295)),
    (1.745, (-0.00016044, -0.014967, 0.13582), (0.025506, 0.0253, 0.0034179)),
    (2.5097, (-0.0048135, -0.0281, -0.084154), (0.01016, 0.0065685, 0.0069078)),
    (1.1136, (-0.0018844, 0.0069001, 0.1341), (0.013557, 0.013555, 0.0013658)),
    (1.5625, (0.0061133, -0.023697, 0.076416), (0.0047328, 0.0029676, 0.0031762)),
    (0.3292, (-8.0726e-06, 0.0085838, -0.0049566), (0.00031105, 0.00021549, 0.00035976)),
    (0.8, (0.0, 0.0, 0.06), (0.0012, 0.0012, 0.0006)),
]
LIMB_JOINT_NAMES = ['right_j%d' % i for i in range(7)]
FINGER_JOINT_NAMES = ['right_gripper_l_finger_joint',
                      'right_gripper_r_finger_joint']
FINGER_STROKE = 0.0208
LINK_NAMES = (['right_l%d' % i for i in range(7)] +
              ['right_hand', 'right_gripper_l_finger_tip',
               'right_gripper_r_finger_tip'])
FINGER_TIP_OFFSET = 0.14
# half-width added around each link's segment.  Link 0 is the round base housing that
# only turns about the vertical axis: its box is the one INSCRIBED in the 7 cm housing
# (0.07 / sqrt 2), because the corners of a circumscribed box would sweep through space
# the housing never enters and brush bodies resting at the near table edge (x = 0.22)
_LINK_RADIUS = [0.05, 0.07, 0.06, 0.06, 0.05, 0.05, 0.045]


def make_arm(base_pos=(0.0, 0.0, 0.0), base_rpy=(0.0, 0.0, 0.0), finger_accel=2.0):
    """``finger_accel``: acceleration limit of the finger joints (m/s^2); with the force-limited
    gripper (PHYSICS.FINGER_DYNAMICS) it is FINGER_MAX_FORCE / FINGER_MASS."""
    arm = abi.rv_arm()
    abi.assign(arm.base_pos, base_pos)
    abi.assign(arm.base_quat, _rpy_quat(*base_rpy).tolist())
    for i, (xyz, rpy) in enumerate(SAWYER_JOINT_ORIGINS):
        abi.assign(arm.jpos[i], xyz)
        abi.assign(arm.jquat[i], _rpy_quat(*rpy).tolist())
    for j in range(7):
        arm.q_lo[j], arm.q_hi[j] = SAWYER_LIMITS[j]
        arm.v_max[j] = SAWYER_MAX_VELOCITY[j]
        arm.a_max[j] = SAWYER_MAX_ACCEL[j]
        arm.inv_tau_max[j] = 1.0 / SAWYER_EFFORT[j]
    for i, (m, com, inertia) in enumerate(SAWYER_INERTIAL):
        arm.link_mass[i] = m
        abi.assign(arm.link_com[i], com)
        abi.assign(arm.link_inertia[i], inertia)
    # electric parallel gripper: left finger 0..+stroke, right -stroke..0
    arm.q_lo[7], arm.q_hi[7] = 0.0, FINGER_STROKE
    arm.q_lo[8], arm.q_hi[8] = -FINGER_STROKE, 0.0
    arm.v_max[7] = arm.v_max[8] = 0.1
    arm.a_max[7] = arm.a_max[8] = float(finger_accel)
    arm.inv_tau_max[7] = arm.inv_tau_max[8] = 1.0 / FINGER_EFFORT
    # open gap (2 x (0.004 + stroke) - pad thickness) ~ 3.8 cm: narrower than the smallest movable, so a
    # push cannot straddle a body
    arm.finger_y0[0], arm.finger_y0[1] = 0.004, -0.004
    # collider boxes: limb link i spans from its frame to the next joint origin
    for i in range(7):
        nxt = np.asarray(SAWYER_JOINT_ORIGINS[i + 1][0])
        arm.col_frame[i] = i
        abi.assign(arm.col_center[i], (0.5 * nxt).tolist())
        abi.assign(arm.col_half[i],
                   (0.5 * np.abs(nxt) + _LINK_RADIUS[i]).tolist())
    arm.col_frame[7] = 7  # gripper base on the hand frame
    abi.assign(arm.col_center[7], (0.0, 0.0, 0.04))
    abi.assign(arm.col_half[7], (0.03, 0.05, 0.03))
    for k in range(2):    # finger pads
        arm.col_frame[8 + k] = 8 + k
        abi.assign(arm.col_center[8 + k], (0.0, 0.0, 0.105))
        abi.assign(arm.col_half[8 + k], (0.008, 0.006, 0.035))
    return arm


def make_scene(shape_hulls=None, env_cfg=None, arm=None):
    """Build the ``rv_scene`` (shape templates + arm).  Returns
    (scene, names).  ``env_cfg``: the env config, for the settings that live in the scene
    (the finger acceleration limit of the force-limited gripper).  ``arm``: an ``rv_arm`` to use
    instead of the built-in one, e.g. ``io.asset_ingest.arm_from_urdf(ARM_URDF, 'right_hand')``."""
    if shape_hulls is None:
        shape_hulls = default_shape_hulls()
    assert len(shape_hulls) <= abi.RV_MAX_SHAPES
    scene = abi.rv_scene()
    scene.n_shapes = len(shape_hulls)
    names = []
    for i, (name, hulls) in enumerate(shape_hulls):
        scene.shapes[i] = make_shape(hulls)
        names.append(name)
    accel = 2.0
    if env_cfg is not None and env_cfg.PHYSICS.get('FINGER_DYNAMICS'):
        accel = env_cfg.PHYSICS.FINGER_MAX_FORCE / env_cfg.PHYSICS.FINGER_MASS
    if arm is None:
        scene.arm = make_arm(finger_accel=accel)
    else:
        scene.arm = arm
        scene.arm.a_max[7] = scene.arm.a_max[8] = float(accel)
    # PHYSICS.ARM_ACCEL_SCALE (tools/sensitivity.py): the BUILD-CHOSEN acceleration limits of the limb joints scaled
    # (PyBullet's POSITION_CONTROL motors have none: with their large default force they reach the commanded
    # velocity within a step)
    if env_cfg is not None and env_cfg.PHYSICS.get('ARM_ACCEL_SCALE') is not None:
        for j in range(7):
            scene.arm.a_max[j] = float(scene.arm.a_max[j]) * float(env_cfg.PHYSICS.ARM_ACCEL_SCALE)
    return scene, names
