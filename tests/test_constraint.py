"""User constraints (Simulator.add_constraint, simulator.py:166-224; ControllableConstraint,
controllable_constraint.py:21-170): a fixed joint between a movable body and a frame of the world,
limited to max_force.  Known answers on the oracle and on the HIP library, HIP == oracle, and the
reference-shaped object API on the GPU."""
import numpy as np
import pytest

from robovat_amd import abi, configs, scenes

import test_kat_contact as T

BACKENDS = ['oracle64', 'oracle32', pytest.param('hip', marks=pytest.mark.gpu)]
Q0 = (0, 0, 0, 1)


@pytest.mark.parametrize('backend', BACKENDS)
def test_constraint_holds_a_body_against_gravity_within_its_force_limit(backend):
    """0.2 kg box held 10 cm above the table: with 50 N per row it hangs at the target; with 1 N
    (less than its weight, 1.96 N) it sinks to the table; removing the constraint drops it."""
    w, cfg = T._world(backend)
    T._bodies(w, [(0, 0.2, 0.5, (0.6, 0.0, 0.131), Q0, (0, 0, 0))])
    w.set_constraint(0, [0.6, 0.0, 0.131, 0, 0, 0, 1], max_force=50.0)
    w.step_sub(800)
    st = w.body_state()[0, 0]
    assert np.abs(st[:3] - [0.6, 0.0, 0.131]).max() < 1.5e-4 and np.abs(st[7:13]).max() < 1e-3, st[:3]
    # move the world frame: the body follows (stiffness erp / dt: 1 mm per substep at most here)
    w.set_constraint(0, [0.65, 0.02, 0.16, 0, 0, np.sin(0.3), np.cos(0.3)], max_force=50.0)
    w.step_sub(1500)
    st = w.body_state()[0, 0]
    assert np.abs(st[:3] - [0.65, 0.02, 0.16]).max() < 3e-4
    assert min(np.abs(st[3:7] - [0, 0, np.sin(0.3), np.cos(0.3)]).max(), np.abs(st[3:7] + [0, 0, np.sin(0.3), np.cos(0.3)]).max()) < 2e-3
    # too weak to carry the weight: the body comes down onto the table
    w.set_constraint(0, [0.65, 0.02, 0.16, 0, 0, np.sin(0.3), np.cos(0.3)], max_force=1.0)
    w.step_sub(1500)
    assert w.body_state()[0, 0, 2] < 0.04
    w.remove_constraint(0)
    w.step_sub(300)
    assert abs(w.body_state()[0, 0, 2] - 0.031) < 2e-3
    if hasattr(w, 'w'):
        w.close()


@pytest.mark.parametrize('backend', BACKENDS)
def test_constraint_with_an_offset_frame_and_contacts(backend):
    """The joint frame sits on top of the box; the constraint presses the box onto the table next to
    a second, free box -- contacts and constraint rows in one system."""
    w, cfg = T._world(backend)
    T._bodies(w, [(0, 0.2, 0.5, (0.6, 0.0, 0.031), Q0, (0, 0, 0)), (0, 0.3, 0.5, (0.6, 0.2, 0.031), Q0, (0, 0, 0))])
    w.set_constraint(0, [0.62, 0.01, 0.061, 0, 0, 0, 1], frame7=[0, 0, 0.03, 0, 0, 0, 1], max_force=20.0)
    w.step_sub(1000)
    st = w.body_state()[0]
    assert np.abs(st[0, :2] - [0.62, 0.01]).max() < 5e-4 and abs(st[0, 2] - 0.031) < 1e-3      # dragged along the table
    assert np.abs(st[1, :3] - [0.6, 0.2, 0.031]).max() < 1e-3                                  # the other box is untouched
    if hasattr(w, 'w'):
        w.close()


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize('backend', BACKENDS)
def test_point2point_pivot_holds_and_the_body_swings(backend):
    """pybullet JOINT_POINT2POINT to the world: the pivot (a corner-side point of the box) stays at its world
    point while the box swings under it like a pendulum -- three rows, the rotation is free."""
    w, cfg = T._world(backend)
    T._bodies(w, [(0, 0.2, 0.5, (0.6, 0.0, 0.25), Q0, (0, 0, 0))])
    piv_local = np.array([0.03, 0.0, 0.0])
    piv_world = np.array([0.63, 0.0, 0.25])
    w.set_constraint(0, list(piv_world) + [0, 0, 0, 1], frame7=list(piv_local) + [0, 0, 0, 1], max_force=50.0, joint_type='point2point')
    zmin, tilt = 1.0, 0.0
    for _ in range(40):
        w.step_sub(25)
        st = np.asarray(w.body_state())[0, 0]
        piv = st[:3] + _rot(st[3:7]) @ piv_local
        assert np.abs(piv - piv_world).max() < 1e-3, piv                      # the pivot holds
        zmin = min(zmin, st[2]); tilt = max(tilt, abs(st[4]))
    assert zmin < 0.25 - 0.02 and tilt > 0.3                                  # the centre swung down under the pivot: it turned about y
    if hasattr(w, 'w'):
        w.close()


@pytest.mark.parametrize('backend', BACKENDS)
def test_fixed_joint_between_two_bodies(backend):
    """A fixed joint between two movable bodies: body 0 is carried by a world constraint and moved; body 1, fixed to
    body 0 at an offset of 8 cm along y, follows it through the air keeping the relative pose; removing the
    body--body joint drops body 1 only."""
    w, cfg = T._world(backend)
    T._bodies(w, [(0, 0.2, 0.5, (0.6, 0.0, 0.15), Q0, (0, 0, 0)), (0, 0.1, 0.5, (0.6, 0.08, 0.15), Q0, (0, 0, 0))])
    w.set_constraint(0, [0.6, 0.0, 0.15, 0, 0, 0, 1], max_force=100.0)
    # the joint frame = body 1's frame, seen from body 0 (the child here) at (0, 0.08, 0)
    w.set_constraint(1, [0, 0.08, 0, 0, 0, 0, 1], max_force=100.0, child=0)
    w.step_sub(400)
    st = np.asarray(w.body_state())[0]
    assert np.abs(st[1, :3] - st[0, :3] - [0, 0.08, 0]).max() < 1e-3 and abs(st[0, 2] - 0.15) < 1e-3 and abs(st[1, 2] - 0.15) < 2e-3
    for i in range(60):        # carry body 0 along x and turn it about z: body 1 goes round with it
        a = 0.3 * (i + 1) / 60
        w.set_constraint(0, [0.6 + 0.001 * (i + 1), 0.0, 0.15, 0, 0, np.sin(a / 2), np.cos(a / 2)], max_force=100.0)
        w.step_sub(10)
    w.step_sub(300)
    st = np.asarray(w.body_state())[0]
    want = st[0, :3] + _rot(st[0, 3:7]) @ [0, 0.08, 0]
    assert abs(st[0, 0] - 0.66) < 1e-3 and np.abs(st[1, :3] - want).max() < 1.5e-3, (st[0, :3], st[1, :3], want)
    assert min(np.abs(st[1, 3:7] - st[0, 3:7]).max(), np.abs(st[1, 3:7] + st[0, 3:7]).max()) < 3e-3
    assert abs(abs(st[0, 5]) - np.sin(0.15)) < 3e-3
    w.remove_constraint(1)
    w.step_sub(400)
    st = np.asarray(w.body_state())[0]
    assert st[1, 2] < 0.04 and abs(st[0, 2] - 0.15) < 1e-3
    if hasattr(w, 'w'):
        w.close()



@pytest.mark.gpu
def test_hip_equals_oracle_with_constraints():
    from robovat_amd import lib
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(n_envs=6, seed=8, shape_names=names)
    w, ref = lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=False)
    w.reset(); ref.reset()
    st = ref.body_state()[0]
    tgt = [float(st[1, 0]) + 0.03, float(st[1, 1]) - 0.02, float(st[1, 2]) + 0.05, 0, 0, np.sin(0.2), np.cos(0.2)]
    for x in (w, ref):
        x.set_constraint(1, tgt, frame7=[0.01, 0, 0.0, 0, 0, 0, 1], max_force=30.0)
    w.step_sub(400); ref.step_sub(400)
    assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    a = ref.policy_random(0)
    w.set_actions(a); ref.set_actions(a); w.step_macro(); ref.step_macro()      # a push with the constraint in place
    assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    for x in (w, ref):
        x.remove_constraint(1)
    w.step_sub(300); ref.step_sub(300)
    assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    # a point-to-point joint to the world on body 1 and a fixed joint body 2 -> body 0, then a push through them
    st = ref.body_state()[0]
    for x in (w, ref):
        x.set_constraint(1, [float(st[1, 0]), float(st[1, 1]), float(st[1, 2]) + 0.04, 0, 0, 0, 1], frame7=[0.02, 0.01, 0.0, 0, 0, 0, 1],
                         max_force=30.0, joint_type='point2point')
        x.set_constraint(2, [0.0, 0.0, 0.07, 0, 0, 0, 1], max_force=40.0, child=0)
    for k in range(3):
        w.step_sub(150); ref.step_sub(150)
        assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32)), k
    a = ref.policy_random(1)
    w.set_actions(a); ref.set_actions(a); w.step_macro(); ref.step_macro()
    assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    # prismatic joints: body 1 on a rail through the air (world), body 3 sliding on body 0
    st = ref.body_state()[0]
    qz = [0, 0, np.sin(0.4), np.cos(0.4)]
    for x in (w, ref):
        x.remove_constraint(1); x.remove_constraint(2)
        x.set_constraint(1, [float(st[1, 0]), float(st[1, 1]), float(st[1, 2]) + 0.03] + qz, frame7=[0, 0, 0.01] + qz, max_force=60.0, joint_type='prismatic')
        x.set_constraint(3, [0.0, 0.0, 0.08, 0, 0, 0, 1], max_force=40.0, child=0, joint_type='prismatic')
        x.set_constraint(2, [float(st[2, 0]), float(st[2, 1]), float(st[2, 2]) + 0.02] + qz, frame7=[0.015, 0, 0.0] + qz, max_force=50.0, joint_type='revolute')
    for k in range(3):
        w.step_sub(150); ref.step_sub(150)
        assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32)), k
    a = ref.policy_random(2)
    w.set_actions(a); ref.set_actions(a); w.step_macro(); ref.step_macro()
    assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    w.close()


@pytest.mark.gpu
def test_simulator_add_constraint_and_pose_servo():
    """The reference-shaped API: Simulator.add_constraint(is_controllable=True) + set_target_pose; the
    servo moves the world frame by max_linear_velocity * dt per Simulator.step()."""
    from robovat_amd.simulation import Simulator
    sim = Simulator(physics_backend='HipPhysics')
    sim.reset()
    sim.start()
    body = sim.add_body('box', pose=[[0.6, 0.0, 0.1], [0, 0, 0]], name='box')
    con = sim.add_constraint(body, None, joint_type='fixed', max_force=40.0, max_linear_velocity=0.1,
                             max_angular_velocity=1.0, is_controllable=True, name='mocap')
    assert 'mocap' in sim.constraints and con.max_force == 40.0 and con.is_ready()
    p0 = np.asarray(body.position).copy()
    for _ in range(200):
        sim.step()
    assert np.abs(np.asarray(body.position) - p0).max() < 5e-4            # held where it was created
    con.set_target_pose([[p0[0] + 0.03, p0[1], p0[2] + 0.02], [0, 0, 0]], timeout=2.0)
    assert not con.is_ready()
    for _ in range(600):
        sim.step()
    assert con.is_ready()                                                 # reached (checked every 100 steps)
    # (the servo stops within POSITION_THRESHOLD = 1 cm of the target, controllable_constraint.py:16,135-156)
    assert np.abs(np.asarray(body.position) - [p0[0] + 0.03, p0[1], p0[2] + 0.02]).max() < 0.0105
    assert np.abs(np.asarray(body.position) - p0).max() > 0.015
    with pytest.raises(ValueError):
        sim.add_constraint(body, body, joint_type='fixed')
    with pytest.raises(NotImplementedError):
        sim.add_constraint(body, None, joint_type='gear')
    sim.remove_constraint('mocap')
    for _ in range(400):
        sim.step()
    assert body.position[2] < p0[2] - 0.03                                # free again: it falls
    # a prismatic joint along joint_axis = y (given in the joint frame): under gravity tilted towards +y the body
    # rides the rail -- y grows, x and z stay
    sim.physics.set_gravity([0.0, 1.5, -9.8])
    rail = sim.add_body('box', pose=[[0.45, -0.2, 0.25], [0, 0, 0]], name='rail_rider')
    sim.add_constraint(rail, None, joint_type='prismatic', joint_axis=[0, 1, 0], max_force=100.0, name='rail')
    q0 = np.asarray(rail.position).copy()
    for _ in range(300):
        sim.step()
    q1 = np.asarray(rail.position)
    assert 0.05 < q1[1] - q0[1] < 0.075 and abs(q1[0] - q0[0]) < 1e-3 and abs(q1[2] - q0[2]) < 1e-3, (q0, q1)


@pytest.mark.gpu
def test_constraint_entry_point_rejects_what_is_not_built():
    """rv_set_constraint_ex: unknown joint types are RV_ERR_NOTIMPL, a bad body / child slot or a null target RV_ERR_VALUE
    (the reference's JOINT_TYPES_MAPPING, bullet_physics.py:20-25, has revolute / prismatic / fixed / point2point: all four are built)."""
    import ctypes as C
    from robovat_amd import lib
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(n_envs=2, seed=3, shape_names=names)
    w = lib.World(cfg, scene, device=0)
    w.reset()
    L = lib.load()
    t7 = (C.c_float * 7)(0.6, 0.0, 0.1, 0, 0, 0, 1)
    call = lambda body, child, jt, tgt, f: L.rv_set_constraint_ex(w.h, body, child, jt, None, tgt, f)
    assert call(0, -1, 0, t7, 10.0) == abi.RV_OK                      # pybullet.JOINT_REVOLUTE ('revolute' of the reference's mapping)
    assert call(0, -1, 6, t7, 10.0) == abi.RV_ERR_NOTIMPL            # pybullet.JOINT_GEAR
    assert call(abi.RV_MAXB, -1, 4, t7, 10.0) == abi.RV_ERR_VALUE     # not a movable body slot
    assert call(0, 0, 4, t7, 10.0) == abi.RV_ERR_VALUE                # a body cannot be its own child
    assert call(0, -1, 4, None, 10.0) == abi.RV_ERR_VALUE             # no target
    assert call(0, 1, 5, t7, 10.0) == abi.RV_OK                       # point2point between two bodies
    assert call(0, -1, 4, None, -1.0) == abi.RV_OK                    # removal needs no target
    assert call(0, -1, 1, t7, 10.0) == abi.RV_OK                      # pybullet.JOINT_PRISMATIC
    with pytest.raises(NotImplementedError):
        w.set_constraint(0, [0.6, 0, 0.1, 0, 0, 0, 1], joint_type='gear')
    w.close()


@pytest.mark.parametrize('backend', BACKENDS)
def test_prismatic_joint_slides_along_its_axis_only(backend):
    """pybullet JOINT_PRISMATIC to the world: the box can only slide along the x axis of the joint frame (here the
    world's x turned by 30 degrees about z).  Under a gravity vector with a horizontal component it accelerates along
    that axis with the component of gravity along it (semi-implicit Euler with Bullet's damping), stays on the line
    and keeps its orientation."""
    gx, gy = 2.0, 1.0
    w, cfg = T._world(backend, **{'PHYSICS.GRAVITY_XY': (gx, gy)})
    T._bodies(w, [(0, 0.2, 0.5, (0.5, -0.1, 0.2), Q0, (0, 0, 0))])
    a = np.radians(30.0)
    qz = [0, 0, np.sin(a / 2), np.cos(a / 2)]
    ax = np.array([np.cos(a), np.sin(a), 0.0])
    # the joint frame of the body is turned like the world frame's, so that the body keeps the identity orientation
    w.set_constraint(0, [0.5, -0.1, 0.2] + qz, frame7=[0, 0, 0] + qz, max_force=100.0, joint_type='prismatic')
    Tn = 300
    w.step_sub(Tn)
    st = np.asarray(w.body_state())[0, 0]
    acc, v, x, damp = gx * ax[0] + gy * ax[1], 0.0, 0.0, float(cfg.lin_damp)
    for _ in range(Tn):
        v = (v + acc * float(cfg.dt)) * damp; x += v * float(cfg.dt)
    d = st[:3] - [0.5, -0.1, 0.2]
    along, off = float(d @ ax), d - (d @ ax) * ax
    assert abs(along - x) < 0.02 * x + 1e-4, (along, x)
    assert np.abs(off).max() < 1e-3, off                                           # neither across the axis nor down
    assert np.abs(st[3:6]).max() < 2e-3 and np.abs(st[10:13]).max() < 2e-2          # no rotation
    assert abs(float(st[7:10] @ ax) - v) < 0.02 * v + 1e-4
    if hasattr(w, 'w'):
        w.close()


def _qmat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize('backend', BACKENDS)
def test_prismatic_joint_between_two_free_bodies_conserves_momentum(backend):
    """A body sliding on ANOTHER free body (pybullet JOINT_PRISMATIC, child = a movable body), no gravity, no damping,
    nothing else in contact: the joint forces are internal, so the pair's linear momentum and its angular momentum about
    the origin must not change -- however far the slider has travelled along the axis.  (Round-4 advisor: the two linear
    rows acted at the parent's pivot on the parent but at the joint-frame ORIGIN on the child; along the slide axis the
    two points drift apart and equal and opposite impulses at different points are a spurious torque.  Both parties
    now take them at the parent's pivot, as Bullet's slider does.)"""
    w, cfg = T._world(backend, **{'PHYSICS.GRAVITY_Z': 0.0, 'PHYSICS.LINEAR_DAMPING': 0.0, 'PHYSICS.ANGULAR_DAMPING': 0.0, 'PHYSICS.SLEEP_STEPS': 0})
    m = (0.2, 0.35)
    # body 0: the slider, 8 cm above body 1 (the carrier), moving along the rail (x of the joint frame) and pushed sideways
    T._bodies(w, [(0, m[0], 0.5, (0.5, 0.0, 0.38), Q0, (0.4, 0.15, 0.0)), (0, m[1], 0.5, (0.5, 0.0, 0.30), Q0, (0.0, 0.0, 0.0))])
    w.set_constraint(0, [0.0, 0.0, 0.08, 0, 0, 0, 1], max_force=200.0, child=1, joint_type='prismatic')
    scene, _ = scenes.make_scene()
    ik = np.array(list(scene.shapes[0].inertia_k))

    def momenta():
        st = np.asarray(w.body_state())[0]
        P, L = np.zeros(3), np.zeros(3)
        for b in (0, 1):
            x, q, v, om = st[b, :3], st[b, 3:7], st[b, 7:10], st[b, 10:13]
            R = _qmat(q)
            I = R @ np.diag(m[b] * ik) @ R.T
            P += m[b] * v; L += np.cross(x, m[b] * v) + I @ om
        return P, L, st
    P0, L0, _ = momenta()
    w.step_sub(250)
    P1, L1, st = momenta()
    slid = float(st[0, 0] - st[1, 0])
    assert slid > 0.05, slid                                                       # the slider did travel along the rail
    tol = 2e-6 if backend == 'oracle64' else 2e-4
    assert np.abs(P1 - P0).max() < tol * 10, (P0, P1)
    assert np.abs(L1 - L0).max() < tol, (L0, L1, slid)                             # (was ~1e-2 with the two anchors)
    # ... and it is a joint: seen from the carrier (which the off-centre sideways push has turned) the slider is on the rail
    rel = _qmat(st[1, 3:7]).T @ (st[0, :3] - st[1, :3])
    assert abs(rel[1]) < 1.5e-3 and abs(rel[2] - 0.08) < 1.5e-3 and rel[0] > 0.05, rel
    if hasattr(w, 'w'):
        w.close()


@pytest.mark.parametrize('backend', BACKENDS)
def test_revolute_joint_is_a_hinge(backend):
    """'revolute' (the reference's JOINT_TYPES_MAPPING, bullet_physics.py:20-25): the body turns freely about the x axis of
    the joint frame and about nothing else, and its pivot stays put.  A box is hinged to the world about a horizontal axis
    (the world's x turned 25 degrees about z) through a pivot 3 cm off its centre of mass: under gravity it swings like a
    physical pendulum about that axis -- the angular velocity stays along the axis, the pivot within a millimetre of its
    world point, and at the bottom of the first swing the kinetic energy equals the potential energy it gave up (minus
    Bullet's damping: a few per cent)."""
    w, cfg = T._world(backend, **{'PHYSICS.SLEEP_STEPS': 0})
    m = 0.25
    x0 = np.array([0.5, 0.0, 0.35])
    T._bodies(w, [(0, m, 0.5, tuple(x0), Q0, (0, 0, 0))])
    a = np.radians(25.0)
    qz = [0, 0, np.sin(a / 2), np.cos(a / 2)]
    ax = np.array([np.cos(a), np.sin(a), 0.0])
    perp = np.array([-np.sin(a), np.cos(a), 0.0])
    lp = 0.03 * perp                                  # pivot in the body frame (identity orientation at the start): 3 cm sideways
    piv = x0 + lp
    w.set_constraint(0, list(piv) + qz, frame7=list(lp) + qz, max_force=200.0, joint_type='revolute')
    scene, _ = scenes.make_scene()
    ik = np.array(list(scene.shapes[0].inertia_k))
    z_min, ke_at_min = 1e9, 0.0
    for k in range(16):
        w.step_sub(25)
        st = np.asarray(w.body_state())[0, 0]
        R = _qmat(st[3:7])
        assert np.linalg.norm(st[:3] + R @ lp - piv) < 1e-3                                   # the pivot stays where it is
        om = st[10:13]
        assert np.linalg.norm(om - (om @ ax) * ax) < 0.03 * max(1.0, abs(om @ ax)), (k, om)   # it turns about the hinge axis only
        assert np.linalg.norm(R @ np.array([np.cos(a), np.sin(a), 0.0]) - ax) < 2e-3          # the axis itself does not tilt
        if st[2] < z_min:
            I = R @ np.diag(m * ik) @ R.T
            z_min, ke_at_min = st[2], 0.5 * m * st[7:10] @ st[7:10] + 0.5 * om @ I @ om
    drop = x0[2] - z_min
    assert 0.02 < drop <= 0.03 + 1e-3, drop                                                     # it swung down (to at most the arm's length)
    assert abs(ke_at_min - m * T.G * drop) < 0.12 * m * T.G * drop, (ke_at_min, m * T.G * drop)
    if hasattr(w, 'w'):
        w.close()


def _qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _attach_to_hand(w, n, link=7, drop=0.20, max_force=300.0, joint_type='fixed'):
    """Put body 0 of every env `drop` below frame `link` of the arm and tie it to that frame; returns the relative pose."""
    lp = np.asarray(w.link_poses(), np.float64)[:, link]
    st = np.asarray(w.body_state(), np.float64).copy()
    st[:, 0, :3] = lp[:, :3] + [0.0, 0.0, -drop]
    st[:, 0, 3:7] = [0, 0, 0, 1]; st[:, 0, 7:] = 0
    w.set_body_state(st)
    rel = []
    for i in range(1):          # (one constraint for every env of the world: the hand poses agree after the same reset command)
        Rh = _qmat(lp[i, 3:7])
        tp = Rh.T @ (st[i, 0, :3] - lp[i, :3])
        qh_inv = lp[i, 3:7] * [-1, -1, -1, 1]
        tq = _qmul(qh_inv, st[i, 0, 3:7])
        rel = list(tp) + list(tq)
    w.set_constraint(0, rel, max_force=max_force, child=abi.RV_CHILD_LINK(link), joint_type=joint_type)
    return np.array(rel)


@pytest.mark.parametrize('backend', BACKENDS)
def test_a_body_fixed_to_the_hand_link_follows_the_arm(backend):
    """A link of the arm as the other party of a constraint (bullet_physics.py:773-790: createConstraint with a
    (body, link) entity), e.g. an object attached to the hand: the box hangs 20 cm under `right_hand` (clear of the finger pads) by a fixed joint,
    the arm carries it through a 15 cm move, and the box stays where the joint says it is IN THE HAND FRAME (2 mm /
    5e-3 in the quaternion while moving: the link is kinematic, the rows see its twist); released, it falls."""
    w, cfg = T._world(backend)
    w.reset()
    rel = _attach_to_hand(w, 1)
    lp0 = np.asarray(w.link_poses(), np.float64)[0, 7]
    target = lp0.copy(); target[0] += 0.10; target[1] -= 0.08; target[2] += 0.08
    w.set_link_target(np.asarray(target, np.float32)[None])
    worst_p = worst_q = 0.0
    for k in range(12):
        w.step_sub(150)
        lp = np.asarray(w.link_poses(), np.float64)[0, 7]
        st = np.asarray(w.body_state(), np.float64)[0, 0]
        want_p = lp[:3] + _qmat(lp[3:7]) @ rel[:3]
        want_q = _qmul(lp[3:7], rel[3:7])
        worst_p = max(worst_p, float(np.linalg.norm(st[:3] - want_p)))
        worst_q = max(worst_q, float(min(np.abs(st[3:7] - want_q).max(), np.abs(st[3:7] + want_q).max())))
    moved = np.linalg.norm(np.asarray(w.link_poses(), np.float64)[0, 7, :3] - lp0[:3])
    assert moved > 0.12, moved
    assert worst_p < 2e-3 and worst_q < 5e-3, (worst_p, worst_q)
    z_held = float(np.asarray(w.body_state())[0, 0, 2])
    w.remove_constraint(0)
    w.step_sub(250)
    assert float(np.asarray(w.body_state())[0, 0, 2]) < z_held - 0.05          # released: it falls
    if hasattr(w, 'w'):
        w.close()


@pytest.mark.gpu
def test_hip_equals_oracle_with_a_body_tied_to_a_link():
    """HIP == float oracle bit for bit while the arm carries a body by a fixed joint to the hand and another swings from
    a point-to-point joint on a finger link -- through link targets, a push of the other bodies and a release."""
    from robovat_amd import lib
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(n_envs=32, seed=17, shape_names=names)
    w, ref = lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=False)
    w.reset(); ref.reset()
    rel = _attach_to_hand(T._Np(w), 32)
    _attach_to_hand(ref, 32)
    lp = ref.link_poses()[:, 7]
    for x in (w, ref):
        x.set_constraint(1, [0.0, 0.0, -0.06, 0, 0, 0, 1], frame7=[0.01, 0.0, 0.02, 0, 0, 0, 1], max_force=80.0, child=abi.RV_CHILD_LINK(8), joint_type='point2point')
    tgt = lp.copy(); tgt[:, 0] += 0.08; tgt[:, 2] -= 0.05
    for x in (w, ref):
        x.set_link_target(tgt.astype(np.float32))
    for k in range(4):
        w.step_sub(300); ref.step_sub(300)
        assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32)), k
        assert np.array_equal(w.joint_state().cpu().numpy(), ref.joint_state().astype(np.float32)), k
    a = ref.policy_random(3)
    w.set_actions(a); ref.set_actions(a); w.step_macro(); ref.step_macro()
    assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    assert np.array_equal(w.env_counters().cpu().numpy(), ref.env_counters())
    for x in (w, ref):
        x.remove_constraint(0)
    w.step_sub(200); ref.step_sub(200)
    assert np.array_equal(w.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    L = lib.load()
    import ctypes as C
    t7 = (C.c_float * 7)(0, 0, 0, 0, 0, 0, 1)
    assert L.rv_set_constraint_ex(w.h, 0, abi.RV_CHILD_LINK(7), 1, None, t7, 10.0) == abi.RV_ERR_NOTIMPL      # prismatic to a link
    assert L.rv_set_constraint_ex(w.h, 0, abi.RV_CHILD_LINK(abi.RV_NFRAME), 4, None, t7, 10.0) == abi.RV_ERR_VALUE
    w.close()
