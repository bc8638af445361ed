"""The run-time PyBullet probe (SURVEY.md 8c last row): its status comes from an executed import, and the pose-parity
harness runs end to end -- here against a stand-in module that implements the handful of pybullet calls the harness makes
(free flight only: gravity + damping, no contacts), so that the harness' own code (scene building, state hand-over,
error statistics) is exercised on a machine without the wheel.  With the real wheel the same harness runs in
tests/test_gpu_pybullet_parity.py and bench.py."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import pybullet_parity  # noqa: E402


def test_the_status_is_what_the_import_did():
    pb, status = pybullet_parity.probe()
    if pb is None:
        assert 'raised' in status and ('ModuleNotFoundError' in status or 'Error' in status), status
    else:
        assert 'importable' in status and 'version' in status


def _free_flight_module():
    """the pybullet calls of tools/pybullet_parity.py on bodies that never touch anything"""
    m = types.ModuleType('pybullet')
    m.DIRECT, m.GEOM_BOX, m.GEOM_MESH, m.ACTIVATION_STATE_DISABLE_SLEEPING = 2, 3, 5, 2
    clients = {}

    def connect(mode):
        cid = len(clients)
        clients[cid] = {'bodies': [], 'dt': 1.0 / 240, 'g': np.zeros(3)}
        return cid
    m.connect = connect
    m.disconnect = lambda physicsClientId=0: clients.pop(physicsClientId)
    m.resetSimulation = lambda physicsClientId=0: None
    m.getAPIVersion = lambda: 'stand-in'

    def setTimeStep(dt, physicsClientId=0):
        clients[physicsClientId]['dt'] = dt
    m.setTimeStep = setTimeStep

    def setGravity(x, y, z, physicsClientId=0):
        clients[physicsClientId]['g'] = np.array([x, y, z])
    m.setGravity = setGravity
    m.createCollisionShape = lambda *a, **k: 0
    m.createCollisionShapeArray = lambda *a, **k: 0

    def createMultiBody(baseMass=0.0, baseCollisionShapeIndex=-1, basePosition=(0, 0, 0), physicsClientId=0):
        bodies = clients[physicsClientId]['bodies']
        bodies.append({'mass': baseMass, 'pos': np.array(basePosition, float), 'quat': np.array([0, 0, 0, 1.0]),
                       'lin': np.zeros(3), 'ang': np.zeros(3), 'damp': 0.04})
        return len(bodies) - 1
    m.createMultiBody = createMultiBody

    def changeDynamics(uid, link, physicsClientId=0, **kw):
        if 'linearDamping' in kw:
            clients[physicsClientId]['bodies'][uid]['damp'] = kw['linearDamping']
    m.changeDynamics = changeDynamics

    def resetBasePositionAndOrientation(uid, pos, quat, physicsClientId=0):
        b = clients[physicsClientId]['bodies'][uid]
        b['pos'], b['quat'] = np.array(pos, float), np.array(quat, float)
    m.resetBasePositionAndOrientation = resetBasePositionAndOrientation

    def resetBaseVelocity(uid, lin, ang, physicsClientId=0):
        b = clients[physicsClientId]['bodies'][uid]
        b['lin'], b['ang'] = np.array(lin, float), np.array(ang, float)
    m.resetBaseVelocity = resetBaseVelocity

    def stepSimulation(physicsClientId=0):
        c = clients[physicsClientId]
        for b in c['bodies']:
            if b['mass'] > 0:
                b['lin'] = (b['lin'] + c['dt'] * c['g']) * (1.0 - b['damp']) ** c['dt']
                b['pos'] = b['pos'] + c['dt'] * b['lin']
    m.stepSimulation = stepSimulation
    m.getBasePositionAndOrientation = lambda uid, physicsClientId=0: (tuple(clients[physicsClientId]['bodies'][uid]['pos']),
                                                                      tuple(clients[physicsClientId]['bodies'][uid]['quat']))
    m.getBaseVelocity = lambda uid, physicsClientId=0: (tuple(clients[physicsClientId]['bodies'][uid]['lin']),
                                                         tuple(clients[physicsClientId]['bodies'][uid]['ang']))
    return m


def test_the_harness_end_to_end_on_bodies_in_free_flight(scene_and_names):
    """bodies lifted 0.3 m above the table, not spinning: PyBullet stand-in (symplectic Euler + damping, which is what the
    oracle integrates too) and the FP64 oracle must agree to rounding over 100 substeps -- which checks that the harness hands
    the same state, masses, damping, gravity and time step to both sides and reads the poses back in the same layout"""
    from robovat_amd import configs
    from oracle import orc
    scene, names = scene_and_names
    cfg = configs.make_rv_config(n_envs=4, shape_names=names, seed=3)
    f64 = orc.OracleWorld(cfg, scene, double=True)
    f64.reset()
    state, params = f64.body_state(), f64.body_params()
    state[:, :, 2] += 0.3
    state[:, :, 7:13] = 0.0
    state[:, :, 7] = 0.2
    env_cfg = configs.push_env_config()
    out = pybullet_parity.pose_parity(_free_flight_module(), cfg, scene, state, params, {'f64_oracle': f64}, env_cfg=env_cfg)
    for h in (1, 10, 100):
        e = out['f64_oracle']['substeps_%d' % h]
        assert e['max_pos_m'] < 1e-7, (h, e)      # (the config's per-substep damping factor is rounded to float32)
        assert e['max_angle_rad'] < 1e-6, (h, e)
