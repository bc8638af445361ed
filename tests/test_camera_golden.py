"""Camera / point-cloud math against vectors generated from the reference
(tests/golden/gen_camera_golden.py), and the point-cloud ORACLE against the
reference-shaped pipeline render -> deproject_depth_image -> convert_segment_ids ->
group_by_labels (camera_obs.py:182-238)."""
import json
import os

import numpy as np
import pytest

from robovat_amd import abi, configs, scenes
from robovat_amd.perception import Camera, intrinsic_to_projection_matrix, point_cloud_utils as pcu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def golden():
    with open(os.path.join(HERE, 'golden', 'camera_golden.json')) as f:
        return json.load(f)


def test_projection_matrix(golden):
    for g in golden['projection']:
        m = intrinsic_to_projection_matrix(np.array(g['intrinsics']), g['height'], g['width'], g['near'], g['far'],
                                           upside_down=g['upside_down'])
        assert np.allclose(m, g['matrix'], rtol=1e-6, atol=1e-7)


def test_camera_pose_project_deproject(golden):
    for g in golden['cameras']:
        cam = Camera(height=g['height'], width=g['width'], intrinsics=g['intrinsics'],
                     translation=g['translation'], rotation=g['rotation'])
        pose = cam.pose
        assert np.allclose(pose.position, g['pose_position'], atol=1e-6)
        assert np.allclose(pose.matrix3, g['pose_matrix3'], atol=1e-6)
        assert np.array_equal(cam.project_point(np.array(g['points'])), np.array(g['pixels']))
        assert np.array_equal(cam.project_point(np.array(g['points_cam_frame']), is_world_frame=False),
                              np.array(g['pixels_cam_frame']))
        for px, z, want, want_c in zip(g['deproject_pixels'], g['deproject_depths'], g['deprojected'], g['deprojected_cam_frame']):
            assert np.allclose(cam.deproject_pixel(np.array(px), z), want, atol=1e-6)
            assert np.allclose(cam.deproject_pixel(np.array(px), z, is_world_frame=False), want_c, atol=1e-9)
        image = np.array(g['depth_image'], dtype=np.float32)
        assert np.allclose(cam.deproject_depth_image(image), g['cloud'], atol=1e-6)
        assert np.allclose(cam.deproject_depth_image(image, is_world_frame=False), g['cloud_cam_frame'], atol=1e-9)


def test_segment_ids_and_grouping_rules(golden):
    for g in golden['grouping']:
        cloud, seg = np.array(g['cloud']), np.array(g['segmask'], dtype=np.int32)
        conv = pcu.convert_segment_ids(seg, g['body_ids'])
        assert np.array_equal(conv, np.array(g['converted']))
        np.random.seed(0)
        grouped = pcu.group_by_labels(cloud, conv, 4, g['num_samples'])
        assert list(grouped.shape) == g['shape'] and str(grouped.dtype) == g['dtype']
        for i in range(4):
            n = g['counts'][i]
            assert bool((grouped[i] == 0).all()) == g['zero_rows'][i] == (n == 0)
            if n == 0:
                continue
            members = {tuple(r) for r in cloud[conv == i].astype(np.float32)}
            rows = [tuple(r) for r in grouped[i]]
            assert all(r in members for r in rows)                       # set membership
            # without replacement iff the label has at least num_samples points
            assert (len(set(rows)) == g['num_samples']) == (n >= g['num_samples'])
            assert (g['distinct_samples'][i] == g['num_samples']) == (n >= g['num_samples'])


def test_grasp2d_from_vector_and_4dof(golden):
    from robovat_amd.envs.grasp.grasp_2d import Grasp2D
    assert len(golden['grasp2d']) >= 20
    for g in golden['grasp2d']:
        cam = Camera(height=424, width=512, intrinsics=g['intrinsics'], translation=g['translation'], rotation=g['rotation'])
        gr = Grasp2D.from_vector(np.array(g['vector']), camera=cam)
        assert np.allclose(gr.center, g['center']) and abs(gr.angle - g['angle']) < 1e-12 and gr.depth == g['depth']
        assert abs(gr.width - g['width']) < 1e-9 and abs(gr.width_pixel - g['width_pixel']) < 1e-9
        assert np.allclose(gr.as_4dof(), g['as_4dof'], atol=1e-6)
        assert np.allclose(gr.vector, g['vector_back'], atol=1e-9)


def _oracle_world(n, seed, **over):
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n, seed=seed, shape_names=names)
    return orc.OracleWorld(cfg, scene), cfg


def _host_camera(cfg):
    fx, fy, cx, cy, sk = list(cfg.cam_intrinsics)
    return Camera(height=cfg.cam_height, width=cfg.cam_width, intrinsics=[[fx, sk, cx], [0, fy, cy], [0, 0, 1]],
                  translation=list(cfg.cam_translation), rotation=np.array(list(cfg.cam_rotation)).reshape(3, 3))


@pytest.mark.parametrize('over', [dict(), dict(TASK_NAME='crossing', LAYOUT_ID=0, MOVABLE_NAME='CONCAVE', MIN_MOVABLE_BODIES=2)])
def test_point_cloud_oracle_vs_reference_shaped_pipeline(over):
    """Every point the oracle emits for body b is the deprojection of a pixel that the
    render labels b; with >= P such pixels the P points are distinct pixels, with fewer
    they are draws with replacement, with none the row is zero."""
    w, cfg = _oracle_world(6, seed=4, **over)
    w.reset()
    cloud = w.point_cloud()
    P = int(cfg.num_points)
    cam = _host_camera(cfg)
    mask = w.observe()[1]
    seen_small = seen_big = 0
    for i in range(w.n):
        depth, seg = w.render(i)
        pts = cam.deproject_depth_image(depth)                       # [H*W, 3], row-major
        labels = pcu.convert_segment_ids(seg.reshape(-1).astype(np.int32), list(range(abi.RV_MAXB)))
        for b in range(abi.RV_MAXB):
            members = pts[labels == b]
            got = cloud[i, b].astype(np.float64)
            if mask[i, b] == 0 or len(members) == 0:
                assert (got == 0).all()
                continue
            # membership: nearest member within float32 rounding of the device arithmetic
            d = np.abs(got[:, None, :] - members[None, :, :]).max(-1)
            assert d.min(axis=1).max() < 2e-6, d.min(axis=1).max()
            idx = d.argmin(axis=1)
            if len(members) >= P:
                assert len(set(idx.tolist())) == P; seen_big += 1
                # emitted in the order of the per-pixel keys (np.random.choice returns a random permutation,
                # point_cloud_utils.py:23-39), not in scan order: no slice of the cloud is a spatial slice
                assert not (np.diff(idx) > 0).all()
                half = idx[:P // 2]
                assert half.min() < len(members) * 0.25 and half.max() > len(members) * 0.75
            else:
                assert set(idx.tolist()) <= set(range(len(members))); seen_small += 1
            # centroid / extent statistics of the sample vs the full visible set
            assert np.abs(got.mean(0) - members.mean(0)).max() < 0.01
            assert (got.min(0) >= members.min(0) - 2e-6).all() and (got.max(0) <= members.max(0) + 2e-6).all()
    assert seen_big > 0
