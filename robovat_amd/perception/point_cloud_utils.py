"""Segmentation-id conversion and per-body grouping of a point cloud, with the
behaviour of ``robovat/perception/point_cloud_utils.py:23-39,110-157`` (pinned by
``tests/golden/camera_golden.json``: shapes, zero fill, with-replacement rule)."""
import numpy as np


def downsample(point_cloud, num_samples):
    """``num_samples`` points drawn from the cloud: without replacement when it has
    enough points, with replacement otherwise."""
    n = point_cloud.shape[0]
    inds = np.random.choice(np.arange(n), size=num_samples, replace=n < num_samples)
    return point_cloud[inds]


def convert_segment_ids(segmask, body_ids):
    """Body uids in a segmentation mask -> their index in ``body_ids``; -1 elsewhere."""
    segmask = np.asarray(segmask)
    out = np.full_like(segmask, -1)
    for i, uid in enumerate(body_ids):
        out[segmask == uid] = i
    return out


def group_by_labels(point_cloud, segmask, num_clusters, num_samples):
    """[num_clusters, num_samples, 3] float32: per label a downsampled copy of its points,
    zeros for labels without points."""
    out = np.zeros([num_clusters, num_samples, 3], dtype=np.float32)
    for i in range(num_clusters):
        inds = np.where(segmask == i)[0]
        if len(inds) > 0:
            out[i] = downsample(point_cloud[inds], num_samples)
    return out
