"""Heuristic push sampler, host (numpy) edition.

Interface of ``robovat/envs/push/heuristic_push_sampler.py:16-168``.  The batched
device version is ``rv_policy_heuristic`` (one wave per env, 64 candidates per
round, lowest successful attempt wins, Philox draws); this class keeps the
reference's numpy-RNG behaviour for single-env users and golden tests.
"""
import numpy as np

SEED = 42


class HeuristicPushSampler(object):

    def __init__(self, cspace_low, cspace_high, translation_x, translation_y,
                 start_margin=0.05, motion_margin=0.01, max_attemps=20000):
        self.cspace_low = np.array(cspace_low)
        self.cspace_high = np.array(cspace_high)
        self.cspace_offset = 0.5 * (self.cspace_high + self.cspace_low)
        self.cspace_range = 0.5 * (self.cspace_high - self.cspace_low)
        self.translation_x = translation_x
        self.translation_y = translation_y
        self.start_margin = start_margin
        self.motion_margin = motion_margin
        self.max_attemps = max_attemps
        self.last_end = None

    def sample(self, position, body_mask, num_episodes, num_steps, num_samples=1):
        return np.stack([self._sample(position, body_mask, num_episodes, num_steps)
                         for _ in range(num_samples)], axis=0)

    def _sample(self, position, body_mask, num_episodes, num_steps):
        num_bodies = int(np.sum(body_mask))
        body_id = int(num_episodes) % num_bodies
        position = np.asarray(position)[:num_bodies]
        target = position[body_id:body_id + 1]
        base_angle = (num_episodes * SEED) % (2 * np.pi)
        if num_steps == 0:
            self.last_end = None
        start = motion = None
        for _ in range(self.max_attemps):
            start = np.random.uniform(-1., 1., [2])
            angle = base_angle + np.random.uniform(-0.25 * np.pi, 0.25 * np.pi)
            motion = np.array([np.cos(angle), np.sin(angle)], dtype=np.float32)
            motion = np.clip(motion + np.random.uniform(-0.3, 0.3, [2]), -1.0, 1.0)
            waypoints = self.get_waypoints(start, motion)
            if not self.is_waypoint_clear(waypoints[0], None, position, self.start_margin):
                continue
            if self.is_waypoint_clear(waypoints[0], waypoints[1], target, self.motion_margin):
                continue
            self.last_end = (np.asarray(waypoints[1]) - self.cspace_offset[:2]) / self.cspace_range[:2]
            break
        return np.concatenate([np.array(start, dtype=np.float32), np.array(motion, dtype=np.float32)], axis=-1)

    def get_waypoints(self, start, motion):
        motion = np.reshape(motion, [-1, 2])
        x = start[0] * self.cspace_range[0] + self.cspace_offset[0]
        y = start[1] * self.cspace_range[1] + self.cspace_offset[1]
        waypoints = [[x, y]]
        for i in range(motion.shape[0]):
            x = np.clip(x + motion[i, 0] * self.translation_x, self.cspace_low[0], self.cspace_high[0])
            y = np.clip(y + motion[i, 1] * self.translation_y, self.cspace_low[1], self.cspace_high[1])
            waypoints.append([x, y])
        return waypoints

    def is_waypoint_clear(self, waypoint1, waypoint2, position, margin):
        """Distance test against the segment ENDPOINTS only, as the reference does
        (heuristic_push_sampler.py:148-168)."""
        position = np.asarray(position)
        d1 = np.hypot(position[..., 0] - waypoint1[0], position[..., 1] - waypoint1[1])
        if waypoint2 is None:
            return bool(np.all(d1 > margin))
        d2 = np.hypot(position[..., 0] - waypoint2[0], position[..., 1] - waypoint2[1])
        return bool(np.all((d1 >= margin) & (d2 >= margin)))
