"""Heuristic push sampler, host (numpy) edition.

Same interface and the same stream of ``np.random`` draws per attempt as the
reference class (``robovat/envs/push/heuristic_push_sampler.py:16-168``: start
xy, angle jitter, motion noise -- in that order), so a seeded reference run and
this one agree (tests/golden/heuristic_golden.json).  The batched device version
is ``rv_policy_heuristic`` (one wave per env, 64 candidates per round, lowest
successful attempt wins, Philox draws); this class serves single-env users.

A candidate push is accepted when its start point keeps ``start_margin`` from
every body and one of its two end points comes within ``motion_margin`` of the
target body (the body picked by ``num_episodes``).
"""
import numpy as np

SEED = 42   # multiplies the episode index into the base push direction


def _dist(points, xy):
    points = np.asarray(points, dtype=np.float64)
    return np.hypot(points[..., 0] - xy[0], points[..., 1] - xy[1])


class HeuristicPushSampler(object):

    def __init__(self, cspace_low, cspace_high, translation_x, translation_y,
                 start_margin=0.05, motion_margin=0.01, max_attemps=20000):
        lo, hi = np.array(cspace_low), np.array(cspace_high)
        self.cspace_low, self.cspace_high = lo, hi
        self.cspace_offset, self.cspace_range = 0.5 * (hi + lo), 0.5 * (hi - lo)
        self.translation_x, self.translation_y = translation_x, translation_y
        self.start_margin, self.motion_margin = start_margin, motion_margin
        self.max_attemps = max_attemps
        self.last_end = None      # normalised end point of the last accepted push

    # -- public interface -------------------------------------------------
    def sample(self, position, body_mask, num_episodes, num_steps, num_samples=1):
        draws = [self._sample(position, body_mask, num_episodes, num_steps) for _ in range(num_samples)]
        return np.stack(draws, axis=0)

    def get_waypoints(self, start, motion):
        """Start point and the clipped end point of every motion segment (world xy)."""
        pts = [self._to_world(start)]
        for seg in np.reshape(motion, [-1, 2]):
            step = np.array([seg[0] * self.translation_x, seg[1] * self.translation_y])
            nxt = np.clip(np.asarray(pts[-1]) + step, self.cspace_low[:2], self.cspace_high[:2])
            pts.append([nxt[0], nxt[1]])
        return pts

    def is_waypoint_clear(self, waypoint1, waypoint2, position, margin):
        """True when the bodies keep ``margin`` from the END POINTS of the segment (the
        reference tests only those, heuristic_push_sampler.py:148-168):
        strictly for a lone point, non-strictly for a pair."""
        if waypoint2 is None:
            return bool(np.all(_dist(position, waypoint1) > margin))
        return bool(np.all(np.minimum(_dist(position, waypoint1), _dist(position, waypoint2)) >= margin))

    # -- internals --------------------------------------------------------
    def _to_world(self, unit_xy):
        return [unit_xy[0] * self.cspace_range[0] + self.cspace_offset[0],
                unit_xy[1] * self.cspace_range[1] + self.cspace_offset[1]]

    def _to_unit(self, world_xy):
        return (np.asarray(world_xy) - self.cspace_offset[:2]) / self.cspace_range[:2]

    @staticmethod
    def _propose(base_angle):
        """One candidate: three np.random calls, in the reference's order."""
        start = np.random.uniform(-1., 1., [2])
        heading = base_angle + np.random.uniform(-0.25 * np.pi, 0.25 * np.pi)
        push = np.array([np.cos(heading), np.sin(heading)], dtype=np.float32)
        push = np.clip(push + np.random.uniform(-0.3, 0.3, [2]), -1.0, 1.0)
        return start, push

    def _sample(self, position, body_mask, num_episodes, num_steps):
        count = int(np.sum(body_mask))
        bodies = np.asarray(position)[:count]
        target = bodies[int(num_episodes) % count][None]
        base_angle = (num_episodes * SEED) % (2 * np.pi)
        if num_steps == 0:
            self.last_end = None
        start = push = None
        for _ in range(self.max_attemps):
            start, push = self._propose(base_angle)
            begin, end = self.get_waypoints(start, push)[:2]
            starts_free = self.is_waypoint_clear(begin, None, bodies, self.start_margin)
            reaches_target = not self.is_waypoint_clear(begin, end, target, self.motion_margin)
            if starts_free and reaches_target:
                self.last_end = self._to_unit(end)
                break
        return np.concatenate([np.asarray(start, np.float32), np.asarray(push, np.float32)], axis=-1)
