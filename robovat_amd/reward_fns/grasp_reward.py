"""GraspReward with the reference's API (``robovat/reward_fns/grasp_reward.py:14-80``).

On the backend the reward is computed on the device at the end of ``rv_step_macro`` (wait until
the graspable is stable, success = the arm still touches it; ``rv_dev_env.h: genv_step``); this
class reads it back and keeps the reference's success history.
"""
import numpy as np


class GraspReward(object):

    def __init__(self, name, end_effector_name=None, graspable_name=None, terminate_after_grasp=True,
                 streaming_length=1000):
        self.name = name
        self.end_effector_name, self.graspable_name = end_effector_name, graspable_name
        self.terminate_after_grasp = terminate_after_grasp
        self.streaming_length = streaming_length
        self.env = None
        self.history = []

    def initialize(self, env):
        self.env = env

    def on_episode_start(self):
        pass

    def get_reward(self):
        """(success, termination) of the grasp the env has just executed."""
        reward, done = self.env._vec.world.reward()
        success = bool(reward[0].item() > 0.5)
        self.history.append(success)
        self.history = self.history[-self.streaming_length:]
        return success, self.terminate_after_grasp

    @property
    def success_rate(self):
        return float(np.mean(self.history or [-1]))
