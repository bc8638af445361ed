"""Policies (``robovat/policies/*``): ``policy.action(observation) -> action``."""
import numpy as np

from robovat_amd import configs
from robovat_amd.envs.push.heuristic_push_sampler import HeuristicPushSampler


class Policy(object):
    def __init__(self, env, config=None):
        self.env = env
        self.config = config

    def action(self, observation):
        return self._action(observation)


class RandomPolicy(Policy):
    """``env.action_space.sample()`` (random_policy.py:14-23).  With a
    ``VecPushEnv`` the draw happens on the device (rv_policy_random)."""

    def _action(self, observation):
        if hasattr(self.env, 'sample_random_actions'):
            return self.env.sample_random_actions()
        return self.env.action_space.sample()


class HeuristicPushPolicy(Policy):
    """push_policy.py:12-52; batched envs use rv_policy_heuristic."""

    def __init__(self, env, config=None):
        config = config or configs.AttrDict(configs.HEURISTIC_PUSH_POLICY_CONFIG)
        super(HeuristicPushPolicy, self).__init__(env, config)
        self._sampler = HeuristicPushSampler(
            cspace_low=config.ACTION.CSPACE.LOW, cspace_high=config.ACTION.CSPACE.HIGH,
            translation_x=config.ACTION.MOTION.TRANSLATION_X, translation_y=config.ACTION.MOTION.TRANSLATION_Y,
            max_attemps=config.HEURISTICS.MAX_ATTEMPS)

    def _action(self, observation):
        if hasattr(self.env, 'sample_heuristic_actions'):
            return self.env.sample_heuristic_actions(self.config.HEURISTICS.MAX_ATTEMPS)
        return self._sampler.sample(np.asarray(observation['position']), np.asarray(observation['body_mask']),
                                    int(observation['num_episodes']), int(observation['num_steps']), num_samples=1)[0]
