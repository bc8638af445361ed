from robovat_amd.envs.push.push_env import PushEnv, VecPushEnv  # noqa: F401
from robovat_amd.envs.grasp import Grasp4DofEnv, VecGrasp4DofEnv, Grasp2D  # noqa: F401
