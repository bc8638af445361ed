from robovat_amd.envs.grasp.grasp_2d import Grasp2D  # noqa: F401
from robovat_amd.envs.grasp.grasp_4dof_env import Grasp4DofEnv, VecGrasp4DofEnv  # noqa: F401
