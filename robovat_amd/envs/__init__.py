from robovat_amd.envs.push.push_env import PushEnv, VecPushEnv  # noqa: F401
