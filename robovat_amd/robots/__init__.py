"""Robot facades (``robovat/robots``): ``RobotCommand`` and the simulated Sawyer."""
from robovat_amd.robots.robot_command import RobotCommand  # noqa: F401
from robovat_amd.robots.sawyer_sim import SawyerSim, factory  # noqa: F401
