from robovat_amd.math.pose import Pose, Orientation, Euler, Quaternion, Point, get_transform  # noqa: F401
