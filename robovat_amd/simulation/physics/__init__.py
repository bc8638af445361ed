from robovat_amd.simulation.physics.physics import Physics  # noqa: F401
from robovat_amd.simulation.physics.hip_physics import HipPhysics  # noqa: F401
