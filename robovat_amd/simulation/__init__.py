from robovat_amd.simulation.simulator import Simulator, Body, ControllableBody  # noqa: F401
