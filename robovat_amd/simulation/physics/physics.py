"""Abstract physics plugin (``robovat/simulation/physics/physics.py:14-20``)."""


class Physics(object):
    """Base class resolved by name in ``Simulator(physics_backend=...)``."""

    def __init__(self, *args, **kwargs):
        pass
