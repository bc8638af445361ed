import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The CPU checkers (oracle, lane emulator) are OpenMP programs: one thread per hardware thread of the NODE would be throttled
# into the ground inside a container that is a slice of it (a 16-CPU cgroup quota on a 256-thread box) -- as many threads as
# this process may keep busy, set before libgomp is loaded
from oracle import orc as _orc  # noqa: E402
os.environ.setdefault('OMP_NUM_THREADS', str(_orc.effective_cpus()))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def scene_and_names():
    from robovat_amd import scenes
    return scenes.make_scene()


@pytest.fixture(scope='session')
def hip_lib():
    """The product library; GPU tests fail loudly if it is missing."""
    from robovat_amd import lib
    return lib.load()
