import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
