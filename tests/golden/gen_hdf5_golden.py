#!/usr/bin/env python
"""Golden for the episode writer: the REFERENCE's ``write_data_to_hdf5`` (robovat/io/hdf5_utils.py:16-54)
is imported from /root/reference (build container only; h5py is stubbed -- the function only needs a
group object) and run against a recording group on a deterministic episode of the generate_episode
shape (episode_generation.py:47-67).  The sequence of create_group / create_dataset / __setitem__
calls (path, shape, dtype, compression keywords, a checksum of the data) is the golden the build's
writer must reproduce call for call.

    python tests/golden/gen_hdf5_golden.py      # writes tests/golden/hdf5_golden.json
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def make_episode():
    """Deterministic stand-in for what generate_episode returns for PushEnv (push_env.py:169-236)."""
    rng = np.random.RandomState(12345)
    trans = []
    for t in range(3):
        state = {'position': rng.rand(4, 3).astype(np.float32), 'body_mask': np.ones(4, np.float32),
                 'point_cloud': rng.rand(4, 64, 3).astype(np.float32),
                 'num_episodes': np.int64(7), 'num_steps': np.int64(t), 'layout_id': np.int64(0),
                 'is_safe': np.int64(1), 'is_effective': np.int64(t % 2)}
        trans.append({'state': state, 'action': rng.uniform(-1, 1, 4).astype(np.float32), 'reward': float(t) - 0.5, 'info': None})
    return {'hostname': 'golden-host', 'timestamp': '2020-01-01-00-00-00', 'transitions': trans}


class Recorder(object):
    """h5py-like group that records what is done to it."""

    def __init__(self, calls, path=''):
        self.calls, self.path = calls, path

    def create_group(self, key):
        self.calls.append(['group', self.path + '/' + key])
        return Recorder(self.calls, self.path + '/' + key)

    @staticmethod
    def _desc(value):
        a = np.asarray(value)
        if a.dtype.kind in 'US':
            return {'str': str(value)}
        return {'shape': list(a.shape), 'dtype': str(a.dtype), 'sha1': hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()}

    def create_dataset(self, key, data=None, **kw):
        self.calls.append(['dataset', self.path + '/' + key, self._desc(data), {k: kw[k] for k in sorted(kw)}])

    def __setitem__(self, key, value):
        self.calls.append(['set', self.path + '/' + key, self._desc(value)])


def record(writer):
    calls = []
    writer(Recorder(calls), make_episode())
    return calls


if __name__ == '__main__':
    sys.path.insert(0, '/root/reference')
    sys.modules['h5py'] = types.ModuleType('h5py')       # only imported, never used by write_data_to_hdf5
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_hdf5_utils', '/root/reference/robovat/io/hdf5_utils.py')
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    calls = record(ref.write_data_to_hdf5)
    with open(os.path.join(HERE, 'hdf5_golden.json'), 'w') as f:
        json.dump({'source': 'robovat/io/hdf5_utils.py:16-54 write_data_to_hdf5 on gen_hdf5_golden.make_episode()', 'calls': calls}, f, indent=0)
    print('wrote hdf5_golden.json:', len(calls), 'calls')
