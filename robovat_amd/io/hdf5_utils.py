"""Episode files with the reference's HDF5 layout (``robovat/io/hdf5_utils.py:16-95``,
``tools/run_env.py:229-247``): one group named by a UUID per episode; a dict is a group, a list
is a group ``<key>[]`` with one sub-group per element named by its index, ``None`` is the string
``'None'``, arrays of >= 100 elements are gzip-9 compressed datasets.

``write_data_to_hdf5`` / ``read_data_from_hdf5`` work on any h5py-like group (``create_group``,
``create_dataset``, ``__setitem__``, ``items``).  h5py is not part of this image: ``open_store``
returns an ``h5py.File`` when h5py is importable and otherwise an ``NpzStore`` -- the same tree of
groups kept in one ``.npz`` file whose keys are the HDF5 paths -- so that the layout can be written,
read back and tested here; files written with h5py are readable by the reference's own reader.

``episodes_from_rollout`` turns the ``[K, N, ...]`` buffers of ``World.rollout_record_full`` (per-step
observations, actions, rewards, dones, and the observation ``env.reset()`` returned before every step
that an auto-reset preceded) into the episode dictionaries ``generate_episode`` produces
(``episode_generation.py:36-67``), one per env and episode, without stepping the envs again and
without losing a transition.
"""
import socket
import time
import uuid

import numpy as np


def write_data_to_hdf5(f, data, compress_size_thresh=100):
    for key, value in data.items():
        if isinstance(value, dict):
            write_data_to_hdf5(f.create_group(key), value, compress_size_thresh)
        elif isinstance(value, list):
            group_list = f.create_group(key + '[]')
            for i, value_i in enumerate(value):
                if not isinstance(value_i, (dict, np.ndarray)):
                    raise ValueError("List '%s' has a %s element; lists hold dicts or numpy arrays." % (key, type(value_i)))
                group = group_list.create_group(str(i))
                write_data_to_hdf5(group, value_i if isinstance(value_i, dict) else {'value': value_i}, compress_size_thresh)
        elif value is None:
            f[key] = 'None'
        else:
            try:
                value = np.array(value)
                if value.dtype == object:
                    raise TypeError
            except Exception:
                raise ValueError("Unsupported data '%s' of type %s." % (key, type(value)))
            if np.prod(value.shape) >= compress_size_thresh:
                f.create_dataset(key, data=value, compression='gzip', compression_opts=9)
            else:
                f.create_dataset(key, data=value)


def _is_group(x):
    return hasattr(x, 'items') and hasattr(x, 'create_group')


def read_data_from_hdf5(f):
    data = dict()
    for key, value in f.items():
        if _is_group(value):
            if key[-2:] != '[]':
                data[key] = read_data_from_hdf5(value)
            else:
                out = [None] * len(value)
                for ind, element in value.items():
                    out[int(ind)] = read_data_from_hdf5(element)
                data[key[:-2]] = out
        else:
            value = value[()] if hasattr(value, 'shape') or hasattr(value, '__getitem__') else value
            if isinstance(value, bytes):
                value = value.decode()
            if isinstance(value, (str, np.str_)) and str(value) == 'None':
                data[key] = None
            else:
                value = np.array(value)
                data[key] = value.item() if value.shape == () else value
    return data


class _NpzGroup(object):
    """h5py-like group over a flat dict path -> array."""

    def __init__(self, store, prefix):
        self._store, self._prefix = store, prefix

    def _path(self, key):
        return self._prefix + '/' + key if self._prefix else key

    def create_group(self, key):
        path = self._path(key)
        self._store.setdefault('__groups__', set()).add(path)
        return _NpzGroup(self._store, path)

    def create_dataset(self, key, data=None, compression=None, compression_opts=None):
        self._store[self._path(key)] = np.asarray(data)

    def __setitem__(self, key, value):
        self._store[self._path(key)] = np.asarray(value)

    def items(self):
        n = len(self._prefix) + 1 if self._prefix else 0
        seen = {}
        for path in sorted(self._store.get('__groups__', ())):
            if path.startswith(self._prefix + '/' if self._prefix else '') and '/' not in path[n:]:
                seen[path[n:]] = _NpzGroup(self._store, path)
        for path, value in self._store.items():
            if path == '__groups__':
                continue
            if (not self._prefix or path.startswith(self._prefix + '/')) and '/' not in path[n:]:
                seen[path[n:]] = value
        return seen.items()

    def __len__(self):
        return len(dict(self.items()))


class NpzStore(_NpzGroup):
    """The group tree of an episode file in one .npz (fallback when h5py is absent)."""

    def __init__(self, filename, mode='a'):
        import os
        self.filename, self.mode = filename, mode
        store = {}
        if mode in ('a', 'r') and os.path.exists(filename):
            with np.load(filename, allow_pickle=False) as z:
                for k in z.files:
                    if k == '__groups__':
                        store['__groups__'] = set(str(x) for x in z[k])
                    else:
                        store[k] = z[k]
        _NpzGroup.__init__(self, store, '')

    def close(self):
        if self.mode == 'r':
            return                                  # nothing to write back
        import os
        out = {k: v for k, v in self._store.items() if k != '__groups__'}
        out['__groups__'] = np.array(sorted(self._store.get('__groups__', ())), dtype=str)
        tmp = self.filename + '.tmp.npz'            # (a crash while saving must not corrupt the episodes already on disk)
        np.savez_compressed(tmp, **out)
        os.replace(tmp, self.filename)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def open_store(filename, mode='a'):
    """``h5py.File`` if h5py is importable, else an ``NpzStore`` (filename + '.npz')."""
    try:
        import h5py
        return h5py.File(filename, mode)
    except ImportError:
        return NpzStore(filename if filename.endswith('.npz') else filename + '.npz', mode)


def append_episode(store, episode):
    """tools/run_env.py:241-244: a new group named by a UUID holds the episode."""
    name = str(uuid.uuid4())
    write_data_to_hdf5(store.create_group(name), episode)
    return name


def episodes_from_rollout(first_obs, obs, actions, rewards, dones, reset=None, reset_obs=None, auto_reset=True):
    """Episode dictionaries from ``World.rollout_record_full`` buffers.

    first_obs: observation dict of the state before the first step ([N, ...]); obs: dict of
    [K, N, ...] per-step observations; actions [K, N, ...]; rewards [K, N]; dones [K, N];
    reset [K, N] (1 = an auto-reset preceded step k) and reset_obs (dict of [K, N, ...]: what
    ``env.reset()`` returned then), as ``rollout_record_full`` gives them.  An env contributes one
    episode per ``done`` (its tail without a done is a last, open episode); every transition is
    (state before the step, action, reward, info) as in ``generate_episode``
    (episode_generation.py:47-67), the first state of an episode being the reset observation.
    Without ``reset_obs`` (plain ``rollout_record``) the first transition of every episode after
    an auto-reset has no recorded state and is dropped -- the number dropped is returned in the
    ``dropped_transitions`` attribute of the list.  ``reset`` and ``reset_obs`` come together
    (ValueError otherwise).  ``auto_reset=False`` says the rollout ran without auto-reset: the rows
    after an env's ``done`` are steps that were never taken (reward 0, done) and are neither
    transitions nor drops; with ``reset`` given the same is read off the flags (no reset flag
    after a done = the env stopped).
    """
    to_np = lambda x: x.cpu().numpy() if hasattr(x, 'cpu') else np.asarray(x)
    first_obs = {k: to_np(v) for k, v in first_obs.items()}
    obs = {k: to_np(v) for k, v in obs.items()}
    actions, rewards, dones = to_np(actions), to_np(rewards), to_np(dones)
    if (reset is None) != (reset_obs is None):
        raise ValueError('episodes_from_rollout: reset and reset_obs are given together or not at all')
    if reset_obs is not None:
        reset_obs = {k: to_np(v) for k, v in reset_obs.items()}
        reset = to_np(reset)
    K, N = rewards.shape
    host, stamp = socket.gethostname(), time.strftime('%Y-%m-%d-%H-%M-%S')

    class _Episodes(list):
        dropped_transitions = 0
    episodes = _Episodes()
    for i in range(N):
        state = {k: v[i] for k, v in first_obs.items()}
        transitions = []
        for k in range(K):
            if reset_obs is not None and reset[k, i]:
                state = {key: v[k, i] for key, v in reset_obs.items()}      # what env.reset() returned
            if state is None and (not auto_reset or reset_obs is not None):
                break               # episode over and no reset: the remaining rows are steps that were never taken
            if state is not None:
                transitions.append({'state': state, 'action': actions[k, i], 'reward': float(rewards[k, i]), 'info': None})
            else:
                episodes.dropped_transitions += 1
            state = {key: v[k, i] for key, v in obs.items()}
            if dones[k, i]:
                if transitions:
                    episodes.append({'hostname': host, 'timestamp': stamp, 'transitions': transitions, 'env': i})
                transitions, state = [], None       # the next step starts a new episode from a reset
        if transitions:
            episodes.append({'hostname': host, 'timestamp': stamp, 'transitions': transitions, 'env': i})
    return episodes
