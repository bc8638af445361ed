"""The C-ABI library loads and exports every symbol include/rovat.h declares;
without a GPU it refuses to run (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from robovat_amd import abi, configs, scenes, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'rovat.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rv_[a-z_0-9]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    handle = C.CDLL(lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(handle, name), name


def test_struct_sizes_match_c_layout():
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "rovat.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rv_shape), sizeof(rv_arm), sizeof(rv_scene), sizeof(rv_config), sizeof(rv_macro_stats), sizeof(rv_obs_buffers), sizeof(rv_state_view));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(src)
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 't.c'), '-o', os.path.join(d, 't')], check=True)
        out = subprocess.run([os.path.join(d, 't')], capture_output=True, text=True, check=True).stdout.split()
    sizes = [C.sizeof(x) for x in (abi.rv_shape, abi.rv_arm, abi.rv_scene, abi.rv_config, abi.rv_macro_stats,
                                    abi.rv_obs_buffers, abi.rv_state_view)]
    assert [int(x) for x in out] == sizes


def test_binary_carries_the_hash_of_its_sources():
    lib.build()
    assert lib.built_source_hash() == lib.source_hash() and len(lib.source_hash()) == 64


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    handle = lib.load()
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(n_envs=2, shape_names=names)
    out = C.c_void_p()
    rc = handle.rv_create(C.byref(cfg), C.byref(scene), 0, C.byref(out))
    assert rc == abi.RV_ERR_HIP and b'no CPU fallback' in handle.rv_last_error()
    with pytest.raises(RuntimeError):
        lib.check(rc)
