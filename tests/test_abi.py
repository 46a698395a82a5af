"""CPU: the C-ABI library loads and exports every symbol include/perceive_hip.h declares; host-only
entry points behave; device entry points fail loudly (no fallback) when no GPU is visible."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import perceive_amd as pa
from perceive_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "perceive_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    names = declared_symbols()
    assert len(names) >= 40
    out = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (pcv_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in perceive_hip.h but not exported: {missing}"


def test_python_binding_covers_header():
    assert sorted(_ffi.SYMBOLS) == declared_symbols()
    _ffi.lib()  # binds every symbol; AttributeError if one is missing


def test_no_torch_or_oracle_linked():
    out = subprocess.run(["ldd", _ffi.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in out and "oracle" not in out
    assert "libamdhip64" in out


def test_blob_codec_host_only():
    v = np.array([1.0, -2.5, 0.0, 3.4028235e38], np.float32)
    blob = pa.serialize_embedding(v)
    assert blob == struct.pack("<4f", *v)
    np.testing.assert_array_equal(pa.deserialize_embedding(blob), v)
    with pytest.raises(pa.PcvError) as e:
        pa.deserialize_embedding(b"\x00\x01\x02")  # reference: chunk[3] out of bounds -> panic
    assert "whole number of f32" in str(e.value)


def test_minilm_desc():
    d = _ffi.ModelDesc()
    _ffi.lib().pcv_model_desc_minilm_l6(d)
    assert (d.vocab_size, d.hidden, d.layers, d.heads, d.intermediate) == (30522, 384, 6, 12, 1536)
    assert d.normalize == 1 and d.pooling == _ffi.POOL_MEAN and abs(d.layer_norm_eps - 1e-12) < 1e-18


def test_fails_loudly_without_gpu():
    if pa.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pa.PcvError) as e:
        pa.Context(0)
    assert e.value.status == 2 and "no HIP device" in str(e.value)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "perceive_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle_ffi" not in src and "liboracle" not in src and '"oracle.h"' not in src, f
