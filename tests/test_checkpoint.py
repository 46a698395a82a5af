"""Checkpoint readers of the library (pcv_checkpoint_visit, pcv_model_create_from_dir): the reference's
`rust_model.ot` (configs.rs:109,112; VarStore::load at model.rs:117-124), `pytorch_model.bin` and
`model.safetensors`.  The `.ot` fixture under tests/golden/ot/ was written by libtorch's OutputArchive — the call
tch's Tensor::save_multi makes — see tests/golden/gen_ot_fixture.py; no file from the reference's own pipeline
exists offline."""
import json
import os
import zipfile

import numpy as np
import pytest

import perceive_amd as pa


@pytest.fixture(scope="module")
def ot_dir(golden_dir):
    return os.path.join(golden_dir, "ot")


@pytest.fixture(scope="module")
def expected(ot_dir):
    return json.load(open(os.path.join(ot_dir, "expected.json")))


def _matches(arr, want):
    flat = arr.ravel()
    return (list(arr.shape) == want["shape"] and abs(float(flat.astype(np.float64).sum()) - want["sum"]) <= 1e-6 * max(1.0, abs(want["sum"]))
            and np.array_equal(flat[:3], np.float32(want["head"])) and np.array_equal(flat[-3:], np.float32(want["tail"])))


def test_ot_archive_tensors(ot_dir, expected):
    got = pa.checkpoint_tensors(os.path.join(ot_dir, "rust_model.ot"))
    assert list(got) == list(expected["tensors"])  # file order = the order save_multi was given
    for name, want in expected["tensors"].items():
        assert _matches(got[name], want), name
    dense = pa.checkpoint_tensors(os.path.join(ot_dir, "2_Dense", "rust_model.ot"))
    assert list(dense) == ["linear.weight", "linear.bias"]
    for name, want in expected["dense"].items():
        assert _matches(dense[name], want), name


def test_state_dict_bin_and_safetensors_agree(tmp_path):
    import torch
    from safetensors.torch import save_file

    torch.manual_seed(0)
    sd = {"w": torch.randn(5, 7), "h": torch.randn(4, 3).half(), "b": torch.randn(6).bfloat16(), "d": torch.randn(3, 2).double(),
          "scalar": torch.tensor(2.5), "ids": torch.arange(8).reshape(1, 8), "flag": torch.tensor([True, False])}
    view = torch.arange(24.0).reshape(4, 6).t()[1:4]  # offset + strides: not contiguous
    torch.save({**sd, "view": view}, tmp_path / "pytorch_model.bin")
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    b = pa.checkpoint_tensors(tmp_path / "pytorch_model.bin")
    s = pa.checkpoint_tensors(tmp_path / "model.safetensors")
    for k, v in sd.items():
        if v.dtype in (torch.int64, torch.bool):
            assert b[k] is None and s[k] is None  # reported, not converted
            continue
        want = v.float().numpy()
        assert b[k].shape == want.shape and s[k].shape == want.shape
        np.testing.assert_array_equal(b[k], want)
        np.testing.assert_array_equal(s[k], want)
    np.testing.assert_array_equal(b["view"], view.numpy())
    # nn.Module state dicts carry an OrderedDict with _metadata; {"state_dict": ...} wrappers are unwrapped
    lin = torch.nn.Linear(3, 2)
    torch.save(lin.state_dict(), tmp_path / "lin.bin")
    torch.save({"state_dict": lin.state_dict(), "epoch": 3}, tmp_path / "wrapped.bin")
    for name in ("lin.bin", "wrapped.bin"):
        t = pa.checkpoint_tensors(tmp_path / name)
        np.testing.assert_array_equal(t["weight"], lin.weight.detach().numpy())
        np.testing.assert_array_equal(t["bias"], lin.bias.detach().numpy())


def test_archive_reader_refuses_what_it_does_not_know(tmp_path, ot_dir):

    # a pickle that wants to call something: refused, never executed
    marker = tmp_path / "executed"
    evil = b"\x80\x02cos\nsystem\nX" + len(f"touch {marker}").to_bytes(4, "little") + f"touch {marker}".encode() + b"\x85R."
    with zipfile.ZipFile(tmp_path / "evil.bin", "w", zipfile.ZIP_STORED) as z:
        z.writestr("archive/data.pkl", evil)
        z.writestr("archive/version", "3\n")
    with pytest.raises(pa.ModelError, match="does not know"):
        pa.checkpoint_tensors(tmp_path / "evil.bin")
    assert not marker.exists()
    # truncated archive, not-an-archive, missing storage, compressed data.pkl
    raw = open(os.path.join(ot_dir, "2_Dense", "rust_model.ot"), "rb").read()
    (tmp_path / "cut.ot").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(pa.ModelError):
        pa.checkpoint_tensors(tmp_path / "cut.ot")
    (tmp_path / "junk.ot").write_bytes(b"PK\x03\x04" + b"\0" * 100)
    with pytest.raises(pa.ModelError):
        pa.checkpoint_tensors(tmp_path / "junk.ot")
    with zipfile.ZipFile(os.path.join(ot_dir, "2_Dense", "rust_model.ot")) as src, zipfile.ZipFile(tmp_path / "nostorage.ot", "w") as dst:
        for info in src.infolist():
            if not info.filename.endswith("/data/0"):
                dst.writestr(info.filename, src.read(info.filename), zipfile.ZIP_STORED)
    with pytest.raises(pa.ModelError, match="missing"):
        pa.checkpoint_tensors(tmp_path / "nostorage.ot")
    with zipfile.ZipFile(os.path.join(ot_dir, "2_Dense", "rust_model.ot")) as src, zipfile.ZipFile(tmp_path / "deflated.ot", "w") as dst:
        for info in src.infolist():
            dst.writestr(info.filename, src.read(info.filename), zipfile.ZIP_DEFLATED)
    with pytest.raises(pa.ModelError, match="compressed"):
        pa.checkpoint_tensors(tmp_path / "deflated.ot")
    with pytest.raises(pa.ModelError):
        pa.checkpoint_tensors(tmp_path / "does-not-exist.ot")


def _patched_state_dict(tmp_path, name, size, stride):
    """torch.save of {"w": 16 f32} with the (size, stride) tuples of its view record — pickled `K\x10\x85 q<memo> K\x01\x85` —
    replaced by the given pickle bytes (views no tensor API would make)."""
    import re

    import torch

    torch.save({"w": torch.arange(16, dtype=torch.float32)}, tmp_path / "plain.bin")
    with zipfile.ZipFile(tmp_path / "plain.bin") as src, zipfile.ZipFile(tmp_path / name, "w", zipfile.ZIP_STORED) as dst:
        for info in src.infolist():
            data = src.read(info.filename)
            if info.filename.endswith("data.pkl"):
                pat = re.compile(rb"K\x10\x85(q.)K\x01\x85", re.S)
                assert len(pat.findall(data)) == 1, data
                data = pat.sub(lambda m: size + m.group(1) + stride, data)
            dst.writestr(info.filename, data)
    return tmp_path / name


def test_archive_reader_refuses_views_that_leave_their_storage(tmp_path):
    """Shapes, strides and offsets come straight out of the pickle / the JSON header: sums and products that wrap around
    int64 (stride 2^62 + 2^24 over three elements wraps `last` negative and walked far outside the mapping),
    broadcast views that would allocate what the file never held, and safetensors shapes that are not whole numbers."""
    tup1 = lambda v: (b"K" + bytes([v]) if 0 <= v < 256 else b"\x8a\x08" + int(v).to_bytes(8, "little", signed=True)) + b"\x85"
    assert pa.checkpoint_tensors(_patched_state_dict(tmp_path, "same.bin", tup1(16), tup1(1)))["w"].tolist() == list(range(16))
    assert pa.checkpoint_tensors(_patched_state_dict(tmp_path, "strided.bin", tup1(4), tup1(5)))["w"].tolist() == [0, 5, 10, 15]
    for name, size, stride in [("wrap.bin", 3, 2**62 + 2**24),  # `last` wraps negative
                               ("bcast.bin", 2**40, 0),         # 4 TB of values out of 64 bytes
                               ("huge.bin", 2**62, 0),
                               ("past.bin", 2, 16),             # second element is one past the end
                               ("neg.bin", 4, -1)]:
        with pytest.raises(pa.ModelError):
            pa.checkpoint_tensors(_patched_state_dict(tmp_path, name, tup1(size), tup1(stride)))
    # safetensors: header numbers are JSON doubles
    import json
    import struct

    def st_file(name, shape, offsets, nbytes=64):
        hdr = json.dumps({"w": {"dtype": "F32", "shape": shape, "data_offsets": offsets}}).encode()
        (tmp_path / name).write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * nbytes)
        return tmp_path / name

    assert pa.checkpoint_tensors(st_file("ok.safetensors", [4, 4], [0, 64]))["w"].shape == (4, 4)
    for name, shape, offs in [("frac.safetensors", [2.5, 4], [0, 40]), ("neg.safetensors", [-1, 4], [0, 64]),
                              ("wrap.safetensors", [2**31, 2**31, 4], [0, 64]), ("big.safetensors", [1e300], [0, 64]),
                              ("offs.safetensors", [16], [1e300, 64]), ("past.safetensors", [32], [0, 128])]:
        with pytest.raises(pa.ModelError):
            pa.checkpoint_tensors(st_file(name, shape, offs))


@pytest.mark.gpu
def test_new_pretrained_from_rust_model_ot(ctx, ot_dir, expected):
    """The reference's layout end to end: JSON configs + vocab.txt + rust_model.ot (+ 2_Dense/rust_model.ot) -> text in,
    embedding out, against Hugging Face running the same weights (values stored by the generator)."""
    m = pa.new_pretrained(ctx, ot_dir)
    assert m.desc.dense_out == 64 and m.desc.max_seq_length == 24
    out = m.encode(expected["texts"])
    ref = np.asarray(expected["embeddings"], dtype=np.float32)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 1e-4
    m.close()


@pytest.mark.gpu
def test_ot_wins_over_other_formats_and_missing_tensors_are_named(ctx, ot_dir, tmp_path, expected):
    import shutil

    from safetensors.numpy import save_file

    d = tmp_path / "m"
    shutil.copytree(ot_dir, d)
    # a safetensors file with wrong values next to rust_model.ot: the reference's file is the one read
    good = pa.checkpoint_tensors(d / "rust_model.ot")
    save_file({k: np.zeros_like(v) for k, v in good.items()}, str(d / "model.safetensors"))
    m = pa.new_pretrained(ctx, str(d))
    assert np.abs(m.encode(expected["texts"]) - np.asarray(expected["embeddings"], dtype=np.float32)).max() < 1e-4
    m.close()
    # without the .ot the safetensors is read; one tensor short -> the error names it
    os.remove(d / "rust_model.ot")
    save_file({k: v for k, v in good.items() if "output.LayerNorm.bias" not in k or "attention" in k}, str(d / "model.safetensors"))
    with pytest.raises(pa.ModelError, match="output.LayerNorm.bias"):
        pa.new_pretrained(ctx, str(d))
