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
