"""CPU: on-disk formats and optimizer-state layout either side of the hot path: HDF5 wrappers
(on h5py or, as in this image, on the HDF5 C library itself; a stand-in module only where neither exists), scaler statistics, pickled model.conf, Adam state in torch.optim.Adam's layout, the
weighted gradient-exchange protocol."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

from qpnet_amd import loaders
from qpnet_amd.config import TINY

STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def _real_hdf5_backend():
    """h5py, or qpnet_amd._hdf5 on an installed libhdf5 (what loaders._h5py() picks), or None."""
    try:
        import h5py
        if "stubs" not in getattr(h5py, "__file__", ""):
            return h5py
    except ImportError:
        pass
    try:
        from qpnet_amd import _hdf5
        _hdf5._lib()
        return _hdf5
    except ImportError:
        return None


@pytest.fixture
def h5stub(monkeypatch):
    """the wrappers run on real HDF5 wherever h5py or the HDF5 C library exists (this image has the library); only otherwise on the pickle stand-in"""
    if _real_hdf5_backend() is None:
        monkeypatch.syspath_prepend(STUBS)
        monkeypatch.delitem(sys.modules, "h5py", raising=False)
    yield
    sys.modules.pop("h5py", None) if "stubs" in getattr(sys.modules.get("h5py"), "__file__", "") else None


def _hdf5_tool(name):
    import shutil
    for d in (None, "/opt/conda/bin", "/usr/bin", "/usr/local/bin"):
        p = shutil.which(name) if d is None else os.path.join(d, name)
        if p and os.path.exists(p):
            return p
    return None


def test_hdf5_files_meet_the_hdf5_projects_own_tools(tmp_path):
    """The /world wrappers against REAL HDF5 without h5py: a file loaders.write_hdf5 wrote (through libhdf5, qpnet_amd/_hdf5.py) is listed by h5ls and
    its data dumped by h5dump bit for bit -- groups created on the way for /world/mean as h5py does --, and a file h5import built from text, i.e. one
    this repo's code never touched, is read by loaders.read_hdf5 / shape_hdf5 / read_features (reference utils.py:43-128)."""
    import subprocess
    backend = _real_hdf5_backend()
    h5dump, h5ls, h5import = _hdf5_tool("h5dump"), _hdf5_tool("h5ls"), _hdf5_tool("h5import")
    if backend is None or not (h5dump and h5ls and h5import):
        pytest.skip("needs h5py or libhdf5, and the HDF5 command-line tools")
    assert loaders._h5py() is backend
    w = np.random.RandomState(2).randn(41, 39)
    f = str(tmp_path / "mine.h5")
    loaders.write_hdf5(f, "/world", w.astype(np.float32))
    loaders.write_hdf5(f, "/world_stats/mean", w.mean(0))
    loaders.write_hdf5(f, "/world_stats/scale", w.std(0))
    listing = subprocess.run([h5ls, "-r", f], capture_output=True, text=True, check=True).stdout
    assert "/world " in listing and "Dataset {41, 39}" in listing and "/world_stats/scale" in listing and "Group" in listing
    for path, ref in (("/world", w.astype(np.float32)), ("/world_stats/mean", w.mean(0))):
        raw = str(tmp_path / "raw.bin")
        subprocess.run([h5dump, "-d", path, "-b", "LE", "-o", raw, f], capture_output=True, check=True)
        np.testing.assert_array_equal(np.fromfile(raw, dtype=ref.dtype.newbyteorder("<")).reshape(ref.shape), ref)
    txt = str(tmp_path / "w.txt")
    np.savetxt(txt, w, fmt="%.17g")
    g = str(tmp_path / "imported.h5")
    subprocess.run([h5import, txt, "-dims", "41,39", "-path", "/world", "-type", "TEXTFP", "-size", "64", "-o", g], capture_output=True, check=True)
    np.testing.assert_array_equal(loaders.read_hdf5(g, "/world"), w)
    np.testing.assert_array_equal(loaders.read_features(g), w)
    assert tuple(loaders.shape_hdf5(g, "/world")) == w.shape
    with pytest.raises(KeyError):
        loaders.read_hdf5(g, "/world/mean")


def test_hdf5_roundtrip_and_errors(h5stub, tmp_path):
    f = str(tmp_path / "sub" / "a.h5")
    w = np.arange(12, dtype=np.float32).reshape(4, 3)
    loaders.write_hdf5(f, "/world", w)
    np.testing.assert_array_equal(loaders.read_hdf5(f, "/world"), w)
    loaders.write_hdf5(f, "/world", w * 2)                                  # overwrite is the default (utils.py:94-98)
    np.testing.assert_array_equal(loaders.read_hdf5(f, "/world"), w * 2)
    with pytest.raises(KeyError):
        loaders.write_hdf5(f, "/world", w, is_overwrite=False)
    with pytest.raises(KeyError):
        loaders.read_hdf5(f, "/nothing")
    with pytest.raises(FileNotFoundError):
        loaders.read_hdf5(str(tmp_path / "missing.h5"), "/world")
    np.testing.assert_array_equal(loaders.read_features(f), w * 2)
    np.save(str(tmp_path / "b.npy"), w)
    np.testing.assert_array_equal(loaders.read_features(str(tmp_path / "b.npy")), w)


def test_hdf5_against_a_real_h5py(tmp_path):
    """On any box that HAS h5py (this image does not: skipped here) the /world wrappers meet real HDF5 files: one written by h5py the way
    the reference's feature extraction writes it (utils.py:90-128: a float dataset under /world) is read by loaders.read_hdf5 / shape_hdf5,
    and one written by loaders.write_hdf5 is opened by h5py."""
    h5py = pytest.importorskip("h5py")
    assert "stubs" not in getattr(h5py, "__file__", "")
    w = np.random.RandomState(1).randn(57, 39).astype(np.float32)
    f = str(tmp_path / "real.h5")
    with h5py.File(f, "w") as fh:
        fh.create_dataset("/world", data=w)
        fh.create_dataset("/world_stats/mean", data=w.mean(0))
    np.testing.assert_array_equal(loaders.read_hdf5(f, "/world"), w)
    np.testing.assert_array_equal(loaders.read_features(f), w)
    assert tuple(loaders.shape_hdf5(f, "/world")) == w.shape
    g = str(tmp_path / "mine.h5")
    loaders.write_hdf5(g, "/world", w * 3)
    with h5py.File(g, "r") as fh:
        np.testing.assert_array_equal(fh["/world"][()], w * 3)


def test_scaler_stats_like_calc_stats(h5stub, tmp_path):
    """calc_stats.py:19-37: StandardScaler.partial_fit over dims 1.., uv dim keeps mean 0 / scale 1; read back the way
    the task scripts rebuild their scaler (qpnet_train.py:433-436)."""
    from sklearn.preprocessing import StandardScaler
    rs = np.random.RandomState(0)
    feats = [rs.randn(30 + 7 * i, 39) * 3 + 1 for i in range(4)]
    sc = StandardScaler()
    for f in feats:
        sc.partial_fit(f[:, 1:])
    mine = loaders.calc_stats(feats)
    np.testing.assert_allclose(mine.mean_[1:], sc.mean_, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(mine.scale_[1:], sc.scale_, rtol=1e-12)
    assert mine.mean_[0] == 0.0 and mine.scale_[0] == 1.0
    stats = str(tmp_path / "stats.h5")
    loaders.write_hdf5(stats, "/world/mean", mine.mean_)
    loaders.write_hdf5(stats, "/world/scale", mine.scale_)
    back = loaders.read_scaler_stats(stats)
    ref = StandardScaler(); ref.mean_ = mine.mean_; ref.scale_ = mine.scale_; ref.n_features_in_ = 39
    np.testing.assert_allclose(back.transform(feats[0]), ref.transform(feats[0]), rtol=1e-14)
    np.savez(str(tmp_path / "stats.npz"), mean=mine.mean_, scale=mine.scale_)
    np.testing.assert_array_equal(loaders.read_scaler_stats(str(tmp_path / "stats.npz")).scale_, mine.scale_)


def test_model_conf_is_a_pickled_namespace(tmp_path):
    """the trainer torch.save()s its argparse.Namespace (qpnet_train.py:389); decode torch.load()s it (qpnet_decode.py:245)"""
    ns = argparse.Namespace(feature_type="world", feature_format="h5", dense_factor=8, **TINY.kwargs())
    path = loaders.save_model_conf(str(tmp_path / "model.conf"), ns)
    conf = loaders.load_model_conf(path)
    assert isinstance(conf, argparse.Namespace) and conf.n_resch == 32 and conf.feature_format == "h5"
    assert loaders.model_kwargs(conf) == TINY.kwargs()


def test_adam_state_uses_torch_adam_layout():
    """FusedTrainer / FlatAdam checkpoints == torch.optim.Adam.state_dict(): a torch Adam loads ours and continues, and
    ours loads a torch Adam's (what a reference-made checkpoint holds, qpnet_train.py:346-352)."""
    from qpnet_amd.qpnet import QPNet
    from qpnet_amd.train import adam_state_to_torch, adam_state_from_torch
    torch.manual_seed(0)
    m = QPNet(**TINY.kwargs())
    n = sum(p.numel() for p in m.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    mom, var = torch.randn(n), torch.rand(n)
    sd = adam_state_to_torch(m, mom, var, 7, dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0))
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    opt.load_state_dict(sd)                                  # torch accepts the layout ...
    assert opt.param_groups[0]["lr"] == 2e-4
    st = opt.state[next(iter(m.parameters()))]
    assert float(st["step"]) == 7.0
    for p in m.parameters():
        p.grad = torch.randn_like(p)
    opt.step()                                               # ... and can continue from it
    sd2 = opt.state_dict()
    m2, v2, steps, hyper = adam_state_from_torch(m, sd2, flat)
    assert steps == 8 and hyper["lr"] == 2e-4
    # one Adam step by hand on the flat buffers equals what torch did
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    np.testing.assert_allclose(m2.numpy(), (mom + (g - mom) * 0.1).numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(v2.numpy(), (var * 0.999 + 0.001 * g * g).numpy(), rtol=1e-6, atol=1e-7)
    # a fresh torch Adam (no state yet) and an old-style int step both load
    m3, v3, s3, _ = adam_state_from_torch(m, torch.optim.Adam(m.parameters()).state_dict(), flat)
    assert s3 == 0 and float(m3.abs().max()) == 0.0
    for k in sd2["state"]:
        sd2["state"][k]["step"] = 8                          # PyTorch 1.3 stored python ints
    assert adam_state_from_torch(m, sd2, flat)[2] == 8


def test_weighted_exchange_protocol_single_process():
    """[n_r g_r | n_r, 0, 0, 0] summed over ranks, divided by the summed count == gradient of the global mean loss."""
    from qpnet_amd import parallel
    rs = np.random.RandomState(1)
    g = [torch.from_numpy(rs.randn(50).astype(np.float32)) for _ in range(3)]
    n = [1200.0, 900.0, 1500.0]
    bufs = []
    for gi, ni in zip(g, n):
        b = torch.zeros(50 + parallel.TRAILER); b[:50] = gi * ni; b[50] = ni
        bufs.append(b)
    tot = sum(bufs)
    np.testing.assert_allclose((tot[:50] / tot[50]).numpy(), (sum(gi * ni for gi, ni in zip(g, n)) / sum(n)).numpy(), rtol=1e-6)
    assert float(tot[51:].abs().max()) == 0.0
    assert parallel.exchange(bufs[0]) is bufs[0]             # no process group: identity
