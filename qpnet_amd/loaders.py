"""In-memory counterparts of the reference's batch generators and checkpoint helpers.

Same batching / chunking semantics as the reference task scripts, but over arrays instead of .wav/.h5 files
(file formats are rank 4 and stay out of scope; h5py is not even installed here):

  decode_generator  <- src/bin/qpnet_decode.py:122-209  (sort by length, array_split batching, F0 scaling,
                       d = fs/(f0*dense) extended x U, scaler, zero pad_list, n_samples = F*U - 1)
  train_generator   <- src/bin/qpnet_train.py:200-335   (running x/h/d buffers over utterances, receptive field from the
                       buffer's max d, batch_length shrunk to max_length and to a multiple of U, chunks of RF+BL samples
                       that overlap by RF, x[:-1] / x[1:] input-target shift)
  save_checkpoint / load_checkpoint <- src/bin/qpnet_train.py:338-353,481-499,557-563 ({"model","optimizer","iterations"})
"""
import math
import os

import numpy as np
import torch

from . import harness
from .qpnet import encode_mu_law


def decode_generator(feats, fs, feat_ids=None, wav_transform=None, feat_transform=None, dense_factor=8, batch_size=32,
                     upsampling_factor=80, f0_factor=1.0, f0_dim_index=1, extra_memory=False, device=None):
    """feats: list of (F_i, n_aux) float arrays.  Yields (feat_ids, batch_x, batch_h, n_samples_list, batch_d) exactly as
    the reference generator does (batch_d is a numpy float64 array unless extra_memory)."""
    if feat_ids is None:
        feat_ids = ["utt%04d" % i for i in range(len(feats))]
    idx = np.argsort([f.shape[0] for f in feats])                     # sort with the feature length (:152-155)
    order = [int(i) for i in idx]
    n_batch = math.ceil(len(order) / batch_size)
    for part in np.array_split(np.array(order, dtype=np.int64), n_batch):   # (:157-160)
        bx, bh, bd, ids, ns = [], [], [], [], []
        for i in part.tolist():
            x = np.zeros((1))
            h = np.array(feats[i], copy=True)
            # `if f0_factor is not 1.0` in the reference is an identity test that is true for any parsed float,
            # so F0 is always scaled and d always rebuilt (:172-175)
            h[:, f0_dim_index] = h[:, f0_dim_index] * f0_factor
            d = harness.dilated_factor(harness.batch_f0(h), fs, dense_factor)
            d = harness.extend_time(np.expand_dims(d, -1), upsampling_factor)
            if wav_transform is not None:
                x = wav_transform(x)
            if feat_transform is not None:
                h = feat_transform(h)
            bx.append(x); bh.append(h); bd.append(d); ids.append(feat_ids[i]); ns.append(h.shape[0] * upsampling_factor - 1)
        batch_x = torch.from_numpy(np.stack(bx, axis=0)).long()
        batch_h = torch.from_numpy(harness.pad_list(bh)).float().transpose(1, 2)
        batch_d = harness.pad_list(bd)
        batch_d = torch.from_numpy(batch_d).float().squeeze(-1) if extra_memory else batch_d.squeeze(-1)
        if device is not None:
            batch_x, batch_h = batch_x.to(device), batch_h.to(device)
            if extra_memory:
                batch_d = batch_d.to(device)
        yield ids, batch_x, batch_h, ns, batch_d


def train_generator(utterances, model_receptiveCausal, model_receptiveF, model_receptiveA, fs, wav_transform=None,
                    feat_transform=None, dense_factor=8, batch_length=20000, batch_size=1, max_length=23070,
                    f0_threshold=0, upsampling_factor=80, shuffle=True, device=None, epochs=None):
    """utterances: list of (x float waveform in [-1,1], h (F, n_aux)) pairs.  Yields (batch_x, batch_h, batch_t, batch_d,
    batch_b) like the reference generator; endless unless `epochs` is given."""
    n_files = len(utterances)
    order = list(np.random.permutation(n_files)) if shuffle else list(range(n_files))
    x_buffer = h_buffer = d_buffer = None
    epoch = 0
    while epochs is None or epoch < epochs:
        bx, bh, bt, bd, bb = [], [], [], [], []
        batch_count = batch_size
        for i in order:
            x, h = utterances[i]
            x = np.array(x, dtype=np.float32)
            x, h = harness.validate_length(x, np.asarray(h), upsampling_factor)
            d = harness.dilated_factor(harness.batch_f0(h, f0_threshold), fs, dense_factor)
            d = np.squeeze(harness.extend_time(np.expand_dims(d, -1), upsampling_factor), -1)
            if x_buffer is None:
                x_buffer = np.empty((0), dtype=np.float32)
                h_buffer = np.empty((0, h.shape[1]), dtype=np.float32)
                d_buffer = np.empty((0), dtype=np.float32)
            x_buffer = np.concatenate([x_buffer, x], axis=0)
            h_buffer = np.concatenate([h_buffer, h], axis=0)
            d_buffer = np.concatenate([d_buffer, d], axis=0)
            rf = harness.receptive_field(model_receptiveCausal, model_receptiveF, model_receptiveA, d_buffer)
            mod1 = max(rf + batch_length - max_length, 0)                  # avoid out-of-memory (:273-275)
            bl = batch_length - mod1
            bl -= (rf + bl) % upsampling_factor                            # meet the upsampling ratio (:276-278)
            h_bs = (rf + bl) // upsampling_factor
            x_bs = h_bs * upsampling_factor + 1
            while len(h_buffer) > (batch_count * h_bs) and len(x_buffer) > (batch_count * x_bs):
                h_, x_, d_ = h_buffer[:h_bs, :], x_buffer[:x_bs], d_buffer[:x_bs]
                if wav_transform is not None:
                    x_ = wav_transform(x_)
                if feat_transform is not None:
                    h_ = feat_transform(h_)
                x_ = torch.from_numpy(np.asarray(x_)).long()
                h_ = torch.from_numpy(np.asarray(h_)).float()
                d_ = torch.from_numpy(np.asarray(d_)).float()
                bh.append(h_.transpose(0, 1)); bx.append(x_[:-1]); bt.append(x_[1:]); bd.append(d_[:-1]); bb.append(bl)
                batch_count -= 1
                h_ss = bl // upsampling_factor                              # shift = batch_length: chunks overlap by RF
                x_ss = h_ss * upsampling_factor
                h_buffer, x_buffer, d_buffer = h_buffer[h_ss:, :], x_buffer[x_ss:], d_buffer[x_ss:]
                if len(bx) == batch_size:
                    out = (torch.stack(bx), torch.stack(bh), torch.stack(bt), torch.stack(bd), torch.tensor(bb))
                    if device is not None:
                        out = tuple(o.to(device) for o in out)
                    yield out
                    bx, bh, bt, bd, bb = [], [], [], [], []
                    batch_count = batch_size
        if shuffle:
            order = list(np.random.permutation(n_files))
        epoch += 1


def mu_law_transform(n_quantize=256):
    """wav_transform of the reference scripts: encode_mu_law(x, n_quantize)."""
    return lambda x: encode_mu_law(x, n_quantize)


def save_checkpoint(checkpoint_dir, model, optimizer, iterations):
    """reference _save_checkpoint (qpnet_train.py:338-353)."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, "checkpoint-%d.pkl" % iterations)
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "iterations": iterations}, path)
    return path


def save_final(checkpoint_dir, model):
    """reference final model file: {"model": state_dict} only (qpnet_train.py:557-563)."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, "checkpoint-final.pkl")
    torch.save({"model": model.state_dict()}, path)
    return path


def load_checkpoint(path, model, optimizer=None, map_location="cpu"):
    """reference --resume (qpnet_train.py:481-499): restores model (+optimizer, iteration count)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ck["model"])
    if optimizer is not None and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    return int(ck.get("iterations", 0))


# ---------------------------------------------------------------- on-disk formats
def samples_to_int16(samples, n_quantize=256):
    """mu-law class ids -> int16 PCM exactly as the reference decode script writes it (qpnet_decode.py:316-319):
    decode_mu_law, x 32768, clip to [-32768, 32767], truncate to int16."""
    from .qpnet import decode_mu_law
    wav = decode_mu_law(np.asarray(samples), n_quantize)
    return np.clip(wav * 32768, -32768, 32767).astype(np.int16)


def write_wav(path, fs, samples, n_quantize=256):
    """Write one decoded utterance as 16-bit PCM (scipy.io.wavfile, as the reference does)."""
    from scipy.io import wavfile
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    wavfile.write(path, fs, samples_to_int16(samples, n_quantize))
    return path


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError as e:     # not installed in the build image; the corpus scripts that need it are out of scope
        raise ImportError("h5py is required for the reference's .h5 feature files (utils.py:43-92); "
                          "pass arrays to decode_generator / train_generator instead") from e


def read_hdf5(hdf5_name, hdf5_path):
    """reference utils.read_hdf5 (utils.py:43-68): dataset values, e.g. read_hdf5(f, "/world")."""
    h5py = _h5py()
    if not os.path.exists(hdf5_name):
        raise FileNotFoundError("There is no such a hdf5 file. (%s)" % hdf5_name)
    with h5py.File(hdf5_name, "r") as f:
        if hdf5_path not in f:
            raise KeyError("There is no such a data in hdf5 file. (%s)" % hdf5_path)
        return f[hdf5_path][()]


def write_hdf5(hdf5_name, hdf5_path, write_data, is_overwrite=True):
    """reference utils.write_hdf5 (utils.py:71-116)."""
    h5py = _h5py()
    write_data = np.array(write_data)
    os.makedirs(os.path.dirname(os.path.abspath(hdf5_name)) or ".", exist_ok=True)
    with h5py.File(hdf5_name, "a") as f:
        if hdf5_path in f:
            if not is_overwrite:
                raise KeyError("Dataset in hdf5 file already exists. (%s)" % hdf5_path)
            del f[hdf5_path]
        f.create_dataset(hdf5_path, data=write_data)
