"""In-memory counterparts of the reference's batch generators and checkpoint helpers.

Same batching / chunking semantics as the reference task scripts, but over arrays instead of .wav/.h5 files
(the file formats of rank 4 are at the end of this module):

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


class _SampleWindow:
    """The trainer's running window over the concatenated utterance stream: one waveform-rate array pair (samples,
    dilated factors) and one frame-rate array (features), consumed from the front by explicit offsets.  Storage is
    compacted only when the consumed prefix outweighs the live part, so appending an utterance costs its own length."""

    def __init__(self, n_feat, feat_dtype):
        self.x = np.empty(0, dtype=np.float32)
        self.d = np.empty(0, dtype=np.float64)          # float32 buffer + float64 factors promote to float64 upstream too
        self.h = np.empty((0, n_feat), dtype=np.result_type(np.float32, feat_dtype))
        self.s0 = 0                                     # first live sample
        self.f0 = 0                                     # first live frame

    def append(self, x, h, d):
        if self.s0 > len(self.x) - self.s0:             # drop the consumed prefix
            self.x, self.d, self.h = self.x[self.s0:], self.d[self.s0:], self.h[self.f0:]
            self.s0 = self.f0 = 0
        self.x = np.concatenate([self.x, np.asarray(x, dtype=np.float32)])
        self.d = np.concatenate([self.d, np.asarray(d, dtype=np.float64)])
        self.h = np.concatenate([self.h, h.astype(self.h.dtype, copy=False)])

    @property
    def n_samples(self):
        return len(self.x) - self.s0

    @property
    def n_frames(self):
        return len(self.h) - self.f0

    def max_factor(self):
        return self.d[self.s0:]

    def front(self, frames, samples):
        return (self.x[self.s0:self.s0 + samples], self.h[self.f0:self.f0 + frames], self.d[self.s0:self.s0 + samples])

    def advance(self, frames, samples):
        self.f0 += frames
        self.s0 += samples


def train_generator(utterances, model_receptiveCausal, model_receptiveF, model_receptiveA, fs, wav_transform=None,
                    feat_transform=None, dense_factor=8, batch_length=20000, batch_size=1, max_length=23070,
                    f0_threshold=0, upsampling_factor=80, shuffle=True, device=None, epochs=None):
    """Chunked teacher-forcing batches over a stream of utterances (reference generator: src/bin/qpnet_train.py:200-335).

    utterances: list of (x float waveform in [-1,1], h (F, n_aux)) pairs, or zero-argument callables returning such a
    pair (lazy file loading).  Yields (batch_x, batch_h, batch_t, batch_d, batch_b); endless unless `epochs` is given.

    Behaviour kept from the reference (pinned by tests/test_loaders_cpu.py): utterances are concatenated into one stream
    that survives epoch boundaries; after each appended utterance the receptive field is taken from the largest dilated
    factor still in the window, the batch length is cut to fit max_length and a whole number of frames, and chunks of
    RF + BL (+1 sample for the input/target shift) are cut while MORE than `slots_left` chunks' worth of frames and
    samples remain, each advancing the window by BL (so consecutive chunks overlap by RF)."""
    U = int(upsampling_factor)
    n_files = len(utterances)
    order = np.random.permutation(n_files) if shuffle else np.arange(n_files)
    win = None
    epoch = 0
    while epochs is None or epoch < epochs:
        rows = []                                       # (x, h, t, d, bl) of the batch being filled
        slots_left = batch_size
        for i in order:
            item = utterances[int(i)]
            x, h = item() if callable(item) else item
            x, h = harness.validate_length(np.array(x, dtype=np.float32), np.asarray(h), U)
            d = harness.dilated_factor(harness.batch_f0(h, f0_threshold), fs, dense_factor)
            if win is None:
                win = _SampleWindow(h.shape[1], h.dtype)
            win.append(x, h, np.repeat(d, U))
            rf = harness.receptive_field(model_receptiveCausal, model_receptiveF, model_receptiveA, win.max_factor())
            bl, frames, samples = harness.chunk_plan(rf, batch_length, max_length, U)
            hop_frames = bl // U
            while win.n_frames > slots_left * frames and win.n_samples > slots_left * samples:
                xs, hs, ds = win.front(frames, samples)
                if wav_transform is not None:
                    xs = wav_transform(xs)
                if feat_transform is not None:
                    hs = feat_transform(hs)
                xs = torch.from_numpy(np.asarray(xs)).long()
                rows.append((xs[:-1], torch.from_numpy(np.asarray(hs)).float().transpose(0, 1), xs[1:],
                             torch.from_numpy(np.asarray(ds)).float()[:-1], bl))
                slots_left -= 1
                win.advance(hop_frames, hop_frames * U)
                if len(rows) == batch_size:
                    out = tuple(torch.stack([r[k] for r in rows]) for k in range(4)) + (torch.tensor([r[4] for r in rows]),)
                    if device is not None:
                        out = tuple(o.to(device) for o in out)
                    yield out
                    rows, slots_left = [], batch_size
        if shuffle:
            order = np.random.permutation(n_files)
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


def read_wav(path):
    """(fs, float32 waveform in [-1, 1)) the way the trainer reads it (qpnet_train.py:250-251: int16 / 32768)."""
    from scipy.io import wavfile
    fs, x = wavfile.read(path)
    return fs, np.array(x, dtype=np.float32) / 32768


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError as e:     # not installed in the build image
        raise ImportError("h5py is required for the reference's .h5 feature files (utils.py:43-92); "
                          "use feature_format 'npy' or pass arrays to decode_generator / train_generator instead") from e


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


def read_features(path, feature_type="world"):
    """(F, n_aux) acoustic features of one utterance: the `/world` dataset of an .h5 file (layout
    [uv, cont_f0, mcep.., codeap..], feature_extract.py:337-343), or a plain .npy array of the same layout."""
    if path.endswith(".npy"):
        return np.load(path)
    return read_hdf5(path, "/%s" % feature_type)


class FeatureScaler:
    """The StandardScaler the task scripts rebuild from the stats file (qpnet_train.py:433-440): transform = (x - mean) / scale."""

    def __init__(self, mean, scale):
        self.mean_ = np.asarray(mean, dtype=np.float64)
        self.scale_ = np.asarray(scale, dtype=np.float64)
        self.n_features_in_ = self.mean_.shape[0]

    def transform(self, x):
        return (np.asarray(x) - self.mean_) / self.scale_

    __call__ = transform


def read_scaler_stats(path, feature_type="world"):
    """`/world/mean`, `/world/scale` of the stats file calc_stats.py writes (:19-37; the uv dimension has mean 0 / scale 1),
    or an .npz with arrays `mean` and `scale`."""
    if path.endswith(".npz"):
        z = np.load(path)
        return FeatureScaler(z["mean"], z["scale"])
    return FeatureScaler(read_hdf5(path, "/%s/mean" % feature_type), read_hdf5(path, "/%s/scale" % feature_type))


def calc_stats(feature_arrays):
    """mean / scale over a list of (F, n_aux) feature arrays the way calc_stats.py does: population statistics of every
    dimension but the first (uv), which keeps mean 0 / scale 1; zero variance -> scale 1 (sklearn's rule)."""
    n, s1, s2 = 0, 0.0, 0.0
    for f in feature_arrays:
        f = np.asarray(f, dtype=np.float64)[:, 1:]
        n += f.shape[0]; s1 = s1 + f.sum(0); s2 = s2 + (f * f).sum(0)
    mean = s1 / n
    var = np.maximum(s2 / n - mean * mean, 0.0)
    scale = np.sqrt(var)
    scale[scale < 10 * np.finfo(np.float64).eps] = 1.0
    return FeatureScaler(np.concatenate([[0.0], mean]), np.concatenate([[1.0], scale]))


def save_model_conf(path, args):
    """the trainer pickles its argparse.Namespace as the model config (qpnet_train.py:389)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    torch.save(args, path)
    return path


def load_model_conf(path):
    """model.conf = a pickled argparse.Namespace (read with a bare torch.load at qpnet_decode.py:245; torch >= 2.6 needs
    weights_only=False for it).  Returns the Namespace."""
    return torch.load(path, map_location="cpu", weights_only=False)


_MODEL_KEYS = ("n_quantize", "n_aux", "n_resch", "n_skipch", "dilationF_depth", "dilationF_repeat",
               "dilationA_depth", "dilationA_repeat", "kernel_size", "upsampling_factor")


def model_kwargs(conf):
    """QPNet constructor kwargs out of a model config Namespace (qpnet_decode.py:275-285)."""
    return {k: getattr(conf, k) for k in _MODEL_KEYS}


def read_txt(path):
    """one entry per line (utils.read_txt, utils.py:151-162)."""
    with open(path) as f:
        return [ln.strip() for ln in f if ln.strip()]


def find_files(directory, pattern="*.wav"):
    import fnmatch
    out = []
    for root, _, names in os.walk(directory, followlinks=True):
        out += [os.path.join(root, n) for n in fnmatch.filter(names, pattern)]
    return sorted(out)


def file_lists(waveforms, feats, feature_format="h5"):
    """(wav_list, feat_list) from a directory pair or a pair of list files (qpnet_train.py:442-455)."""
    if os.path.isdir(waveforms):
        wavs = find_files(waveforms, "*.wav")
        return wavs, [os.path.join(feats, os.path.basename(w).replace(".wav", "." + feature_format)) for w in wavs]
    if os.path.isfile(waveforms):
        wavs, fts = read_txt(waveforms), read_txt(feats)
        assert len(wavs) == len(fts)
        return wavs, fts
    raise FileNotFoundError("--waveforms should be directory or list: %s" % waveforms)
