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
    """The trainer's running window over the concatenated utterance stream, kept as a queue of per-utterance segments
    (samples, frame-rate dilated factors + their suffix maxima, features) and consumed from the front by a frame offset.
    Appending an utterance is O(1): nothing is concatenated until a chunk is actually cut (`front`), and the largest
    dilated factor still in the window -- which sets the receptive field after every append -- comes from the suffix
    maxima instead of a pass over the sample-rate array.  Every utterance holds exactly U samples per frame
    (harness.validate_length), so the window's sample offset is always U times its frame offset."""

    def __init__(self, n_feat, feat_dtype, U):
        import collections
        self.U = int(U)
        self.segs = collections.deque()                 # [x (F*U,) f32 | None, d (F,) f64, sufmax (F,) f64, h (F, D) | None, load, part]
        self.f0 = 0                                     # frames of the first segment already consumed
        self.n_frames = 0                               # live frames / samples in the window
        self.h_dtype = np.result_type(np.float32, feat_dtype)
        self.n_feat = n_feat

    @property
    def n_samples(self):
        return self.n_frames * self.U

    def append(self, x, h, d_frames, sufmax=None, load=None, part=None):
        """One utterance.  x / h may be None with `load` a callable returning the (validated) pair: the PLAN of the stream -- where chunks begin and end,
        which receptive field each has -- needs only the frame count and the dilated factors; samples and features are fetched when a chunk
        that is actually cut (`front`) reaches into the utterance (a data-parallel rank cuts every world-th chunk only).  `part(s0, s1, f0, f1)`, when the
        utterance's source has it, returns just samples [s0, s1) and feature rows [f0, f1): a rank whose chunks are a world-th of the stream then reads a
        world-th of the BYTES, however long the utterances are (its consecutive chunks lie `world` chunks apart: nothing of a whole-utterance load is reused)."""
        d = np.asarray(d_frames, dtype=np.float64)      # float32 buffer + float64 factors promote to float64 upstream too
        if x is not None:
            assert len(x) == len(d) * self.U == h.shape[0] * self.U
        if len(d) == 0:
            return
        if sufmax is None:
            sufmax = np.fmax.accumulate(d[::-1])[::-1]  # (fmax: a NaN factor is ignored, as by the reference's np.nanmax; all-NaN tails stay NaN)
        self.segs.append([None if x is None else np.asarray(x, dtype=np.float32), d, sufmax,
                          None if h is None else h.astype(self.h_dtype, copy=False), load, part])
        self.n_frames += len(d)

    def max_factor(self):
        """largest dilated factor among the live samples (what np.nanmax over the reference's d_buffer returns)."""
        m = np.nan
        for k, seg in enumerate(self.segs):
            m = np.fmax(m, seg[2][self.f0 if k == 0 else 0])
        return float(m)                                 # (NaN only when every live factor is NaN: np.nanmax's answer as well)

    def front(self, frames, samples):
        """first `frames` feature rows and `samples` (= frames*U + 1) samples / sample-rate factors of the window."""
        U = self.U
        xs, ds, hs = [], [], []
        need_s, need_f, off = samples, frames, self.f0
        for seg in self.segs:
            if need_s <= 0:
                break
            d = seg[1]
            take_s = min(need_s, len(d) * U - off * U)
            take_fd = -(-take_s // U)                   # frames whose factors those samples repeat
            take_f = min(need_f, len(d) - off) if need_f > 0 else 0
            if seg[0] is None and seg[5] is not None:   # planned only, and its source can hand out a slice: exactly what this chunk needs of it
                xp, hp = seg[5](off * U, off * U + take_s, off, off + take_f)
                assert len(xp) == take_s and hp.shape[0] == take_f, "an utterance changed between the plan and the load"
                xs.append(np.asarray(xp, dtype=np.float32))
                if take_f > 0:
                    hs.append(hp.astype(self.h_dtype, copy=False))
            else:
                if seg[0] is None:                      # planned only so far: fetch the utterance now (kept until the window has moved past it)
                    x, h = seg[4]()
                    assert len(x) == len(d) * U == h.shape[0] * U, "an utterance changed between the plan and the load"
                    seg[0], seg[3] = np.asarray(x, dtype=np.float32), h.astype(self.h_dtype, copy=False)
                xs.append(seg[0][off * U:off * U + take_s])
                if take_f > 0:
                    hs.append(seg[3][off:off + take_f])
            ds.append(np.repeat(d[off:off + take_fd], U)[:take_s])
            need_f -= take_f
            need_s -= take_s
            off = 0
        cat = lambda parts: parts[0] if len(parts) == 1 else np.concatenate(parts)      # noqa: E731
        return cat(xs), cat(hs), cat(ds)

    def advance(self, frames, samples):
        assert samples == frames * self.U
        self.n_frames -= frames
        self.f0 += frames
        while self.segs and self.f0 >= len(self.segs[0][1]):
            self.f0 -= len(self.segs[0][1])
            self.segs.popleft()


class _UtterancePlan:
    """What the chunk plan needs of every utterance -- frame count after validate_length, per-frame dilated factors and their suffix maxima, the feature
    dtype / width -- computed ONCE per utterance and kept for the later epochs (a few KB each).  Source, cheapest first: the item's own `plan()` hook
    (runners' file-backed loaders: the wav header and the feature file, no waveform), else one full load of the item."""

    def __init__(self, utterances, U, f0_threshold, fs, dense_factor):
        self.utts, self.U, self.f0_threshold, self.fs, self.dense = utterances, int(U), f0_threshold, fs, dense_factor
        self.meta = {}
        self.loads = 0                                  # full loads of an utterance (samples + features) so far: what a sharded rank should need few of
        self.parts = 0                                  # partial loads (the item's `part` hook)

    def load(self, i):
        item = self.utts[i]
        x, h = item() if callable(item) else item
        if callable(item):
            self.loads += 1
        return harness.validate_length(np.array(x, dtype=np.float32), np.asarray(h), self.U)

    def _entry(self, n_wav, h):
        frames = h.shape[0]
        shortfall = frames * self.U - n_wav             # (harness.validate_length on the lengths alone)
        if shortfall > 0:
            frames = max(frames - (shortfall // self.U + 1), 0)
        d = np.asarray(harness.dilated_factor(harness.batch_f0(h[:frames], self.f0_threshold), self.fs, self.dense), dtype=np.float64)
        return (frames, d, np.fmax.accumulate(d[::-1])[::-1] if frames else d, h.shape[1], h.dtype)

    def part_hook(self, i):
        item = self.utts[i]
        hook = getattr(item, "part", None) if callable(item) else None
        if hook is None:
            return None

        def part(s0, s1, f0, f1):
            self.parts += 1
            return hook(s0, s1, f0, f1)
        return part

    def get(self, i):
        """-> (entry, loaded pair or None)"""
        e = self.meta.get(i)
        if e is not None:
            return e, None
        item = self.utts[i]
        hook = getattr(item, "plan", None) if callable(item) else None
        if hook is not None:
            n_wav, h = hook()
            e, pair = self._entry(int(n_wav), np.asarray(h)), None
        else:
            pair = self.load(i)
            e = self._entry(len(pair[0]), pair[1])
        self.meta[i] = e
        return e, pair


def train_generator(utterances, model_receptiveCausal, model_receptiveF, model_receptiveA, fs, wav_transform=None,
                    feat_transform=None, dense_factor=8, batch_length=20000, batch_size=1, max_length=23070,
                    f0_threshold=0, upsampling_factor=80, shuffle=True, device=None, epochs=None, shard=None):
    """Chunked teacher-forcing batches over a stream of utterances (reference generator: src/bin/qpnet_train.py:200-335).

    utterances: list of (x float waveform in [-1,1], h (F, n_aux)) pairs, or zero-argument callables returning such a
    pair (lazy file loading).  Yields (batch_x, batch_h, batch_t, batch_d, batch_b); endless unless `epochs` is given.

    Behaviour kept from the reference (pinned by tests/test_loaders_cpu.py): utterances are concatenated into one stream
    that survives epoch boundaries; after each appended utterance the receptive field is taken from the largest dilated
    factor still in the window, the batch length is cut to fit max_length and a whole number of frames, and chunks of
    RF + BL (+1 sample for the input/target shift) are cut while MORE than `slots_left` chunks' worth of frames and
    samples remain, each advancing the window by BL (so consecutive chunks overlap by RF).

    shard = (rank, world): data-parallel ranks walk the SAME stream (same seed, same chunk boundaries) but only batch
    numbers rank, rank + world, ... are materialised and yielded; for the others the window just advances -- no mu-law
    encoding, scaling or tensor is made for a chunk another rank consumes, and the stream is laid out from per-utterance
    metadata (frames + dilated factors, computed once per utterance -- through the item's `plan()` hook when it has one --
    and cached over the epochs): an utterance's samples and features are loaded only when one of this rank's own chunks
    reaches into it (no counterpart in the reference, whose only multi-GPU path is a dead DataParallel wrapper,
    qpnet_train.py:416-423).  An item with a `part(s0, s1, f0, f1)` hook (runners' file-backed
    loaders) is read in slices: exactly the samples / feature rows a chunk needs.  `train_generator.last_stats` counts the full ("loads") and
    partial ("parts") loads of the most recent generator."""
    U = int(upsampling_factor)
    rank, world = (0, 1) if shard is None else (int(shard[0]), int(shard[1]))
    n_batches = 0                                       # batches cut so far, over all ranks
    n_files = len(utterances)
    order = np.random.permutation(n_files) if shuffle else np.arange(n_files)
    win = None
    epoch = 0
    # With world > 1 the stream is PLANNED from per-utterance metadata (frames, dilated factors: _UtterancePlan, cached across epochs) and an utterance's
    # samples / features are loaded only when one of THIS rank's chunks reaches into it (every rank used to load, validate and take the
    # factors of every utterance, so a rank's host work per step grew with the world size).  world == 1 touches every utterance anyway: loaded on append.
    plan = _UtterancePlan(utterances, U, f0_threshold, fs, dense_factor) if world > 1 else None
    stats = {"loads": 0, "parts": 0}                    # (train_generator.last_stats: full / partial utterance loads of the most recent generator, for tests / tools)
    train_generator.last_stats = stats
    while epochs is None or epoch < epochs:
        rows = []                                       # (x, h, t, d, bl) of the batch being filled
        slots_left = batch_size
        for i in order:
            if plan is None:
                item = utterances[int(i)]
                x, h = item() if callable(item) else item
                stats["loads"] += 1
                x, h = harness.validate_length(np.array(x, dtype=np.float32), np.asarray(h), U)
                d = harness.dilated_factor(harness.batch_f0(h, f0_threshold), fs, dense_factor)
                if win is None:
                    win = _SampleWindow(h.shape[1], h.dtype, U)
                win.append(x, h, d)
            else:
                (nfr, d, sufmax, n_feat, h_dtype), pair = plan.get(int(i))
                if win is None:
                    win = _SampleWindow(n_feat, h_dtype, U)

                def load(i=int(i)):
                    out = plan.load(i)
                    stats["loads"] = plan.loads
                    return out
                if pair is not None:                    # (the plan had to load it: keep what was loaded)
                    stats["loads"] = plan.loads
                    win.append(pair[0], pair[1], d, sufmax)
                else:
                    win.append(None, None, d, sufmax, load, plan.part_hook(int(i)))
            rf = harness.receptive_field(model_receptiveCausal, model_receptiveF, model_receptiveA, win.max_factor())
            bl, frames, samples = harness.chunk_plan(rf, batch_length, max_length, U)
            hop_frames = bl // U
            while win.n_frames > slots_left * frames and win.n_samples > slots_left * samples:
                mine = n_batches % world == rank
                if mine:
                    xs, hs, ds = win.front(frames, samples)
                    if plan is not None:
                        stats["parts"] = plan.parts
                    if wav_transform is not None:
                        xs = wav_transform(xs)
                    if feat_transform is not None:
                        hs = feat_transform(hs)
                    xs = torch.from_numpy(np.asarray(xs)).long()
                    rows.append((xs[:-1], torch.from_numpy(np.asarray(hs)).float().transpose(0, 1), xs[1:],
                                 torch.from_numpy(np.asarray(ds)).float()[:-1], bl))
                else:
                    rows.append(None)
                slots_left -= 1
                win.advance(hop_frames, hop_frames * U)
                if len(rows) == batch_size:
                    if mine:
                        out = tuple(torch.stack([r[k] for r in rows]) for k in range(4)) + (torch.tensor([r[4] for r in rows]),)
                        if device is not None:
                            out = tuple(o.to(device) for o in out)
                        yield out
                    rows, slots_left = [], batch_size
                    n_batches += 1
        if shuffle:
            order = order[np.random.permutation(n_files)]      # the reference re-shuffles the ALREADY shuffled list (:331-335)
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
    """h5py, or -- where only the HDF5 C library is installed (this image) -- the same few calls on libhdf5 itself (qpnet_amd/_hdf5.py): either
    way the files are real HDF5, the format the reference's feature extraction writes and its trainers read (utils.py:43-128)."""
    try:
        import h5py
        return h5py
    except ImportError:
        pass
    try:
        from . import _hdf5
        _hdf5._lib()
        return _hdf5
    except ImportError as e:
        raise ImportError("neither h5py nor an HDF5 library (libhdf5, QPN_LIBHDF5=<path>) is available for the reference's .h5 feature files "
                          "(utils.py:43-92); use feature_format 'npy' or pass arrays to decode_generator / train_generator instead") from e


def read_hdf5(hdf5_name, hdf5_path):
    """reference utils.read_hdf5 (utils.py:43-68): dataset values, e.g. read_hdf5(f, "/world")."""
    h5py = _h5py()
    if not os.path.exists(hdf5_name):
        raise FileNotFoundError("There is no such a hdf5 file. (%s)" % hdf5_name)
    with h5py.File(hdf5_name, "r") as f:
        if hdf5_path not in f:
            raise KeyError("There is no such a data in hdf5 file. (%s)" % hdf5_path)
        return f[hdf5_path][()]


def check_hdf5(hdf5_name, hdf5_path):
    """reference utils.check_hdf5 (utils.py:19-40): the file exists and holds that dataset."""
    if not os.path.exists(hdf5_name):
        return False
    with _h5py().File(hdf5_name, "r") as f:
        return hdf5_path in f


def shape_hdf5(hdf5_name, hdf5_path):
    """reference utils.shape_hdf5 (utils.py:71-88): the dataset's shape without reading it (the reference exits where this raises)."""
    if not check_hdf5(hdf5_name, hdf5_path):
        raise KeyError("There is no such a file or dataset. (%s, %s)" % (hdf5_name, hdf5_path))
    with _h5py().File(hdf5_name, "r") as f:
        return tuple(f[hdf5_path].shape)


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
