"""Task entry points over the native hot path: train / SD-update / validate / decode.

Same command-line flags and on-disk artefacts as the reference task scripts
  src/bin/qpnet_train.py    (main loop :517-567, resume :481-499, checkpoints :338-353)
  src/bin/qpnet_update.py   (--pretrain / --resume :444-464)
  src/bin/qpnet_validate.py (validation_result.yml :409-437)
  src/bin/qpnet_decode.py   (per-GPU split :258-259,322-331; wav writing :315-320)
but the loop body is the fused step behind the C ABI (FusedTrainer) and multi-GPU is one process per GPU
(torch.distributed / RCCL): rank r consumes chunks r, r+N, ... of the generator stream and the flat gradient is
all-reduced once per step; decode splits the utterance list over ranks with no communication.

    python -m qpnet_amd.run_train    --waveforms .. --feats .. --stats .. --expdir .. --config .. [--n_gpus N]
    python -m qpnet_amd.run_update   ... --pretrain checkpoint-final.pkl
    python -m qpnet_amd.run_validate ... --checkpoint .. --resultdir ..
    python -m qpnet_amd.run_decode   --feats .. --stats .. --config .. --checkpoint .. --outdir out/feat_id.wav

With --n_gpus N > 1 and no torchrun environment the entry point re-launches itself as N ranks (before any GPU call).
"""
import argparse
import logging
import os
import queue
import subprocess
import sys
import threading
import time

import numpy as np
import torch
import yaml

from . import loaders


# ---------------------------------------------------------------- shared plumbing
def _setup_logging(verbose):
    level = logging.INFO if verbose == 1 else logging.DEBUG if verbose > 1 else logging.WARN
    logging.basicConfig(level=level, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s",
                        datefmt="%m/%d/%Y %I:%M:%S")


def _fix_seed(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def launch_ranks(n_gpus, module, argv):
    """Re-launch `python -m module argv` as n_gpus ranks under torch.distributed.run (the parent never touches the GPU);
    returns the children's exit code, or None when this process already is a rank / runs single-GPU."""
    if n_gpus <= 1 or "RANK" in os.environ:
        return None
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", module] + list(argv)
    return subprocess.call(cmd)


def dist_context():
    """(rank, world, device): initialises the process group when launched as ranks (backend nccl = RCCL; gloo when
    QPN_DIST_BACKEND=gloo, used by one-GPU rehearsals)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("qpnet_amd needs an AMD GPU: there is no CPU fallback for the hot path")
    backend = os.environ.get("QPN_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("QPN_BENCH_ONE_GPU") else local
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    return rank, world, dev


class PinnedStager:
    """Host-to-device copies of the loader's batches through a small ring of REUSED pinned staging buffers.

    `tensor.to(device)` from pageable memory goes through the runtime's own bounce buffers and blocks the calling thread: measured
    1150 chunks/s of 0.5 MB on the GPU box (tools/loader_rate.py) -- no faster than one GPU consumes them.  Staging into pinned memory
    first makes the copy a plain DMA (non_blocking=True really is): the ring is `depth` deep because a buffer may only be reused once the
    copy that read it has completed (an event per slot)."""

    def __init__(self, device, depth=4):
        self.device, self.depth = device, depth
        self.slots = [None] * depth                        # one pinned byte buffer per slot (grown on demand)
        self.events = [None] * depth
        self.i = 0
        # the copies run on a stream of their own: on the compute stream a 0.5 MB chunk is a DMA packet BETWEEN two of the step's kernels (the queue is in
        # order: ~25 us of every 705 us step with the compute units idle, tools/step_timeline.py on the run_train loop).  QPN_STAGE_STREAM=0: the caller's stream.
        self.copy_stream = torch.cuda.Stream(device) if (torch.device(device).type == "cuda" and os.environ.get("QPN_STAGE_STREAM", "1") != "0") else None

    def __call__(self, named):
        """named: {name: cpu tensor} -> {name: device tensor}.  ONE asynchronous copy on the current stream: the tensors are packed into the slot's
        pinned buffer at 256-byte offsets and land in one device buffer the results are views of (a chunk's four small tensors as four copies cost
        four in-stream DMA packets of ~5 us each between the step's kernels; tools/runner_prof.py)."""
        k = self.i
        if self.events[k] is not None:
            self.events[k].synchronize()                  # the copy that last read this slot is done
        offs, total = {}, 0
        for name, t in named.items():
            offs[name] = total
            total += (t.numel() * t.element_size() + 255) // 256 * 256
        buf = self.slots[k]
        if buf is None or buf.numel() < total:
            buf = torch.empty(max(total, 1) * 5 // 4, dtype=torch.uint8).pin_memory()
            self.slots[k] = buf
        for name, t in named.items():
            n = t.numel() * t.element_size()
            buf[offs[name]:offs[name] + n].view(t.dtype).view(t.shape).copy_(t)
        cs = self.copy_stream
        if cs is not None:
            with torch.cuda.stream(cs):
                dbuf = buf[:total].to(self.device, non_blocking=True)
        else:
            dbuf = buf[:total].to(self.device, non_blocking=True)
        out = {name: dbuf[offs[name]:offs[name] + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for name, t in named.items()}
        ev = torch.cuda.Event()
        ev.record(cs if cs is not None else torch.cuda.current_stream(self.device))
        self.events[k] = ev
        if cs is not None:
            # whoever consumes the batch joins the copy: FusedTrainer.step / forward_loss and QPNet.forward look at ALL their inputs (train.join_staged)
            for t in out.values():
                t.__dict__["_qpn_staged"] = (ev, dbuf)
        self.i = (self.i + 1) % self.depth
        return out


class Prefetcher:
    """Runs a generator on a daemon thread, `depth` items ahead (the reference wraps its generators in a depth-2
    background queue, utils.py:165-214)."""
    _END = object()

    def __init__(self, gen, depth=2):
        depth = int(os.environ.get("QPN_PREFETCH_DEPTH", depth))
        self.q = queue.Queue(maxsize=depth)
        self.err = None
        self.t = threading.Thread(target=self._run, args=(gen,), daemon=True)
        self.t.start()

    def _run(self, gen):
        try:
            for item in gen:
                self.q.put(item)
        except BaseException as e:      # surfaced on the consumer side
            self.err = e
        self.q.put(self._END)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is self._END:
            if self.err is not None:
                raise self.err
            raise StopIteration
        return item


class _FileUtterance:
    """One (wav, feature file) pair of the corpus: called, it reads both (waveform in [-1, 1), features); `plan()` is what the sharded generator needs to lay
    the chunk stream out WITHOUT the waveform -- its length from the wav header (a memory map: no sample is read) and the features (the f0 column sets the
    dilated factors) -- so that a data-parallel rank reads the samples of the utterances its own chunks reach into only (loaders.train_generator)."""

    def __init__(self, wav, feat, feature_type):
        self.wav, self.feat, self.feature_type = wav, feat, feature_type
        self._x = self._h = None                       # memory maps of the wav samples / .npy features, opened by the first plan() / part() and kept

    def __call__(self):
        return loaders.read_wav(self.wav)[1], loaders.read_features(self.feat, self.feature_type)

    def _maps(self):
        if self._x is None:
            from scipy.io import wavfile
            self._x = wavfile.read(self.wav, mmap=True)[1]
            if self.feat.endswith(".npy"):
                self._h = np.load(self.feat, mmap_mode="r")
        return self._x, self._h

    def plan(self):
        x, h = self._maps()
        return int(x.shape[0]), (np.array(h) if h is not None else loaders.read_features(self.feat, self.feature_type))

    def part(self, s0, s1, f0, f1):
        """samples [s0, s1) as read_wav scales them and feature rows [f0, f1): the wav through a memory map (only the pages of the slice are read), .npy
        features likewise; an .h5 dataset is read whole (150 KB at 39 features x 5 ms frames) and sliced."""
        x, h = self._maps()
        xs = np.array(x[s0:s1], dtype=np.float32) / 32768
        if f1 <= f0:
            return xs, np.zeros((0, 0), dtype=np.float32)
        return xs, (np.array(h[f0:f1]) if h is not None else loaders.read_features(self.feat, self.feature_type)[f0:f1])


def _utterance_loaders(wav_list, feat_list, feature_type):
    """loaders (waveform in [-1,1), features) per utterance: files are read when the generator reaches them."""
    return [_FileUtterance(w, f, feature_type) for w, f in zip(wav_list, feat_list)]


def _batches(args, conf, model, shuffle, epochs, device, rank=0, world=1):
    """Training / validation batches on `device`.  With world > 1 rank r takes chunks r, r+N, ... of the generator's
    stream (same seed on every rank: identical shuffles and chunk boundaries) INSIDE the generator: a rank encodes,
    scales, takes ceil(max d) of and copies to its GPU only its own chunks."""
    wavs, feats = loaders.file_lists(args.waveforms, args.feats, conf.feature_format)
    logging.info("number of utterances = %d." % len(wavs))
    scaler = loaders.read_scaler_stats(args.stats, conf.feature_type)
    fs = loaders.read_wav(wavs[0])[0]
    gen = loaders.train_generator(
        _utterance_loaders(wavs, feats, conf.feature_type), model.receptiveCausal_field, model.receptiveF_field,
        model.receptiveA_field, fs, wav_transform=loaders.mu_law_transform(conf.n_quantize), feat_transform=scaler,
        dense_factor=conf.dense_factor, batch_length=args.batch_length, batch_size=args.batch_size, max_length=args.max_length,
        f0_threshold=args.f0_threshold, upsampling_factor=conf.upsampling_factor, shuffle=shuffle, epochs=epochs,
        shard=(rank, world) if world > 1 else None)

    stage = PinnedStager(device)

    def with_maxd():            # ceil(max d) is known on the host here: the fused step then needs no device read-back
        for bx, bh, bt, bd, bb in gen:
            maxd = int(np.ceil(float(bd.max())))
            dv = stage({"x": bx, "h": bh, "t": bt, "d": bd})
            yield (dv["x"], dv["h"], dv["t"], dv["d"], bb, maxd)
    return with_maxd()


def _build_model(conf, device):
    from .qpnet import QPNet, initialize
    model = QPNet(**loaders.model_kwargs(conf))
    model.apply(initialize)
    return model.to(device)


# ---------------------------------------------------------------- train / update
def _train_args(update):
    p = argparse.ArgumentParser()
    for name in ("waveforms", "feats", "stats", "expdir", "config"):
        p.add_argument("--" + name, required=True, type=str)
    if update:
        p.add_argument("--pretrain", required=True, type=str, help="SI model to adapt")
    else:
        for name, dflt in (("n_quantize", 256), ("n_aux", 39), ("n_resch", 512), ("n_skipch", 256), ("dilationF_depth", 4),
                           ("dilationF_repeat", 3), ("dilationA_depth", 4), ("dilationA_repeat", 1), ("kernel_size", 2),
                           ("dense_factor", 8), ("upsampling_factor", 110)):
            p.add_argument("--" + name, default=dflt, type=int)
        p.add_argument("--feature_type", default="world", type=str)
        p.add_argument("--feature_format", default="h5", type=str)
    p.add_argument("--batch_length", default=20000, type=int)
    p.add_argument("--batch_size", default=1, type=int)
    p.add_argument("--max_length", default=30000, type=int)
    p.add_argument("--f0_threshold", default=0, type=int)
    p.add_argument("--lr", default=1e-4, type=float)
    p.add_argument("--weight_decay", default=0.0, type=float)
    p.add_argument("--iters", default=3000 if update else 200000, type=int)
    p.add_argument("--checkpoint_interval", default=10000, type=int)
    p.add_argument("--intervals", default=100, type=int)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--resume", default=None, nargs="?", type=str)
    p.add_argument("--n_gpus", default=1, type=int)
    p.add_argument("--verbose", default=1, type=int)
    return p


def run_train(argv=None, update=False):
    argv = sys.argv[1:] if argv is None else argv
    args = _train_args(update).parse_args(argv)
    rc = launch_ranks(args.n_gpus, "qpnet_amd.run_update" if update else "qpnet_amd.run_train", argv)
    if rc is not None:
        return rc
    _setup_logging(args.verbose)
    rank, world, dev = dist_context()
    os.makedirs(args.expdir, exist_ok=True)
    _fix_seed(args.seed)                       # same seed on every rank: identical shuffles, so the chunk stream is shared
    if update:
        conf = loaders.load_model_conf(args.config)
    else:
        conf = args
        if args.upsampling_factor <= 0:
            logging.warning("upsampling_factor should larger than 0!")
            return 0
        if rank == 0:
            loaders.save_model_conf(args.config, args)
    from .train import FusedTrainer
    from . import parallel
    model = _build_model(conf, dev).train()
    trainer = FusedTrainer(model, lr=args.lr, weight_decay=args.weight_decay, world_size=world)
    iterations, loss_record = 0, []
    flossyml = os.path.join(args.expdir, "loss-final.yml")
    if args.resume and os.path.exists(args.resume):
        iterations = loaders.load_checkpoint(args.resume, model, trainer)
        logging.info("restored from %d-iter checkpoint." % iterations)
        if os.path.exists(flossyml):
            with open(flossyml, encoding="utf-8") as yf:
                loss_record = yaml.safe_load(yf) or []
    elif update:
        loaders.load_checkpoint(args.pretrain, model, None)
        logging.info("updating based on %s." % args.pretrain)
    from .train import ensure_flat
    parallel.broadcast_parameters(ensure_flat(model, dev))
    stream = Prefetcher(_batches(args, conf, model, True, None, dev, rank, world))      # rank r: chunks r, r+N, r+2N, ...
    loss = total = 0.0
    logging.info("training start!")
    # The reference reads loss.item() at every step (qpnet_train.py:533) and reports its average over `intervals` steps (:536-541).  Here a step
    # returns the loss of the step BEFORE it (copied out behind that step's kernels: the device never idles while the host enqueues) and the last
    # one of an interval is fetched at the report: the same averages, the per-batch debug line one step late.  QPN_RUN_TRAIN_SYNC_LOSS=1: in-step.
    lagged = os.environ.get("QPN_RUN_TRAIN_SYNC_LOSS", "0") != "1"
    for i in range(iterations, args.iters):
        start = time.time()
        bx, bh, bt, bd, bb, maxd = next(stream)
        batch_loss = trainer.step(bx, bh, bt, bd, bb, want_loss="lagged" if lagged else True, maxd=maxd)
        if batch_loss is not None:
            loss += batch_loss
            logging.debug("batch loss = %.3f" % batch_loss)
        total += time.time() - start
        if (i + 1) % args.intervals == 0 or (i + 1) % args.checkpoint_interval == 0 or i + 1 == args.iters:
            last = trainer.flush_loss() if lagged else None
            if last is not None:
                loss += last
            # the device-side checks of the last steps (a dilated factor outside the layer input, a target outside [0, n_quantize), an abandoned
            # stack launch: the reference asserts in every step, qpnet.py:294, qpnet_train.py:525) are collected HERE -- before an interval
            # is reported, before a checkpoint and before the final model are written: nothing flagged reaches the disk
            trainer.check_status()
        if (i + 1) % args.intervals == 0:
            logging.info("(iter:%d) average loss = %.6f (%.3f sec / batch)" % (i + 1, loss / args.intervals, total / args.intervals))
            loss_record.append(loss / args.intervals)
            loss = total = 0.0
        if (i + 1) % args.checkpoint_interval == 0 and rank == 0:
            loaders.save_checkpoint(args.expdir, model, trainer, i + 1)
    if world > 1:
        # every rank applied the same Adam step to the same summed gradient: the replicas must still be bit-identical (a rank that consumed a different
        # chunk count, a diverged knob or a lost collective shows here, before the final model is written)
        drift = parallel.replica_drift(ensure_flat(model, dev))
        if drift != 0.0:
            raise RuntimeError("data-parallel replicas drifted apart: max |w_r - w_0| = %g" % drift)
        logging.info("replicas identical on %d ranks." % world)
    if rank == 0:
        loaders.save_final(args.expdir, model)
        logging.info("final checkpoint created.")
        with open(flossyml, "w", encoding="utf-8") as yf:
            yaml.safe_dump([float(v) for v in loss_record], yf)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def run_update(argv=None):
    return run_train(argv, update=True)


# ---------------------------------------------------------------- validate
def run_validate(argv=None):
    p = argparse.ArgumentParser()
    for name in ("waveforms", "feats", "stats", "resultdir", "config", "checkpoint"):
        p.add_argument("--" + name, required=True, type=str)
    p.add_argument("--batch_length", default=20000, type=int)
    p.add_argument("--batch_size", default=1, type=int)
    p.add_argument("--max_length", default=30000, type=int)
    p.add_argument("--f0_threshold", default=0, type=int)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--n_gpus", default=1, type=int)
    p.add_argument("--verbose", default=1, type=int)
    args = p.parse_args(sys.argv[1:] if argv is None else argv)
    _setup_logging(args.verbose)
    # validation is one forward-only pass over the list (reference qpnet_validate.py:409-437): a single process.  --n_gpus is
    # accepted for command-line compatibility; under a multi-rank launcher rank 0 evaluates and writes the yml, the other
    # ranks leave at once (no process group is created, nothing to tear down)
    if int(os.environ.get("RANK", "0")) != 0:
        return 0
    if args.n_gpus > 1:
        logging.warning("validation runs on one GPU; --n_gpus %d ignored" % args.n_gpus)
    if not torch.cuda.is_available():
        raise RuntimeError("qpnet_amd needs an AMD GPU: there is no CPU fallback for the hot path")
    local = 0 if os.environ.get("QPN_BENCH_ONE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _fix_seed(args.seed)
    conf = loaders.load_model_conf(args.config)
    from .train import FusedTrainer
    model = _build_model(conf, dev).eval()
    loaders.load_checkpoint(args.checkpoint, model, None)
    logging.info("load %s." % args.checkpoint)
    trainer = FusedTrainer(model)
    flossyml = os.path.join(args.resultdir, "validation_result.yml")
    results = {}
    if os.path.exists(flossyml):
        with open(flossyml, encoding="utf-8") as yf:
            results = yaml.safe_load(yf) or {}
    loss, n = 0.0, 0
    for bx, bh, bt, bd, bb, maxd in _batches(args, conf, model, False, 1, dev):
        v = trainer.forward_loss(bx, bh, bt, bd, bb, maxd=maxd)
        loss += v; n += 1
        logging.info("(batch:%d) batch loss = %.6f" % (n, v))
    results[str(os.path.basename(args.checkpoint))] = float(loss / max(n, 1))
    os.makedirs(args.resultdir, exist_ok=True)
    with open(flossyml, "w", encoding="utf-8") as yf:
        yaml.safe_dump(results, yf)
    return 0


# ---------------------------------------------------------------- decode
def run_decode(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    p = argparse.ArgumentParser()
    for name in ("feats", "stats", "config", "outdir", "checkpoint"):
        p.add_argument("--" + name, required=True, type=str)
    p.add_argument("--fs", default=22050, type=int)
    p.add_argument("--batch_size", default=1, type=int)
    p.add_argument("--extra_memory", default=False, type=lambda s: str(s).lower() in ("1", "true", "yes", "y", "t", "on"))
    p.add_argument("--intervals", default=1000, type=int)
    p.add_argument("--seed", default=100, type=int)
    p.add_argument("--n_gpus", default=1, type=int)
    p.add_argument("--verbose", default=1, type=int)
    p.add_argument("--f0_factor", default=1.0, type=float)
    p.add_argument("--f0_dim_index", default=1, type=int)
    p.add_argument("--mode", default="sampling", choices=["sampling", "argmax"], help="the reference script always samples")
    args = p.parse_args(argv)
    rc = launch_ranks(args.n_gpus, "qpnet_amd.run_decode", argv)
    if rc is not None:
        return rc
    _setup_logging(args.verbose)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("QPN_BENCH_ONE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)            # replicas only: no process group, no collective
    dev = torch.device("cuda", local)
    outdir = os.path.dirname(args.outdir)
    if outdir:
        os.makedirs(outdir, exist_ok=True)
    _fix_seed(args.seed)
    conf = loaders.load_model_conf(args.config)
    ext = "." + conf.feature_format
    feat_list = loaders.find_files(args.feats, "*" + ext) if os.path.isdir(args.feats) else loaders.read_txt(args.feats)
    mine = np.array_split(np.array(feat_list, dtype=object), world)[rank].tolist()      # qpnet_decode.py:258-259
    scaler = loaders.read_scaler_stats(args.stats, conf.feature_type)
    with torch.no_grad():
        model = _build_model(conf, dev).eval()
        loaders.load_checkpoint(args.checkpoint, model, None)
        feats = [loaders.read_features(f, conf.feature_type) for f in mine]
        ids = [os.path.basename(f).replace(ext, "") for f in mine]
        from .qpnet import encode_mu_law
        gen = loaders.decode_generator(feats, args.fs, ids, wav_transform=lambda x: encode_mu_law(x, conf.n_quantize),
                                       feat_transform=scaler, dense_factor=conf.dense_factor, batch_size=args.batch_size,
                                       upsampling_factor=conf.upsampling_factor, f0_factor=args.f0_factor,
                                       f0_dim_index=args.f0_dim_index, extra_memory=args.extra_memory, device=dev)
        for feat_ids, bx, bh, ns, bd in gen:
            logging.info("decoding start!")
            outs = model.batch_fast_generate(bx, bh, ns, bd, intervals=args.intervals, mode=args.mode, extra_memory=args.extra_memory)
            for feat_id, samples in zip(feat_ids, outs):
                name = args.outdir.replace("feat_id", feat_id)
                loaders.write_wav(name, args.fs, samples, conf.n_quantize)
                logging.info("wrote %s." % name)
    return 0
