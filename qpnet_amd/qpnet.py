"""Host-side mirror of the reference module ``src/nets/qpnet.py`` (bigpon/QPNet).

Same public names, constructor kwargs, attributes, state_dict keys/shapes and call
signatures, but NO torch compute op on the hot path: ``forward`` and
``batch_fast_generate`` hand raw device pointers to the hand-written HIP kernels of
``libqpnet_hip.so`` through the C ABI in ``include/qpnet_hip.h``.  torch is used for device
memory, streams and autograd plumbing only.  There is no CPU fallback: calling the hot path
without the built library or without a GPU raises.

    encode_mu_law / decode_mu_law   <- reference qpnet.py:22-45   (host numpy, as in the reference)
    initialize                      <- reference qpnet.py:47-58
    QPNet                           <- reference qpnet.py:160-686
"""
import logging
import sys
import time

import numpy as np
import torch
from torch import nn

from .config import QPNetConfig
from . import _lib


def encode_mu_law(x, mu=256):
    """mu-law companding to integer classes 0..mu-1 (reference qpnet.py:22-32)."""
    mu = mu - 1
    fx = np.sign(x) * np.log(1 + mu * np.abs(x)) / np.log(1 + mu)
    return np.floor((fx + 1) / 2 * mu + 0.5).astype(np.int64)


def decode_mu_law(y, mu=256):
    """inverse of encode_mu_law (reference qpnet.py:34-45)."""
    mu = mu - 1
    fx = (y - 0.5) / mu * 2 - 1
    return np.sign(fx) / mu * ((1 + mu) ** np.abs(fx) - 1)


def initialize(m):
    """Xavier-uniform conv weights, zero bias; upsampling kernel 1 / bias 0 (reference qpnet.py:47-58).
    Works with ``model.apply(initialize)`` because parameters are held by stock nn.Conv1d /
    nn.ConvTranspose2d containers (holders only - their forward is never called)."""
    if isinstance(m, nn.Conv1d):
        nn.init.xavier_uniform_(m.weight)
        nn.init.constant_(m.bias, 0.0)
    if isinstance(m, nn.ConvTranspose2d):
        nn.init.constant_(m.weight, 1.0)
        nn.init.constant_(m.bias, 0.0)


class _TapConv(nn.Module):
    """parameter holder: ``.conv`` = 2-tap (dilated) causal conv weights (C_out, C_in, 2)."""

    def __init__(self, cin, cout, ksize, dilation=1):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, ksize, padding=0, dilation=dilation)


class _PitchTapPair(nn.Module):
    """parameter holder: ``.convC`` (current sample) and ``.convP`` (pitch-dependent past sample)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.convC = nn.Conv1d(cin, cout, 1)
        self.convP = nn.Conv1d(cin, cout, 1)


class _FrameUpsampler(nn.Module):
    """parameter holder: ``.conv`` = (1,1,1,U) transposed-conv kernel + scalar bias."""

    def __init__(self, factor):
        super().__init__()
        self.conv = nn.ConvTranspose2d(1, 1, kernel_size=(1, factor), stride=(1, factor))


class QPNet(nn.Module):
    """Quasi-Periodic WaveNet vocoder; drop-in for the reference class (qpnet.py:160)."""

    def __init__(self, n_quantize=256, n_aux=39, n_resch=512, n_skipch=256,
                 dilationF_depth=4, dilationF_repeat=3, dilationA_depth=4, dilationA_repeat=1,
                 kernel_size=2, upsampling_factor=110):
        super().__init__()
        cfg = QPNetConfig(n_quantize, n_aux, n_resch, n_skipch, dilationF_depth, dilationF_repeat,
                          dilationA_depth, dilationA_repeat, kernel_size, upsampling_factor)
        # unsupported geometries fail HERE, not at the first forward / decode (the library validates without a GPU)
        import ctypes as _C
        if _lib.lib().qpn_param_count(_C.byref(_lib.make_config(cfg))) < 0:
            raise ValueError("qpnet_amd.QPNet: " + _lib.lib().qpn_last_error().decode("utf-8", "replace"))
        self.cfg = cfg
        self.n_quantize, self.n_aux, self.n_resch, self.n_skipch = n_quantize, n_aux, n_resch, n_skipch
        self.kernel_size, self.upsampling_factor = kernel_size, upsampling_factor
        self.dilationF_depth, self.dilationF_repeat = dilationF_depth, dilationF_repeat
        self.dilationA_depth, self.dilationA_repeat = dilationA_depth, dilationA_repeat
        self.receptiveCausal_field = cfg.receptiveCausal_field
        self.dilationsF = cfg.dilationsF
        self.receptiveF_field = cfg.receptiveF_field
        self.dilationsA = cfg.dilationsA
        self.receptiveA_field = cfg.receptiveA_field
        self.n_ch = n_resch
        C, S, Q, A = n_resch, n_skipch, n_quantize, n_aux
        # registration order == state_dict order == flat parameter order (config.param_layout)
        self.causal = _TapConv(Q, C, kernel_size)
        if upsampling_factor > 0:
            self.upsampling = _FrameUpsampler(upsampling_factor)
        self.dilF_sigmoid = nn.ModuleList(_TapConv(C, C, kernel_size, d) for d in self.dilationsF)
        self.dilF_tanh = nn.ModuleList(_TapConv(C, C, kernel_size, d) for d in self.dilationsF)
        self.auxF_1x1_sigmoid = nn.ModuleList(nn.Conv1d(A, C, 1) for _ in self.dilationsF)
        self.auxF_1x1_tanh = nn.ModuleList(nn.Conv1d(A, C, 1) for _ in self.dilationsF)
        self.skipF_1x1 = nn.ModuleList(nn.Conv1d(C, S, 1) for _ in self.dilationsF)
        self.resF_1x1 = nn.ModuleList(nn.Conv1d(C, C, 1) for _ in self.dilationsF)
        self.dilA_sigmoid = nn.ModuleList(_PitchTapPair(C, C) for _ in self.dilationsA)
        self.dilA_tanh = nn.ModuleList(_PitchTapPair(C, C) for _ in self.dilationsA)
        self.auxA_1x1_sigmoid = nn.ModuleList(nn.Conv1d(A, C, 1) for _ in self.dilationsA)
        self.auxA_1x1_tanh = nn.ModuleList(nn.Conv1d(A, C, 1) for _ in self.dilationsA)
        self.skipA_1x1 = nn.ModuleList(nn.Conv1d(C, S, 1) for _ in self.dilationsA)
        self.resA_1x1 = nn.ModuleList(nn.Conv1d(C, C, 1) for _ in self.dilationsA)
        self.conv_post_1 = nn.Conv1d(S, S, 1)
        self.conv_post_2 = nn.Conv1d(S, Q, 1)
        assert [k for k, _ in self.named_parameters()] == [k for k, _ in cfg.param_layout()]
        self._handle = None
        self._handle_dev = None
        self._qpn_sync_status = False
        self.status_check = "backward"     # "backward": loss.backward() waits for the forward's device-side check (raised before ANY optimizer steps); "lazy": never waits (train.QPNetFunction.backward)
        self.last_decode_kernel_ms = 0.0
        self.last_decode_plan = ""
        self.sampling_seed = None          # set to an int to pin the sampling-mode random stream
        self._n_generate_calls = 0
        self.last_sampling_seed = None

    # ------------------------------------------------------------------ native handle
    def _native(self, device):
        """libqpnet_hip handle bound to `device` (created lazily, one per module)."""
        if device.type != "cuda":
            raise RuntimeError("qpnet_amd.QPNet runs on an AMD GPU only (tensors are on %s); there is no CPU fallback" % device)
        L = _lib.lib()
        if self._handle is None or self._handle_dev != device:
            self._release()
            import ctypes as C
            hp = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(L.qpn_create(C.byref(_lib.make_config(self.cfg)), C.byref(hp)))
            self._handle, self._handle_dev = hp, device
        return L, self._handle

    def _replicate_for_data_parallel(self):
        """torch.nn.DataParallel with more than one device (the reference wraps the model so, src/bin/qpnet_train.py:416-423, but forces one GPU:
        runQP.py:84-88) would call this once per replica.  The module's state -- the native handle, its workspace with the forward's activations, the
        flat parameter buffer the parameters are views of -- belongs to ONE device and is not carried by replicate(): refuse instead of training on
        replicas that silently share or lose it.  Multi-GPU here is one process per GPU (python -m qpnet_amd.run_train --n_gpus N: RCCL all-reduce of
        the flat gradient); DataParallel over a single device never replicates and works."""
        raise RuntimeError("qpnet_amd.QPNet cannot be replicated by torch.nn.DataParallel over several devices (its native handle and flat parameter "
                           "buffer are per-device state); use one process per GPU: python -m qpnet_amd.run_train --n_gpus N")

    def check_status(self):
        """Raise what the device-side checks of the training forwards so far have found (a dilated factor outside the layer input --
        the reference's gather assert, qpnet.py:294 -- or a target outside [0, n_quantize)); collected without a stream drain.
        backward(), FlatAdam.step() and the next forward call it too; call it after the last forward of a loop that has none of them."""
        if self._handle is not None:
            with torch.cuda.device(self._handle_dev):
                _lib.check(_lib.lib().qpn_train_status_collect(self._handle))

    def _release(self):
        if self._handle is not None:
            try:
                self.check_status()
            except Exception as e:          # (a destructor must not raise: say it)
                import logging
                logging.warning("qpnet_amd: unreported device-side status at release: %s", e)
            _lib.lib().qpn_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def flat_parameters(self):
        """fp32 parameter vector in state_dict order (the layout qpn_set_weights expects)."""
        return torch.cat([p.detach().reshape(-1) for p in self.parameters()]).float().contiguous()

    def _bind_decode_weights(self, L, hd, dev, stream):
        """qpn_set_weights on the model's flat parameter buffer.  The parameters are views of ONE flat buffer
        (train.ensure_flat: built once, re-built only if .to()/.cuda() replaced the storages), so a decode call costs no
        2-96 MB concatenation; the library's tile re-pack is stream-ordered with no host synchronisation (one gather pass
        over the weights, ~0.1 ms for the 96 MB geometry) and is done on every call -- cheaper than any scheme that has to
        notice writes made through `p.data`, which torch's version counters do not see."""
        from .train import ensure_flat
        flat = ensure_flat(self, dev)
        _lib.check(L.qpn_set_weights(hd, flat.data_ptr(), flat.numel(), stream))
        return flat

    # ------------------------------------------------------------------ training forward
    def forward(self, x, h, dilated_factors, blength):
        """Teacher-forced forward (reference qpnet.py:239-312): x (B,T) long, h (B,n_aux,F),
        dilated_factors (B,T) float, blength (B) -> (B, batch_length, n_quantize) logits."""
        from . import train   # native training path (autograd.Function over the C ABI)
        return train.qpnet_forward(self, x, h, dilated_factors, blength)

    # ------------------------------------------------------------------ autoregressive decode
    def batch_fast_generate(self, x, h, n_samples_list, dilated_factors,
                            intervals=None, mode="sampling", extra_memory=False):
        """Batch fast generation (reference qpnet.py:314-559).

        x (B,T0) long seed, h (B,n_aux,F) float, n_samples_list list[int] (consumed exactly like
        the reference does), dilated_factors (B,T): numpy float64 when extra_memory is False,
        float tensor otherwise.  Returns the list of int64 ndarrays in completion order
        (ascending length, ties in input order)."""
        if mode not in ("sampling", "argmax"):
            logging.error("mode should be sampling or argmax")
            sys.exit(1)
        import ctypes as C
        dev = x.device
        L, hd = self._native(dev)
        B = len(n_samples_list)
        assert x.shape[0] == B and h.shape[0] == B
        # maxd exactly as the reference computes it (qpnet.py:347-350)
        if extra_memory:
            maxd = int(torch.max(dilated_factors.ceil()))
            d_dev = dilated_factors.to(dev, torch.float32).contiguous()
            d_is_f32 = 1
        else:
            dnp = np.asarray(dilated_factors)
            maxd = int(np.nanmax(np.ceil(dnp)))
            d_dev = torch.from_numpy(np.ascontiguousarray(dnp, dtype=np.float64)).to(dev)
            d_is_f32 = 0
        assert d_dev.dim() == 2 and d_dev.shape[0] == B
        xd = x.to(dev, torch.int64).contiguous()
        hd_ = h.to(dev, torch.float32).contiguous()
        assert hd_.shape[1] == self.n_aux
        ns = [int(n) for n in n_samples_list]
        max_n = max(ns)
        out = torch.empty((B, max(max_n, 1)), dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        # sampling: a counter-based generator keyed by torch's seed (torch.manual_seed(args.seed) in the reference
        # decode script, qpnet_decode.py:236-239) plus a per-call counter; draws are reproducible, but they are not
        # torch.distributions.Categorical's stream (parity with the reference is statistical in this mode)
        seed = self.sampling_seed if self.sampling_seed is not None else (torch.initial_seed() + self._n_generate_calls) % (1 << 64)
        self._n_generate_calls += 1
        self.last_sampling_seed = seed
        t_start = time.time()
        with torch.cuda.device(dev):
            self._bind_decode_weights(L, hd, dev, stream)
            arr = (C.c_int64 * B)(*ns)
            _lib.check(L.qpn_decode(hd, B, xd.shape[1], hd_.shape[2], d_dev.shape[1],
                                    xd.data_ptr(), hd_.data_ptr(), d_dev.data_ptr(), d_is_f32,
                                    arr, maxd, 1 if mode == "sampling" else 0, seed, None, out.data_ptr(), None, stream))
            self.last_decode_kernel_ms = float(L.qpn_last_decode_kernel_ms(hd))
            self.last_decode_plan = L.qpn_last_decode_plan(hd).decode("utf-8", "replace")
        out_np = out.cpu().numpy()
        if intervals is not None and intervals > 0 and max_n > 0:
            # progress lines of the reference loop (qpnet.py:519-524).  The whole call is one persistent launch, so they are
            # written once it has returned, with the measured mean time per sample (the estimate the reference prints is
            # the same quantity taken over the last `intervals` samples)
            per = (time.time() - t_start) / max_n
            for i in range(intervals, max_n + 1, intervals):
                logging.info("%d/%d estimated time = %.3f sec (%.3f sec / sample)" % (i, max_n, (max_n - i) * per, per))
        # completion order + in-place consumption of n_samples_list (reference qpnet.py:527-557)
        order = sorted(range(B), key=lambda i: ns[i])
        result = [out_np[i, :ns[i]].copy() for i in order]
        keep = ns[order[-1]]
        del n_samples_list[:]
        n_samples_list.append(keep)
        return result

    # test/diagnostic helper (not part of the reference surface): teacher-forced streaming logits
    def _stream_logits(self, x, h, dilated_factors, teacher, n_samples):
        import ctypes as C
        dev = x.device
        L, hd = self._native(dev)
        dnp = np.asarray(dilated_factors)
        maxd = int(np.nanmax(np.ceil(dnp)))
        d_dev = torch.from_numpy(np.ascontiguousarray(dnp, dtype=np.float64)).to(dev)
        B = x.shape[0]
        out = torch.empty((B, n_samples), dtype=torch.int64, device=dev)
        logits = torch.empty((B, n_samples, self.n_quantize), dtype=torch.float32, device=dev)
        tch = teacher.to(dev, torch.int64).contiguous()
        xd = x.to(dev, torch.int64).contiguous(); hh = h.to(dev, torch.float32).contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            self._bind_decode_weights(L, hd, dev, stream)
            arr = (C.c_int64 * B)(*([n_samples] * B))
            _lib.check(L.qpn_decode(hd, B, xd.shape[1], hh.shape[2], d_dev.shape[1], xd.data_ptr(), hh.data_ptr(),
                                    d_dev.data_ptr(), 0, arr, maxd, 0, 0, tch.data_ptr(), out.data_ptr(),
                                    logits.data_ptr(), stream))
        return out, logits
