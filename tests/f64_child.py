"""Child process of test_full_size_gradient_error_is_fp32_reassociation_vs_a_float64_oracle: runs with QPN_LIB = the -DQPN_TESTING build
(qpnet_amd/libqpnet_hip_testing.so), whose hook qpn_test_postnet_activations hands out the rectified post-net activations of a forward -- the ReLU
sides the GPU took.  Prints one line per gradient path and F64_CHILD_OK.
    python tests/f64_child.py [bench|small]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gpu_sides(m, dev, B, BL, S):
    import torch
    from qpnet_amd import _lib
    L, hd = m._native(dev)
    fn = L.qpn_test_postnet_activations
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    s0 = torch.empty((B, BL, S), dtype=torch.float32, device=dev); y0 = torch.empty_like(s0)
    _lib.check(fn(hd, s0.data_ptr(), y0.data_ptr(), s0.numel(), torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    return (s0 > 0).cpu().numpy(), (y0 > 0).cpu().numpy()


def main(which):
    import torch
    import util
    from oracle import train_oracle as TO
    from qpnet_amd import synth
    from qpnet_amd.config import PAPER
    from qpnet_amd.train import FusedTrainer
    dev = torch.device("cuda:0")
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    if which == "bench":          # bench.py's chunk 0 (BASELINE config[1]): 20 900 samples, 19 954 rows
        x, h, t, d, b = synth.train_inputs(cfg, 20000, 5000, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True)
    else:
        x, h, t, d, b = synth.train_inputs(cfg, 1500, 61, 30000)
    BL = int(b[0]); maxd = int(np.ceil(d).max())
    to = lambda *a: [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in a]
    xt, ht, tt, dt = to(x, h, t, d)
    # float32 numpy oracle and the float64 yardstick on ITS sides
    lg, c32 = TO.forward(cfg, flat, x, h, d, b)
    _, dl = TO.ce_loss(lg, t[:, -BL:])
    g32 = TO.backward(cfg, flat, c32, dl).astype(np.float64)
    with TO.precision(np.float64):
        f64 = flat.astype(np.float64)
        lg64, c64 = TO.forward(cfg, f64, x, h.astype(np.float64), d, b)
        _, dl64 = TO.ce_loss(lg64, t[:, -BL:])
        pre = [{k: c[k].copy() for k in ("s0", "y0")} for c in c64]          # the float64 pre-activations (the sides are imposed below)

        def yardstick(sides):
            """float64 gradient with the given ReLU sides [(s0 > 0, y0 > 0) per batch row]; returns it and the largest |float64 pre-activation| among
            the units whose side had to be changed (must be rounding-level: otherwise a side differs for a reason that is not rounding)."""
            moved = 0.0
            for c, p0, (ms, my) in zip(c64, pre, sides):
                for key, mk in (("s0", ms), ("y0", my)):
                    v = p0[key].copy()
                    flip = (v > 0) != mk
                    if flip.any():
                        moved = max(moved, float(np.abs(v[flip]).max()))
                    v[flip] = np.where(mk[flip], 1e-300, -1e-300)
                    c[key] = v
            return TO.backward(cfg, f64, c64, dl64), moved
        g64_o, moved_o = yardstick([(c["s0"] > 0, c["y0"] > 0) for c in c32])
        # the fused step (what bench.py times) and the autograd path, each with the float64 yardstick on the sides THAT forward took
        runs = {}
        m = util.build_model(cfg, flat, dev).train()
        tr = FusedTrainer(m, lr=1e-4)
        tr.step(xt, ht, tt, dt, b, want_loss=False, maxd=maxd)
        tr.check_status()
        ms, my = gpu_sides(m, dev, 1, BL, cfg.n_skipch)
        runs["fused"] = (tr.g[:flat.size].cpu().numpy().astype(np.float64), [(ms[0], my[0])])
        m2 = util.build_model(cfg, flat, dev).train()
        logits = m2(xt, ht, dt, torch.from_numpy(b))
        ms, my = gpu_sides(m2, dev, 1, BL, cfg.n_skipch)
        torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1)).backward()
        runs["autograd"] = (torch.cat([p.grad.reshape(-1) for p in m2.parameters()]).cpu().numpy().astype(np.float64), [(ms[0], my[0])])
        offs, _ = cfg.param_offsets()
        floor = 1e-6 * np.abs(g64_o).max()
        ok = moved_o < 1e-5
        for name, (g, sides) in runs.items():
            g64, moved = yardstick(sides)
            nflip = int(sum(((a != (c["s0"] > 0)).sum() + (bb != (c["y0"] > 0)).sum()) for (a, bb), c in zip(sides, c32)))
            ratios = []
            for k, (o, shp) in offs.items():
                n = int(np.prod(shp))
                e_gpu = np.abs(g[o:o + n] - g64[o:o + n]).max()
                e_o32 = np.abs(g32[o:o + n] - g64_o[o:o + n]).max()
                ratios.append((e_gpu / (e_o32 + floor), k, e_gpu, e_o32))
            w = max(ratios)
            rel = np.abs(g - g64).max() / np.abs(g64).max()
            print("F64 %s %s: worst tensor %s |g_gpu - g64| %.3e vs |g_o32 - g64| %.3e ratio %.2f; median ratio %.2f; max |g_gpu - g64| / max |g64| %.2e "
                  "(oracle32: %.2e); %d post-net units on the other side than the float32 oracle's, largest float64 pre-activation among the moved sides %.2e"
                  % (which, name, w[1], w[2], w[3], w[0], float(np.median([r[0] for r in ratios])), rel, np.abs(g32 - g64_o).max() / np.abs(g64_o).max(), nflip, moved))
            ok = ok and w[0] <= 4.0 and moved < 1e-5
    print("F64_CHILD_OK" if ok else "F64_CHILD_FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "bench"))
