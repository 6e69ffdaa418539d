"""Child process of the two give-up tests: runs with QPN_LIB = the -DQPN_TESTING build (qpnet_amd/libqpnet_hip_testing.so), the only
build that contains the fault-injection hooks, and the hook's environment variable set by the parent.
    python tests/giveup_child.py stack|pipe|coopb"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stack(cuda):
    """Every wait of the one-launch residual stack is bounded.  With the published flags made unrecognisable (the situation of a peer
    workgroup that never runs) the first dependent tile's wait runs out, the launch drains, the step is REPORTED as invalid (status bit 4 ->
    QPN_ENODEV) -- and the handle runs a launch per layer from then on: the next forward is correct."""
    import torch
    import util
    from qpnet_amd import _lib, synth
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    x, h, t, d, b = synth.train_inputs(cfg, 1500, 91, 30000)
    xt, ht, dt, bt = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in (x, h, d, b)]
    os.environ.pop("QPN_TEST_STACK_GIVES_UP")
    ref = util.build_model(cfg, flat, cuda)
    with torch.no_grad():
        good = ref(xt, ht, dt, bt).cpu().numpy()
    os.environ["QPN_TEST_STACK_GIVES_UP"] = "1"          # (read when the handle's training state is created: at m's first forward)
    m = util.build_model(cfg, flat, cuda)
    with torch.no_grad():
        try:
            m(xt, ht, dt, bt)
            raise SystemExit("the forward did not raise")
        except _lib.QpnError as e:
            assert e.code == -2 and "residual stack" in str(e), str(e)
        again = m(xt, ht, dt, bt).cpu().numpy()           # the hook is still armed: this forward no longer uses the queue
    assert np.array_equal(again.view(np.uint32), good.view(np.uint32))


def pipe(cuda):
    """A multi-workgroup decode launch whose workgroups are not co-resident (CU-masked / shared GPU) times out and drains; the call is then
    re-run on the one-CU kernel instead of failing with QPN_ENODEV.  The give-up is injected."""
    import torch
    import util
    from oracle import cpu_oracle as oracle
    from qpnet_amd import synth
    from qpnet_amd.config import PAPER
    cfg = PAPER
    specs = [(300 + b, 6 + (b * 5) % 9, [1.0, 0.5, 1.5][b % 3]) for b in range(5)]      # (= test_decode_gpu._paper_batch(5))
    flat = synth.make_weights(cfg, 13)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = synth.decode_batch(cfg, specs)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="argmax")
    assert "timed out, retried" in m.last_decode_plan and "pipe rows=0" in m.last_decode_plan.split("retried:")[1], m.last_decode_plan
    order = np.argsort(ns, kind="stable")
    maxd = int(np.ceil(np.nanmax(bd)))
    for k in range(5):
        bb = int(order[k])
        x, h, d, n = synth.decode_inputs(cfg, specs[bb][1], specs[bb][0], specs[bb][2])
        np.testing.assert_array_equal(outs[k], oracle.decode(cfg, flat, h, d, x, n, maxd=maxd)["samples"])


def coopb(cuda):
    """The utterance-batched cooperative launch (decode_coopb.hip) that gives up is re-run on the per-utterance cooperative kernel (half the
    workgroups per utterance), and the samples are the oracle's."""
    import torch
    import util
    from oracle import cpu_oracle as oracle
    from qpnet_amd import synth
    from qpnet_amd.config import DEFAULT
    cfg = DEFAULT
    os.environ.pop("QPN_DECODE_COOPB", None)
    specs = [(21, 1, 1.0), (22, 2, 1.5), (23, 1, 0.5)]
    flat = synth.make_weights(cfg, 17)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = synth.decode_batch(cfg, specs)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="argmax")
    plan = m.last_decode_plan
    assert plan.startswith("coopb ") and "timed out, retried: coop G=" in plan, plan
    o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd)
    for a, b in zip(outs, o_outs):
        np.testing.assert_array_equal(a, b)


if __name__ == "__main__":
    import torch
    assert torch.cuda.is_available()
    from qpnet_amd import _lib
    assert _lib.LIB_PATH.endswith("libqpnet_hip_testing.so"), _lib.LIB_PATH
    {"stack": stack, "pipe": pipe, "coopb": coopb}[sys.argv[1]](torch.device("cuda:0"))
    print("GIVEUP_CHILD_OK", sys.argv[1])
