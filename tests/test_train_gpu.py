"""GPU parity of the training path: HIP forward / backward / CE / Adam vs the numpy oracle
(itself pinned to the reference's autograd by tests/golden/train.npz) and vs the fixtures."""
import numpy as np
import pytest

from cases import FORWARD_CASES, TRAIN_CASES
from qpnet_amd import synth
import util

pytestmark = pytest.mark.gpu


def _to(dev, *arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


@pytest.mark.parametrize("case", FORWARD_CASES, ids=[c[0] for c in FORWARD_CASES])
def test_forward_logits_vs_reference(case, cuda, golden_dir):
    import torch
    name, cfg, wseed, dseed, bl, ml = case
    g = np.load(golden_dir + "/forward.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda)
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, ml)
    xt, ht, dt, bt = _to(cuda, x, h, d, b)
    with torch.no_grad():
        logits = m(xt, ht, dt, bt)
    lg = logits.cpu().numpy()
    ref = g[name + "_logits"]
    assert lg.shape == ref.shape
    np.testing.assert_allclose(lg, ref, atol=2e-5, rtol=0)      # fp32 reassociation only
    BL = int(b[0])
    lse = np.log(np.exp(lg[0].astype(np.float64)).sum(1))
    loss = (lse - lg[0][np.arange(BL), t[0, -BL:]]).mean()
    assert abs(loss - float(g[name + "_loss"])) < 1e-4          # north_star tolerance
