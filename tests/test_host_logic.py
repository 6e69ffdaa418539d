"""CPU: host-side mirror of the reference interface (module surface, harness helpers)."""
import numpy as np
import pytest

from qpnet_amd import harness, synth
from qpnet_amd.config import TINY, PAPER, DEFAULT


def test_dilated_factor_kat():
    # SURVEY §8a a18 [probed on the reference]
    d = harness.dilated_factor(np.array([0.0, 100.0, 200.0, 45.0]), 22050, 8)
    np.testing.assert_allclose(d, [1.0, 27.5625, 13.78125, 61.25])
    assert harness.receptive_field(1, 45, 15, d) == 976


def test_validate_length_kat():
    x, h = harness.validate_length(np.zeros(1000), np.zeros((10, 39)), 110)
    assert x.shape[0] == 990 and h.shape[0] == 9


def test_extend_time_is_repeat():
    f = np.arange(6.0).reshape(3, 2)
    e = harness.extend_time(f, 4)
    assert e.shape == (12, 2) and (e[4:8] == f[1]).all()


def test_receptive_fields():
    assert (PAPER.receptiveF_field, PAPER.receptiveA_field, PAPER.receptiveCausal_field) == (15, 15, 1)
    assert DEFAULT.receptiveF_field == 45 and DEFAULT.receptive_field(62) == 976
    assert PAPER.receptive_field(62) == 946


def test_chunk_geometry_paper():
    # max d 61.25 -> maxd 62, RF 946, chunk % 110 == 0
    d = np.full(30000, 61.25)
    rf, bl, h_bs, x_bs = harness.train_chunk_geometry(PAPER, d, 20000, 30000)
    assert rf == 946 and (rf + bl) % 110 == 0 and bl <= 20000 and x_bs == h_bs * 110 + 1


def test_module_surface_and_state_dict():
    import torch
    from qpnet_amd.qpnet import QPNet, initialize
    m = QPNet(**TINY.kwargs())
    keys = list(m.state_dict().keys())
    assert keys == [k for k, _ in TINY.param_layout()]
    for k, shp in TINY.param_layout():
        assert tuple(m.state_dict()[k].shape) == shp
    assert (m.receptiveCausal_field, m.receptiveF_field, m.receptiveA_field) == (1, 3, 1)
    assert m.dilationsF == [1, 2] and m.dilationsA == [1] and m.n_quantize == 256 and m.upsampling_factor == 110
    m.apply(initialize)
    assert float(m.upsampling.conv.weight.min()) == 1.0 and float(m.causal.conv.bias.abs().max()) == 0.0
    w = m.dilF_sigmoid[0].conv.weight
    bound = np.sqrt(6.0 / (32 * 2 + 32 * 2))
    assert float(w.abs().max()) <= bound + 1e-6
    flat = synth.make_weights(TINY, 5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(TINY, flat).items()})
    np.testing.assert_array_equal(m.flat_parameters().numpy(), flat)


def test_bad_mode_exits():
    import torch
    from qpnet_amd.qpnet import QPNet
    m = QPNet(**TINY.kwargs())
    with pytest.raises(SystemExit):
        m.batch_fast_generate(torch.zeros(1, 1, dtype=torch.long), torch.zeros(1, 39, 4), [10], np.ones((1, 440)), mode="bogus")


def test_synth_is_deterministic():
    a = synth.make_features(50, 3); b = synth.make_features(50, 3)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (50, 39) and a[:, 1].min() >= 45 and a[:, 1].max() <= 450
    assert set(np.unique(a[:, 0])) <= {0.0, 1.0}


def test_data_parallel_replicas_are_refused():
    """torch.nn.DataParallel over several devices would replicate the module (reference wrap: src/bin/qpnet_train.py:416-423, forced to one GPU by
    runQP.py:84-88); the native handle and the flat parameter buffer are per-device state replicate() does not carry: a clear error, not silence."""
    import pytest
    import torch
    from qpnet_amd.config import TINY
    from qpnet_amd.qpnet import QPNet
    m = QPNet(**TINY.kwargs())
    wrapped = torch.nn.DataParallel(m)                  # wrapping (and a single device, which never replicates) stays possible
    assert wrapped.module is m
    with pytest.raises(RuntimeError, match="one process per GPU"):
        m._replicate_for_data_parallel()                # what torch.nn.parallel.replicate calls for every replica


def test_optimizer_step_hooks_leave_other_modules_alone():
    """qpnet_amd.train registers global torch.optim step hooks (the reference loop's stock Adam is stepped by the library, train._adam_prehook): an Adam over
    any other module's parameters -- or an optimizer of another class -- is looked at once and never again, and steps exactly as without the hooks."""
    import torch
    import qpnet_amd.train  # noqa: F401  (registers the hooks)
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    ref = torch.nn.Linear(5, 3)
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(7, 5)
    opt = torch.optim.Adam(lin.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad(); lin(x).pow(2).sum().backward(); opt.step()
    assert opt.__dict__.get("_qpn_adopt") is False and len(opt.param_groups[0]["params"]) == 2
    # the same three steps computed by hand (torch's documented update rule)
    ps = list(ref.parameters()); m = [torch.zeros_like(p) for p in ps]; v = [torch.zeros_like(p) for p in ps]
    for k in range(1, 4):
        for p in ps:
            p.grad = None
        ref(x).pow(2).sum().backward()
        with torch.no_grad():
            for p, mi, vi in zip(ps, m, v):
                mi.mul_(0.9).add_(p.grad, alpha=0.1); vi.mul_(0.999).addcmul_(p.grad, p.grad, value=0.001)
                p.addcdiv_(mi / (1 - 0.9 ** k), (vi / (1 - 0.999 ** k)).sqrt() + 1e-8, value=-1e-2)
    for a, b in zip(lin.parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=1e-6, rtol=0)
    sgd = torch.optim.SGD(lin.parameters(), lr=0.1)
    sgd.zero_grad(); lin(x).sum().backward(); sgd.step()
    assert "_qpn_adopt" not in sgd.__dict__
