"""Helpers shared by the parity tests (inputs regenerated from seeds via qpnet_amd.synth)."""
import numpy as np
from qpnet_amd import synth, harness


decode_batch = synth.decode_batch


def build_model(cfg, flat, device):
    import torch
    from qpnet_amd.qpnet import QPNet
    m = QPNet(**cfg.kwargs())
    sd = {k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()}
    m.load_state_dict(sd)
    return m.to(device).eval()


def run_giveup_child(which, env_name):
    """The fault-injection hooks exist only in the -DQPN_TESTING build of the library (qpnet_amd/libqpnet_hip_testing.so, built by
    __graft_entry__.build()): the scenario runs in a child process bound to that build (tests/giveup_child.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "qpnet_amd", "libqpnet_hip_testing.so")
    assert os.path.exists(lib), "build the testing library first: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, QPN_LIB=lib, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env[env_name] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "giveup_child.py"), which], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GIVEUP_CHILD_OK " + which in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
