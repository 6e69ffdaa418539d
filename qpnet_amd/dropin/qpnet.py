"""Drop-in replacement for the reference's `src/nets/qpnet.py`.

The reference's task scripts do `from qpnet import encode_mu_law, decode_mu_law, initialize, QPNet`
with PYTHONPATH=src/utils:src/nets (reference src/runQP.py:81-82, bin/qpnet_train.py:35-37).
Put THIS directory in front of src/nets on PYTHONPATH and every stage of
run_QP.sh picks up the MI355X-native module unchanged."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from qpnet_amd.qpnet import QPNet, encode_mu_law, decode_mu_law, initialize  # noqa: E402,F401
