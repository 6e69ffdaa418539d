"""python -m qpnet_amd.run_train -- counterpart of the reference's src/bin/qpnet_train.py over the native hot path (see runners.py)."""
import sys

from .runners import run_train

if __name__ == "__main__":
    sys.exit(run_train())
