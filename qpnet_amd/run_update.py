"""python -m qpnet_amd.run_update -- counterpart of the reference's src/bin/qpnet_update.py over the native hot path (see runners.py)."""
import sys

from .runners import run_update

if __name__ == "__main__":
    sys.exit(run_update())
