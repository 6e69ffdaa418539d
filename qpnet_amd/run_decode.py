"""python -m qpnet_amd.run_decode -- counterpart of the reference's src/bin/qpnet_decode.py over the native hot path (see runners.py)."""
import sys

from .runners import run_decode

if __name__ == "__main__":
    sys.exit(run_decode())
