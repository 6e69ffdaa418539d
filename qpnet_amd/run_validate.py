"""python -m qpnet_amd.run_validate -- counterpart of the reference's src/bin/qpnet_validate.py over the native hot path (see runners.py)."""
import sys

from .runners import run_validate

if __name__ == "__main__":
    sys.exit(run_validate())
