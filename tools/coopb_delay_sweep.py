"""Dev: sweep of the first-poll delay of k_decode_coopb's layer gathers (QPN_COOPB_DELAY_G / _X / _T, units of s_sleep(2) = 128 clocks):
    python tools/coopb_delay_sweep.py <B> [g:x:t,g:x:t,...]."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else "20"
GRID = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [(0, 0, 0), (4, 0, 0), (8, 0, 0), (12, 0, 0), (0, 4, 0), (0, 8, 0), (8, 4, 0), (12, 8, 0), (16, 8, 0)]
for g in GRID:
    dg, dx, dt = (g + (0,))[:3]
    env = dict(os.environ, QPN_COOPB_DELAY_G=str(dg), QPN_COOPB_DELAY_X=str(dx), QPN_COOPB_DELAY_T=str(dt))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "coopb_phases.py"), B, "100"], env=env, capture_output=True, text=True, timeout=300)
    line = [x for x in r.stdout.splitlines() if "samples/s" in x]
    print("delay g=%2d x=%2d t=%2d: %s" % (dg, dx, dt, line[-1] if line else r.stderr[-300:]), flush=True)
