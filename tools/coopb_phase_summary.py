"""Dev: turn the stamp lines of tools/coopb_phases.py logs (stderr of a -DQPN_ENABLE_STAMPS library) into microseconds per phase.
    python tools/coopb_phase_summary.py gpurun_out/ph4.log gpurun_out/ph20.log ..."""
import re
import sys

NAMES = ["P1 tags, tap distances", "P2 layer-0 input published, aux(0)", "P3 gather x(0) [+ past rows 0, 1: compute waves]",
         "S1 current-row dot (until barrier A)", "S2 gate: close the tree, qgate, publish", "S2 gather g [+ next past-row dot: compute waves] (until B)",
         "S3 residual + skip dot (until C)", "S4 block output: close the tree, publish", "S4 gather x(l+1) [+ past rows l+2: compute waves] (until D)",
         "tail: y1 / post1 / y2 / post2 / logits (3 edges, 2 dots)", "pick",
         "  compute wave, S3: the dot product", "  compute wave, S3: issue the next fragments", "  compute wave, S3: partial tile -> LDS", "  compute wave, S3: wait at C"]
PER_LAYER = {3, 4, 5, 6, 7, 8, 11, 12, 13, 14}


def main():
    for path in sys.argv[1:]:
        txt = open(path).read()
        runs = re.findall(r"stamp\s+(\d+): w0\s+(\d+)", txt)
        m = re.search(r"B=(\d+) x (\d+) samples: ([\d.]+) k samples/s, kernel ([\d.]+) ms = ([\d.]+) us/step, plan (.*)", txt)
        if not runs or not m:
            continue
        last = {}
        for k, v in runs:
            last[int(k)] = int(v)                      # (the log holds the warm-up call and the timed one: the later wins)
        cyc = [last.get(k + 1, 0) for k in range(15)]
        total = sum(cyc[:11])
        step_us = float(m.group(5))
        L = 16
        print("%s: %s k samples/s, %s us per step (stamps build), plan %s" % (path.split("/")[-1], m.group(3), m.group(5), m.group(6)))
        for k, c in enumerate(cyc):
            us = c / total * step_us
            print("  %-82s %6.2f us/step%s" % (NAMES[k], us, "  = %.2f us per layer" % (us / L) if k in PER_LAYER else ""))
        print()


if __name__ == "__main__":
    main()
