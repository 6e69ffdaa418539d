"""The backward work-queue kernel publishes the previous tile in the MIDDLE of its contraction behind a COUNTED wait (csrc/train_stack.hip,
k_stack_bwd: `s_waitcnt vmcnt(9)`, 10 in wave 0; 15 / 16 in the aux-hoist form): everything older than the tile's request group -- i.e. the previous
tile's outputs -- must have completed, the group itself stays in flight.  That count is only right while hipcc issues the group as at least that many
instructions (eleven requests, twelve in wave 0; seventeen / eighteen in the aux-hoist form).  This test compiles the file to gfx950 assembly and counts them."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_request_group_of_the_backward_queue_is_at_least_as_long_as_the_counted_wait(tmp_path):
    out = tmp_path / "stack.s"
    src = os.path.join(ROOT, "qpnet_amd", "csrc", "train_stack.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "-S", "--cuda-device-only",
                    "-I" + os.path.dirname(src), src, "-o", str(out)], check=True, capture_output=True, timeout=600)
    text = out.read_text().split("\n")
    # <11>: the K = 176 form (aux features at sample rate), <8>: K = 128 (aux hoist: six more requests in the group, a 64-MFMA contraction)
    for inst, waits, nloads, mfma_lo, mfma_hi in (("ILi11E", (9, 10), 11, 40, 56), ("ILi8E", (15, 16), 17, 28, 36)):
        start = next(i for i, l in enumerate(text) if re.match(r"^_Z11k_stack_bwd%s\w*:" % inst, l))
        end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
        body = text[start:end]
        counted = [i for i, l in enumerate(body) if re.search(r"s_waitcnt vmcnt\((%d|%d)\)\s*$" % waits, l)]
        first = next(i for i in counted if re.search(r"vmcnt\(%d\)" % waits[0], body[i]))
        assert any(re.search(r"vmcnt\(%d\)" % waits[1], body[i]) for i in counted), "the two counted waits (waves 1-3 / wave 0) of the publish point"
        barrier = max(i for i in range(first) if "s_barrier" in body[i])                       # B2
        vmem = [l for l in body[barrier:first] if re.search(r"^\s*(global|buffer)_(load|store|atomic)", l)]
        loads = [l for l in vmem if "_load_" in l]
        atomics = [l for l in vmem if "_atomic_" in l]
        assert not [l for l in vmem if "_store_" in l], "nothing may be stored between B2 and the publish point"
        assert len(loads) >= waits[0] and len(loads) == nloads, "requests of the group: %d" % len(loads)      # waves 1-3 wait down to two below the group's length
        assert len(loads) + len(atomics) >= waits[1] and len(atomics) == 1                      # wave 0 (its ticket)
        mfma_before = sum("v_mfma" in l for l in body[barrier:first])
        assert mfma_lo <= mfma_before <= mfma_hi, "the publish point sits halfway through the contraction (%d MFMAs in front of it)" % mfma_before
