"""Register / scratch budget of the hot kernels, read from hipcc's resource report (no GPU needed).

A value that spills to scratch memory is re-read inside the per-sample or per-chunk loop with an s_waitcnt in front of it: the
pipelined decode kernel lost 10 % to seven such reloads before its biases moved to LDS, the GEMM loader
80 bytes per lane per chunk before its staging registers became scalars (section 5b).  This pins the state they are in now."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qpnet_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernel-name fragment -> (file, max scratch bytes per lane, min waves per SIMD)
BUDGET = {
    "k_decode_pipe12DecodeParams": ("decode_pipe.hip", 0, 2),    # one utterance per five-role group: the latency-optimal kernel, nothing in scratch
    "k_decode_pipe_nILi2E": ("decode_pipe.hip", 32, 2),           # two per group: three weight pairs of the post-2 role spill (whole-kernel allocation:
                                                                  # the K role needs all 256 registers); non-inlined roles cost 2.6 KB of scratch instead
    "k_decode_coop": ("decode_coop.hip", 0, 3),             # 768 threads: three waves per SIMD, nothing in scratch memory
    "k_layer_fwd_pILi11ELb0E": ("train_fwd.hip", 0, 2),     # persistent layer forward: two workgroups per CU, weights in registers
    "k_layer_fwd_pILi8ELb0E": ("train_fwd.hip", 0, 2),      # ... the K = 128 form (aux 1x1 at frame rate)
    "k_layer_bwd_pILi8ELb0E": ("train_bwd.hip", 0, 2),
    "k_stack_fwdILi8EE": ("train_stack.hip", 0, 2),         # the one-launch residual stack (work queue), K = 128: two workgroups per CU, nothing in scratch
    "k_stack_bwdILi8EE": ("train_stack.hip", 0, 2),
    "k_stack_fwdILi11EE": ("train_stack.hip", 0, 2),        # ... and the K = 176 form (upsampling_factor 0 / < 16, QPN_AUX_HOIST=0)
    "k_stack_bwdILi11EE": ("train_stack.hip", 0, 2),
    "k_post_fwd_wILi5E": ("train_fwd.hip", 0, 2),           # 80-row post-net tiles: one 512-thread workgroup per CU
    "k_post_bwd_wILi5E": ("train_bwd.hip", 0, 2),
    "k_post_fb_wILi5E": ("train_bwd.hip", 0, 2),            # forward + backward of a post-net row tile in one kernel (qpn_train_step)
    "k_layer_fwdILi1E": ("train_fwd.hip", 0, 5),          # five 16-row workgroups per CU must be co-resident
    "k_post_fwdILi1E": ("train_fwd.hip", 0, 5),
    "k_layer_bwdILi1E": ("train_bwd.hip", 0, 5),
    "k_wgrad3ILi1ELi4ELi4ELb0ELi2EE": ("train_bwd.hip", 0, 2),      # post-net weight gradients (both in one launch: 512 workgroups, two per CU)
    "k_wgrad3ILi0ELi4ELi4ELb0ELi2EE": ("train_bwd.hip", 0, 2),      # skip 1x1 (B = the gate product)
    "k_wgrad3ILi3ELi2ELi11ELb0ELi1EE": ("train_bwd.hip", 0, 2),     # dW1: 159 VGPRs + 88 accumulators, two workgroups per CU
    "k_wgrad3ILi3ELi2ELi8ELb0ELi1EE": ("train_bwd.hip", 0, 2),      # dW1 at K = 128 (aux 1x1 at frame rate)
    "k_aux_tail": ("train_bwd.hip", 0, 2),                             # 64 accumulator rows in registers at once, nothing in scratch
    "k_wgrad3ILi0ELi1ELi4ELb0ELi1EE": ("train_bwd.hip", 0, 4),      # residual 1x1: one A array (dXout summed in place), one B array (the gate product); memory-bound: occupancy is what it lives on
    "k_wgrad3ILi0ELi1ELi4ELb1ELi1EE": ("train_bwd.hip", 0, 4),      # ... with the two-part dXout of the per-layer backward launches
    "k_up_bwd": ("train_bwd.hip", 0, 2),                               # a row's 16 float4 words in registers at once, still nothing in scratch
    "k_gemm_nnILi0ELi1E": ("train_gemm.hip", 0, 3),
    "k_gemm_nnILi1ELi4E": ("train_gemm.hip", 0, 3),
    "k_gemm_tnILi3E": ("train_gemm.hip", 0, 3),
}


def _report(fname):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-c",
                          os.path.join(CSRC, fname), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); kernels[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+?):\s+(\d+)", line)
        if m and cur:
            kernels[cur][m.group(1).strip()] = int(m.group(2))
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hot_kernels_stay_within_their_register_budget():
    files = sorted({v[0] for v in BUDGET.values()})
    with ThreadPoolExecutor(max_workers=4) as ex:
        reports = dict(zip(files, ex.map(_report, files)))
    for frag, (fname, max_scratch, min_occ) in BUDGET.items():
        hits = {k: v for k, v in reports[fname].items() if frag in k}
        assert hits, "no kernel matching %s in %s: %s" % (frag, fname, sorted(reports[fname]))
        for name, r in hits.items():
            scratch = r.get("ScratchSize [bytes/lane]", -1)
            occ = r.get("Occupancy [waves/SIMD]", -1)
            assert 0 <= scratch <= max_scratch, "%s spills %d bytes per lane to scratch memory" % (name, scratch)
            assert occ >= min_occ, "%s: %d waves per SIMD, needs %d" % (name, occ, min_occ)
