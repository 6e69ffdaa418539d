import os, sys, time
os.environ["QPN_PIPE_STAMPS"]="1"
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
import util
cuda = torch.device("cuda:0"); cfg=PAPER
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda)
bx, bh, bd, ns = synth.decode_batch(cfg, [(100, 300, 1.0)])
xb, hb = torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda)
m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
print("samples", ns[0], "kernel ms", m.last_decode_kernel_ms, "us/sample", m.last_decode_kernel_ms*1e3/ns[0])
