#!/usr/bin/env python3
"""Decodes the lane layout of v_mfma_f64_4x4x4_4b_f64 on the GPU at hand (libqpx_bench.so, kernel 26): which lanes'
A and B operands feed the accumulator of each lane."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "qpth_amd", "libqpx_bench.so"))
lib.qpx_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
out = torch.zeros(32768, dtype=torch.float64, device=dev)
inp = torch.ones(4096, dtype=torch.float64, device=dev)
assert lib.qpx_bench(26, 1, 1, 0, out.data_ptr(), inp.data_ptr(), None) == 0
torch.cuda.synchronize()
o = out[:19 * 64].cpu().numpy().reshape(19, 64)
bits = lambda v: [k for k in range(16) if (int(v) >> k) & 1]
print("D lane : A lanes (mod 16) summed over k | B lanes (mod 16) summed over k | sees block-0 A")
for lane in range(64):
    print("%2d : %-16s | %-16s | %d" % (lane, bits(o[16, lane]), bits(o[17, lane]), int(o[18, lane])))
print("B lane x (mod 16) -> for each D lane (block 0) the A lane that multiplies it")
for x in range(16):
    print("%2d : %s" % (x, [(lane, bits(o[x, lane])) for lane in range(16) if o[x, lane] != 0]))
