#!/usr/bin/env python3
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "qpth_amd", "libqpx_bench.so"))
lib.qpx_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
out = torch.zeros(20480 + 4096, dtype=torch.float64, device=dev)
inp = torch.rand(4096, dtype=torch.float64, device=dev) + 0.5
lib.qpx_bench(9, 1, 20000, 0, out.data_ptr(), inp.data_ptr(), None)
torch.cuda.synchronize()
o = out.cpu().numpy()
print("rcp_ (estimate + 2 Newton) max rel err vs 1/x: %.3e ; raw v_rcp_f64: %.3e" % (o[4096:4160].max(), o[8192:8256].max()))
