#!/usr/bin/env python3
"""Dry run of `-m gpu` tests in the GPU-less container: the bodies of the named tests of tests/test_gpu_parity.py are
executed on CPU tensors with the host-thread emulator as the backend (tests/emu), so that a change of the Python
surface or of the argument checks is seen before it costs a GPU visit.  It proves nothing about the HIP build.
    gpu_tests_on_emu.py test_golden_batches test_every_loop_kernel_form ...      (full-size tests take minutes)"""
import sys, inspect, traceback, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pytest
import test_gpu_parity as T
from emu.harness import emu_lib
from qpth_amd import _lib
torch.cuda.synchronize = lambda *a, **k: None
lib = emu_lib()
_lib.set_test_backend(lib)
_lib.hip = lambda: lib            # variant knobs go to the emulator
dev = torch.device("cpu")
os.environ["QPX_EMU_THREADS"] = "256"
class Capsys:                                    # stand-in for pytest's fixture: what was printed since the last call
    def __init__(self):
        import io
        self.buf = io.StringIO()
        self._real = sys.stdout
        sys.stdout = self.buf

    def readouterr(self):
        class R:
            pass
        r = R()
        r.out, r.err = self.buf.getvalue(), ""
        self.buf.seek(0)
        self.buf.truncate()
        return r

    def close(self):
        sys.stdout = self._real
names = sys.argv[1:]
for name in names:
    fn = getattr(T, name)
    params = [{}]
    for mk in getattr(fn, "pytestmark", []):
        if mk.name == "parametrize":                       # stacked marks: the cartesian product
            argn = [a.strip() for a in mk.args[0].split(",")]
            one = [dict(zip(argn, v if isinstance(v, (tuple, list)) and len(argn) > 1 else (v,))) for v in mk.args[1]]
            params = [dict(a, **b) for a in params for b in one]
    for prm in params:
        kw = dict(prm) if prm else {}
        if "dev" in inspect.signature(fn).parameters:
            kw["dev"] = dev
        cap = None
        if "capsys" in inspect.signature(fn).parameters:
            cap = kw["capsys"] = Capsys()
        try:
            fn(**kw)
            verdict = ("ok  ", name, prm if prm else "")
        except Exception as ex:
            verdict = ("FAIL", name, prm, type(ex).__name__, str(ex)[:300])
        if cap:
            cap.close()
        print(*verdict)
