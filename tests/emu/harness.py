"""Loads tests/emu/_build/libqpx_emu.so (the HIP kernel bodies run by host threads) behind the
package's ctypes marshalling layer.  TEST INFRASTRUCTURE ONLY -- see tests/emu/qpx_platform.h."""
import contextlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_SO = os.environ.get("QPX_EMU_SO") or os.path.join(HERE, "_build", "libqpx_emu.so")


def build():
    subprocess.check_call(["make", "-C", HERE, "-s"], stdout=subprocess.DEVNULL)
    return EMU_SO


_EMU = None


def emu_lib():
    global _EMU
    if _EMU is None:
        from qpth_amd import _lib
        build()
        _EMU = _lib.QpxLib(EMU_SO)
    return _EMU


@contextlib.contextmanager
def emulated(threads=128, variant=0):
    """Run qpth_amd on CPU tensors through the emulator inside this block.  `variant` is the
    qpx_set_ipm_variant knob (which kernel family / form runs, see include/qpx.h)."""
    from qpth_amd import _lib
    old = os.environ.get("QPX_EMU_THREADS")
    os.environ["QPX_EMU_THREADS"] = str(threads)
    lib = emu_lib()
    old_variant = lib.dll.qpx_set_ipm_variant(int(variant))
    _lib.set_test_backend(lib)
    try:
        yield
    finally:
        _lib.set_test_backend(None)
        lib.dll.qpx_set_ipm_variant(old_variant)
        if old is None:
            os.environ.pop("QPX_EMU_THREADS", None)
        else:
            os.environ["QPX_EMU_THREADS"] = old
