"""Python mirror of the offsets in qpth_amd/csrc/qpx_layout.h that tooling needs (profiling only)."""


def _align4(x):
    return (x + 3) & ~3


def _tri(i):
    return i * (i + 1) // 2


def prof_offset(n, m, q, images):
    """offset (elements) of the 8 phase timers the profiling build (-DQPX_PROFILE) dumps into a blob;
    images = 0: workgroup-kernel family, else thread-grid / tile family (fac_layout in qpx_layout.h)"""
    if images == 0:
        sizes = (_tri(n), n, n * m, _tri(m), n * q, q * m, _tri(q), q, m)
    else:
        sizes = (n * n, n * m, q * n, m * q, q * q)
    return sum(_align4(s) for s in sizes) + 4
