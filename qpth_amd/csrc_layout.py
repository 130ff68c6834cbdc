"""Python mirror of the offsets in qpth_amd/csrc/qpx_layout.h that tooling needs (profiling only)."""


def _align4(x):
    return (x + 3) & ~3


def _tri(i):
    return i * (i + 1) // 2


def fac_layout_T_offset(n, m, q):
    o = 0
    for sz in (_tri(n), n, n * m, _tri(m), n * q, q * m, _tri(q), q, m):
        o += _align4(sz)
    return o + 4
