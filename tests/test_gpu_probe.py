"""cos_hbm_probe: the empirical-ceiling diagnostic bench.py reports next to the 8 TB/s spec figure."""
import ctypes as C

import pytest


@pytest.mark.gpu
def test_hbm_probe_runs_and_is_plausible():
    from cosdata_amd import _lib
    L = _lib.lib()
    g = C.c_double(0.0)
    for kind, rb in ((0, 0), (1, 0), (2, 768), (2, 1024)):
        _lib.check(L.cos_hbm_probe(0, kind, 256 << 20, rb, 2, C.byref(g)))
        assert 50.0 < g.value < 20000.0, (kind, g.value)   # GB/s; an MI355X streams a few TB/s, MALL hits can exceed HBM
    # argument checking
    assert L.cos_hbm_probe(0, 2, 256 << 20, 100, 1, C.byref(g)) != 0   # row_bytes not a multiple of 16
    assert L.cos_hbm_probe(0, 7, 256 << 20, 0, 1, C.byref(g)) != 0
