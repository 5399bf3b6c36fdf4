"""Host-side logic of bench.py that needs no GPU: launch shaping (exactly K timed steps) and the usable-core count."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_shape_times_exactly_k_steps_when_it_can():
    import bench
    assert bench.launch_shape(1024, 64, 32) == (32, 32, 2)          # default: 32 launches of 32 client batches
    for steps in (32, 64, 100, 50, 1000, 20, 96, 8):
        C, n_launch, _ = bench.launch_shape(steps, 64, 32)
        assert C * n_launch == steps and C * 4 >= 32, (steps, C, n_launch)
    for steps in (53, 5, 1, 31 * 7 + 0):                            # no usable divisor: rounded up to whole launches
        C, n_launch, _ = bench.launch_shape(steps, 64, 32)
        assert C * n_launch >= steps and (C * n_launch - steps) < C
    C, n_launch, n_warm = bench.launch_shape(1024, 0, 32)
    assert n_warm == 0
    assert bench.launch_shape(10, 3, 1) == (1, 10, 3)               # coalescing off


def test_effective_cores_is_bounded_by_affinity():
    import bench
    n = bench.effective_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
