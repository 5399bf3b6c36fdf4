"""`bench.py --gpus N` must start N ranks itself when no launcher did (the flag used to be parsed and ignored, so
`python bench.py --gpus 8` silently ran one rank).  Checked on CPU with the launcher self-test mode: two real processes, gloo
rendezvous on 127.0.0.1, the packed all-gather of cosdata_amd/sharding.py, MAX-over-ranks timing, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env=None, timeout=300):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def test_gpus_2_without_world_size_launches_two_gloo_ranks():
    r = _run(["--gpus", "2", "--launcher-selftest", "--steps", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                     # exactly one JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and j["asked_gpus"] == 2
    assert j["distinct_processes"] == 2 and len(set(j["rank_pids"])) == 2   # two real processes took part
    assert j["shards_in_merged_answer"] == [0, 1]                          # the merged answer draws on both shards
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr


def test_gpus_flag_must_agree_with_the_launchers_world_size():
    r = _run(["--gpus", "4", "--launcher-selftest"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in r.stderr


def test_more_gpus_than_visible_is_refused_loudly():
    # no GPU in the CPU container (or fewer than 64 anywhere): the launcher must refuse instead of running fewer ranks
    r = _run(["--gpus", "64"])
    assert r.returncode != 0 and "refusing to run fewer ranks" in r.stderr


def test_gpus_8_launches_eight_gloo_ranks_and_every_shard_reaches_the_merge():
    """the shape of BASELINE configs[3] — 8 ranks, ONE packed all-gather per step, a merge that draws on all eight shards — through the
    launcher the driver uses, on CPU (gloo): eight real processes, one JSON line, max-over-ranks timing"""
    r = _run(["--gpus", "8", "--launcher-selftest", "--steps", "3"], env={"OMP_NUM_THREADS": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["ranks"] == 8 and j["asked_gpus"] == 8
    # the workload `value` is quoted on at this N: eight 12.5M x 1024 shards = BASELINE configs[3] itself, exchange inside the step
    w = j["config"]["value_workload_at_this_n"]
    assert "configs[3]" in w and "8 id-range shard" in w and "12500000 x 1024" in w and "all-gather" in w
    assert j["config"]["n_gpus"] == 8 and j["config"]["dim"] == 1024
    assert j["distinct_processes"] == 8
    assert j["shards_in_merged_answer"] == list(range(8))
    assert "--nproc-per-node=8" in r.stderr
