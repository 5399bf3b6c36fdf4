"""COS_HOST_STAGE_THREADS (engine.hip search_host_pipelined + host_stage.h): a big host-buffer call staged through the pipe's pinned
buffer by helper threads returns what the same call returns without staging.  The variable is read once per process, so the staged
run is a child process.  A candidate of the end of round 4 (written without a device at hand): runs when COS_CANDIDATES=1."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
candidates = pytest.mark.skipif(os.environ.get("COS_CANDIDATES", "") != "1", reason="candidate paths: set COS_CANDIDATES=1")

CHILD = r"""
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
import cosdata_amd as ca
rng = np.random.default_rng(5)
n, d = 30000, 96
c = rng.standard_normal((40, d)).astype(np.float32)
X = (c[rng.integers(0, 40, n)] + 0.35 * rng.standard_normal((n, d))).astype(np.float32)
ix = ca.HNSWIndex(d, ca.HNSWHyperParams(ef_search=48), ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), (-4.0, 4.0))
ix.upload_vectors(X)
ix.build(2048)
out = {}
for B in (256, 8192, 20000, 33000):          # below the pipeline threshold, exactly at it, ragged chunk sizes
    Q = (c[rng.integers(0, 40, B)] + 0.4 * rng.standard_normal((B, d))).astype(np.float32)
    Q[min(B - 1, 9000 % B)] = 0.0            # one zero query: its status, nobody else's
    ids, sc, cnt, rc, st = ix.batch_search(Q, 10, return_status=True)
    out[str(B)] = [hashlib.sha256(a.tobytes()).hexdigest() for a in (ids, sc.view(np.uint32), cnt, st)] + [int(rc)]
print(json.dumps(out))
"""


def _run(threads):
    env = dict(os.environ)
    env.pop("COS_HOST_STAGE_THREADS", None)
    if threads:
        env["COS_HOST_STAGE_THREADS"] = str(threads)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@candidates
def test_staged_host_call_returns_the_unstaged_answers():
    plain = _run(0)
    for threads in (1, 3):
        assert _run(threads) == plain, threads
