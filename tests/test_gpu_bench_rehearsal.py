"""bench.py's N-rank frame loop on one GPU: a one-rank RCCL process group (backend "nccl") through
which every frame runs the collective calls of an N-rank run, the gather of the packed tile
buffers or the reduce of the int64 sums (main.scala:87-91 is the plain sum being shared), the
barriers and the all-reduce of the timing.  The picture that came through them must equal the
one resolved straight from the sums."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shard", ["tiles", "samples"])
def test_one_rank_group_runs_the_collectives(ht, shard):
    assert ht.device_count() >= 1, "GPU tests need a HIP device (libhiptrace has no CPU path)"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "scene", "--spp", "8", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--rehearse-collectives", "--shard", shard],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["config"]["shard"] == shard
    assert d["rehearsal"] and d["rehearsal_picture_equals_direct_resolve"] is True
    assert d["value"] > 0
