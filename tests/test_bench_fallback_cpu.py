"""bench.py's guard around the first real multi-GPU run (VERDICT r2 #10b): the RCCL communicator behind the C ABI has
only ever been created at world size 1, so every rank reports over the host channel whether its bsc_comm_init_rank
came up; if ANY failed, ALL drop theirs and build the torch.distributed fallback group together.  Driven here under
gloo with two CPU ranks and a simulated failure -- on one rank, on both, on none."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, failing):
    port = 29500 + ((os.getpid() * 7 + len(failing) * 131) % 2000)
    out = str(tmp_path / ("rank%d_" + "_".join(map(str, failing)) + ".json"))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_fallback_worker.py"), out,
                                       ",".join(map(str, failing))], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return [json.load(open(out % r)) for r in range(2)]


@pytest.mark.parametrize("failing", [(1,), (0, 1), ()])
def test_ranks_agree_on_the_exchange(tmp_path, failing):
    r0, r1 = _run(tmp_path, failing)
    for r in (r0, r1):
        assert r["sum"] == 3.0                          # the collective over whatever was agreed completes on both ranks
    if failing:
        # every rank -- also one whose own communicator came up -- drops it and joins the fallback group
        assert r0["log"] == r1["log"] == ["init", "drop", "fallback"]
        assert r0["group"] and r1["group"]
        assert "failed on %d rank(s)" % len(failing) in r0["note"]
        assert ("this rank" in r0["note"]) == (0 in failing) and ("this rank" in r1["note"]) == (1 in failing)
    else:
        assert r0["log"] == r1["log"] == ["init"] and not r0["group"] and r0["note"] is None
