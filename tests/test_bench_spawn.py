"""bench.py's own launcher (no GPU here): `python bench.py --gpus N` without WORLD_SIZE must start
N rank processes itself -- before importing torch or touching a GPU -- and exit non-zero, promptly,
when a rank fails (here every rank fails: there is no GPU)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_multi_gpu_invocation_spawns_ranks_and_reports_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the failure leg needs a box without a GPU")
    env = dict(os.environ, BSC_BENCH_TIMEOUT_S="240")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                           "--warmup", "1", "--rows", "1000"], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)
    err = proc.stderr.decode()
    assert proc.returncode != 0
    assert "bench.py: rank" in err and "exited with status" in err, err[-2000:]
    assert "launch with torch.distributed.run" not in err
    assert time.time() - t0 < 200


def test_parent_does_not_import_torch_before_spawning():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src.split("def run_rank", 1)[0]
    body = head.split("def spawn_ranks", 1)[1].split("\ndef ", 1)[0]
    assert "import torch" not in body and "cuda" not in body
    main = src.split("def main():", 1)[1]
    assert main.index("spawn_ranks") < main.index("run_rank")


def test_rank_rows_partition_is_contiguous_and_complete():
    sys.path.insert(0, ROOT)
    import bench
    for total in (1_000_000, 999_983, 50_000):
        for world in (1, 2, 4, 8):
            cuts = [bench.rank_rows(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
    assert bench.rank_rows(1_000_000, 3, 8) == (375_000, 500_000)      # r*N/p


def test_chunked_survey_inputs_equal_the_one_shot_stream():
    """bench.py draws SURVEY 8(d)'s RandomState matrices in row chunks; a rank of a strong-scaling
    run keeps rows [r0, r1) of the same stream."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from oracle import svi
    X, y, _ = svi.make_cfg2(1001, 7)
    got = bench._normal_f32(np, 1234, (1001, 7), chunk_rows=97)
    np.testing.assert_array_equal(got, X)
    np.testing.assert_array_equal(bench._normal_f32(np, 1234, (700, 7), chunk_rows=64), X[:700])


def test_lowered_tree_baseline_equals_the_oracle_pass():
    import numpy as np
    from oracle import lowered_baseline as lb
    from oracle import svi
    X, y, _ = svi.make_cfg2(5000, 32)
    W = (np.random.RandomState(0).standard_normal((4, 32)) / 16).astype(np.float32)
    Q, G = lb.blr_data_pass_lowered(X, y, W)
    Q0, G0 = svi.blr_data_pass(X, y, W)
    np.testing.assert_allclose(Q, Q0, rtol=1e-5)                       # float32 BLAS vs float64
    np.testing.assert_allclose(G, G0, rtol=1e-4, atol=1e-4 * np.abs(G0).max())
    assert "_tensordot(W, _dimshuffle(X, 1, 0), [1], [0])" in lb.blr_pass_functions()["lowered"]
    with lb.threads(1):
        Q1, _ = lb.blr_data_pass_lowered(X, y, W)
    np.testing.assert_allclose(Q1, Q0, rtol=1e-5)
    assert lb.host_facts()["cpu_count"] == os.cpu_count()


def test_sweep_bookkeeping_of_the_config2_driver():
    """BLRReparamSVI alternates the sweep order per PASS; a freshly swapped-in batch is streamed once.
    Host logic only -- no device.  (The number of passes per update is the library's answer,
    bsc_blr_pass_count: tests/test_blr_gpu.py::test_pass_count_is_the_librarys_answer.)"""
    from types import SimpleNamespace
    from bayesic_amd.svi.blr import BLRReparamSVI, SWEEP_STREAM, SWEEP_FORWARD_KEEP, SWEEP_BACKWARD_KEEP

    def fake(S, D=256, sweep="alternate", reproducible=False):
        passes = 1 if S <= 8 else 2
        return SimpleNamespace(S=S, D=D, sweep=sweep, reproducible=reproducible, _fresh_batch=False,
                               _sweep_next=SWEEP_FORWARD_KEEP, _passes_per_update=lambda: passes)

    me = [None]
    # without the library (a test double as context) the driver assumes eight draws per pass
    plain = SimpleNamespace(S=20, D=128, _yarg=0, ctx=SimpleNamespace())
    assert BLRReparamSVI._passes_per_update(plain) == 3
    # one pass per update: forward, backward, forward, ...
    me[0] = fake(8)
    assert [BLRReparamSVI._take_sweep(me[0]) for _ in range(4)] == [1, 2, 1, 2]
    # two passes per update (S = 24): every update starts where the previous one ended
    me[0] = fake(24)
    assert [BLRReparamSVI._take_sweep(me[0]) for _ in range(3)] == [1, 1, 1]
    # a fresh batch is streamed, then walked back from its end
    me[0] = fake(8)
    me[0]._fresh_batch = True
    assert [BLRReparamSVI._take_sweep(me[0]) for _ in range(3)] == [SWEEP_STREAM, SWEEP_BACKWARD_KEEP,
                                                                    SWEEP_FORWARD_KEEP]
    me[0] = fake(8, sweep="stream")
    assert BLRReparamSVI._take_sweep(me[0]) == SWEEP_STREAM
    me[0] = fake(8, reproducible=True)
    assert BLRReparamSVI._take_sweep(me[0]) == SWEEP_STREAM
