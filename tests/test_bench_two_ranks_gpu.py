"""bench.py's N > 1 control flow (barriers, agreed step counts, the hand-off inside the timed region, the untimed full hand-off leg,
rank 0's single JSON line) on a single-GPU box: two ranks on cuda:0 over gloo (MTFJSP_BENCH_ONE_DEVICE=1).  The numbers mean nothing
— the ranks share one GPU, and their single-launch GIN kernels cannot be co-resident, so this also drives the time-out → streaming
fallback → restart path under real contention."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_with_two_ranks_and_reports_the_handoff():
    env = dict(os.environ, MTFJSP_BENCH_ONE_DEVICE="1", MTFJSP_NO_RESIDENT_GIN="1")   # (streaming GIN: two resident grids on one GPU only time out)
    # the plain form the driver uses: bench.py starts its two ranks itself (a child torch.distributed.run on 127.0.0.1)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "10", "--batch", "256",
           "--min-seconds", "0.05", "--min-warmup-seconds", "0.1"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    S, B = 5 * 36, 256
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["handoff"]["world"] == 2 and d["handoff"]["allgather_bytes_per_rank"] == 4 * S * B * 4
    hf = d["handoff_full"]
    assert "error" not in hf, hf
    assert hf["world"] == 2 and hf["allgather_bytes_per_rank"] == 16 * S * B * 4      # 16 tensors x [S, B_local] f32
    assert "cpu_baseline" not in d                                                    # N = 1 only


def test_bench_launcher_form_still_works_and_mislaunch_exits_at_once():
    """`python -m torch.distributed.run ... bench.py --gpus 2` (ranks already started) must not start ranks again; a WORLD_SIZE that
    disagrees with --gpus exits before the CPU legs or torch are touched"""
    env = dict(os.environ, MTFJSP_BENCH_ONE_DEVICE="1", MTFJSP_NO_RESIDENT_GIN="1")
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--batch", "128",
           "--min-seconds", "0.05", "--min-warmup-seconds", "0.1", "--no-full-handoff"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    import time
    t0 = time.time()
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=dict(os.environ, RANK="0", WORLD_SIZE="2"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr and time.time() - t0 < 20
