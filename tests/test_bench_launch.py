"""bench.py's launch contract on a box without a GPU: `python bench.py --gpus N` with WORLD_SIZE unset starts the N
ranks itself (child process under torch.distributed.run) and relays rank 0's line and the return code; inside a
torch.distributed.run job it runs as a rank.  ZS_BENCH_RENDEZVOUS_ONLY stops each rank after the rendezvous."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ZS_BENCH_RENDEZVOUS_ONLY="1", OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr
    return json.loads(lines[0])


def test_plain_start_launches_the_ranks_itself():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "0"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert _line(out) == {"rendezvous_only": True, "n_gpus": 2, "steps": 3}


def test_single_gpu_start_does_not_spawn():
    out = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert _line(out)["n_gpus"] == 1


def test_under_torchrun_it_is_a_rank_and_failures_propagate():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29731", "bench.py", "--gpus", "2", "--steps", "1"]
    out = _run(cmd)
    assert out.returncode == 0, out.stderr[-2000:]
    assert _line(out)["n_gpus"] == 2
    # a world size that contradicts --gpus is an error, and the self-launcher relays a worker's failure
    bad = _run([sys.executable, "bench.py", "--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert bad.returncode != 0
