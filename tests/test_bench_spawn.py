"""bench.py's launch contract: `python bench.py --gpus N` must run N ranks -- under the driver's torchrun, or by
launching them itself -- and rank 0 prints exactly ONE line with n_gpus == N; it must never run fewer ranks than asked and say N.
No GPU needed: --dry-spawn stops after the rendezvous (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _json_lines(txt):
    out = []
    for ln in txt.splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            out.append(json.loads(ln))
    return out


def test_self_spawn_two_ranks_one_line():
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-spawn"], capture_output=True, text=True, timeout=240, env=_env())
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = _json_lines(pr.stdout)
    assert len(lines) == 1, pr.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["local_ranks_seen"] == [0, 1]


def test_under_torchrun_two_ranks_one_line():
    """the driver's form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           BENCH, "--gpus", "2", "--dry-spawn"]
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=_env(OMP_NUM_THREADS="1"))
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = _json_lines(pr.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_more_gpus_than_visible_fails_loudly():
    """on a box with fewer GPUs than asked (here: none) the launcher refuses instead of printing n_gpus: 1"""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    ask = have + 1 if have else 2
    pr = subprocess.run([sys.executable, BENCH, "--gpus", str(ask)], capture_output=True, text=True, timeout=240, env=_env())
    assert pr.returncode != 0
    assert f"{ask} GPUs requested, {have} visible" in pr.stderr + pr.stdout
    assert not _json_lines(pr.stdout)


def test_gpus_flag_must_match_world_size():
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-spawn"], capture_output=True, text=True, timeout=120,
                        env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert pr.returncode != 0 and "--gpus 4" in pr.stderr + pr.stdout
    assert not _json_lines(pr.stdout)


def test_eight_ranks_from_a_cold_start():
    """`bench.py --gpus 8` -- what the driver runs on an 8-GPU node -- starts its eight ranks without manual steps: every rank opens an engine
    (store-only here: no GPU), loads the graph through the ABI, takes ITS rotation of the request stream, configures the 8-way type-hash sharding of the extra
    leg, and the ranks agree on the max-over-ranks clock and gather their records over gloo; rank 0 prints ONE line with n_gpus == 8."""
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-spawn"], capture_output=True, text=True, timeout=600, env=_env(OMP_NUM_THREADS="1"))
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = _json_lines(pr.stdout)
    assert len(lines) == 1, pr.stdout
    ln = lines[0]
    assert ln["n_gpus"] == 8 and ln["local_ranks_seen"] == list(range(8)) and ln["scaling"] == "weak"
    assert sorted(r_["rank"] for r_ in ln["per_rank"]) == list(range(8)) and abs(ln["max_elapsed_s"] - 1.07) < 1e-9  # the slowest rank's clock
    assert ln["distinct_request_streams"] >= 7  # every rank answers its own rotation of the stream
    assert ln["sharded_leg"] == "on" and all(0 <= v < 8 for v in ln["shard_of_type"].values())
