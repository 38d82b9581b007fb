"""bench.py --gpus N must start N ranks itself: the launch / rendezvous / barrier / final-gather path of a
multi-rank run, rehearsed on CPU with the gloo backend (no GPU work, the printed line carries no measurement), and the
refusal to run a smaller job under the label of a bigger one."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_spawns_its_ranks_and_gathers_over_the_process_group():
    r = _run(["--gpus", "2", "--rehearse-distributed"], {"BENCH_DIST_BACKEND": "gloo", "BENCH_SHARE_DEVICE": "1"})
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["rehearsal"] is True and out["value"] is None and out["metric"] is None  # cannot pass for a result
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1] and out["distinct_processes"] == 2


def test_bench_refuses_more_gpus_than_there_are():
    import torch

    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 8), "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 2, (r.returncode, r.stdout, r.stderr)
    assert "refusing" in r.stderr and "n_gpus" not in r.stdout


def test_gpus_flag_must_match_the_world_size_of_an_external_launcher():
    r = _run(["--gpus", "4", "--rehearse-distributed"], {"BENCH_DIST_BACKEND": "gloo", "WORLD_SIZE": "1", "RANK": "0"})
    # (_run drops inherited rank variables, then WORLD_SIZE=1 is put back: a launcher that started one rank for --gpus 4)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
