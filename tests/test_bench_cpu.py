"""bench.py's rank launcher, on a box without a GPU: ``--gpus N`` starts N rank processes itself
(torch.distributed.run, one per GPU) and relays their exit status; a launcher whose world size
disagrees with ``--gpus`` is refused."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE",
                                                           "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_world_size_must_agree_with_gpus():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "WORLD_SIZE=4" in r.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the launcher's behaviour on a box WITHOUT a GPU")
def test_gpus_2_starts_two_ranks_and_relays_their_failure():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0                                           # the ranks' status, not 0
    # both ranks got as far as the GPU check (each prints the message once)
    assert r.stderr.count("bench.py needs an AMD GPU") >= 2, r.stderr[-2000:]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour on a box WITHOUT a GPU")
def test_single_rank_runs_in_process_and_fails_loudly_without_a_gpu():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "needs an AMD GPU" in r.stderr
