"""bench.py's own launcher: `python bench.py --gpus N` without torch.distributed.run around it starts
the per-GPU processes itself -- as a CHILD process, relaying the one JSON line and the exit status.  Here (no GPU) the ranks
can only fail with bench.py's "needs a GPU" message: what is checked is that they were started, two of them, and that their
failure is ours.  The GPU box runs the real thing (tests/test_gpu_full.py::test_bench_self_launch_two_ranks_on_one_gpu)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_world_size_starts_torchrun_as_a_child():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the launch is exercised for real by the gpu-marked test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0
    assert "needs torch.distributed.run" not in p.stderr            # round 3's refusal is gone
    assert p.stderr.count("bench.py needs a GPU") >= 2, p.stderr[-2000:]   # both ranks ran bench.py's main
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
