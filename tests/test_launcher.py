"""CPU suite: `python bench.py --gpus N` outside a launcher starts its own ranks (bench.launch_ranks) and never reports a line for
fewer GPUs than it was asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_refuses_when_the_node_has_fewer_devices(capsys, monkeypatch):
    import bench
    monkeypatch.delenv("YK_BENCH_BACKEND", raising=False)
    rc = bench.launch_ranks(8, argv=["--gpus", "8"], count=lambda: 1)
    out = capsys.readouterr()
    assert rc != 0 and out.out == "" and "needs 8 HIP devices" in out.err


def test_rehearsal_is_limited_to_six_ranks_per_card(capsys, monkeypatch):
    import bench
    monkeypatch.setenv("YK_BENCH_BACKEND", "gloo")
    assert bench.launch_ranks(8, argv=["--gpus", "8"], count=lambda: 1) != 0
    assert "at most 6 ranks" in capsys.readouterr().err


def test_spawns_ranks_and_relays_only_a_result_line():
    """Two real rank processes (torch.distributed.run, gloo rehearsal switch).  This box has no GPU, so every rank stops with bench.py's
    "needs a HIP device" status: the launcher must pass that failure on and print no JSON line (in particular no n_gpus: 1 line)."""
    env = dict(os.environ, YK_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    code = ("import bench, sys; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu'];"
            "sys.exit(bench.launch_ranks(2, count=lambda: 1))")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():                           # on a GPU box the rehearsal really runs: one line, for two ranks
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["n_gpus"] == 2
    else:
        assert out.returncode != 0
        assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout
        assert "HIP device" in out.stderr
