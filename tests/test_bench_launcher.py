"""`python bench.py --gpus N` (N > 1) with no launcher around it starts torchrun as a CHILD process before torch is
imported, relays rank 0's one JSON line and exits with the child's code.  No GPU needed."""
import importlib.util
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_decision_and_command(bench):
    a2 = bench.parse_args(["--gpus", "2", "--steps", "3"])
    a1 = bench.parse_args([])
    assert a1.gpus == 1 and not bench.needs_self_launch(a1, {})
    assert bench.needs_self_launch(a2, {}) and not bench.needs_self_launch(a2, {"WORLD_SIZE": "2"})
    cmd = bench.self_launch_command(["--gpus", "2", "--steps", "3"], 2, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "2", "--steps", "3"]


def test_main_takes_the_launcher_path_and_relays(bench, monkeypatch, capsys):
    seen = {}
    line = json.dumps({"metric": "sampled edges/sec", "value": 1.0, "n_gpus": 2})

    def fake_run(cmd, env=None, stdout=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout=("NCCL version banner\n" + line + "\n").encode())

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setitem(sys.modules, "torch", None)       # importing torch on this path would raise: it must not happen
    with pytest.raises(SystemExit) as ex:
        bench.main(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert ex.value.code == 0
    assert capsys.readouterr().out.strip() == line
    assert "torch.distributed.run" in seen["cmd"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["cmd"][-6:] == ["--gpus", "2", "--steps", "2", "--warmup", "1"]


def test_child_failure_becomes_the_exit_code(bench, monkeypatch, capsys):
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=3, stdout=b""))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main(["--gpus", "4"])
    assert ex.value.code == 3 and capsys.readouterr().out == ""


def test_plain_spelling_really_starts_ranks():
    """`python bench.py --gpus 2` end to end on this CPU-only container: torchrun starts two ranks, each dies for lack of
    a GPU (the HIP library / device is missing) -- the launcher itself must have worked: non-zero exit, no result line,
    and the ranks' error mentions the device or library, not the old 'launch N>1 through torch.distributed.run' refusal."""
    env = dict(os.environ, TG_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--scale", "10", "--batches-per-step", "4", "--no-secondary", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "launch N>1 through" not in p.stderr
    assert "WORLD_SIZE" not in p.stderr or "but WORLD_SIZE" not in p.stderr
