"""`python bench.py --gpus N` started as a plain script (no WORLD_SIZE) must start its own ranks through
torch.distributed.run in a child process and pass rank 0's result line through.  The child is stubbed here: the test
checks the command line, the environment and the relaying, without a GPU."""
import json
import subprocess
import sys
import types

import pytest

import bench


def _stub(monkeypatch, stdout, rc=0):
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=rc, stdout=out_text)

    out_text = stdout
    monkeypatch.setattr(subprocess, "run", fake_run)
    return seen


def test_gpus_n_without_world_size_launches_ranks(monkeypatch, capsys):
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 2})
    seen = _stub(monkeypatch, "some rank chatter\n" + line + "\n")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--workload", "c4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = [c for c in cmd if c.endswith("bench.py")]
    assert script and cmd[cmd.index(script[0]) + 1:] == ["--gpus", "2", "--steps", "3", "--workload", "c4"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == line            # exactly one line on stdout: rank 0's
    assert "some rank chatter" in out.err     # anything else goes to stderr


def test_launcher_passes_the_exit_code_on(monkeypatch, capsys):
    _stub(monkeypatch, "", rc=3)
    assert bench.launch_ranks(4, ["--gpus", "4"]) == 3
    _stub(monkeypatch, "no result here\n", rc=0)
    assert bench.launch_ranks(4, ["--gpus", "4"]) == 1   # ranks that print no line are a failure, not an empty success


def test_rank_processes_do_not_launch_again(monkeypatch):
    """Under torch.distributed.run (WORLD_SIZE set) the script is a rank: --gpus must agree with the world size."""
    def boom(*a, **k):
        raise AssertionError("a rank must not start ranks")
    monkeypatch.setattr(subprocess, "run", boom)
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=4" in str(e.value.code)
