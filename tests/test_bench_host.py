"""CPU: host logic of bench.py -- the N>1 self-launch (a CHILD torchrun, never exec), the --gpus /
WORLD_SIZE check, the host-core count."""
import os
import subprocess
import sys
import types

import pytest

from conftest import ROOT


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def test_self_launch_starts_child_torchrun_and_relays_its_code(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(types.SimpleNamespace(gpus=4))
    assert e.value.code == 7                                       # non-zero child -> non-zero parent
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[-7]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_must_match_world_size(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match WORLD_SIZE" in str(e.value)


def test_host_cores_is_positive_and_uncapped():
    bench = _bench()
    n = bench.host_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
