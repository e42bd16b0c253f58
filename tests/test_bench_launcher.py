"""bench.py --gpus N started as a plain command becomes the launcher of its N ranks (cf/main.py:47-70 is started by mpirun;
the driver may start `python3 bench.py --gpus 8` by itself).  CPU-only: the ranks are replaced by a stub."""
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_starts_torchrun_and_relays_one_json_line(capsys):
    seen = {}

    def runner(cmd, env, stdout, text):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='NCCL version banner\n{"metric": "m", "value": 1.0, "n_gpus": 2}\n')

    rc = bench.self_launch(2, ["--gpus", "2", "--steps", "3"], runner=runner)
    out = capsys.readouterr()
    assert rc == 0
    assert json.loads(out.out.strip()) == {"metric": "m", "value": 1.0, "n_gpus": 2}      # ONE line on stdout
    assert "NCCL version banner" in out.err
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_self_launch_propagates_failure(capsys):
    rc = bench.self_launch(2, [], runner=lambda *a, **k: types.SimpleNamespace(returncode=3, stdout="boom\n"))
    assert rc == 3 and capsys.readouterr().out == ""
    rc = bench.self_launch(2, [], runner=lambda *a, **k: types.SimpleNamespace(returncode=0, stdout=""))
    assert rc == 1            # ranks that exit 0 without a result line are a failure too


def test_main_launches_before_touching_the_gpu(monkeypatch):
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: calls.append((n, argv)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0 and calls == [(4, ["--gpus", "4", "--steps", "2", "--warmup", "1"])]
