"""`python bench.py --gpus N` as typed (no torchrun around it): the parent must start N ranks as a CHILD
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, relay rank 0's one JSON line
and return the child's exit code — without ever touching the GPU itself."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeProc:
    def __init__(self, cmd, env, rc, lines):
        self.cmd, self.env, self.rc = cmd, env, rc
        self.stdout = io.StringIO("".join(lines))

    def wait(self):
        return self.rc


def _run(monkeypatch, rc, lines, argv):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    import subprocess
    seen = {}

    def fake_popen(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return FakeProc(cmd, env, rc, lines)
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    out = io.StringIO()
    monkeypatch.setattr(sys, "stdout", out)
    code = bench.self_launch(4)
    return code, out.getvalue(), seen


def test_parent_starts_the_ranks_and_relays_rank0(monkeypatch):
    line = json.dumps({"metric": "queries/sec", "value": 1.0, "n_gpus": 4})
    code, out, seen = _run(monkeypatch, 0, ["some library chatter\n", line + "\n"], ["--gpus", "4", "--steps", "3"])
    assert code == 0 and out.strip() == line
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_child_failure_is_the_parents_exit_code(monkeypatch):
    code, out, _ = _run(monkeypatch, 3, ["Traceback ...\n"], ["--gpus", "4"])
    assert code == 3 and out.strip() == ""
    code, out, _ = _run(monkeypatch, 0, [], ["--gpus", "4"])       # exit 0 without a line is still a failure
    assert code == 1
