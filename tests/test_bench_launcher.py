"""`python bench.py --gpus N` outside a launcher starts its own ranks: the command line it builds."""
import importlib.util
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _FakeProc:
    def __init__(self, cmd, **kw):
        _FakeProc.cmd = list(cmd)
        line = {"metric": "m", "value": 1.0, "n_gpus": 2}
        self.stdout = io.StringIO("some rank chatter\n" + json.dumps(line) + "\n")

    def wait(self):
        return 0


def test_launcher_passes_arguments_through(monkeypatch, capsys):
    bench = _load_bench()
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", _FakeProc)
    monkeypatch.setattr(bench, "cpu_baseline", lambda D, dt: {"value": 2.0, "cores": 1, "kind": "port"})
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", "windows", "--backend", "gloo",
            "--problems", "8"]
    try:
        bench.main(argv)
    except SystemExit as e:
        assert e.code == 0
    cmd = _FakeProc.cmd
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:k + 1 + len(argv)] == argv          # every argument, in order
    assert cmd[-1] == "--no-cpu"                        # the ranks do not repeat the CPU baseline
    captured = capsys.readouterr()
    out = captured.out.strip().splitlines()
    assert len(out) == 1                                # stdout carries the JSON line alone
    assert "some rank chatter" in captured.err          # what else the ranks print goes to stderr
    line = json.loads(out[-1])
    assert line["n_gpus"] == 2 and line["cpu_baseline"]["value"] == 2.0 and "launcher" in line


def test_under_a_launcher_the_world_size_must_match(monkeypatch):
    bench = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", "4")
    try:
        bench.main(["--gpus", "2", "--no-cpu"])
    except SystemExit as e:
        assert "WORLD_SIZE=4" in str(e.code)
    else:
        raise AssertionError("expected a refusal")
