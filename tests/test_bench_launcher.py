"""`python bench.py --gpus N` starts its N ranks itself (no torch.distributed.run around it): the parent process
makes no GPU call, gives every rank the rendezvous environment, forwards rank 0's JSON line and fails when any
rank fails.  Checked here with a stand-in rank program, and once end to end through the real argument parser."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (module import is cheap: torch is only imported inside main())


def _stub(tmp_path, body):
    path = tmp_path / "rank_stub.py"
    path.write_text(textwrap.dedent(body))
    return [sys.executable, str(path)]


def test_launcher_gives_every_rank_its_environment_and_forwards_rank_0(tmp_path, capsys):
    child = _stub(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        # a two-rank rendezvous over the address the launcher handed out (gloo: runs on the CPU)
        import torch, torch.distributed as dist
        dist.init_process_group(backend="gloo")
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        dist.destroy_process_group()
        print("noise before the line")
        if rank == 0:
            print(json.dumps({"n_gpus": world, "sum": int(t.item()), "argv": sys.argv[1:], "local_rank": os.environ["LOCAL_RANK"],
                              "addr": os.environ["MASTER_ADDR"], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
        else:
            print(json.dumps({"this is": "not rank 0"}))
    """)
    rc = bench.launch_ranks(2, ["--steps", "3"], child=child)
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0
    assert len(out) == 1, out  # exactly one line, rank 0's
    line = json.loads(out[0])
    assert line == {"n_gpus": 2, "sum": 3, "argv": ["--steps", "3"], "local_rank": "0", "addr": "127.0.0.1", "ipc": "0"}


def test_launcher_fails_when_a_rank_fails(tmp_path, capsys):
    child = _stub(tmp_path, """
        import os, sys
        if os.environ["RANK"] == "1":
            sys.exit(7)
        print('{"n_gpus": 2}')
    """)
    assert bench.launch_ranks(2, [], child=child) == 7
    assert capsys.readouterr().out.strip() == ""  # no line of a failed job is forwarded


def test_launcher_fails_when_rank_0_prints_nothing(tmp_path):
    assert bench.launch_ranks(2, [], child=_stub(tmp_path, "pass\n")) != 0


def test_a_rank_started_by_a_launcher_does_not_start_ranks_again(tmp_path):
    """With WORLD_SIZE set (torch.distributed.run, or our own launcher) `--gpus N` only names the job size."""
    probe = tmp_path / "probe.py"
    probe.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        bench.launch_ranks = lambda *a, **k: (_ for _ in ()).throw(SystemExit(99))
        sys.modules["torch"] = None  # main() stops at its first torch import: far enough for this check
        sys.argv = ["bench.py", "--gpus", "2"]
        try:
            bench.main()
        except ImportError:
            sys.exit(0)
    """))
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1")
    assert subprocess.call([sys.executable, str(probe)], env=env) == 0
    env.pop("WORLD_SIZE")
    assert subprocess.call([sys.executable, str(probe)], env=env) == 99
