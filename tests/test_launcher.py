"""`python bench.py --gpus N` with WORLD_SIZE unset starts its own rank processes (fresh children, environment set as
torch.distributed.run would): covered here on CPU with world size 2 over gloo."""
import json
import os

import bench

PROBE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "launcher_probe.py")


def test_spawn_ranks_starts_a_working_process_group(capfd):
    rc = bench.spawn_ranks(2, ["--tag", "x"], script=PROBE)
    assert rc == 0
    line = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got == {"world": 2, "sum": 3.0, "argv": ["--tag", "x"]}


def test_spawn_ranks_reports_a_failing_rank(capfd):
    assert bench.spawn_ranks(2, ["--fail"], script=PROBE) == 3


def test_spawn_ranks_terminates_the_survivors_when_a_rank_dies(capfd):
    """A rank that dies early must not leave its peers hanging in a collective / rendezvous: the launcher watches every child, terminates the rest
    and returns the failing rank's exit code -- promptly."""
    import time
    t0 = time.monotonic()
    assert bench.spawn_ranks(2, ["--crash-early"], script=PROBE) == 7
    assert time.monotonic() - t0 < 60.0
    assert "crashing before the rendezvous" in capfd.readouterr().err      # every rank's stderr is kept


def test_gpus_gt_1_without_rank_env_goes_through_the_launcher(monkeypatch):
    calls = {}
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv, **kw: calls.setdefault("n", n) and 0)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr("sys.argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code in (0, 2) or e.code is None or e.code == 2
    assert calls["n"] == 2
