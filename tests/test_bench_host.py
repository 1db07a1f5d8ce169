"""Host-side logic of bench.py that needs no GPU: the N > 1 self-launch (`python bench.py --gpus N` without a launcher around it
must start the N ranks itself -- the driver's scaling run issues exactly that shape of command)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_command_only_when_no_launcher_is_around():
    b = _bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = b.self_launch_command(8, argv, {}, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv      # the ranks re-run THIS file with THESE flags
    # a rank started by a launcher (the driver's torch.distributed.run, or our own children) never launches again
    assert b.self_launch_command(8, argv, {"WORLD_SIZE": "8", "RANK": "3"}) is None
    assert b.self_launch_command(8, argv, {"RANK": "0"}) is None
    assert b.self_launch_command(1, argv, {}) is None
    # a free port is picked when none is given
    port = int(b.self_launch_command(2, argv, {})[b.self_launch_command(2, argv, {}, port=1).index("--master-port") + 1])
    assert 1024 < port < 65536


def test_self_launch_spawns_the_ranks_and_relays_their_exit_code(tmp_path):
    """The real spawn path, with a stand-in for the rank program: self_launch() must start torch.distributed.run with N children, each of
    which sees WORLD_SIZE / RANK (so it does not launch again), pass rank 0's stdout through and exit with the launcher's code."""
    script = tmp_path / "fake_bench.py"
    script.write_text(
        "import importlib.util, os, sys\n"
        "spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "b.__file__ = __file__\n"
        "b.self_launch(2, sys.argv[1:])\n"                                # parent: never returns; children: returns None
        "assert os.environ['WORLD_SIZE'] == '2'\n"
        "if os.environ['RANK'] == '0': print('LINE from rank 0 of', os.environ['WORLD_SIZE'], sys.argv[1:], flush=True)\n"
        "sys.exit(7 if '--fail' in sys.argv else 0)\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script), "--steps", "3"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("LINE from rank 0 of 2 ['--steps', '3']") == 1, r.stdout
    r = subprocess.run([sys.executable, str(script), "--fail"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
