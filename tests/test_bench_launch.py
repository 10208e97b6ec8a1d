"""bench.py --gpus N without a launcher starts torch.distributed.run itself, as a child process, before it touches a GPU,
and relays the child's output and exit code (the driver's contract command is `python bench.py --gpus N ...`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_relays_child_output_and_exit_code(tmp_path):
    stub = tmp_path / "stub_launcher.py"
    stub.write_text(
        "import json, sys\n"
        "print(json.dumps({'stub_argv': sys.argv[1:]}), flush=True)\n"
        "sys.exit(7)\n")
    env = dict(os.environ, SARPRO_BENCH_LAUNCHER=str(stub))
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--mode", "stripe"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 7, p.stderr[-500:]
    import json
    argv = json.loads(p.stdout.strip().splitlines()[-1])["stub_argv"]
    # one rank per GPU on 127.0.0.1, then bench.py with the caller's own arguments
    assert "--nproc-per-node=2" in argv and "127.0.0.1" in argv
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "2", "--steps", "1", "--warmup", "0", "--mode", "stripe"]


def test_launcher_is_not_used_for_one_gpu_or_inside_a_launch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if args.gpus > 1 and "WORLD_SIZE" not in os.environ:' in src
    assert "os.exec" not in src  # never replace a process that may have touched the GPU runtime
