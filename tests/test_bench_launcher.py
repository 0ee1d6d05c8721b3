"""`python bench.py --gpus N` must work as typed (the driver runs exactly that): with no WORLD_SIZE in the environment
the process launches N ranks itself.  Rehearsed here on the CPU (gloo, --dry-run: no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_dry_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["dry_run"] is True


def test_bench_launcher_does_not_touch_the_gpu_in_the_parent():
    """The launching parent must not import torch / the HIP library before it starts its children."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def launch_children")]
    assert "import torch" not in head and "import kmerlr_amd" not in head
