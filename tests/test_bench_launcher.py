"""bench.py --gpus N launches its own ranks: the parent makes no GPU call, starts
``python -m torch.distributed.run`` as a child, relays rank 0's JSON line and propagates the exit status.  The
``--backend gloo --dry-run`` path drives that same launcher and the product's tile sharding + FrameExchange on host
tensors, so it runs here without a GPU."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _bench(*args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("n,image,counts", [(2, 200, [8, 8]), (3, 200, [6, 5, 5])])
def test_self_launch_dry_run(n, image, counts):
    r = _bench("--gpus", str(n), "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "1", "--image", str(image), "--tile", "64")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["dry_run"] is True and out["frame_ok"] is True
    assert out["tiles_per_rank"] == counts and out["steps"] == 3 and out["warmup"] == 1


def test_refuses_to_report_fewer_gpus_than_asked():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("a full node: the launch would succeed")
    r = _bench("--gpus", "8", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())      # no valid-looking line
    assert "GPU(s) visible" in r.stderr


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--backend", "gloo", "--dry-run"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
