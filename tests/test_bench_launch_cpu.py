"""bench.py's multi-GPU plumbing on a box without GPUs: `--gpus N` starts N ranks (or refuses loudly), never prints a
line for fewer GPUs than asked for, and the job's time / ray totals are reduced over the ranks as the metric says
(MAX of the elapsed times, SUM of the rays).  The reference's precedent for the sharding is thrender_no_rayon
(camera.rs:319-328)."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

BENCH = os.path.join(ROOT, "bench.py")


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


@pytest.mark.timeout(120)
def test_gpus_2_without_two_gpus_fails_loudly():
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "visible" in r.stderr
    assert _json_lines(r.stdout) == []          # in particular no line that says n_gpus: 1


@pytest.mark.timeout(120)
def test_gpus_flag_must_agree_with_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "2 rank" in r.stderr
    assert _json_lines(r.stdout) == []


def test_launcher_command(monkeypatch):
    """N ranks through torch.distributed.run on 127.0.0.1, the flags passed through, the child's status returned."""
    seen = {}

    def fake_call(cmd, **kw):
        seen["cmd"] = cmd
        return 7

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(subprocess, "call", fake_call)
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "2", "--warmup", "1"])
    cmd = seen["cmd"]
    assert rc == 7
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) <= 65535
    at = cmd.index(BENCH)
    assert cmd[at + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    # more ranks than GPUs: refused before anything is started
    seen.clear()
    assert bench.launch_ranks(8, ["--gpus", "8"]) != 0 and "cmd" not in seen


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launch(monkeypatch):
    """The real launcher with two ranks on a box without GPUs: every rank refuses to render, and the launch reports it."""
    if torch.cuda.device_count() >= 1:
        pytest.skip("needs a box without GPUs")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)   # (gets the launcher past its own check)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.launch_ranks(2, ["--gpus", "2", "--steps", "1", "--warmup", "0"]) != 0


def _reduce_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = bench.reduce_over_ranks(1.0 + rank, 10.0 * (rank + 1), 3.0 * (rank + 1), dist, "cpu")
        with open(os.path.join(out_dir, f"r{rank}.json"), "w") as fh:
            json.dump(got, fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_time_is_the_slowest_ranks_and_rays_add_up(tmp_path, world):
    port = 29700 + (os.getpid() % 1500) + world
    mp.spawn(_reduce_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        with open(tmp_path / f"r{r}.json") as fh:
            elapsed, rays, paths = json.load(fh)
        assert elapsed == float(world)                       # MAX over ranks
        assert rays == 10.0 * world * (world + 1) / 2        # SUM over ranks
        assert paths == 3.0 * world * (world + 1) / 2
    assert bench.reduce_over_ranks(1.5, 2.0, 3.0, None, "cpu") == (1.5, 2.0, 3.0)


def test_traffic_model_agrees_with_the_profiled_traffic_on_the_headline_workload():
    """bench.py's roofline_hbm is priced from the run's own ray and path counts (traffic_model); the committed rocprofv3
    pass of the same command (FETCH_SIZE doubled + WRITE_SIZE, profiles/traffic.json) must agree within 15 % on C2,
    where every byte is streamed (no gathers).  Both are per ray of the dominant kernel's launches."""
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
        prof = json.load(fh)["C2"]
    with open(os.path.join(ROOT, "profiles", "r03_C2_bench.json")) as fh:
        line = json.loads([l for l in fh if '"metric"' in l][-1])
    counts = line["roofline_hbm"]["traffic_model"]
    model = bench.traffic_model("k_bounce_lockstep", [], counts["rays"], counts["paths"])
    assert prof["kernel"] == "k_bounce_lockstep"
    assert abs(model["bytes_per_ray"] / prof["hbm_bytes_per_ray"] - 1.0) < 0.15, (model["bytes_per_ray"], prof["hbm_bytes_per_ray"])
    # and the line itself carries both, with the commit the profile was taken at
    assert line["roofline_hbm"]["traffic_model"]["bytes_per_ray"] == pytest.approx(model["bytes_per_ray"], rel=1e-9)
    assert bench.traffic_model("k_finish", [], 10.0, 2.0) is None and bench.traffic_model("k_bounce_lockstep", [], 0.0, 0.0) is None
