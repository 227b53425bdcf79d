"""N > 1 path of bench.py on CPU: two ranks over gloo (world_size 2) through the real launcher, stand-in step.
Checks the rank/segment sharding, the barrier-bracketed timing and the max-over-ranks aggregation."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["scaling"] == "weak" and j["first_frame_rank0"] == 0
    # rank 1 sleeps 20 ms per step, rank 0 10 ms: the aggregate time is the slower rank's
    assert j["ms_per_step"] >= 19.0
    assert abs(j["value"] - 2 * j["frames_per_step"] * 5 / (j["ms_per_step"] * 5e-3)) < 1e-6 * j["value"] + 1e-6
    # the end-to-end leg runs on EVERY rank between barriers: frames of both ranks over the slower rank's time
    # (rank 1 sleeps 40 ms), host threads shared between the ranks
    e = j["e2e_gpu_entropy"]
    assert e["ranks"] == 2 and e["host_threads_per_rank"] >= 1
    assert 2 * j["frames_per_step"] / 0.2 < j["e2e_gpu_entropy_frames_per_s"] <= 2 * j["frames_per_step"] / 0.04 + 1e-6


def test_single_rank_dry_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu", "--steps", "2"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["ms_per_step"] >= 9.0


def test_segment_sharding_is_disjoint():
    sys.path.insert(0, ROOT)
    import bench
    starts = [bench.segment_of_rank(r, 32) for r in range(8)]
    assert starts == [0, 32, 64, 96, 128, 160, 192, 224]


def test_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with NO external launcher: the script starts its two ranks (the flag was parsed and
    ignored, a driver-run `--gpus 8` would have measured one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ms_per_step"] >= 19.0


def test_world_size_without_gpus_flag_is_taken_from_the_launcher():
    """a launcher run that sets WORLD_SIZE without repeating --gpus must not exit"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu", "--steps", "1"], env=env, capture_output=True, text=True,
                         timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]


def test_a_dying_rank_takes_the_others_down(tmp_path):
    """spawn_ranks polls its children: when one exits non-zero the rest are terminated instead of waiting in a barrier"""
    import time
    sys.path.insert(0, ROOT)
    code = ("import os, sys, time\n"
            "sys.exit(3) if os.environ['RANK'] == '1' else time.sleep(600)\n")
    script = tmp_path / "bench.py"
    src = open(os.path.join(ROOT, "bench.py")).read()
    script.write_text(src.replace("def main():\n    args = parse()", "def main():\n    args = parse()\n    if 'RANK' in os.environ:\n        exec(%r)" % code, 1))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, str(script), "--gpus", "2", "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode != 0 and "rank(s) failed" in out.stderr and time.time() - t0 < 60


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=env, capture_output=True,
                         text=True, timeout=120, cwd=ROOT)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in out.stderr
