"""bench.py's own launcher: a bare `python bench.py --gpus N` starts its N ranks as fresh child processes, relays
rank 0's single JSON line and the ranks' exit code.  Runs the launcher path with a stub body (no GPU work, gloo on the CPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 4])
def test_bare_gpus_n_starts_its_own_ranks_and_prints_one_line(n):
    r = run(["--gpus", str(n), "--stub-body"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line on stdout and nothing else
    out = json.loads(lines[0])
    assert out["world_size_seen"] == n and out["n_gpus"] == n and out["replicas_identical"] is True
    assert out["value"] == n * (n + 1) / 2  # the all-reduce really spanned the n ranks
    assert out["launched_by"] == "bench.py"


@pytest.mark.parametrize("code", [3, 4])
def test_a_failing_rank_ends_the_run_with_its_exit_code(code):
    """Exit codes 3 (replicas diverged) and 4 (a rank's graph self-check raised) must reach the caller, and the surviving rank,
    which waits in a barrier for the failed one, must not be left behind."""
    r = run(["--gpus", "2", "--stub-body"], {"PUNET_BENCH_STUB_FAIL": f"1:{code}"}, timeout=120)
    assert r.returncode == code, (r.returncode, r.stderr[-2000:])
    assert "exit codes by rank" in r.stderr


def test_under_an_external_launcher_the_world_sizes_must_agree():
    r = run(["--gpus", "4", "--stub-body"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in r.stderr


def test_single_process_default_is_untouched_by_the_launcher():
    r = run(["--stub-body"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 1 and out["launched_by"] == "external launcher"


def test_roofline_traffic_is_found_in_the_committed_counter_files():
    """bench.py's roofline.traffic is read from the newest committed PMC passes (profiles/rNN/pmc_hbm_traffic*.json).  The kernels' names
    carry their template arguments and gained one in rounds 3 and 5: a lookup that no longer matches must not pass as a silent
    `traffic: null` in the driver's line (it did for a few commits in round 5).  The x6 file must hold the forward, per-image,
    bf16-plane instantiation, with a plausible byte count, the commit it was collected at and the digest of the kernel sources."""
    sys.path.insert(0, ROOT)
    import bench
    traffic, note = bench.traffic_from_profiles("x6")
    assert traffic is not None, note
    assert 1e8 < traffic < 1e9, traffic                      # algorithmic 250 MB per launch; halo re-reads on top
    assert "conv3x3_x6b_kernel<3, false" in note and "true" not in note.split(")")[0], note
    assert "collected at commit" in note and "kernel sources unchanged since that collection: " in note
    t32, n32 = bench.traffic_from_profiles("f32")
    assert t32 is not None, n32


def test_inference_hbm_report_reads_the_committed_counter_file():
    """configs[4]'s hbm_report: the counted bytes per tile come from profiles/rNN/pmc_hbm_traffic_inference_1024.json."""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.inference_hbm_report(1024, 1024, 10.5)
    assert r["bound_in_this_design"] == "mfma" and r["algorithmic_min_bytes"] == int(78.1e6 + 9 * 1024 * 1024)
    assert 1e10 < r["counted_hbm_bytes_per_step"] < 3e10, r.get("counted_note")
    assert 0.1 < r["counted_frac_of_hbm_peak"] < 0.5
