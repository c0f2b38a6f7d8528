"""GPU: bench.py's contract - the `--gpus N` self-launcher (N ranks over gloo on this one-GPU box, the data-parallel path
of itts_hip/dp.py: replicate_packed / partition / gather) and the shape of the JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=dict(os.environ, **(env or {})),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_micro_line_single_gpu():
    j = run_bench("--micro", "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    assert j["n_gpus"] == 1 and j["metric"] == "audio_sec_per_sec" and j["value"] > 0 and j["scaling"] == "weak"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(j["roofline"])


def test_bench_gpus2_self_launch_over_gloo():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts 2 ranks as a child process group and
    relays ONE line with n_gpus = 2; both ranks share this box's GPU (backend gloo), weights arrive by broadcast."""
    j1 = run_bench("--micro", "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    j2 = run_bench("--gpus", "2", "--micro", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", env={"ITTS_DIST_BACKEND": "gloo"})
    assert j2["n_gpus"] == 2
    a1, a2 = j1["config"]["audio_sec_per_step_per_gpu"], j2["config"]["audio_sec_per_step_per_gpu"]
    assert abs(a1 - a2) < 1e-6  # weak scaling: the per-GPU work is unchanged, `value` aggregates both ranks
    # BASELINE config 4 (N x 32 utterances, data-parallel) rides in the same line at N > 1, config 3 at N = 1
    c4 = j2["also"]["config4_batch32_per_gpu"]
    assert c4["config"]["utterances_per_gpu"] == 32 and c4["n_gpus"] == 2 and c4["value"] > 0
    assert j1["also"]["config3_batch32"]["config"]["utterances_per_gpu"] == 32 and "product_loop" in j1["also"]
    d3 = j1["also"]["reference_default_mode_3_beams"]  # the reference's default generate() mode rides along too
    assert d3["value"] > 0 and d3["config"]["decode_batch"] == 3 * j1["config"]["decode_batch"]
    assert j2["dist"]["world_size"] == 2 and j2["dist"]["backend"] == "gloo" and j2["dist"]["collectives_in_timed_region"] == 0
