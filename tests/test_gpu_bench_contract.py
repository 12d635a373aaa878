"""bench.py prints exactly ONE JSON line on stdout with the fields the driver reads (task contract): metric / value /
unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload, plus `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "12", "--warmup", "3", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "patches/s" and d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"].startswith("f32") and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 256 * 1000.0 / d["ms_per_step"]) <= 1e-6 * d["value"]      # whole-job patches / time
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    # peak: the MFMA ceiling of the kernel's own mix of f32-input, three-piece bf16 (six MFMAs per product) and
    # two-piece fp16 (three) segments, between the pure cases
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 157.3 <= rf["peak"] <= 2500.0 / 3 + 1e-6
    seg = rf["flops_by_mfma_kind"]
    assert abs(sum(seg.values()) - rf["flops_per_launch"]) < 1.0
    assert seg.get("split2", 0.0) > 0.0 and seg.get("split", 0.0) > 0.0        # conv1's taps | conv0 + conv2 of the fused launch
    want_peak = sum(seg.values()) / (seg.get("f32", 0.0) / 157.3 + seg.get("split", 0.0) / (2500.0 / 6) +
                                     seg.get("split2", 0.0) / (2500.0 / 3))
    assert abs(rf["peak"] - want_peak) < 1e-6 * want_peak
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 1.0
    assert rf["launches_timed"] == 12
    assert rf["traffic"] is None or rf["traffic"] > 1e6     # null when the kernels changed since the recorded PMC pass
    assert len(d["roofline_others"]) == 2
    assert all(v == v for v in d["final_losses"].values())                               # finite


def _run_bench_ranks(args, env, timeout):
    """bench.py with its own ranks; exit code 3 is the start-up watchdog (cmlpl_amd.distributed: the rendezvous or the
    first collective did not finish in time -- seen once on a box that was loading RCCL for the first time): such a
    run is repeated ONCE, the library then being resident; any other failure is the test's."""
    for attempt in range(2):
        r = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
        if r.returncode != 3:
            break
        # a watchdog exit is made visible (pytest -rP / the captured output of a failing run): a RECURRING one is a
        # real rendezvous / first-collective hang, not a cold library
        print(f"[bench-retry] bench.py {' '.join(args)} ended with the start-up watchdog (exit 3) on attempt {attempt + 1}:\n"
              f"{r.stderr[-1500:]}", file=sys.stderr)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_retries.log"), "a") as f:
            f.write(f"attempt {attempt + 1}: bench.py {' '.join(args)} -> exit 3\n{r.stderr[-1500:]}\n")
    return r


def test_bench_self_launch_path():
    """`python bench.py --gpus N` typed directly starts its own ranks (cmlpl_amd/launch.py).  One GPU here, so the
    spawn path is exercised at N = 1 (CMLPL_BENCH_SPAWN=1 forces it): the child is a real rank with its own
    rendezvous environment, runs the data-parallel engine on RCCL at world size 1, and the parent relays its line."""
    env = dict(os.environ, CMLPL_BENCH_SPAWN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = _run_bench_ranks(["--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--workload", "B5",
                          "--global-batch", "64+512"], env, 900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["global_batch"] == 576
    assert d["config"]["parallelism"] == "dp1" and d["value"] > 0


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` as the driver types it for N > 1 without torchrun: the parent starts two ranks, they
    rendezvous, run the sharded step and rank 0's line comes back with n_gpus = 2 and the whole-job value.  One GPU
    here, so both ranks share it and the collectives run on gloo (CMLPL_ONE_GPU / CMLPL_DIST_BACKEND): this checks
    the wiring, not the speed."""
    env = dict(os.environ, CMLPL_ONE_GPU="1", CMLPL_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = _run_bench_ranks(["--gpus", "2", "--steps", "8", "--warmup", "2", "--workload", "B3"], env, 900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 1024
    assert d["config"]["parallelism"] == "dp2" and d["cpu_baseline"] is None
    assert abs(d["value"] - 1024 * 1000.0 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert all(v == v for v in d["final_losses"].values())
