"""Child of tests/test_gpu_env_paths.py: ONE process walks a list of kernel-variant settings.

The CMLPL_* planner switches are read once per process; through round 5 every variant therefore cost its own
interpreter + library load + device start-up (35 child processes, a third of the GPU suite's run time).  The library
now re-reads the switches on request (cmlpl_debug_reload_switches, a test aid), so this process sets a job's
environment, reloads, and runs the job -- a pytest selection of the ordinary parity tests (`pytest.main`), or a Philox
trajectory whose printed lines the parent compares between two settings.

usage: _env_paths_child.py JOBS.json RESULTS.json      (results are re-written after every job: a progress file)
"""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(jobs_path, out_path):
    import pytest

    from cmlpl_amd import _lib
    from _philox_traj_child import run as run_traj
    jobs = json.load(open(jobs_path))
    lib = _lib.load()
    mine = set()
    results = {}
    for job in jobs:
        for k in mine:                                  # the previous job's switches
            os.environ.pop(k, None)
        mine = set(job["env"])
        os.environ.update(job["env"])
        _lib.check("cmlpl_debug_reload_switches", lib.cmlpl_debug_reload_switches())
        t0 = time.time()
        if job["kind"] == "pytest":
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
                rc = int(pytest.main(["-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"] + job["select"]))
            res = {"rc": rc, "tail": buf.getvalue()[-3000:]}
        else:
            try:
                res = {"rc": 0, "lines": run_traj(*job["args"])}
            except Exception as e:          # noqa: BLE001  (reported to the parent, which fails the test)
                import traceback
                res = {"rc": 1, "tail": traceback.format_exc()[-3000:], "lines": [repr(e)]}
        res["seconds"] = round(time.time() - t0, 2)
        results[job["name"]] = res
        with open(out_path + ".tmp", "w") as f:
            json.dump(results, f)
        os.replace(out_path + ".tmp", out_path)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
