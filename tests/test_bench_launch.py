"""bench.py --gpus N without a launcher must start N ranks itself (the flag was parsed and ignored).  On a box
without a GPU every rank refuses to run; what is checked here is that N fresh rank processes were started (before any
GPU call) and that the failure reaches the exit status."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--small", "--backend", "gloo"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                   # no JSON line without a measurement
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr[-2000:]


def _run_watchdog(body):
    code = ("import sys, time\nsys.path.insert(0, %r)\nimport bench\n" % ROOT) + body + "\ntime.sleep(30)\nprint('NOT REACHED')\n"
    return subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)


def test_watchdog_prints_the_fallback_line_and_always_exits():
    """the sharded leg's watchdog died of a NameError in its timer thread (the line builders were defined
    after the leg) and os._exit was never reached.  The watchdog now is bench.arm_watchdog: whatever its last_words do --
    print the fallback line, raise, or call a name that does not exist yet -- the process ends with the given status."""
    r = _run_watchdog("bench.arm_watchdog(0.2, lambda: print('{\"fallback\": 1}', flush=True), bench.EXIT_SHARDED_FAILED)")
    assert r.returncode == 3 and r.stdout.strip() == '{"fallback": 1}', (r.returncode, r.stdout, r.stderr)
    r = _run_watchdog("def lw():\n    raise RuntimeError('boom')\nbench.arm_watchdog(0.2, lw, bench.EXIT_SHARDED_FAILED)")
    assert r.returncode == 3 and "NOT REACHED" not in r.stdout and "boom" in r.stderr, (r.returncode, r.stdout, r.stderr)
    r = _run_watchdog("def lw():\n    not_defined_yet()\nbench.arm_watchdog(0.2, lw, 7)")
    assert r.returncode == 7 and "NOT REACHED" not in r.stdout, (r.returncode, r.stdout, r.stderr)
    # disarmed in time: nothing happens
    r = subprocess.run([sys.executable, "-c", "import sys, time\nsys.path.insert(0, %r)\nimport bench\nd = bench.arm_watchdog(0.5, lambda: None, 3)\n"
                        "d.cancel()\ntime.sleep(1.0)\nprint('alive')" % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "alive"


def test_line_builders_are_defined_before_the_sharded_leg():
    """the fallback line is built by make_line(): it must exist when the watchdog is armed"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("def make_line(") < src.index("arm_watchdog(a.sharded_timeout") and src.index("def roofline(") < src.index("arm_watchdog(a.sharded_timeout")
