"""Performance regression guard: the headline configuration over 100 steps must stay within 10 % of the figure committed under
profiles/ (profiles/perf_guard.json).  Part of the speed comes from compiler flags and code-generation properties of this hipcc
build (av_aloha_amd/build.py: no SLP vectoriser, -O2, DPP bound_ctrl), which nothing else would notice losing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_config2_throughput_within_10_percent_of_the_committed_figure():
    ref = json.load(open(os.path.join(ROOT, "profiles", "perf_guard.json")))["config2_steps100_env_steps_per_s"]
    best = 0.0
    for attempt in range(2):            # a second run if the first one falls short (a cold box, a busy host)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10", "--no-extras", "--no-cpu-baseline"],
                             capture_output=True, text=True, timeout=900, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert out.returncode == 0 and lines, out.stdout[-1000:] + out.stderr[-2000:]
        r = json.loads(lines[-1])
        assert r["is_headline_metric"] and r["dtype"] == "f32" and r["config"]["envs_total"] == 4096 and r["config"]["nan_envs"] == 0
        best = max(best, r["value"])
        if best >= 0.9 * ref:
            break
    assert best >= 0.9 * ref, f"config 2 runs at {best:.0f} env-steps/s, below 0.9 x the committed {ref} (profiles/perf_guard.json)"
