"""The three BASELINE workloads at their TRUE depth and width against the CPU oracle (-m gpu): see tests/real_width_case.py.
Each case is its own process (dmae_vtp's package name collides with base_vtp's; the l14 model's memory leaves with it) and prints one
`REALWIDTH {...}` line with every measured deviation; the gates live next to the numbers in real_width_case.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "real_width_case.py")


def run_case(case, dev, env_extra=None, timeout=1500):
    env = dict(os.environ)
    env.pop("ANTMMF_HIP_LIB", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, SCRIPT, case, dev], capture_output=True, text=True, timeout=timeout, env=env)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("REALWIDTH ")]
    assert line, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    rep = json.loads(line[-1][len("REALWIDTH "):])
    print(line[-1])
    out = os.environ.get("ANTMMF_REAL_WIDTH_OUT")   # tools/gpu_*.sh: keep the report next to the other measurements of the run
    if out:
        with open(out, "a") as f:
            f.write(line[-1][len("REALWIDTH "):] + "\n")
    assert p.returncode == 0 and not rep["failed_gates"], rep
    return rep


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["l14", "b16", "vtp8", "vtp8t", "dmae12", "dmae12tpm"])
def test_real_width_step_vs_oracle(case):
    rep = run_case(case, "cuda:0")
    assert rep["full"]


SLOW = pytest.mark.skipif(not os.environ.get("ANTMMF_SLOW_TESTS"), reason="set ANTMMF_SLOW_TESTS=1 (2-3 min per case on the lane emulator)")


@SLOW
@pytest.mark.parametrize("case", ["l14", "vtp8", "dmae12"])
def test_real_width_plumbing_on_emulator(case):
    """the same script at toy dimensions on the CPU lane emulator: names, shapes, oracle calls and gates are exercised where there is no GPU"""
    emu = os.path.join(ROOT, "tests", "emu", "build", "libantmmf_emu.so")
    rep = run_case(case, "cpu", dict(ANTMMF_HIP_LIB=emu, ANTMMF_ALLOW_EMULATOR="1", ANTMMF_REAL_WIDTH_SMALL="1"), timeout=2400)
    assert not rep["full"]
