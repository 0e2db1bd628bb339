"""`bench.py --gpus 1` on the device: the contract line the driver reads.

The default invocation must print ONE short JSON line (last line of stdout) that carries the contract keys, `roofline` and `cpu_baseline`, write the tables
behind it to bench_detail.json, and a second run must reproduce `value` within 3 % — the N = 1 point of a scaling run has to agree with the plain bench."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    return last, r.stdout


@pytest.mark.gpu
def test_default_bench_line_parses_is_short_and_reproduces():
    last, stdout = _run(["--steps", "20", "--warmup", "5"])
    assert len(last) <= 6000, len(last)
    assert len(stdout) <= 8000, "stdout beyond the contract line: a log tail would cut the line (%d bytes)" % len(stdout)
    line = json.loads(last)
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["unit"] == "mel-frames/s" and line["dtype"] == "bf16"
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["config"]["per_gpu_frames_per_step"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    rf = line["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    for leg in ("hifigan", "xvapitch_c5"):
        assert line[leg]["ms_per_step"] > 0 and "error" not in line[leg], line[leg]
    detail = json.load(open(os.path.join(ROOT, line["detail"])))
    assert detail["value"] == line["value"] and "by_kernel" in detail["roofline"]["all_gemm"]
    # the N = 1 number is reproducible: a second, FastPitch-only run agrees within 3 %
    last2, _ = _run(["--steps", "20", "--warmup", "5", "--no-hifigan", "--no-xvapitch", "--no-fp32-parity", "--no-cpu-baseline", "--no-roofline"])
    v2 = json.loads(last2)["value"]
    assert abs(v2 - line["value"]) <= 0.03 * line["value"], (v2, line["value"])
