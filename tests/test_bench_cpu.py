"""bench.py on a machine without a GPU: argument handling, the traffic model lookup, and the refusal to run
(there is no CPU fallback: the product path must fail loudly when there is no device)."""
import importlib.util
import json
import os
import subprocess
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_configs_are_baseline_json_configs():
    b = _bench()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert b.CONFIGS[1][0] == "three_sphere" and "three-sphere" in base["configs"][1] and "1024 spp" in base["configs"][1]
    assert b.CONFIGS[2][0] == "cube" and "cube.obj" in base["configs"][2]
    assert b.CONFIGS[3] == ("monkey", 1920, 1080, 1024, 8) and "low_poly_monkey" in base["configs"][3] and "8 bounces" in base["configs"][3]
    assert b.CONFIGS[4] == ("monkey", 3840, 2160, 4096, 8) and "3840\u00d72160" in base["configs"][4] and "4096 spp" in base["configs"][4]
    assert "1920\u00d71080" in base["configs"][1] and "1920\u00d71080" in base["configs"][3]
    assert b.weak_image(1920, 1080, 2) == (2720, 1530) and b.weak_image(1920, 1080, 4) == (3840, 2160) and b.weak_image(1920, 1080, 8) == (5424, 3051)


def test_traffic_lookup_resolves_the_drivers_shape():
    b = _bench()
    # the driver runs --steps 20 --warmup 5: one launch of 20 frames; profiles/traffic.json must resolve it
    t, src = b.measured_traffic("monkey", 1920, 1080, 1024, 8, 1920 * 1080, 20.0, True)
    assert t is not None and t > 20 * 12.0 * 1920 * 1080 and src
    # a frame count that was not profiled goes through the fitted line; a rank's share scales with its pixels
    t10, _ = b.measured_traffic("monkey", 1920, 1080, 1024, 8, 1920 * 1080, 10.0, True)
    t10h, _ = b.measured_traffic("monkey", 1920, 1080, 1024, 8, 1920 * 1080 // 2, 10.0, True)
    assert t10 is not None and 0.3 * t < t10 < 0.7 * t and 0.45 * t10 < t10h < 0.6 * t10
    assert b.measured_traffic("monkey", 640, 480, 16, 8, 640 * 480, 1.0, True) == (None, None)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "no GPU" in (r.stderr + r.stdout) and not [l for l in r.stdout.splitlines() if l.startswith("{")]
