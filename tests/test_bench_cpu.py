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


def test_reference_default_workload_and_argument_defaults():
    b = _bench()
    # the reference's own defaults: scene 0, 1000x800 (src/camera.cu:4-5), 100 spp x 5 bounces (src/main.cu:318-330)
    assert b.CONFIGS["ref0"] == ("reference_scene0", 1000, 800, 100, 5)
    a = b.parse_args(["--config", "ref0"])
    assert (a.scene, a.width, a.height, a.spp, a.limit) == ("reference_scene0", 1000, 800, 100, 5)
    a = b.parse_args([])
    assert (a.scene, a.width, a.height, a.spp, a.limit, a.gpus, a.partition) == ("monkey", 1920, 1080, 1024, 8, 1, "lists")
    a = b.parse_args(["--config", "2", "--spp", "16", "--scene", "sphere50k"])
    assert (a.scene, a.width, a.spp) == ("sphere50k", 1920, 16)


def test_rccl_failure_is_a_failed_run_unless_the_fallback_is_asked_for(monkeypatch):
    """a gloo measurement must never pass for an xGMI one: without --allow-gloo-fallback an RCCL initialisation
    failure ends every rank with exit code 3"""
    import pytest
    import torch.distributed as dist
    b = _bench()
    calls = []

    def fake_init(backend, **kw):
        calls.append(backend)
        if backend == "nccl":
            raise RuntimeError("ncclInvalidUsage: duplicate GPU")

    class Exit(Exception):
        pass

    def fake_exit(code):
        raise Exit(code)
    monkeypatch.setattr(dist, "init_process_group", fake_init)
    monkeypatch.setattr(dist, "destroy_process_group", lambda: None)
    monkeypatch.setattr(os, "_exit", fake_exit)
    with pytest.raises(Exit) as e:
        b.init_process_group(b.parse_args(["--gpus", "2"]), "cpu", 2)
    assert e.value.args[0] == b.EXIT_NO_RCCL == 3 and calls == ["nccl"]
    calls.clear()
    backend, note = b.init_process_group(b.parse_args(["--gpus", "2", "--allow-gloo-fallback"]), "cpu", 2)
    assert backend == "gloo" and calls == ["nccl", "gloo"] and "failed to initialise" in note
    calls.clear()
    assert b.init_process_group(b.parse_args(["--gpus", "2", "--backend", "gloo"]), "cpu", 2) == ("gloo", None) and calls == ["gloo"]


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


def test_capi_multi_never_runs_under_an_rccl_barrier():
    """if the gloo side group cannot be created, the waiting ranks must not sit in an RCCL barrier (a kernel
    spinning on GPUs 1..N-1) while rank 0's child renders on those GPUs: the child then runs after destroy_process_group."""
    import bench
    f = bench.capi_runs_after_group
    assert f(True, 8, "nccl", False) is True          # RCCL, no side group: after the group is gone
    assert f(True, 8, "nccl", True) is False          # RCCL + gloo side group: under the host-side barrier
    assert f(True, 2, "gloo", False) is False         # a gloo process group waits host-side by itself
    assert f(False, 8, "nccl", False) is False        # nothing to run
    assert f(True, 1, "none", False) is False         # one GPU: --capi-multi runs inline
    a = bench.parse_args(["--gpus", "2", "--no-host-group", "--self-test"])
    assert a.no_host_group and a.self_test and not a.no_self_test
