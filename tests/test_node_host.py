"""The Node.js host layer: N-API addon + HipWorker (the object handed to the reference through `workerOrUrl`)."""
import os
import shutil
import subprocess

import pytest

from __graft_entry__ import ROOT, build

pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
ADDON = os.path.join(ROOT, "spectroplot-js_amd", "lib", "spectroplot_hip.node")


def _node(script, *args, timeout=600, node_flags=(), env=None):
    if not os.path.exists(ADDON):
        build()
    return subprocess.run(["node"] + list(node_flags) + [os.path.join(ROOT, "tests", "js", script)] + list(args), capture_output=True,
                          text=True, timeout=timeout, env=None if env is None else dict(os.environ, **env))


def test_addon_loads_and_host_helpers_match_reference_kats():
    out = _node("check_addon_cpu.js")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "addon cpu checks ok" in out.stdout


def test_caller_side_parameter_helpers_match_reference():
    out = _node("check_params.js")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "params checks ok" in out.stdout


def test_side_output_consumers_match_reference_draw_calls():
    """js/consumers.js (colour ramp, dBfs markers, histogram outlines) call for call against the reference's drawColorRamp /
    drawHistograms run on a recording 2-D context (tests/golden/consumers.json)."""
    out = _node("check_consumers.js")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "consumers checks ok" in out.stdout


def test_software_rasteriser_replays_the_references_gauge_and_ramp_calls():
    """js/raster.js: the reference's recorded gauge rectangles and colour-ramp calls replayed on the software surface, pixel for pixel
    against a per-pixel model; composePlot's layout."""
    out = _node("check_raster.js")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "raster checks ok" in out.stdout


def test_slice_merge_and_gauges_reproduce_the_references_own_caller():
    """renderSliced / stripPlacement / gaugeColumns against what the reference's startWorkers + processData did with 1, 2 and 8
    workers (tests/golden/caller.json), the workers being the JavaScript oracle here."""
    out = _node("check_caller.js", "oracle")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "caller checks ok (oracle): 9" in out.stdout


@pytest.mark.gpu
def test_hip_worker_pool_reproduces_the_references_own_caller():
    """The same with a pool of HipWorker instances, the product's workers."""
    out = _node("check_caller.js", "hip")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "caller checks ok (hip): 9" in out.stdout


@pytest.mark.gpu
def test_hip_worker_reproduces_golden_vectors():
    import re
    import subprocess as sp
    sp.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])        # the librccl test double of the multi-member exchange
    out = _node("check_hip_worker.js")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bit-for-bit" in out.stdout
    # the sliced cases with 2, 3 and 8 workers also went through addon.groupRender's multi-member RCCL exchange (tests/cpp/rccl_shim.cpp)
    m = re.search(r"(\d+) through the multi-member exchange", out.stdout)
    assert m and int(m.group(1)) >= 9, out.stdout[-500:]


@pytest.mark.gpu
def test_named_requests_and_the_command_line_renderer():
    """HipWorker.renderNamed (sp_render_named through N-API) on the golden vectors, plan reuse, and js/cli.js on a capture file."""
    out = _node("check_named.js")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "named requests ok" in out.stdout


@pytest.mark.gpu
def test_config5_at_full_size_through_the_javascript_boundary():
    """BASELINE config 5 (64 MSample cs12, n = 8192, zoom x8, 2 GiB of RGBA) through renderSliced with two HipWorker slices."""
    out = _node("check_config5_full.js", timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "config 5 at full size through renderSliced ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode,env", [("default", {}), ("nopin", {"SPECTROPLOT_HIP_POOL_PINNED_MB": "0"}),
                                      ("tight", {"SPECTROPLOT_HIP_POOL_PINNED_MB": "8", "SPECTROPLOT_HIP_POOL_KEEP_MB": "16"})])
def test_reply_image_pool_keeps_to_its_limits(mode, env):
    """The addon's pool of reply images: recycled blocks carry correct images, page-locked bytes and kept bytes stay below the limits
    set through the environment."""
    out = _node("check_pool.js", mode, node_flags=["--expose-gc"], env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pool checks ok (%s)" % mode in out.stdout
