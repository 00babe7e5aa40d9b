"""Host logic: the exact edge tables a plan is built from (sphost::build_thresholds, found from analytic guesses) are bit-identical
to a full bisection over all positive doubles with the same pixel arithmetic, over a spread of gains, ranges, norms and LUT lengths,
and the indices are monotone in the neighbourhood of every edge (the assumption a bisection rests on)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_edge_tables_equal_full_bisection(tmp_path):
    pkg = os.path.join(ROOT, "spectroplot-js_amd")
    exe = str(tmp_path / "thresholds_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(pkg, "csrc"), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "thresholds_check.cpp"), os.path.join(pkg, "csrc", "sp_host.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 differing edges" in r.stdout, r.stdout
    # ... and the index is monotone where a bisection needs it: no double within 64 ulps of an edge (4096 for the default request) lies on
    # the wrong side of it (the restated log10 does not wobble around the steps)
    assert " 0 on the wrong side of their edge" in r.stdout and "doubles around the edges probed" in r.stdout, r.stdout
