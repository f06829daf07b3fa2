"""The reference's own signatures on the host shims: ORBmatcher::SearchByProjection x 3 / SearchByBruceMatching and
Optimizer::PoseOptimization / CFSE3ObjStateOptimization / ObjectLocalBundleAdjustment as templates over a Frame-shaped type
(tests/cpp/frame_view.h carries the reference's member names), run on the GPU and compared with the CPU checker through an
independent marshalling (tests/cpp/shim_ref_check.cpp)."""
import json
import os
import subprocess

import pytest

import oracle_lib  # noqa: F401  (builds oracle/liboracle.so when missing)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "shim_ref_check")


def _build():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "pointslot_amd", "host"), "-I", os.path.join(ROOT, "include"),
                           EXE + ".cpp", "-o", EXE, "-L", os.path.join(ROOT, "oracle"), "-loracle", "-L", os.path.join(ROOT, "pointslot_amd"), "-lpointslot_hip",
                           "-pthread", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath," + os.path.join(ROOT, "pointslot_amd"), "-Wl,-rpath-link,/opt/rocm/lib"])


def test_reference_signatures_compile_against_frame_shaped_types():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_reference_signatures_match_the_cpu_checker():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    summary = json.loads(lines[-1])
    assert summary["failed"] == 0 and summary["checks"] >= 12, out.stdout
    for name in ("SearchByProjection(CurrentFrame, LastFrame, th, bMono)", "SearchByProjection(F, vpMapPoints, th)",
                 "SearchByProjection(F, nOrder, vpMapObjectPoints, th)", "SearchByBruceMatching(", "Optimizer::PoseOptimization(Frame*)",
                 "Optimizer::CFSE3ObjStateOptimization(Frame*", "Optimizer::ObjectLocalBundleAdjustment(ObjectKeyFrame*"):
        assert any(l.startswith("ok") and name in l for l in lines), name
