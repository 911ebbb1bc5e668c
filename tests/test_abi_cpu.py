"""The C-ABI shared library on a machine without a GPU: it loads, exports every
symbol ``include/vo_hip.h`` declares, and rejects bad arguments before touching
the device.  Also: the host-side pieces of the package that need no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from visual_odometry_amd import _lib
    names = _declared()
    assert len(names) >= 15
    assert names == _lib.EXPORTS                      # the ctypes table covers the header exactly
    lib = _lib.load()
    for n in names:
        assert getattr(lib, n) is not None
    assert _lib.version() == _lib.ABI_VERSION == 301


def test_bad_arguments_are_rejected_without_a_gpu():
    from visual_odometry_amd import _lib
    lib = _lib.load()
    assert lib.vo_kabsch_batch(None, None, None, None, None, None, 1, 3, None) == _lib.VO_ERR_INVALID_ARG
    assert b"null" in lib.vo_last_error()
    assert lib.vo_disparity_to_depth(None, None, 10, 1.0, 1, None) == _lib.VO_ERR_INVALID_ARG
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.vo_corr_volume_2d(p, p, p, 1, 2, 2, 48, None) == _lib.VO_ERR_UNSUPPORTED   # D % 32 != 0
    assert b"multiple of 32" in lib.vo_last_error()
    assert lib.vo_corr_lookup_2d(p, None, None, None, p, p, 0, 1, 2, 2, 2, 2, 1, 7, None) == _lib.VO_ERR_UNSUPPORTED
    assert lib.vo_convex_upsample(p, p, p, 1, 2, 2, 2, 2, 3, 1.0, None) == _lib.VO_ERR_UNSUPPORTED
    assert lib.vo_corr_pyramid_2d(p, None, p, None, 1, 4, 4, None) == _lib.VO_ERR_INVALID_ARG
    with pytest.raises(ValueError):
        _lib.call("vo_keypoints_to_3d", None, 0, None, None, None, 1, 1, 4, 4, None)


def test_ops_refuse_cpu_tensors():
    import torch
    from visual_odometry_amd import ops
    with pytest.raises(ValueError, match="GPU tensor"):
        ops.disparity_to_depth(torch.zeros(8), 1.0)
    with pytest.raises(ValueError, match="GPU tensor"):
        ops.corr_volume_2d(torch.zeros(1, 2, 2, 32), torch.zeros(1, 2, 2, 32))


def test_plugins_need_a_gpu_and_say_so():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from visual_odometry_amd.geometry import HipCameraSvdXform
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipCameraSvdXform().compute_camera_xform(np.zeros((4, 3)), np.ones((4, 3)))


def test_host_rotation_fit_matches_lapack():
    """The 3x3 rotation fit of the Kabsch kernel, compiled for the host."""
    from oracle import geometry_ref as ref
    from tests.golden import cases
    so = os.path.join(ROOT, "tests", "_host_checks.so")
    if not os.path.exists(so):
        from visual_odometry_amd.build import build_host_checks
        build_host_checks()
    lib = ctypes.CDLL(so)
    fit = lib.host_rotation_from_covariance
    fit.argtypes = [ctypes.c_void_p, ctypes.c_void_p]

    def rot(P, Q):
        H = np.ascontiguousarray((P - P.mean(0)).T @ (Q - Q.mean(0)))
        R = np.zeros((3, 3))
        fit(H.ctypes.data, R.ctypes.data)
        return R

    for name, (P, Q, _) in cases.kabsch_cases().items():
        R = rot(P, Q)
        assert np.abs(R - ref.kabsch(P, Q)[0]).max() < 1e-12, name
        assert abs(np.linalg.det(R) - 1) < 1e-12
    # degenerate covariances still give a proper rotation
    for H in (np.zeros((3, 3)), np.outer([1.0, 2.0, 3.0], [0.5, -1.0, 2.0]), np.diag([3.0, 2.0, 0.0])):
        R = np.zeros((3, 3))
        H = np.ascontiguousarray(H)
        fit(H.ctypes.data, R.ctypes.data)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12


def test_uniform_keypoints_host_matches_golden():
    from tests.conftest import golden
    from tests.golden import cases
    from visual_odometry_amd.geometry import UniformKeyPoints
    g = golden("g1_keypoint_grid")
    for (h, w, n, margin) in cases.GRID_CASES:
        for mname, mask in cases.grid_masks(h, w, seed=h + n + margin).items():
            got = UniformKeyPoints(mask, margin).get_keypoints(None, n)
            assert np.array_equal(got, g[f"{h}x{w}_n{n}_m{margin}_{mname}"])


_REFERENCE = "/root/reference"

_DROP_IN_CHECK = r"""
import contextlib, io, sys, types
sys.dont_write_bytecode = True
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
with_cv2_stub = sys.argv[3] == "stub"
if with_cv2_stub:
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
from visual_odometry_amd import geometry, interfaces, raft_plugins, rectification, synthetic as syn
from flow.flow_interfaces import OpticalFlowInterface
from stereo.stereo_interfaces import StereoDisparityInterface, StereoParamsInterface, StereoRectificationInterface
from camera.camera_interfaces import CameraXformInterface
from stereo.stereo_params_YAML import StereoParamsYAML
assert interfaces.REFERENCE_ABCS["flow"] and interfaces.REFERENCE_ABCS["stereo"] and interfaces.REFERENCE_ABCS["camera"]
assert interfaces.REFERENCE_ABCS["keypoints"] == with_cv2_stub
assert issubclass(raft_plugins.HipOpticalFlowRAFT, OpticalFlowInterface)
assert issubclass(raft_plugins.HipDisparityRAFT, StereoDisparityInterface)
assert issubclass(rectification.HipStereoRectification, StereoRectificationInterface)
assert issubclass(geometry.HipCameraSvdXform, CameraXformInterface)
assert issubclass(geometry.HipCameraRansacXform, CameraXformInterface)
if with_cv2_stub:
    from keypoints.keypoints_interfaces import KeyPointsInterface, Keypoints3DFlowInterface, Keypoints3DInterface
    assert issubclass(geometry.UniformKeyPoints, KeyPointsInterface)
    assert issubclass(geometry.HipKeypoints3DXform, Keypoints3DInterface)
    assert issubclass(geometry.HipKeypoints3DFlow, Keypoints3DFlowInterface)
with contextlib.redirect_stdout(io.StringIO()):
    params = StereoParamsYAML(syn.RIG_YAML)           # a genuine reference parameter object
assert isinstance(params, StereoParamsInterface)
depth = geometry.HipStereoDepth(params)               # the constructor needs no GPU
assert depth.fx == params.focal_length_px and depth.baseline == params.get_baseline()
print("ok")
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(_REFERENCE, "flow")), reason="reference checkout not present")
@pytest.mark.parametrize("cv2", ["stub", "absent"])
def test_backends_subclass_the_reference_interfaces(cv2):
    """With the reference's root on sys.path every Hip* class derives from the reference's OWN
    abstract class; where cv2 is missing only the keypoint interfaces (whose module imports cv2)
    fall back to the local stand-ins, and HipStereoDepth still accepts a reference
    StereoParamsYAML.  Runs in a child process: the choice is made at import time."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _DROP_IN_CHECK, _REFERENCE, ROOT, cv2], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def _call_arguments(text, start):
    """The top-level arguments of the call whose opening parenthesis is at text[start]."""
    depth, args, cur = 0, [], []
    for ch in text[start:]:
        if ch in "([{":
            depth += 1
            if depth == 1:
                continue
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                args.append("".join(cur).strip())
                return args
        if ch == "," and depth == 1:
            args.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    raise AssertionError("unbalanced call")


def test_every_16_byte_buffer_store_has_a_literal_scalar_offset():
    """A 16-byte buffer store whose scalar-offset operand lives in a register may store the NEW contents of its
    data registers when a vector instruction overwrites them right behind it (the compiler inserts the wait states
    only for a constant scalar offset; measured on gfx950: sporadically wrong lanes).  Every such store in the
    library therefore passes an integer literal as its fourth argument -- this test reads the sources and fails
    when one does not."""
    import re
    csrc = os.path.join(ROOT, "visual_odometry_amd", "csrc")
    found = 0
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(csrc, name)).read()
        text = re.sub(r"//[^\n]*", "", text)                      # comments mention the builtin too
        for m in re.finditer(r"__builtin_amdgcn_raw_buffer_store_b128\s*\(", text):
            args = _call_arguments(text, m.end() - 1)
            assert len(args) == 5, (name, args)
            assert re.fullmatch(r"\d+", args[3]), f"{name}: 16-byte buffer store with a non-literal scalar offset: {args[3]!r}"
            found += 1
        # the inline-assembly spelling of the same store must not appear at all
        assert not re.search(r"buffer_store_dwordx4|buffer_store_b128\s+v", text), name
    assert found >= 6                                               # the check did see the stores it is about
