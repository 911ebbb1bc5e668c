"""The plugin interfaces the HIP backends implement.

Inside the reference's source tree (its root on ``sys.path``) the backends
subclass the reference's OWN abstract classes, so they are drop-ins wherever the
drivers accept ``OpticalFlowInterface`` / ``StereoDisparityInterface`` /
``KeyPointsInterface`` / ``Keypoints3DInterface`` / ``Keypoints3DFlowInterface``
/ ``CameraXformInterface`` objects (flow/flow_interfaces.py:4,
stereo/stereo_interfaces.py:5-98, keypoints/keypoints_interfaces.py:6-98,
camera/camera_interfaces.py:5).  Stand-alone (this repository on its own, e.g. on
the GPU test box) equivalent ABCs with the same names, method names and argument
order are defined here instead.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from enum import Enum


def _try_import(module: str, names):
    """The named classes of one reference module, or None when the reference tree is not on
    ``sys.path`` or that module cannot be imported (keypoints/keypoints_interfaces.py:2 imports
    cv2, which may be absent).  Each module is tried on its own: one failing import must not
    turn the other interfaces into local stand-ins, or a genuine reference object (say a
    ``StereoParamsYAML``) would fail the ``isinstance`` checks of the classes built on them."""
    try:
        mod = __import__(module, fromlist=list(names))
        return [getattr(mod, n) for n in names]
    except Exception:
        return None


_flow = _try_import("flow.flow_interfaces", ["OpticalFlowInterface"])
_stereo = _try_import("stereo.stereo_interfaces", ["CameraParametersInterface", "StereoParamsInterface",
                                                   "StereoRectificationInterface", "StereoDisparityInterface"])
_keypoints = _try_import("keypoints.keypoints_interfaces", ["KeyPointsInterface", "Keypoints3DInterface",
                                                            "Keypoints3DFlowInterface"])
_camera = _try_import("camera.camera_interfaces", ["CameraXformInterface"])
# which interface families come from the reference's own modules
REFERENCE_ABCS = {"flow": _flow is not None, "stereo": _stereo is not None, "keypoints": _keypoints is not None,
                  "camera": _camera is not None}
USING_REFERENCE_ABCS = all(REFERENCE_ABCS.values())

if _flow is not None:
    (OpticalFlowInterface,) = _flow
else:
    class OpticalFlowInterface(ABC):
        @abstractmethod
        def compute_flow(self, img1, img2):
            """Two images -> flow (2,H,W) float32, channel 0 = du, channel 1 = dv."""

if _stereo is not None:
    CameraParametersInterface, StereoParamsInterface, StereoRectificationInterface, StereoDisparityInterface = _stereo
else:
    class CameraParametersInterface(ABC):
        @abstractmethod
        def get_intrinsics(self): ...

        @abstractmethod
        def get_distortion_coeffs(self): ...

        @abstractmethod
        def get_resolution(self): ...

    class StereoParamsInterface(ABC):
        class StereoCamera(Enum):
            LEFT = "left"
            RIGHT = "right"

        @abstractmethod
        def load_params(self, input_data): ...

        @abstractmethod
        def get_intrinsics(self): ...

        @abstractmethod
        def get_baseline(self): ...

        @abstractmethod
        def get_z_max(self, d_min=1.0): ...

    class StereoRectificationInterface(ABC):
        @abstractmethod
        def rectify_images(self, img_left, img_right): ...

        @abstractmethod
        def get_rectification_matrices(self): ...

        @abstractmethod
        def get_rectification_masks(self): ...

    class StereoDisparityInterface(ABC):
        @abstractmethod
        def compute_disparity(self, img_left, img_right):
            """Rectified pair -> disparity (H,W) float32 (negative: left->right x-flow)."""

if _keypoints is not None:
    KeyPointsInterface, Keypoints3DInterface, Keypoints3DFlowInterface = _keypoints
else:
    class KeyPointsInterface(ABC):
        @abstractmethod
        def get_keypoints(self, image, max_number: int): ...

    class Keypoints3DInterface(ABC):
        @abstractmethod
        def to_3d(self, keypoints, depth_map): ...

        @abstractmethod
        def to_2d(self, points_3D): ...

    class Keypoints3DFlowInterface(ABC):
        @abstractmethod
        def compute_2d_flow(self, keypoints, uv_flow): ...

        @abstractmethod
        def compute_3d_flow(self, keypoints, depth1, depth2, uv_flow): ...

if _camera is not None:
    (CameraXformInterface,) = _camera
else:
    class CameraXformInterface(ABC):
        @abstractmethod
        def compute_camera_xform(self, P, Q): ...
