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


def _try_reference():
    try:
        from flow.flow_interfaces import OpticalFlowInterface as F                      # type: ignore
        from stereo.stereo_interfaces import (CameraParametersInterface as C,           # type: ignore
                                              StereoDisparityInterface as D,
                                              StereoParamsInterface as P,
                                              StereoRectificationInterface as R)
        from keypoints.keypoints_interfaces import (KeyPointsInterface as K,            # type: ignore
                                                    Keypoints3DInterface as K3,
                                                    Keypoints3DFlowInterface as K3F)
        from camera.camera_interfaces import CameraXformInterface as X                   # type: ignore
        return dict(F=F, C=C, D=D, P=P, R=R, K=K, K3=K3, K3F=K3F, X=X)
    except Exception:       # reference tree absent, or one of its imports (cv2) is
        return None


_ref = _try_reference()
USING_REFERENCE_ABCS = _ref is not None

if _ref is not None:
    OpticalFlowInterface = _ref["F"]
    CameraParametersInterface = _ref["C"]
    StereoDisparityInterface = _ref["D"]
    StereoParamsInterface = _ref["P"]
    StereoRectificationInterface = _ref["R"]
    KeyPointsInterface = _ref["K"]
    Keypoints3DInterface = _ref["K3"]
    Keypoints3DFlowInterface = _ref["K3F"]
    CameraXformInterface = _ref["X"]
else:
    class OpticalFlowInterface(ABC):
        @abstractmethod
        def compute_flow(self, img1, img2):
            """Two images -> flow (2,H,W) float32, channel 0 = du, channel 1 = dv."""

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

    class CameraXformInterface(ABC):
        @abstractmethod
        def compute_camera_xform(self, P, Q): ...
