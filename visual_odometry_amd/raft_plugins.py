"""HIP-backed drop-ins for the reference's RAFT wrappers.

  HipOpticalFlowRAFT   flow/flow_map_RAFT.py:20-106        (OpticalFlowInterface)
  HipDisparityRAFT     stereo/stereo_disparity_RAFT.py:18-110 (StereoDisparityInterface)

Same constructor arguments, same single-frame methods (host images in, freshly
allocated host array out, ``ValueError`` on ``None`` images, ``TypeError`` on
unsupported image types), plus ``*_batch`` methods that keep a chunk of frames on
the GPU.  ``checkpoint=None`` builds the network with seeded random weights (no
trained checkpoints are available offline).

The reference's "warm start" replaces the carried flow by zeros before every call
(flow_map_RAFT.py:75-78, stereo_disparity_RAFT.py:85-88), i.e. every call starts
from zero flow; nothing is carried between calls here either.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .interfaces import OpticalFlowInterface, StereoDisparityInterface
from .raft.models import RAFTFlow, RAFTFlowConfig, RAFTStereo, RAFTStereoConfig, _checked, load_checkpoint


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP backend needs a visible AMD GPU (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _unpad(x: torch.Tensor, pads) -> torch.Tensor:
    ht, wd = x.shape[-2:]
    return x[..., pads[2]:ht - pads[3], pads[0]:wd - pads[1]]


def load_image(image_param, device, allow_path: bool = True) -> torch.Tensor:
    """Host image -> (1,3,H,W) float32 0..255 on the GPU.  A 2-D gray array is
    replicated into three equal channels; a 3-channel array is taken as BGR and
    flipped to RGB; a string is a file path opened with PIL."""
    if isinstance(image_param, str) and allow_path:
        from PIL import Image
        img = Image.open(image_param)
        if img.mode in ("L", "LA"):
            img = img.convert("RGB")
        arr = np.array(img).astype(np.uint8)
    elif isinstance(image_param, np.ndarray):
        arr = image_param
        if arr.ndim == 2:
            arr = np.repeat(arr[:, :, None], 3, axis=2)
        elif arr.ndim == 3 and arr.shape[2] == 3:
            arr = arr[:, :, ::-1]
        elif arr.ndim == 3 and arr.shape[2] == 1:
            arr = np.repeat(arr, 3, axis=2)
        else:
            raise TypeError("Unsupported image format")
        arr = np.ascontiguousarray(arr).astype(np.uint8)
    else:
        raise TypeError("Unsupported image format")
    return torch.from_numpy(arr).to(device).permute(2, 0, 1).float()[None].contiguous()


def gray_batch_to_rgb(frames: torch.Tensor) -> torch.Tensor:
    """(B,H,W) uint8/float GPU frames -> (B,3,H,W) float32 0..255."""
    return frames.float()[:, None].expand(-1, 3, -1, -1).contiguous()


class HipOpticalFlowRAFT(OpticalFlowInterface):
    def __init__(self, checkpoint: Optional[str] = None, rectification=None, iters: int = 32, seed: int = 0,
                 mixed_precision: bool = False, conv_arithmetic: str = "f16x3"):
        self.device = _device()
        self.iters = iters
        self.flow = None
        self.rectification = rectification
        torch.manual_seed(seed)
        self.model = RAFTFlow(RAFTFlowConfig())
        if checkpoint is not None:
            load_checkpoint(self.model, checkpoint, self.device)
        self.model.to(self.device).eval()
        self.model.mixed_precision = bool(mixed_precision)      # flow/flow_map_RAFT.py:38 passes False
        self.model.conv_arithmetic = _checked(conv_arithmetic)   # how the heavy float32 products are formed (raft.layers.Arithmetic)

    def compute_flow(self, img1, img2) -> np.ndarray:
        if img1 is None or img2 is None:
            raise ValueError("One or both images were not found or invalid.")
        if self.rectification:
            img1, _ = self.rectification.rectify_images(img1, None)
            img2, _ = self.rectification.rectify_images(img2, None)
        image1, image2 = load_image(img1, self.device), load_image(img2, self.device)
        flow_up = self.compute_flow_batch(image1, image2)
        self.flow = flow_up
        return flow_up[0].cpu().numpy()

    def compute_flow_batch(self, image1: torch.Tensor, image2: torch.Tensor, fmap1=None, fmap2=None) -> torch.Tensor:
        """(B,3,H,W) float 0..255 GPU images -> flow (B,2,H,W) float32 on the GPU."""
        pads = ops.pad_amounts(image1.shape[-2], image1.shape[-1], 8)
        if any(pads):
            image1 = F.pad(image1, pads, mode="replicate")
            image2 = F.pad(image2, pads, mode="replicate")
        _, flow_up = self.model(image1, image2, iters=self.iters, fmap1=fmap1, fmap2=fmap2)
        return _unpad(flow_up, pads).contiguous() if any(pads) else flow_up


    def compute_flow_sequence(self, gray: torch.Tensor) -> torch.Tensor:
        """(F,H,W) uint8 GPU frames -> (F-1,2,H,W) flow between consecutive frames.
        The input stage (padding, normalisation) is one HIP kernel, the three equal channels of a gray
        frame stay one plane (the stems sum their filters over the input channels instead), and the
        feature encoder runs once per frame, not twice per pair."""
        imgs, pads = ops.prepare_images(gray, 8, normalize=True, channels=1)
        fm = self.model.features(imgs)
        _, flow_up = self.model(imgs[:-1], imgs[1:], iters=self.iters, fmap1=fm[:-1], fmap2=fm[1:], normalized=True)
        return _unpad(flow_up, pads).contiguous() if any(pads) else flow_up


class HipDisparityRAFT(StereoDisparityInterface):
    def __init__(self, checkpoint: Optional[str] = None, rectification=None, iters: int = 32,
                 config: Optional[RAFTStereoConfig] = None, seed: int = 1, mixed_precision: bool = False,
                 conv_arithmetic: str = "f16x3"):
        self.device = _device()
        self.iters = iters
        self.flow = None
        self.rectification = rectification
        torch.manual_seed(seed)
        self.model = RAFTStereo(config or RAFTStereoConfig())
        if checkpoint is not None:
            load_checkpoint(self.model, checkpoint, self.device)
        self.model.to(self.device).eval()
        self.model.mixed_precision = bool(mixed_precision)      # stereo/stereo_disparity_RAFT.py:43 passes False
        self.model.conv_arithmetic = _checked(conv_arithmetic)

    def compute_disparity(self, img_left, img_right) -> np.ndarray:
        if img_left is None or img_right is None:
            raise ValueError("One or both images were not found or invalid.")
        if isinstance(img_left, str) or isinstance(img_right, str):
            from PIL import Image            # the reference reads paths as grayscale
            if isinstance(img_left, str):
                img_left = np.array(Image.open(img_left).convert("L"))
            if isinstance(img_right, str):
                img_right = np.array(Image.open(img_right).convert("L"))
        if self.rectification:
            img_left, img_right = self.rectification.rectify_images(img_left, img_right)
        image1 = load_image(img_left, self.device, allow_path=False)
        image2 = load_image(img_right, self.device, allow_path=False)
        disp = self.compute_disparity_batch(image1, image2)
        self.flow = disp
        return disp[0].cpu().numpy()

    def compute_disparity_batch(self, image1: torch.Tensor, image2: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) float 0..255 GPU images -> disparity (B,H,W) float32 (negative)."""
        pads = ops.pad_amounts(image1.shape[-2], image1.shape[-1], 32)
        if any(pads):
            image1 = F.pad(image1, pads, mode="replicate")
            image2 = F.pad(image2, pads, mode="replicate")
        _, disp = self.model(image1, image2, iters=self.iters)
        disp = _unpad(disp, pads) if any(pads) else disp
        return disp[:, 0].contiguous()

    def compute_disparity_gray_batch(self, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
        """(F,H,W) uint8 GPU frames -> disparity (F,H,W) float32; input stage on the GPU."""
        image1, pads = ops.prepare_images(left, 32, normalize=True, channels=1)
        image2, _ = ops.prepare_images(right, 32, normalize=True, channels=1)
        _, disp = self.model(image1, image2, iters=self.iters, normalized=True)
        disp = _unpad(disp, pads) if any(pads) else disp
        return disp[:, 0].contiguous()
