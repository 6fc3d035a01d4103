"""Feature extractors with the interface of pyvisim/features/_features.py: `extractor(image) -> (n, output_dim)`
float descriptors plus `.output_dim`.

  SIFT / RootSIFT   OpenCV's detector is third-party CPU code and out of scope; it
                    is imported lazily, so the classes exist without cv2 and fail only when called.  RootSIFT
                    additionally exposes `raw(image)` so that the encoders can hand the *raw* uint8 SIFT rows
                    to the GPU and fuse the RootSIFT tail (d /= sum+1e-7; sqrt, _features.py:112-114) there.
  Lambda            any user function (the descriptor-level door used by tests and synthetic benchmarks).
  DeepConvFeature   conv feature maps of a torch model on PyTorch-ROCm (torch is plumbing here).  torchvision
                    is absent offline, so the default network is an own VGG16 `features` stack with RANDOM
                    weights unless a model is passed (the reference's default argument downloads weights at
                    import, _features.py:179 -- never attempted).
"""
from __future__ import annotations

from functools import wraps
from typing import Callable

import numpy as np

from .._base_classes import FeatureExtractorBase


def _check_output_shape(func) -> Callable:
    """Extractor outputs must be a 2-D ndarray (n, output_dim); None becomes an empty (0, D) array
    (same contract as the reference wrapper, _features.py:24-51)."""

    @wraps(func)
    def wrapper(self, *args, **kwargs) -> np.ndarray:
        image = args[0]
        if type(image).__module__.startswith("torch"):
            raise TypeError("Torch images are not supported yet. Please convert to NumPy.")
        feats = func(self, *args, **kwargs)
        if feats is None:
            return np.zeros((0, self.output_dim), dtype=np.float32)
        if not isinstance(feats, np.ndarray):
            raise ValueError(f"Expected output to be a NumPy array, got {type(feats)} instead.")
        if feats.ndim != 2:
            raise ValueError(f"Feature extractor output must be 2D. Got shape {feats.shape}.")
        if feats.shape[1] != self.output_dim:
            raise ValueError(f"Expected feat_vecs.shape[1] == {self.output_dim}, but got {feats.shape[1]}.")
        return feats

    return wrapper


def _cv2():
    try:
        import cv2
    except ImportError as e:  # pragma: no cover - cv2 is absent in the build image
        raise ImportError("OpenCV (cv2) is required for SIFT keypoint detection; pass descriptors through "
                          "`Lambda` or `encode_descriptors` instead") from e
    return cv2


class SIFT(FeatureExtractorBase):
    """Lowe's SIFT descriptors (n, 128), integer valued float32, via OpenCV."""

    def __init__(self):
        super().__init__()
        self._output_dim = 128

    @property
    def output_dim(self) -> int:
        return self._output_dim

    @_check_output_shape
    def __call__(self, image: np.ndarray, /) -> np.ndarray:
        super().__call__(image)
        _, descriptors = _cv2().SIFT.create().detectAndCompute(image, None)
        return descriptors

    def __repr__(self):
        return f"SIFT(output_dim={self.output_dim})"


class RootSIFT(FeatureExtractorBase):
    """SIFT + Hellinger normalisation (Arandjelovic & Zisserman 2012)."""
    fused_rootsift = True   # encoders may call raw() and let the GPU apply the RootSIFT tail

    def __init__(self):
        super().__init__()
        self._output_dim = 128

    @property
    def output_dim(self) -> int:
        return self._output_dim

    def raw(self, image: np.ndarray) -> np.ndarray:
        """Raw OpenCV SIFT rows (n, 128) float32 with integer values 0..255 (empty -> (0, 128))."""
        FeatureExtractorBase.__call__(self, image)
        _, descriptors = _cv2().SIFT.create().detectAndCompute(image, None)
        return np.zeros((0, 128), np.float32) if descriptors is None else descriptors

    @_check_output_shape
    def __call__(self, image: np.ndarray, /) -> np.ndarray:
        super().__call__(image)
        _, descriptors = _cv2().SIFT.create().detectAndCompute(image, None)
        if descriptors is not None:
            descriptors /= (descriptors.sum(axis=1, keepdims=True) + 1e-7)
            descriptors = np.sqrt(descriptors)
        return descriptors

    def __repr__(self):
        return f"RootSIFT(output_dim={self.output_dim})"


class Lambda(FeatureExtractorBase):
    """Wraps any `func(image) -> (n, output_dim)`."""

    def __init__(self, func: Callable, output_dim: int):
        super().__init__()
        if not callable(func):
            raise ValueError(f"Argument func must be a callable object, got {type(func)} instead")
        self._output_dim = output_dim
        self.func = func

    @property
    def output_dim(self) -> int:
        return self._output_dim

    @_check_output_shape
    def __call__(self, image: np.ndarray, /) -> np.ndarray:
        super().__call__(image)
        return self.func(image)


def _vgg16_features():
    """VGG16 `features` stack (13 conv + 5 max-pool), same module indices as torchvision's (conv at
    0,2,5,7,10,12,14,17,19,21,24,26,28) so that layer_index=-1 hooks `features.28`."""
    import torch.nn as nn
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    layers, c_in = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c_in, v, 3, padding=1), nn.ReLU(inplace=True)]
            c_in = v

    class VGG16Features(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(*layers)

        def forward(self, x):
            return self.features(x)

    return VGG16Features()


class DeepConvFeature(FeatureExtractorBase):
    """Feature map of one conv layer, flattened to (H*W, C [+2]) descriptors (reference: _features.py:151-306).

    The hook sits on the Conv2d itself, i.e. PRE-ReLU (:254-261); the default transform is ToTensor +
    Resize(224, 224) with NO mean/std normalisation (:192-194); `spatial_encoding` appends (x/W, y/H).
    `batch(images)` runs many images in one forward and keeps the features on the device."""

    def __init__(self, model=None, target_submodule: str = None, layer_index: int = -1,
                 spatial_encoding: bool = True, device=None, transform=None):
        import torch
        super().__init__()
        if model is None:
            model = _vgg16_features()
        if not isinstance(model, torch.nn.Module):
            raise TypeError(f"Currently, only torch.nn.Module is supported. Got {type(model)} instead.")
        self._model = model
        self.layer_index = layer_index
        self.spatial_encoding = spatial_encoding
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.transform = transform
        if target_submodule is not None and not hasattr(model, target_submodule):
            raise AttributeError(f"Model {model._get_name()} has no submodule named {target_submodule}.")
        self._modules = model if target_submodule is None else getattr(model, target_submodule)
        self._conv_layers = self.list_conv_layers()
        if not self._conv_layers:
            raise ValueError(f"No convolutional layers found in model {model._get_name()}.")
        try:
            _, self.selected_layer_name, self.selected_layer_module = self._conv_layers[layer_index]
        except IndexError:
            raise IndexError(f"Model {model._get_name()} has only {len(self._conv_layers)} convolutional layers. "
                             f"Got layer_index={layer_index}.")
        c = self.selected_layer_module.out_channels
        self._output_dim = c + 2 if spatial_encoding else c
        self.buffer = None
        self.hook = self.selected_layer_module.register_forward_hook(self._hook_fn)
        self._model.eval().to(self.device)

    def _hook_fn(self, module, inputs, output):
        self.buffer = output.detach()

    @property
    def output_dim(self) -> int:
        return self._output_dim

    @property
    def model(self):
        return self._model

    def list_conv_layers(self):
        import torch
        out, idx = [], 0
        for name, module in self._modules.named_modules():
            if isinstance(module, torch.nn.Conv2d):
                out.append((idx, name, module))
                idx += 1
        return out

    def _to_tensor(self, image: np.ndarray):
        import torch
        import torch.nn.functional as F
        if self.transform is not None:
            return self.transform(image)
        t = torch.from_numpy(np.ascontiguousarray(image))
        if t.ndim == 2:
            t = t[:, :, None]
        t = t.permute(2, 0, 1)
        t = t.float().div(255) if t.dtype == torch.uint8 else t.float()
        return F.interpolate(t[None], size=(224, 224), mode="bilinear", align_corners=False, antialias=True)[0]

    def batch(self, images):
        """(B, Hf*Wf, D) float32 torch tensor on self.device for a list of images."""
        import torch
        x = torch.stack([self._to_tensor(im) for im in images]).to(self.device)
        with torch.no_grad():
            self._model(x)
        if self.buffer is None:
            raise RuntimeError("Forward hook did not capture any features.")
        fm = self.buffer                                  # (B, C, Hf, Wf)
        b, c, hf, wf = fm.shape
        feats = fm.reshape(b, c, hf * wf).transpose(1, 2)  # (B, Hf*Wf, C), row-major over (y, x)
        if self.spatial_encoding:
            ys, xs = torch.meshgrid(torch.arange(hf, device=fm.device), torch.arange(wf, device=fm.device),
                                    indexing="ij")
            coords = torch.stack([xs.reshape(-1) / wf, ys.reshape(-1) / hf], dim=1).float()
            feats = torch.cat([feats, coords[None].expand(b, -1, -1)], dim=2)
        return feats.contiguous()

    @_check_output_shape
    def __call__(self, image: np.ndarray, /) -> np.ndarray:
        super().__call__(image)
        return self.batch([image])[0].cpu().numpy()

    def __repr__(self):
        return (f"DeepConvFeature(model={self._model._get_name()}, layer_index={self.layer_index}, "
                f"spatial_encoding={self.spatial_encoding}, device={self.device}, "
                f"selected_layer_name={self.selected_layer_name}, output_dim={self.output_dim})")
