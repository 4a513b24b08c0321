"""vello_amd -- MI355X-native replacement for vello's GPU compute path.

Only what the hot path needs lives here: the HIP kernels + C ABI (csrc/engine, include/vello_hip.h),
the C++ mirror of the host-side contract (csrc/host) and thin ctypes bindings that keep the
reference's names (Scene, Renderer, RenderParams, AaConfig; vello/src/lib.rs, vello/src/scene.rs).

There is no CPU fallback: importing works without a GPU (so CPU tests can check the ABI), but
`Renderer(...)` raises when no gfx950 device is present.
"""
from ._lib import load_library, library_path, VelloHipError
from .kurbo import Affine, BezPath, Circle, Rect, RoundedRect, Line, Stroke, Join, Cap
from .scene import (Scene, Fill, Color, BlendMode, Mix, Compose, Gradient, Extend, InterpolationAlphaSpace, ImageData, ImageBrush,
                    ImageFormat, ImageAlphaType, ImageQuality, Resolver, Resolved)
from .renderer import Renderer, RenderParams, AaConfig, RendererOptions, Engine, Layout

__all__ = [
    "load_library", "library_path", "VelloHipError", "Affine", "BezPath", "Circle", "Rect", "RoundedRect", "Line",
    "Stroke", "Join", "Cap", "Scene", "Fill", "Color", "BlendMode", "Mix", "Compose", "Renderer", "RenderParams",
    "AaConfig", "RendererOptions", "Engine", "Layout", "Gradient", "Extend", "InterpolationAlphaSpace", "ImageData", "ImageBrush",
    "ImageFormat", "ImageAlphaType", "ImageQuality", "Resolver", "Resolved",
]
