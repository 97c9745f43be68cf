"""digital_earth_amd — MI355X-native (gfx950) implementation of Digital Earth's spectral path-tracing hot path.

`Renderer` and `EarthViewer` keep the reference's API (renderer.py / earth_viewer.py); the device work is in
libdigitalearth_hip.so (csrc/, C ABI in include/digital_earth.h).  Importing the package does not load the native
library; constructing a Renderer does, and fails loudly when it is missing or no gfx950 device is present.
"""
__all__ = ["Renderer", "EarthViewer", "Camera"]


def __getattr__(name):
    if name == "Renderer":
        from .renderer import Renderer
        return Renderer
    if name in ("EarthViewer", "Camera"):
        from . import earth_viewer
        return getattr(earth_viewer, name)
    raise AttributeError(name)
