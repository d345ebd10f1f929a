"""raytracing_amd — MI355X-native trace / intersect / shade loop of the
Ivoah/RayTracing renderer, behind the C ABI of include/hiptrace.h.

The package holds the HIP kernels and the C ABI (csrc/), and the host-side
mirror of the reference's scene model and renderFrame for the Python harness.
"""
from . import _abi  # noqa: F401
from .scene import (Camera, Checker, Diffuse, Emission, Glass, Glossy, Image, Perlin, SolidColor,  # noqa: F401
                    Sphere, STL, Triangle, XYRect, XZRect, YZRect, flatten, load_scene)
