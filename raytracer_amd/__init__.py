"""raytracer_amd -- MI355X-native path-tracing backend behind villarjorge/raytracer's scene-building API.

Layout: csrc/rt_amd.hip + csrc/rt_core.h (HIP kernels and the C ABI of include/rt_amd.h), csrc/host (C++ restatement
of the reference's Hittable / Material / Texture / Camera constructors, BVH build, .obj loader, flatten), and thin
ctypes plumbing in this package.  There is no CPU rendering path: everything that traces a ray needs the HIP
library and a GPU, and says so loudly when either is missing."""
from ._lib import RtError  # noqa: F401
from .builder import HostBuilder, camera_new, write_png  # noqa: F401
from .scene import Scene, debug_arith  # noqa: F401
from . import scenes  # noqa: F401
