"""renderer_amd -- MI355X (gfx950) implementation of the indexed-triangle draw path the
reference renderer hands to its Vulkan driver.  The compute path is the HIP library
``librenderer_amd.so`` behind the C ABI of ``include/renderer_amd.h``; this package is the
Python binding used by the tests and the benchmark."""
from . import api, hostmath, scenes  # noqa: F401
from .runner import SceneRunner  # noqa: F401
