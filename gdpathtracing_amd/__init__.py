"""gdpathtracing_amd -- MI355X-native (gfx950 HIP) back end for the GDPathTracing hot path.

Only what the path needs lives here: `csrc/` (HIP kernels, host builder, the C ABI of include/jpt.h),
`capi` (ctypes loader, fails loudly when the HIP library is missing), `host` (host-side mirror of the
reference's GeometryGroup3D / PathTracingCamera / ProgressiveRendering interface), `scenes` and `wire`
(synthetic inputs and wire formats).  The CPU oracle is NOT part of this package (see oracle/).
"""
__version__ = "0.1.0"

import os as _os

# Hardware queues.  The HIP runtime reads GPU_MAX_HW_QUEUES when it STARTS -- at the first HIP call of the process, not when
# it is loaded (measured: tools/hwq_probe.py) -- and the library's queued renders want more than its default pool of four
# (csrc/jpt_capi.hip, HwQueueRequest: the library asks for 16 itself when it is loaded, which is in time for a C++ host).
# A Python process usually imports torch first; importing this package before the first torch.cuda call still gets the
# request in.  A value the host has set is kept.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
