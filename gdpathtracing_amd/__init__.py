"""gdpathtracing_amd -- MI355X-native (gfx950 HIP) back end for the GDPathTracing hot path.

Only what the path needs lives here: `csrc/` (HIP kernels, host builder, the C ABI of include/jpt.h),
`capi` (ctypes loader, fails loudly when the HIP library is missing), `host` (host-side mirror of the
reference's GeometryGroup3D / PathTracingCamera / ProgressiveRendering interface), `scenes` and `wire`
(synthetic inputs and wire formats).  The CPU oracle is NOT part of this package (see oracle/).
"""
__version__ = "0.1.0"
