// jpt_kernels_post.hip -- multi-GPU epilogue: the rank-major pieces an RCCL gather delivers are
// scattered back to image rows, and the display image (ACES of the mean, progressive_rendering.glsl:39-45)
// is re-derived from the assembled sums.  No reference counterpart (the reference is single-device).
#include "jpt_kernels.h"

namespace jpt {

__global__ __launch_bounds__(256) void assemble_kernel(const float4* __restrict__ gathered, int world, int width, int height,
                                                       int max_local_rows, float4* __restrict__ accum_full,
                                                       uint32_t* __restrict__ ldr_full, uint32_t frame_count)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= width || y >= height) return;
    const int strip = y / kStripRows;
    const int rank = strip % world;
    const int ly = (strip / world) * kStripRows + (y - strip * kStripRows);
    const float4 v = gathered[((size_t)rank * max_local_rows + ly) * width + x];
    const size_t idx = (size_t)y * width + x;
    accum_full[idx] = v;
    const float fc = (float)(frame_count ? frame_count : 1u);
    const f3 col = aces_film(mk3(v.x / fc, v.y / fc, v.z / fc) * 1.0f);
    ldr_full[idx] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
}

void launch_assemble(hipStream_t stream, const float4* gathered, int world, int width, int height, int max_local_rows,
                     float4* accum_full, uint32_t* ldr_full, uint32_t frame_count)
{
    dim3 grid((width + 255) / 256, height), block(256);
    hipLaunchKernelGGL(assemble_kernel, grid, block, 0, stream, gathered, world, width, height, max_local_rows, accum_full,
                       ldr_full, frame_count);
}

}  // namespace jpt
