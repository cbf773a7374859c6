// Does a VALU instruction whose EXEC mask has whole 16-lane quarters empty issue faster on gfx950?
// Each wave runs a chain of dependent f32 FMAs with only `active` lanes enabled (contiguous from lane 0, or strided).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void chain(float* out, int active, int stride_mode, int iters)
{
    const int lane = threadIdx.x & 63;
    const bool on = stride_mode ? ((lane % (64 / active)) == 0) : (lane < active);
    float a = (float)threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    if (on) {
        for (int i = 0; i < iters; i++) {
            a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f;
            a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main()
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++)
        for (int active : {64, 48, 32, 16, 8, 1}) {
            if (mode && 64 % active) continue;
            hipLaunchKernelGGL(chain, dim3(256 * 8), dim3(256), 0, 0, out, active, mode, 1000);
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(256 * 8), dim3(256), 0, 0, out, active, mode, 20000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s active=%2d: %.3f ms\n", mode ? "strided   " : "contiguous", active, ms);
        }
    return 0;
}
