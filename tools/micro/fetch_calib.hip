// fetch_calib.hip -- what do rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ count for THIS path's access shapes?
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE for wide coalesced streaming reads only (128-byte requests tallied at 64 B: double
// it) and says every other width is uncalibrated.  wf2_trace does not stream: each lane gathers one 64-byte record (four
// dwordx4 loads at rec, rec + 16, + 32, + 48) at an address of its own.  VERDICT r04 weak 4: the "0.63 of the HBM peak" of the
// 4 M-triangle scene rests on the factor 2.  This program reads a table that is larger than every cache (1 GiB) with known byte
// counts in four shapes, one kernel name each, so that `rocprofv3 --kernel-trace --pmc <counter>` yields counter / known bytes:
//
//   calib_stream     every lane a float4 of a contiguous range (the guide's case: the control)
//   calib_gather64   every lane ONE random 64-byte-aligned 64-byte record, all 64 bytes (four dwordx4): wf2_trace's record step;
//                    one half of a 128-byte line each, no line twice -- a miss that filled whole lines would show as 2x
//   calib_gather128  every lane one random 128-byte line, both 64-byte halves (eight dwordx4): two sibling records side by side
//   calib_gather16   every lane 16 bytes of a random 64-byte record (one dwordx4): a partial record
//   calib_gather48   every lane a random 48-byte triangle record at a 48-byte stride (three dwordx4; straddles 64-byte lines)
//
// Every record is touched exactly once per launch (index = bijection of the thread id over a power-of-two table), nothing is
// re-read, and the table is 4x the Infinity Cache: what the counters report is what the fabric moved for those bytes.
//
//   hipcc --offload-arch=gfx950 -O2 -o fetch_calib fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- ./fetch_calib          (tools/fetch_calib.sh runs the passes)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                   \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) {                                 \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                       \
        }                                                       \
    } while (0)

__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

// a bijection of [0, 2^bits): odd multiplier, then a rotate of the low `bits` bits (neighbouring threads land far apart)
__device__ __forceinline__ uint32_t scatter(uint32_t i, uint32_t bits)
{
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t x = (i * 2654435761u) & mask;
    x ^= x >> (bits / 2);   // (xorshift of the low bits: invertible)
    return (x * 81007u) & mask;   // (another odd multiplier)
}

__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ t, size_t n4, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    // four float4 per thread, each a fully coalesced wave-wide 1 KiB read
    for (int k = 0; k < 4; k++) {
        const size_t j = i + (size_t)k * gridDim.x * blockDim.x;
        if (j < n4) acc += sum4(t[j]);
    }
    if (acc == 12345.678f) out[0] = acc;
}

// (distinct kernel names for the profiler's per-kernel rows)
__global__ __launch_bounds__(256) void calib_gather64(const char* __restrict__ t, uint32_t bits, float* __restrict__ out)
{
    // ONE 64-byte half of a 128-byte line per lane, and no line touched twice: if a miss filled the whole line, the counters
    // would show twice the known bytes
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = scatter(i, bits);
    const char* p = t + (size_t)r * 128 + ((i * 0x9e3779b9u) >> 31) * 64u;
    const float acc = sum4(*reinterpret_cast<const float4*>(p)) + sum4(*reinterpret_cast<const float4*>(p + 16)) +
                      sum4(*reinterpret_cast<const float4*>(p + 32)) + sum4(*reinterpret_cast<const float4*>(p + 48));
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather128(const char* __restrict__ t, uint32_t bits, float* __restrict__ out)
{
    const uint32_t r = scatter(blockIdx.x * blockDim.x + threadIdx.x, bits);
    const char* p = t + (size_t)r * 128;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) acc += sum4(*reinterpret_cast<const float4*>(p + 16 * k));
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather16(const char* __restrict__ t, uint32_t bits, float* __restrict__ out)
{
    const uint32_t r = scatter(blockIdx.x * blockDim.x + threadIdx.x, bits);
    const float acc = sum4(*reinterpret_cast<const float4*>(t + (size_t)r * 64));
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather48(const char* __restrict__ t, uint32_t bits, float* __restrict__ out)
{
    const uint32_t r = scatter(blockIdx.x * blockDim.x + threadIdx.x, bits);
    const char* p = t + (size_t)r * 48;
    const float acc = sum4(*reinterpret_cast<const float4*>(p)) + sum4(*reinterpret_cast<const float4*>(p + 16)) +
                      sum4(*reinterpret_cast<const float4*>(p + 32));
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const size_t table_bytes = (size_t)1 << 30;
    char* t;
    float* out;
    CK(hipMalloc(&t, table_bytes + 256));
    CK(hipMalloc(&out, 256));
    CK(hipMemset(t, 0, table_bytes + 256));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, double bytes, auto launch) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("%-16s launch %d: known bytes %.0f, %.3f ms, %.2f TB/s\n", name, rep, bytes, ms, bytes / (ms * 1e-3) / 1e12);
        }
    };
    // every kernel reads 256 MiB of distinct bytes out of the 1 GiB table
    const double known = 256.0 * 1048576.0;
    {   // stream: 2^24 float4, four per thread
        const size_t n4 = (size_t)1 << 24;
        const uint32_t threads = (uint32_t)(n4 / 4);
        timed("calib_stream", known, [&] { hipLaunchKernelGGL(calib_stream, dim3(threads / 256), dim3(256), 0, 0, (const float4*)t, n4, out); });
    }
    {   // 2^22 lanes x 64 B: one half of 2^22 of the table's 2^23 lines
        const uint32_t lanes = 1u << 22;
        timed("calib_gather64", known, [&] { hipLaunchKernelGGL(calib_gather64, dim3(lanes / 256), dim3(256), 0, 0, t, 23u, out); });
    }
    {   // 2^21 lanes x 128 B out of 2^23 lines
        const uint32_t lanes = 1u << 21;
        timed("calib_gather128", known, [&] { hipLaunchKernelGGL(calib_gather128, dim3(lanes / 256), dim3(256), 0, 0, t, 23u, out); });
    }
    {   // 2^24 lanes x 16 B, one per 64-byte record of the whole table
        const uint32_t lanes = 1u << 24;
        timed("calib_gather16", known, [&] { hipLaunchKernelGGL(calib_gather16, dim3(lanes / 256), dim3(256), 0, 0, t, 24u, out); });
    }
    {   // 2^22 lanes x 48 B at a 48-byte stride (2^22 * 48 = 192 MiB of known bytes)
        const uint32_t lanes = 1u << 22;
        timed("calib_gather48", 48.0 * lanes, [&] { hipLaunchKernelGGL(calib_gather48, dim3(lanes / 256), dim3(256), 0, 0, t, 24u, out); });
    }
    std::printf("%s done\n", prop.gcnArchName);
    return 0;
}
