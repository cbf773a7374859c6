// node_fetch.hip -- how fast can a CU of gfx950 serve DIVERGENT 128-byte record fetches (every lane its own record, the
// next record's index read from the one just fetched: the access pattern of a BVH walk)?
//
// A table of `n` 128-byte records (7 x float4 of payload + 4 "child" indices, like WideNode4); every lane walks its own
// pseudo-random chain through it: fetch the record as 8 x 16-byte loads, do `valu` dependent FMAs per step on the
// payload (stand-in for the slab tests), continue at child[k].  Reported per configuration: record fetches per
// microsecond per CU, bytes per cycle per CU, and the time one dependent step takes a wave.  Sweeps: table size
// (6 MB = L2-resident like the C3 scene, 96 MB = Infinity Cache, 768 MB = HBM), waves per SIMD, enabled lanes per wave,
// VALU work per step.
//
//   hipcc --offload-arch=gfx950 -O2 -o node_fetch node_fetch.hip && ./node_fetch
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(128) Rec {
    float4 q[7];
    int32_t child[4];
};

__global__ __launch_bounds__(256) void walk(const Rec* __restrict__ table, uint32_t n, int steps, int lanes_on, int valu, float* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (lane >= lanes_on) return;
    uint32_t cur = (tid * 2654435761u) % n;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        const Rec* r = table + cur;
        const float4 a = r->q[0], b = r->q[1], c = r->q[2], d = r->q[3], e = r->q[4], f = r->q[5], g = r->q[6];
        const int4 ch = *reinterpret_cast<const int4*>(r->child);
        // every component is used, so the compiler keeps the eight 16-byte loads (27 adds: the floor of VALU work per step)
        float v = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)) +
                  ((e.x + e.y) + (e.z + e.w)) + ((f.x + f.y) + (f.z + f.w)) + ((g.x + g.y) + (g.z + g.w));
        for (int k = 0; k < valu; k++) v = v * 1.0001f + 0.5f;   // dependent chain: `valu` VALU instructions
        acc += v;
        const int k = (int)(s & 3);
        cur = (uint32_t)(k == 0 ? ch.x : k == 1 ? ch.y : k == 2 ? ch.z : ch.w);
    }
    out[tid] = acc;
}

#define CK(x)                                                               \
    do {                                                                    \
        hipError_t e_ = (x);                                                \
        if (e_ != hipSuccess) {                                             \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));             \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::printf("%s, %d CUs\n", prop.gcnArchName, n_cu);
    float* out;
    CK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * sizeof(float)));
    for (uint32_t n : {49152u, 786432u, 6291456u}) {   // 6 MB, 96 MB, 768 MB
        std::vector<Rec> h(n);
        uint32_t s = 12345u;
        for (uint32_t i = 0; i < n; i++) {
            for (int k = 0; k < 7; k++) h[i].q[k] = make_float4(1e-3f * (float)(i & 255), 0.5f, 0.25f, 0.125f);
            for (int k = 0; k < 4; k++) {
                s = s * 1664525u + 1013904223u;
                h[i].child[k] = (int32_t)((s >> 4) % n);
            }
        }
        Rec* d;
        CK(hipMalloc(&d, (size_t)n * sizeof(Rec)));
        CK(hipMemcpy(d, h.data(), (size_t)n * sizeof(Rec), hipMemcpyHostToDevice));
        for (int waves : {1, 2, 4, 7, 8})
            for (int lanes_on : {64, 32, 16})
                for (int valu : {0, 64, 192}) {
                    if ((lanes_on != 64 || valu != 64) && waves != 7) continue;   // full sweep only at 7 waves/SIMD
                    const int steps = 400;
                    const int blocks = n_cu * waves;
                    hipLaunchKernelGGL(walk, dim3(blocks), dim3(256), 0, 0, d, n, 20, lanes_on, valu, out);
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0));
                    CK(hipEventCreate(&e1));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(walk, dim3(blocks), dim3(256), 0, 0, d, n, steps, lanes_on, valu, out);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    const double fetches = (double)blocks * 4.0 * lanes_on * steps;
                    const double per_us_cu = fetches / (ms * 1e3) / n_cu;
                    std::printf("table %4.0f MB  waves/SIMD %d  lanes %2d  valu/step %3d: %7.1f record fetches/us/CU = %5.1f B/cycle/CU at 2.4 GHz, "
                                "%6.1f Grec/s chip, step time per wave %.0f ns\n",
                                (double)n * 128.0 / 1048576.0, waves, lanes_on, valu, per_us_cu, per_us_cu * 128.0 / 2400.0, fetches / (ms * 1e6),
                                ms * 1e6 / steps);
                    CK(hipEventDestroy(e0));
                    CK(hipEventDestroy(e1));
                }
        CK(hipFree(d));
    }
    return 0;
}
