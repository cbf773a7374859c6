// node_fetch.hip -- what does the vector-memory path of a gfx950 CU charge for DIVERGENT record fetches (every lane its
// own record, the next record's index read from the one just fetched: the access pattern of a BVH walk)?
//
// A table of 128-byte-aligned records; every lane walks its own pseudo-random chain through it.  Per step a lane loads
// the first NLOADS x 16 bytes of its record (NLOADS = 1, 2, 4, 7, 8: how much of the cost is per load instruction and how
// much per byte), with 64-bit per-lane addresses or with a scalar base + 32-bit per-lane offset, does `valu` dependent
// FMAs on the payload, and continues at the index stored in the record.  Reported per configuration: record fetches
// per microsecond per CU, the CU's time per wave-level load instruction, and the time one dependent step takes a wave.
// Sweeps: table size (6 MB = L2-resident like the C3 scene, 96 MB = Infinity Cache), waves per SIMD, enabled lanes.
//
// Findings on MI355X (profiles/r02/node_fetch.txt): the rate is set per CU by the vector-memory path, not by L2 or HBM
// bandwidth: a load instruction costs the CU about (fixed + per-enabled-lane) cycles, so what a record costs is its
// number of 16-byte load instructions, whatever the table size up to the Infinity Cache's.
//
//   hipcc --offload-arch=gfx950 -O2 -o node_fetch node_fetch.hip && ./node_fetch
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(128) Rec {
    float4 q[8];   // q[0].w holds the next record's index (bits)
};

template <int NLOADS, bool SADDR>
__global__ __launch_bounds__(256) void walk(const Rec* __restrict__ table, uint32_t n, int steps, int lanes_on, int valu, float* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (lane >= lanes_on) return;
    uint32_t cur = (tid * 2654435761u) % n;
    float acc = 0.0f;
    const char* base = reinterpret_cast<const char*>(table);
    for (int s = 0; s < steps; s++) {
        float4 v[NLOADS];
        if (SADDR) {
            const uint32_t off = cur << 7;   // scalar base + 32-bit offset (tables up to 4 GB)
#pragma unroll
            for (int k = 0; k < NLOADS; k++) v[k] = *reinterpret_cast<const float4*>(base + (off + 16u * k));
        } else {
            const Rec* r = table + cur;
#pragma unroll
            for (int k = 0; k < NLOADS; k++) v[k] = r->q[k];
        }
        float x = 0.0f;
#pragma unroll
        for (int k = 0; k < NLOADS; k++) x += (v[k].x + v[k].y) + (v[k].z + (k == 0 ? 0.0f : v[k].w));   // every component is used: the loads stay whole
        for (int k = 0; k < valu; k++) x = x * 1.0001f + 0.5f;   // dependent chain: `valu` VALU instructions
        acc += x;
        cur = __float_as_uint(v[0].w);
    }
    out[tid] = acc;
}

// QUAD-cooperative fetch (round 5): the four lanes of a quad fetch the quad's four 64-byte records together -- load j asks, in every
// lane c of the quad, for bytes 16c..16c+15 of the record of the quad's lane j -- so a load instruction touches one 64-byte line per
// quad (fully used) instead of one per lane (a quarter used).  Does the vector-memory path charge per lane or per line?  (No
// transpose here: every lane sums what it loaded; the chain continues at the index in the first 16 bytes, which lane 0 of the
// record's quad slot holds and broadcasts.)
template <bool LDS_TRANSPOSE>
__global__ __launch_bounds__(256) void walk_quad(const Rec* __restrict__ table, uint32_t n, int steps, int lanes_on, int valu, float* __restrict__ out)
{
    __shared__ float4 xch[256 * 4];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // `lanes_on` active rays per wave, spread over the wave like a walk's survivors: lane i holds a ray iff (i * 37) % 64 < lanes_on
    const bool has_ray = ((lane * 37) & 63) < lanes_on;
    uint32_t cur = (tid * 2654435761u) % n;
    float acc = 0.0f;
    const char* base = reinterpret_cast<const char*>(table);
    const uint32_t c = (uint32_t)(lane & 3);
    for (int s = 0; s < steps; s++) {
        float4 v[4];
        const uint32_t my = has_ray ? (cur << 7) : 0xffffffffu;
        // the record offset of the quad's lane j (quad_perm:[j,j,j,j] broadcast), this lane's 16 bytes of it
        const uint32_t off4[4] = {(uint32_t)__builtin_amdgcn_mov_dpp((int)my, 0x00, 0xf, 0xf, true), (uint32_t)__builtin_amdgcn_mov_dpp((int)my, 0x55, 0xf, 0xf, true),
                                  (uint32_t)__builtin_amdgcn_mov_dpp((int)my, 0xaa, 0xf, 0xf, true), (uint32_t)__builtin_amdgcn_mov_dpp((int)my, 0xff, 0xf, 0xf, true)};
#pragma unroll
        for (int j = 0; j < 4; j++)
            v[j] = off4[j] != 0xffffffffu ? *reinterpret_cast<const float4*>(base + (off4[j] + 16u * c)) : make_float4(0, 0, 0, 0);
        float4 r[4];
        if (LDS_TRANSPOSE) {
            // through LDS: lane c of the quad holds chunk c of records 0..3; record j's chunks go to the slot of the quad's lane j
            const uint32_t q4 = (threadIdx.x & ~3u);
#pragma unroll
            for (int j = 0; j < 4; j++) xch[(q4 + j) * 4 + c] = v[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = xch[threadIdx.x * 4 + k];
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = v[k];
        }
        float x = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) x += (r[k].x + r[k].y) + (r[k].z + (LDS_TRANSPOSE && k == 0 ? 0.0f : r[k].w));
        for (int k = 0; k < valu; k++) x = x * 1.0001f + 0.5f;
        acc += x;
        // the next index: in the first 16 bytes of my own record = what lane 0 of my quad loaded in load (lane & 3) (or r[0].w after the transpose)
        uint32_t nxt;
        if (LDS_TRANSPOSE) nxt = __float_as_uint(r[0].w);
        else {
            // lane 0 of the quad holds chunk 0 of records 0..3 in v[0..3].w; lane j needs v[j].w of lane 0
            const uint32_t w0 = __float_as_uint(v[0].w), w1 = __float_as_uint(v[1].w), w2 = __float_as_uint(v[2].w), w3 = __float_as_uint(v[3].w);
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)w0, 0x00, 0xf, 0xf, true), b1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)w1, 0x00, 0xf, 0xf, true);
            const uint32_t b2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)w2, 0x00, 0xf, 0xf, true), b3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)w3, 0x00, 0xf, 0xf, true);
            nxt = c == 0 ? b0 : (c == 1 ? b1 : (c == 2 ? b2 : b3));
        }
        if (has_ray) cur = nxt % n;
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

template <int NLOADS, bool SADDR>
static void run(const Rec* d, uint32_t n, int n_cu, int waves, int lanes_on, int valu, float* out)
{
    const int steps = 300;
    const int blocks = n_cu * waves;
    hipLaunchKernelGGL((walk<NLOADS, SADDR>), dim3(blocks), dim3(256), 0, 0, d, n, 20, lanes_on, valu, out);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((walk<NLOADS, SADDR>), dim3(blocks), dim3(256), 0, 0, d, n, steps, lanes_on, valu, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fetches = (double)blocks * 4.0 * lanes_on * steps;
    const double per_us_cu = fetches / (ms * 1e3) / n_cu;
    const double instr_per_cu = (double)waves * 4.0 * steps * NLOADS;   // wave-level load instructions one CU issued
    std::printf("table %4.0f MB  %s  loads/record %d  waves/SIMD %d  lanes %2d  valu/step %3d: %7.1f records/us/CU, %5.1f ns of CU time per load "
                "instruction, step time per wave %.0f ns\n",
                (double)n * 128.0 / 1048576.0, SADDR ? "sbase+off32" : "addr64     ", NLOADS, waves, lanes_on, valu, per_us_cu,
                ms * 1e6 / instr_per_cu, ms * 1e6 / steps);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

template <bool LDS_TRANSPOSE>
static void run_quad(const Rec* d, uint32_t n, int n_cu, int waves, int lanes_on, int valu, float* out)
{
    const int steps = 300;
    const int blocks = n_cu * waves;
    hipLaunchKernelGGL((walk_quad<LDS_TRANSPOSE>), dim3(blocks), dim3(256), 0, 0, d, n, 20, lanes_on, valu, out);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((walk_quad<LDS_TRANSPOSE>), dim3(blocks), dim3(256), 0, 0, d, n, steps, lanes_on, valu, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fetches = (double)blocks * 4.0 * lanes_on * steps;
    std::printf("table %4.0f MB  QUAD-cooperative%s  loads/record 4  waves/SIMD %d  rays per wave %2d  valu/step %3d: %7.1f records/us/CU, %5.1f ns of CU time per load "
                "instruction, step time per wave %.0f ns\n",
                (double)n * 128.0 / 1048576.0, LDS_TRANSPOSE ? " + LDS transpose" : "                ", waves, lanes_on, valu, fetches / (ms * 1e3) / n_cu,
                ms * 1e6 / ((double)waves * 4.0 * steps * 4), ms * 1e6 / steps);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::printf("%s, %d CUs\n", prop.gcnArchName, n_cu);
    float* out;
    CK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * sizeof(float)));
    for (uint32_t n : {49152u, 786432u}) {   // 6 MB, 96 MB
        std::vector<Rec> h(n);
        uint32_t s = 12345u;
        for (uint32_t i = 0; i < n; i++) {
            for (int k = 0; k < 8; k++) h[i].q[k] = make_float4(1e-3f * (float)(i & 255), 0.5f, 0.25f, 0.125f);
            s = s * 1664525u + 1013904223u;
            const uint32_t next = (s >> 4) % n;
            h[i].q[0].w = *reinterpret_cast<const float*>(&next);
        }
        Rec* d;
        CK(hipMalloc(&d, (size_t)n * sizeof(Rec)));
        CK(hipMemcpy(d, h.data(), (size_t)n * sizeof(Rec), hipMemcpyHostToDevice));
        for (int lanes_on : {64, 24, 16}) {
            run<1, false>(d, n, n_cu, 7, lanes_on, 64, out);
            run<2, false>(d, n, n_cu, 7, lanes_on, 64, out);
            run<4, false>(d, n, n_cu, 7, lanes_on, 64, out);
            run<7, false>(d, n, n_cu, 7, lanes_on, 64, out);
            run<8, false>(d, n, n_cu, 7, lanes_on, 64, out);
            run<4, true>(d, n, n_cu, 7, lanes_on, 64, out);
            run<7, true>(d, n, n_cu, 7, lanes_on, 64, out);
        }
        for (int lanes_on : {64, 24, 16}) {
            run_quad<false>(d, n, n_cu, 7, lanes_on, 64, out);
            run_quad<true>(d, n, n_cu, 7, lanes_on, 64, out);
        }
        for (int lanes_on : {64, 24}) {   // (the plain walk with rays spread over the wave the same way: lanes_on counts rays, not a prefix)
            run<4, true>(d, n, n_cu, 7, lanes_on, 128, out);
            run_quad<true>(d, n, n_cu, 7, lanes_on, 128, out);
        }
        if (n == 49152u) {
            for (int waves : {2, 4, 8}) run<7, true>(d, n, n_cu, waves, 24, 64, out);
            for (int valu : {0, 192, 384}) run<7, true>(d, n, n_cu, 7, 24, valu, out);
            for (int valu : {0, 192, 384}) run<4, true>(d, n, n_cu, 7, 24, valu, out);
        }
        CK(hipFree(d));
    }
    return 0;
}
