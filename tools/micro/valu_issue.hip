// valu_issue.hip -- how many cycles does one SIMD of gfx950 need per wave64 VALU instruction?
//
// The question behind DESIGN.md's issue-rate accounting: MI355X_MICROARCH.md says "SIMD-32, 2 cycles per wave64
// instruction with >= 2 waves per SIMD, 4 for a wave alone"; the SQ counters of round 1 (SQ_ACTIVE_INST_VALU /
// SQ_INSTS_VALU = 1.0 quad-cycle) read 4.  This program measures it directly: every wave runs a loop of 16
// INDEPENDENT instructions of one kind (16 accumulators, inline asm so nothing is folded), with 1, 2, 4 or 8 waves
// resident per SIMD (blocks of 256 threads = one wave per SIMD, `w` blocks per CU), and reports
//     cycles per instruction per SIMD = wave cycles (s_memtime) / (waves per SIMD x instructions per wave)
// from the median wave, plus the same figure from wall time at the clock the chip held (s_memrealtime).
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind { FMA = 0, MUL, MIN, MAX3, CNDMASK, PK_FMA, PK_MUL, RCP, FMA_DEP, FMA_16LANES, MIX_SALU, N_KINDS };
static const char* kKindName[N_KINDS] = {"v_fma_f32", "v_mul_f32", "v_min_f32", "v_max3_f32", "v_cndmask_b32", "v_pk_fma_f32",
                                         "v_pk_mul_f32", "v_rcp_f32", "v_fma_f32 (one dependent chain)", "v_fma_f32, 16 of 64 lanes enabled",
                                         "v_fma_f32 + s_add_u32 interleaved (VALU count only)"};

template <int KIND>
__global__ __launch_bounds__(256) void issue_loop(float* __restrict__ out, unsigned long long* __restrict__ cycles,
                                                  unsigned long long* __restrict__ realtime, int iters)
{
    float a[16];
    float2 p[16];
    for (int k = 0; k < 16; k++) {
        a[k] = (float)(threadIdx.x + k) * 1e-3f;
        p[k] = make_float2(a[k], a[k] + 1.0f);
    }
    float b = 1.0001f, c = 1e-6f;
    float2 pb = make_float2(1.0001f, 0.9999f), pc = make_float2(1e-6f, 2e-6f);
    unsigned int s = 0;
    const bool on = KIND != FMA_16LANES || (threadIdx.x & 63) < 16;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int i = 0; i < iters; i++) {
            if (KIND == FMA || KIND == FMA_16LANES) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                REP16(X)
#undef X
            } else if (KIND == MUL) {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                REP16(X)
#undef X
            } else if (KIND == MIN) {
#define X(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                REP16(X)
#undef X
            } else if (KIND == MAX3) {
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                REP16(X)
#undef X
            } else if (KIND == CNDMASK) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
                REP16(X)
#undef X
            } else if (KIND == PK_FMA) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
                REP16(X)
#undef X
            } else if (KIND == PK_MUL) {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
                REP16(X)
#undef X
            } else if (KIND == RCP) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                REP16(X)
#undef X
            } else if (KIND == FMA_DEP) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                REP16(X)
#undef X
            } else if (KIND == MIX_SALU) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 1" : "+v"(a[k]), "+s"(s) : "v"(b), "v"(c));
                REP16(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float sum = (float)s;
    for (int k = 0; k < 16; k++) sum += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        cycles[w] = t1 - t0;
        realtime[w] = r1 - r0;
    }
}

template <int KIND>
static void run_kind(float* out, unsigned long long* d_cyc, unsigned long long* d_rt, int n_cu)
{
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        const int blocks = n_cu * w;
        hipLaunchKernelGGL(issue_loop<KIND>, dim3(blocks), dim3(256), 0, 0, out, d_cyc, d_rt, 200);  // warm-up
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(issue_loop<KIND>, dim3(blocks), dim3(256), 0, 0, out, d_cyc, d_rt, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> cyc((size_t)blocks * 4), rt((size_t)blocks * 4);
        hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(rt.data(), d_rt, rt.size() * 8, hipMemcpyDeviceToHost);
        std::sort(cyc.begin(), cyc.end());
        std::sort(rt.begin(), rt.end());
        const double med = (double)cyc[cyc.size() / 2], med_rt = (double)rt[rt.size() / 2];
        const double n_inst = (double)iters * 16.0;
        const double clock_ghz = med / (med_rt * 10.0);  // s_memrealtime ticks at 100 MHz
        std::printf("%-52s waves/SIMD %d: %6.3f cycles per instruction per SIMD (median wave %.0f cycles, in-kernel clock %.2f GHz, "
                    "launch %.3f ms => %.3f from wall time)\n",
                    kKindName[KIND], w, med / ((double)w * n_inst), med, clock_ghz, ms,
                    (double)ms * 1e-3 * clock_ghz * 1e9 / ((double)w * n_inst));
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    std::printf("%s, %d CUs, %d MHz\n", prop.gcnArchName, n_cu, prop.clockRate / 1000);
    float* out;
    unsigned long long *d_cyc, *d_rt;
    hipMalloc(&out, (size_t)n_cu * 8 * 256 * sizeof(float));
    hipMalloc(&d_cyc, (size_t)n_cu * 8 * 4 * 8);
    hipMalloc(&d_rt, (size_t)n_cu * 8 * 4 * 8);
    run_kind<FMA>(out, d_cyc, d_rt, n_cu);
    run_kind<MUL>(out, d_cyc, d_rt, n_cu);
    run_kind<MIN>(out, d_cyc, d_rt, n_cu);
    run_kind<MAX3>(out, d_cyc, d_rt, n_cu);
    run_kind<CNDMASK>(out, d_cyc, d_rt, n_cu);
    run_kind<PK_FMA>(out, d_cyc, d_rt, n_cu);
    run_kind<PK_MUL>(out, d_cyc, d_rt, n_cu);
    run_kind<RCP>(out, d_cyc, d_rt, n_cu);
    run_kind<FMA_DEP>(out, d_cyc, d_rt, n_cu);
    run_kind<FMA_16LANES>(out, d_cyc, d_rt, n_cu);
    run_kind<MIX_SALU>(out, d_cyc, d_rt, n_cu);
    return 0;
}
