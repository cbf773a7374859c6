// jpt_debug.hip -- audit entry points of the C ABI (include/jpt.h, jpt_debug_*): pieces of the native walk run on
// caller-made inputs, so that properties the images only sample can be tested directly.
//
// jpt_debug_node_step4: the four-child record step of the native walk (Traversal<.., W4>::node_step4, jpt_trace_core.h) on
// (record, ray, distance bound) triples -- which of the four children does the walk keep?  The step's box tests are
// CONSERVATIVE tests on quantised planes (jpt_nodeq.h) with reciprocals from v_rcp_f32; what the tests must never do is
// drop a child whose box holds a triangle that Moller-Trumbore as written accepts (tests/test_quantized_walk.py feeds
// adversarial triples: axis-parallel rays, flat nodes, far ray origins, slivers, planes on grid steps).  On a device the
// kernel below runs the very function the tracing kernels inline; with JPT_DEVICE_HOST_ONLY a host restatement of the same
// arithmetic runs, its reciprocals perturbed by a chosen number of ulps (v_rcp_f32 is accurate to 1 ulp).
#include "../../include/jpt.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "jpt_nodeq.h"
#include "jpt_trace_core.h"

using namespace jpt;

namespace {

thread_local std::string g_debug_error;

struct StepCase {   // 32 bytes
    float o[3], d[3];
    float t_max;    // hitInfo.t when the record is expanded
    uint32_t node;  // which record
};
static_assert(sizeof(StepCase) == 32, "StepCase");

__global__ void node_step4_probe(const WideNodeQ* __restrict__ nodes, const StepCase* __restrict__ cases, uint32_t n, uint8_t* __restrict__ taken)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const StepCase c = cases[i];
    WideSceneDev sc;
    std::memset(&sc, 0, sizeof sc);
    sc.nodesq = nodes;
    sc.n_instances = 1;
    Traversal<false, true> tr;
    tr.begin(sc, mk3(c.o[0], c.o[1], c.o[2]), mk3(c.d[0], c.d[1], c.d[2]));   // (sets the slab constants the way a walk does)
    tr.hit.t = c.t_max;
    tr.cur = (int32_t)c.node;
    tr.in_blas = true;
    tr.have = true;
    int32_t pushed[4] = {0, 0, 0, 0};
    const Traversal<false, true>::Stack st{nullptr, pushed, 0, 0, 4};   // every push lands in `pushed`
    DevCounters cnt = {};
    tr.node_step4(sc, st, cnt);
    // the children the walk keeps: the one it descends into and the ones it pushed.  The probe's records name child k
    // as reference k + 1 (never dereferenced).
    uint32_t mask = 0;
    if (tr.have && tr.cur >= 1 && tr.cur <= 4) mask |= 1u << (tr.cur - 1);
    for (int k = 0; k < tr.sp && k < 4; k++)
        if (pushed[k] >= 1 && pushed[k] <= 4) mask |= 1u << (pushed[k] - 1);
    taken[i] = (uint8_t)mask;
}

float step_ulps(float x, int ulps)
{
    for (int k = 0; k < (ulps < 0 ? -ulps : ulps); k++) x = std::nextafterf(x, ulps > 0 ? INFINITY : -INFINITY);
    return x;
}

// the arithmetic of node_step4's box tests, restated for the host (same operations in the same order; std::fmaf is the
// fused multiply-add, std::fmaxf / fminf ignore a NaN operand like v_max_f32 / v_min_f32)
uint8_t node_step4_host(const WideNodeQ& q, const StepCase& c, int rcp_ulps)
{
    float rD[3], ood[3], a[3], nb[3], fb[3];
    const float origin[3] = {q.ox, q.oy, q.oz}, scale[3] = {q.sx, q.sy, q.sz};
    const uint32_t lo[3] = {q.lo_x, q.lo_y, q.lo_z}, hi[3] = {q.hi_x, q.hi_y, q.hi_z};
    uint32_t nw[3], fw[3];
    for (int k = 0; k < 3; k++) {
        float r = 1.0f / c.d[k];
        if (std::isfinite(r) && r != 0.0f) r = step_ulps(r, r > 0.0f ? rcp_ulps : -rcp_ulps);   // |r| larger for ulps > 0
        r = r > kRcpClamp ? kRcpClamp : (r < -kRcpClamp ? -kRcpClamp : r);                         // (set_level keeps them finite)
        rD[k] = r;
        ood[k] = -(c.o[k] * rD[k]);
        a[k] = scale[k] * rD[k];
        const float b = std::fmaf(origin[k], rD[k], ood[k]);
        const float m = std::fmaf(kWalkSlackOverEps, std::fabs(a[k]), std::fabs(b)) + std::fabs(ood[k]);
        nb[k] = std::fmaf(-kWalkEps, m, b);
        fb[k] = std::fmaf(kWalkEps, m, b);
        uint32_t bits;
        std::memcpy(&bits, &c.d[k], 4);
        const bool neg = (int32_t)bits < 0;
        nw[k] = neg ? hi[k] : lo[k];
        fw[k] = neg ? lo[k] : hi[k];
    }
    uint32_t mask = 0;
    for (int k = 0; k < 4; k++) {
        float t_in = 0.0f, t_out = c.t_max;
        float tin_axes[3], tout_axes[3];
        for (int ax = 0; ax < 3; ax++) {
            tin_axes[ax] = std::fmaf((float)((nw[ax] >> (8 * k)) & 255u), a[ax], nb[ax]);
            tout_axes[ax] = std::fmaf((float)((fw[ax] >> (8 * k)) & 255u), a[ax], fb[ax]);
        }
        t_in = std::fmax(std::fmax(std::fmax(tin_axes[0], tin_axes[1]), tin_axes[2]), 0.0f);
        t_out = std::fmin(std::fmin(std::fmin(tout_axes[0], tout_axes[1]), tout_axes[2]), c.t_max);
        if (t_in <= t_out && q.child[k] != kEmptyChild) mask |= 1u << k;
    }
    return (uint8_t)mask;
}

}  // namespace

extern "C" {

const char* jpt_debug_last_error(void) { return g_debug_error.c_str(); }

int jpt_debug_quantize_nodes4(const void* nodes4, uint32_t n_nodes, void* nodesq_out)
{
    if ((n_nodes && !nodes4) || !nodesq_out) return JPT_E_INVALID;
    const WideNode4* in = static_cast<const WideNode4*>(nodes4);
    WideNodeQ* out = static_cast<WideNodeQ*>(nodesq_out);
    for (uint32_t i = 0; i < n_nodes; i++) {
        WideNode4 n;
        std::memcpy(&n, reinterpret_cast<const char*>(in) + (size_t)i * sizeof(WideNode4), sizeof n);
        WideNodeQ q;
        quantize_node4(n, q);
        std::memcpy(reinterpret_cast<char*>(out) + (size_t)i * sizeof(WideNodeQ), &q, sizeof q);
    }
    return JPT_OK;
}

int jpt_debug_node_step4(int device_id, const void* nodes4, uint32_t n_nodes, const void* cases32, uint32_t n_cases, int32_t host_rcp_ulps,
                         uint8_t* taken_out)
{
    if (!nodes4 || !cases32 || !taken_out || n_nodes == 0) {
        g_debug_error = "null argument";
        return JPT_E_INVALID;
    }
    std::vector<WideNodeQ> q(n_nodes);
    if (jpt_debug_quantize_nodes4(nodes4, n_nodes, q.data()) != JPT_OK) return JPT_E_INVALID;
    const StepCase* cases = static_cast<const StepCase*>(cases32);
    for (uint32_t i = 0; i < n_cases; i++)
        if (cases[i].node >= n_nodes) {
            g_debug_error = "a case names a record that does not exist";
            return JPT_E_INVALID;
        }
    if (device_id == JPT_DEVICE_HOST_ONLY) {
        for (uint32_t i = 0; i < n_cases; i++) taken_out[i] = node_step4_host(q[cases[i].node], cases[i], host_rcp_ulps);
        return JPT_OK;
    }
    auto hip_fail = [](hipError_t e, const char* what) {
        g_debug_error = std::string(what) + ": " + hipGetErrorString(e);
        return JPT_E_DEVICE;
    };
    hipError_t e;
    if ((e = hipSetDevice(device_id)) != hipSuccess) return hip_fail(e, "hipSetDevice");
    WideNodeQ* d_nodes = nullptr;
    StepCase* d_cases = nullptr;
    uint8_t* d_taken = nullptr;
    int rc = JPT_OK;
    if ((e = hipMalloc((void**)&d_nodes, (size_t)n_nodes * sizeof(WideNodeQ))) != hipSuccess) rc = hip_fail(e, "hipMalloc");
    if (rc == JPT_OK && n_cases && (e = hipMalloc((void**)&d_cases, (size_t)n_cases * sizeof(StepCase))) != hipSuccess) rc = hip_fail(e, "hipMalloc");
    if (rc == JPT_OK && n_cases && (e = hipMalloc((void**)&d_taken, n_cases)) != hipSuccess) rc = hip_fail(e, "hipMalloc");
    if (rc == JPT_OK && (e = hipMemcpy(d_nodes, q.data(), (size_t)n_nodes * sizeof(WideNodeQ), hipMemcpyHostToDevice)) != hipSuccess) rc = hip_fail(e, "hipMemcpy");
    if (rc == JPT_OK && n_cases) {
        if ((e = hipMemcpy(d_cases, cases, (size_t)n_cases * sizeof(StepCase), hipMemcpyHostToDevice)) != hipSuccess) rc = hip_fail(e, "hipMemcpy");
        if (rc == JPT_OK) {
            hipLaunchKernelGGL(node_step4_probe, dim3((n_cases + 255u) / 256u), dim3(256), 0, nullptr, d_nodes, d_cases, n_cases, d_taken);
            if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) rc = hip_fail(e, "node_step4_probe");
        }
        if (rc == JPT_OK && (e = hipMemcpy(taken_out, d_taken, n_cases, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "hipMemcpy");
    }
    if (d_nodes) (void)hipFree(d_nodes);
    if (d_cases) (void)hipFree(d_cases);
    if (d_taken) (void)hipFree(d_taken);
    return rc;
}

}  // extern "C"
