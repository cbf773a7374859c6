// jpt_nodeq.h -- the 64-byte quantised form of a four-child record, made from the float form (WideNode4) by one function
// that the host (upload) and the device (TLAS refit) share.
//
// Why: a record step of wf2_trace pays twice -- in VALU issue (the kernel uses about two thirds of the chip's VALU issue
// capacity, profiles/current_sq.json: valu_issue_frac) and in the CU's vector-memory path, which charges about 16 ns of CU
// time per wave-level load instruction at the ~24 lanes a record step has enabled (tools/micro/node_fetch.hip: 7.7 ns +
// 0.33 ns per enabled lane, whatever the table size up to the Infinity Cache's) -- and not in bytes.  The float record
// needs seven 16-byte loads and 199 VALU instructions per step; this one four loads and 131.
//
//   bytes  0..15   origin.xyz (the lo corner of the union of the children's boxes), scale.x
//         16..31   scale.y, scale.z, lo_x, lo_y        each plane word holds the four children's planes, one byte each
//         32..47   lo_z, hi_x, hi_y, hi_z              (child k in byte k)
//         48..63   child[0..3]                         same references as WideNode4
//
// A child's box is [origin + (lo - kPlaneSlack) * scale, origin + (hi + kPlaneSlack) * scale] per axis, scale = extent /
// 254 (the grid's ends lie on the node's own faces), kPlaneSlack = 1/256 of a step.  lo is the largest and hi the
// smallest step for which that interval contains the float box, so a quantised box contains the float box and grows by
// less than a step = extent / 254 per side.  The slack is what keeps FLAT children flat: a wall that lies in a face of
// its parent has lo = hi (the float box is only as thick as the builder's padding, a fraction of the slack), and a
// bounce ray that starts 0.001 above the wall does not enter its box again -- with planes rounded outwards to whole
// steps the wall became a slab a step thick, and every ray leaving a wall re-tested the wall's triangles (+30 %
// triangle tests on the Cornell scenes).  The kernel applies the slack as two biases per axis (jpt_trace_core.h).
#pragma once

#include "jpt_instance_math.h"   // JPT_HD, imin_, imax_
#include "jpt_types.h"

namespace jpt {

struct alignas(64) WideNodeQ {
    float ox, oy, oz, sx;
    float sy, sz;
    uint32_t lo_x, lo_y;
    uint32_t lo_z, hi_x, hi_y, hi_z;
    int32_t child[4];
};
static_assert(sizeof(WideNodeQ) == 64, "WideNodeQ");

constexpr float kPlaneSlack = 1.0f / 256.0f;

// What the WALK adds to a plane's distance on top of the slack, per axis: kWalkEps * (|b| + |o * rD|), b = the distance of
// the record's origin plane.  The step evaluates t = q * (scale * rD) + (origin * rD - o * rD) with rD from v_rcp_f32 (1 ulp)
// and three roundings; the reference's own box test -- the one the reach records repeat -- evaluates (plane - o) * (1 / d)
// with three others.  Each is off by a few 2^-24 of |b| + |o * rD| of ITS axis, independently per axis, which at a ray
// origin tens of mesh sizes away exceeds the builder's padding (2e-6 of the mesh's largest coordinate): without this
// term a ray through the corner of a far, small box could pass the reference's test and fail this one
// (tests/test_quantized_walk.py).  2^-20 = 16 x 2^-24 covers both evaluations with a factor two to spare.  The slack the
// walk applies is 1/256 + 1/16384 of a step: the second part covers the rounding of scale * rD times a plane number.
constexpr float kWalkEps = 1.0f / 1048576.0f;
constexpr float kWalkSlackOverEps = 4160.0f;   // (1/256 + 1/16384) / kWalkEps, exactly
constexpr float kRcpClamp = 1.2676506002282294e30f;   // 2^100: the walk's reciprocal directions are clamped to +-this (jpt_trace_core.h, set_level)

// smallest power of two >= x (x > 0, finite); tiny or zero extents get the smallest normal number
JPT_HD float pow2_at_least(float x)
{
    union {
        float f;
        uint32_t u;
    } v;
    v.f = x;
    uint32_t e = (v.u >> 23) & 0xffu;
    if ((v.u & 0x7fffffu) != 0u) e++;
    if (e < 1u) e = 1u;
    if (e > 254u) e = 254u;
    v.u = e << 23;
    return v.f;
}

// floor / ceil of a plane number.  The callers clamp the result to [0, 255]; values outside [-4, 300] (huge or infinite boxes
// of degenerate or hostile records) are clamped BEFORE the integer conversion, which is undefined for them, and a NaN is
// passed through (the callers give a NaN box the whole range).  In range: exactly what the plain conversion gave.
JPT_HD double qfloor_(double x)
{
    if (!(x >= -4.0 && x <= 300.0)) return x != x ? x : (x < 0.0 ? -4.0 : 300.0);
    const double t = (double)(long long)x;
    return t > x ? t - 1.0 : t;
}
JPT_HD double qceil_(double x)
{
    if (!(x >= -4.0 && x <= 300.0)) return x != x ? x : (x < 0.0 ? -4.0 : 300.0);
    const double t = (double)(long long)x;
    return t < x ? t + 1.0 : t;
}

JPT_HD void quantize_node4(const WideNode4& n, WideNodeQ& q)
{
    float lo[3] = {0.0f, 0.0f, 0.0f}, hi[3] = {0.0f, 0.0f, 0.0f};
    bool any = false;
    for (int k = 0; k < 4; k++) {
        if (n.child[k] == kEmptyChild) continue;
        const float cl[3] = {n.lo_x[k], n.lo_y[k], n.lo_z[k]}, ch[3] = {n.hi_x[k], n.hi_y[k], n.hi_z[k]};
        for (int a = 0; a < 3; a++) {
            lo[a] = any ? imin_(lo[a], cl[a]) : cl[a];
            hi[a] = any ? imax_(hi[a], ch[a]) : ch[a];
        }
        any = true;
    }
    float s[3];
    for (int a = 0; a < 3; a++) {
        const float ext = hi[a] - lo[a];
        s[a] = (ext > 0.0f && ext < 3.0e38f) ? ext / 254.0f : 0.0f;
        if (!(s[a] >= 1.17549435e-38f)) s[a] = 1.17549435e-38f;   // flat or degenerate node: the smallest normal number
    }
    q.ox = lo[0]; q.oy = lo[1]; q.oz = lo[2];
    q.sx = s[0]; q.sy = s[1]; q.sz = s[2];
    uint32_t wlo[3] = {0u, 0u, 0u}, whi[3] = {0u, 0u, 0u};
    for (int k = 0; k < 4; k++) {
        q.child[k] = n.child[k];
        const bool empty = n.child[k] == kEmptyChild;
        const float cl[3] = {n.lo_x[k], n.lo_y[k], n.lo_z[k]}, ch[3] = {n.hi_x[k], n.hi_y[k], n.hi_z[k]};
        for (int a = 0; a < 3; a++) {
            double ql = 255.0, qh = 0.0;   // an empty slot: lo beyond hi
            if (!empty) {
                // exact differences and quotients of floats in double, up to 1e-16 relative: far inside the builder's padding
                ql = qfloor_(((double)cl[a] - (double)lo[a]) / (double)s[a] + (double)kPlaneSlack);
                qh = qceil_(((double)ch[a] - (double)lo[a]) / (double)s[a] - (double)kPlaneSlack);
                ql = ql >= 0.0 ? (ql > 255.0 ? 255.0 : ql) : 0.0;   // (NaN boxes of degenerate scenes: whole range)
                qh = qh <= 255.0 ? (qh < 0.0 ? 0.0 : qh) : 255.0;
            }
            wlo[a] |= (uint32_t)ql << (8 * k);
            whi[a] |= (uint32_t)qh << (8 * k);
        }
    }
    q.lo_x = wlo[0]; q.lo_y = wlo[1]; q.lo_z = wlo[2];
    q.hi_x = whi[0]; q.hi_y = whi[1]; q.hi_z = whi[2];
}

}  // namespace jpt
