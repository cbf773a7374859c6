// jpt_nodeq.h -- the 64-byte quantised form of a four-child record, made from the float form (WideNode4) by one function
// that the host (upload) and the device (TLAS refit) share.
//
// Why: wf2_trace is bound by the CU's vector-memory path, which charges about 16 ns of CU time per wave-level load
// instruction at the ~24 lanes a record step has enabled (tools/micro/node_fetch.hip: 7.7 ns + 0.33 ns per enabled
// lane, whatever the table size up to the Infinity Cache's), not by VALU issue and not by bytes: a record step costs
// what its number of 16-byte loads costs.  The float record needs seven of them; this one needs four.
//
//   bytes  0..15   origin.xyz (the lo corner of the union of the children's boxes), scale.x
//         16..31   scale.y, scale.z, lo_x, lo_y        each plane word holds the four children's planes, one byte each
//         32..47   lo_z, hi_x, hi_y, hi_z              (child k in byte k): plane = origin + q * scale, scale a power of 2
//         48..63   child[0..3]                         same references as WideNode4
//
// The planes are rounded outwards (lo down, hi up), so a quantised box contains the float box; with a power-of-two
// scale and q <= 255 the products q * scale are exact.  Boxes grow by at most scale = extent / 255 .. extent / 127 per
// side, i.e. a child of half the node's size is inflated by 1-2 %.
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

JPT_HD float qfloor_(float x) { const float t = (float)(int)x; return t > x ? t - 1.0f : t; }   // |x| < 2^23 here
JPT_HD float qceil_(float x) { const float t = (float)(int)x; return t < x ? t + 1.0f : t; }

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
        // 254 steps for the extent leave the last step for the rounding of (plane - origin)
        s[a] = (ext > 0.0f && ext < 3.0e38f) ? pow2_at_least(ext / 254.0f) : pow2_at_least(0.0f);
    }
    q.ox = lo[0]; q.oy = lo[1]; q.oz = lo[2];
    q.sx = s[0]; q.sy = s[1]; q.sz = s[2];
    uint32_t wlo[3] = {0u, 0u, 0u}, whi[3] = {0u, 0u, 0u};
    for (int k = 0; k < 4; k++) {
        q.child[k] = n.child[k];
        const bool empty = n.child[k] == kEmptyChild;
        const float cl[3] = {n.lo_x[k], n.lo_y[k], n.lo_z[k]}, ch[3] = {n.hi_x[k], n.hi_y[k], n.hi_z[k]};
        for (int a = 0; a < 3; a++) {
            float ql = 255.0f, qh = 0.0f;   // an empty slot: lo beyond hi
            if (!empty) {
                ql = qfloor_((cl[a] - lo[a]) / s[a]);
                qh = qceil_((ch[a] - lo[a]) / s[a]);
                ql = ql < 0.0f ? 0.0f : (ql > 255.0f ? 255.0f : ql);
                qh = qh < 0.0f ? 0.0f : (qh > 255.0f ? 255.0f : qh);
            }
            wlo[a] |= (uint32_t)ql << (8 * k);
            whi[a] |= (uint32_t)qh << (8 * k);
        }
    }
    q.lo_x = wlo[0]; q.lo_y = wlo[1]; q.lo_z = wlo[2];
    q.hi_x = whi[0]; q.hi_y = whi[1]; q.hi_z = whi[2];
}

}  // namespace jpt
