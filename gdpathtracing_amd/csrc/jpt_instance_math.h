// jpt_instance_math.h -- the BLASInstance record of one instance from its transform (geometry_group3d.cpp:322-341,
// bvh.h:81-115), written once for the host builder (jpt_builder.cpp) and for the device refit kernel
// (jpt_kernels_post.hip): plain float arithmetic only, compiled without contraction on both sides, so the two give
// the same bits.
#pragma once

#include "jpt_types.h"

#if defined(__HIPCC__)
#define JPT_HD __host__ __device__ inline
#else
#define JPT_HD inline
#endif

namespace jpt {

JPT_HD float imin_(float a, float b) { return (b < a) ? b : a; }  // std::min
JPT_HD float imax_(float a, float b) { return (a < b) ? b : a; }  // std::max
JPT_HD float iabs_(float a) { return a < 0.0f ? -a : a; }

// Transform3D (rows of the basis, then the origin) -> column-major mat4 (geometry_group3d.cpp:330-333)
JPT_HD void transform12_to_mat16(const float* t, float* m)
{
    for (int c = 0; c < 3; c++) {
        m[c * 4 + 0] = t[0 * 3 + c];
        m[c * 4 + 1] = t[1 * 3 + c];
        m[c * 4 + 2] = t[2 * 3 + c];
        m[c * 4 + 3] = 0.0f;
    }
    m[12] = t[9];
    m[13] = t[10];
    m[14] = t[11];
    m[15] = 1.0f;
}

// godot Basis::invert (cofactors) + Transform3D::affine_inverse, float.  godot-cpp (branch 4.3) is an
// absent submodule; this is its published algorithm.  Runs before the path: it only fills the matrices
// of the BLASInstance record.
JPT_HD void affine_inverse12(const float* t, float* o)
{
#define JPT_R(r, c) t[(r) * 3 + (c)]
#define JPT_COF(r1, c1, r2, c2) (JPT_R(r1, c1) * JPT_R(r2, c2) - JPT_R(r1, c2) * JPT_R(r2, c1))
    const float co0 = JPT_COF(1, 1, 2, 2), co1 = JPT_COF(1, 2, 2, 0), co2 = JPT_COF(1, 0, 2, 1);
    const float det = JPT_R(0, 0) * co0 + JPT_R(0, 1) * co1 + JPT_R(0, 2) * co2;
    const float s = 1.0f / det;
    float m[9];
    m[0] = co0 * s; m[1] = JPT_COF(0, 2, 2, 1) * s; m[2] = JPT_COF(0, 1, 1, 2) * s;
    m[3] = co1 * s; m[4] = JPT_COF(0, 0, 2, 2) * s; m[5] = JPT_COF(0, 2, 1, 0) * s;
    m[6] = co2 * s; m[7] = JPT_COF(0, 1, 2, 0) * s; m[8] = JPT_COF(0, 0, 1, 1) * s;
#undef JPT_COF
#undef JPT_R
    for (int k = 0; k < 9; k++) o[k] = m[k];
    const float nx = -t[9], ny = -t[10], nz = -t[11];
    o[9] = m[0] * nx + m[1] * ny + m[2] * nz;
    o[10] = m[3] * nx + m[4] * ny + m[5] * nz;
    o[11] = m[6] * nx + m[7] * ny + m[8] * nz;
}

// aabbMin / aabbMax of an instance from its column-major transform and the box of its BLAS root
// (BLASInstance::update_aabb, bvh.h:90-115); `pad_box` = the native builder's outward padding of the world box
JPT_HD void instance_world_box(const float* transform16, const Vec4& root_min, const Vec4& root_max, bool pad_box, Vec4& lo_out, Vec4& hi_out)
{
    Vec4 lo = Vec4{1e34f, 1e34f, 1e34f, 1.0f};
    Vec4 hi = Vec4{-1e34f, -1e34f, -1e34f, 1.0f};
    for (int i = 0; i < 8; i++) {
        const float corner[4] = {(i & 1) ? root_max.x : root_min.x, (i & 2) ? root_max.y : root_min.y,
                                 (i & 4) ? root_max.z : root_min.z, 1.0f};
        float tc[4] = {0.0f, 0.0f, 0.0f, 1.0f};
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) tc[j] += transform16[k * 4 + j] * corner[k];
        const float s = 2.0f / tc[3];
        const Vec4 c{tc[0] * s, tc[1] * s, tc[2] * s, tc[3] * s};
        lo = Vec4{imin_(lo.x, c.x), imin_(lo.y, c.y), imin_(lo.z, c.z), imin_(lo.w, c.w)};
        hi = Vec4{imax_(hi.x, c.x), imax_(hi.y, c.y), imax_(hi.z, c.z), imax_(hi.w, c.w)};
    }
    if (pad_box) {
        // pad the world box like the BLAS boxes (the affine map of the corners rounds)
        float m = 0.0f;
        m = imax_(m, imax_(iabs_(lo.x), iabs_(hi.x)));
        m = imax_(m, imax_(iabs_(lo.y), iabs_(hi.y)));
        m = imax_(m, imax_(iabs_(lo.z), iabs_(hi.z)));
        const float pad = m * 2e-6f;
        lo = Vec4{lo.x - pad, lo.y - pad, lo.z - pad, 1.0f};
        hi = Vec4{hi.x + pad, hi.y + pad, hi.z + pad, 1.0f};
    }
    lo_out = lo;
    hi_out = hi;
}

// transform, inverse_transform, aabbMin, aabbMax of `inst` from the instance's Transform3D and the box of its BLAS
// root (BLASInstance::set_transform, bvh.h:81-88)
JPT_HD void instance_record(const float* t12, const Vec4& root_min, const Vec4& root_max, bool pad_box, RefInstance& inst)
{
    float inv12[12];
    affine_inverse12(t12, inv12);
    transform12_to_mat16(t12, inst.transform);
    transform12_to_mat16(inv12, inst.inverse_transform);
    instance_world_box(inst.transform, root_min, root_max, pad_box, inst.aabbMin, inst.aabbMax);
}

}  // namespace jpt
