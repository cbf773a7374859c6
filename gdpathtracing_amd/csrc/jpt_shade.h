// jpt_shade.h -- per-path device code shared by every tracing kernel: RNG, primary-ray generation,
// shading-record fetch, the diffuse + GGX mixture BRDF with VNDF sampling, and the path update.
// Replaces main.glsl:163-222,372-421 and brdfs.glsl:1-138 of the reference
// (project/addons/jar_path_tracing/src/shaders/).
#pragma once

#include "jpt_device_math.h"
#include "jpt_types.h"

namespace jpt {

#define JPT_PI 3.141592653589793238462643f  // brdfs.glsl:1

struct Ray {  // main.glsl:26-30
    f3 o, d, rD;
};

struct Hit {  // the fields of HitInfo (main.glsl:62-71) that survive traversal
    float t;
    float u, v;
    uint32_t tri;   // index into the reference-order triangle arrays
    uint32_t inst;  // BLAS instance id
    f3 lo, ld;      // ray origin / direction in the hit instance's local space (position = lo + t*ld)
};

struct Shading {  // main.glsl:73-82
    f3 position, normal, out_dir;
    float lambert_out;
    f3 emission, diffuse_albedo, fresnel_0;
    float roughness;
};

struct SceneShading {  // cold, once-per-hit data: kept in the reference layout
    const ShadeTri* __restrict__ tri_data;
    const RefInstance* __restrict__ instances;
    const RefMaterial* __restrict__ materials;
    const uint8_t* __restrict__ tex;
    const ReachTri* __restrict__ reach_tri;    // reach records (jpt_types.h), null when the scene has none
    const ReachInst* __restrict__ reach_inst;
    bool retrace_ties;                         // the reference's own trees are on the device: a hit the walk flagged as an exact
                                               // distance tie is set aside and traced again on them (wf2_finish)
    uint32_t n_materials, n_instances;
    int32_t tex_res, n_layers;
    int32_t sampler_mode;  // JPT_SAMPLER_*: bit 0 repeat, bit 1 linear
};

#if defined(__HIPCC__)   // (everything below is device code; the host layer -- jpt_capi.cpp, jpt_multi.cpp -- sees the structs above only)

// ---- RNG (main.glsl:163-181) -----------------------------------------------------------------

__device__ __forceinline__ void pcg2d(uint32_t& sx, uint32_t& sy, float& rx, float& ry)
{
    uint32_t x = 1664525u * sx + 1013904223u;
    uint32_t y = 1664525u * sy + 1013904223u;
    x += 1664525u * y;
    y += 1664525u * x;
    x ^= x >> 16;
    y ^= y >> 16;
    x += 1664525u * y;
    y += 1664525u * x;
    x ^= x >> 16;
    y ^= y >> 16;
    sx = x;
    sy = y;
    rx = (float)x * 2.32830643654e-10f;
    ry = (float)y * 2.32830643654e-10f;
}

__device__ __forceinline__ void prng_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t& sx, uint32_t& sy)
{
    uint32_t x = px * 0x9e3779b9u + frame;
    uint32_t y = py * 0x9e3779b9u + frame;
    x ^= x >> 16;
    y ^= y >> 16;
    sx = x * 0x9e3779b9u;
    sy = y * 0x9e3779b9u;
}

// ---- primary ray (main.glsl:405-421, box_muller :183-187) -------------------------------------

__device__ __forceinline__ Ray primary_ray(const RefCamera& cam, int width, int height, int px, int py, uint32_t frame,
                                           uint32_t& sx, uint32_t& sy)
{
    prng_seed((uint32_t)px, (uint32_t)py, frame, sx, sy);
    float r0, r1;
    pcg2d(sx, sy, r0, r1);
    r1 = r1 * 0.25f;
    // box_muller: R = sqrt(-2 log(r0)) is dead code in the reference; only theta is used
    float js, jc;
    sincos_(6.2831853f * r1, js, jc);
    const float scx = ((float)px + jc) / (float)width * 2.0f - 1.0f;
    const float scy = ((float)py + js) / (float)height * 2.0f - 1.0f;
    const float nx = scx, ny = -scy;
    const float* m = cam.ivp;
    float wx = m[0] * nx + m[4] * ny + m[8] + m[12];
    float wy = m[1] * nx + m[5] * ny + m[9] + m[13];
    float wz = m[2] * nx + m[6] * ny + m[10] + m[14];
    const float ww = m[3] * nx + m[7] * ny + m[11] + m[15];
    wx = wx / ww;
    wy = wy / ww;
    wz = wz / ww;
    Ray ray;
    ray.o = mk3(cam.position.x, cam.position.y, cam.position.z);
    ray.d = normalize3(mk3(wx, wy, wz) - ray.o);
    ray.rD = rcp3(ray.d);
    return ray;
}

// The ray through an exact raster position (no jitter): the arithmetic of primary_ray from `scx` on.  Used to bound what the
// jittered rays of a pixel can see (wf2_accumulate); `ww_out` = the clip-space w the position divides by.
__device__ __forceinline__ f3 raster_direction(const RefCamera& cam, int width, int height, float fx, float fy, float& ww_out)
{
    const float scx = fx / (float)width * 2.0f - 1.0f;
    const float scy = fy / (float)height * 2.0f - 1.0f;
    const float nx = scx, ny = -scy;
    const float* m = cam.ivp;
    float wx = m[0] * nx + m[4] * ny + m[8] + m[12];
    float wy = m[1] * nx + m[5] * ny + m[9] + m[13];
    float wz = m[2] * nx + m[6] * ny + m[10] + m[14];
    const float ww = m[3] * nx + m[7] * ny + m[11] + m[15];
    ww_out = ww;
    wx = wx / ww;
    wy = wy / ww;
    wz = wz / ww;
    return normalize3(mk3(wx, wy, wz) - mk3(cam.position.x, cam.position.y, cam.position.z));
}

// The same direction to a few ulp, for callers that only BOUND what a pixel's rays can see (wf2_accumulate's sky cells, which
// keep a margin a thousand times the rounding of this arithmetic): the six divisions and the square root of raster_direction as
// reciprocal estimates (v_rcp_f32 / v_rsq_f32, 1 ulp each), 30 instructions instead of 90.  Never used for a value that is stored.
__device__ __forceinline__ f3 raster_direction_approx(const RefCamera& cam, float two_over_w, float two_over_h, float fx, float fy, float& ww_out)
{
    const float nx = fx * two_over_w - 1.0f, ny = -(fy * two_over_h - 1.0f);
    const float* m = cam.ivp;
    const float wx = m[0] * nx + m[4] * ny + m[8] + m[12];
    const float wy = m[1] * nx + m[5] * ny + m[9] + m[13];
    const float wz = m[2] * nx + m[6] * ny + m[10] + m[14];
    const float ww = m[3] * nx + m[7] * ny + m[11] + m[15];
    ww_out = ww;
    const float iw = __builtin_amdgcn_rcpf(ww);
    const f3 v = mk3(wx * iw - cam.position.x, wy * iw - cam.position.y, wz * iw - cam.position.z);
    const float il = __builtin_amdgcn_rsqf(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk3(v.x * il, v.y * il, v.z * il);
}

__device__ __forceinline__ f3 sample_sky(f3 d)  // main.glsl:189-192
{
    const float t = 0.5f * (d.y + 1.0f);
    return mk3(mix_(0.95f, 0.9f, t) * 1.0f, mix_(0.95f, 0.94f, t) * 1.0f, mix_(0.95f, 1.0f, t) * 1.0f);
}

// texture(textureArray, vec3(uv, layer)) (main.glsl:214).  The sampler state is a parameter (jpt_set_params); what
// each mode computes is pinned in oracle/oracle_trace.c::sample_texture (Vulkan texel addressing: nearest floor(u*res),
// linear around u*res - 0.5, clamp-to-edge or non-negative modulo, mix() of the four UNORM8 texels, no sRGB decode).
__device__ __forceinline__ int tex_index(float f, int res, bool repeat)
{
    if (f != f) return 0;
    if (!repeat) {   // clamp-to-edge saturates BEFORE the cast: +inf and 1e9 are the last texel, -inf the first
        if (f >= (float)(res - 1)) return res - 1;
        return f <= 0.0f ? 0 : (int)f;
    }
    if (f >= 1073741824.0f || f <= -1073741824.0f) return 0;   // repeat: the modulo of a float this large is not defined by the pin
    int i = (int)f;
    i %= res;
    return i < 0 ? i + res : i;
}
__device__ __forceinline__ f3 texel(const SceneShading& sc, int layer, int ix, int iy)
{
    const uint32_t p = reinterpret_cast<const uint32_t*>(sc.tex)[((size_t)layer * sc.tex_res + iy) * sc.tex_res + ix];
    return mk3(from_unorm8(p & 255u), from_unorm8((p >> 8) & 255u), from_unorm8((p >> 16) & 255u));
}
// FILTER: 0 the sampler mode's own bit decides at run time, 1 nearest only, 2 linear only (the shading kernel is instantiated
// per filter: the bilinear path's four texels and weights cost it a wave per SIMD)
template <int FILTER = 0>
__device__ __forceinline__ f3 sample_texture(const SceneShading& sc, float u, float v, int layer)
{
    if (!sc.tex || sc.n_layers <= 0 || sc.tex_res <= 0) return mk3(0.0f, 0.0f, 0.0f);
    if (layer >= sc.n_layers) layer = sc.n_layers - 1;
    const int res = sc.tex_res;
    const bool repeat = (sc.sampler_mode & 1) != 0, linear = FILTER == 0 ? (sc.sampler_mode & 2) != 0 : FILTER == 2;
    const float x = u * (float)res, y = v * (float)res;
    if (!linear) return texel(sc, layer, tex_index(__builtin_floorf(x), res, repeat), tex_index(__builtin_floorf(y), res, repeat));
    const float xs = x - 0.5f, ys = y - 0.5f;
    const float fx = __builtin_floorf(xs), fy = __builtin_floorf(ys);
    float a = xs - fx, b = ys - fy;
    if (a != a) a = 0.0f;
    if (b != b) b = 0.0f;
    const int x0 = tex_index(fx, res, repeat), x1 = tex_index(fx + 1.0f, res, repeat);
    const int y0 = tex_index(fy, res, repeat), y1 = tex_index(fy + 1.0f, res, repeat);
    const f3 t00 = texel(sc, layer, x0, y0), t10 = texel(sc, layer, x1, y0);
    const f3 t01 = texel(sc, layer, x0, y1), t11 = texel(sc, layer, x1, y1);
    const f3 r0 = mk3(mix_(t00.x, t10.x, a), mix_(t00.y, t10.y, a), mix_(t00.z, t10.z, a));
    const f3 r1 = mk3(mix_(t01.x, t11.x, a), mix_(t01.y, t11.y, a), mix_(t01.z, t11.z, a));
    return mk3(mix_(r0.x, r1.x, b), mix_(r0.y, r1.y, b), mix_(r0.z, r1.z, b));
}

// ---- shading record (main.glsl:194-222) -------------------------------------------------------

// the triangle's shading record as four aligned 16-byte loads: n0.xyz n1.x | n1.yz n2.xy | n2.z uv0 uv1.x | uv1.y uv2 slot
struct ShadeTriRegs {
    float4 q0, q1, q2, q3;
};
__device__ __forceinline__ ShadeTriRegs load_shade_tri(const SceneShading& sc, uint32_t tri)
{
    const float4* tq = reinterpret_cast<const float4*>(sc.tri_data + tri);
    return ShadeTriRegs{tq[0], tq[1], tq[2], tq[3]};
}

// (the record is passed in so that a caller can ask for it early, together with its other gathers)
// TEX = 0: the scene has no texture array, so texture() returns zero for every material that names a layer (what
// sample_texture answers then, without carrying the sampler code in the kernel)
// (TEX: 0 no texture array, 1 nearest filter, 2 linear filter, 3 either, decided by the sampler mode at run time)
template <int TEX = 3>
__device__ __forceinline__ Shading get_shading_data(const SceneShading& sc, const Hit& h, bool front, const ShadeTriRegs& tr)
{
    Shading s;
    const float4 q0 = tr.q0, q1 = tr.q1, q2 = tr.q2, q3 = tr.q3;
    const RefInstance& b = sc.instances[h.inst];
    const uint32_t slot = __float_as_uint(q3.w);
    // b.materials[tri.materialIndex] is unchecked in the reference (main.glsl:198): slots past 2 read on into the next
    // instance's record; a read past the END of the instance buffer returns 0 (Vulkan robust buffer access; the same
    // pin as the oracle's), so no uploaded material_index can make the kernel read outside the array
    const unsigned long long word = (unsigned long long)h.inst * 44ull + 41ull + (unsigned long long)slot;
    uint32_t mat_id = word < (unsigned long long)sc.n_instances * 44ull ? reinterpret_cast<const uint32_t*>(sc.instances)[word] : 0u;
    if (mat_id >= sc.n_materials) mat_id = 0;
    const RefMaterial& material = sc.materials[mat_id];

    const float u = h.u, v = h.v;
    const float w0 = 1.0f - u - v;
    // (the texture first, while little else is live: the sampler's four texels and weights are the kernel's register peak;
    // the operations and their order per value are those of main.glsl:200-218 wherever they stand)
    f3 albedo = mk3(material.albedo.x, material.albedo.y, material.albedo.z);
    if (material.albedo_texture_index >= 0) {
        const float uvx = q2.y * w0 + q2.w * u + q3.y * v;
        const float uvy = q2.z * w0 + q3.x * u + q3.z * v;
        albedo = albedo * (TEX == 0 ? mk3(0.0f, 0.0f, 0.0f) : sample_texture<(TEX == 3 ? 0 : TEX)>(sc, uvx, uvy, material.albedo_texture_index));
    }
    const f3 lpos = h.lo + h.ld * h.t;  // hitInfo.position = ray.o + t * ray.d (main.glsl:249)
    s.position = xform_point(b.transform, lpos);
    s.out_dir = normalize3(xform_dir(b.transform, -h.ld));
    f3 n = mk3(q0.x, q0.y, q0.z) * w0 + mk3(q0.w, q1.x, q1.y) * u + mk3(q1.z, q1.w, q2.x) * v;
    n = normalize3(xform_dir(b.transform, n));
    s.normal = front ? n : -n;

    s.lambert_out = dot3(s.normal, s.out_dir);
    const float em = fmax_(0.0f, material.emission.w);
    s.emission = mk3(material.emission.x * em, material.emission.y * em, material.emission.z * em);

    const float metalicity = material.metallic;
    s.fresnel_0 = mk3(mix_(0.02f, albedo.x, metalicity), mix_(0.02f, albedo.y, metalicity), mix_(0.02f, albedo.z, metalicity));
    s.diffuse_albedo = albedo - albedo * metalicity;
    s.roughness = fmax_(0.006f, material.roughness);
    return s;
}

// ---- BRDF (brdfs.glsl) ------------------------------------------------------------------------

__device__ __forceinline__ float schlick_factor(float cosine_theta)  // brdfs.glsl:4-6
{
    const float factor = 1.0f - cosine_theta;
    const float factor_squared = factor * factor;
    return factor_squared * factor_squared * factor;
}

__device__ __forceinline__ f3 brdf_eval(const Shading& sh, f3 l)  // brdfs.glsl:10-38
{
    const float n_dot_light = dot3(sh.normal, l);
    const float n_dot_view = sh.lambert_out;
    if (fmin_(n_dot_light, n_dot_view) < 0.0f) return mk3(0.0f, 0.0f, 0.0f);

    const f3 half_vector = normalize3(l + sh.out_dir);
    const float half_dot_view = dot3(half_vector, sh.out_dir);

    const float f90 = (half_dot_view * half_dot_view) * (2.0f * sh.roughness) + 0.5f;
    const float diffuse_fresnel = mix_(1.0f, f90, schlick_factor(n_dot_view)) * mix_(1.0f, f90, schlick_factor(n_dot_light));
    f3 r = mk3(diffuse_fresnel * sh.diffuse_albedo.x, diffuse_fresnel * sh.diffuse_albedo.y,
               diffuse_fresnel * sh.diffuse_albedo.z);

    const float half_dot_normal = dot3(half_vector, sh.normal);
    const float roughness_sq = sh.roughness * sh.roughness;
    const float denominator = half_dot_normal * (roughness_sq - 1.0f) + 1.0f;  // un-squared n.h, as the reference
    const float distribution = roughness_sq / (denominator * denominator);

    const float masking = n_dot_light * __builtin_sqrtf((n_dot_view - roughness_sq * n_dot_view) * n_dot_view + roughness_sq);
    const float shadowing =
        n_dot_view * __builtin_sqrtf((n_dot_light - roughness_sq * n_dot_light) * n_dot_light + roughness_sq);
    const float geometry = 0.5f / (masking + shadowing);

    const float ff = schlick_factor(fmax_(0.0f, half_dot_view));
    const f3 spec_f = mk3(mix_(sh.fresnel_0.x, 1.0f, ff), mix_(sh.fresnel_0.y, 1.0f, ff), mix_(sh.fresnel_0.z, 1.0f, ff));
    const float dg = distribution * geometry;
    r = r + mk3(dg * spec_f.x, dg * spec_f.y, dg * spec_f.z);
    return r / JPT_PI;
}

__device__ __forceinline__ float diffuse_probability(const Shading& sh)  // brdfs.glsl:107-110
{
    return fmin_(0.5f, dot3(sh.diffuse_albedo, mk3(0.2126f, 0.7152f, 0.0722f)));
}

__device__ __forceinline__ f3 sample_brdf(const Shading& sh, float xi0, float xi1)  // brdfs.glsl:112-128
{
    // get_shading_space (brdfs.glsl:83-93)
    const f3 nrm = sh.normal;
    const float sign = nrm.z > 0.0f ? 1.0f : -1.0f;
    const float a = -1.0f / (sign + nrm.z);
    const float b = nrm.x * nrm.y * a;
    const f3 c0 = mk3(1.0f + sign * nrm.x * nrm.x * a, sign * b, -sign * nrm.x);
    const f3 c1 = mk3(b, sign + nrm.y * nrm.y * a, -nrm.y);
    const f3 c2 = nrm;

    const float diffuse_prob = diffuse_probability(sh);
    // The two strategies share their azimuth: sin / cos of 2 pi xi0' with xi0' rescaled per strategy.  The lanes of a wave
    // choose at random, so a wave runs both sides of the branch; the 60-instruction sincos_ is kept out of it.
    const bool diffuse = xi0 < diffuse_prob;
    if (diffuse) xi0 = xi0 / diffuse_prob;
    else xi0 = (xi0 - diffuse_prob) / (1.0f - diffuse_prob);
    float sp, cp;
    sincos_((2.0f * JPT_PI) * xi0, sp, cp);
    f3 local;
    if (diffuse) {
        // sample_hemisphere_psa (brdfs.glsl:95-101)
        const float radius = __builtin_sqrtf(xi1);
        const float z = __builtin_sqrtf(1.0f - radius * radius);
        local = mk3(radius * cp, radius * sp, z);
    } else {
        const f3 view = mk3(dot3(c0, sh.out_dir), dot3(c1, sh.out_dir), dot3(c2, sh.out_dir));
        // sample_ggx_vndf (brdfs.glsl:40-54), roughness = vec2(r)
        const float rg = sh.roughness;
        const f3 tv = normalize3(mk3(view.x * rg, view.y * rg, view.z));
        const float z = 1.0f - xi1 * (1.0f + tv.z);
        const float sin_theta = __builtin_sqrtf(fmax_(0.0f, 1.0f - z * z));
        const f3 hs = mk3(sin_theta * cp, sin_theta * sp, z);
        const f3 sum = hs + tv;
        const f3 h = normalize3(mk3(sum.x * rg, sum.y * rg, sum.z));
        // sample_ggx_in_dir: -reflect(view, h) (brdfs.glsl:69-72)
        const float k = 2.0f * dot3(h, view);
        local = -(view - h * k);
    }
    return mk3(c0.x * local.x + c1.x * local.y + c2.x * local.z, c0.y * local.x + c1.y * local.y + c2.y * local.z,
               c0.z * local.x + c1.z * local.y + c2.z * local.z);
}

__device__ __forceinline__ float brdf_density(const Shading& sh, f3 l)  // brdfs.glsl:130-138, :74-81, :56-67, :103-105
{
    const float diffuse_prob = diffuse_probability(sh);
    const f3 half_vector = normalize3(l + sh.out_dir);
    const float half_dot_view = dot3(half_vector, sh.out_dir);
    const float half_dot_normal = dot3(half_vector, sh.normal);
    float vndf = 0.0f;
    if (!(half_dot_normal < 0.0f)) {
        const float n_dot_view = sh.lambert_out;
        const float roughness_sq = sh.roughness * sh.roughness;
        const float inv_roughness_sq = 1.0f - roughness_sq;
        const float denominator = n_dot_view + __builtin_sqrtf(roughness_sq + inv_roughness_sq * n_dot_view * n_dot_view);
        const float d_vis = fmax_(0.0f, half_dot_view) * (2.0f / JPT_PI) / denominator;
        const float m_sq_term = 1.0f - inv_roughness_sq * half_dot_normal * half_dot_normal;
        vndf = d_vis * roughness_sq / (m_sq_term * m_sq_term);
    }
    const float specular_density = vndf / (4.0f * half_dot_view);
    const float diffuse_density = fmax_(0.0f, dot3(sh.normal, l)) / JPT_PI;
    return mix_(specular_density, diffuse_density, diffuse_prob);
}

// One path vertex after a hit (main.glsl:381-394).  Returns false when the path ends
// (lambert_in <= 0); otherwise `ray` is the next segment and `throughput` is updated.
__device__ __forceinline__ bool bounce_step(const Shading& s, uint32_t& sx, uint32_t& sy, Ray& ray, f3& throughput)
{
    ray.o = s.position + s.normal * 0.001f;
    float xi0, xi1;
    pcg2d(sx, sy, xi0, xi1);
    ray.d = sample_brdf(s, xi0, xi1);
    ray.rD = rcp3(ray.d);
    const float density = brdf_density(s, ray.d);
    const float lambert_in = dot3(s.normal, ray.d);
    if (lambert_in <= 0.0f) return false;
    const f3 f = (brdf_eval(s, ray.d) * lambert_in) / density;
    throughput = throughput * f;
    return true;
}

__device__ __forceinline__ f3 aces_film(f3 x)  // progressive_rendering.glsl:19-26
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return mk3(clamp_((x.x * (a * x.x + b)) / (x.x * (c * x.x + d) + e), 0.0f, 1.0f),
               clamp_((x.y * (a * x.y + b)) / (x.y * (c * x.y + d) + e), 0.0f, 1.0f),
               clamp_((x.z * (a * x.z + b)) / (x.z * (c * x.z + d) + e), 0.0f, 1.0f));
}

#endif  // __HIPCC__

}  // namespace jpt
