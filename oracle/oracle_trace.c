/*
 * oracle_trace.c -- line-by-line CPU restatement of main.glsl, brdfs.glsl and
 * progressive_rendering.glsl.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see
 * jpt_oracle.h).  Citations are to
 *   M = project/addons/jar_path_tracing/src/shaders/main.glsl
 *   B = project/addons/jar_path_tracing/src/shaders/brdfs.glsl
 *   P = project/addons/jar_path_tracing/src/shaders/progressive_rendering.glsl
 */
#include "jpt_oracle.h"
#include "oracle_pins.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define PI_F 3.141592653589793238462643f /* B:1 */

/* ------------------------------------------------------------------ RNG */

/* M:163-174 */
static void pcg2d(uint32_t seed[2], float out[2])
{
    uint32_t x = seed[0], y = seed[1];
    x = 1664525u * x + 1013904223u;
    y = 1664525u * y + 1013904223u;
    x += 1664525u * y;
    y += 1664525u * x;
    x ^= (x >> 16u);
    y ^= (y >> 16u);
    x += 1664525u * y;
    y += 1664525u * x;
    x ^= (x >> 16u);
    y ^= (y >> 16u);
    seed[0] = x;
    seed[1] = y;
    out[0] = (float)x * 2.32830643654e-10f;
    out[1] = (float)y * 2.32830643654e-10f;
}

/* M:176-181 */
static void prng_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t seed[2])
{
    uint32_t x = px, y = py;
    x = x * 0x9e3779b9u + frame;
    y = y * 0x9e3779b9u + frame;
    x ^= x >> 16u;
    y ^= y >> 16u;
    seed[0] = x * 0x9e3779b9u;
    seed[1] = y * 0x9e3779b9u;
}

/* M:183-187.  R = sqrt(-2 log(rands.x)) is computed and discarded by the reference. */
static void box_muller(const float rands[2], float out[2])
{
    float theta = 6.2831853f * rands[1];
    float s, c;
    p_sincos(theta, &s, &c);
    out[0] = c;
    out[1] = s;
}

/* ------------------------------------------------------------------ types */

typedef struct { v3 d, o, rD; } ray_t;             /* M:26-30 */

typedef struct {                                    /* M:62-71 */
    v3 position;
    float t;
    uint32_t blas;
    uint32_t triangle;
    uint32_t steps;
    float bary_u, bary_v;
    int front;
    v3 out_dir;
} hit_t;

typedef struct {                                    /* M:73-82 */
    v3 position;
    v3 normal;
    v3 out_dir;
    float lambert_out;
    v3 emission;
    v3 diffuse_albedo;
    v3 fresnel_0;
    float roughness;
} shading_t;

typedef struct {
    const jpto_scene_view *sc;
    uint32_t flags;
    jpto_counters cnt;
} ctx_t;

static inline v3 v4xyz(jpto_vec4 v) { return v3_make(v.x, v.y, v.z); }

/* ------------------------------------------------------------------ BRDF (brdfs.glsl) */

/* B:3-8 */
static v3 fresnel_schlick(v3 f0, v3 f90, float cosine_theta)
{
    float factor = 1.0f - cosine_theta;
    float factor_squared = factor * factor;
    float factor_fifth = factor_squared * factor_squared * factor;
    return v3_make(p_mix(f0.x, f90.x, factor_fifth), p_mix(f0.y, f90.y, factor_fifth),
                   p_mix(f0.z, f90.z, factor_fifth));
}

/* B:10-38 */
static v3 brdf(const shading_t *sh, v3 light_dir)
{
    float n_dot_light = v3_dot(sh->normal, light_dir);
    float n_dot_view = sh->lambert_out;

    if (p_min(n_dot_light, n_dot_view) < 0.0f) return v3_make(0.0f, 0.0f, 0.0f);

    v3 half_vector = v3_normalize(v3_add(light_dir, sh->out_dir));
    float half_dot_view = v3_dot(half_vector, sh->out_dir);

    float f90 = (half_dot_view * half_dot_view) * (2.0f * sh->roughness) + 0.5f;
    v3 one = v3_make(1.0f, 1.0f, 1.0f);
    v3 f90v = v3_make(f90, f90, f90);
    float diffuse_fresnel = fresnel_schlick(one, f90v, n_dot_view).x * fresnel_schlick(one, f90v, n_dot_light).x;

    v3 r = v3_make(diffuse_fresnel * sh->diffuse_albedo.x, diffuse_fresnel * sh->diffuse_albedo.y,
                   diffuse_fresnel * sh->diffuse_albedo.z);

    float half_dot_normal = v3_dot(half_vector, sh->normal);
    float roughness_sq = sh->roughness * sh->roughness;
    float denominator = half_dot_normal * (roughness_sq - 1.0f) + 1.0f; /* un-squared n.h, as written (B:27) */
    float distribution = roughness_sq / (denominator * denominator);

    float masking = n_dot_light * sqrtf((n_dot_view - roughness_sq * n_dot_view) * n_dot_view + roughness_sq);
    float shadowing = n_dot_view * sqrtf((n_dot_light - roughness_sq * n_dot_light) * n_dot_light + roughness_sq);
    float geometry = 0.5f / (masking + shadowing);

    v3 specular_fresnel = fresnel_schlick(sh->fresnel_0, one, p_max(0.0f, half_dot_view));
    float dg = distribution * geometry;
    r = v3_add(r, v3_make(dg * specular_fresnel.x, dg * specular_fresnel.y, dg * specular_fresnel.z));

    return v3_divs(r, PI_F);
}

/* B:40-54 */
static v3 sample_ggx_vndf(v3 view_dir, float rough_x, float rough_y, const float random_sample[2])
{
    v3 transformed_view = v3_normalize(v3_make(view_dir.x * rough_x, view_dir.y * rough_y, view_dir.z));
    float phi = (2.0f * PI_F) * random_sample[0];
    float z = 1.0f - random_sample[1] * (1.0f + transformed_view.z);

    float sin_theta = sqrtf(p_max(0.0f, 1.0f - z * z));
    float sp, cp;
    p_sincos(phi, &sp, &cp);
    v3 hemisphere_sample = v3_make(sin_theta * cp, sin_theta * sp, z);

    v3 sum = v3_add(hemisphere_sample, transformed_view);
    v3 half_vector = v3_normalize(v3_make(sum.x * rough_x, sum.y * rough_y, sum.z));
    return half_vector;
}

/* B:56-67 */
static float get_ggx_vndf_density(float n_dot_view, float half_dot_normal, float half_dot_view, float roughness)
{
    if (half_dot_normal < 0.0f) return 0.0f;

    float roughness_sq = roughness * roughness;
    float inv_roughness_sq = 1.0f - roughness_sq;
    float denominator = n_dot_view + sqrtf(roughness_sq + inv_roughness_sq * n_dot_view * n_dot_view);

    float d_vis = p_max(0.0f, half_dot_view) * (2.0f / PI_F) / denominator;
    float m_sq_term = 1.0f - inv_roughness_sq * half_dot_normal * half_dot_normal;

    return d_vis * roughness_sq / (m_sq_term * m_sq_term);
}

/* B:69-72.  reflect(I,N) = I - (2*dot(N,I))*N */
static v3 sample_ggx_in_dir(v3 view_dir, float roughness, const float random_sample[2])
{
    v3 half_vector = sample_ggx_vndf(view_dir, roughness, roughness, random_sample);
    float k = 2.0f * v3_dot(half_vector, view_dir);
    v3 refl = v3_sub(view_dir, v3_scale(half_vector, k));
    return v3_neg(refl);
}

/* B:74-81 */
static float get_ggx_in_dir_density(float n_dot_view, v3 view_dir, v3 light_dir, v3 normal, float roughness)
{
    v3 half_vector = v3_normalize(v3_add(light_dir, view_dir));
    float half_dot_view = v3_dot(half_vector, view_dir);
    float half_dot_normal = v3_dot(half_vector, normal);

    float density = get_ggx_vndf_density(n_dot_view, half_dot_normal, half_dot_view, roughness);
    return density / (4.0f * half_dot_view);
}

/* B:83-93: columns c0,c1,c2 */
static void get_shading_space(v3 normal, v3 *c0, v3 *c1, v3 *c2)
{
    float sign = normal.z > 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign + normal.z);
    float b = normal.x * normal.y * a;

    *c0 = v3_make(1.0f + sign * normal.x * normal.x * a, sign * b, -sign * normal.x);
    *c1 = v3_make(b, sign + normal.y * normal.y * a, -normal.y);
    *c2 = normal;
}

/* B:95-101 */
static v3 sample_hemisphere_psa(const float random_sample[2])
{
    float phi = (2.0f * PI_F) * random_sample[0];
    float radius = sqrtf(random_sample[1]);
    float z = sqrtf(1.0f - radius * radius);
    float sp, cp;
    p_sincos(phi, &sp, &cp);
    return v3_make(radius * cp, radius * sp, z);
}

/* B:103-105 */
static float get_hemisphere_psa_density(float sampled_z) { return p_max(0.0f, sampled_z) / PI_F; }

/* B:107-110 */
static float get_diffuse_sampling_probability(const shading_t *sh)
{
    float luminance = v3_dot(sh->diffuse_albedo, v3_make(0.2126f, 0.7152f, 0.0722f));
    return p_min(0.5f, luminance);
}

static inline v3 m3_mul(v3 c0, v3 c1, v3 c2, v3 v)
{
    return v3_make(c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z,
                   c0.z * v.x + c1.z * v.y + c2.z * v.z);
}

/* B:112-128 */
static v3 sample_brdf(const shading_t *sh, const float random_in[2])
{
    float random_sample[2] = {random_in[0], random_in[1]};
    v3 c0, c1, c2;
    get_shading_space(sh->normal, &c0, &c1, &c2);
    float diffuse_prob = get_diffuse_sampling_probability(sh);

    v3 sampled_dir;
    if (random_sample[0] < diffuse_prob) {
        random_sample[0] /= diffuse_prob;
        sampled_dir = m3_mul(c0, c1, c2, sample_hemisphere_psa(random_sample));
    } else {
        random_sample[0] = (random_sample[0] - diffuse_prob) / (1.0f - diffuse_prob);
        v3 local_view = v3_make(v3_dot(c0, sh->out_dir), v3_dot(c1, sh->out_dir), v3_dot(c2, sh->out_dir));
        v3 local_light = sample_ggx_in_dir(local_view, sh->roughness, random_sample);
        sampled_dir = m3_mul(c0, c1, c2, local_light);
    }
    return sampled_dir;
}

/* B:130-138 */
static float get_brdf_density(const shading_t *sh, v3 sampled_dir)
{
    float diffuse_prob = get_diffuse_sampling_probability(sh);
    float specular_density =
        get_ggx_in_dir_density(sh->lambert_out, sh->out_dir, sampled_dir, sh->normal, sh->roughness);
    float diffuse_density = get_hemisphere_psa_density(v3_dot(sh->normal, sampled_dir));
    return p_mix(specular_density, diffuse_density, diffuse_prob);
}

/* ------------------------------------------------------------------ scene queries (main.glsl) */

/* M:189-192 */
static v3 sample_sky(v3 direction)
{
    float t = 0.5f * (direction.y + 1.0f);
    return v3_make(p_mix(0.95f, 0.9f, t) * 1.0f, p_mix(0.95f, 0.94f, t) * 1.0f, p_mix(0.95f, 1.0f, t) * 1.0f);
}

/* texture(textureArray, vec3(uv, layer)) (M:214).  The sampler state lives in the absent gdcs, so the mode is a
 * parameter (sampler_mode: bit 0 = repeat instead of clamp-to-edge, bit 1 = linear instead of nearest); what each
 * mode computes is pinned here after the Vulkan texel-addressing rules:
 *   unnormalised coordinate  x = u * res                        (one float multiply)
 *   nearest                  i = floor(x)
 *   linear                   i0 = floor(x - 0.5), i1 = i0 + 1, weight a = (x - 0.5) - floor(x - 0.5)
 *   clamp-to-edge            i = clamp(i, 0, res - 1);     repeat   i = i mod res (non-negative remainder)
 *   NaN                      index 0 (weight 0);  repeat with |floor| >= 2^30: index 0;  clamp-to-edge saturates first,
 *                            so +inf / 1e9 address the last texel and -inf the first, as Vulkan's clamp does
 *   filter                   mix(mix(t00, t10, a), mix(t01, t11, a), b) per channel, mix = a*(1-t) + b*t (oracle_pins.h);
 *                            texels are UNORM8 -> float by p_from_unorm8, no sRGB decode (path_tracing_camera.cpp:182) */
static int32_t tex_index(float f, int32_t res, int repeat)
{
    if (f != f) return 0;
    if (!repeat) {   /* clamp-to-edge saturates BEFORE the cast: +inf and 1e9 are the last texel, -inf the first */
        if (f >= (float)(res - 1)) return res - 1;
        return f <= 0.0f ? 0 : (int32_t)f;   /* f is integral */
    }
    if (f >= 1073741824.0f || f <= -1073741824.0f) return 0;
    int32_t i = (int32_t)f;
    i %= res;
    return i < 0 ? i + res : i;
}
static v3 texel(const jpto_scene_view *sc, int32_t layer, int32_t ix, int32_t iy)
{
    const int32_t res = sc->tex_res;
    const uint8_t *p = sc->tex_rgba8 + (((size_t)layer * res + iy) * res + ix) * 4;
    return v3_make(p_from_unorm8(p[0]), p_from_unorm8(p[1]), p_from_unorm8(p[2]));
}
static v3 sample_texture(const jpto_scene_view *sc, float u, float v, int32_t layer)
{
    if (!sc->tex_rgba8 || sc->n_layers <= 0 || sc->tex_res <= 0) return v3_make(0.0f, 0.0f, 0.0f);
    if (layer >= sc->n_layers) layer = sc->n_layers - 1;
    const int32_t res = sc->tex_res;
    const int repeat = sc->sampler_mode & 1, linear = (sc->sampler_mode >> 1) & 1;
    const float x = u * (float)res, y = v * (float)res;
    if (!linear) return texel(sc, layer, tex_index(floorf(x), res, repeat), tex_index(floorf(y), res, repeat));
    const float xs = x - 0.5f, ys = y - 0.5f;
    const float fx = floorf(xs), fy = floorf(ys);
    float a = xs - fx, b = ys - fy;
    if (a != a) a = 0.0f;
    if (b != b) b = 0.0f;
    const int32_t x0 = tex_index(fx, res, repeat), x1 = tex_index(fx + 1.0f, res, repeat);
    const int32_t y0 = tex_index(fy, res, repeat), y1 = tex_index(fy + 1.0f, res, repeat);
    const v3 t00 = texel(sc, layer, x0, y0), t10 = texel(sc, layer, x1, y0);
    const v3 t01 = texel(sc, layer, x0, y1), t11 = texel(sc, layer, x1, y1);
    const v3 r0 = v3_make(p_mix(t00.x, t10.x, a), p_mix(t00.y, t10.y, a), p_mix(t00.z, t10.z, a));
    const v3 r1 = v3_make(p_mix(t01.x, t11.x, a), p_mix(t01.y, t11.y, a), p_mix(t01.z, t11.z, a));
    return v3_make(p_mix(r0.x, r1.x, b), p_mix(r0.y, r1.y, b), p_mix(r0.z, r1.z, b));
}

/* M:194-222 */
static void get_shading_data(ctx_t *cx, const hit_t *h, shading_t *s)
{
    const jpto_scene_view *sc = cx->sc;
    cx->cnt.shaded_hits++;
    const jpto_tri_data *tri = &sc->tri_data[h->triangle];
    const jpto_blas_instance *b = &sc->instances[h->blas];
    /* b.materials[tri.materialIndex] has no bounds check in the reference (M:198): indices past 2 read the words
     * that follow in the blas_instances buffer (the next instance's transform).  Pinned: a read past the END of
     * that buffer returns 0, the robust-buffer-access rule of Vulkan storage buffers. */
    uint32_t mslot = tri->material_index;
    uint64_t word = (uint64_t)h->blas * 44u + 41u + (uint64_t)mslot;   /* 176 B = 44 words; material[] starts at word 41 */
    uint32_t mat_id = word < (uint64_t)sc->n_inst * 44u ? ((const uint32_t *)sc->instances)[word] : 0u;
    if (mat_id >= sc->n_mat) mat_id = 0;
    const jpto_material *material = &sc->materials[mat_id];

    s->position = m4_point(b->transform, h->position);
    s->out_dir = v3_normalize(m4_dir(b->transform, h->out_dir));
    float u = h->bary_u;
    float v = h->bary_v;
    float w0 = 1.0f - u - v;

    float uvx = tri->uvs[0][0] * w0 + tri->uvs[1][0] * u + tri->uvs[2][0] * v;
    float uvy = tri->uvs[0][1] * w0 + tri->uvs[1][1] * u + tri->uvs[2][1] * v;
    v3 n0 = v3_make(tri->n0[0], tri->n0[1], tri->n0[2]);
    v3 n = v3_add(v3_add(v3_scale(n0, w0), v3_scale(v4xyz(tri->n1), u)), v3_scale(v4xyz(tri->n2), v));
    n = v3_normalize(m4_dir(b->transform, n));
    s->normal = h->front ? n : v3_neg(n);

    s->lambert_out = v3_dot(s->normal, s->out_dir);
    float em = p_max(0.0f, material->emission.w);
    s->emission = v3_make(material->emission.x * em, material->emission.y * em, material->emission.z * em);
    v3 albedo = v3_make(material->albedo.x, material->albedo.y, material->albedo.z);
    if (material->albedo_texture_index >= 0)
        albedo = v3_mul(albedo, sample_texture(sc, uvx, uvy, material->albedo_texture_index));

    float metalicity = material->metallic;
    s->fresnel_0 = v3_make(p_mix(0.02f, albedo.x, metalicity), p_mix(0.02f, albedo.y, metalicity),
                           p_mix(0.02f, albedo.z, metalicity));
    s->diffuse_albedo = v3_sub(albedo, v3_scale(albedo, metalicity));
    s->roughness = p_max(0.006f, material->roughness);
}

/* M:224-257 */
static int intersect_triangle(ctx_t *cx, const ray_t *ray, uint32_t tri_index, hit_t *hit)
{
    hit->steps++;
    cx->cnt.tri_tests++;
    const jpto_tri_geometry *tri = &cx->sc->tri_geom[tri_index];
    v3 v0 = v4xyz(tri->vertices[0]);
    v3 v1 = v4xyz(tri->vertices[1]);
    v3 v2 = v4xyz(tri->vertices[2]);

    v3 edge1 = v3_sub(v1, v0);
    v3 edge2 = v3_sub(v2, v0);

    v3 pvec = v3_cross(ray->d, edge2);
    float det = v3_dot(edge1, pvec);

    if (p_abs(det) < 1e-5f) return 0;
    float invDet = 1.0f / det;
    v3 tvec = v3_sub(ray->o, v0);
    float u = v3_dot(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 qvec = v3_cross(tvec, edge1);
    float v = v3_dot(ray->d, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return 0;

    float t = v3_dot(edge2, qvec) * invDet;
    if (t < 0.0f || t > hit->t) return 0;

    hit->position = v3_add(ray->o, v3_scale(ray->d, t));
    hit->t = t;
    hit->triangle = tri_index;
    hit->bary_u = u;
    hit->bary_v = v;
    hit->out_dir = v3_neg(ray->d);
    v3 geometricNormal = v3_cross(edge1, edge2);
    hit->front = (v3_dot(geometricNormal, ray->d) > 0.0f);
    return 1;
}

/* M:259-268 */
static float intersect_aabb(const ray_t *ray, v3 bmin, v3 bmax)
{
    float tx1 = (bmin.x - ray->o.x) * ray->rD.x, tx2 = (bmax.x - ray->o.x) * ray->rD.x;
    float tmin = p_min(tx1, tx2), tmax = p_max(tx1, tx2);
    float ty1 = (bmin.y - ray->o.y) * ray->rD.y, ty2 = (bmax.y - ray->o.y) * ray->rD.y;
    tmin = p_max(tmin, p_min(ty1, ty2)), tmax = p_min(tmax, p_max(ty1, ty2));
    float tz1 = (bmin.z - ray->o.z) * ray->rD.z, tz2 = (bmax.z - ray->o.z) * ray->rD.z;
    tmin = p_max(tmin, p_min(tz1, tz2)), tmax = p_min(tmax, p_max(tz1, tz2));
    if (tmax >= tmin && tmax > 0.0f) return tmin; else return 1e30f;
}

#define STACK_MAX 64
#define PUSH(stack, sp, v) do { if ((sp) < STACK_MAX) (stack)[(sp)++] = (v); else cx->cnt.stack_overflow++; } while (0)

/* M:270-303 */
static int ray_trace_blas(ctx_t *cx, uint32_t root, const ray_t *ray, hit_t *hit)
{
    const jpto_bvh_node *bvh = cx->sc->bvh_nodes;
    const int no_cull = (cx->flags & JPTO_FLAG_NO_CULL) != 0;
    const int reach_only = (cx->flags & JPTO_FLAG_REACH_ONLY) != 0;
    uint32_t stack[STACK_MAX];
    uint32_t sp = 0;
    stack[sp++] = root;

    while (sp > 0) {
        const jpto_bvh_node *node = &bvh[stack[--sp]];

        if (node->tri_count > 0) { /* leaf */
            for (uint32_t i = 0; i < node->tri_count; i++)
                intersect_triangle(cx, ray, node->first_tri_index + i, hit);
            continue;
        }
        cx->cnt.blas_expand++;
        const jpto_bvh_node *childL = &bvh[node->left_child];
        const jpto_bvh_node *childR = &bvh[node->right_child];
        float d1 = intersect_aabb(ray, v4xyz(childL->aabbMin), v4xyz(childL->aabbMax));
        float d2 = intersect_aabb(ray, v4xyz(childR->aabbMin), v4xyz(childR->aabbMax));
        int leftValid = no_cull || (d1 < hit->t);
        int rightValid = no_cull || (d2 < hit->t);
        if (reach_only) { /* see JPTO_FLAG_REACH_ONLY: only a LEAF's own box decides */
            leftValid = childL->tri_count > 0 ? d1 < 1e30f : 1;
            rightValid = childR->tri_count > 0 ? d2 < 1e30f : 1;
        }

        if (d1 < d2) {
            if (rightValid) PUSH(stack, sp, node->right_child);
            if (leftValid) PUSH(stack, sp, node->left_child);
        } else {
            if (leftValid) PUSH(stack, sp, node->left_child);
            if (rightValid) PUSH(stack, sp, node->right_child);
        }
    }
    return hit->t < 1e9f;
}

/* M:305-350 */
static int ray_trace_tlas(ctx_t *cx, const ray_t *ray, hit_t *hit)
{
    const jpto_tlas_node *tlas = cx->sc->tlas_nodes;
    const int no_cull = (cx->flags & JPTO_FLAG_NO_CULL) != 0;
    const int reach_only = (cx->flags & JPTO_FLAG_REACH_ONLY) != 0;
    uint32_t stack[STACK_MAX];
    uint32_t sp = 0;
    stack[sp++] = 0;
    float minT = 1e9f;

    while (sp > 0) {
        const jpto_tlas_node *node = &tlas[stack[--sp]];

        if (node->leftRight == 0) {
            cx->cnt.inst_visits++;
            const jpto_blas_instance *b = &cx->sc->instances[node->blas];
            ray_t b_ray;
            b_ray.o = m4_point(b->inverse_transform, ray->o);
            b_ray.d = m4_dir(b->inverse_transform, ray->d);
            b_ray.rD = v3_make(1.0f / b_ray.d.x, 1.0f / b_ray.d.y, 1.0f / b_ray.d.z);
            ray_trace_blas(cx, b->blas_index, &b_ray, hit);

            if (hit->t < minT) {
                hit->blas = node->blas;
                minT = hit->t;
            }
            continue;
        }
        cx->cnt.tlas_expand++;
        uint32_t left = node->leftRight & 0xFFFF;
        uint32_t right = node->leftRight >> 16;
        const jpto_tlas_node *childL = &tlas[left];
        const jpto_tlas_node *childR = &tlas[right];
        float d1 = intersect_aabb(ray, v3_make(childL->aabbMin[0], childL->aabbMin[1], childL->aabbMin[2]),
                                  v3_make(childL->aabbMax[0], childL->aabbMax[1], childL->aabbMax[2]));
        float d2 = intersect_aabb(ray, v3_make(childR->aabbMin[0], childR->aabbMin[1], childR->aabbMin[2]),
                                  v3_make(childR->aabbMax[0], childR->aabbMax[1], childR->aabbMax[2]));
        int leftValid = no_cull || (d1 < hit->t);
        int rightValid = no_cull || (d2 < hit->t);
        if (reach_only) {
            leftValid = childL->leftRight == 0 ? d1 < 1e30f : 1;
            rightValid = childR->leftRight == 0 ? d2 < 1e30f : 1;
        }

        if (d1 < d2) {
            if (rightValid) PUSH(stack, sp, right);
            if (leftValid) PUSH(stack, sp, left);
        } else {
            if (leftValid) PUSH(stack, sp, left);
            if (rightValid) PUSH(stack, sp, right);
        }
    }
    return hit->t < 1e9f;
}

/* M:352-370 */
static int ray_trace(ctx_t *cx, const ray_t *ray, shading_t *s)
{
    hit_t hit;
    memset(&hit, 0, sizeof hit);
    hit.t = 1e9f;
    hit.steps = 0;
    cx->cnt.rays++;
    int is_hit = (cx->sc->n_tlas > 0 && cx->sc->n_inst > 0) ? ray_trace_tlas(cx, ray, &hit) : 0;
    if (cx->flags & JPTO_FLAG_DEBUG_STEPS) {   /* #ifdef DEBUG_STEPS (M:358-361) */
        float g = p_clamp((float)hit.steps / 256.0f, 0.0f, 1.0f);
        s->emission = v3_make(g, g, g);
        return 0;
    }
    if (is_hit) {
        get_shading_data(cx, &hit, s);
        return 1;
    } else {
        s->emission = sample_sky(ray->d);
        return 0;
    }
}

/* M:372-401; the literal 5 is max_bounces + 1 */
static v3 path_trace(ctx_t *cx, ray_t ray, uint32_t seed[2], float *depth, float cam_far, int32_t max_bounces)
{
    *depth = cam_far;
    v3 radiance = v3_make(0.0f, 0.0f, 0.0f);
    v3 throughput = v3_make(1.0f, 1.0f, 1.0f);
    for (int i = 0; i < max_bounces + 1; i++) {
        shading_t s;
        int is_hit = ray_trace(cx, &ray, &s);
        radiance = v3_add(radiance, v3_mul(throughput, s.emission));
        if (is_hit) {
            if (i == 0) *depth = v3_length(v3_sub(s.position, ray.o));

            ray.o = v3_add(s.position, v3_scale(s.normal, 0.001f));
            float xi[2];
            pcg2d(seed, xi);
            ray.d = sample_brdf(&s, xi);
            ray.rD = v3_make(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);

            float density = get_brdf_density(&s, ray.d);
            float lambert_in = v3_dot(s.normal, ray.d);
            if (lambert_in <= 0.0f) break;

            v3 f = v3_divs(v3_scale(brdf(&s, ray.d), lambert_in), density);
            throughput = v3_mul(throughput, f);
        } else {
            break;
        }
    }
    return radiance;
}

/* M:405-421 */
static void primary_ray(const jpto_camera *cam, int32_t width, int32_t height, int32_t px, int32_t py, ray_t *ray,
                        uint32_t seed[2])
{
    prng_seed((uint32_t)px, (uint32_t)py, cam->frame_index, seed);
    float r[2], jitter[2];
    pcg2d(seed, r);
    r[0] = r[0] * 0.25f;
    r[1] = r[1] * 0.25f;
    box_muller(r, jitter);
    float sx = ((float)px + jitter[0]) / (float)width * 2.0f - 1.0f;
    float sy = ((float)py + jitter[1]) / (float)height * 2.0f - 1.0f;
    float nx = sx, ny = -sy;
    const float *m = cam->ivp;
    float wx = m[0] * nx + m[4] * ny + m[8] + m[12];
    float wy = m[1] * nx + m[5] * ny + m[9] + m[13];
    float wz = m[2] * nx + m[6] * ny + m[10] + m[14];
    float ww = m[3] * nx + m[7] * ny + m[11] + m[15];
    wx = wx / ww;
    wy = wy / ww;
    wz = wz / ww;
    v3 cam_pos = v3_make(cam->position.x, cam->position.y, cam->position.z);
    ray->o = cam_pos;
    ray->d = v3_normalize(v3_sub(v3_make(wx, wy, wz), cam_pos));
    ray->rD = v3_make(1.0f / ray->d.x, 1.0f / ray->d.y, 1.0f / ray->d.z);
}

static void add_counters(jpto_counters *dst, const jpto_counters *src)
{
    dst->rays += src->rays;
    dst->blas_expand += src->blas_expand;
    dst->tri_tests += src->tri_tests;
    dst->tlas_expand += src->tlas_expand;
    dst->inst_visits += src->inst_visits;
    dst->shaded_hits += src->shaded_hits;
    dst->stack_overflow += src->stack_overflow;
}

/* M:404-436 over rows [y0,y1); outputs are indexed relative to row `row_base` */
static void trace_rows(const jpto_scene_view *scene, const jpto_camera *camera, int32_t width, int32_t height,
                       int32_t max_bounces, uint32_t flags, int32_t y0, int32_t y1, int32_t row_base,
                       float *radiance_rgba, float *depth_out, jpto_counters *counters)
{
    ctx_t cx;
    memset(&cx, 0, sizeof cx);
    cx.sc = scene;
    cx.flags = flags;
    for (int32_t py = y0; py < y1; py++) {
        for (int32_t px = 0; px < width; px++) {
            ray_t ray;
            uint32_t seed[2];
            primary_ray(camera, width, height, px, py, &ray, seed);
            float depth = camera->far_;
            v3 radiance;
            if (flags & JPTO_FLAG_DEBUG_STEPS) {   /* M:423-427: one ray_trace, its "emission" is the image; depth stays far */
                shading_t s;
                (void)ray_trace(&cx, &ray, &s);
                radiance = s.emission;
            } else {
                radiance = path_trace(&cx, ray, seed, &depth, camera->far_, max_bounces);
            }
            depth = camera->far_ / (camera->far_ - camera->near_) * (1.0f - camera->near_ / depth);
            size_t idx = (size_t)(py - row_base) * width + px;
            if (radiance_rgba) {
                radiance_rgba[idx * 4 + 0] = radiance.x;
                radiance_rgba[idx * 4 + 1] = radiance.y;
                radiance_rgba[idx * 4 + 2] = radiance.z;
                radiance_rgba[idx * 4 + 3] = 1.0f;
            }
            if (depth_out) depth_out[idx] = depth;
        }
    }
    if (counters) add_counters(counters, &cx.cnt);
}

void jpto_trace_frame(const jpto_scene_view *scene, const jpto_camera *camera, int32_t width, int32_t height,
                      int32_t max_bounces, uint32_t flags, int32_t y0, int32_t y1, float *radiance_rgba, float *depth_out,
                      jpto_counters *counters)
{
    trace_rows(scene, camera, width, height, max_bounces, flags, y0, y1, 0, radiance_rgba, depth_out, counters);
}

/* P:19-26 */
static v3 aces_film(v3 x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float in[3] = {x.x, x.y, x.z}, out[3];
    for (int k = 0; k < 3; k++) {
        float v = in[k];
        out[k] = p_clamp((v * (a * v + b)) / (v * (c * v + d) + e), 0.0f, 1.0f);
    }
    return v3_make(out[0], out[1], out[2]);
}

typedef struct {
    const jpto_scene_view *scene;
    jpto_camera camera;
    int32_t width, height, max_bounces, n_frames, accum_mode;
    uint32_t first_frame_index, flags;
    int32_t y0, y1;
    float *accum;
    uint8_t *ldr;
    float *depth;
    jpto_counters cnt;
} job_t;

/* One thread: all frames of rows [y0,y1).  Per pixel the frame order is sequential, as in the
 * reference (one dispatch of main + one of progressive_rendering per frame). */
static void *job_main(void *arg)
{
    job_t *jb = (job_t *)arg;
    int32_t rows = jb->y1 - jb->y0;
    if (rows <= 0) return NULL;
    size_t npx = (size_t)rows * jb->width;
    float *rad = (float *)malloc(npx * 4 * sizeof(float));
    float *dep = (float *)malloc(npx * sizeof(float));
    for (int32_t f = 0; f < jb->n_frames; f++) {
        jpto_camera cam = jb->camera;
        cam.frame_index = jb->first_frame_index + (uint32_t)f; /* path_tracing_camera.cpp:199 */
        uint32_t frame_count = (uint32_t)f + 1;                /* progressive_rendering.cpp:53-60 */
        trace_rows(jb->scene, &cam, jb->width, jb->height, jb->max_bounces, jb->flags, jb->y0, jb->y1, jb->y0, rad, dep,
                   &jb->cnt);
        for (size_t i = 0; i < npx; i++) {
            size_t g = (size_t)jb->y0 * jb->width + i;
            float cur[3];
            if (jb->accum_mode == JPTO_ACCUM_REF_LDR8) {
                /* imageStore(outputImage rgba8) (M:434) then imageLoad(screenTexture) (P:33) */
                for (int k = 0; k < 3; k++) cur[k] = p_from_unorm8(p_unorm8(rad[i * 4 + k]));
            } else {
                for (int k = 0; k < 3; k++) cur[k] = rad[i * 4 + k];
            }
            float sum[3];
            for (int k = 0; k < 3; k++) {
                sum[k] = cur[k];
                if (frame_count > 1) sum[k] = sum[k] + jb->accum[g * 4 + k]; /* P:34-36 */
                jb->accum[g * 4 + k] = sum[k];                               /* P:37 */
            }
            jb->accum[g * 4 + 3] = 1.0f;
            if (jb->ldr) {
                float fc = (float)frame_count;
                v3 avg = v3_make(sum[0] / fc, sum[1] / fc, sum[2] / fc); /* P:39 */
                v3 col = aces_film(v3_scale(avg, 1.0f));                 /* P:41-43 */
                jb->ldr[g * 4 + 0] = p_unorm8(col.x);
                jb->ldr[g * 4 + 1] = p_unorm8(col.y);
                jb->ldr[g * 4 + 2] = p_unorm8(col.z);
                jb->ldr[g * 4 + 3] = 255;
            }
            if (jb->depth) jb->depth[g] = dep[i];
        }
    }
    free(rad);
    free(dep);
    return NULL;
}

typedef struct { job_t *jobs; int32_t first, step, n; volatile int32_t *next; } worker_t;
static void *worker_main(void *arg)
{
    worker_t *w = (worker_t *)arg;
    /* strips are taken from a shared counter: sky strips cost ~100x less than strips through the box */
    for (;;) {
        int32_t s = __sync_fetch_and_add(w->next, 1);
        if (s >= w->n) break;
        job_main(&w->jobs[s]);
    }
    return NULL;
}

int32_t jpto_render(const jpto_scene_view *scene, const jpto_camera *camera, int32_t width, int32_t height,
                    int32_t max_bounces, int32_t n_frames, uint32_t first_frame_index, int32_t accum_mode, uint32_t flags,
                    int32_t n_threads, float *accum_rgba, uint8_t *ldr_rgba8, float *depth, jpto_counters *counters)
{
    if (n_threads <= 0) {
        long n = sysconf(_SC_NPROCESSORS_ONLN);
        n_threads = n > 0 ? (int32_t)n : 1;
    }
    /* rows are dealt in small interleaved strips so threads get similar work */
    const int32_t strip = 2;
    int32_t n_strips = (height + strip - 1) / strip;
    if (n_threads > n_strips) n_threads = n_strips > 0 ? n_strips : 1;
    job_t *jobs = (job_t *)calloc((size_t)n_strips, sizeof(job_t));
    for (int32_t s = 0; s < n_strips; s++) {
        job_t *jb = &jobs[s];
        jb->scene = scene;
        jb->camera = *camera;
        jb->width = width;
        jb->height = height;
        jb->max_bounces = max_bounces;
        jb->n_frames = n_frames;
        jb->accum_mode = accum_mode;
        jb->first_frame_index = first_frame_index;
        jb->flags = flags;
        jb->y0 = s * strip;
        jb->y1 = (s + 1) * strip < height ? (s + 1) * strip : height;
        jb->accum = accum_rgba;
        jb->ldr = ldr_rgba8;
        jb->depth = depth;
    }
    /* worker w handles strips w, w+T, w+2T, ... */
    worker_t *ws = (worker_t *)calloc((size_t)n_threads, sizeof(worker_t));
    pthread_t *tids = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    volatile int32_t next_strip = 0;
    for (int32_t w = 0; w < n_threads; w++) {
        ws[w].next = &next_strip;
        ws[w].jobs = jobs;
        ws[w].first = w;
        ws[w].step = n_threads;
        ws[w].n = n_strips;
        if (n_threads == 1) worker_main(&ws[w]);
        else pthread_create(&tids[w], NULL, worker_main, &ws[w]);
    }
    if (n_threads > 1)
        for (int32_t w = 0; w < n_threads; w++) pthread_join(tids[w], NULL);
    if (counters)
        for (int32_t s = 0; s < n_strips; s++) add_counters(counters, &jobs[s].cnt);
    free(ws);
    free(tids);
    free(jobs);
    return n_threads;
}


/* ------------------------------------------------------------------ KAT entry points */

void jpto_prng_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t seed_out[2]) { prng_seed(px, py, frame, seed_out); }
void jpto_pcg2d(uint32_t seed[2], float out[2]) { pcg2d(seed, out); }
void jpto_sincos(float x, float *s, float *c) { p_sincos(x, s, c); }

float jpto_intersect_aabb(const float o[3], const float rD[3], const float bmin[3], const float bmax[3])
{
    ray_t r;
    r.o = v3_make(o[0], o[1], o[2]);
    r.rD = v3_make(rD[0], rD[1], rD[2]);
    r.d = v3_make(0, 0, 0);
    return intersect_aabb(&r, v3_make(bmin[0], bmin[1], bmin[2]), v3_make(bmax[0], bmax[1], bmax[2]));
}

int jpto_intersect_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3],
                            float t_max, float out_tuv[3], int *front)
{
    jpto_tri_geometry g;
    memset(&g, 0, sizeof g);
    const float *vs[3] = {v0, v1, v2};
    for (int k = 0; k < 3; k++) {
        g.vertices[k].x = vs[k][0];
        g.vertices[k].y = vs[k][1];
        g.vertices[k].z = vs[k][2];
        g.vertices[k].w = 1.0f;
    }
    jpto_scene_view sc;
    memset(&sc, 0, sizeof sc);
    sc.tri_geom = &g;
    sc.n_tri = 1;
    ctx_t cx;
    memset(&cx, 0, sizeof cx);
    cx.sc = &sc;
    ray_t r;
    r.o = v3_make(o[0], o[1], o[2]);
    r.d = v3_make(d[0], d[1], d[2]);
    r.rD = v3_make(1.0f / d[0], 1.0f / d[1], 1.0f / d[2]);
    hit_t h;
    memset(&h, 0, sizeof h);
    h.t = t_max;
    int ok = intersect_triangle(&cx, &r, 0, &h);
    if (ok) {
        out_tuv[0] = h.t;
        out_tuv[1] = h.bary_u;
        out_tuv[2] = h.bary_v;
        if (front) *front = h.front;
    }
    return ok;
}

void jpto_primary_ray(const jpto_camera *camera, int32_t width, int32_t height, int32_t px, int32_t py, float o[3],
                      float d[3], uint32_t seed_after[2])
{
    ray_t r;
    uint32_t seed[2];
    primary_ray(camera, width, height, px, py, &r, seed);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
    d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
    seed_after[0] = seed[0];
    seed_after[1] = seed[1];
}

static void to_shading(const jpto_shading *in, shading_t *s)
{
    memset(s, 0, sizeof *s);
    s->normal = v3_make(in->normal[0], in->normal[1], in->normal[2]);
    s->out_dir = v3_make(in->out_dir[0], in->out_dir[1], in->out_dir[2]);
    s->lambert_out = in->lambert_out;
    s->diffuse_albedo = v3_make(in->diffuse_albedo[0], in->diffuse_albedo[1], in->diffuse_albedo[2]);
    s->fresnel_0 = v3_make(in->fresnel_0[0], in->fresnel_0[1], in->fresnel_0[2]);
    s->roughness = in->roughness;
}

void jpto_brdf(const jpto_shading *in, const float l[3], float out[3])
{
    shading_t s;
    to_shading(in, &s);
    v3 r = brdf(&s, v3_make(l[0], l[1], l[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void jpto_sample_brdf(const jpto_shading *in, const float xi[2], float out[3])
{
    shading_t s;
    to_shading(in, &s);
    v3 r = sample_brdf(&s, xi);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

float jpto_brdf_density(const jpto_shading *in, const float l[3])
{
    shading_t s;
    to_shading(in, &s);
    return get_brdf_density(&s, v3_make(l[0], l[1], l[2]));
}

uint8_t jpto_unorm8(float x) { return p_unorm8(x); }

void jpto_sample_texture(const jpto_scene_view *scene, float u, float v, int32_t layer, float out[3])
{
    v3 t = sample_texture(scene, u, v, layer);
    out[0] = t.x; out[1] = t.y; out[2] = t.z;
}

void jpto_aces(const float in[3], float out[3])
{
    v3 r = aces_film(v3_make(in[0], in[1], in[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
