/*
 * oracle_pins.h -- pinned float32 semantics for the constructs GLSL leaves
 * implementation-defined.  TEST INFRASTRUCTURE ONLY.
 *
 * Compile with -ffp-contract=off and without -ffast-math: every + - * / sqrt
 * below is one IEEE-754 binary32 round-to-nearest-even operation, in the
 * order written.  The HIP kernels implement the same definitions
 * independently (gdpathtracing_amd/csrc/jpt_device_math.h); DESIGN.md lists
 * them.
 */
#ifndef JPT_ORACLE_PINS_H
#define JPT_ORACLE_PINS_H

#include <math.h>
#include <stdint.h>

typedef struct { float x, y, z; } v3;

/* min/max: IEEE-754-2008 minNum/maxNum (a NaN operand is ignored) -- what
 * GLSL min()/max() lower to on AMD hardware (v_min_f32/v_max_f32). */
static inline float p_min(float a, float b) { if (a != a) return b; if (b != b) return a; return (b < a) ? b : a; }
static inline float p_max(float a, float b) { if (a != a) return b; if (b != b) return a; return (a < b) ? b : a; }
static inline float p_clamp(float x, float lo, float hi) { return p_min(p_max(x, lo), hi); }
static inline float p_abs(float x) { return fabsf(x); }
/* mix(a,b,t) = a*(1-t) + b*t */
static inline float p_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_divs(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* dot: left to right, no FMA */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* normalize(v) = v * (1 / sqrt(dot(v,v))) */
static inline v3 v3_normalize(v3 a)
{
    float inv = 1.0f / sqrtf(v3_dot(a, a));
    return v3_scale(a, inv);
}
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }

/* mat4 * (v,1) and mat4 * (v,0), column-major float[16] (utils.h:15-37):
 * c0*x + c1*y + c2*z (+ c3), summed left to right.  The w = 0 form omits the
 * c3*0 term. */
static inline v3 m4_point(const float *m, v3 p)
{
    return v3_make(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                   m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                   m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
static inline v3 m4_dir(const float *m, v3 d)
{
    return v3_make(m[0] * d.x + m[4] * d.y + m[8] * d.z,
                   m[1] * d.x + m[5] * d.y + m[9] * d.z,
                   m[2] * d.x + m[6] * d.y + m[10] * d.z);
}

/*
 * sin/cos: one shared routine, arguments here are in [0, 2*pi].
 * Cephes-style: j = floor(x * 4/pi), made even; three-part Cody-Waite
 * reduction by pi/4; degree-7/8 polynomials on [-pi/4, pi/4].
 */
static inline void p_sincos(float x, float *s_out, float *c_out)
{
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    float ax = fabsf(x);
    float y = floorf(ax * FOPI);
    int32_t j = (int32_t)y;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    float r = ((ax - y * DP1) - y * DP2) - y * DP3;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    float s, c;
    switch (j) {
    case 0: s = ps;  c = pc;  break;
    case 2: s = pc;  c = -ps; break;
    case 4: s = -ps; c = -pc; break;
    default: s = -pc; c = ps; break; /* 6 */
    }
    if (x < 0.0f) s = -s;
    *s_out = s;
    *c_out = c;
}

/* imageStore to rgba8: floor(clamp01(x)*255 + 0.5); NaN -> 0 via maxNum */
static inline uint8_t p_unorm8(float x)
{
    float c = p_min(p_max(x, 0.0f), 1.0f);
    return (uint8_t)floorf(c * 255.0f + 0.5f);
}
/* ivec2(vec2): truncation toward zero; out of range saturates and NaN gives 0 (GLSL leaves both undefined;
 * this is what v_cvt_i32_f32 does) */
static inline int32_t p_f2i(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
/* imageLoad from rgba8 */
static inline float p_from_unorm8(uint8_t q) { return (float)q / 255.0f; }

#endif
