// jpt_device_math.h -- float32 building blocks of the gfx950 kernels.
//
// Every + - * / sqrt is one IEEE-754 binary32 operation in the order written: the library is compiled
// with -ffp-contract=off, hipcc's default correctly-rounded fp32 divide/sqrt, and denormals on.  The
// definitions of the constructs GLSL leaves open (normalize, mix, min/max with NaN, mat*vec order,
// sin/cos, UNORM8 conversion) are listed in DESIGN.md "Pinned semantics"; the CPU oracle implements
// the same definitions independently.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jpt {

struct f3 {
    float x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }

// minNum / maxNum: a NaN operand is ignored (v_min_f32 / v_max_f32).
__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
__device__ __forceinline__ float mix_(float a, float b, float t) { return a * (1.0f - t) + b * t; }

__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ f3 normalize3(f3 a)
{
    float inv = 1.0f / __builtin_sqrtf(dot3(a, a));
    return a * inv;
}
__device__ __forceinline__ float length3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 rcp3(f3 d) { return f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z}; }

// column-major 4x4 (utils.h:15-37) times (p,1) / (d,0); c0*x + c1*y + c2*z (+ c3), left to right
__device__ __forceinline__ f3 xform_point(const float* __restrict__ m, f3 p)
{
    return f3{m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
__device__ __forceinline__ f3 xform_dir(const float* __restrict__ m, f3 d)
{
    return f3{m[0] * d.x + m[4] * d.y + m[8] * d.z, m[1] * d.x + m[5] * d.y + m[9] * d.z,
              m[2] * d.x + m[6] * d.y + m[10] * d.z};
}

// sin and cos of x (|x| small, here [0, 2*pi]): octant j = floor(|x| * 4/pi) rounded up to even,
// three-constant Cody-Waite reduction by pi/4, single-precision minimax polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void sincos_(float x, float& s_out, float& c_out)
{
    const float ax = __builtin_fabsf(x);
    float y = __builtin_floorf(ax * 1.27323954473516f);
    int j = (int)y;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    j &= 7;
    const float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc =
        ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    float s = (j == 0) ? ps : (j == 2) ? pc : (j == 4) ? -ps : -pc;
    float c = (j == 0) ? pc : (j == 2) ? -ps : (j == 4) ? -pc : ps;
    if (x < 0.0f) s = -s;
    s_out = s;
    c_out = c;
}

// rgba8 UNORM store / load
__device__ __forceinline__ uint32_t unorm8(float x)
{
    return (uint32_t)__builtin_floorf(clamp_(x, 0.0f, 1.0f) * 255.0f + 0.5f);
}
// q / 255.0f, correctly rounded, for q in 0..255 (rgba8 imageLoad, texel decode) without the ~11-instruction division
// sequence: y = q * c with c = RN(1 / 255), one residual and one correction step, three 2-cycle instructions.  Equal to
// the IEEE quotient for ALL 256 inputs -- checked exhaustively in exact rational arithmetic
// (tests/test_oracle_kats.py::test_unorm8_decode_without_division_is_exact); other inputs do not occur.
__device__ __forceinline__ float from_unorm8(uint32_t q)
{
    const float x = (float)q, c = 1.0f / 255.0f;
    const float y = x * c;
    const float r = __builtin_fmaf(-255.0f, y, x);
    return __builtin_fmaf(r, c, y);
}
// ivec2(vec2): truncation; out of range saturates, NaN gives 0 (pinned; GLSL leaves both undefined)
__device__ __forceinline__ int32_t f2i_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)f;
}

}  // namespace jpt
