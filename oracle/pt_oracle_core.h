// pt_oracle_core.h — CPU parity oracle, part 1: vector math, samplers, textures, volumes and the
// per-mesh BVH builder.  Part 2 (traversal, shading programs, frame loop, C interface) is
// pt_oracle.cc, the only file that includes this header.
//
// TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  PARITY UNPINNED: the reference holds no golden
// vectors for this path and cannot be built here; this file restates its algorithm from the
// sources, citing the file:line each function follows (paths relative to the reference root).
//
// Build: g++ -O2 -ffp-contract=off (no fast-math): every float operation below is meant to be
// evaluated as written, one IEEE-754 binary32 rounding per operation.
//
// glm is not available; its functions are restated from their published definitions:
//   dot(vec3)        = (x*x' + y*y') + z*z'            mat4*vec4 = (c0*x + c1*y) + (c2*z + c3*w)
//   normalize(v)     = v * (1/sqrt(dot(v,v)))          mat3*vec3 = (c0*x + c1*y) + c2*z
//   min(a,b)         = (b<a)?b:a      max(a,b) = (a<b)?b:a      clamp = min(max(x,lo),hi)
//   mix(a,b,t)       = a*(1-t) + b*t
//   reflect(I,N)     = I - N*dot(N,I)*2
//   refract(I,N,eta) : k = 1 - eta*eta*(1 - d*d), d = dot(N,I); k<0 ? 0 : eta*I - (eta*d + sqrt(k))*N

#pragma once
#include "pt_oracle.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <thread>
#include <vector>

namespace orc {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr float kPi = 3.14159265358979323846264338327950288f;

// ------------------------------------------------------------------------------------------
// vector math with glm semantics
// ------------------------------------------------------------------------------------------
struct V2 {
    float x = 0, y = 0;
};
struct V3 {
    float x = 0, y = 0, z = 0;
    V3() = default;
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(float s) : x(s), y(s), z(s) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V4 {
    float x = 0, y = 0, z = 0, w = 0;
    V4() = default;
    V4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    V4(V3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
};
struct M3 {
    V3 c[3];
};
struct M4 {
    V4 c[4];
};

inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator*(V2 a, float s) { return {a.x * s, a.y * s}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(float s, V3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline bool operator!=(V3 a, V3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator*(V4 a, V4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V3 xyz(V4 v) { return {v.x, v.y, v.z}; }

inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float gmin(float a, float b) { return (b < a) ? b : a; }
inline float gmax(float a, float b) { return (a < b) ? b : a; }
inline V3 gmin(V3 a, V3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline V3 gmax(V3 a, V3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }
inline float clampf(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline V3 mix(V3 a, V3 b, float t) { return {mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)}; }
inline V4 mix(V4 a, V4 b, float t) {
    return {mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t), mixf(a.w, b.w, t)};
}
inline V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }
inline V3 refract(V3 I, V3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k >= 0.0f) return eta * I - (eta * d + std::sqrt(k)) * N;
    return V3(0.0f);
}
inline V4 mul(const M4& m, V4 v) {
    V4 a = m.c[0] * v.x, b = m.c[1] * v.y, c = m.c[2] * v.z, d = m.c[3] * v.w;
    return (a + b) + (c + d);
}
inline V3 mul(const M3& m, V3 v) {
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z, m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
inline M4 mul(const M4& a, const M4& b) {  // glm mat4*mat4: left-to-right sum of four column products
    M4 r;
    for (int j = 0; j < 4; ++j) {
        const V4 bj = b.c[j];
        V4 t = a.c[0] * bj.x;
        t = t + a.c[1] * bj.y;
        t = t + a.c[2] * bj.z;
        t = t + a.c[3] * bj.w;
        r.c[j] = t;
    }
    return r;
}
inline M4 transpose(const M4& m) {
    M4 r;
    const float* s = &m.c[0].x;
    float* d = &r.c[0].x;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) d[i * 4 + j] = s[j * 4 + i];
    return r;
}
inline M4 load_m4(const float* p) {
    M4 m;
    std::memcpy(static_cast<void*>(&m), p, sizeof(float) * 16);
    return m;
}
// glm::inverse(mat4) — cofactor expansion in glm's operation order.
inline M4 inverse(const M4& mm) {
    auto m = [&](int c, int r) { return (&mm.c[c].x)[r]; };
    float c00 = m(2, 2) * m(3, 3) - m(3, 2) * m(2, 3), c02 = m(1, 2) * m(3, 3) - m(3, 2) * m(1, 3),
          c03 = m(1, 2) * m(2, 3) - m(2, 2) * m(1, 3);
    float c04 = m(2, 1) * m(3, 3) - m(3, 1) * m(2, 3), c06 = m(1, 1) * m(3, 3) - m(3, 1) * m(1, 3),
          c07 = m(1, 1) * m(2, 3) - m(2, 1) * m(1, 3);
    float c08 = m(2, 1) * m(3, 2) - m(3, 1) * m(2, 2), c10 = m(1, 1) * m(3, 2) - m(3, 1) * m(1, 2),
          c11 = m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2);
    float c12 = m(2, 0) * m(3, 3) - m(3, 0) * m(2, 3), c14 = m(1, 0) * m(3, 3) - m(3, 0) * m(1, 3),
          c15 = m(1, 0) * m(2, 3) - m(2, 0) * m(1, 3);
    float c16 = m(2, 0) * m(3, 2) - m(3, 0) * m(2, 2), c18 = m(1, 0) * m(3, 2) - m(3, 0) * m(1, 2),
          c19 = m(1, 0) * m(2, 2) - m(2, 0) * m(1, 2);
    float c20 = m(2, 0) * m(3, 1) - m(3, 0) * m(2, 1), c22 = m(1, 0) * m(3, 1) - m(3, 0) * m(1, 1),
          c23 = m(1, 0) * m(2, 1) - m(2, 0) * m(1, 1);
    V4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11), f3(c12, c12, c14, c15),
        f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    V4 v0(m(1, 0), m(0, 0), m(0, 0), m(0, 0)), v1(m(1, 1), m(0, 1), m(0, 1), m(0, 1)),
        v2(m(1, 2), m(0, 2), m(0, 2), m(0, 2)), v3(m(1, 3), m(0, 3), m(0, 3), m(0, 3));
    auto sub = [](V4 a, V4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
    V4 i0 = sub(v1 * f0, v2 * f1) + v3 * f2;
    V4 i1 = sub(v0 * f0, v2 * f3) + v3 * f4;
    V4 i2 = sub(v0 * f1, v1 * f3) + v3 * f5;
    V4 i3 = sub(v0 * f2, v1 * f4) + v2 * f5;
    V4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    M4 inv;
    inv.c[0] = i0 * sa;
    inv.c[1] = i1 * sb;
    inv.c[2] = i2 * sa;
    inv.c[3] = i3 * sb;
    V4 row0(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    V4 d0 = mm.c[0] * row0;
    float det = (d0.x + d0.y) + (d0.z + d0.w);
    float ood = 1.0f / det;
    for (auto& col : inv.c) col = col * ood;
    return inv;
}
// glm::lookAt (right-handed) and glm::perspectiveRH with the -1..1 clip depth glm defaults to.
inline M4 look_at_rh(V3 eye, V3 center, V3 up) {
    V3 f = normalize(center - eye);
    V3 s = normalize(cross(f, up));
    V3 u = cross(s, f);
    M4 r;
    r.c[0] = V4(s.x, u.x, -f.x, 0);
    r.c[1] = V4(s.y, u.y, -f.y, 0);
    r.c[2] = V4(s.z, u.z, -f.z, 0);
    r.c[3] = V4(-dot(s, eye), -dot(u, eye), dot(f, eye), 1);
    return r;
}
inline M4 perspective_rh(float fovy, float aspect, float zn, float zf) {
    float th = std::tan(fovy / 2.0f);
    M4 r;  // zero-initialised
    r.c[0].x = 1.0f / (aspect * th);
    r.c[1].y = 1.0f / th;
    r.c[2].z = -(zf + zn) / (zf - zn);
    r.c[2].w = -1.0f;
    r.c[3].z = -(2.0f * zf * zn) / (zf - zn);
    return r;
}

// ------------------------------------------------------------------------------------------
// portable transcendental kernels (math_mode 1) — fixed polynomial evaluation order with explicit
// fused multiply-adds, so a device implementation of the same formulas returns the same bits.
// ------------------------------------------------------------------------------------------
inline float p_fma(float a, float b, float c) { return std::fma(a, b, c); }
inline void portable_sincos(float x, float* s_out, float* c_out) {  // x in [0, 2*pi]
    const float two_over_pi = 0.636619772367581343f;
    const float pio2_hi = 1.57079625129699707031f;   // high bits of pi/2
    const float pio2_lo = 7.54978995489188216e-08f;  // pi/2 - pio2_hi
    float qf = std::nearbyint(x * two_over_pi);
    int q = static_cast<int>(qf);
    float r = p_fma(-qf, pio2_hi, x);
    r = p_fma(-qf, pio2_lo, r);
    float r2 = r * r;
    float sp = p_fma(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = p_fma(sp, r2, -1.6666654611e-1f);
    float s = p_fma(sp * r2, r, r);
    float cp = p_fma(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = p_fma(cp, r2, 4.166664568298827e-2f);
    float c = p_fma(cp * r2, r2, p_fma(-0.5f, r2, 1.0f));
    switch (q & 3) {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}
inline float portable_atan(float x) {  // cephes-style atanf
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = 1.57079632679489661923f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = p_fma(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = p_fma(p, z, 1.99777106478e-1f);
    p = p_fma(p, z, -3.33329491539e-1f);
    y = y + p_fma(p * z, x, x);
    return sign * y;
}
inline float portable_atan2(float y, float x) {
    if (x == 0.0f) {
        if (y > 0.0f) return 1.57079632679489661923f;
        if (y < 0.0f) return -1.57079632679489661923f;
        return 0.0f;
    }
    float a = portable_atan(y / x);
    if (x < 0.0f) return (y < 0.0f) ? a - kPi : a + kPi;
    return a;
}
inline float portable_asin(float x) {  // cephes-style asinf
    float sign = 1.0f, a = x;
    if (x < 0.0f) { sign = -1.0f; a = -x; }
    if (a > 1.0f) a = 1.0f;
    bool big = a > 0.5f;
    float z, w;
    if (big) { z = 0.5f * (1.0f - a); w = std::sqrt(z); }
    else { w = a; z = w * w; }
    float p = p_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = p_fma(p, z, 4.5470025998e-2f);
    p = p_fma(p, z, 7.4953002686e-2f);
    p = p_fma(p, z, 1.6666752422e-1f);
    float r = p_fma(p * z, w, w);
    if (big) r = 1.57079632679489661923f - (r + r);
    return sign * r;
}
inline float portable_exp(float x) {  // cephes-style expf; callers pass x <= 0 in practice
    if (x != x) return x;
    if (x > 88.0f) return kInf;
    if (x < -87.0f) return 0.0f;
    float nf = std::floor(p_fma(x, 1.44269504088896341f, 0.5f));
    int n = static_cast<int>(nf);
    float r = p_fma(nf, -0.693359375f, x);
    r = p_fma(nf, 2.12194440e-4f, r);
    float r2 = r * r;
    float p = p_fma(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = p_fma(p, r, 8.3334519073e-3f);
    p = p_fma(p, r, 4.1665795894e-2f);
    p = p_fma(p, r, 1.6666665459e-1f);
    p = p_fma(p, r, 5.0000001201e-1f);
    float e = p_fma(p, r2, r) + 1.0f;
    uint32_t bits = static_cast<uint32_t>(n + 127) << 23;  // 2^n, n in [-126, 127]
    float scale;
    std::memcpy(&scale, &bits, 4);
    return e * scale;
}
struct Mathx {
    int mode = ORC_MATH_LIBM;
    void sincos(float x, float* s, float* c) const {
        if (mode == ORC_MATH_PORTABLE) portable_sincos(x, s, c);
        else { *c = std::cos(x); *s = std::sin(x); }
    }
    float atan2(float y, float x) const { return mode == ORC_MATH_PORTABLE ? portable_atan2(y, x) : std::atan2(y, x); }
    float asin(float x) const { return mode == ORC_MATH_PORTABLE ? portable_asin(x) : std::asin(x); }
    float exp(float x) const { return mode == ORC_MATH_PORTABLE ? portable_exp(x) : std::exp(x); }
};

// ------------------------------------------------------------------------------------------
// counter-based sampler (same definition as the GPU library)
// ------------------------------------------------------------------------------------------
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
inline uint32_t rng_camera_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = mix32(seed ^ 0x85ebca6bU);
    k = mix32(k + pixel);
    return mix32(k ^ (sample * 0x9e3779b9U + 0x7f4a7c15U));
}
inline float rng_uniform(uint32_t key, uint32_t j) {
    return static_cast<float>(mix32(key + (j + 1u) * 0x9e3779b9U) >> 8) * (1.0f / 16777216.0f);
}
inline uint32_t rng_child(uint32_t key, uint32_t lobe) { return mix32((key ^ 0x68bc21ebU) + (lobe + 1u) * 0x02e5be93U); }
enum { LOBE_PASS = 0, LOBE_DIFFUSE = 1, LOBE_TRANSMIT = 2, LOBE_REFLECT = 3, LOBE_AO0 = 4 };
// 5-bit tag of a triangle (same definition as the GPU library's pass_hash)
inline uint32_t pass_hash(uint32_t object, uint32_t triangle) { return ((triangle * 0x9e3779b9U) ^ (object * 0x85ebca6bU)) >> 27; }

// ------------------------------------------------------------------------------------------
// textures (src/cpu_framebuffer.h:14-59, 90-123)
// ------------------------------------------------------------------------------------------
inline float linear_to_srgb(float c) {  // cpu_framebuffer.h:14-20
    if (c <= 0.0031308f) return 12.92f * c;
    return 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f;
}
inline float srgb_to_linear(float c) {  // cpu_framebuffer.h:21-27
    if (c <= 0.04045f) return c / 12.92f;
    return std::pow((c + 0.055f) / 1.055f, 2.4f);
}
struct Tex8 {
    const uint8_t* px = nullptr;  // RGBA8
    int w = 1, h = 1;
};
struct TexF {
    const float* px = nullptr;  // RGBA32F
    int w = 1, h = 1;
};
inline V4 texel8(const Tex8& t, int x, int y, bool srgb) {  // pixel_to_float / srgb_pixel_to_float, :35-50
    const uint8_t* p = t.px + (static_cast<size_t>(y) * t.w + x) * 4;
    float r = static_cast<float>(p[0]) / 255.0f, g = static_cast<float>(p[1]) / 255.0f,
          b = static_cast<float>(p[2]) / 255.0f, a = static_cast<float>(p[3]) / 255.0f;
    if (srgb) return V4(srgb_to_linear(r), srgb_to_linear(g), srgb_to_linear(b), a);
    return V4(r, g, b, a);
}
inline V4 texelf(const TexF& t, int x, int y) {
    const float* p = t.px + (static_cast<size_t>(y) * t.w + x) * 4;
    return V4(p[0], p[1], p[2], p[3]);
}
struct BilinearTaps {
    int x0, y0, x1, y1;
    float tx, ty;
};
inline BilinearTaps bilinear_taps(V2 uv, int w, int h) {  // cpu_framebuffer.h:91-102
    float xf = uv.x * static_cast<float>(w) - 0.5f;
    float yf = uv.y * static_cast<float>(h) - 0.5f;
    BilinearTaps b;
    b.tx = xf - std::floor(xf);
    b.ty = yf - std::floor(yf);
    int x0 = static_cast<int>(std::floor(xf)), y0 = static_cast<int>(std::floor(yf));
    int x1 = x0 + 1, y1 = y0 + 1;
    b.x0 = ((x0 % w) + w) % w;
    b.y0 = ((y0 % h) + h) % h;
    b.x1 = ((x1 % w) + w) % w;
    b.y1 = ((y1 % h) + h) % h;
    return b;
}
inline V4 sample_bilinear(const Tex8& t, V2 uv, bool srgb) {  // cpu_framebuffer.h:90-123
    BilinearTaps b = bilinear_taps(uv, t.w, t.h);
    V4 c00 = texel8(t, b.x0, b.y0, srgb), c10 = texel8(t, b.x1, b.y0, srgb), c01 = texel8(t, b.x0, b.y1, srgb),
       c11 = texel8(t, b.x1, b.y1, srgb);
    return mix(mix(c00, c10, b.tx), mix(c01, c11, b.tx), b.ty);
}
inline V4 sample_bilinear(const TexF& t, V2 uv) {
    BilinearTaps b = bilinear_taps(uv, t.w, t.h);
    V4 c00 = texelf(t, b.x0, b.y0), c10 = texelf(t, b.x1, b.y0), c01 = texelf(t, b.x0, b.y1),
       c11 = texelf(t, b.x1, b.y1);
    return mix(mix(c00, c10, b.tx), mix(c01, c11, b.tx), b.ty);
}

// ------------------------------------------------------------------------------------------
// scene data (src/model_loader.h:14-65)
// ------------------------------------------------------------------------------------------
struct Vertex {
    V3 position, normal;
    V4 tangent;
    V2 uv;
};
struct Mesh {
    const uint32_t* indices;
    uint32_t n_indices;
    const orc_vertex* vertices;
    uint32_t n_vertices;
    uint32_t material;
};
struct Object {
    M4 model, normal;
    uint32_t mesh;
};
inline Vertex load_vertex(const orc_vertex& v) {
    Vertex r;
    r.position = V3(v.position[0], v.position[1], v.position[2]);
    r.normal = V3(v.normal[0], v.normal[1], v.normal[2]);
    r.tangent = V4(v.tangent[0], v.tangent[1], v.tangent[2], v.tangent[3]);
    r.uv = V2{v.uv[0], v.uv[1]};
    return r;
}
inline V3 vpos(const orc_vertex& v) { return V3(v.position[0], v.position[1], v.position[2]); }

// ------------------------------------------------------------------------------------------
// rays and hits (src/ray_program.h:16-56)
// ------------------------------------------------------------------------------------------
struct Ray {
    V3 origin, direction;
};
struct RayP : Ray {
    V3 payload = V3(1.0f);
    uint8_t depth = 0;
    bool any_hit = false;
    uint32_t key = 0;  // counter-sampler state (the reference has none)
    uint32_t pass_tag = 0;  // pass-through rays: tag of the triangle they passed (the reference has none: see pbr())
};
struct VolumeHit {  // ray_volume_hit_info, ray_program.h:40-48
    bool hit = false;
    float t0 = 0.0f, t1 = 0.0f;
    bool forward() const { return hit && (t1 > 0.0f); }
    float forward_distance() const { return forward() ? (t0 > 0.0f ? t0 : 0.0f) : kInf; }
};
struct TriHit {  // barycentric_coords + ray_triangle_hit_info, ray_program.h:29-38, 50-56
    float A = 0.0f, B = 0.0f, t = kInf;
    bool hit = false, backface = false;
    uint32_t object = 0, mesh = 0, triangle = 0;
    float C() const { return 1.0f - A - B; }
    bool forward() const { return hit && (t > 0.0f); }
};

struct Counters {
    uint64_t rays = 0, nodes = 0, tris = 0, inst_tested = 0, inst_entered = 0;
};
inline thread_local Counters tl_counters;

// Moller-Trumbore, src/acceleration_structure.cc:16-49
inline TriHit ray_triangle(const Ray& r, V3 p1, V3 p2, V3 p3, bool cull) {
    TriHit miss;
    V3 e1 = p2 - p1, e2 = p3 - p1;
    V3 pv = cross(r.direction, e2);
    float det = dot(e1, pv);
    if (det == 0.0f) return miss;
    if (cull && det < 0.0f) return miss;
    float inv_det = 1.0f / det;
    V3 tv = r.origin - p1;
    float u = dot(tv, pv) * inv_det;
    if (u < 0.0f || u > 1.0f) return miss;
    V3 qv = cross(tv, e1);
    float v = dot(r.direction, qv) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return miss;
    float t = dot(e2, qv) * inv_det;
    TriHit h;
    h.A = 1.0f - u - v;
    h.B = u;
    h.t = t;
    h.hit = true;
    h.backface = (det < 0.0f) && !cull;
    return h;
}

// BBox, src/acceleration_structure.cc:72-127
struct BBox {
    V3 lo = V3(std::numeric_limits<float>::max()), hi = V3(std::numeric_limits<float>::lowest());
    bool empty() const { return hi.x < lo.x || hi.y < lo.y || hi.z < lo.z; }
    void grow(V3 p) { lo = gmin(lo, p); hi = gmax(hi, p); }
    void grow(const BBox& b) { lo = gmin(lo, b.lo); hi = gmax(hi, b.hi); }
    float half_area() const {  // surface_area() of the reference returns half the area, :124-127
        V3 d = hi - lo;
        return d.x * d.y + d.x * d.z + d.y * d.z;
    }
    VolumeHit test(const Ray& r, V3 inv_dir) const {  // :109-122
        V3 ta = (lo - r.origin) * inv_dir, tb = (hi - r.origin) * inv_dir;
        V3 t0 = gmin(ta, tb), t1 = gmax(ta, tb);
        t0.x = (t0.x > t0.y) ? t0.x : t0.y;
        t0.x = (t0.x > t0.z) ? t0.x : t0.z;
        t1.x = (t1.x < t1.y) ? t1.x : t1.y;
        t1.x = (t1.x < t1.z) ? t1.x : t1.z;
        return {t0.x <= t1.x, t0.x, t1.x};
    }
};

// 7-axis DOP, src/acceleration_structure.h:33-59, .cc:144-206
constexpr float kInvSqrt3 = 0.577350269189625764509148780501957456f;
inline const V3 kDopAxes[7] = {V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1), V3(kInvSqrt3, kInvSqrt3, kInvSqrt3),
                        V3(kInvSqrt3, kInvSqrt3, -kInvSqrt3), V3(kInvSqrt3, -kInvSqrt3, kInvSqrt3),
                        V3(-kInvSqrt3, kInvSqrt3, kInvSqrt3)};
struct Dop {
    V2 mm[7];
    Dop() {
        for (auto& m : mm) m = V2{std::numeric_limits<float>::max(), std::numeric_limits<float>::lowest()};
    }
    void grow(V3 p) {  // DOP::expand(point) == expand(DOP{point}), :145-169
        for (int i = 0; i < 7; ++i) {
            float pr = dot(p, kDopAxes[i]);
            mm[i].x = std::min(mm[i].x, pr);
            mm[i].y = std::max(mm[i].y, pr);
        }
    }
    void grow(const Dop& d) {
        for (int i = 0; i < 7; ++i) {
            mm[i].x = std::min(mm[i].x, d.mm[i].x);
            mm[i].y = std::max(mm[i].y, d.mm[i].y);
        }
    }
    VolumeHit test(const std::array<V2, 7>& proj) const {  // :192-206
        float tmin = std::numeric_limits<float>::lowest(), tmax = std::numeric_limits<float>::max();
        for (int i = 0; i < 7; ++i) {
            float t0 = (mm[i].x - proj[i].x) * proj[i].y;
            float t1 = (mm[i].y - proj[i].x) * proj[i].y;
            if (t0 > t1) std::swap(t0, t1);
            tmin = std::max(t0, tmin);
            tmax = std::min(t1, tmax);
            if (tmax < tmin) return {false, 0.0f, 0.0f};
        }
        return {true, tmin, tmax};
    }
};
inline std::array<V2, 7> dop_projections(const Ray& r) {  // :593-597
    std::array<V2, 7> p;
    for (int i = 0; i < 7; ++i) p[i] = V2{dot(r.origin, kDopAxes[i]), 1.0f / dot(r.direction, kDopAxes[i])};
    return p;
}

struct ObjectData {  // src/acceleration_structure.h:72-79
    Dop volume;
    M4 model, inv_model;
    uint32_t object, mesh, complexity;
};
struct VolumeCandidate {  // volume_hit_and_obj_index, :81-87
    VolumeHit hit;
    uint32_t slot;
    bool operator<(const VolumeCandidate& o) const { return hit.forward_distance() > o.hit.forward_distance(); }
};

// Per-mesh binary BVH, src/acceleration_structure.h:122-163, .cc:392-585
struct BvhTri {
    V3 p1, p2, p3;
    uint32_t index;
};
struct BvhNode {
    BBox box;
    bool leaf = true;
    uint32_t left = 0, right = 0;   // children (inner)
    uint32_t first = 0, count = 0;  // triangle range (leaf)
};
inline int float_to_int_x86(float f) {
    // `int bin = float_expr;` at :472 is undefined for NaN/out-of-range values; x86 cvttss2si, which
    // every build of the reference uses, returns INT_MIN there.  Restated so the result is defined.
    if (!(f == f) || f >= 2147483648.0f || f < -2147483648.0f) return std::numeric_limits<int>::min();
    return static_cast<int>(f);
}
struct MeshBvh {
    std::vector<BvhNode> nodes;
    std::vector<BvhTri> tris;
    bool double_sided = false;
    uint32_t max_leaf = 8;

    void build(const Mesh& mesh, bool ds, uint32_t max_tris) {  // :392-404
        double_sided = ds;
        max_leaf = max_tris;
        tris.reserve(mesh.n_indices / 3);
        for (uint32_t i = 0; i + 2 < mesh.n_indices; i += 3)
            tris.push_back({vpos(mesh.vertices[mesh.indices[i]]), vpos(mesh.vertices[mesh.indices[i + 1]]),
                            vpos(mesh.vertices[mesh.indices[i + 2]]), i});
        nodes.emplace_back();
        parse(0, static_cast<uint32_t>(tris.size()), 0);
    }
    static BBox bounds(const BvhTri* b, const BvhTri* e) {  // triangles_to_bbox, :572-585
        BBox box;
        for (; b != e; ++b) {
            BBox t;
            t.lo = b->p1;
            t.hi = b->p1;
            t.grow(b->p2);
            t.grow(b->p3);
            box.grow(t);
        }
        return box;
    }
    void parse(uint32_t begin, uint32_t end, uint32_t id) {  // :406-447
        BBox box = bounds(tris.data() + begin, tris.data() + end);
        nodes[id] = BvhNode{};
        nodes[id].box = box;
        nodes[id].first = begin;
        nodes[id].count = end - begin;
        if (end - begin <= max_leaf) return;
        uint32_t mid = split(begin, end, box);
        if (mid == end) return;  // no split found: stays a leaf
        uint32_t l = static_cast<uint32_t>(nodes.size());
        nodes[id].leaf = false;
        nodes[id].left = l;
        nodes[id].right = l + 1;
        nodes.emplace_back();
        nodes.emplace_back();
        parse(begin, mid, l);
        parse(mid, end, l + 1);
    }
    static int longest_axis(const BBox& b) {  // :51-59
        V3 e = b.hi - b.lo;
        if (e.x >= e.y && e.x >= e.z) return 0;
        if (e.y >= e.x && e.y >= e.z) return 1;
        return 2;
    }
    uint32_t split(uint32_t begin, uint32_t end, const BBox& box) {  // split_triangles, :449-536
        const int n = static_cast<int>(end - begin);
        const float parent = box.empty() ? kInf : box.half_area() * n;  // :60-63
        float best = parent;
        constexpr int NB = 32;
        struct Bin {
            BBox box;
            int n = 0;
        };
        V3 dim = box.hi - box.lo;
        V3 scale = (NB * 1.0f) / dim;
        int best_axis = longest_axis(box);
        float best_center = 0.0f;
        BvhTri* tb = tris.data() + begin;
        BvhTri* te = tris.data() + end;
        for (int axis = 0; axis < 3; ++axis) {
            std::array<Bin, NB> bins;
            for (BvhTri* t = tb; t != te; ++t) {
                float c = (t->p1[axis] + t->p2[axis] + t->p3[axis]) / 3.0f;
                int id = float_to_int_x86((c - box.lo[axis]) * scale[axis]);
                id = std::clamp(id, 0, NB - 1);
                ++bins[id].n;
                bins[id].box.grow(t->p1);
                bins[id].box.grow(t->p2);
                bins[id].box.grow(t->p3);
            }
            std::array<Bin, NB - 1> ls, rs;
            ls[0] = bins[0];
            for (int i = 1; i < NB - 1; ++i) {
                ls[i] = ls[i - 1];
                ls[i].box.grow(bins[i].box);
                ls[i].n += bins[i].n;
            }
            rs[NB - 2] = bins[NB - 1];
            for (int i = NB - 3; i >= 0; --i) {
                rs[i] = rs[i + 1];
                rs[i].box.grow(bins[i].box);  // bins[i], not bins[i+1]: restated as the reference has it (:492-493)
                rs[i].n += bins[i].n;
            }
            for (int i = 0; i < NB - 1; ++i) {
                float sah = (ls[i].box.empty() || rs[i].box.empty())
                                ? kInf
                                : ls[i].box.half_area() * ls[i].n + rs[i].box.half_area() * rs[i].n;  // :64-67
                if (sah < best) {
                    best_center = box.lo[axis] + (i + 1) * dim[axis] * (1.0f / NB);
                    best = sah;
                    best_axis = axis;
                }
            }
        }
        if (best == parent) {
            best_axis = longest_axis(box);
            best_center = box.lo[best_axis] + 0.5f * dim[best_axis];
        }
        BvhTri* mid = std::partition(tb, te, [best_center, best_axis](const BvhTri& t) {
            float c = (t.p1[best_axis] + t.p2[best_axis] + t.p3[best_axis]) / 3.0f;
            return c < best_center;
        });
        // An empty left part (every centroid right of the bins' plane once compared in binary32) makes the reference
        // recurse without end (:443-444); the median split is taken instead — the one deviation from its tree.
        const bool empty_left = (best != parent) && mid == tb;
        if (best != parent && !empty_left) return begin + static_cast<uint32_t>(mid - tb);
        if (!empty_left && static_cast<uint32_t>(n) <= max_leaf) return end;
        if (mid == tb || mid == te || mid == tb + 1 || mid + 1 == te) {
            std::sort(tb, te, [best_axis](const BvhTri& a, const BvhTri& b) {
                return (a.p1[best_axis] + a.p2[best_axis] + a.p3[best_axis]) <
                       (b.p1[best_axis] + b.p2[best_axis] + b.p3[best_axis]);
            });
            mid = tb + n / 2;
        }
        return begin + static_cast<uint32_t>(mid - tb);
    }
    void info(orc_bvh_info* out) const {  // collect_tree_info, :538-570
        orc_bvh_info r{};
        r.min_depth = 0xffffffffu;
        r.min_tris_in_leaf = 0xffffffffu;
        uint64_t depth_sum = 0, tri_sum = 0, h = 1469598103934665603ull;
        auto fnv = [&h](uint32_t v) {
            for (int i = 0; i < 4; ++i) {
                h ^= (v >> (8 * i)) & 0xffu;
                h *= 1099511628211ull;
            }
        };
        std::vector<std::pair<uint32_t, uint32_t>> stack{{0u, 0u}};
        while (!stack.empty()) {
            auto [id, depth] = stack.back();
            stack.pop_back();
            const BvhNode& nd = nodes[id];
            if (!nd.leaf) {
                ++r.n_nodes;
                stack.push_back({nd.right, depth + 1});
                stack.push_back({nd.left, depth + 1});
                continue;
            }
            ++r.n_leaves;
            r.min_depth = std::min(r.min_depth, depth);
            r.max_depth = std::max(r.max_depth, depth);
            depth_sum += depth;
            r.min_tris_in_leaf = std::min(r.min_tris_in_leaf, nd.count);
            r.max_tris_in_leaf = std::max(r.max_tris_in_leaf, nd.count);
            tri_sum += nd.count;
            fnv(nd.count);
            for (uint32_t k = 0; k < nd.count; ++k) fnv(tris[nd.first + k].index);
        }
        r.n_triangles = static_cast<uint32_t>(tris.size());
        r.mean_depth = static_cast<float>(depth_sum) / r.n_leaves;
        r.mean_tris_in_leaf = static_cast<float>(tri_sum) / r.n_leaves;
        r.checksum = h;
        *out = r;
    }
};

}  // namespace orc
