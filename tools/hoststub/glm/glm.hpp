// Minimal glm stand-in for building the reference hot path headless (TEST INFRASTRUCTURE ONLY).
//
// glm is a system dependency of the reference (source/layermap.h:25, Makefile:9) that is NOT
// vendored under /root/reference and is not installed in this image. This header restates the
// definitions of glm 0.9.9 for exactly the surface the hot path uses:
//   normalize(v) = v * (T(1)/sqrt(dot(v,v)))      (glm/detail/func_geometric.inl, func_exponential.inl)
//   dot: vec2 x+y, vec3 x+y+z (left to right), vec4 (x+y)+(z+w)
//   cross(x,y) = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
//   mix(x,y,a:U) = vec<T>( vec<U>(x)*(U(1)-a) + vec<U>(y)*a )
//   round -> std::round (C++11 STL path), floor -> std::floor, fract(x) = x - floor(x)
// No reference test pins these semantics ("parity unpinned" at the glm boundary).
#pragma once
#include <cmath>
#include <cstddef>
#include <type_traits>

namespace glm {

typedef unsigned int uint;

template <typename T> struct tvec2;
template <typename T> struct tvec3;
template <typename T> struct tvec4;

template <typename T> struct tvec2 {
  T x{}, y{};
  constexpr tvec2() = default;
  template <typename S, typename = std::enable_if_t<std::is_arithmetic_v<S>>> tvec2(S s) : x(static_cast<T>(s)), y(static_cast<T>(s)) {}
  template <typename A, typename B> tvec2(A a, B b) : x(static_cast<T>(a)), y(static_cast<T>(b)) {}
  template <typename U> tvec2(const tvec2<U>& v) : x(static_cast<T>(v.x)), y(static_cast<T>(v.y)) {}
  T& operator[](int i) { return (&x)[i]; }
  const T& operator[](int i) const { return (&x)[i]; }
  tvec2& operator+=(const tvec2& o) { x += o.x; y += o.y; return *this; }
  tvec2& operator-=(const tvec2& o) { x -= o.x; y -= o.y; return *this; }
};

template <typename T> struct tvec3 {
  T x{}, y{}, z{};
  constexpr tvec3() = default;
  template <typename S, typename = std::enable_if_t<std::is_arithmetic_v<S>>> tvec3(S s) : x(static_cast<T>(s)), y(static_cast<T>(s)), z(static_cast<T>(s)) {}
  template <typename A, typename B, typename C>
  tvec3(A a, B b, C c) : x(static_cast<T>(a)), y(static_cast<T>(b)), z(static_cast<T>(c)) {}
  template <typename U> tvec3(const tvec3<U>& v)
      : x(static_cast<T>(v.x)), y(static_cast<T>(v.y)), z(static_cast<T>(v.z)) {}
  template <typename U> explicit tvec3(const tvec4<U>& v);
  T& operator[](int i) { return (&x)[i]; }
  const T& operator[](int i) const { return (&x)[i]; }
  tvec3& operator+=(const tvec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
  tvec3& operator-=(const tvec3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
};

template <typename T> struct tvec4 {
  T x{}, y{}, z{}, w{};
  constexpr tvec4() = default;   // (constexpr: `new vec4[N]{...}` of the host's lbmwind.h must not expand into N constructor calls)
  template <typename S, typename = std::enable_if_t<std::is_arithmetic_v<S>>> tvec4(S s)
      : x(static_cast<T>(s)), y(static_cast<T>(s)), z(static_cast<T>(s)), w(static_cast<T>(s)) {}
  template <typename A, typename B, typename C, typename D>
  tvec4(A a, B b, C c, D d)
      : x(static_cast<T>(a)), y(static_cast<T>(b)), z(static_cast<T>(c)), w(static_cast<T>(d)) {}
  template <typename U> tvec4(const tvec4<U>& v)
      : x(static_cast<T>(v.x)), y(static_cast<T>(v.y)), z(static_cast<T>(v.z)), w(static_cast<T>(v.w)) {}
  T& operator[](int i) { return (&x)[i]; }
  const T& operator[](int i) const { return (&x)[i]; }
};

template <typename T> template <typename U>
tvec3<T>::tvec3(const tvec4<U>& v) : x(static_cast<T>(v.x)), y(static_cast<T>(v.y)), z(static_cast<T>(v.z)) {}

typedef tvec2<float> vec2;
typedef tvec3<float> vec3;
typedef tvec4<float> vec4;
typedef tvec2<int> ivec2;
typedef tvec3<int> ivec3;
typedef tvec2<bool> bvec2;
typedef tvec2<double> dvec2;
typedef tvec3<double> dvec3;
typedef tvec4<double> dvec4;

// ---- vec2 arithmetic (component-wise, same T) ----
template <typename T> tvec2<T> operator+(const tvec2<T>& a, const tvec2<T>& b) { return tvec2<T>(a.x + b.x, a.y + b.y); }
template <typename T> tvec2<T> operator-(const tvec2<T>& a, const tvec2<T>& b) { return tvec2<T>(a.x - b.x, a.y - b.y); }
template <typename T> tvec2<T> operator*(const tvec2<T>& a, const tvec2<T>& b) { return tvec2<T>(a.x * b.x, a.y * b.y); }
template <typename T> tvec2<T> operator/(const tvec2<T>& a, const tvec2<T>& b) { return tvec2<T>(a.x / b.x, a.y / b.y); }
template <typename T> tvec2<T> operator+(const tvec2<T>& a, T s) { return tvec2<T>(a.x + s, a.y + s); }
template <typename T> tvec2<T> operator-(const tvec2<T>& a, T s) { return tvec2<T>(a.x - s, a.y - s); }
template <typename T> tvec2<T> operator*(const tvec2<T>& a, T s) { return tvec2<T>(a.x * s, a.y * s); }
template <typename T> tvec2<T> operator/(const tvec2<T>& a, T s) { return tvec2<T>(a.x / s, a.y / s); }
template <typename T> tvec2<T> operator*(T s, const tvec2<T>& a) { return tvec2<T>(s * a.x, s * a.y); }
template <typename T> tvec2<T> operator-(const tvec2<T>& a) { return tvec2<T>(-a.x, -a.y); }
template <typename T> bool operator==(const tvec2<T>& a, const tvec2<T>& b) { return a.x == b.x && a.y == b.y; }

// ---- vec3 arithmetic ----
template <typename T> tvec3<T> operator+(const tvec3<T>& a, const tvec3<T>& b) { return tvec3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> tvec3<T> operator-(const tvec3<T>& a, const tvec3<T>& b) { return tvec3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> tvec3<T> operator*(const tvec3<T>& a, const tvec3<T>& b) { return tvec3<T>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename T> tvec3<T> operator/(const tvec3<T>& a, const tvec3<T>& b) { return tvec3<T>(a.x / b.x, a.y / b.y, a.z / b.z); }
template <typename T> tvec3<T> operator*(const tvec3<T>& a, T s) { return tvec3<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> tvec3<T> operator/(const tvec3<T>& a, T s) { return tvec3<T>(a.x / s, a.y / s, a.z / s); }
template <typename T> tvec3<T> operator*(T s, const tvec3<T>& a) { return tvec3<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> tvec3<T> operator-(const tvec3<T>& a) { return tvec3<T>(-a.x, -a.y, -a.z); }

// ---- vec4 arithmetic ----
template <typename T> tvec4<T> operator+(const tvec4<T>& a, const tvec4<T>& b) { return tvec4<T>(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <typename T> tvec4<T> operator*(const tvec4<T>& a, T s) { return tvec4<T>(a.x * s, a.y * s, a.z * s, a.w * s); }
template <typename T> tvec4<T> operator/(const tvec4<T>& a, T s) { return tvec4<T>(a.x / s, a.y / s, a.z / s, a.w / s); }
template <typename T> tvec4<T> operator*(const tvec4<T>& a, const tvec4<T>& b) { return tvec4<T>(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
template <typename T> tvec4<T> operator/(const tvec4<T>& a, const tvec4<T>& b) { return tvec4<T>(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
template <typename T> tvec4<T> operator-(const tvec4<T>& a, const tvec4<T>& b) { return tvec4<T>(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// ---- common ----
inline vec2 floor(const vec2& v) { return vec2(std::floor(v.x), std::floor(v.y)); }
inline vec2 round(const vec2& v) { return vec2(std::round(v.x), std::round(v.y)); }
inline vec2 fract(const vec2& v) { return v - floor(v); }

template <typename T> T dot(const tvec2<T>& a, const tvec2<T>& b) { tvec2<T> t(a * b); return t.x + t.y; }
template <typename T> T dot(const tvec3<T>& a, const tvec3<T>& b) { tvec3<T> t(a * b); return t.x + t.y + t.z; }
template <typename T> T length(const tvec2<T>& v) { return std::sqrt(dot(v, v)); }
template <typename T> T length(const tvec3<T>& v) { return std::sqrt(dot(v, v)); }
template <typename T> tvec2<T> normalize(const tvec2<T>& v) { return v * (static_cast<T>(1) / std::sqrt(dot(v, v))); }
template <typename T> tvec3<T> normalize(const tvec3<T>& v) { return v * (static_cast<T>(1) / std::sqrt(dot(v, v))); }
template <typename T> tvec3<T> cross(const tvec3<T>& x, const tvec3<T>& y) {
  return tvec3<T>(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}

template <typename T, typename U> tvec2<T> mix(const tvec2<T>& x, const tvec2<T>& y, U a) {
  return tvec2<T>(tvec2<U>(x) * (static_cast<U>(1) - a) + tvec2<U>(y) * a);
}
template <typename T, typename U> tvec3<T> mix(const tvec3<T>& x, const tvec3<T>& y, U a) {
  return tvec3<T>(tvec3<U>(x) * (static_cast<U>(1) - a) + tvec3<U>(y) * a);
}
template <typename T, typename U> tvec4<T> mix(const tvec4<T>& x, const tvec4<T>& y, U a) {
  return tvec4<T>(tvec4<U>(x) * (static_cast<U>(1) - a) + tvec4<U>(y) * a);
}

// ---- mat4: only so that render-side host code (scene.h, the pipeline lambda of SoilMachine.cpp) compiles ----
struct mat4 {
  float m[16];
  mat4() { for (int i = 0; i < 16; i++) m[i] = 0.f; }
  mat4(float d) { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? d : 0.f; }
  mat4(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8, float a9, float a10,
       float a11, float a12, float a13, float a14, float a15) { float t[16] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15}; for (int i = 0; i < 16; i++) m[i] = t[i]; }
};
inline mat4 operator*(const mat4& a, const mat4& b) {
  mat4 r;
  for (int c = 0; c < 4; c++) for (int rr = 0; rr < 4; rr++) { float s = 0; for (int k = 0; k < 4; k++) s += a.m[k * 4 + rr] * b.m[c * 4 + k]; r.m[c * 4 + rr] = s; }
  return r;
}

// render-side matrix builders the host tree's own scene.h / lbmwind.h name (values are never looked at headless)
template <typename T = float> mat4 ortho(T, T, T, T, T, T) { return mat4(1.0f); }
template <typename A, typename B, typename C> mat4 lookAt(const A&, const B&, const C&) { return mat4(1.0f); }
template <typename V> mat4 scale(const mat4& m, const V&) { return m; }
template <typename V> mat4 translate(const mat4& m, const V&) { return m; }
template <typename V> mat4 rotate(const mat4& m, float, const V&) { return m; }
inline float radians(float d) { return d * 0.017453292519943295f; }

template <typename T> bvec2 greaterThanEqual(const tvec2<T>& a, const tvec2<T>& b) { return bvec2(a.x >= b.x, a.y >= b.y); }
template <typename T> bvec2 lessThan(const tvec2<T>& a, const tvec2<T>& b) { return bvec2(a.x < b.x, a.y < b.y); }
inline bool all(const bvec2& v) { return v.x && v.y; }

}  // namespace glm
