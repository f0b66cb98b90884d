// glsl_cs.h -- just enough of GLSL 4.60 compute to compile the reference's LBM shaders AS C++ (TEST INFRASTRUCTURE ONLY).
//
// oracle/glsl2cpp.py turns the storage declarations of source/include/lbmwind/shader/LBM/{lbm,init,collide,stream}.cs and
// shader/move.cs into macros (nothing else of the text changes); this header gives those macros and the vector types a
// meaning, and oracle/lbm_ref.cpp runs main() once per invocation. Compiled with -fsingle-precision-constant (a GLSL
// literal `0.05` is a float) and -ffp-contract=off. What stays outside any pin: a GLSL compiler may contract/reassociate
// float arithmetic and fold constants at another precision; this build evaluates every expression as written, in float.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace glsl {
using std::sqrt;
typedef unsigned int uint;
struct ivec3; struct ivec4;
struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  vec3(float a) : x(a), y(a), z(a) {}
  vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  vec3(const ivec3& v);                                       // implicit int -> float conversion (GLSL 4.1.10)
  vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
};
struct ivec3 {
  int x, y, z;
  ivec3() : x(0), y(0), z(0) {}
  ivec3(int a, int b, int c) : x(a), y(b), z(c) {}
  struct from_u { uint x, y, z; };
  explicit ivec3(const from_u& u) : x((int)u.x), y((int)u.y), z((int)u.z) {}
};
inline vec3::vec3(const ivec3& v) : x((float)v.x), y((float)v.y), z((float)v.z) {}
struct uvec3 {                                                // gl_GlobalInvocationID: .x .y .z and the .xyz swizzle
  union { struct { uint x, y, z; }; ivec3::from_u xyz; };
  uvec3() : x(0), y(0), z(0) {}
};
struct vec4 {
  float x, y, z, w;
  vec4() : x(0), y(0), z(0), w(0) {}
  vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
  vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
  vec4(const ivec4& v);
  vec4& operator+=(const vec4& o) { x += o.x; y += o.y; z += o.z; w += o.w; return *this; }
};
struct ivec4 {
  int x, y, z, w;
  explicit ivec4(int a) : x(a), y(a), z(a), w(a) {}
  explicit ivec4(const vec4& v) : x((int)v.x), y((int)v.y), z((int)v.z), w((int)v.w) {}   // truncation toward zero
  ivec4(int a, int b, int c, int d) : x(a), y(b), z(c), w(d) {}
};
inline vec4::vec4(const ivec4& v) : x((float)v.x), y((float)v.y), z((float)v.z), w((float)v.w) {}
inline ivec3 operator+(const ivec3& a, const ivec3& b) { return ivec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline ivec4 operator+(const ivec4& a, const ivec4& b) { return ivec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec3 operator*(float s, const vec3& v) { return vec3(s * v.x, s * v.y, s * v.z); }
inline vec3 operator*(float s, const ivec3& v) { return s * vec3(v); }
inline vec3 operator/(const vec3& v, float s) { return vec3(v.x / s, v.y / s, v.z / s); }
inline vec4 operator*(float s, const vec4& v) { return vec4(s * v.x, s * v.y, s * v.z, s * v.w); }
inline vec4 operator+(const vec4& a, const vec4& b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4& a, const vec4& b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline float dot(const vec3& a, const vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// A shader storage block: a runtime-sized array. `weak_after_strong`: see FBUF below.
template <class T> struct buf {
  T* d = nullptr;
  T& operator[](size_t i) const { return d[i]; }
};
// The one place the reference leaves to the GL implementation: stream.cs lets a cell on a driven face overwrite its own
// distributions (stream.cs:27-34) in the SAME dispatch in which its neighbours push theirs into those very slots (:9-20) --
// a data race whose outcome GLSL does not define. The shader's comment ("Wetnode Approach") wants the overwrite to stand:
// buffer F is therefore a proxy in which a write by the invocation that OWNS the cell is final for the rest of the dispatch,
// whatever order the invocations run in. (lbm_oracle.cpp and the HIP kernel take the same decision.)
struct FBUF {
  float* d = nullptr; unsigned char* fin = nullptr; const size_t* owner_cell = nullptr; int q = 19;
  struct ref {
    const FBUF& b; size_t i;
    operator float() const { return b.d[i]; }
    ref& operator=(float v) {
      if (i / (size_t)b.q == *b.owner_cell) { b.d[i] = v; if (b.fin) b.fin[i] = 1; }
      else if (!b.fin || !b.fin[i]) b.d[i] = v;
      return *this;
    }
  };
  ref operator[](size_t i) const { return ref{*this, i}; }
};
}  // namespace glsl

#define GLSL_LOCAL_SIZE(...)
#define GLSL_UNIFORM static
#define GLSL_BUFFER(T, NAME) static glsl::buf<T> NAME;
#define GLSL_BUFFER_F(NAME) static glsl::FBUF NAME;
