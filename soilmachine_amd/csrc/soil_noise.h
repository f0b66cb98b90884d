// soil_noise.h -- OpenSimplex2 3-D + FBm, the only FastNoiseLite path Layermap::initialize uses
// (layermap.h:181-196 -> surface.h:82-99 -> FastNoiseLite.h:321-345,473-552,686-727,865-885,1053-1150).
// Integer hash + float arithmetic in source order; compiled with -ffp-contract=off so every cell gets
// the bit pattern the reference's CPU code produces.
//
// simplex2_3d / fbm below restate the algorithm of FastNoiseLite (the noise library the reference vendors as
// source/include/FastNoiseLite.h, v1.0.1): bit-exact float arithmetic leaves one legal order of operations, so the
// statements follow the library's. FastNoiseLite is distributed under the MIT License:
//
//   MIT License
//   Copyright(c) 2020 Jordan Peck (jordan.me2@gmail.com)
//   Copyright(c) 2020 Contributors
//
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files(the "Software"), to deal in the Software without restriction, including without limitation the
//   rights to use, copy, modify, merge, publish, distribute, sublicense, and / or sell copies of the Software, and to
//   permit persons to whom the Software is furnished to do so, subject to the following conditions :
//
//   The above copyright notice and this permission notice shall be included in all copies or substantial portions of
//   the Software.
//
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//   WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR
//   COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//   OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
#pragma once
#include <stdint.h>

#ifndef SMX_D
#define SMX_D __device__ __forceinline__
#endif

namespace smx {

struct NoiseCfg {
  int seed;            // 1337: the reference never seeds the noise object (FastNoiseLite.h:114-116)
  int octaves;
  float frequency, lacunarity, gain, bounding;
};

// 12 cube-edge gradients repeated 5x + 4 extras = 64 entries (FastNoiseLite.h:2529-2546)
SMX_D void grad3(int e, float& gx, float& gy, float& gz) {
  int b = e < 60 ? e % 12 : -1;
  if (b < 0) {
    const int t = e - 60;   // (1,1,0) (0,-1,1) (-1,1,0) (0,-1,-1)
    gx = (t == 0) ? 1.f : (t == 2 ? -1.f : 0.f);
    gy = (t & 1) ? -1.f : 1.f;
    gz = (t == 1) ? 1.f : (t == 3 ? -1.f : 0.f);
    return;
  }
  const int grp = b >> 2, k = b & 3;          // grp 0: (0,±1,±1)  1: (±1,0,±1)  2: (±1,±1,0)
  const float a = (k & 1) ? -1.f : 1.f;       // first varying component: + - + -
  const float c = (k & 2) ? -1.f : 1.f;       // second varying component: + + - -
  if (grp == 0) { gx = 0.f; gy = a; gz = c; }
  else if (grp == 1) { gx = a; gy = 0.f; gz = c; }
  else { gx = a; gy = c; gz = 0.f; }
}

SMX_D float gradcoord(int sd, int xp, int yp, int zp, float xd, float yd, float zd) {
  uint32_t h = (uint32_t)sd ^ (uint32_t)xp ^ (uint32_t)yp ^ (uint32_t)zp;
  h *= 0x27d4eb2du;
  int hash = (int)h;
  hash ^= hash >> 15;
  hash &= 63 << 2;
  float xg, yg, zg;
  grad3(hash >> 2, xg, yg, zg);
  return xd * xg + yd * yg + zd * zg;
}

SMX_D int fastround(float f) { return f >= 0 ? (int)(f + 0.5f) : (int)(f - 0.5f); }

SMX_D float simplex2_3d(int sd, float x, float y, float z) {   // SingleOpenSimplex2 (FastNoiseLite.h:1053-1150)
  const int PX = 501125321, PY = 1136930381, PZ = 1720413743;
  int i = fastround(x), j = fastround(y), k = fastround(z);
  float x0 = (float)(x - i), y0 = (float)(y - j), z0 = (float)(z - k);
  int xs = (int)(-1.0f - x0) | 1, ys = (int)(-1.0f - y0) | 1, zs = (int)(-1.0f - z0) | 1;
  float ax0 = xs * -x0, ay0 = ys * -y0, az0 = zs * -z0;
  i = (int)((uint32_t)i * (uint32_t)PX); j = (int)((uint32_t)j * (uint32_t)PY); k = (int)((uint32_t)k * (uint32_t)PZ);
  float value = 0;
  float a = (0.6f - x0 * x0) - (y0 * y0 + z0 * z0);
  for (int l = 0;; l++) {
    if (a > 0) value += (a * a) * (a * a) * gradcoord(sd, i, j, k, x0, y0, z0);
    float b = a + 1;
    int i1 = i, j1 = j, k1 = k;
    float x1 = x0, y1 = y0, z1 = z0;
    if (ax0 >= ay0 && ax0 >= az0) {
      x1 += xs; b -= xs * 2 * x1; i1 = (int)((uint32_t)i1 - (uint32_t)(xs * PX));
    } else if (ay0 > ax0 && ay0 >= az0) {
      y1 += ys; b -= ys * 2 * y1; j1 = (int)((uint32_t)j1 - (uint32_t)(ys * PY));
    } else {
      z1 += zs; b -= zs * 2 * z1; k1 = (int)((uint32_t)k1 - (uint32_t)(zs * PZ));
    }
    if (b > 0) value += (b * b) * (b * b) * gradcoord(sd, i1, j1, k1, x1, y1, z1);
    if (l == 1) break;
    ax0 = 0.5f - ax0; ay0 = 0.5f - ay0; az0 = 0.5f - az0;
    x0 = xs * ax0; y0 = ys * ay0; z0 = zs * az0;
    a += (0.75f - ax0) - (ay0 + az0);
    i = (int)((uint32_t)i + (uint32_t)((xs >> 1) & PX));
    j = (int)((uint32_t)j + (uint32_t)((ys >> 1) & PY));
    k = (int)((uint32_t)k + (uint32_t)((zs >> 1) & PZ));
    xs = -xs; ys = -ys; zs = -zs;
    sd = ~sd;
  }
  return value * 32.69428253173828125f;
}

SMX_D float noise_fbm(const NoiseCfg& c, float x, float y, float z) {   // GetNoise 3-D + GenFractalFBm
  x *= c.frequency; y *= c.frequency; z *= c.frequency;
  const float R3 = (float)(2.0 / 3.0);            // TransformType3D_DefaultOpenSimplex2 (:712-720)
  float r = (x + y + z) * R3;
  x = r - x; y = r - y; z = r - z;
  int sd = c.seed;
  float sum = 0, amp = c.bounding;
  for (int o = 0; o < c.octaves; o++) {
    float n = simplex2_3d(sd++, x, y, z);
    sum += n * amp;
    amp *= 1.0f + 0.0f * (((n + 1) * 0.5f) - 1.0f);   // Lerp(1, (n+1)*0.5, mWeightedStrength = 0)
    x *= c.lacunarity; y *= c.lacunarity; z *= c.lacunarity;
    amp *= c.gain;
  }
  return sum;
}

}  // namespace smx
