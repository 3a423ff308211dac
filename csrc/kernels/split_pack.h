// Filter layout of the split-operand ("bf16x6") convolution kernel, shared by its own pack kernel and
// by pack_multi: bf16 element i of the packed filter of a 3x3, C -> C layer, i running over
// [K step][output-channel tile][part: h, m, l][lane][8]. Lane (q = lane >> 4, n = lane & 15) of an MFMA
// A operand holds row n (output channel 16 tile + n), K indices 8 q .. 8 q + 7 of the step:
//   16 channels: K = 32 is two taps (2 s, 2 s + 1) x 16 channels; tap 9 does not exist: zeros
//   32 channels: one tap x 32 channels
//   64 channels: half a tap: channels 32 (s & 1) .. + 31 of tap s >> 1
#pragma once
#include <hip/hip_runtime.h>

namespace dtf {

constexpr int split_ksteps(int C) { return (9 * C + 31) / 32; }
constexpr long long split_packed_bf16(int C) { return (long long)split_ksteps(C) * (C / 16) * 3 * 64 * 8; }

// bf16 bits of v, round to nearest even (finite v)
__device__ __forceinline__ unsigned short bf16_rne_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// part 0, 1, 2 of x ~= h + m + l: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), each rounded to
// nearest (the subtractions are exact; the remainder is below 2^-26 |x|)
__device__ __forceinline__ unsigned short split_part_bits(float v, int part) {
  const unsigned short h = bf16_rne_bits(v);
  if (part == 0) return h;
  const float r1 = v - __uint_as_float((unsigned)h << 16);
  const unsigned short m = bf16_rne_bits(r1);
  if (part == 1) return m;
  return bf16_rne_bits(r1 - __uint_as_float((unsigned)m << 16));
}

// flip: the data-gradient filter W'[kh][kw][co][ci] = W[2 - kh][2 - kw][ci][co] (rows = input channels of the forward layer)
__device__ __forceinline__ unsigned short split_packed_element(const float* w, int C, int flip, int i) {
  const int NT = C / 16;
  const int j = i & 7, lane = (i >> 3) & 63, part = (i >> 9) % 3, ct = ((i >> 9) / 3) % NT, ks = (i >> 9) / (3 * NT);
  const int q = lane >> 4, row = ct * 16 + (lane & 15);
  const int kk = 8 * q + j;
  int tap, k;
  if (C == 16) { tap = 2 * ks + (kk >> 4); k = kk & 15; }
  else if (C == 32) { tap = ks; k = kk; }
  else { tap = ks >> 1; k = 32 * (ks & 1) + kk; }
  float v = 0.f;
  if (tap < 9) {
    const int kh = tap / 3, kw = tap % 3;
    v = flip ? w[(((2 - kh) * 3 + (2 - kw)) * C + row) * C + k] : w[((kh * 3 + kw) * C + k) * C + row];
  }
  return split_part_bits(v, part);
}

}  // namespace dtf
