// Shared pieces of the PNG kernels (png.hip: stored-block encoder + decoder; png_huff.hip: Huffman-compressing encoder).
#pragma once
#include "common.h"

namespace gpemsr {
namespace png {

constexpr uint32_t CRC_POLY = 0xEDB88320u;         // reflected CRC-32 (ISO 3309 / PNG)
constexpr int ADLER_SEG = 4096;                    // raw bytes per Adler segment
constexpr int CRC_SEG = 256;                       // message bytes per CRC segment (one lane each)
constexpr uint32_t ADLER_MOD = 65521u;
constexpr int STORED_MAX = 65535;                  // bytes per stored deflate block

__device__ __forceinline__ uint32_t crc_table_entry(uint32_t i) {
  uint32_t c = i;
#pragma unroll
  for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
  return c;
}

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0)
__host__ __device__ inline uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (uint32_t m = 1u << 31; m != 0 && (a & (m | (m - 1))) != 0; m >>= 1) {      // stops once no term of a is left
    if (a & m) p ^= b;
    b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
  }
  return p;
}
// x^(8 n) mod P
__host__ __device__ inline uint32_t gf2_x8n(unsigned long long n) {
  uint32_t sq = 1u << 30;                          // x^1
  sq = gf2_mulmod(sq, sq); sq = gf2_mulmod(sq, sq); sq = gf2_mulmod(sq, sq);      // x^8
  uint32_t p = 1u << 31;                           // x^0
  while (n) {
    if (n & 1ull) p = gf2_mulmod(sq, p);
    sq = gf2_mulmod(sq, sq);
    n >>= 1;
  }
  return p;
}

__device__ __forceinline__ uint32_t crc_raw_bytes(const uint32_t* tab, const uint8_t* p, int len) {      // initial value 0, no final inversion
  uint32_t c = 0;
  for (int i = 0; i < len; ++i) c = tab[(c ^ p[i]) & 255u] ^ (c >> 8);
  return c;
}

}  // namespace png
}  // namespace gpemsr
