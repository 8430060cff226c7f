// The PNG edges of the inference loop on the device (R:output_GPEMSR.py:88-95 `cv2.imwrite(path, output)` per slice and
// R:data/util.py:75-88 `cv2.imread` -> float32 / 255): byte and integer work, bound by HBM bandwidth (encode) or by the serial inflate
// (decode; the LR inputs are 128 x 128).
//
// ENCODE  gpemsr_png_encode_gray8: 8-bit grayscale images [n][h][w] -> n complete PNG files in device memory.  The zlib stream uses
//   STORED deflate blocks (the 8-bit EM images compress to ~0.75 with zlib and the host's compression is what does not scale: 2,000
//   output MP/s are 2,000 images per second at 10-15 ms of zlib each); every byte of the file except the two checksums is a pure function of its
//   offset, so the file is assembled by one coalesced pass; Adler-32 (over the filtered scanlines) and CRC-32 (over the IDAT chunk) are
//   computed per 4 KB / 1 KB segment in parallel and folded with the checksums' own combine rules (sum rules for Adler; multiplication by
//   x^(8 len) modulo the CRC polynomial, zlib's crc32_combine recipe).
// DECODE  gpemsr_png_decode_gray8: the concatenated IDAT payloads of n non-interlaced 8-bit grayscale PNGs -> float32 [n][h][w] / 255.
//   One lane per image inflates (stored / fixed / dynamic Huffman blocks, the LZ77 window is the output itself), verifies Adler-32 and
//   undoes the five scanline filters; a second kernel converts.  Chunk parsing (lengths, IHDR fields) stays on the host: it touches 50 bytes.
#include "png_common.h"

namespace gpemsr {
namespace {
using namespace png;

struct PngGeo {
  int n, h, w;
  long long img_stride; int row_stride;            // source image: bytes between images / rows
  long long out_stride;                            // bytes between files
  long long raw;                                   // h * (w + 1)
  int nblk;                                        // stored blocks
  long long zlen;                                  // zlib stream bytes = 2 + 5 nblk + raw + 4
  long long total;                                 // file bytes = 57 + zlen
  // CRC-32 of the IDAT chunk (type + data = clen bytes): segments of CRC_SEG bytes counted from the END of the message (a raw CRC, initial
  // value 0, ignores leading zeros, so the message is thought left-padded to 256 * per whole segments); x_* = powers of x modulo the
  // polynomial, computed on the host
  long long clen; int nseg, per;
  uint32_t x_seg, x_lvl[8], x_len;
};

// byte `o` of the file (checksum fields read as 0)
__device__ __forceinline__ unsigned file_byte(const PngGeo& G, const uint8_t* img, long long o) {
  if (o < 33) {
    switch ((int)o) {
      case 0: return 0x89; case 1: return 'P'; case 2: return 'N'; case 3: return 'G'; case 4: return 0x0D; case 5: return 0x0A; case 6: return 0x1A; case 7: return 0x0A;
      case 8: case 9: case 10: return 0; case 11: return 13;
      case 12: return 'I'; case 13: return 'H'; case 14: return 'D'; case 15: return 'R';
      case 16: return (unsigned)(G.w >> 24) & 255u; case 17: return (unsigned)(G.w >> 16) & 255u; case 18: return (unsigned)(G.w >> 8) & 255u; case 19: return (unsigned)G.w & 255u;
      case 20: return (unsigned)(G.h >> 24) & 255u; case 21: return (unsigned)(G.h >> 16) & 255u; case 22: return (unsigned)(G.h >> 8) & 255u; case 23: return (unsigned)G.h & 255u;
      case 24: return 8;                              // bit depth; colour type 0, compression 0, filter 0, no interlace follow
      default: return 0;                              // 25-28, and the IHDR CRC (29-32) patched later
    }
  }
  if (o < 41) {
    if (o < 37) return (unsigned)(G.zlen >> (8 * (36 - o))) & 255u;
    const char t[4] = {'I', 'D', 'A', 'T'};
    return (unsigned)t[o - 37];
  }
  long long z = o - 41;                              // offset in the zlib stream
  if (z < G.zlen) {
    if (z == 0) return 0x78;
    if (z == 1) return 0x01;
    z -= 2;
    const long long body = 5ll * G.nblk + G.raw;
    if (z >= body) return 0;                         // Adler-32, patched later
    const long long blk = z / (STORED_MAX + 5);
    const int within = (int)(z - blk * (STORED_MAX + 5));
    if (within < 5) {
      const long long left = G.raw - blk * STORED_MAX;
      const unsigned len = left < STORED_MAX ? (unsigned)left : (unsigned)STORED_MAX;
      switch (within) {
        case 0: return blk == G.nblk - 1 ? 1u : 0u;  // BFINAL, BTYPE = 00 (byte-aligned: every block starts on a byte)
        case 1: return len & 255u; case 2: return len >> 8;
        case 3: return (~len) & 255u; default: return ((~len) >> 8) & 255u;
      }
    }
    const long long r = blk * STORED_MAX + (within - 5);
    const int row = (int)(r / (G.w + 1)), col = (int)(r - (long long)row * (G.w + 1)) - 1;
    return col < 0 ? 0u : img[(long long)row * G.row_stride + col];      // filter type 0 (None)
  }
  z -= G.zlen;                                       // IDAT CRC (0-3, patched later), then IEND
  if (z < 4) return 0;
  const unsigned char iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
  return iend[z - 4];
}

__global__ __launch_bounds__(256) void png_assemble_kernel(PngGeo G, const uint8_t* __restrict__ src, uint8_t* __restrict__ out) {
  const int img = blockIdx.y;
  const long long o0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
  if (o0 >= G.total) return;
  const uint8_t* ip = src + (long long)img * G.img_stride;
  uint8_t* op = out + (long long)img * G.out_stride;
  unsigned v[4] = {0, 0, 0, 0};
  const int cnt = G.total - o0 < 16 ? (int)(G.total - o0) : 16;
  // fast path: 16 bytes inside one scanline of one stored block
  bool fast = false;
  if (cnt == 16 && o0 >= 43) {
    const long long z = o0 - 43, body = 5ll * G.nblk + G.raw;
    if (z + 16 <= body) {
      const long long blk = z / (STORED_MAX + 5);
      const int within = (int)(z - blk * (STORED_MAX + 5));
      if (within >= 5 && within + 16 <= STORED_MAX + 5) {
        const long long r = blk * STORED_MAX + (within - 5);
        const int row = (int)(r / (G.w + 1)), col = (int)(r - (long long)row * (G.w + 1)) - 1;
        if (col >= 0 && col + 16 <= G.w) {
          // 16 pixels from aligned dwords (the scanline's filter byte shifts every row by one: the source is rarely 16-byte aligned)
          const uintptr_t a = reinterpret_cast<uintptr_t>(ip + (long long)row * G.row_stride + col);
          const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
          const unsigned sh = (unsigned)(a & 3) * 8u;
          const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
          if (sh == 0) { v[0] = d0; v[1] = d1; v[2] = d2; v[3] = d3; }
          else {
            const uint32_t d4 = q[4];
            v[0] = (d0 >> sh) | (d1 << (32u - sh)); v[1] = (d1 >> sh) | (d2 << (32u - sh));
            v[2] = (d2 >> sh) | (d3 << (32u - sh)); v[3] = (d3 >> sh) | (d4 << (32u - sh));
          }
          fast = true;
        }
      }
    }
  }
  if (!fast)
    for (int k = 0; k < cnt; ++k) v[k >> 2] |= file_byte(G, ip, o0 + k) << (8 * (k & 3));
  if (cnt == 16 && ((reinterpret_cast<uintptr_t>(op) + (uintptr_t)o0) & 15) == 0) *reinterpret_cast<uint4*>(op + o0) = make_uint4(v[0], v[1], v[2], v[3]);
  else for (int k = 0; k < cnt; ++k) op[o0 + k] = (uint8_t)(v[k >> 2] >> (8 * (k & 3)));
}

// Adler-32 partial sums of raw segment s: a = sum d_j, b = sum (len - j) d_j  (j local)
__global__ __launch_bounds__(256) void png_adler_partial_kernel(PngGeo G, const uint8_t* __restrict__ src, unsigned long long* __restrict__ part, int nseg) {
  const int img = blockIdx.y, s = blockIdx.x;
  const long long r0 = (long long)s * ADLER_SEG;
  const int len = G.raw - r0 < ADLER_SEG ? (int)(G.raw - r0) : ADLER_SEG;
  const uint8_t* ip = src + (long long)img * G.img_stride;
  unsigned long long a = 0, b = 0;
  for (int j = threadIdx.x; j < len; j += 256) {
    const long long r = r0 + j;
    const int row = (int)(r / (G.w + 1)), col = (int)(r - (long long)row * (G.w + 1)) - 1;
    const unsigned d = col < 0 ? 0u : ip[(long long)row * G.row_stride + col];
    a += d; b += (unsigned long long)(len - j) * d;
  }
  __shared__ unsigned long long sa[256], sb[256];
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sa[threadIdx.x] += sa[threadIdx.x + o]; sb[threadIdx.x] += sb[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((long long)img * nseg + s) * 2] = sa[0]; part[((long long)img * nseg + s) * 2 + 1] = sb[0]; }
}

// raw CRC-32 of every segment of the IDAT chunk's type + data (file offsets 37 .. 37 + clen) with the Adler-32 field still zero; segment j
// (counted from the front of the padded message) ends (nseg - j - 1) whole segments before the end, the first one may be short; one lane each
__global__ __launch_bounds__(256) void png_crc_partial_kernel(PngGeo G, const uint8_t* __restrict__ out, uint32_t* __restrict__ part) {
  __shared__ uint32_t tab[256];
  tab[threadIdx.x] = crc_table_entry(threadIdx.x);
  __syncthreads();
  const int img = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
  if (s >= G.nseg) return;
  const long long end = G.clen - (long long)(G.nseg - 1 - s) * CRC_SEG;
  const long long beg = end - CRC_SEG > 0 ? end - CRC_SEG : 0;
  part[(long long)img * G.nseg + s] = crc_raw_bytes(tab, out + (long long)img * G.out_stride + 37 + beg, (int)(end - beg));
}

// one workgroup per image: Adler-32 from the partial sums; the chunk CRCs.  Adler: A = 1 + sum d_i, B = n + sum (n - i) d_i, and a segment's
// share of B is b_s + a_s * (bytes after the segment).  CRC: lane t folds `per` consecutive segments (crc <- crc * x^(8 SEG) + next), a tree
// over the 256 lanes multiplies by x^(8 * run * 2^k) per level; the Adler field's four bytes enter by linearity (they are the message's tail:
// raw(M) = raw(M with zeros there) + raw(those four bytes)); standard CRC = raw + 0xFFFFFFFF * x^(8 clen) + 0xFFFFFFFF.
__global__ __launch_bounds__(256) void png_final_kernel(PngGeo G, const unsigned long long* __restrict__ apart, int na, const uint32_t* __restrict__ cpart,
                                                        uint8_t* __restrict__ out) {
  __shared__ uint32_t tab[256];
  __shared__ unsigned long long sa[256], sb[256];
  __shared__ uint32_t sc[256];
  const int img = blockIdx.x, t = threadIdx.x;
  tab[t] = crc_table_entry(t);
  unsigned long long a = 0, b = 0;
  for (int s = t; s < na; s += 256) {
    const long long r0 = (long long)s * ADLER_SEG;
    const long long len = G.raw - r0 < ADLER_SEG ? G.raw - r0 : ADLER_SEG;
    const unsigned long long as = apart[((long long)img * na + s) * 2], bs = apart[((long long)img * na + s) * 2 + 1];
    a += as;
    b += bs % ADLER_MOD + (as % ADLER_MOD) * ((unsigned long long)(G.raw - r0 - len) % ADLER_MOD);
  }
  sa[t] = a; sb[t] = b % ADLER_MOD;
  // CRC: this lane's run of the padded message (segments t * per .. + per; the first 256 * per - nseg of them are padding = zero)
  const int pad = 256 * G.per - G.nseg;
  uint32_t c = 0;
  for (int k = 0; k < G.per; ++k) {
    const int s = t * G.per + k - pad;
    c = gf2_mulmod(G.x_seg, c);
    if (s >= 0) c ^= cpart[(long long)img * G.nseg + s];
  }
  sc[t] = c;
  __syncthreads();
#pragma unroll
  for (int lv = 0; lv < 8; ++lv) {
    const int o = 1 << lv;
    if ((t & (2 * o - 1)) == 0) { sa[t] += sa[t + o]; sb[t] += sb[t + o]; sc[t] = gf2_mulmod(G.x_lvl[lv], sc[t]) ^ sc[t + o]; }
    __syncthreads();
  }
  if (t != 0) return;
  uint8_t* op = out + (long long)img * G.out_stride;
  const unsigned long long A = (1ull + sa[0]) % ADLER_MOD, B = ((unsigned long long)(G.raw % ADLER_MOD) + sb[0]) % ADLER_MOD;
  const uint32_t adler = (uint32_t)((B << 16) | A);
  const long long pa = 41 + G.zlen - 4;
  const uint8_t ab[4] = {(uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler};
  op[pa] = ab[0]; op[pa + 1] = ab[1]; op[pa + 2] = ab[2]; op[pa + 3] = ab[3];
  const uint32_t crc = sc[0] ^ crc_raw_bytes(tab, ab, 4) ^ gf2_mulmod(G.x_len, 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
  op[pa + 4] = (uint8_t)(crc >> 24); op[pa + 5] = (uint8_t)(crc >> 16); op[pa + 6] = (uint8_t)(crc >> 8); op[pa + 7] = (uint8_t)crc;
  uint32_t hc = 0xFFFFFFFFu;                              // "IHDR" + 13 data bytes
  for (int i = 12; i < 29; ++i) hc = tab[(hc ^ op[i]) & 255u] ^ (hc >> 8);
  hc ^= 0xFFFFFFFFu;
  op[29] = (uint8_t)(hc >> 24); op[30] = (uint8_t)(hc >> 16); op[31] = (uint8_t)(hc >> 8); op[32] = (uint8_t)hc;
}

// ------------------------------------------------------------------------------------------------------------------ decode
// One workgroup (one wave) per image.  Inflate is serial by nature: lane 0 walks the bit stream; what keeps a single lane tolerable is that
// nothing it touches is further away than LDS -- the Huffman tables (a 512-entry direct table for codes of up to 9 bits, the canonical
// count / symbol lists for the longer ones), the code lengths, and the LZ77 window (the last 32 KB of output, circular); the output itself is
// only STORED to global memory.  Afterwards all 64 lanes check Adler-32 (partial-sum rule), undo the scanline filters row by row (None / Up in
// parallel, Sub as a wave-wide prefix sum, Average / Paeth serially over the two rows held in LDS) and write float32 pixel / 255.
constexpr int WIN = 32768;
constexpr int FAST_BITS = 9;
constexpr int MAX_W = 16384;                       // two scanlines in LDS

struct BitReader {
  const uint8_t* p; long long n, pos; unsigned long long buf; int cnt; bool over;
  // refill: one aligned 32-bit load per four bytes wherever the position allows (a lone lane pays ~100 cycles per global load, hit or not)
  __device__ __forceinline__ void fill() {
    while (cnt <= 32 && pos + 4 <= n && ((reinterpret_cast<uintptr_t>(p) + (uintptr_t)pos) & 3) == 0) {
      buf |= (unsigned long long)(*reinterpret_cast<const uint32_t*>(p + pos)) << cnt; cnt += 32; pos += 4;
    }
    while (cnt <= 56 && pos < n) { buf |= (unsigned long long)p[pos++] << cnt; cnt += 8; }
  }
  __device__ __forceinline__ unsigned peek(int k) { if (cnt < k) fill(); return (unsigned)(buf & ((1ull << k) - 1ull)); }
  __device__ __forceinline__ void drop(int k) { if (cnt < k) over = true; else { buf >>= k; cnt -= k; } }
  __device__ __forceinline__ unsigned bits(int k) { const unsigned v = peek(k); drop(k); return v; }      // k <= 16
  __device__ __forceinline__ void align() { const int d = cnt & 7; buf >>= d; cnt -= d; }
};

struct Huff {                                      // in LDS
  unsigned short count[16];                        // codes per length
  unsigned short symbol[288];                      // symbols ordered by code
  unsigned short fast[1 << FAST_BITS];             // (symbol << 4) | length for codes of <= FAST_BITS bits, 0 = longer code (or invalid)
};

// canonical Huffman code from code lengths; returns 0 for a complete code, > 0 incomplete, < 0 over-subscribed
__device__ int huff_build(Huff& H, const unsigned char* len, int n) {
  for (int i = 0; i < 16; ++i) H.count[i] = 0;
  for (int i = 0; i < n; ++i) H.count[len[i]]++;
  for (int i = 0; i < (1 << FAST_BITS); ++i) H.fast[i] = 0;
  if (H.count[0] == n) return 0;
  int left = 1;
  for (int l = 1; l < 16; ++l) { left <<= 1; left -= H.count[l]; if (left < 0) return left; }
  unsigned short offs[16], code[16];
  offs[1] = 0; code[1] = 0;
  for (int l = 1; l < 15; ++l) { offs[l + 1] = offs[l] + H.count[l]; code[l + 1] = (unsigned short)((code[l] + H.count[l]) << 1); }
  for (int i = 0; i < n; ++i) {
    const int l = len[i];
    if (!l) continue;
    H.symbol[offs[l]++] = (unsigned short)i;
    const unsigned c = code[l]++;                  // the code, most significant bit first; the stream delivers it first bit = lowest bit
    if (l <= FAST_BITS) {
      const unsigned rev = __brev(c) >> (32 - l);
      for (unsigned j = rev; j < (1u << FAST_BITS); j += 1u << l) H.fast[j] = (unsigned short)((i << 4) | l);
    }
  }
  return left;
}
__device__ __forceinline__ int huff_decode(BitReader& R, const Huff& H) {
  const unsigned e = H.fast[R.peek(FAST_BITS)];
  if (e) { R.drop((int)(e & 15u)); return R.over ? -1 : (int)(e >> 4); }
  int code = 0, first = 0, index = 0;              // longer than FAST_BITS (or the stream's last bits): bit by bit
  for (int l = 1; l < 16; ++l) {
    code |= (int)R.bits(1);
    if (R.over) return -1;
    const int cnt = H.count[l];
    if (code - cnt < first) return H.symbol[index + (code - first)];
    index += cnt; first += cnt; first <<= 1; code <<= 1;
  }
  return -1;
}

__device__ const unsigned short LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const unsigned char LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const unsigned short DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const unsigned char DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const unsigned char CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct InflateLds {
  Huff HL, HD;
  unsigned char lens[320];
  unsigned short len_base[29], dist_base[30];
  unsigned char len_extra[29], dist_extra[30];
  uint8_t win[WIN];
};

// status: 0 ok; 1 bad zlib header; 2 bad block type / stored length; 3 bad Huffman table; 4 bad symbol / distance; 5 output size mismatch;
//         6 input exhausted; 7 Adler-32 mismatch; 8 bad filter type; 9 image too wide for the device path
__device__ int inflate_image(InflateLds& S, const uint8_t* z, long long zn, uint8_t* raw, long long want) {
  if (zn < 6) return 6;
  const unsigned cmf = z[0], flg = z[1];
  if ((cmf & 15u) != 8u || ((cmf << 8) | flg) % 31u != 0u || (flg & 32u)) return 1;
  BitReader R{z + 2, zn - 6, 0, 0ull, 0, false};
  long long out = 0;
  unsigned char* lens = S.lens;
  for (;;) {
    const unsigned last = R.bits(1), type = R.bits(2);
    if (R.over) return 6;
    if (type == 0) {
      R.align();
      const unsigned len = R.bits(16), nlen = R.bits(16);
      if (R.over) return 6;
      if ((len ^ nlen) != 0xFFFFu) return 2;
      if (out + len > want) return 5;
      for (unsigned i = 0; i < len; ++i) {
        const unsigned b = R.bits(8);
        if (R.over) return 6;
        S.win[out & (WIN - 1)] = (uint8_t)b; raw[out++] = (uint8_t)b;
      }
    } else if (type == 1 || type == 2) {
      if (type == 1) {
        for (int i = 0; i < 144; ++i) lens[i] = 8;
        for (int i = 144; i < 256; ++i) lens[i] = 9;
        for (int i = 256; i < 280; ++i) lens[i] = 7;
        for (int i = 280; i < 288; ++i) lens[i] = 8;
        huff_build(S.HL, lens, 288);
        for (int i = 0; i < 30; ++i) lens[i] = 5;
        huff_build(S.HD, lens, 30);
      } else {
        const int nlen = (int)R.bits(5) + 257, ndist = (int)R.bits(5) + 1, ncode = (int)R.bits(4) + 4;
        if (R.over) return 6;
        if (nlen > 286 || ndist > 30) return 3;
        for (int i = 0; i < 19; ++i) lens[i] = 0;
        for (int i = 0; i < ncode; ++i) lens[CL_ORDER[i]] = (unsigned char)R.bits(3);
        if (R.over) return 6;
        if (huff_build(S.HL, lens, 19) != 0) return 3;
        int idx = 0;
        while (idx < nlen + ndist) {
          const int sym = huff_decode(R, S.HL);
          if (sym < 0) return R.over ? 6 : 3;
          if (sym < 16) lens[idx++] = (unsigned char)sym;
          else {
            int prev = 0, rep;
            if (sym == 16) { if (idx == 0) return 3; prev = lens[idx - 1]; rep = 3 + (int)R.bits(2); }
            else if (sym == 17) rep = 3 + (int)R.bits(3);
            else rep = 11 + (int)R.bits(7);
            if (R.over) return 6;
            if (idx + rep > nlen + ndist) return 3;
            while (rep--) lens[idx++] = (unsigned char)prev;
          }
        }
        if (lens[256] == 0) return 3;
        int e = huff_build(S.HD, lens + nlen, ndist);          // distances first: the literal / length build below reuses nothing of `lens`
        if (e < 0 || (e > 0 && ndist - S.HD.count[0] != 1)) return 3;
        e = huff_build(S.HL, lens, nlen);
        if (e < 0 || (e > 0 && nlen - S.HL.count[0] != 1)) return 3;
      }
      for (;;) {
        int sym = huff_decode(R, S.HL);
        if (sym < 256) {
          if (sym < 0) return R.over ? 6 : 4;
          if (out >= want) return 5;
          S.win[out & (WIN - 1)] = (uint8_t)sym; raw[out++] = (uint8_t)sym;
        } else if (sym == 256) break;
        else {
          sym -= 257;
          if (sym >= 29) return 4;
          const int len = S.len_base[sym] + (int)R.bits(S.len_extra[sym]);
          const int ds = huff_decode(R, S.HD);
          if (ds < 0 || ds >= 30) return R.over ? 6 : 4;
          const long long dist = S.dist_base[ds] + (long long)R.bits(S.dist_extra[ds]);
          if (R.over) return 6;
          if (dist > out) return 4;
          if (out + len > want) return 5;
          for (int i = 0; i < len; ++i) {
            const uint8_t v = S.win[(out - dist) & (WIN - 1)];
            S.win[out & (WIN - 1)] = v; raw[out++] = v;
          }
        }
      }
    } else return 2;
    if (last) break;
  }
  return out == want ? 0 : 5;
}

__global__ __launch_bounds__(64) void png_decode_kernel(const uint8_t* __restrict__ z, const long long* __restrict__ offs, int n, int h, int w,
                                                        uint8_t* __restrict__ raw, float* __restrict__ outp, float divisor, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  InflateLds& S = *reinterpret_cast<InflateLds*>(dsm);
  uint8_t* rows = reinterpret_cast<uint8_t*>(dsm + sizeof(InflateLds));           // [2][w]
  __shared__ int st_sh;
  __shared__ unsigned long long ra[64], rb[64];
  const int img = blockIdx.x, lane = threadIdx.x;
  const long long want = (long long)h * (w + 1);
  uint8_t* r = raw + (long long)img * want;
  const uint8_t* zp = z + offs[img];
  const long long zn = offs[img + 1] - offs[img];
  if (lane < 29) { S.len_base[lane] = LEN_BASE[lane]; S.len_extra[lane] = LEN_EXTRA[lane]; }
  if (lane < 30) { S.dist_base[lane] = DIST_BASE[lane]; S.dist_extra[lane] = DIST_EXTRA[lane]; }
  __syncthreads();
  if (lane == 0) st_sh = w > MAX_W ? 9 : inflate_image(S, zp, zn, r, want);
  __threadfence_block();
  __syncthreads();
  int st = st_sh;
  if (st != 0) { if (lane == 0) status[img] = st; return; }
  // Adler-32 of the inflated scanlines: A = 1 + sum d_i, B = n + sum (n - i) d_i
  unsigned long long a = 0, b = 0;
  for (long long i = lane; i < want; i += 64) { const unsigned d = r[i]; a += d; b += (unsigned long long)(want - i) * d; }
  ra[lane] = a; rb[lane] = b % ADLER_MOD;
  __syncthreads();
  if (lane == 0) {
    unsigned long long A = 1, B = (unsigned long long)(want % ADLER_MOD);
    for (int k = 0; k < 64; ++k) { A += ra[k]; B += rb[k]; }
    A %= ADLER_MOD; B %= ADLER_MOD;
    const uint8_t* t = zp + zn - 4;
    const unsigned adler = ((unsigned)t[0] << 24) | ((unsigned)t[1] << 16) | ((unsigned)t[2] << 8) | t[3];
    st_sh = adler == (unsigned)((B << 16) | A) ? 0 : 7;
  }
  __syncthreads();
  st = st_sh;
  if (st != 0) { if (lane == 0) status[img] = st; return; }
  // scanline filters (PNG specification, section 6; one byte per pixel), rows[cur] / rows[prev] in LDS
  float* op = outp + (long long)img * h * w;
  const int chunk = (w + 63) / 64;
  __shared__ unsigned tot[64];
  for (int y = 0; y < h; ++y) {
    uint8_t* cur = rows + (y & 1) * w;
    const uint8_t* up = rows + ((y & 1) ^ 1) * w;
    const uint8_t* src = r + (long long)y * (w + 1);
    const int ft = src[0];
    if (ft > 4) { if (lane == 0) status[img] = 8; return; }
    for (int x = lane; x < w; x += 64) cur[x] = src[1 + x];
    __syncthreads();
    if (ft == 2) {
      if (y) for (int x = lane; x < w; x += 64) cur[x] = (uint8_t)(cur[x] + up[x]);
    } else if (ft == 1) {                             // Sub: prefix sum modulo 256 -- per-lane runs, a wave-wide scan of their totals
      const int x0 = lane * chunk, x1 = x0 + chunk < w ? x0 + chunk : w;
      unsigned sum = 0;
      for (int x = x0; x < x1; ++x) { sum += cur[x]; cur[x] = (uint8_t)sum; }
      tot[lane] = sum;
      __syncthreads();
      unsigned before = 0;
      for (int k = 0; k < lane; ++k) before += tot[k];
      for (int x = x0; x < x1; ++x) cur[x] = (uint8_t)(cur[x] + before);
    } else if (ft == 3 || ft == 4) {
      if (lane == 0) {
        int left = 0, ul = 0;
        for (int x = 0; x < w; ++x) {
          const int bb = y ? up[x] : 0;
          int pred;
          if (ft == 3) pred = (left + bb) >> 1;
          else {
            const int p = left + bb - ul, pa = p > left ? p - left : left - p, pb = p > bb ? p - bb : bb - p, pc = p > ul ? p - ul : ul - p;
            pred = (pa <= pb && pa <= pc) ? left : (pb <= pc ? bb : ul);
          }
          left = (cur[x] + pred) & 255; ul = bb;
          cur[x] = (uint8_t)left;
        }
      }
    }
    __syncthreads();
    for (int x = lane; x < w; x += 64) op[(long long)y * w + x] = (float)cur[x] / divisor;      // IEEE division, as numpy's float32 / 255.
    __syncthreads();
  }
  if (lane == 0) status[img] = 0;
}

PngGeo png_geo(int n, int h, int w, long long img_stride, int row_stride, long long out_stride) {
  PngGeo G;
  G.n = n; G.h = h; G.w = w; G.img_stride = img_stride; G.row_stride = row_stride; G.out_stride = out_stride;
  G.raw = (long long)h * (w + 1);
  G.nblk = (int)((G.raw + STORED_MAX - 1) / STORED_MAX);
  G.zlen = 2 + 5ll * G.nblk + G.raw + 4;
  G.total = 57 + G.zlen;
  G.clen = 4 + G.zlen;
  G.nseg = (int)((G.clen + CRC_SEG - 1) / CRC_SEG);
  G.per = (G.nseg + 255) / 256;
  G.x_seg = gf2_x8n(CRC_SEG);
  G.x_len = gf2_x8n((unsigned long long)G.clen);
  G.x_lvl[0] = gf2_x8n((unsigned long long)G.per * CRC_SEG);
  for (int k = 1; k < 8; ++k) G.x_lvl[k] = gf2_mulmod(G.x_lvl[k - 1], G.x_lvl[k - 1]);
  return G;
}

}  // namespace
}  // namespace gpemsr

using namespace gpemsr;

extern "C" int64_t gpemsr_png_gray8_size(int h, int w) {
  if (h <= 0 || w <= 0) return -1;
  return png_geo(1, h, w, 0, w, 0).total;
}

extern "C" int64_t gpemsr_png_encode_workspace(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0) return -1;
  const PngGeo G = png_geo(n, h, w, 0, w, 0);
  const long long na = (G.raw + ADLER_SEG - 1) / ADLER_SEG;
  return (int64_t)n * (na * 16 + (long long)G.nseg * 4);
}

extern "C" int gpemsr_png_encode_gray8(const uint8_t* img, int n, int h, int w, int64_t img_stride, int row_stride, uint8_t* out, int64_t out_stride,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
  GP_REQUIRE(img && out && workspace && n > 0 && h > 0 && w > 0, "png_encode_gray8: null pointer or empty image");
  GP_REQUIRE(row_stride >= w && img_stride >= 0, "png_encode_gray8: bad strides");
  const PngGeo G = png_geo(n, h, w, img_stride, row_stride, out_stride);
  GP_REQUIRE(out_stride >= G.total, "png_encode_gray8: out_stride %lld < file size %lld", (long long)out_stride, G.total);
  GP_REQUIRE(G.zlen < (1ll << 31), "png_encode_gray8: image too large for one IDAT chunk");
  GP_REQUIRE(workspace_bytes >= gpemsr_png_encode_workspace(n, h, w) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "png_encode_gray8: workspace too small or misaligned");
  const int na = (int)((G.raw + ADLER_SEG - 1) / ADLER_SEG);
  unsigned long long* apart = reinterpret_cast<unsigned long long*>(workspace);
  uint32_t* cpart = reinterpret_cast<uint32_t*>(apart + (long long)n * na * 2);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned gx = (unsigned)((G.total + 4095) / 4096);
  hipLaunchKernelGGL(png_assemble_kernel, dim3(gx, n), dim3(256), 0, st, G, img, out);
  hipLaunchKernelGGL(png_adler_partial_kernel, dim3(na, n), dim3(256), 0, st, G, img, apart, na);
  hipLaunchKernelGGL(png_crc_partial_kernel, dim3((G.nseg + 255) / 256, n), dim3(256), 0, st, G, out, cpart);
  hipLaunchKernelGGL(png_final_kernel, dim3(n), dim3(256), 0, st, G, apart, na, cpart, out);
  return check_launch("png_encode_gray8");
}

extern "C" int gpemsr_png_decode_gray8(const uint8_t* idat, const int64_t* offsets, int n, int h, int w, uint8_t* raw, float* out, float divisor,
                                       int32_t* status, void* stream) {
  GP_REQUIRE(idat && offsets && raw && out && status && n > 0 && h > 0 && w > 0, "png_decode_gray8: null pointer or empty image");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = sizeof(InflateLds) + 2 * (size_t)(w <= MAX_W ? w : 1);
  static dev_once_t attr{0};
  if (dev_once_begin(attr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(png_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
      return fail(GPEMSR_ELAUNCH, "png_decode_gray8: cannot raise the dynamic LDS limit");
    dev_once_done(attr);
  }
  hipLaunchKernelGGL(png_decode_kernel, dim3(n), dim3(64), lds, st, idat, reinterpret_cast<const long long*>(offsets), n, h, w, raw, out, divisor, status);
  return check_launch("png_decode_gray8");
}
