// PNG encoder with a COMPRESSED zlib stream, on the device (R:output_GPEMSR.py:95 `cv2.imwrite`): one dynamic-Huffman deflate
// block of literals per image.  8-bit EM slices are noise over structure -- LZ77 matches are rare, almost all of what zlib gains on them is the
// entropy code, and an entropy code is embarrassingly parallel once the code lengths are known:
//   1. histograms of the scanline bytes for filter type 0 (None) and type 1 (Sub), LDS atomics;
//   2. one workgroup per image: picks the filter with the lower order-0 entropy, builds the length-limited (15 bit) canonical Huffman code
//      (parallel rank sort of the 257 symbols, the two-queue merge and zlib's overflow rule by one lane), the code-length code (7 bit), and
//      writes the block header bits;
//   3. code lengths summed per 4,096-symbol block, and
//   4. every thread writes its 16 codes at bit offset (header + blocks before + symbols before) with atomic ORs into the zeroed stream;
//   5. Adler-32 of the filtered scanlines and CRC-32 of the chunk as in png.hip (partial sums / raw remainders folded by a tree); the
//      chunk length, and with it the file size, is known only now: the sizes go back to the host beside the files.
// The stored-block encoder of png.hip stays the default (0.12 ms per 16 slices, fixed file size); this one trades ~3x the time for files
// of the size the reference's zlib produces.
#include "png_common.h"

namespace gpemsr {
namespace {
using namespace png;

constexpr int HSEG = 4096;                         // symbols per pack block: 256 threads x 16
constexpr int NSYM = 257;                          // literals + end-of-block
constexpr int HDR_BYTES = 512;                     // block header (3 + 14 + 57 + 258 x <= 7 bits = 1,880 bits at most)

struct HuffGeo {
  int n, h, w;
  long long img_stride; int row_stride;
  long long out_stride;
  long long raw;                                   // h * (w + 1) scanline bytes
  int nblk;                                        // pack blocks
  int na;                                          // Adler segments
  int nseg, per;                                   // CRC segments of the LONGEST possible chunk, segments per lane of the tree
  uint32_t x_seg, x_lvl[8];
};

// per-image scratch (device): everything the kernels hand to one another
struct HuffWork {
  unsigned hist[2][256];
  unsigned code[NSYM];                             // (length << 16) | code bits, first-transmitted bit lowest
  int filter;                                      // 0 None, 1 Sub
  unsigned hdr_bits;
  unsigned char hdr[HDR_BYTES];
  unsigned long long total_bits;                   // header + symbols + end-of-block
  long long clen;                                  // chunk type + data bytes
};

__device__ __forceinline__ unsigned scan_sym(const HuffGeo& G, const uint8_t* ip, long long r, int filter) {
  const int row = (int)(r / (G.w + 1)), col = (int)(r - (long long)row * (G.w + 1)) - 1;
  if (col < 0) return (unsigned)filter;
  const uint8_t* p = ip + (long long)row * G.row_stride + col;
  const unsigned v = p[0];
  return (filter && col > 0) ? ((v - p[-1]) & 255u) : v;
}

__global__ __launch_bounds__(256) void huff_hist_kernel(HuffGeo G, const uint8_t* __restrict__ src, HuffWork* __restrict__ W) {
  __shared__ unsigned hs[2][256];
  const int img = blockIdx.y, t = threadIdx.x;
  hs[0][t] = 0; hs[1][t] = 0;
  __syncthreads();
  const uint8_t* ip = src + (long long)img * G.img_stride;
  const long long r0 = (long long)blockIdx.x * HSEG;
  for (int j = t; j < HSEG; j += 256) {
    const long long r = r0 + j;
    if (r >= G.raw) break;
    atomicAdd(&hs[0][scan_sym(G, ip, r, 0)], 1u);
    atomicAdd(&hs[1][scan_sym(G, ip, r, 1)], 1u);
  }
  __syncthreads();
  if (hs[0][t]) atomicAdd(&W[img].hist[0][t], hs[0][t]);
  if (hs[1][t]) atomicAdd(&W[img].hist[1][t], hs[1][t]);
}

// ---- length-limited canonical Huffman code over n symbols (n <= 257), all in LDS; lane 0 does the serial parts -------------------------
struct HuffLds {
  unsigned cnt[NSYM];                              // symbol counts
  unsigned short order[NSYM];                      // used symbols, ascending (count, symbol)
  unsigned wt[2 * NSYM];                           // node weights: leaves (sorted) then internal nodes
  unsigned short parent[2 * NSYM];
  unsigned char depth[2 * NSYM];
  unsigned char len[NSYM];                         // result: code length per symbol (0 = unused)
  unsigned code[NSYM];                             // result: (length << 16) | bit-reversed code
  int nused;
};

// all threads of the workgroup call this; n symbols with counts in S.cnt; maxlen 15 or 7
__device__ void build_code(HuffLds& S, int n, int maxlen) {
  const int t = threadIdx.x;
  for (int s = t; s < n; s += blockDim.x) {
    S.len[s] = 0;
    const unsigned c = S.cnt[s];
    if (!c) continue;
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const unsigned cj = S.cnt[j];
      rank += (cj && (cj < c || (cj == c && j < s))) ? 1 : 0;
    }
    S.order[rank] = (unsigned short)s;
  }
  if (t == 0) { int u = 0; for (int s = 0; s < n; ++s) u += S.cnt[s] ? 1 : 0; S.nused = u; }
  __syncthreads();
  if (t == 0) {
    const int L = S.nused;
    if (L == 1) S.len[S.order[0]] = 1;
    else if (L > 1) {
      for (int i = 0; i < L; ++i) S.wt[i] = S.cnt[S.order[i]];
      int i = 0, j = L, k = L;                      // next leaf, next unmerged internal node, next free node
      for (int m = 0; m < L - 1; ++m) {
        int a, b;
        if (i < L && (j >= k || S.wt[i] <= S.wt[j])) a = i++; else a = j++;
        if (i < L && (j >= k || S.wt[i] <= S.wt[j])) b = i++; else b = j++;
        S.wt[k] = S.wt[a] + S.wt[b];
        S.parent[a] = (unsigned short)k; S.parent[b] = (unsigned short)k;
        ++k;
      }
      S.depth[2 * L - 2] = 0;
      for (int v = 2 * L - 3; v >= 0; --v) S.depth[v] = (unsigned char)(S.depth[S.parent[v]] + 1);
      // zlib's gen_bitlen rule for codes longer than maxlen: clamp, then move leaves down from the deepest non-full level until the code is
      // complete again; the longest lengths go to the rarest symbols
      int bl[16];
      for (int b = 0; b < 16; ++b) bl[b] = 0;
      int overflow = 0;                             // zlib counts EVERY node below the limit, internal ones too (their clamped parents put them there)
      for (int v = 0; v < 2 * L - 2; ++v) overflow += S.depth[v] > maxlen ? 1 : 0;
      for (int v = 0; v < L; ++v) { const int d = S.depth[v]; bl[d > maxlen ? maxlen : d]++; }
      while (overflow > 0) {
        int bits = maxlen - 1;
        while (bl[bits] == 0) --bits;
        bl[bits]--; bl[bits + 1] += 2; bl[maxlen]--;
        overflow -= 2;
      }
      int v = 0;                                    // leaves in ascending frequency: the rarest get the longest codes
      for (int bits = maxlen; bits >= 1; --bits)
        for (int c2 = bl[bits]; c2 > 0; --c2) S.len[S.order[v++]] = (unsigned char)bits;
    }
    // canonical codes in symbol order; stored bit-reversed (a Huffman code enters the stream most significant bit first)
    int blc[16], next[16];
    for (int b = 0; b < 16; ++b) blc[b] = 0;
    for (int s = 0; s < n; ++s) blc[S.len[s]]++;
    blc[0] = 0;
    int code = 0;
    for (int b = 1; b < 16; ++b) { code = (code + blc[b - 1]) << 1; next[b] = code; }
    for (int s = 0; s < n; ++s) {
      const int l = S.len[s];
      S.code[s] = l ? (((unsigned)l << 16) | (__brev((unsigned)next[l]++) >> (32 - l))) : 0u;
    }
  }
  __syncthreads();
}

struct BitOut {
  unsigned char* p; unsigned nbits; unsigned long long acc; int cnt;
  __device__ void put(unsigned v, int k) {
    acc |= (unsigned long long)v << cnt; cnt += k; nbits += k;
    while (cnt >= 8) { *p++ = (unsigned char)acc; acc >>= 8; cnt -= 8; }
  }
  __device__ void flush() { if (cnt > 0) { *p++ = (unsigned char)acc; acc = 0; cnt = 0; } }
};

__device__ const unsigned char HCL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__global__ __launch_bounds__(256) void huff_build_kernel(HuffGeo G, HuffWork* __restrict__ W) {
  __shared__ HuffLds S;
  __shared__ float ent[2][256];
  __shared__ unsigned char litlen[NSYM + 1];
  __shared__ unsigned hdrw[HDR_BYTES / 4];
  const int img = blockIdx.x, t = threadIdx.x;
  HuffWork& w = W[img];
  // filter choice: the lower order-0 entropy = the larger sum of c log2 c (both streams have the same length)
  for (int f = 0; f < 2; ++f) { const float c = (float)w.hist[f][t]; ent[f][t] = c > 0.f ? c * log2f(c) : 0.f; }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { ent[0][t] += ent[0][t + o]; ent[1][t] += ent[1][t + o]; }
    __syncthreads();
  }
  const int filt = ent[1][0] > ent[0][0] ? 1 : 0;
  S.cnt[t] = w.hist[filt][t];
  if (t == 0) S.cnt[256] = 1;                         // end-of-block
  __syncthreads();
  build_code(S, NSYM, 15);
  for (int s = t; s < NSYM; s += 256) { w.code[s] = S.code[s]; litlen[s] = S.len[s]; }
  if (t == 0) { litlen[NSYM] = 0; w.filter = filt; }   // the one distance code, of zero bits: no distance codes are used at all (RFC 1951 3.2.7)
  __syncthreads();
  // code-length code over the 258 lengths, each sent as itself (symbols 0..15; no run-length symbols)
  if (t < 19) S.cnt[t] = 0;
  __syncthreads();
  if (t == 0) for (int s = 0; s <= NSYM; ++s) S.cnt[litlen[s]]++;
  __syncthreads();
  build_code(S, 16, 7);
  for (int i = t; i < HDR_BYTES / 4; i += 256) hdrw[i] = 0;
  __syncthreads();
  if (t == 0) {
    BitOut B{reinterpret_cast<unsigned char*>(hdrw), 0u, 0ull, 0};
    B.put(1u, 1); B.put(2u, 2);                       // BFINAL, BTYPE = dynamic Huffman
    B.put(0u, 5); B.put(0u, 5); B.put(15u, 4);        // HLIT = 257 codes, HDIST = 1 code, HCLEN = 19 lengths
    for (int i = 0; i < 19; ++i) { const int s = HCL_ORDER[i]; B.put(s < 16 ? (unsigned)S.len[s] : 0u, 3); }
    for (int s = 0; s <= NSYM; ++s) { const unsigned c = S.code[litlen[s]]; B.put(c & 0xFFFFu, (int)(c >> 16)); }
    w.hdr_bits = B.nbits;
    B.flush();
  }
  __syncthreads();
  for (int i = t; i < HDR_BYTES / 4; i += 256) reinterpret_cast<unsigned*>(w.hdr)[i] = hdrw[i];
}

// bits of every pack block's symbols
__global__ __launch_bounds__(256) void huff_blocksum_kernel(HuffGeo G, const uint8_t* __restrict__ src, const HuffWork* __restrict__ W, unsigned* __restrict__ bsum) {
  __shared__ unsigned char len[256];
  __shared__ unsigned red[256];
  const int img = blockIdx.y, t = threadIdx.x;
  len[t] = (unsigned char)(W[img].code[t] >> 16);
  __syncthreads();
  const int filt = W[img].filter;
  const uint8_t* ip = src + (long long)img * G.img_stride;
  const long long r0 = (long long)blockIdx.x * HSEG + (long long)t * 16;
  unsigned s = 0;
  for (int k = 0; k < 16; ++k) if (r0 + k < G.raw) s += len[scan_sym(G, ip, r0 + k, filt)];
  red[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  if (t == 0) bsum[(long long)img * G.nblk + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void huff_pack_kernel(HuffGeo G, const uint8_t* __restrict__ src, HuffWork* __restrict__ W, const unsigned* __restrict__ bsum,
                                                        uint8_t* __restrict__ out) {
  __shared__ unsigned code[NSYM];
  __shared__ unsigned long long pre[256];
  __shared__ unsigned scan[256];
  const int img = blockIdx.y, t = threadIdx.x, blk = blockIdx.x;
  for (int s = t; s < NSYM; s += 256) code[s] = W[img].code[s];
  // bits in front of this block: header + the blocks before (<= a few hundred: every thread sums a stride of them)
  unsigned long long p = 0;
  for (int b = t; b < blk; b += 256) p += bsum[(long long)img * G.nblk + b];
  pre[t] = p;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) pre[t] += pre[t + o]; __syncthreads(); }
  const unsigned long long base = (unsigned long long)W[img].hdr_bits + pre[0];
  const int filt = W[img].filter;
  const uint8_t* ip = src + (long long)img * G.img_stride;
  const long long r0 = (long long)blk * HSEG + (long long)t * 16;
  unsigned cs[16]; unsigned mine = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { cs[k] = (r0 + k < G.raw) ? code[scan_sym(G, ip, r0 + k, filt)] : 0u; mine += cs[k] >> 16; }
  scan[t] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {                 // inclusive scan of the threads' bit counts
    const unsigned v = t >= o ? scan[t - o] : 0u;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  // the deflate data begins at file byte 43; the file is 16-byte aligned: bit b of the data is bit (43 * 8 + b) of the file's dword array
  unsigned long long bit = 43ull * 8ull + base + (scan[t] - mine);
  unsigned* words = reinterpret_cast<unsigned*>(out + (long long)img * G.out_stride);
  long long wi = (long long)(bit >> 5);
  unsigned long long acc = 0; int cnt = (int)(bit & 31ull);
  const bool last_thread = (blk == G.nblk - 1) && (r0 < G.raw) && (r0 + 16 >= G.raw);
#pragma unroll
  for (int k = 0; k <= 16; ++k) {
    unsigned c;
    if (k < 16) c = cs[k]; else c = last_thread ? code[256] : 0u;       // end-of-block behind the last symbol
    const int l = (int)(c >> 16);
    if (!l) continue;
    acc |= (unsigned long long)(c & 0xFFFFu) << cnt; cnt += l;
    if (cnt >= 32) { atomicOr(&words[wi++], (unsigned)acc); acc >>= 32; cnt -= 32; }
  }
  if (cnt > 0 && acc) atomicOr(&words[wi], (unsigned)acc);
  if (last_thread) W[img].total_bits = base + scan[t] + (code[256] >> 16);
  if (blk == 0) {                                     // the block header (its bits start at data bit 0 = file bit 344)
    const unsigned hb = W[img].hdr_bits, nw = (hb + 31) / 32;
    for (unsigned i = t; i < nw; i += 256) {
      const unsigned long long v = (unsigned long long)reinterpret_cast<const unsigned*>(W[img].hdr)[i] << 24;      // 344 bits = 10 words + 24 bits
      atomicOr(&words[10 + i], (unsigned)v);
      if (v >> 32) atomicOr(&words[11 + i], (unsigned)(v >> 32));
    }
  }
}

// Adler-32 partial sums of the FILTERED scanlines (png.hip's rule: a = sum d, b = sum (len - j) d_j per segment)
__global__ __launch_bounds__(256) void huff_adler_partial_kernel(HuffGeo G, const uint8_t* __restrict__ src, const HuffWork* __restrict__ W,
                                                                 unsigned long long* __restrict__ part) {
  const int img = blockIdx.y, s = blockIdx.x, t = threadIdx.x;
  const long long r0 = (long long)s * ADLER_SEG;
  const int len = G.raw - r0 < ADLER_SEG ? (int)(G.raw - r0) : ADLER_SEG;
  const int filt = W[img].filter;
  const uint8_t* ip = src + (long long)img * G.img_stride;
  unsigned long long a = 0, b = 0;
  for (int j = t; j < len; j += 256) { const unsigned d = scan_sym(G, ip, r0 + j, filt); a += d; b += (unsigned long long)(len - j) * d; }
  __shared__ unsigned long long sa[256], sb[256];
  sa[t] = a; sb[t] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) { sa[t] += sa[t + o]; sb[t] += sb[t + o]; } __syncthreads(); }
  if (t == 0) { part[((long long)img * G.na + s) * 2] = sa[0]; part[((long long)img * G.na + s) * 2 + 1] = sb[0]; }
}

// one workgroup per image: everything around the deflate data except the chunk CRC
__global__ __launch_bounds__(256) void huff_frame_kernel(HuffGeo G, HuffWork* __restrict__ W, const unsigned long long* __restrict__ apart, uint8_t* __restrict__ out) {
  __shared__ uint32_t tab[256];
  __shared__ unsigned long long sa[256], sb[256];
  const int img = blockIdx.x, t = threadIdx.x;
  tab[t] = crc_table_entry(t);
  unsigned long long a = 0, b = 0;
  for (int s = t; s < G.na; s += 256) {
    const long long r0 = (long long)s * ADLER_SEG;
    const long long len = G.raw - r0 < ADLER_SEG ? G.raw - r0 : ADLER_SEG;
    const unsigned long long as = apart[((long long)img * G.na + s) * 2], bs = apart[((long long)img * G.na + s) * 2 + 1];
    a += as;
    b += bs % ADLER_MOD + (as % ADLER_MOD) * ((unsigned long long)(G.raw - r0 - len) % ADLER_MOD);
  }
  sa[t] = a; sb[t] = b % ADLER_MOD;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) { sa[t] += sa[t + o]; sb[t] += sb[t + o]; } __syncthreads(); }
  if (t != 0) return;
  uint8_t* op = out + (long long)img * G.out_stride;
  const unsigned long long A = (1ull + sa[0]) % ADLER_MOD, B = ((unsigned long long)(G.raw % ADLER_MOD) + sb[0]) % ADLER_MOD;
  const uint32_t adler = (uint32_t)((B << 16) | A);
  const long long dbytes = (long long)((W[img].total_bits + 7ull) / 8ull);
  const long long zlen = 2 + dbytes + 4;
  const unsigned char head[16] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A, 0, 0, 0, 13, 'I', 'H', 'D', 'R'};
  for (int i = 0; i < 16; ++i) op[i] = head[i];
  op[16] = (uint8_t)(G.w >> 24); op[17] = (uint8_t)(G.w >> 16); op[18] = (uint8_t)(G.w >> 8); op[19] = (uint8_t)G.w;
  op[20] = (uint8_t)(G.h >> 24); op[21] = (uint8_t)(G.h >> 16); op[22] = (uint8_t)(G.h >> 8); op[23] = (uint8_t)G.h;
  op[24] = 8; op[25] = 0; op[26] = 0; op[27] = 0; op[28] = 0;
  uint32_t hc = 0xFFFFFFFFu;
  for (int i = 12; i < 29; ++i) hc = tab[(hc ^ op[i]) & 255u] ^ (hc >> 8);
  hc ^= 0xFFFFFFFFu;
  op[29] = (uint8_t)(hc >> 24); op[30] = (uint8_t)(hc >> 16); op[31] = (uint8_t)(hc >> 8); op[32] = (uint8_t)hc;
  op[33] = (uint8_t)(zlen >> 24); op[34] = (uint8_t)(zlen >> 16); op[35] = (uint8_t)(zlen >> 8); op[36] = (uint8_t)zlen;
  op[37] = 'I'; op[38] = 'D'; op[39] = 'A'; op[40] = 'T';
  op[41] = 0x78; op[42] = 0x01;
  const long long pa = 43 + dbytes;
  op[pa] = (uint8_t)(adler >> 24); op[pa + 1] = (uint8_t)(adler >> 16); op[pa + 2] = (uint8_t)(adler >> 8); op[pa + 3] = (uint8_t)adler;
  W[img].clen = 4 + zlen;
}

// raw CRC-32 (initial value 0) of the chunk's segments, counted from the END of this image's chunk; the standard CRC's initial value
// 0xFFFFFFFF = the first four message bytes inverted
__global__ __launch_bounds__(256) void huff_crc_partial_kernel(HuffGeo G, const HuffWork* __restrict__ W, const uint8_t* __restrict__ out, uint32_t* __restrict__ part) {
  __shared__ uint32_t tab[256];
  tab[threadIdx.x] = crc_table_entry(threadIdx.x);
  __syncthreads();
  const int img = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
  if (s >= G.nseg) return;
  const long long clen = W[img].clen;
  const long long end = clen - (long long)(G.nseg - 1 - s) * CRC_SEG;
  uint32_t c = 0;
  if (end > 0) {
    const long long beg = end - CRC_SEG > 0 ? end - CRC_SEG : 0;
    const uint8_t* p = out + (long long)img * G.out_stride + 37;
    for (long long m = beg; m < end; ++m) c = tab[(c ^ (p[m] ^ (m < 4 ? 0xFFu : 0u))) & 255u] ^ (c >> 8);
  }
  part[(long long)img * G.nseg + s] = c;
}

__global__ __launch_bounds__(256) void huff_crc_final_kernel(HuffGeo G, const HuffWork* __restrict__ W, const uint32_t* __restrict__ cpart, uint8_t* __restrict__ out,
                                                             long long* __restrict__ sizes) {
  __shared__ uint32_t sc[256];
  const int img = blockIdx.x, t = threadIdx.x;
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
    if ((t & (2 * o - 1)) == 0) sc[t] = gf2_mulmod(G.x_lvl[lv], sc[t]) ^ sc[t + o];
    __syncthreads();
  }
  if (t != 0) return;
  const uint32_t crc = sc[0] ^ 0xFFFFFFFFu;
  const long long clen = W[img].clen;
  uint8_t* op = out + (long long)img * G.out_stride + 37 + clen;
  op[0] = (uint8_t)(crc >> 24); op[1] = (uint8_t)(crc >> 16); op[2] = (uint8_t)(crc >> 8); op[3] = (uint8_t)crc;
  const unsigned char iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
  for (int i = 0; i < 12; ++i) op[4 + i] = iend[i];
  sizes[img] = 37 + clen + 16;
}

HuffGeo huff_geo(int n, int h, int w, long long img_stride, int row_stride, long long out_stride) {
  HuffGeo G;
  G.n = n; G.h = h; G.w = w; G.img_stride = img_stride; G.row_stride = row_stride; G.out_stride = out_stride;
  G.raw = (long long)h * (w + 1);
  G.nblk = (int)((G.raw + HSEG - 1) / HSEG);
  G.na = (int)((G.raw + ADLER_SEG - 1) / ADLER_SEG);
  const long long clen_max = 4 + 2 + (HDR_BYTES + 2 * G.raw + 2) + 4;      // no code is longer than 15 bits
  G.nseg = (int)((clen_max + CRC_SEG - 1) / CRC_SEG);
  G.per = (G.nseg + 255) / 256;
  G.x_seg = gf2_x8n(CRC_SEG);
  G.x_lvl[0] = gf2_x8n((unsigned long long)G.per * CRC_SEG);
  for (int k = 1; k < 8; ++k) G.x_lvl[k] = gf2_mulmod(G.x_lvl[k - 1], G.x_lvl[k - 1]);
  return G;
}

long long huff_capacity(long long raw) { return (37 + 4 + 2 + HDR_BYTES + 2 * raw + 2 + 4 + 16 + 15) / 16 * 16; }

}  // namespace
}  // namespace gpemsr

using namespace gpemsr;

extern "C" int64_t gpemsr_png_huff_capacity(int h, int w) {
  if (h <= 0 || w <= 0) return -1;
  return huff_capacity((long long)h * (w + 1));
}

extern "C" int64_t gpemsr_png_huff_workspace(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0) return -1;
  const HuffGeo G = huff_geo(n, h, w, 0, w, 0);
  long long per = (long long)sizeof(HuffWork) + 8;
  per = (per + 15) / 16 * 16;
  return (int64_t)n * (per + (long long)G.nblk * 4 + (long long)G.na * 16 + (long long)G.nseg * 4 + 64);
}

extern "C" int gpemsr_png_encode_gray8_huff(const uint8_t* img, int n, int h, int w, int64_t img_stride, int row_stride, uint8_t* out, int64_t out_stride,
                                            int64_t* sizes, void* workspace, int64_t workspace_bytes, void* stream) {
  GP_REQUIRE(img && out && sizes && workspace && n > 0 && h > 0 && w > 0, "png_encode_gray8_huff: null pointer or empty image");
  GP_REQUIRE(row_stride >= w && img_stride >= 0, "png_encode_gray8_huff: bad strides");
  const HuffGeo G = huff_geo(n, h, w, img_stride, row_stride, out_stride);
  GP_REQUIRE(out_stride >= huff_capacity(G.raw) && out_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
             "png_encode_gray8_huff: out_stride must be >= gpemsr_png_huff_capacity and a multiple of 16, out 16-byte aligned");
  GP_REQUIRE(2 * G.raw < (1ll << 31), "png_encode_gray8_huff: image too large for one IDAT chunk");
  GP_REQUIRE(workspace_bytes >= gpemsr_png_huff_workspace(n, h, w) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
             "png_encode_gray8_huff: workspace too small or misaligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  HuffWork* W = reinterpret_cast<HuffWork*>(workspace);
  char* wp = reinterpret_cast<char*>(workspace) + ((long long)n * (long long)sizeof(HuffWork) + 15) / 16 * 16;
  unsigned long long* apart = reinterpret_cast<unsigned long long*>(wp); wp += (long long)n * G.na * 16;
  unsigned* bsum = reinterpret_cast<unsigned*>(wp); wp += ((long long)n * G.nblk * 4 + 15) / 16 * 16;
  uint32_t* cpart = reinterpret_cast<uint32_t*>(wp);
  if (hipMemsetAsync(W, 0, (size_t)n * sizeof(HuffWork), st) != hipSuccess || hipMemsetAsync(out, 0, (size_t)n * (size_t)out_stride, st) != hipSuccess)
    return fail(GPEMSR_ELAUNCH, "png_encode_gray8_huff: memset failed");
  hipLaunchKernelGGL(huff_hist_kernel, dim3(G.nblk, n), dim3(256), 0, st, G, img, W);
  hipLaunchKernelGGL(huff_build_kernel, dim3(n), dim3(256), 0, st, G, W);
  hipLaunchKernelGGL(huff_blocksum_kernel, dim3(G.nblk, n), dim3(256), 0, st, G, img, W, bsum);
  hipLaunchKernelGGL(huff_pack_kernel, dim3(G.nblk, n), dim3(256), 0, st, G, img, W, bsum, out);
  hipLaunchKernelGGL(huff_adler_partial_kernel, dim3(G.na, n), dim3(256), 0, st, G, img, W, apart);
  hipLaunchKernelGGL(huff_frame_kernel, dim3(n), dim3(256), 0, st, G, W, apart, out);
  hipLaunchKernelGGL(huff_crc_partial_kernel, dim3((G.nseg + 255) / 256, n), dim3(256), 0, st, G, W, out, cpart);
  hipLaunchKernelGGL(huff_crc_final_kernel, dim3(n), dim3(256), 0, st, G, W, cpart, out, reinterpret_cast<long long*>(sizes));
  return check_launch("png_encode_gray8_huff");
}
