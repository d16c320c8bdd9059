// Block codecs of the GeoTIFF side of the zonal loop (host code, no GPU): TIFF-flavoured LZW and the horizontal
// differencing predictor.
//
// The reference reaches these through rasterio / GDAL / libtiff: windowed reads of the input rasters
// (flair_zonal_detection/dataset.py:89-117) and the LZW-compressed uint8 window writes of the prediction rasters
// (flair_zonal_detection/inference.py:157-208 profile "compress": "lzw", :342-352 dst.write).  GDAL is absent from
// the build image, so flair_zonal_detection/geotiff.py parses / emits the TIFF structure itself and calls in here
// for the per-block byte work (a 25 000 x 25 000 px prediction raster is 9 600 blocks of 64 KiB).
//
// Bit-stream conventions are libtiff's ("new style" LZW, TIFF 6.0 section 13): codes are packed MSB first, start
// at 9 bits, ClearCode = 256, EndOfInformation = 257, first free code 258, and the code width grows one code EARLY
// (when the next free code reaches 2^bits - 1).  The encoder resets its table at 4094 entries, as libtiff does.
#include <stdint.h>
#include <string.h>

#include "ffa_common_host.h"

namespace {

constexpr int kClear = 256;
constexpr int kEoi = 257;
constexpr int kFirst = 258;
constexpr int kMaxBits = 12;
constexpr int kTable = 1 << kMaxBits;

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc = 0;
  int have = 0;
  // next `bits`-wide code (bits <= 12), or -1 when the stream is exhausted
  inline int get(int bits) {
    if (have < bits) {
      if (end - p >= 8) {  // refill up to 56 bits with one unaligned big-endian load
        uint64_t w;
        memcpy(&w, p, 8);
        w = __builtin_bswap64(w);
        const int take = (64 - have) >> 3;  // whole bytes that fit
        acc = (take == 8) ? w : ((acc << (8 * take)) | (w >> (64 - 8 * take)));
        have += 8 * take;
        p += take;
      } else {
        while (have < bits) {
          if (p == end) return -1;
          acc = (acc << 8) | *p++;
          have += 8;
        }
      }
    }
    have -= bits;
    return (int)((acc >> have) & ((1u << bits) - 1));
  }
};

struct BitWriter {
  uint8_t* p;
  uint8_t* end;
  uint64_t acc = 0;
  int have = 0;
  bool overflow = false;
  void put(int code, int bits) {
    acc = (acc << bits) | (uint32_t)code;
    have += bits;
    while (have >= 8) {
      have -= 8;
      if (p == end) {
        overflow = true;
        return;
      }
      *p++ = (uint8_t)(acc >> have);
    }
  }
  void flush() {
    if (have > 0) {
      if (p == end) {
        overflow = true;
        return;
      }
      *p++ = (uint8_t)(acc << (8 - have));
      have = 0;
    }
  }
};

}  // namespace

// Worst-case size of ffa_tiff_lzw_encode's output for n input bytes (every byte its own 12-bit code, plus the
// clear codes and the end marker).
extern "C" long long ffa_tiff_lzw_bound(long long n) { return n + n / 2 + n / 1024 + 16; }

// Decodes one LZW strip / tile.  Stops at EndOfInformation, at the end of the input, or when `cap` bytes have been
// produced (writers pad blocks, readers know the decoded size).  Returns the number of bytes written, or
// FFA_ERR_ARG for a corrupt stream.
extern "C" long long ffa_tiff_lzw_decode(const uint8_t* src, long long n, uint8_t* dst, long long cap) {
  if (!src || !dst || n < 0 || cap < 0) {
    ffa_set_error("lzw_decode: bad arguments");
    return FFA_ERR_ARG;
  }
  if (n >= 2 && src[0] == 0 && (src[1] & 1)) {
    ffa_set_error("lzw_decode: old-style (LSB-first) LZW streams are not supported");
    return FFA_ERR_UNSUPPORTED;
  }
  // The table holds no strings, only where in the OUTPUT each one was first written: the entry created after
  // emitting string(old) is string(old) + first byte of the next string, and those bytes sit contiguously in dst
  // starting at the position string(old) was emitted at.  Decoding a code is one short copy within dst.
  // (thread-local storage is resolved ONCE into plain pointers: in a -fPIC library every direct access to a
  // thread_local array is a call to __tls_get_addr)
  static thread_local uint32_t tl_where[kTable];
  static thread_local uint16_t tl_length[kTable];
  uint32_t* const where = tl_where;
  uint16_t* const length = tl_length;
  BitReader br{src, src + n};
  int bits = 9, next = kFirst;
  bool have_prev = false;
  long long out = 0, prev_pos = 0;
  int prev_len = 0;
  while (out < cap) {
    const int code = br.get(bits);
    if (code < 0 || code == kEoi) break;
    if (code == kClear) {
      bits = 9;
      next = kFirst;
      have_prev = false;
      continue;
    }
    const long long pos = out;
    int len;
    if (code < 256) {
      dst[out++] = (uint8_t)code;
      len = 1;
    } else if (!have_prev) {  // first code after a clear must be a literal
      ffa_set_error("lzw_decode: corrupt stream (code %d after clear)", code);
      return FFA_ERR_ARG;
    } else if (code < next) {
      len = length[code];
      const uint8_t* from = dst + where[code];  // written earlier in full (where + len <= out): no overlap
      const long long m = (out + len <= cap) ? len : cap - out;
      if (out + 16 <= cap && len <= 16 && where[code] + 16 <= out) {
        // short string away from the write position and from the end of the buffer: two blind 8-byte moves (the
        // bytes past `len` are overwritten by the next code)
        uint64_t a, b;
        memcpy(&a, from, 8);
        memcpy(&b, from + 8, 8);
        memcpy(dst + out, &a, 8);
        memcpy(dst + out + 8, &b, 8);
      } else if (m <= 16) {
        for (long long i = 0; i < m; ++i) dst[out + i] = from[i];
      } else {
        memcpy(dst + out, from, (size_t)m);
      }
      out += m;
    } else if (code == next && next < kTable) {
      // string(old) + its own first byte: the source run ends on the byte being written -> forward byte copy
      len = prev_len + 1;
      const uint8_t* from = dst + prev_pos;
      const long long m = (out + len <= cap) ? len : cap - out;
      for (long long i = 0; i < m; ++i) dst[out + i] = from[i];
      out += m;
    } else {
      ffa_set_error("lzw_decode: corrupt stream (code %d, table holds %d)", code, next);
      return FFA_ERR_ARG;
    }
    if (have_prev && next < kTable) {
      where[next] = (uint32_t)prev_pos;
      length[next] = (uint16_t)(prev_len + 1);
      ++next;
      if (next >= (1 << bits) - 1 && bits < kMaxBits) ++bits;
    }
    have_prev = true;
    prev_pos = pos;
    prev_len = len;
  }
  return out;
}

// Encodes n bytes as one LZW block.  Returns the encoded size, or FFA_ERR_WORKSPACE when `cap` is too small
// (ffa_tiff_lzw_bound(n) always suffices).
extern "C" long long ffa_tiff_lzw_encode(const uint8_t* src, long long n, uint8_t* dst, long long cap) {
  if (!src || !dst || n < 0 || cap < 0) {
    ffa_set_error("lzw_encode: bad arguments");
    return FFA_ERR_ARG;
  }
  // open-addressing hash (prefix code, byte) -> code; a generation stamp makes the table reset free
  constexpr int kHash = 1 << 14;
  static thread_local uint32_t tl_hkey[kHash];
  static thread_local uint16_t tl_hval[kHash];
  static thread_local uint32_t tl_hgen[kHash];
  static thread_local uint32_t tl_gen = 0;
  uint32_t* const hkey = tl_hkey;  // resolved once, see ffa_tiff_lzw_decode
  uint16_t* const hval = tl_hval;
  uint32_t* const hgen = tl_hgen;
  uint32_t gen = tl_gen;
  auto reset = [&]() {
    if (++gen == 0) {
      memset(hgen, 0, sizeof(tl_hgen));
      gen = 1;
    }
  };
  BitWriter bw{dst, dst + cap};
  int bits = 9, next = kFirst;
  reset();
  bw.put(kClear, bits);
  if (n > 0) {
    int w = src[0];
    for (long long i = 1; i < n; ++i) {
      const int k = src[i];
      const uint32_t key = ((uint32_t)w << 8) | (uint32_t)k;
      uint32_t h = (key * 2654435761u) >> 18;  // 14 bits
      int found = -1;
      while (hgen[h] == gen) {
        if (hkey[h] == key) {
          found = hval[h];
          break;
        }
        h = (h + 1) & (kHash - 1);
      }
      if (found >= 0) {
        w = found;
        continue;
      }
      bw.put(w, bits);
      hgen[h] = gen;
      hkey[h] = key;
      hval[h] = (uint16_t)next;
      ++next;
      if (next == kTable - 2) {  // libtiff: table full at 4094 entries -> clear
        bw.put(kClear, bits);
        bits = 9;
        next = kFirst;
        reset();
      } else if (next > (1 << bits) - 1) {
        ++bits;
      }
      w = k;
    }
    bw.put(w, bits);
    // the decoder adds one more entry after this code: keep the width of the end marker in step with it
    ++next;
    if (next == kTable - 2) {
      bw.put(kClear, bits);
      bits = 9;
    } else if (next > (1 << bits) - 1 && bits < kMaxBits) {
      ++bits;
    }
  }
  tl_gen = gen;
  bw.put(kEoi, bits);
  bw.flush();
  if (bw.overflow) {
    ffa_set_error("lzw_encode: output buffer too small (%lld bytes for %lld input bytes)", cap, n);
    return FFA_ERR_WORKSPACE;
  }
  return (long long)(bw.p - dst);
}

// TIFF Predictor = 2 (horizontal differencing) over `rows` rows of `row_samples` native-endian samples of
// `sample_bytes` (1, 2 or 4) bytes each, `stride` samples between horizontally neighbouring values of one band
// (= SamplesPerPixel for pixel-interleaved blocks, 1 for band-separate ones).  undo != 0 accumulates (read side),
// undo == 0 differences (write side).  In place.
extern "C" int ffa_tiff_hpredict(void* buf, long long rows, long long row_samples, int sample_bytes, int stride,
                                 int undo) {
  if (!buf || rows < 0 || row_samples < 0 || stride < 1 ||
      !(sample_bytes == 1 || sample_bytes == 2 || sample_bytes == 4)) {
    ffa_set_error("hpredict: bad arguments");
    return FFA_ERR_ARG;
  }
#define FFA_HP(T_)                                                                 \
  {                                                                                \
    T_* p = static_cast<T_*>(buf);                                                 \
    for (long long r = 0; r < rows; ++r, p += row_samples) {                       \
      if (undo) {                                                                  \
        for (long long i = stride; i < row_samples; ++i) p[i] = (T_)(p[i] + p[i - stride]); \
      } else {                                                                     \
        for (long long i = row_samples - 1; i >= stride; --i) p[i] = (T_)(p[i] - p[i - stride]); \
      }                                                                            \
    }                                                                              \
  }
  if (sample_bytes == 1) FFA_HP(uint8_t)
  else if (sample_bytes == 2) FFA_HP(uint16_t)
  else FFA_HP(uint32_t)
#undef FFA_HP
  return FFA_OK;
}
