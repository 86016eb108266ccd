// Host-side audio reader of the decode CLI: RIFF/WAVE and FLAC files -> rows of one
// zero-padded f32 batch matrix, decoded by a small thread pool straight into the (pinned) buffer the H2D
// copy reads.  Pure host code: no HIP calls, usable without a GPU.
//
// Reference behaviour restated:
//   soundfile.read(path, dtype=float32) as used by espnet2/fileio/sound_scp.py:13-155 and the `sound` entry of
//   espnet2/train/iterable_dataset.py:44-67 (libsndfile's float conversion: integer PCM of w bytes ->
//   value / 2^(8w-1), 8-bit is unsigned with offset 128, IEEE float taken as is), and
//   CommonCollateFn (espnet2/train/collate_fn.py:17-95): pad every utterance of the batch with 0.0 to the longest.
// The Python reader espnet_amd/fileio/sound_scp.py:read_wav is the same restatement; tests compare the two
// bit for bit.  Only what the batched fast path needs is handled here (mono PCM 8/16/24/32 and IEEE float
// 32/64); everything else reports EM_ERR_UNSUPPORTED per file and the Python reader takes that window.
//
// FLAC (the recipes' default `audio_format=flac`, egs2/TEMPLATE/asr1/asr.sh:56; libsndfile decodes it for the
// reference) is decoded here from the format specification (RFC 9639): STREAMINFO, frame headers with CRC-8,
// constant / verbatim / fixed / LPC subframes, Rice and Rice2 partitioned residuals incl. escaped partitions,
// wasted bits, left-side / right-side / mid-side decorrelation, frame CRC-16; samples -> value / 2^(bits-1).
// Pinned by the specification's own self-checking example (MD5 + CRCs) and by round trips through an
// independent encoder in tests/ over every subframe type; the Python decoder in fileio/sound_scp.py is a second
// implementation the tests compare against bit for bit.
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/espnet_amd.h"

namespace {

constexpr uint16_t FMT_PCM = 0x0001, FMT_FLOAT = 0x0003, FMT_EXT = 0xFFFE;
constexpr int32_t FMT_FLAC = EM_AUDIO_FORMAT_FLAC;  // EmWavInfo.format of a FLAC stream (not a WAVE tag)

// ---------------------------------------------------------------------------------------------- FLAC (RFC 9639)
struct BitReader {  // MSB-first, 64-bit window refilled a byte at a time
  const unsigned char* p;
  size_t n, byte = 0;
  uint64_t cache = 0;  // next bits left aligned
  int cbits = 0;       // valid bits in cache
  bool bad = false;
  BitReader(const unsigned char* p_, size_t n_) : p(p_), n(n_) {}
  inline void refill() {
    if (byte + 8 <= n) {  // one unaligned big-endian 64-bit load tops the window up to >= 57 bits
      uint64_t w;
      memcpy(&w, p + byte, 8);
      w = __builtin_bswap64(w);
      cache |= cbits ? (w >> cbits) : w;
      const int add = (64 - cbits) >> 3;  // whole bytes that fit
      byte += (size_t)add;
      cbits += add * 8;
      if (cbits < 64) cache &= ~0ull << (64 - cbits);  // invariant: bits beyond cbits are zero
      return;
    }
    while (cbits <= 56 && byte < n) {
      cache |= (uint64_t)p[byte++] << (56 - cbits);
      cbits += 8;
    }
  }
  inline size_t pos() const { return byte * 8 - (size_t)cbits; }  // bits consumed so far
  inline uint64_t bits32(int k) {  // 1 <= k <= 32
    if (cbits < k) {
      refill();
      if (cbits < k) {
        bad = true;
        return 0;
      }
    }
    const uint64_t v = cache >> (64 - k);
    cache <<= k;
    cbits -= k;
    return v;
  }
  inline uint64_t bits(int k) {  // 0 <= k <= 57
    if (k == 0) return 0;
    if (k <= 32) return bits32(k);
    const uint64_t hi = bits32(k - 32);
    return (hi << 32) | bits32(32);
  }
  inline uint32_t bit() { return (uint32_t)bits32(1); }
  inline int64_t sbits(int k) {
    if (k == 0) return 0;
    const uint64_t v = bits(k);
    const uint64_t sign = 1ull << (k - 1);
    return (int64_t)(v ^ sign) - (int64_t)sign;
  }
  inline uint32_t unary() {  // number of 0 bits before the next 1
    uint32_t q = 0;
    for (;;) {
      if (cbits == 0) {
        refill();
        if (cbits == 0) {
          bad = true;
          return q;
        }
      }
      if (cache == 0) {  // only zeros in the window
        q += (uint32_t)cbits;
        cbits = 0;
        if (q > (1u << 24)) {
          bad = true;
          return q;
        }
        continue;
      }
      const int lz = __builtin_clzll(cache);  // < cbits: bits beyond cbits are zero, the one found is valid
      q += (uint32_t)lz;
      cache <<= lz;  // lz <= 63
      cache <<= 1;
      cbits -= lz + 1;
      return q;
    }
  }
  inline void align() {
    const int drop = cbits & 7;
    cache <<= drop;
    cbits -= drop;
  }
};

inline uint8_t crc8(const unsigned char* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
  }
  return c;
}

struct Crc16Table {  // slicing-by-4 tables of the MSB-first CRC-16 (poly 0x8005, init 0)
  uint16_t t[4][256];
  Crc16Table() {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
      t[0][i] = c;
    }
    for (int i = 0; i < 256; ++i)
      for (int s = 1; s < 4; ++s) t[s][i] = (uint16_t)((t[s - 1][i] << 8) ^ t[0][t[s - 1][i] >> 8]);
  }
};
inline uint16_t crc16(const unsigned char* p, size_t n) {
  static const Crc16Table tab;
  uint16_t c = 0;
  size_t i = 0;
  for (; i + 4 <= n; i += 4)  // four message bytes per step: the first two meet the running remainder
    c = (uint16_t)(tab.t[3][(c >> 8) ^ p[i]] ^ tab.t[2][(c & 0xFF) ^ p[i + 1]] ^ tab.t[1][p[i + 2]] ^ tab.t[0][p[i + 3]]);
  for (; i < n; ++i) c = (uint16_t)((c << 8) ^ tab.t[0][((c >> 8) ^ p[i]) & 0xFF]);
  return c;
}

struct FlacStreamInfo {
  int min_block = 0, max_block = 0, rate = 0, channels = 0, bits = 0;
  int64_t total = 0;
  size_t first_frame = 0;  // byte offset of the first frame
};

// metadata blocks: "fLaC", then (last flag | type, 24-bit length, body)*; STREAMINFO (type 0) must come first
int flac_stream_info(const unsigned char* d, size_t n, FlacStreamInfo* si) {
  if (n < 4 + 4 + 34 || memcmp(d, "fLaC", 4)) return EM_ERR_UNSUPPORTED;
  size_t o = 4;
  bool first = true, last = false;
  while (!last) {
    if (o + 4 > n) return EM_ERR_UNSUPPORTED;
    last = (d[o] & 0x80) != 0;
    const int type = d[o] & 0x7F;
    const size_t len = ((size_t)d[o + 1] << 16) | ((size_t)d[o + 2] << 8) | d[o + 3];
    o += 4;
    if (o + len > n) return EM_ERR_UNSUPPORTED;
    if (first) {
      if (type != 0 || len < 34) return EM_ERR_UNSUPPORTED;
      const unsigned char* s = d + o;
      si->min_block = (s[0] << 8) | s[1];
      si->max_block = (s[2] << 8) | s[3];
      si->rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      si->channels = ((s[12] >> 1) & 7) + 1;
      si->bits = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      si->total = ((int64_t)(s[13] & 0x0F) << 32) | ((int64_t)s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17];
      first = false;
    }
    o += len;
  }
  si->first_frame = o;
  return (si->bits >= 4 && si->bits <= 32 && si->rate > 0) ? EM_OK : EM_ERR_UNSUPPORTED;
}

// out[k] += (sum_j coef[j] * out[k-1-j]) >> shift for k >= ORDER; the common orders are unrolled.  The sums
// wrap modulo 2^64 (unsigned arithmetic): valid streams never get near that, damaged ones must not be UB.
template <int ORDER>
inline void flac_predict_n(int64_t* out, int bs, const int64_t* coef, int shift) {
  uint64_t c[ORDER];
  for (int j = 0; j < ORDER; ++j) c[j] = (uint64_t)coef[j];
  for (int k = ORDER; k < bs; ++k) {
    uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < ORDER; ++j) acc += c[j] * (uint64_t)out[k - 1 - j];
    out[k] = (int64_t)((uint64_t)out[k] + (uint64_t)((int64_t)acc >> shift));
  }
}
inline void flac_predict(int64_t* out, int bs, int order, const int64_t* coef, int shift) {
  switch (order) {
    case 0: return;
    case 1: return flac_predict_n<1>(out, bs, coef, shift);
    case 2: return flac_predict_n<2>(out, bs, coef, shift);
    case 3: return flac_predict_n<3>(out, bs, coef, shift);
    case 4: return flac_predict_n<4>(out, bs, coef, shift);
    case 5: return flac_predict_n<5>(out, bs, coef, shift);
    case 6: return flac_predict_n<6>(out, bs, coef, shift);
    case 7: return flac_predict_n<7>(out, bs, coef, shift);
    case 8: return flac_predict_n<8>(out, bs, coef, shift);
    case 9: return flac_predict_n<9>(out, bs, coef, shift);
    case 10: return flac_predict_n<10>(out, bs, coef, shift);
    case 11: return flac_predict_n<11>(out, bs, coef, shift);
    case 12: return flac_predict_n<12>(out, bs, coef, shift);
  }
  for (int k = order; k < bs; ++k) {
    uint64_t acc = 0;
    for (int j = 0; j < order; ++j) acc += (uint64_t)coef[j] * (uint64_t)out[k - 1 - j];
    out[k] = (int64_t)((uint64_t)out[k] + (uint64_t)((int64_t)acc >> shift));
  }
}

// one subframe of `bs` samples at `bps` bits into out[] (int64: 33-bit side channels of 32-bit streams fit)
int flac_subframe(BitReader& br, int bs, int bps, int64_t* out) {
  if (br.bit()) return EM_ERR_UNSUPPORTED;  // padding bit
  const int type = (int)br.bits(6);
  int wasted = 0;
  if (br.bit()) wasted = (int)br.unary() + 1;
  if (wasted >= bps) return EM_ERR_UNSUPPORTED;
  bps -= wasted;
  int order = 0;
  if (type == 0) {  // constant
    const int64_t v = br.sbits(bps);
    for (int i = 0; i < bs; ++i) out[i] = v;
  } else if (type == 1) {  // verbatim
    for (int i = 0; i < bs; ++i) out[i] = br.sbits(bps);
  } else if ((type >= 8 && type <= 12) || type >= 32) {
    const bool lpc = type >= 32;
    order = lpc ? (type & 31) + 1 : type - 8;
    if (order > bs) return EM_ERR_UNSUPPORTED;
    for (int i = 0; i < order; ++i) out[i] = br.sbits(bps);
    int prec = 0, shift = 0;
    int64_t coef[32];
    if (lpc) {
      prec = (int)br.bits(4) + 1;
      if (prec == 16) return EM_ERR_UNSUPPORTED;
      shift = (int)br.sbits(5);
      if (shift < 0) return EM_ERR_UNSUPPORTED;
      for (int j = 0; j < order; ++j) coef[j] = br.sbits(prec);
    } else {
      static const int64_t fixed[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
      for (int j = 0; j < order; ++j) coef[j] = fixed[order][j];
    }
    // residual: coding method, partition order, then 2^po partitions of Rice codes (or raw, when escaped)
    const int method = (int)br.bits(2);
    if (method > 1) return EM_ERR_UNSUPPORTED;
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int po = (int)br.bits(4);
    const int parts = 1 << po;
    if ((bs >> po) << po != bs && po > 0) return EM_ERR_UNSUPPORTED;
    if ((bs >> po) < order && po > 0) return EM_ERR_UNSUPPORTED;
    int i = order;
    for (int pt = 0; pt < parts && !br.bad; ++pt) {
      const int cnt = (bs >> po) - (pt == 0 ? order : 0);
      if (cnt < 0 || i + cnt > bs) return EM_ERR_UNSUPPORTED;
      const int param = (int)br.bits(pbits);
      if (param == esc) {
        const int nb = (int)br.bits(5);
        for (int k = 0; k < cnt; ++k) out[i + k] = br.sbits(nb);
      } else {
        for (int k = 0; k < cnt; ++k) {
          uint64_t u;
          if (br.cbits < 48) br.refill();
          const int lz = br.cache ? __builtin_clzll(br.cache) : 64;
          if (lz + 1 + param <= br.cbits && lz + 1 + param <= 63) {  // whole code inside the window: no calls
            const uint64_t rest = br.cache << (lz + 1);
            u = ((uint64_t)lz << param) | (param ? rest >> (64 - param) : 0);
            br.cache = rest << param;
            br.cbits -= lz + 1 + param;
          } else {
            const uint64_t q = br.unary();
            u = (q << param) | br.bits(param);
          }
          out[i + k] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
        }
      }
      i += cnt;
    }
    if (br.bad || i != bs) return EM_ERR_UNSUPPORTED;
    flac_predict(out, bs, order, coef, lpc ? shift : 0);  // prediction + residual (fixed predictors: shift 0)
  } else {
    return EM_ERR_UNSUPPORTED;  // reserved subframe type
  }
  if (br.bad) return EM_ERR_UNSUPPORTED;
  if (wasted)
    for (int i = 0; i < bs; ++i) out[i] = (int64_t)((uint64_t)out[i] << wasted);
  return EM_OK;
}

// decode a whole mono stream into row[0 .. frames) as value / 2^(bits-1); returns the samples written or < 0
int64_t flac_decode_mono(const unsigned char* d, size_t n, const FlacStreamInfo& si, float* row, int64_t cap) {
  std::vector<int64_t> buf;
  size_t o = si.first_frame;
  int64_t written = 0;
  const float scale = 1.0f / (float)(1ull << (si.bits - 1));
  const double dscale = 1.0 / (double)(1ull << (si.bits - 1));
  while (o + 2 <= n && written < cap) {
    if (d[o] != 0xFF || (d[o + 1] & 0xFE) != 0xF8) return EM_ERR_UNSUPPORTED;  // sync code + reserved bit
    BitReader br(d + o, n - o);
    br.bits(16);
    const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), ch_code = (int)br.bits(4);
    const int sz_code = (int)br.bits(3);
    if (br.bit()) return EM_ERR_UNSUPPORTED;
    // UTF-8 style coded frame / sample number (up to 7 bytes)
    int lead = (int)br.bits(8), extra = 0;
    if (lead & 0x80) {
      while (lead & (0x40 >> extra)) ++extra;
      if (extra < 1 || extra > 6) return EM_ERR_UNSUPPORTED;
      for (int k = 0; k < extra; ++k)
        if ((br.bits(8) & 0xC0) != 0x80) return EM_ERR_UNSUPPORTED;
    }
    int bs;
    if (bs_code == 0) return EM_ERR_UNSUPPORTED;
    else if (bs_code == 1) bs = 192;
    else if (bs_code <= 5) bs = 576 << (bs_code - 2);
    else if (bs_code == 6) bs = (int)br.bits(8) + 1;
    else if (bs_code == 7) bs = (int)br.bits(16) + 1;
    else bs = 256 << (bs_code - 8);
    if (sr_code == 12) br.bits(8);
    else if (sr_code == 13 || sr_code == 14) br.bits(16);
    else if (sr_code == 15) return EM_ERR_UNSUPPORTED;
    static const int sizes[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    const int bps = sz_code == 0 ? si.bits : sizes[sz_code];
    if (bps != si.bits || ch_code != 0) return EM_ERR_UNSUPPORTED;  // mono streams only on this path
    if (br.bad || (br.pos() & 7)) return EM_ERR_UNSUPPORTED;
    const size_t hdr = br.pos() >> 3;
    if (hdr + 1 > n - o || crc8(d + o, hdr) != d[o + hdr]) return EM_ERR_IO;
    br.bits(8);
    buf.resize((size_t)bs);
    const int rc = flac_subframe(br, bs, bps, buf.data());
    if (rc != EM_OK) return rc;
    br.align();
    const size_t body = br.pos() >> 3;
    if (body + 2 > n - o) return EM_ERR_IO;
    if (crc16(d + o, body) != (uint16_t)((d[o + body] << 8) | d[o + body + 1])) return EM_ERR_IO;
    o += body + 2;
    int64_t take = bs;
    if (written + take > cap) take = cap - written;
    if (si.bits <= 24) {  // exact in float: integer of <= 24 bits times a power of two
      for (int64_t i = 0; i < take; ++i) row[written + i] = (float)buf[(size_t)i] * scale;
    } else {
      for (int64_t i = 0; i < take; ++i) row[written + i] = (float)((double)buf[(size_t)i] * dscale);
    }
    written += take;
  }
  return written;
}

bool read_file(const char* path, std::vector<unsigned char>& data) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long end = ftell(f);
  fseek(f, 0, SEEK_SET);
  bool ok = end >= 0;
  if (ok) {
    data.resize((size_t)end);
    ok = end == 0 || fread(data.data(), 1, (size_t)end, f) == (size_t)end;
  }
  fclose(f);
  return ok;
}

int probe_flac(FILE* f, EmWavInfo* w) {
  // STREAMINFO is always the first block (4 + 4 + 34 bytes); the first frame offset only needs the headers of
  // the blocks behind it (their bodies -- pictures, seek tables -- are skipped, not read)
  unsigned char head[42];
  fseek(f, 0, SEEK_SET);
  const size_t got = fread(head, 1, sizeof(head), f);
  FlacStreamInfo si;
  if (got < 42 || memcmp(head, "fLaC", 4)) return EM_ERR_UNSUPPORTED;
  fseek(f, 0, SEEK_END);
  const long fsize = ftell(f);
  size_t o = 4;
  bool last = false, first = true;
  while (!last) {
    unsigned char h[4];
    if (fseek(f, (long)o, SEEK_SET) || fread(h, 1, 4, f) != 4) return EM_ERR_UNSUPPORTED;
    last = (h[0] & 0x80) != 0;
    const size_t len = ((size_t)h[1] << 16) | ((size_t)h[2] << 8) | h[3];
    if (first && ((h[0] & 0x7F) != 0 || len < 34)) return EM_ERR_UNSUPPORTED;
    first = false;
    o += 4 + len;
    if ((long)o > fsize) return EM_ERR_UNSUPPORTED;
  }
  head[4] |= 0x80;  // parse STREAMINFO alone
  if (flac_stream_info(head, sizeof(head), &si) != EM_OK) return EM_ERR_UNSUPPORTED;
  w->format = FMT_FLAC;
  w->channels = si.channels;
  w->rate = si.rate;
  w->bits = si.bits;
  w->frames = si.total;
  w->data_offset = (int64_t)o;
  if (si.total <= 0) return EM_ERR_UNSUPPORTED;  // unknown length: the Python reader decodes to find out
  // a damaged STREAMINFO must not make the caller allocate an absurd row: a frame takes >= 10 bytes and holds
  // at most max_block samples, and 2^28 samples (4.6 h at 16 kHz) is more than one utterance ever is
  const int64_t payload = (int64_t)fsize - (int64_t)o;
  const int64_t max_block = si.max_block > 0 ? si.max_block : 65536;
  if (si.total > (1ll << 28) || si.total > (payload / 10 + 1) * max_block) return EM_ERR_UNSUPPORTED;
  return si.channels == 1 ? EM_OK : EM_ERR_UNSUPPORTED;
}

inline uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int probe_one(const char* path, EmWavInfo* w) {
  memset(w, 0, sizeof(*w));
  FILE* f = fopen(path, "rb");
  if (!f) return EM_ERR_IO;
  unsigned char h[12];
  int rc = EM_ERR_UNSUPPORTED;
  bool have_fmt = false;
  const size_t got = fread(h, 1, 12, f);
  if (got >= 4 && !memcmp(h, "fLaC", 4)) {
    rc = probe_flac(f, w);
  } else if (got == 12 && !memcmp(h, "RIFF", 4) && !memcmp(h + 8, "WAVE", 4)) {
    for (;;) {
      unsigned char ck[8];
      if (fread(ck, 1, 8, f) != 8) break;
      const uint32_t size = le32(ck + 4);
      if (!memcmp(ck, "fmt ", 4)) {
        unsigned char raw[40];
        const size_t take = size < sizeof(raw) ? size : sizeof(raw);
        if (size < 16 || fread(raw, 1, take, f) != take) break;
        uint16_t tag = le16(raw);
        if (tag == FMT_EXT && size >= 26) tag = le16(raw + 24);
        w->format = tag;
        w->channels = le16(raw + 2);
        w->rate = (int32_t)le32(raw + 4);
        w->bits = le16(raw + 14);
        have_fmt = true;
        if (size > take && fseek(f, (long)(size - take), SEEK_CUR)) break;
      } else if (!memcmp(ck, "data", 4)) {
        if (!have_fmt) break;
        const long off = ftell(f);
        fseek(f, 0, SEEK_END);
        const long end = ftell(f);
        int64_t bytes = size;
        if (off + bytes > end) bytes = end - off;  // truncated file: what is there (like f.read(size))
        const int bps = w->bits / 8;
        const bool known = (w->format == FMT_PCM && (w->bits == 8 || w->bits == 16 || w->bits == 24 || w->bits == 32)) ||
                           (w->format == FMT_FLOAT && (w->bits == 32 || w->bits == 64));
        if (!known || w->channels < 1) break;
        w->data_offset = off;
        w->frames = bytes / ((int64_t)bps * w->channels);
        rc = w->channels == 1 ? EM_OK : EM_ERR_UNSUPPORTED;
        break;
      } else if (fseek(f, (long)size, SEEK_CUR)) {
        break;
      }
      if ((size & 1) && fseek(f, 1, SEEK_CUR)) break;
    }
  }
  fclose(f);
  return rc;
}

int load_one(const char* path, const EmWavInfo& w, float* row, int64_t ld, std::vector<unsigned char>& tmp) {
  if (w.status != EM_OK || w.frames > ld) return EM_ERR_BAD_ARG;
  const int64_t n = w.frames;
  const int bps = w.bits / 8;
  if (w.format == FMT_FLAC) {
    if (!read_file(path, tmp)) return EM_ERR_IO;
    FlacStreamInfo si;
    si.rate = w.rate;
    si.channels = w.channels;
    si.bits = w.bits;
    si.total = w.frames;
    si.first_frame = (size_t)w.data_offset;
    if (si.first_frame > tmp.size()) return EM_ERR_IO;
    const int64_t got = flac_decode_mono(tmp.data(), tmp.size(), si, row, n);
    if (got < 0) return (int)got;
    if (got != n) return EM_ERR_IO;  // fewer samples than STREAMINFO announces
    if (ld > n) memset(row + n, 0, (size_t)(ld - n) * sizeof(float));
    return EM_OK;
  }
  FILE* f = fopen(path, "rb");
  if (!f) return EM_ERR_IO;
  int rc = EM_OK;
  if (fseek(f, (long)w.data_offset, SEEK_SET)) rc = EM_ERR_IO;
  if (rc == EM_OK) {
    if (w.format == FMT_FLOAT && w.bits == 32) {  // already the output type: read in place
      if ((int64_t)fread(row, 4, (size_t)n, f) != n) rc = EM_ERR_IO;
    } else {
      tmp.resize((size_t)(n * bps));
      if ((int64_t)fread(tmp.data(), (size_t)bps, (size_t)n, f) != n) rc = EM_ERR_IO;
      const unsigned char* p = tmp.data();
      if (rc != EM_OK) {
      } else if (w.format == FMT_FLOAT) {  // f64 -> f32, round to nearest
        for (int64_t i = 0; i < n; ++i) {
          double v;
          memcpy(&v, p + 8 * i, 8);
          row[i] = (float)v;
        }
      } else if (w.bits == 16) {  // exact: int16 times a power of two
        const int16_t* s = (const int16_t*)p;
        for (int64_t i = 0; i < n; ++i) row[i] = (float)s[i] * (1.0f / 32768.0f);
      } else if (w.bits == 8) {
        for (int64_t i = 0; i < n; ++i) row[i] = ((float)p[i] - 128.0f) / 128.0f;
      } else if (w.bits == 24) {
        for (int64_t i = 0; i < n; ++i) {
          int32_t v = p[3 * i] | (p[3 * i + 1] << 8) | (p[3 * i + 2] << 16);
          if (v & 0x800000) v -= 0x1000000;
          row[i] = (float)v * (1.0f / 8388608.0f);
        }
      } else {  // 32-bit PCM: through double like the Python reader (one rounding, to f32)
        for (int64_t i = 0; i < n; ++i) {
          int32_t v;
          memcpy(&v, p + 4 * i, 4);
          row[i] = (float)((double)v / 2147483648.0);
        }
      }
    }
  }
  fclose(f);
  if (rc == EM_OK && ld > n) memset(row + n, 0, (size_t)(ld - n) * sizeof(float));  // collate_fn.py pad value 0.0
  return rc;
}

template <typename Fn>
void run_pool(int32_t n, int32_t threads, Fn fn) {
  int nt = threads < 1 ? 1 : threads;
  if (nt > n) nt = n;
  if (nt > 64) nt = 64;
  std::atomic<int32_t> next(0);
  auto work = [&]() {
    std::vector<unsigned char> tmp;
    for (;;) {
      const int32_t i = next.fetch_add(1);
      if (i >= n) return;
      fn(i, tmp);
    }
  };
  if (nt <= 1) {
    work();
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve((size_t)nt - 1);
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
}

}  // namespace

extern "C" int em_wav_probe(const char* const* paths, int32_t n, EmWavInfo* info, int32_t threads) {
  if (!paths || !info || n < 0) return EM_ERR_BAD_ARG;
  if (n == 0) return EM_OK;
  std::atomic<int> worst(EM_OK);
  run_pool(n, threads, [&](int32_t i, std::vector<unsigned char>&) {
    const int rc = paths[i] ? probe_one(paths[i], &info[i]) : EM_ERR_BAD_ARG;
    info[i].status = rc;
    if (rc != EM_OK) worst.store(rc);
  });
  return worst.load();
}

extern "C" int em_wav_load_rows(const char* const* paths, const EmWavInfo* info, int32_t n, float* out, int64_t ld,
                                int32_t threads) {
  if (!paths || !info || n < 0 || ld < 0 || (!out && ld > 0)) return EM_ERR_BAD_ARG;
  if (n == 0 || ld == 0) return EM_OK;  // a batch of empty files has nothing to decode
  std::atomic<int> worst(EM_OK);
  run_pool(n, threads, [&](int32_t i, std::vector<unsigned char>& tmp) {
    const int rc = paths[i] ? load_one(paths[i], info[i], out + (size_t)i * (size_t)ld, ld, tmp) : EM_ERR_BAD_ARG;
    if (rc != EM_OK) worst.store(rc);
  });
  return worst.load();
}
