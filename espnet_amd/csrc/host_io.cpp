// Host-side audio reader of the decode CLI: RIFF/WAVE files -> rows of one
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
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/espnet_amd.h"

namespace {

constexpr uint16_t FMT_PCM = 0x0001, FMT_FLOAT = 0x0003, FMT_EXT = 0xFFFE;

inline uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int probe_one(const char* path, EmWavInfo* w) {
  memset(w, 0, sizeof(*w));
  FILE* f = fopen(path, "rb");
  if (!f) return EM_ERR_IO;
  unsigned char h[12];
  int rc = EM_ERR_UNSUPPORTED;
  bool have_fmt = false;
  if (fread(h, 1, 12, f) == 12 && !memcmp(h, "RIFF", 4) && !memcmp(h + 8, "WAVE", 4)) {
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
