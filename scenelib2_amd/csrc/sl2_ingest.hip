// Frame ingest: the FileGrabber / FrameGrabber pair of the reference
// (framegrabber/filegrabber.cpp:53-109, framegrabber/framegrabber.cpp:73-104) for a batch of
// sequences, feeding sl2_go_one_step(frames_on_device = 1).
//
//   * listing: every regular file below the sequence's directory, recursively, sorted by full path
//     (byte-wise std::sort on the path strings, filegrabber.cpp:63-83);
//   * decode: 8-bit single-channel, row-major, step == width — the only contract GoOneStep relies on
//     (SURVEY 2, row 14).  The reference decodes through cv::imread(path, 0) (filegrabber.cpp:106-109), a third-party
//     call that is not re-implemented in general; three containers are read here, chosen by the file's magic bytes:
//       - binary PGM (P5, maxval <= 255), the format of the reference's own fixtures;
//       - PNG (the format of the MonoSLAM test sequences): plain or Adam7-interlaced, 8 bits per sample (grey, grey + alpha, RGB,
//         RGBA, palette) or 1/2/4-bit grey / palette; IDAT inflated with zlib, the five scan-line filters undone here.
//         Grey PNGs are delivered byte for byte.  Colour goes to grey the way cv::imread(.., 0) gets it from libpng
//         (png_set_rgb_to_gray, 8-bit path): (9797 R + 19234 G + 3737 B + 16384) >> 15; alpha is dropped;
//       - JPEG (round 4; sl2_jpeg.hpp): sequential or progressive DCT, Huffman, 8 bits, grey or YCbCr - the luminance component through
//         libjpeg's integer inverse DCT, which is what imread(.., 0) delivers (out_color_space = JCS_GRAYSCALE);
//   * a producer thread decodes ahead into pinned host buffers (the reference queues <= 50 frames,
//     framegrabber.cpp:93-104; here `depth` batches), the consumer uploads one batch per call with an
//     asynchronous copy into one of two device buffers, so the copy of frame k+1 overlaps the step on
//     frame k when both are issued on different streams.
#include <dirent.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sl2_common.hpp"
#include "sl2_jpeg.hpp"

namespace sl2 {

constexpr size_t kMaxFramePixels = (size_t)64 << 20;   // a header that asks for more is treated as corrupt, not allocated

static bool list_files_recursive(const std::string& dir, std::vector<std::string>& out) {
  DIR* d = opendir(dir.c_str());
  if (!d) return false;
  while (dirent* ent = readdir(d)) {
    const std::string name = ent->d_name;
    if (name == "." || name == "..") continue;
    const std::string path = dir + "/" + name;
    struct stat st;
    if (stat(path.c_str(), &st) != 0) continue;
    if (S_ISDIR(st.st_mode)) {
      if (!list_files_recursive(path, out)) { closedir(d); return false; }
    } else {
      out.push_back(path);
    }
  }
  closedir(d);
  std::sort(out.begin(), out.end());
  return true;
}

// binary PGM -> grey bytes.  Returns false (and sets the error string) on anything else.
static bool read_pgm(const std::string& path, std::vector<uint8_t>& px, int* w, int* h) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { set_error(("cannot open " + path).c_str()); return false; }
  auto token = [&](std::string& t) -> bool {
    t.clear();
    int c;
    for (;;) {
      c = fgetc(f);
      if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
      if (c == EOF) return false;
      if (c != ' ' && c != '\t' && c != '\n' && c != '\r') break;
    }
    while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r') { t.push_back((char)c); c = fgetc(f); }
    return !t.empty();
  };
  std::string magic, sw, sh, smax;
  bool ok = token(magic) && magic == "P5" && token(sw) && token(sh) && token(smax);
  int W = 0, H = 0, maxval = 0;
  if (ok) { W = atoi(sw.c_str()); H = atoi(sh.c_str()); maxval = atoi(smax.c_str()); }
  ok = ok && W > 0 && H > 0 && maxval > 0 && maxval <= 255 && (size_t)W * (size_t)H <= kMaxFramePixels;
  if (ok) {
    px.resize((size_t)W * H);
    ok = fread(px.data(), 1, px.size(), f) == px.size();
  }
  fclose(f);
  if (!ok) { set_error(("not a binary 8-bit PGM: " + path).c_str()); return false; }
  *w = W; *h = H;
  return true;
}

// PNG -> grey bytes (see the header comment for what is covered).
static bool read_png(const std::string& path, std::vector<uint8_t>& px, int* w, int* h) {
  auto fail = [&](const char* why) { set_error((std::string(why) + ": " + path).c_str()); return false; };
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fail("cannot open");
  std::vector<uint8_t> file;
  {
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
  }
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return fail("not a PNG");
  auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | ((uint32_t)file[o + 1] << 16) | ((uint32_t)file[o + 2] << 8) | file[o + 3]; };
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool seen_end = false;
  for (size_t o = 8; o + 12 <= file.size() && !seen_end;) {
    const uint32_t len = be32(o);
    if (o + 12 + (size_t)len > file.size()) return fail("truncated PNG");
    const char* type = (const char*)&file[o + 4];
    const uint8_t* data = &file[o + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (len < 13) return fail("bad IHDR");
      W = be32(o + 8); H = be32(o + 12);
      depth = data[8]; ctype = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0) return fail("unknown PNG compression / filter method");
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      seen_end = true;
    }
    o += 12 + (size_t)len;
  }
  if (W == 0 || H == 0 || W > 65535 || H > 65535 || idat.empty()) return fail("PNG without image data");
  if ((size_t)W * H > kMaxFramePixels) return fail("PNG larger than 64 M pixels");
  if (interlace != 0 && interlace != 1) return fail("unknown PNG interlace method");
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return fail("unknown PNG colour type");
  }
  const bool sub_byte = depth == 1 || depth == 2 || depth == 4;
  if (!(depth == 8 || (sub_byte && (ctype == 0 || ctype == 3)))) return fail("PNG bit depth not supported (8, or 1/2/4 grey / palette)");
  if (ctype == 3 && plte.size() < 3) return fail("palette PNG without PLTE");
  const size_t bpp = (size_t)std::max(1, channels * depth / 8);            // filter distance in bytes
  // the image, or the seven reduced images of an Adam7-interlaced file (PNG specification, section 8.2), one after the other
  // in the inflated stream: first pixel (x0, y0), every dx-th column and dy-th row
  struct Pass { uint32_t x0, y0, dx, dy; };
  static const Pass kAdam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const Pass kWhole[1] = {{0, 0, 1, 1}};
  const Pass* passes = interlace ? kAdam7 : kWhole;
  const int npass = interlace ? 7 : 1;
  size_t total = 0;
  for (int k = 0; k < npass; ++k) {
    const size_t pw = W > passes[k].x0 ? (W - passes[k].x0 + passes[k].dx - 1) / passes[k].dx : 0;
    const size_t ph = H > passes[k].y0 ? (H - passes[k].y0 + passes[k].dy - 1) / passes[k].dy : 0;
    if (pw && ph) total += ((pw * channels * depth + 7) / 8 + 1) * ph;
  }
  std::vector<uint8_t> raw(total);
  {
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return fail("PNG inflate failed");
  }
  auto to_grey = [](int r, int g, int b) { return (uint8_t)((9797 * r + 19234 * g + 3737 * b + 16384) >> 15); };
  px.resize((size_t)W * H);
  size_t at = 0;
  for (int k = 0; k < npass; ++k) {
    const Pass ps = passes[k];
    const size_t pw = W > ps.x0 ? (W - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = H > ps.y0 ? (H - ps.y0 + ps.dy - 1) / ps.dy : 0;
    if (!pw || !ph) continue;
    const size_t stride = (pw * channels * depth + 7) / 8;
    uint8_t* base = raw.data() + at;
    at += (stride + 1) * ph;
    // undo the scan-line filters in place (PNG specification, section 9: None, Sub, Up, Average, Paeth)
    std::vector<uint8_t> zero(stride, 0);
    for (size_t y = 0; y < ph; ++y) {
      uint8_t* cur = &base[y * (stride + 1) + 1];
      const uint8_t* up = y ? &base[(y - 1) * (stride + 1) + 1] : zero.data();
      const int ft = base[y * (stride + 1)];
      for (size_t x = 0; x < stride; ++x) {
        const int a = x >= bpp ? cur[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
        int pred;
        switch (ft) {
          case 0: pred = 0; break;
          case 1: pred = a; break;
          case 2: pred = b; break;
          case 3: pred = (a + b) >> 1; break;
          case 4: {
            const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
            pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            break;
          }
          default: return fail("bad PNG filter type");
        }
        cur[x] = (uint8_t)(cur[x] + pred);
      }
    }
    for (size_t y = 0; y < ph; ++y) {
      const uint8_t* row = &base[y * (stride + 1) + 1];
      uint8_t* out = &px[(size_t)(ps.y0 + y * ps.dy) * W + ps.x0];
      for (size_t x = 0; x < pw; ++x) {
        int v;     // the sample (or palette index) of pixel x
        if (sub_byte) {
          const size_t bit = x * depth;
          v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        } else {
          v = row[x * channels];
        }
        uint8_t g;
        switch (ctype) {
          case 0: g = sub_byte ? (uint8_t)(v * 255 / ((1 << depth) - 1)) : (uint8_t)v; break;
          case 4: g = (uint8_t)v; break;
          case 3: {
            if ((size_t)v * 3 + 2 >= plte.size()) return fail("PNG palette index out of range");
            g = to_grey(plte[3 * v], plte[3 * v + 1], plte[3 * v + 2]);
            break;
          }
          default: g = to_grey(row[x * channels], row[x * channels + 1], row[x * channels + 2]); break;
        }
        out[x * ps.dx] = g;
      }
    }
  }
  *w = (int)W; *h = (int)H;
  return true;
}

// JPEG (sequential or progressive DCT, Huffman): the luminance component through libjpeg's integer inverse DCT - what cv::imread(path, 0) delivers
static bool read_jpeg(const std::string& path, std::vector<uint8_t>& px, int* w, int* h) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { set_error(("cannot open " + path).c_str()); return false; }
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof(buf), f)) > 0) {
    file.insert(file.end(), buf, buf + got);
    if (file.size() > ((size_t)256 << 20)) break;
  }
  fclose(f);
  std::string err;
  if (!jpeg::decode_grey(file, px, w, h, &err, kMaxFramePixels)) { set_error((path + ": " + err).c_str()); return false; }
  return true;
}

// container chosen by the magic bytes
static bool read_image(const std::string& path, std::vector<uint8_t>& px, int* w, int* h) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { set_error(("cannot open " + path).c_str()); return false; }
  uint8_t m[2] = {0, 0};
  const size_t n = fread(m, 1, 2, f);
  fclose(f);
  if (n == 2 && m[0] == 0x89 && m[1] == 'P') return read_png(path, px, w, h);
  if (n == 2 && m[0] == 0xFF && m[1] == 0xD8) return read_jpeg(path, px, w, h);
  return read_pgm(path, px, w, h);
}

}  // namespace sl2

struct sl2_ingest {
  int device = 0, nseq = 0, width = 0, height = 0, depth = 0;
  std::vector<std::vector<std::string>> files;   // per sequence, sorted
  int n_frames = 0;                               // min over sequences
  // ring of `depth` pinned host batches [nseq][W*H]
  std::vector<uint8_t*> host;
  std::vector<int> state;        // 0 free, 1 decoded, 2 in flight (uploaded, waiting for its event)
  std::vector<hipEvent_t> done;
  int produced = 0, consumed = 0;
  bool stop = false, failed = false;
  std::string fail_msg;
  std::mutex mu;
  std::condition_variable cv;
  std::thread producer;
  uint8_t* dev[2] = {nullptr, nullptr};
  // Upload: frame k goes to dev[k & 1] on a copy stream of the grabber's own, ONE FRAME AHEAD of the caller - the copy of frame
  // k + 1 is issued by the call that hands out frame k and runs under the caller's work on frame k (round 6; before, the copy of
  // frame k was queued on the caller's stream in front of its step: 1.4 ms of a 1.5 ms step at batch 1024, 8 us of a single
  // sequence's frame).  done[slot]: the copy out of pinned slot `slot` has landed (the caller's stream waits for it, and the
  // producer may decode into the slot again); reusable[b]: the caller's stream has got past everything that read dev[b] (the copy
  // stream waits for it before it overwrites the buffer).
  hipStream_t copy_stream = nullptr;
  hipEvent_t reusable[2] = {nullptr, nullptr};
  bool handed_out[2] = {false, false};
  int issued = 0;                // frames whose copy has been issued (consumed <= issued <= consumed + 1 between calls)
  // Small batches (a single 320 x 240 sequence: 77 KB) are not copied at all: the device reads the pinned batch in place - the
  // ~6 us a hipMemcpyAsync costs the host per call, its wait and its events were most of what this grabber added to a 70 us
  // frame, and the step touches each byte of such a frame about once.  zero_copy: batches of at most zc_max bytes
  // (sl2_ingest_set_zero_copy; default 512 KB).  A slot handed out stays with the device (state 2) until an event recorded on the
  // caller's stream at the NEXT call - behind everything that read it - has completed (used_valid).
  size_t zc_max = 512 * 1024;
  std::vector<uint8_t*> host_dev;   // device-side addresses of the pinned batches
  std::vector<char> used_valid;     // done[slot] has been recorded behind the slot's consumer

  void run() {
    const size_t fb = (size_t)width * height;
    std::vector<uint8_t> px;
    for (int k = 0; k < n_frames; ++k) {
      const int slot = k % depth;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || state[slot] == 0; });
        if (stop) return;
      }
      for (int s = 0; s < nseq; ++s) {
        int w = 0, h = 0;
        bool ok = false;
        try { ok = sl2::read_image(files[s][k], px, &w, &h); } catch (const std::exception&) { ok = false; }
        if (!ok || w != width || h != height) {
          std::lock_guard<std::mutex> lk(mu);
          failed = true;
          fail_msg = "frame " + files[s][k] + " is not a " + std::to_string(width) + "x" + std::to_string(height) + " binary PGM / PNG";
          cv.notify_all();
          return;
        }
        memcpy(host[slot] + (size_t)s * fb, px.data(), fb);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        state[slot] = 1;
        ++produced;
      }
      cv.notify_all();
    }
  }
};

namespace sl2 {

// 8-bit greyscale image writer for sl2_save_patch (cv::imwrite's role at monoslam.cpp:1570): binary PGM for ".pgm",
// otherwise PNG (colour type 0, one zlib stream, filter 0 on every scan line - readable by any PNG decoder, and by
// sl2_read_image above).
int write_grey_image(const char* path, const uint8_t* px, int w, int h) {
  const size_t len = strlen(path);
  FILE* f = fopen(path, "wb");
  if (!f) { set_error(std::string("cannot write ") + path); return SL2_ERR_INVALID; }
  bool ok = true;
  if (len >= 4 && strcmp(path + len - 4, ".pgm") == 0) {
    ok = fprintf(f, "P5\n%d %d\n255\n", w, h) > 0 && fwrite(px, 1, (size_t)w * h, f) == (size_t)w * h;
  } else {
    auto be32 = [](uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t n) {
      uint8_t hdr[8];
      be32(hdr, n);
      memcpy(hdr + 4, type, 4);
      uint32_t crc = crc32(0L, hdr + 4, 4);
      if (n) crc = crc32(crc, data, n);
      uint8_t tail[4];
      be32(tail, crc);
      ok = ok && fwrite(hdr, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    ok = fwrite(sig, 1, 8, f) == 8;
    uint8_t ihdr[13];
    be32(ihdr, (uint32_t)w); be32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8; ihdr[9] = 0; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    std::vector<uint8_t> raw((size_t)h * (w + 1));
    for (int r = 0; r < h; ++r) { raw[(size_t)r * (w + 1)] = 0; memcpy(&raw[(size_t)r * (w + 1) + 1], px + (size_t)r * w, w); }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    ok = ok && compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), Z_BEST_SPEED) == Z_OK;
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) { set_error(std::string("write failed: ") + path); return SL2_ERR_INVALID; }
  return SL2_OK;
}

}  // namespace sl2

extern "C" {

int sl2_list_frames(const char* dir, char* buf, size_t capacity, int* count) {
  using namespace sl2;
  if (!dir || !count) return SL2_ERR_INVALID;
  std::vector<std::string> files;
  if (!list_files_recursive(dir, files)) { set_error("sl2_list_frames: provided directory doesn't exist"); return SL2_ERR_INVALID; }
  *count = (int)files.size();
  size_t need = 1;
  for (const auto& f : files) need += f.size() + 1;
  if (!buf) return SL2_OK;
  if (need > capacity) return SL2_ERR_CAPACITY;
  char* p = buf;
  for (const auto& f : files) { memcpy(p, f.c_str(), f.size()); p += f.size(); *p++ = '\n'; }
  *p = 0;
  return SL2_OK;
}

int sl2_read_pgm(const char* path, uint8_t* out, size_t capacity, int* width, int* height) {
  try {
    using namespace sl2;
    if (!path || !width || !height) return SL2_ERR_INVALID;
    std::vector<uint8_t> px;
    if (!read_pgm(path, px, width, height)) return SL2_ERR_INVALID;
    if (!out) return SL2_OK;
    if (px.size() > capacity) return SL2_ERR_CAPACITY;
    memcpy(out, px.data(), px.size());
    return SL2_OK;
  } catch (const std::exception& ex) {   // (allocation failure on a hostile header: never across the ABI)
    sl2::set_error(ex.what());
    return SL2_ERR_INVALID;
  }
}

int sl2_read_image(const char* path, uint8_t* out, size_t capacity, int* width, int* height) {
  try {
    using namespace sl2;
    if (!path || !width || !height) return SL2_ERR_INVALID;
    std::vector<uint8_t> px;
    if (!read_image(path, px, width, height)) return SL2_ERR_INVALID;
    if (!out) return SL2_OK;
    if (px.size() > capacity) return SL2_ERR_CAPACITY;
    memcpy(out, px.data(), px.size());
    return SL2_OK;
  } catch (const std::exception& ex) {   // (allocation failure on a hostile header: never across the ABI)
    sl2::set_error(ex.what());
    return SL2_ERR_INVALID;
  }
}

int sl2_ingest_open(const char* const* dirs, int nseq, int width, int height, int device, int depth, sl2_ingest** out) {
  using namespace sl2;
  if (!dirs || !out || nseq <= 0 || width <= 0 || height <= 0) return SL2_ERR_INVALID;
  if (depth < 2) depth = 2;
  if (depth > 50) depth = 50;      // FrameGrabber::IsFrameBufferFull (framegrabber.cpp:93-104)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device (the engine has no CPU fallback)"); return SL2_ERR_NO_DEVICE; }
  SL2_HIP(hipSetDevice(device));
  sl2_ingest* g = new sl2_ingest();
  g->device = device; g->nseq = nseq; g->width = width; g->height = height; g->depth = depth;
  g->files.resize(nseq);
  int nmin = -1;
  for (int s = 0; s < nseq; ++s) {
    if (!dirs[s] || !list_files_recursive(dirs[s], g->files[s])) {
      delete g;
      set_error("sl2_ingest_open: provided directory doesn't exist");
      return SL2_ERR_INVALID;
    }
    const int n = (int)g->files[s].size();
    nmin = (nmin < 0 || n < nmin) ? n : nmin;
  }
  g->n_frames = nmin;
  const size_t batch = (size_t)nseq * width * height;
  g->host.assign(depth, nullptr);
  g->state.assign(depth, 0);
  g->done.assign(depth, nullptr);
  g->host_dev.assign(depth, nullptr);
  g->used_valid.assign(depth, 0);
  for (int i = 0; i < depth; ++i) {
    if (hipHostMalloc((void**)&g->host[i], batch, hipHostMallocMapped) != hipSuccess || hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming) != hipSuccess ||
        hipHostGetDevicePointer((void**)&g->host_dev[i], g->host[i], 0) != hipSuccess) {
      set_error("sl2_ingest_open: pinned allocation failed");
      sl2_ingest_close(g);
      return SL2_ERR_HIP;
    }
  }
  for (int i = 0; i < 2; ++i)
    if (hipMalloc((void**)&g->dev[i], batch) != hipSuccess || hipEventCreateWithFlags(&g->reusable[i], hipEventDisableTiming) != hipSuccess) {
      set_error("sl2_ingest_open: device allocation failed");
      sl2_ingest_close(g);
      return SL2_ERR_HIP;
    }
  if (hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("sl2_ingest_open: cannot create the copy stream");
    sl2_ingest_close(g);
    return SL2_ERR_HIP;
  }
  g->producer = std::thread([g] { g->run(); });
  *out = g;
  return SL2_OK;
}

int sl2_ingest_frame_count(const sl2_ingest* g) { return g ? g->n_frames : 0; }

// The copy of frame k (host slot k % depth -> dev[k & 1]) on the copy stream; `wait`: block until the producer has decoded it
// (false: only if it is ready now).  0 = issued, 1 = not ready (wait == false only), < 0 = -(error code).
static int issue_copy(sl2_ingest* g, int k, bool wait) {
  using namespace sl2;
  const int slot = k % g->depth, b = k & 1;
  {
    // the slot may still be marked "in flight" from `depth` frames ago: hand it back to the producer first
    bool fly;
    { std::lock_guard<std::mutex> lk(g->mu); fly = g->state[slot] == 2; }
    if (fly) {
      if (!wait && hipEventQuery(g->done[slot]) != hipSuccess) return 1;
      if (hipEventSynchronize(g->done[slot]) != hipSuccess) { set_error("sl2_ingest_next: hipEventSynchronize failed"); return -SL2_ERR_HIP; }
      { std::lock_guard<std::mutex> lk(g->mu); g->state[slot] = 0; }
      g->cv.notify_all();
    }
  }
  {
    std::unique_lock<std::mutex> lk(g->mu);
    if (!wait && g->state[slot] != 1) return 1;
    // a decode failure further ahead does not fail THIS frame if its batch is already decoded
    g->cv.wait(lk, [&] { return g->failed || g->state[slot] == 1; });
    if (g->state[slot] != 1) { set_error(g->fail_msg.c_str()); return -SL2_ERR_INVALID; }
  }
  const size_t batch = (size_t)g->nseq * g->width * g->height;
  // dev[b] was last read by the caller's work on frame k - 2: the copy stream waits until the caller's stream is past it
  if (g->handed_out[b] && hipStreamWaitEvent(g->copy_stream, g->reusable[b], 0) != hipSuccess) { set_error("sl2_ingest_next: hipStreamWaitEvent failed"); return -SL2_ERR_HIP; }
  if (hipMemcpyAsync(g->dev[b], g->host[slot], batch, hipMemcpyHostToDevice, g->copy_stream) != hipSuccess ||
      hipEventRecord(g->done[slot], g->copy_stream) != hipSuccess) {       // (one event: the pinned slot is free AND the device buffer is filled)
    set_error("sl2_ingest_next: the upload could not be queued");
    return -SL2_ERR_HIP;
  }
  { std::lock_guard<std::mutex> lk(g->mu); g->state[slot] = 2; }
  g->issued = k + 1;
  return 0;
}

int sl2_ingest_set_zero_copy(sl2_ingest* g, size_t max_batch_bytes) {
  if (!g) return SL2_ERR_INVALID;
  if (g->consumed > 0) { sl2::set_error("sl2_ingest_set_zero_copy: frames have been handed out already"); return SL2_ERR_INVALID; }
  g->zc_max = max_batch_bytes;
  return SL2_OK;
}

// The zero-copy hand-over of frame k: the pinned batch itself, as the device sees it.
static int next_zero_copy(sl2_ingest* g, hipStream_t st, const uint8_t** d_frames) {
  using namespace sl2;
  const int k = g->consumed, slot = k % g->depth;
  {
    // the slot may still be with the device from `depth` frames ago (a caller that queues far ahead): wait for that consumer -
    // its event was recorded depth - 1 calls ago - and hand the slot to the producer before waiting for the producer
    bool fly;
    { std::lock_guard<std::mutex> lk(g->mu); fly = g->state[slot] == 2; }
    if (fly) {
      if (g->used_valid[slot]) SL2_HIP(hipEventSynchronize(g->done[slot]));
      { std::lock_guard<std::mutex> lk(g->mu); g->state[slot] = 0; }
      g->cv.notify_all();
    }
  }
  {
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return g->failed || g->state[slot] == 1; });
    if (g->state[slot] != 1) { set_error(g->fail_msg.c_str()); return SL2_ERR_INVALID; }
    g->state[slot] = 2;
  }
  g->used_valid[slot] = 0;
  if (k >= 1) {                                   // everything that read the previous frame's batch is queued on the caller's stream by now
    const int prev = (k - 1) % g->depth;
    SL2_HIP(hipEventRecord(g->done[prev], st));
    g->used_valid[prev] = 1;
  }
  for (int back = 2; back <= 3 && back <= k; ++back) {      // older batches whose consumers have finished go back to the producer
    const int old = (k - back) % g->depth;
    bool f2;
    { std::lock_guard<std::mutex> lk(g->mu); f2 = g->state[old] == 2; }
    if (f2 && g->used_valid[old] && hipEventQuery(g->done[old]) == hipSuccess) {
      { std::lock_guard<std::mutex> lk(g->mu); g->state[old] = 0; }
      g->cv.notify_all();
    }
  }
  ++g->consumed;
  *d_frames = g->host_dev[slot];
  return SL2_OK;
}

int sl2_ingest_next(sl2_ingest* g, void* stream, const uint8_t** d_frames, size_t* seq_stride) {
  using namespace sl2;
  if (!g || !d_frames || !seq_stride) return SL2_ERR_INVALID;
  if (g->consumed >= g->n_frames) return SL2_ERR_CAPACITY;     // end of the shortest sequence
  hipStream_t st = (hipStream_t)stream;
  if ((size_t)g->nseq * g->width * g->height <= g->zc_max && g->depth >= 4) {
    *seq_stride = (size_t)g->width * g->height;
    return next_zero_copy(g, st, d_frames);
  }
  SL2_HIP(hipSetDevice(g->device));
  const int k = g->consumed, b = k & 1;
  if (g->issued <= k) {                       // not prefetched (the first frame, or the producer had not decoded it in time)
    const int rc = issue_copy(g, k, true);
    if (rc < 0) return -rc;
  }
  SL2_HIP(hipStreamWaitEvent(st, g->done[k % g->depth], 0));   // the caller's work on frame k queues behind its copy
  g->handed_out[b] = true;
  ++g->consumed;
  // ONE AHEAD: frame k + 1 goes to the other buffer, which the caller's work on frame k - 1 read - everything queued on the
  // caller's stream so far.  Behind that point the buffer may be overwritten; the caller's work on frame k, queued after this
  // call returns, runs beside the copy.
  if (k + 1 < g->n_frames) {
    if (g->handed_out[b ^ 1]) SL2_HIP(hipEventRecord(g->reusable[b ^ 1], st));
    const int rc = issue_copy(g, k + 1, false);
    if (rc < 0) { /* reported by the call whose frame it is */ (void)hipGetLastError(); }
  }
  // The pinned batch of the frame handed out LAST time goes back to the producer: its copy was waited for by the caller's stream a
  // whole call ago, and everything queued behind that wait has been issued since - one query, not one per slot (a HIP call is
  // ~1 us; this function was 12 us of a 70 us frame).  The slot this call used stays "in flight" until the next call.
  for (int back = 1; back <= 2 && back <= k; ++back) {      // (the one before as well, should its copy not have landed a call ago)
    const int prev = (k - back) % g->depth;
    bool f2;
    { std::lock_guard<std::mutex> lk(g->mu); f2 = g->state[prev] == 2; }
    if (f2 && hipEventQuery(g->done[prev]) == hipSuccess) {
      { std::lock_guard<std::mutex> lk(g->mu); g->state[prev] = 0; }
      g->cv.notify_all();
    }
  }
  *d_frames = g->dev[b];
  *seq_stride = (size_t)g->width * g->height;
  return SL2_OK;
}

void sl2_ingest_close(sl2_ingest* g) {
  if (!g) return;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->stop = true;
  }
  g->cv.notify_all();
  if (g->producer.joinable()) g->producer.join();
  hipSetDevice(g->device);
  hipDeviceSynchronize();
  for (auto p : g->host) if (p) hipHostFree(p);
  for (auto e : g->done) if (e) hipEventDestroy(e);
  for (int i = 0; i < 2; ++i) {
    if (g->dev[i]) hipFree(g->dev[i]);
    if (g->reusable[i]) hipEventDestroy(g->reusable[i]);
  }
  if (g->copy_stream) hipStreamDestroy(g->copy_stream);
  delete g;
}

}  // extern "C"
