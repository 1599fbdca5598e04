// rthost_ppm.cpp — the P3 PPM body exactly as the reference's writer thread emits it:
//   header  "P3\n" ++ show imgW ++ " " ++ show imgH ++ "\n255\n"   app/Rendering/ImageGenerator.hs:100
//   rows    unlines (map showPixel pixels)                          app/Rendering/RenderWorker.hs:72
//   pixel   unwords (map show [r,g,b])                              app/Rendering/RenderWorker.hs:108-109
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rthost.h"

namespace rth { int fail(int code, const char *fmt, ...); }
using rth::fail;

namespace {
// appends the decimal text of v (0..255) at p, returns the new end
inline char *put_u8(char *p, unsigned v) {
  if (v >= 100) { *p++ = (char)('0' + v / 100); v %= 100; *p++ = (char)('0' + v / 10); *p++ = (char)('0' + v % 10); }
  else if (v >= 10) { *p++ = (char)('0' + v / 10); *p++ = (char)('0' + v % 10); }
  else *p++ = (char)('0' + v);
  return p;
}
}  // namespace

extern "C" {

int64_t rth_format_ppm(int32_t W, int32_t H, const uint8_t *rgb8, char *out, int64_t cap) {
  if (W <= 0 || H <= 0 || !rgb8) return fail(RT_ERR_INVALID, "bad image");
  char header[64];
  int hl = snprintf(header, sizeof header, "P3\n%d %d\n255\n", W, H);
  const size_t npx = (size_t)W * H;
  int64_t need = hl;
  for (size_t i = 0; i < 3 * npx; ++i) need += (rgb8[i] >= 100 ? 3 : rgb8[i] >= 10 ? 2 : 1) + 1;   // digits + ' ' or '\n'
  if (!out || cap < need) return need;
  char *p = out;
  std::memcpy(p, header, hl); p += hl;
  for (size_t i = 0; i < npx; ++i) {
    p = put_u8(p, rgb8[3 * i]); *p++ = ' ';
    p = put_u8(p, rgb8[3 * i + 1]); *p++ = ' ';
    p = put_u8(p, rgb8[3 * i + 2]); *p++ = '\n';
  }
  return need;
}

int rth_write_ppm(const char *path, int32_t W, int32_t H, const uint8_t *rgb8) {
  if (!path) return fail(RT_ERR_INVALID, "null path");
  int64_t need = rth_format_ppm(W, H, rgb8, nullptr, 0);
  if (need < 0) return (int)need;
  std::vector<char> buf((size_t)need);
  rth_format_ppm(W, H, rgb8, buf.data(), need);
  FILE *f = fopen(path, "wb");
  if (!f) return fail(RT_ERR_INVALID, "cannot open %s for writing", path);
  size_t w = fwrite(buf.data(), 1, buf.size(), f);
  fclose(f);
  if (w != buf.size()) return fail(RT_ERR_INVALID, "short write to %s", path);
  return RT_OK;
}

}  // extern "C"
