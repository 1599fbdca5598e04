// rthost_sky.cpp — host-side sky texture loading, the part of loadSkySphere
// (/root/reference/app/Rendering/SkySphere.hs:18-29) that this mirror can do without JuicyPixels:
// Radiance .hdr (RGBE) -> float RGB (PixelRGBF), PPM -> 8-bit RGB (PixelRGB8).  Row 0 = top, as pixelAt.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rthost.h"

namespace rth { int fail(int code, const char *fmt, ...); }
using rth::fail;

namespace {
bool ends_with(const std::string &s, const char *suf) {
  size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
// RGBE -> float.  PARITY UNPINNED: the reference decodes through JuicyPixels' readImage, whose source is not part of the
// reference checkout and cannot be imported here, and the reference holds no decoded fixture.  This follows JuicyPixels'
// Codec.Picture.HDR.toFloat as recalled - (mantissa + 0.5) * 2^(e - 136) for every exponent, with no special case for
// e = 0 (which yields a denormal ~1e-41 rather than 0: invisible next to the 1e-12 of the parity tests) - and the only
// thing tests/test_config_loader.py pins is this formula itself.  If JuicyPixels truncates instead (mantissa * 2^(e - 136))
// every texel here is half a mantissa step too bright: 0.2-0.4 % on the dominant channel.
void rgbe_to_float(const unsigned char *p, float *out) {
  float f = std::ldexp(1.0f, (int)p[3] - (128 + 8));
  out[0] = (p[0] + 0.5f) * f; out[1] = (p[1] + 0.5f) * f; out[2] = (p[2] + 0.5f) * f;
}
int load_hdr(FILE *f, int32_t *w, int32_t *h, void **pixels) {
  char line[512];
  bool fmt_ok = false;
  if (!fgets(line, sizeof line, f) || strncmp(line, "#?", 2) != 0) return fail(RT_ERR_INVALID, "Failed to load sky texture: not a Radiance file");
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '\n' || (line[0] == '\r' && line[1] == '\n')) break;
    if (!strncmp(line, "FORMAT=32-bit_rle_rgbe", 22)) fmt_ok = true;
  }
  if (!fmt_ok) return fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: only FORMAT=32-bit_rle_rgbe is supported");
  int W = 0, H = 0;
  if (!fgets(line, sizeof line, f) || sscanf(line, "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0)
    return fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: unsupported resolution line");
  float *out = (float *)malloc((size_t)W * H * 3 * sizeof(float));
  if (!out) return fail(RT_ERR_OOM, "sky texture");
  std::vector<unsigned char> scan((size_t)W * 4);
  for (int y = 0; y < H; ++y) {
    unsigned char hd[4];
    if (fread(hd, 1, 4, f) != 4) { free(out); return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated file"); }
    if (W >= 8 && W < 32768 && hd[0] == 2 && hd[1] == 2 && ((hd[2] << 8) | hd[3]) == W) {   // new-style RLE, per channel
      for (int c = 0; c < 4; ++c) {
        int x = 0;
        while (x < W) {
          int n = fgetc(f);
          if (n == EOF) { free(out); return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated file"); }
          if (n > 128) { n -= 128; int v = fgetc(f); if (x + n > W || v == EOF) { free(out); return fail(RT_ERR_INVALID, "bad RLE run"); } while (n--) scan[4 * (x++) + c] = (unsigned char)v; }
          else { if (n == 0 || x + n > W) { free(out); return fail(RT_ERR_INVALID, "bad RLE run"); } while (n--) { int v = fgetc(f); if (v == EOF) { free(out); return fail(RT_ERR_INVALID, "truncated"); } scan[4 * (x++) + c] = (unsigned char)v; } }
        }
      }
    } else {                                                                                  // flat pixels
      memcpy(scan.data(), hd, 4);
      if (W > 1 && fread(scan.data() + 4, 4, (size_t)W - 1, f) != (size_t)W - 1) { free(out); return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated file"); }
    }
    for (int x = 0; x < W; ++x) rgbe_to_float(&scan[4 * x], out + ((size_t)y * W + x) * 3);
  }
  *w = W; *h = H; *pixels = out;
  return RT_OK;
}
int ppm_token(FILE *f, int *v) {
  int c;
  for (;;) {
    c = fgetc(f);
    if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
    if (c == EOF) return 0;
    if (!isspace(c)) break;
  }
  int n = 0, any = 0;
  while (c != EOF && isdigit(c)) { n = n * 10 + (c - '0'); any = 1; c = fgetc(f); }
  *v = n;
  return any;
}
int load_ppm(FILE *f, int32_t *w, int32_t *h, void **pixels) {
  int c0 = fgetc(f), c1 = fgetc(f);
  if (c0 != 'P' || (c1 != '6' && c1 != '3')) return fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: only .hdr and PPM (P3/P6) are supported by this host");
  int W, H, mx;
  if (!ppm_token(f, &W) || !ppm_token(f, &H) || !ppm_token(f, &mx) || W <= 0 || H <= 0 || mx != 255)
    return fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: bad PPM header (maxval must be 255)");
  size_t n = (size_t)W * H * 3;
  unsigned char *out = (unsigned char *)malloc(n);
  if (!out) return fail(RT_ERR_OOM, "sky texture");
  if (c1 == '6') { if (fread(out, 1, n, f) != n) { free(out); return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated PPM"); } }
  else for (size_t i = 0; i < n; ++i) { int v; if (!ppm_token(f, &v)) { free(out); return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated PPM"); } out[i] = (unsigned char)v; }
  *w = W; *h = H; *pixels = out;
  return RT_OK;
}
}  // namespace

extern "C" {
int rth_sky_load(const char *path, int32_t *format, int32_t *w, int32_t *h, void **pixels) {
  if (!path || !format || !w || !h || !pixels) return fail(RT_ERR_INVALID, "null argument");
  FILE *f = fopen(path, "rb");
  if (!f) return fail(RT_ERR_INVALID, "Failed to load sky texture: %s: does not exist", path);
  int rc;
  if (ends_with(path, ".hdr")) { rc = load_hdr(f, w, h, pixels); *format = RT_SKY_RGBF; }     // SkySphere.hs:24-27
  else { rc = load_ppm(f, w, h, pixels); *format = RT_SKY_RGB8; }                             // :28-29
  fclose(f);
  return rc;
}
void rth_sky_free(void *pixels) { free(pixels); }
}
