// rthost_sky.cpp — host-side sky texture loading, the part of loadSkySphere
// (/root/reference/app/Rendering/SkySphere.hs:18-29) that this mirror can do without JuicyPixels:
// Radiance .hdr (RGBE) -> float RGB (PixelRGBF); PNG / BMP / TGA (and PPM, for tests) -> 8-bit RGB (PixelRGB8, as convertRGB8).
// Row 0 = top, as pixelAt.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
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
// Codec.Picture.HDR.toFloat as recalled:
//     f = encodeFloat 1 (e - (128 + 8));  rf = (fromIntegral r + 0.0) * f   (likewise g, b)
// - mantissa * 2^(e - 136) for every exponent (e = 0 gives 0 only because the mantissas of a black texel are 0), WITHOUT the
// half-step offset of Radiance's own (m + 0.5) reader.  Rounds 1-2 of this repository recalled the line with + 0.5; the `+ 0.0`
// is the more specific memory (it reads like the trace of exactly that edit), so it is the default now, and RTH_RGBE_HALF=1 in
// the environment selects the other reading: 0.2-0.4 % brighter on a texel's dominant channel, more on the small mantissas.
// Whichever JuicyPixels does, the render core is handed floats and is exact for them; only this loader is in question.
float rgbe_offset() {
  static const float off = [] { const char *e = std::getenv("RTH_RGBE_HALF"); return e && e[0] == '1' ? 0.5f : 0.0f; }();
  return off;
}
void rgbe_to_float(const unsigned char *p, float *out) {
  const float f = std::ldexp(1.0f, (int)p[3] - (128 + 8)), off = rgbe_offset();
  out[0] = (p[0] + off) * f; out[1] = (p[1] + off) * f; out[2] = (p[2] + off) * f;
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

// ---------------------------------------------------------------- the LDR formats JuicyPixels' readImage also accepts: PNG, BMP, TGA
// loadSkySphere passes every non-".hdr" file through readImage + convertRGB8 (SkySphere.hs:20-29).  PARITY UNPINNED like the RGBE
// rule above (JuicyPixels is not in the reference checkout): restated from its documented behaviour - convertRGB8 drops an
// alpha channel without blending, replicates grey into three channels, takes the HIGH byte of 16-bit samples (from16to8),
// and its PNG decoder expands palettes and scales 1/2/4-bit grey as i * 255 div (2^depth - 1); gAMA and colour profiles are ignored.
// JPEG, GIF and TIFF (which readImage accepts too) stay RT_ERR_UNSUPPORTED.
std::vector<unsigned char> read_all(FILE *f) {
  std::vector<unsigned char> b;
  unsigned char tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n);
  return b;
}
uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

int load_png(const std::vector<unsigned char> &buf, int32_t *w, int32_t *h, void **pixels) {
  size_t pos = 8;
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<unsigned char> idat, plte;
  bool have_ihdr = false;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = be32(&buf[pos]);
    const unsigned char *type = &buf[pos + 4], *data = &buf[pos + 8];
    if (pos + 12 + (size_t)len > buf.size()) return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated PNG chunk");
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; have_ihdr = true;
      if (data[10] != 0 || data[11] != 0) return fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: unknown PNG compression / filter method");
    } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "IEND", 4)) break;
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || W == 0 || H == 0 || W > (1u << 20) || H > (1u << 20)) return fail(RT_ERR_INVALID, "Failed to load sky texture: bad PNG header");
  const int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                      : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
  if (!chans || !depth_ok || interlace > 1) return fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: unsupported PNG colour type %d / depth %d", ctype, depth);
  if (ctype == 3 && plte.size() < 3) return fail(RT_ERR_INVALID, "Failed to load sky texture: palette PNG without PLTE");
  const int bpp_bits = chans * depth, fbpp = std::max(1, bpp_bits / 8);           // bytes per complete pixel for the filters
  // passes: one for a plain image, seven for Adam7
  static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
  const int n_pass = interlace ? 7 : 1;
  size_t raw_size = 0;
  for (int p = 0; p < n_pass; ++p) {
    const uint32_t pw = interlace ? (W + adx[p] - 1 - ax0[p]) / adx[p] : W, ph = interlace ? (H + ady[p] - 1 - ay0[p]) / ady[p] : H;
    if (pw && ph) raw_size += (size_t)ph * (1 + ((size_t)pw * bpp_bits + 7) / 8);
  }
  std::vector<unsigned char> raw(raw_size);
  uLongf got = (uLongf)raw_size;
  if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw_size)
    return fail(RT_ERR_INVALID, "Failed to load sky texture: PNG data does not inflate to the image size");
  unsigned char *out = (unsigned char *)malloc((size_t)W * H * 3);
  if (!out) return fail(RT_ERR_OOM, "sky texture");
  size_t rp = 0;
  for (int p = 0; p < n_pass; ++p) {
    const uint32_t pw = interlace ? (W + adx[p] - 1 - ax0[p]) / adx[p] : W, ph = interlace ? (H + ady[p] - 1 - ay0[p]) / ady[p] : H;
    if (!pw || !ph) continue;
    const size_t stride = ((size_t)pw * bpp_bits + 7) / 8;
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    for (uint32_t y = 0; y < ph; ++y) {
      const int ft = raw[rp++];
      const unsigned char *src = &raw[rp];
      rp += stride;
      for (size_t i = 0; i < stride; ++i) {                                       // un-filter (PNG spec 9.2)
        const int a = i >= (size_t)fbpp ? cur[i - fbpp] : 0, b = prev[i], c = i >= (size_t)fbpp ? prev[i - fbpp] : 0;
        int v = src[i];
        switch (ft) {
          case 0: break;
          case 1: v += a; break;
          case 2: v += b; break;
          case 3: v += (a + b) >> 1; break;
          case 4: v += paeth(a, b, c); break;
          default: free(out); return fail(RT_ERR_INVALID, "Failed to load sky texture: bad PNG filter type %d", ft);
        }
        cur[i] = (unsigned char)v;
      }
      const uint32_t oy = interlace ? ay0[p] + y * ady[p] : y;
      for (uint32_t x = 0; x < pw; ++x) {
        const uint32_t ox = interlace ? ax0[p] + x * adx[p] : x;
        unsigned char *o = out + ((size_t)oy * W + ox) * 3;
        auto sample = [&](int ch) -> int {                                        // 8 significant bits of channel ch (16-bit: the high byte)
          if (depth == 16) return cur[((size_t)x * chans + ch) * 2];
          if (depth == 8) return cur[(size_t)x * chans + ch];
          const size_t bit = (size_t)x * depth;
          return (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        };
        if (ctype == 3) {
          const size_t k = (size_t)sample(0) * 3;
          if (k + 2 < plte.size()) { o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2]; } else { o[0] = o[1] = o[2] = 0; }
        } else if (ctype == 0 || ctype == 4) {
          int g = sample(0);
          if (depth < 8) g = g * 255 / ((1 << depth) - 1);
          o[0] = o[1] = o[2] = (unsigned char)g;
        } else { o[0] = (unsigned char)sample(0); o[1] = (unsigned char)sample(1); o[2] = (unsigned char)sample(2); }
      }
      prev.swap(cur);
    }
  }
  *w = (int32_t)W; *h = (int32_t)H; *pixels = out;
  return RT_OK;
}

int load_bmp(const std::vector<unsigned char> &b, int32_t *w, int32_t *h, void **pixels) {
  auto le32 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8) | ((uint32_t)b[o + 2] << 16) | ((uint32_t)b[o + 3] << 24); };
  auto le16 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8); };
  if (b.size() < 54) return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated BMP");
  const uint32_t off = le32(10), hdr = le32(14);
  const int32_t W = (int32_t)le32(18), Hs = (int32_t)le32(22);
  const uint32_t bpp = le16(28), comp = le32(30);
  const int32_t H = Hs < 0 ? -Hs : Hs;
  if (hdr < 40 || W <= 0 || H <= 0 || (bpp != 24 && bpp != 32) || (comp != 0 && !(comp == 3 && bpp == 32)))
    return fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: only uncompressed 24 / 32-bit BMP is supported by this host");
  const size_t stride = (((size_t)W * bpp + 31) / 32) * 4;
  if ((size_t)off + stride * H > b.size()) return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated BMP");
  unsigned char *out = (unsigned char *)malloc((size_t)W * H * 3);
  if (!out) return fail(RT_ERR_OOM, "sky texture");
  for (int32_t y = 0; y < H; ++y) {
    const unsigned char *row = &b[off + stride * (size_t)(Hs < 0 ? y : H - 1 - y)];       // bottom-up unless the height is negative
    for (int32_t x = 0; x < W; ++x) {
      const unsigned char *px = row + (size_t)x * (bpp / 8);
      unsigned char *o = out + ((size_t)y * W + x) * 3;
      o[0] = px[2]; o[1] = px[1]; o[2] = px[0];                                              // stored B, G, R(, A)
    }
  }
  *w = W; *h = H; *pixels = out;
  return RT_OK;
}

int load_tga(const std::vector<unsigned char> &b, int32_t *w, int32_t *h, void **pixels) {
  if (b.size() < 18) return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated TGA");
  const int idlen = b[0], cmap = b[1], type = b[2], bpp = b[16], desc = b[17];
  const int32_t W = b[12] | (b[13] << 8), H = b[14] | (b[15] << 8);
  const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
  if (cmap != 0 || !(type == 2 || type == 3 || type == 10 || type == 11) || W <= 0 || H <= 0 || !(grey ? bpp == 8 : (bpp == 24 || bpp == 32)))
    return fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: only true-colour / grey TGA (types 2, 3, 10, 11) is supported by this host");
  const int nb = bpp / 8;
  size_t pos = 18 + (size_t)idlen;
  std::vector<unsigned char> px((size_t)W * H * nb);
  if (!rle) {
    if (pos + px.size() > b.size()) return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated TGA");
    memcpy(px.data(), &b[pos], px.size());
  } else {
    size_t o = 0;
    while (o < px.size()) {
      if (pos >= b.size()) return fail(RT_ERR_INVALID, "Failed to load sky texture: truncated TGA");
      const int hd = b[pos++], n = (hd & 127) + 1;
      if (o + (size_t)n * nb > px.size()) return fail(RT_ERR_INVALID, "Failed to load sky texture: bad TGA run");
      if (hd & 128) { if (pos + nb > b.size()) return fail(RT_ERR_INVALID, "truncated TGA"); for (int k = 0; k < n; ++k) { memcpy(&px[o], &b[pos], nb); o += nb; } pos += nb; }
      else { if (pos + (size_t)n * nb > b.size()) return fail(RT_ERR_INVALID, "truncated TGA"); memcpy(&px[o], &b[pos], (size_t)n * nb); o += (size_t)n * nb; pos += (size_t)n * nb; }
    }
  }
  unsigned char *out = (unsigned char *)malloc((size_t)W * H * 3);
  if (!out) return fail(RT_ERR_OOM, "sky texture");
  const bool top = (desc & 0x20) != 0, right = (desc & 0x10) != 0;
  for (int32_t y = 0; y < H; ++y)
    for (int32_t x = 0; x < W; ++x) {
      const unsigned char *s = &px[((size_t)(top ? y : H - 1 - y) * W + (right ? W - 1 - x : x)) * nb];
      unsigned char *o = out + ((size_t)y * W + x) * 3;
      if (grey) o[0] = o[1] = o[2] = s[0]; else { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; }
    }
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
  else {                                                                                       // :28-29 readImage + convertRGB8: by content
    *format = RT_SKY_RGB8;
    unsigned char magic[8] = {0};
    const size_t got = fread(magic, 1, 8, f);
    rewind(f);
    static const unsigned char png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (got == 8 && !memcmp(magic, png_sig, 8)) rc = load_png(read_all(f), w, h, pixels);
    else if (got >= 2 && magic[0] == 'B' && magic[1] == 'M') rc = load_bmp(read_all(f), w, h, pixels);
    else if (got >= 2 && magic[0] == 'P' && (magic[1] == '3' || magic[1] == '6')) rc = load_ppm(f, w, h, pixels);
    else if (ends_with(path, ".tga")) rc = load_tga(read_all(f), w, h, pixels);               // (TGA has no signature)
    else rc = fail(RT_ERR_UNSUPPORTED, "Failed to load sky texture: this host reads .hdr, PNG, BMP, TGA and PPM (JuicyPixels also reads JPEG, GIF, TIFF)");
  }
  fclose(f);
  return rc;
}
void rth_sky_free(void *pixels) { free(pixels); }
}
