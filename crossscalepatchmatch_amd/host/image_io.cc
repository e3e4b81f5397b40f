// image_io.cc -- cv::imread / cv::imwrite for the CLI (main.cc:68-69,133-134): 8-bit PNG through zlib (no
// libpng headers in this image) and binary PGM/PPM.  Host-side I/O only.
#include "cv_compat.h"
#ifndef CSPM_USE_OPENCV
#include <zlib.h>

namespace {

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<unsigned char> &v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s)); }

bool read_file(const std::string &path, std::vector<unsigned char> &out) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? n : 0);
  bool ok = n >= 0 && std::fread(out.data(), 1, out.size(), f) == out.size();
  std::fclose(f);
  return ok;
}

int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

cv::Mat read_png(const std::vector<unsigned char> &buf) {
  static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8)) return cv::Mat();
  size_t pos = 8;
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<unsigned char> idat, plte;
  while (pos + 12 <= buf.size()) {
    uint32_t len = be32(&buf[pos]);
    std::string tag((const char *)&buf[pos + 4], 4);
    if (pos + 12 + len > buf.size()) return cv::Mat();
    const unsigned char *d = &buf[pos + 8];
    if (tag == "IHDR") { W = be32(d); H = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
    else if (tag == "PLTE") plte.assign(d, d + len);
    else if (tag == "IDAT") idat.insert(idat.end(), d, d + len);
    else if (tag == "IEND") break;
    pos += 12 + len;
  }
  if (!W || !H || depth != 8 || interlace != 0) return cv::Mat();
  int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!ch) return cv::Mat();
  std::vector<unsigned char> raw((size_t)H * (1 + (size_t)W * ch));
  uLongf rawlen = raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) return cv::Mat();
  const size_t stride = (size_t)W * ch;
  std::vector<unsigned char> prev(stride, 0), cur(stride);
  cv::Mat out((int)H, (int)W, CV_8UC3);
  for (uint32_t y = 0; y < H; ++y) {
    const unsigned char *row = &raw[y * (stride + 1)];
    const int ft = row[0];
    for (size_t i = 0; i < stride; ++i) {
      int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0, x = row[1 + i];
      switch (ft) {
        case 0: break;
        case 1: x += a; break;
        case 2: x += b; break;
        case 3: x += (a + b) / 2; break;
        case 4: x += paeth(a, b, c); break;
        default: return cv::Mat();
      }
      cur[i] = (unsigned char)x;
    }
    unsigned char *o = out.ptr<unsigned char>((int)y);
    for (uint32_t x = 0; x < W; ++x) {
      unsigned char r, g, b;
      if (ctype == 0 || ctype == 4) r = g = b = cur[x * ch];
      else if (ctype == 3) {
        size_t k = 3 * (size_t)cur[x];
        if (k + 2 >= plte.size()) return cv::Mat();
        r = plte[k]; g = plte[k + 1]; b = plte[k + 2];
      } else { r = cur[x * ch]; g = cur[x * ch + 1]; b = cur[x * ch + 2]; }
      o[3 * x] = b; o[3 * x + 1] = g; o[3 * x + 2] = r;  // BGR, as cv::imread
    }
    prev.swap(cur);
  }
  return out;
}

bool write_png(const std::string &path, const cv::Mat &img) {
  const int ch = img.channels();
  std::vector<unsigned char> raw;
  raw.reserve((size_t)img.rows * (1 + (size_t)img.cols * ch));
  for (int y = 0; y < img.rows; ++y) {
    raw.push_back(0);
    const unsigned char *p = img.ptr<unsigned char>(y);
    for (int x = 0; x < img.cols; ++x)
      if (ch == 1) raw.push_back(p[x]);
      else { raw.push_back(p[3 * x + 2]); raw.push_back(p[3 * x + 1]); raw.push_back(p[3 * x]); }
  }
  uLongf clen = compressBound(raw.size());
  std::vector<unsigned char> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) return false;
  std::vector<unsigned char> out = {137, 80, 78, 71, 13, 10, 26, 10};
  auto chunk = [&](const char *tag, const std::vector<unsigned char> &d) {
    put32(out, (uint32_t)d.size());
    size_t s = out.size();
    out.insert(out.end(), tag, tag + 4);
    out.insert(out.end(), d.begin(), d.end());
    put32(out, (uint32_t)crc32(0, &out[s], (uInt)(out.size() - s)));
  };
  std::vector<unsigned char> ihdr;
  put32(ihdr, img.cols); put32(ihdr, img.rows);
  ihdr.push_back(8); ihdr.push_back(ch == 1 ? 0 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk("IHDR", ihdr);
  comp.resize(clen);
  chunk("IDAT", comp);
  chunk("IEND", {});
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  std::fclose(f);
  return ok;
}

cv::Mat read_pnm(const std::vector<unsigned char> &buf) {
  if (buf.size() < 7 || buf[0] != 'P' || (buf[1] != '5' && buf[1] != '6')) return cv::Mat();
  size_t pos = 2;
  int vals[3], n = 0;
  while (n < 3 && pos < buf.size()) {
    while (pos < buf.size() && (std::isspace(buf[pos]) || buf[pos] == '#')) {
      if (buf[pos] == '#') while (pos < buf.size() && buf[pos] != '\n') ++pos;
      else ++pos;
    }
    int v = 0;
    bool any = false;
    while (pos < buf.size() && std::isdigit(buf[pos])) { v = v * 10 + (buf[pos++] - '0'); any = true; }
    if (!any) return cv::Mat();
    vals[n++] = v;
  }
  ++pos;  // single whitespace after maxval
  const int W = vals[0], H = vals[1], ch = buf[1] == '6' ? 3 : 1;
  if (n < 3 || vals[2] != 255 || pos + (size_t)W * H * ch > buf.size()) return cv::Mat();
  cv::Mat out(H, W, CV_8UC3);
  for (int y = 0; y < H; ++y) {
    const unsigned char *s = &buf[pos + (size_t)y * W * ch];
    unsigned char *o = out.ptr<unsigned char>(y);
    for (int x = 0; x < W; ++x) {
      if (ch == 1) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = s[x];
      else { o[3 * x] = s[3 * x + 2]; o[3 * x + 1] = s[3 * x + 1]; o[3 * x + 2] = s[3 * x]; }
    }
  }
  return out;
}

bool ends_with(const std::string &s, const char *suf) {
  std::string t = s;
  std::transform(t.begin(), t.end(), t.begin(), ::tolower);
  size_t n = std::strlen(suf);
  return t.size() >= n && t.compare(t.size() - n, n, suf) == 0;
}

}  // namespace

namespace cv {

Mat imread(const std::string &path, int) {
  std::vector<unsigned char> buf;
  if (!read_file(path, buf)) return Mat();
  Mat m = read_png(buf);
  if (m.empty()) m = read_pnm(buf);
  return m;
}

bool imwrite(const std::string &path, const Mat &img) {
  if (img.empty() || img.depth() != CV_8U || (img.channels() != 1 && img.channels() != 3)) return false;
  if (ends_with(path, ".pgm") || ends_with(path, ".ppm") || ends_with(path, ".pnm")) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P%c\n%d %d\n255\n", img.channels() == 1 ? '5' : '6', img.cols, img.rows);
    for (int y = 0; y < img.rows; ++y) {
      const unsigned char *p = img.ptr<unsigned char>(y);
      if (img.channels() == 1) std::fwrite(p, 1, img.cols, f);
      else for (int x = 0; x < img.cols; ++x) { unsigned char rgb[3] = {p[3 * x + 2], p[3 * x + 1], p[3 * x]}; std::fwrite(rgb, 1, 3, f); }
    }
    std::fclose(f);
    return true;
  }
  return write_png(path, img);
}

}  // namespace cv
#endif
