// pfm_io.h -- float disparity maps as single-channel PFM ("Pf", little-endian, rows bottom-up), the format the
// Middlebury / KITTI tooling reads.  The reference writes only 8-bit maps quantised by dis_scale
// (CSPM/main.cc:133-134, cs_patchmatch.cc:590-601); the float file keeps the sub-pixel plane disparities.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

inline bool WritePFM(const std::string &path, const double *disp, int w, int h) {
  FILE *fp = std::fopen(path.c_str(), "wb");
  if (!fp) return false;
  std::fprintf(fp, "Pf\n%d %d\n-1.0\n", w, h);  // negative scale = little-endian
  std::vector<float> row((size_t)w);
  bool ok = true;
  for (int y = h - 1; y >= 0 && ok; --y) {
    for (int x = 0; x < w; ++x) row[x] = (float)disp[(size_t)y * w + x];
    ok = std::fwrite(row.data(), sizeof(float), (size_t)w, fp) == (size_t)w;
  }
  return std::fclose(fp) == 0 && ok;
}
