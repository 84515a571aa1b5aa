// See orb_plan.h.  Pure host code (no HIP).
#include "orb_plan.h"
#include "sd_common.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace sd {

static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// reference src/ORBextractor.cc:406-434 (note: the member scaleFactor is a double holding the
// float ctor argument, src/ORBextractor.h:38,78 -- the products below promote accordingly)
void plan_tables(int nfeatures, float scaleFactorArg, int nlevels, HostPlan& hp) {
  const double scaleFactor = scaleFactorArg;
  hp.sf.assign(nlevels, 1.0f);
  hp.sigma2.assign(nlevels, 1.0f);
  for (int i = 1; i < nlevels; i++) {
    hp.sf[i] = (float)(hp.sf[i - 1] * scaleFactor);
    hp.sigma2[i] = hp.sf[i] * hp.sf[i];
  }
  hp.inv_sf.resize(nlevels);
  hp.inv_sigma2.resize(nlevels);
  for (int i = 0; i < nlevels; i++) {
    hp.inv_sf[i] = 1.0f / hp.sf[i];
    hp.inv_sigma2[i] = 1.0f / hp.sigma2[i];
  }
  hp.quota.assign(nlevels, 0);
  float factor = (float)(1.0f / scaleFactor);
  float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    hp.quota[l] = cv_round(nDesired);
    sum += hp.quota[l];
    nDesired *= factor;
  }
  hp.quota[nlevels - 1] = std::max(nfeatures - sum, 0);
}

// cv::resize INTER_LINEAR coefficient tables (OpenCV 3.2 semantics)
static void resize_tables(int sn, int dn, std::vector<int32_t>& ofs, std::vector<int32_t>& ab, bool clamp_x) {
  double inv_scale = (double)dn / sn;
  double scale = 1. / inv_scale;
  ofs.resize(dn);
  ab.resize(dn);
  for (int d = 0; d < dn; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cv_floor(f);
    f -= s;
    if (clamp_x) {
      if (s < 0) { f = 0; s = 0; }
      if (s + 1 >= sn && s >= sn - 1) { f = 0; s = sn - 1; }
    }
    float c0 = 1.f - f, c1 = f;
    int a0 = std::min(std::max(cv_round(c0 * 2048.f), -32768), 32767);
    int a1 = std::min(std::max(cv_round(c1 * 2048.f), -32768), 32767);
    ofs[d] = s;
    ab[d] = (int32_t)((uint32_t)(uint16_t)(int16_t)a0 | ((uint32_t)(uint16_t)(int16_t)a1 << 16));
  }
}

// k_pyr_split's per-group table: for every aligned 4-pixel group of a PADDED row (padded columns 4 gi ... 4 gi + 3, pixels
// X0 + k with X0 = 4 gi - 19 at their REFLECT_101 positions) {base = smallest source column, 4 v_perm selectors (left / right tap
// relative to base), 4 coefficient pairs, Xmin and byte selector of the level-0 copy, 0}: the kernel does no reflection, minimum
// or selector arithmetic of its own.  Returns false if some group's sources do not fit the 8-byte window.
static bool group_table(LevelGeom& L, int src_w, const std::vector<int32_t>* xo, const std::vector<int32_t>* xa, std::vector<int32_t>& coef) {
  while (coef.size() % 4) coef.push_back(0);
  L.cg = (int)coef.size();
  auto refl = [](int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
  };
  bool ok = true;
  const int G = (L.w + 2 * SD_EDGE + 3) / 4;
  for (int gi = 0; gi < G; gi++) {
    int Xr[4], sx[4] = {0, 0, 0, 0};
    uint32_t ab[4] = {0, 0, 0, 0}, selP[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++) Xr[k] = refl(4 * gi - SD_EDGE + k, L.w);
    int base = 0;
    if (xo) {
      for (int k = 0; k < 4; k++) {
        sx[k] = (*xo)[Xr[k]];
        ab[k] = (uint32_t)(*xa)[Xr[k]];
      }
      base = std::min(std::min(sx[0], sx[1]), std::min(sx[2], sx[3]));
      for (int k = 0; k < 4; k++) {
        const int o = sx[k] - base, o1 = std::min(sx[k] + 1, src_w - 1) - base;
        if (o < 0 || o > 7 || o1 < 0 || o1 > 7) ok = false;
        selP[k] = (uint32_t)o | 0x0c00u | ((uint32_t)o1 << 16) | 0x0c000000u;
      }
    }
    const int Xmin = std::min(std::min(Xr[0], Xr[1]), std::min(Xr[2], Xr[3]));
    uint32_t sel0 = 0;
    for (int k = 0; k < 4; k++) {
      if (Xr[k] - Xmin > 3) ok = false;
      sel0 |= (uint32_t)((Xr[k] - Xmin) & 3) << (8 * k);
    }
    const int32_t e[12] = {base, (int32_t)selP[0], (int32_t)selP[1], (int32_t)selP[2], (int32_t)selP[3], (int32_t)ab[0], (int32_t)ab[1],
                           (int32_t)ab[2], (int32_t)ab[3], Xmin, (int32_t)sel0, 0};
    coef.insert(coef.end(), e, e + 12);
  }
  return ok;
}

bool plan_geometry(int nfeatures, int nlevels, int thFAST, int w, int h, HostPlan& hp, const char** why) {
  (void)nfeatures;
  if (nlevels < 1 || nlevels > SD_MAX_LEVELS) { *why = "nlevels out of range"; return false; }
  if (w < 1 || h < 1 || w > SD_MAX_DIM || h > SD_MAX_DIM) { *why = "image size out of range (1..4095)"; return false; }
  OrbPlan& P = hp.plan;
  memset(&P, 0, sizeof(P));
  P.nlevels = nlevels;
  P.thFAST = std::min(std::max(thFAST, 0), 255);
  P.w0 = w;
  P.h0 = h;
  hp.cells.clear();
  hp.blur_tiles.clear();
  hp.coef.clear();
  hp.max_cells_per_level = 0;
  hp.max_cell_pixels = 0;

  size_t off = 0;
  uint32_t cand = 0;
  int sel = 0;
  const float imageRatio = (float)w / h;   // src/ORBextractor.cc:469 (level 0 cols/rows)
  size_t lds_max = 0;
  {
    // option "extract.fast_merge_from" (A/B): first level of the merged FAST launch; >= nlevels = one launch per level.  Measured (1024 VGA
    // frames, 8 levels): full step 163.1 k frames/s with one launch per level, 167.0 k from level 4, 169.8 k from level 3,
    // 165.6 k from level 2 (ORB alone is indifferent up to 3 and loses from 2 on: the merged launch waits for the whole pyramid).
    // r3, after k_fast_cells itself got 12 % shorter (flat phase A, scalar loop control): from level 5 203.3 k, 6: 202.7 k,
    // 4: 197.7-203.9 k (two states), 3: 194.7 k, none: 201.5 k (alternating runs) -- the default moved from 3 to 5; and, after the
    // NMS phase lost its nine dependent LDS round trips (alone another 6 % shorter), to 6: full step 205.4 k vs 205.0 k, TrackWithMotionModel
    // 208.4 k vs 206.6 k, ORB-only 262.4 k vs 260.1 k, 1280x720 93.0 k vs 89.0 k
    hp.fast_merge_from = std::max(1, opt(OPT_FAST_MERGE_FROM));
  }
  for (int l = 0; l < SD_MAX_LEVELS; l++) hp.fast_lds_level[l] = 0;
  double sum_px = 0, px0 = 0, px_last = 0;

  for (int l = 0; l < nlevels; l++) {
    LevelGeom& L = P.lv[l];
    float scale = hp.inv_sf[l];
    L.w = cv_round((float)w * scale);   // src/ORBextractor.cc:683
    L.h = cv_round((float)h * scale);
    if (L.w < 1 || L.h < 1) { *why = "pyramid level collapses to zero size"; return false; }
    L.pstride = (int)up((size_t)L.w + 2 * SD_EDGE, 64);
    L.prows = L.h + 2 * SD_EDGE;
    L.off = (uint32_t)off;
    off += up((size_t)L.pstride * L.prows, 256);
    L.quota = hp.quota[l];
    L.scale = hp.sf[l];
    L.kpsize = (float)(int)(31 * hp.sf[l]);   // const int scaledPatchSize = PATCH_SIZE*mvScaleFactor[level]
    L.sel_off = sel;
    sel += L.quota;
    sum_px += (double)L.w * L.h;
    if (l == 0) px0 = (double)L.w * L.h;
    px_last = (double)L.w * L.h;

    // resize tables (level l from level l-1)
    L.area2x2 = 0;
    L.fast_resize = 0;
    L.cx = L.cy = L.cg = L.cr = 0;
    L.scale_x = L.scale_y = 1.0;
    if (l > 0) {
      const LevelGeom& S = P.lv[l - 1];
      double sx = 1. / ((double)L.w / S.w), sy = 1. / ((double)L.h / S.h);
      L.scale_x = sx;
      L.scale_y = sy;
      int isx = cv_round(sx), isy = cv_round(sy);
      bool fast = std::fabs(sx - isx) < DBL_EPSILON && std::fabs(sy - isy) < DBL_EPSILON;
      if (fast && isx == 2 && isy == 2) {
        L.area2x2 = 1;
      } else {
        std::vector<int32_t> xo, xa, yo, yb;
        resize_tables(S.w, L.w, xo, xa, true);
        resize_tables(S.h, L.h, yo, yb, false);
        L.cx = (int)hp.coef.size();
        hp.coef.insert(hp.coef.end(), xo.begin(), xo.end());
        hp.coef.insert(hp.coef.end(), xa.begin(), xa.end());
        L.cy = (int)hp.coef.size();
        hp.coef.insert(hp.coef.end(), yo.begin(), yo.end());
        hp.coef.insert(hp.coef.end(), yb.begin(), yb.end());
        // k_pyr_resize gathers the sources of 4 adjacent outputs out of 8 consecutive source bytes
        bool ok = L.w >= 8;
        // (every run of 4 consecutive outputs: the border groups of a padded row are such runs at reflected positions, in any order)
        for (int x0 = 0; ok && x0 + 3 < L.w; x0++)
          if (std::min(xo[x0 + 3] + 1, S.w - 1) - xo[x0] > 7 || xo[x0 + 3] < xo[x0]) ok = false;
        for (int x = 0; ok && x < L.w; x++)
          if ((xa[x] & 0xffff) > 2048 || ((uint32_t)xa[x] >> 16) > 2048) ok = false;
        for (int y = 0; ok && y < L.h; y++)
          if (yo[y] < 0 || (yb[y] & 0xffff) > 2048 || ((uint32_t)yb[y] >> 16) > 2048) ok = false;
        L.fast_resize = ok ? 1 : 0;
        if (ok) ok = group_table(L, S.w, &xo, &xa, hp.coef);
        if (ok) {   // per-row table: where the two source rows start inside the padded source level (column 0 of the interior)
          while (hp.coef.size() % 4) hp.coef.push_back(0);
          L.cr = (int)hp.coef.size();
          for (int y = 0; y < L.h; y++) {
            const int sy0 = std::min(yo[y], S.h - 1), sy1 = std::min(yo[y] + 1, S.h - 1);
            hp.coef.push_back((sy0 + SD_EDGE) * S.pstride + SD_EDGE);
            hp.coef.push_back((sy1 + SD_EDGE) * S.pstride + SD_EDGE);
            hp.coef.push_back(yb[y]);
            hp.coef.push_back(0);
          }
        }
        L.fast_resize = ok ? 1 : 0;
      }
    } else {
      L.fast_resize = group_table(L, 0, nullptr, nullptr, hp.coef) ? 1 : 0;   // level 0: the copy of the frame uses the column part only
    }

    // grid (src/ORBextractor.cc:472-488)
    const int nDesired = L.quota;
    const int levelCols = (int)sqrtf((float)nDesired / (5 * imageRatio));
    const int levelRows = (int)(imageRatio * levelCols);
    L.cell0 = (int)hp.cells.size();
    L.cand_off = cand;
    L.cols = L.rows = L.ncells = 0;
    L.cellW = L.cellH = L.nfeaturesCell = 0;
    if (levelCols > 0 && levelRows > 0 && nDesired > 0) {
      const int minBX = SD_EDGE, minBY = SD_EDGE, maxBX = L.w - SD_EDGE, maxBY = L.h - SD_EDGE;
      const int W = maxBX - minBX, H = maxBY - minBY;
      const int cellW = (int)ceilf((float)W / levelCols);
      const int cellH = (int)ceilf((float)H / levelRows);
      L.cols = levelCols;
      L.rows = levelRows;
      L.cellW = cellW;
      L.cellH = cellH;
      L.ncells = levelCols * levelRows;
      L.nfeaturesCell = (int)ceilf((float)nDesired / L.ncells);
      for (int i = 0; i < levelRows; i++) {
        const int iniY = minBY + i * cellH - 3;
        int hY = cellH + 6;
        if (i == levelRows - 1) hY = maxBY + 3 - iniY;
        for (int j = 0; j < levelCols; j++) {
          const int iniX = minBX + j * cellW - 3;
          int hX = cellW + 6;
          if (j == levelCols - 1) hX = maxBX + 3 - iniX;
          CellGeom c;
          c.level = l;
          // Cells the reference skips with `continue` (src/ORBextractor.cc:507-511,526-530) keep
          // bNoMore=false / nTotal=0 and enter the quota loop differently from cells whose FAST
          // call simply finds nothing, so the distinction is carried in `evaluated`.  A cell view
          // outside the level image (cv::Mat::rowRange/colRange would throw; degenerate grids
          // only) is treated as skipped, like the oracle does.
          bool skip = (i == levelRows - 1 && hY <= 0) || (j == levelCols - 1 && hX <= 0) || iniX < 0 ||
                      iniY < 0 || hX < 0 || hY < 0 || iniX + hX > L.w || iniY + hY > L.h;
          bool ok = !skip && hX > 6 && hY > 6;   // FAST scans [3, dim-3) of the cell view
          c.evaluated = skip ? 0 : 1;
          c.zx0 = iniX + 3;
          c.zy0 = iniY + 3;
          c.zw = ok ? hX - 6 : 0;
          c.zh = ok ? hY - 6 : 0;
          // k_fast_cells walks a zone as one flat pixel range and recovers the row with a 32-bit multiply-high magic number, which
          // does not exist for a divisor of 1 (frames a few dozen pixels wide with hundreds of features per level): refused, not guessed
          if (c.zw == 1) { *why = "a grid cell's FAST zone is one pixel wide (frame too small for this many features per level)"; return false; }
          c.cap = ok ? (uint32_t)(((c.zw + 1) / 2) * ((c.zh + 1) / 2)) : 0;
          c.cand_off = cand;
          cand += c.cap;
          c.strip_rows = 0;
          if (ok) hp.max_cell_pixels = std::max(hp.max_cell_pixels, c.zw * c.zh);
          if (ok) {
            // strip height: <= 64 queue chunks per wave (bits in a u64), <= 2^16 queue indices,
            // LDS = pixel tile + score map + per-wave u16 queues (worst case one entry per pixel)
            // Budgets (r2, measured): a cell that fits in one strip within 40 KB is done in one strip (no second staging /
            // barrier round: VGA ORB-only 220 k -> 231 k frames/s, full step 168.6 k -> 173.5 k); cells that need strips
            // anyway (1280x720: 311 x 98 pixels) get 24 KB strips (720p 81.2 k vs 79.9 k at 40 KB, 57.7 k at 60 KB).
            // Options "extract.fast_lds_kb" / "extract.fast_lds_whole_kb" override the two (experiments).
            const size_t lds_kb = (size_t)opt(OPT_FAST_LDS_KB);
            const size_t whole_kb = (size_t)opt(OPT_FAST_LDS_WHOLE_KB);
            int S = c.zh;
            // (levels of the merged launch keep the strip budget: a launch asks for the LARGEST need of its cells, and one
            // 34-KB level would take every small cell of that launch from 6 to 4 workgroups per CU -- 720p: 81.3 k -> 77.7 k)
            bool whole_pass = l < hp.fast_merge_from;   // first candidate: the whole cell under the larger budget
            if (!whole_pass) S = std::min(c.zh, std::max(1, 12288 / c.zw));
            for (;;) {
              size_t tp = up((size_t)c.zw + 6 + 3, 4), sp = up((size_t)c.zw + 2, 4);
              size_t rpw = (size_t)(S + 2 + 3) / 4;
              size_t need = tp * (S + 2 + 6) + sp * (S + 2 + 2) + 2 * 4 * rpw * c.zw + 64;
              const size_t budget = whole_pass ? std::max(whole_kb, lds_kb) : lds_kb;
              bool fits = need <= budget * 1024 && rpw * c.zw <= 4096 && (size_t)(S + 2) * c.zw < 65536;
              if (whole_pass && !fits) {
                whole_pass = false;
                S = std::min(c.zh, std::max(1, 12288 / c.zw));
                continue;
              }
              if (fits || S == 1) {
                lds_max = std::max(lds_max, need);
                hp.fast_lds_level[l] = std::max(hp.fast_lds_level[l], need);
                break;
              }
              S = std::max(1, S - std::max(1, S / 8));
            }
            c.strip_rows = S;
          }
          hp.cells.push_back(c);
        }
      }
    }
    L.cand_cap = cand - L.cand_off;
    hp.max_cells_per_level = std::max(hp.max_cells_per_level, L.ncells);

    {   // k_blur: a 16-lane group per (64-column strip, 32-row band), sixteen groups per workgroup, dealt flat over the level
      const int nstrips = (L.w + 63) / 64, npairs = nstrips * ((L.h + 31) / 32);
      for (int p0 = 0; p0 < npairs; p0 += 16)
        hp.blur_tiles.push_back(BlurTile{l, p0, nstrips, (unsigned)(0xFFFFFFFFull / (unsigned)nstrips) + 1u});
    }
  }
  P.ncells = (int)hp.cells.size();
  P.nsel = sel;
  P.cand_per_frame = cand;
  P.pyr_frame_bytes = off;
  hp.fast_lds_bytes = lds_max;
  if (hp.coef.empty()) hp.coef.push_back(0);

  // algorithmic bytes per frame: each stage reads its input once, writes once
  hp.stage_bytes[0] = (sum_px - px_last) + (sum_px - px0) + px0 * 2;  // pyramid (+ level-0 copy in/out)
  hp.stage_bytes[1] = sum_px;                                         // FAST + NMS scan (read)
  hp.stage_bytes[2] = (double)cand * 0 + (double)sel * 4;             // selection (keys; negligible)
  hp.stage_bytes[3] = 2 * sum_px;                                     // blur read + write
  hp.stage_bytes[4] = (double)sel * (749 + 961 + 28 + 32);            // orientation + rBRIEF gathers + outputs
  return true;
}

}  // namespace sd
