// Host-side plan of one ORB extractor geometry (pyramid sizes, grid cells, resize tables,
// HBM layout).  Restates the *parameter* arithmetic of the reference exactly:
//   scale/sigma tables, per-level quotas   reference src/ORBextractor.cc:406-434
//   level sizes                            src/ORBextractor.cc:682-684
//   grid / cell zones                      src/ORBextractor.cc:469-532
// The structs below are uploaded verbatim and read by the kernels in orb.hip.
#pragma once
#include <stdint.h>
#include <vector>

#define SD_MAX_LEVELS 16
#define SD_EDGE 19        // EDGE_THRESHOLD, src/ORBextractor.cc:75
#define SD_MAX_DIM 4095   // x,y packed into 12 bits each in candidate keys

namespace sd {

struct LevelGeom {
  int w, h;              // level image size (cv::Size sz)
  int pstride;           // padded row stride in bytes (multiple of 64)
  int prows;             // h + 2*19
  uint32_t off;          // byte offset of the padded level inside one frame's pyramid block
  int quota;             // mnFeaturesPerLevel[level]
  int cols, rows;        // levelCols, levelRows
  int cellW, cellH;
  int nfeaturesCell;
  int cell0, ncells;     // cells of this level inside the global cell table
  int sel_off;           // first slot of this level in the per-frame selected-key array
  int area2x2;           // resize takes the exact-2x INTER_AREA fast path
  int cx, cy;            // offsets of the x / y resize tables inside the coefficient array: {ofs[dn], a0|a1<<16 [dn]}
  int cr;                // offset of the per-row table of k_pyr_split (4 dwords per interior row: byte offsets of the two source rows, b0|b1<<16, 0)
  int cg;                // offset of the per-group table of k_pyr_split (12 dwords per 4-pixel group of a padded row, 16-byte aligned)
  int fast_resize;       // level qualifies for k_pyr_resize (bilinear, 4 outputs read <= 8 source bytes)
  double scale_x, scale_y;   // cv::resize: 1. / ((double)dst / src) from level l-1 to l
  float scale;           // mvScaleFactor[level]
  float kpsize;          // (float)(int)(31 * scale)
  uint32_t cand_off;     // first candidate slot of this level inside one frame's block
  uint32_t cand_cap;     // sum of the level's cell capacities
};

struct CellGeom {
  int level;
  int zx0, zy0, zw, zh;  // FAST detection zone of the cell in level coordinates
  uint32_t cand_off;     // candidate slots of this cell inside one frame's block
  uint32_t cap;          // ceil(zw/2)*ceil(zh/2): 3x3 NMS leaves at most one per 2x2 block
  int strip_rows;        // zone rows per LDS strip
  int evaluated;         // 0: the reference `continue`s over this cell (never calls FAST)
};

// k_blur work item: 16 consecutive (64-column strip, 32-row band) pairs of a level, pair p = band * nstrips + strip
struct BlurTile { int level, p0, nstrips; unsigned magic; };   // magic: p / nstrips == umulhi(p, magic)

struct OrbPlan {
  int nlevels, ncells, nsel;   // nsel = sum of quotas (<= nfeatures)
  int thFAST;
  int w0, h0;
  uint32_t cand_per_frame;     // candidate slots per frame
  uint64_t pyr_frame_bytes;    // bytes of one frame's padded pyramid block
  LevelGeom lv[SD_MAX_LEVELS];
};

struct HostPlan {
  OrbPlan plan;
  std::vector<CellGeom> cells;
  std::vector<BlurTile> blur_tiles;
  std::vector<int32_t> coef;   // resize tables
  std::vector<float> sf, inv_sf, sigma2, inv_sigma2;
  std::vector<int> quota;
  size_t fast_lds_bytes;       // dynamic LDS of k_fast_cells (max over cells)
  int fast_merge_from;         // levels >= this share one k_fast_cells launch (orb.hip); they keep the strip budget
  size_t fast_lds_level[SD_MAX_LEVELS];   // ... per level: the small levels' cells need far less than the budget, and a launch
                                          // that asks for less LDS keeps more workgroups per CU
  int max_cells_per_level;
  int max_cell_pixels;         // largest FAST detection zone (pixels): sizes the selection kernels' LDS cell buffers
  double stage_bytes[8];       // algorithmic bytes per frame per stage
};

// Tables that depend only on the ctor arguments.
void plan_tables(int nfeatures, float scaleFactor, int nlevels, HostPlan& hp);
// Geometry for a w x h frame. Returns false (with message) if unsupported.
bool plan_geometry(int nfeatures, int nlevels, int thFAST, int w, int h, HostPlan& hp, const char** why);

}  // namespace sd
