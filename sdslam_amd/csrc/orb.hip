// ORB extraction on MI355X (gfx950): hand-written HIP kernels + the sd_orb_* C ABI.
//
// Replaces SD_SLAM::ORBextractor (reference src/ORBextractor.cc).  Batched-frames-first:
// every kernel takes the frame index as its outermost grid dimension, so one launch covers
// all frames of a batch (and, for blur / selection / descriptors, all pyramid levels).
//
//   k_pyr_split      ComputePyramid: resize + copyMakeBorder   src/ORBextractor.cc:680-700 (one launch per level: aligned 4-pixel
//                    groups of every padded row, table-driven bilinear at REFLECT_101 column indices; border rows as second stores)
//   k_pyr_level      same, single generic pass     (exact-2x INTER_AREA levels, byte-unaligned inputs, tiny levels)
//   k_fast_cells     cv::FAST per grid cell        src/ORBextractor.cc:501-552 (FAST-9/16, score, cell-local 3x3 NMS)
//   k_select_*       quota loop + retainBest       src/ORBextractor.cc:554-605 (wave-parallel libstdc++ introselect replay)
//   k_blur           GaussianBlur 7x7 s=2          src/ORBextractor.cc:659-660 (8-bit fixed point, separable)
//   k_orient_desc    IC_Angle + steered rBRIEF     src/ORBextractor.cc:78-143, 608-618, 669-674
//   k_undistort      Frame::UndistortKeyPoints     src/Frame.cc:335-366
//
// Streams per handle: main (resize chain, selection, descriptors), fast (FAST of a level as soon as the level is
// complete), aux (blur); see launch_pipeline / pipeline_body and DESIGN.md section 5.
// HBM layout (per frame): padded pyramid block (all levels, 19-px border, 64-B aligned rows),
// blurred block (same geometry), candidate keys (u32: response<<24 | y<<12 | x, per cell, raster
// order), selected keys per level, output keypoints (28 B) + descriptors (32 B).
// Integer / byte work limited by VALU issue and load latency: no MFMA anywhere
// (largest dense object is 7 taps).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "introselect.h"
#include "introselect_wave.h"
#include "orb_plan.h"
#include "sd_common.h"
#include "sd_sincosf.h"

namespace sd {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
// Frame / block of this workgroup for a (blocks, frames) grid such that the workgroups of ONE frame run on ONE XCD: the dispatcher
// deals workgroups to the 8 XCDs round-robin in linear order (MI355X_MICROARCH.md, workgroup dispatch), every XCD has its own 4-MB L2,
// and the workgroups of a frame re-read each other's rows (resize: the two source rows of adjacent output rows; FAST / blur: tile
// halos; 128-B lines shared by neighbouring tiles).  Bijective for any grid size; a speed choice only.
#ifndef SD_XCD_REMAP
#define SD_XCD_REMAP 1
#endif
__device__ __forceinline__ void xcd_frame_block(unsigned& frame, unsigned& blk) {
#if SD_XCD_REMAP
  const unsigned nx = gridDim.x, total = nx * gridDim.y;
  const unsigned lin = blockIdx.y * nx + blockIdx.x;
  const unsigned q = total >> 3, r = total & 7u, xcd = lin & 7u;
  const unsigned logical = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (lin >> 3);
  frame = logical / nx;
  blk = logical - frame * nx;
#else
  frame = blockIdx.y;
  blk = blockIdx.x;
#endif
}

__device__ __forceinline__ int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
  unsigned lane = __lane_id();
  return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// ------------------------------------------------------------------------------------------
// k_pyr_level: one pyramid level, written over its whole padded domain in a single pass.
// level 0 : copyMakeBorder(image, REFLECT_101)
// level l : resize(level l-1 -> l, INTER_LINEAR) then copyMakeBorder(REFLECT_101|ISOLATED);
//           a border pixel equals the resize result of its reflected interior pixel.
// Each thread produces 4 horizontally adjacent bytes (one u32 store, coalesced).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);   // global memory runs in unaligned-access mode: one dword load
  return v;
}

// 16-byte window starting at the 4-byte-aligned address at or below p, as two 64-bit halves;
// *sh = p's offset inside the window (0..3).  Aligned dword loads only: byte-unaligned vector
// loads are legal on gfx950 but ran the bilinear level kernels at ~270 GB/s.
__device__ __forceinline__ void ld_window16(const uint8_t* p, unsigned long long* lo, unsigned long long* hi, int* sh) {
  const uintptr_t a = (uintptr_t)p;
  const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
  *sh = (int)(a & 3);
  const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
  *lo = w0 | ((unsigned long long)w1 << 32);
  *hi = w2 | ((unsigned long long)w3 << 32);
}
// cv::resize INTER_LINEAR coefficients of destination index d (OpenCV 3.2):
// fx = (float)((d + 0.5) * scale - 0.5); s = floor(fx); fx -= s; alpha = saturate_cast<short>(w * 2048).
// Recomputed per pixel (a handful of IEEE ops, bit-identical to the host tables) so that the
// pixel loads do not wait behind a dependent table load.
__device__ __forceinline__ void resize_coef(int d, double scale, int sn, bool clamp, int* s_out, int* a0, int* a1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (clamp) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= sn - 1) { f = 0.f; s = sn - 1; }
  }
  *s_out = s;
  *a0 = (int)rintf((1.f - f) * 2048.f);
  *a1 = (int)rintf(f * 2048.f);
}

__device__ __forceinline__ int win_byte(unsigned long long lo, unsigned long long hi, int b) {   // b in 0..15
  return b < 8 ? (int)((lo >> (8 * b)) & 0xff) : (int)((hi >> (8 * (b - 8))) & 0xff);
}

#define PYR_ROWS 1   // rows per thread (multi-row unrolling bought nothing and tripped a codegen problem in the byte packing)

__device__ __forceinline__ uint32_t pyr_px4(const LevelGeom& L, const LevelGeom& S, size_t pyr_frame_bytes, int level, int frame, int px,
                                            int py, const uint8_t* __restrict__ src0, int src_stride, size_t src_frame_stride,
                                            const uint8_t* __restrict__ pyr) {
  const int Y = reflect101(py - SD_EDGE, L.h);
  // fast path: the 4 outputs are interior pixels X0..X0+3 (no reflection, consecutive sources)
  const int X0 = px - SD_EDGE;
  const bool interior = X0 >= 0 && X0 + 3 < L.w;
  uint32_t packed = 0;
  if (level == 0) {
    const uint8_t* s = src0 + (size_t)frame * src_frame_stride + (size_t)Y * src_stride;
    if (interior && X0 >= 4 && X0 + 7 < L.w) {   // aligned window stays inside this source row
      const uintptr_t a = (uintptr_t)(s + X0);
      const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
      const int sh = (int)(a & 3) * 8;
      const unsigned long long w = q[0] | ((unsigned long long)q[1] << 32);
      packed = (uint32_t)(w >> sh);
    } else if (interior) {
      packed = ld_u32_unaligned(s + X0);
    } else {
      uint32_t v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = s[reflect101(min(px + k, L.w + 2 * SD_EDGE - 1) - SD_EDGE, L.w)];   // 4 loads in flight
#pragma unroll
      for (int k = 0; k < 4; k++) packed |= (px + k < L.w + 2 * SD_EDGE ? v[k] : 0u) << (8 * k);
    }
  } else {
    const uint8_t* sb = pyr + (size_t)frame * pyr_frame_bytes + S.off + (size_t)SD_EDGE * S.pstride + SD_EDGE;
    if (L.area2x2) {
      const uint8_t* r0 = sb + (size_t)(2 * Y) * S.pstride;
      const uint8_t* r1 = r0 + S.pstride;
      if (interior) {
        const uint32_t a0 = ld_u32_unaligned(r0 + 2 * X0), a1 = ld_u32_unaligned(r0 + 2 * X0 + 4);
        const uint32_t b0 = ld_u32_unaligned(r1 + 2 * X0), b1 = ld_u32_unaligned(r1 + 2 * X0 + 4);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t ta = k < 2 ? (a0 >> (16 * k)) : (a1 >> (16 * (k - 2)));
          const uint32_t tb = k < 2 ? (b0 >> (16 * k)) : (b1 >> (16 * (k - 2)));
          const uint32_t v = ((ta & 0xff) + ((ta >> 8) & 0xff) + (tb & 0xff) + ((tb >> 8) & 0xff) + 2) >> 2;
          packed |= v << (8 * k);
        }
      } else {
        uint32_t t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {   // loads of the four pixels first (clamped column, result dropped beyond the padded row)
          const int X = reflect101(min(px + k, L.w + 2 * SD_EDGE - 1) - SD_EDGE, L.w);
          t[k] = r0[2 * X] + r0[2 * X + 1] + r1[2 * X] + r1[2 * X + 1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) packed |= (px + k < L.w + 2 * SD_EDGE ? (t[k] + 2) >> 2 : 0u) << (8 * k);
      }
    } else {
      int sy0, b0, b1;
      resize_coef(Y, L.scale_y, S.h, false, &sy0, &b0, &b1);
      int sy1 = sy0 + 1;
      sy0 = sy0 < 0 ? 0 : (sy0 < S.h ? sy0 : S.h - 1);
      sy1 = sy1 < 0 ? 0 : (sy1 < S.h ? sy1 : S.h - 1);
      const uint8_t* r0 = sb + (size_t)__mul24(sy0, S.pstride);
      const uint8_t* r1 = sb + (size_t)__mul24(sy1, S.pstride);
      bool done = false;
      if (interior) {
        // the sources of 4 consecutive outputs span <= 12 bytes for scale factors up to 2:
        // fetch each source row as three dwords and pick bytes out of the registers
        int sx[4], wa0[4], wa1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) resize_coef(X0 + k, L.scale_x, S.w, true, &sx[k], &wa0[k], &wa1[k]);
        const int base = sx[0];
        if (sx[3] + 1 - base <= 11 && base + 15 < S.w + SD_EDGE) {   // 16-byte aligned window stays inside the padded row
          unsigned long long lo0, hi0, lo1, hi1;
          int sh0, sh1;
          ld_window16(r0 + base, &lo0, &hi0, &sh0);
          ld_window16(r1 + base, &lo1, &hi1, &sh1);   // sh1 == sh0 (row pitch is a multiple of 64)
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int o = sx[k] - base;            // 0..10
            const int o1 = (sx[k] + 1 < S.w ? sx[k] + 1 : S.w - 1) - base;
            const int p00 = win_byte(lo0, hi0, sh0 + o), p01 = win_byte(lo0, hi0, sh0 + o1);
            const int p10 = win_byte(lo1, hi1, sh1 + o), p11 = win_byte(lo1, hi1, sh1 + o1);
            const int a0 = wa0[k], a1 = wa1[k];
            // all factors fit 24 bits: v_mul_i32_i24 / v_mad_i32_i24 issue at full rate, v_mul_lo_u32 at a quarter
            const int h0 = __mul24(p00, a0) + __mul24(p01, a1);
            const int h1 = __mul24(p10, a0) + __mul24(p11, a1);
            const int ov = ((__mul24(b0, h0 >> 4) >> 16) + (__mul24(b1, h1 >> 4) >> 16) + 2) >> 2;
            packed |= (uint32_t)(ov < 0 ? 0 : (ov > 255 ? 255 : ov)) << (8 * k);
          }
          done = true;
        }
      }
      if (!done) {
        // border / edge pixels: the 4 x 4 source bytes are fetched first (columns beyond the padded row are clamped, their
        // result is dropped), so a thread has 16 loads in flight instead of four dependent rounds of four
        int p00[4], p01[4], p10[4], p11[4], a0[4], a1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int x = min(px + k, L.w + 2 * SD_EDGE - 1);
          const int X = reflect101(x - SD_EDGE, L.w);
          int sx;
          resize_coef(X, L.scale_x, S.w, true, &sx, &a0[k], &a1[k]);
          const int sx1 = sx + 1 < S.w ? sx + 1 : S.w - 1;
          p00[k] = r0[sx]; p01[k] = r0[sx1]; p10[k] = r1[sx]; p11[k] = r1[sx1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int h0 = __mul24(p00[k], a0[k]) + __mul24(p01[k], a1[k]);
          const int h1 = __mul24(p10[k], a0[k]) + __mul24(p11[k], a1[k]);
          const int o = ((__mul24(b0, h0 >> 4) >> 16) + (__mul24(b1, h1 >> 4) >> 16) + 2) >> 2;
          const uint32_t v = px + k < L.w + 2 * SD_EDGE ? (uint32_t)(o < 0 ? 0 : (o > 255 ? 255 : o)) : 0u;
          packed |= v << (8 * k);
        }
      }
    }
  }
  return packed;
}

// L / S (this level, source level) travel by value in the kernel arguments: one less dependent
// load before the first pixel fetch.
__global__ __launch_bounds__(256) void k_pyr_level(const LevelGeom L, const LevelGeom S, size_t pyr_frame_bytes, int level,
                                                   const uint8_t* __restrict__ src0, int src_stride,
                                                   size_t src_frame_stride, uint8_t* __restrict__ pyr) {
  const int frame = blockIdx.z;
  const int px = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int py0 = (blockIdx.y * 4 + threadIdx.y) * PYR_ROWS;
  if (px >= L.pstride || py0 >= L.prows) return;
  uint8_t* dstbase = pyr + (size_t)frame * pyr_frame_bytes + L.off;
  uint32_t out[PYR_ROWS];
#pragma unroll
  for (int r = 0; r < PYR_ROWS; r++) {
    const int py = min(py0 + r, L.prows - 1);   // clamped duplicates are computed but not stored
    out[r] = pyr_px4(L, S, pyr_frame_bytes, level, frame, px, py, src0, src_stride, src_frame_stride, pyr);
  }
#pragma unroll
  for (int r = 0; r < PYR_ROWS; r++)
    if (py0 + r < L.prows) *(uint32_t*)(dstbase + (size_t)(py0 + r) * L.pstride + px) = out[r];
}

// ------------------------------------------------------------------------------------------
// Split pyramid construction (the default path).  k_pyr_level above evaluates every padded pixel
// through one code path, so the waves at both ends of a row run the slow reflected-border branch
// as well as the interior one.  Here every thread owns an aligned 4-pixel group (padded columns 4 gi ... 4 gi + 3) of three rows:
// coefficient tables (the host's cv::resize tables) read at the REFLECT_101 column index of each pixel, one aligned 12-byte window
// per source row (the four sources of a group lie within 8 bytes whatever their order), v_perm_b32 gathers the 2 x 4 source bytes.
// (r1-r3 had a second role for the border columns and the interior pixels outside the aligned interior groups, on the generic
// per-byte path pyr_px4: a fifth of every launch's workgroups.)  Level 0 is the frame copied into the padded layout by the same kernel.
// ------------------------------------------------------------------------------------------
#ifndef PYR_RPT
#define PYR_RPT 4
#endif
// rows per thread: rows Y, Y + ceil(h / 4), ... share the group's column data and give independent load chains (r1: 3 rows 130.8 k,
// 2 rows 130.4 k frames/s, 4 rows no better than 2; r3, with the per-group / per-row tables as the fixed cost of a thread: 4 rows and
// the descriptor patch on 8-byte loads 226.2 k vs 222.3 k with 3 rows; 4 / 5 / 6 rows 229.4 / 226.9 / 225.4 k, alternating runs)
// **r2**: the rows are PADDED rows (0 .. h + 37): a top / bottom border row is the resize of its REFLECT_101 interior row, computed
// here like any other row instead of being copied by a third kernel after the first two (k_pyr_rows, gone): ONE launch per level
// (k_pyr_split), 8 pyramid launches per step instead of 24.  r3: where the level is tall enough for single reflections the border
// rows are second stores of their source rows (`mirror`).
__device__ __forceinline__ void pyr_resize_body(const LevelGeom& L, const LevelGeom& S, size_t pyr_frame_bytes, int level,
                                                const int32_t* __restrict__ coef, const uint8_t* __restrict__ src0, int src_stride,
                                                size_t src_frame_stride, uint8_t* __restrict__ pyr, int G, unsigned magicG, int Hh,
                                                int frame, unsigned e, int mirror) {
  const unsigned Y0 = __umulhi(e, magicG);        // e / G
  if (Y0 >= (unsigned)Hh) return;
  // r3: G counts the 4-pixel groups of the whole PADDED row (padded columns 4 gi ... 4 gi + 3, gi = 0 ... G - 1): the border columns and
  // the few interior pixels outside the aligned interior groups go through the same code at their REFLECT_101 column positions
  // (their four sources still lie within one 8-byte window, in any order) instead of a second role on the generic per-byte path,
  // which was a fifth of every launch's workgroups.  Everything that depends on the group alone -- window base, v_perm selectors,
  // coefficient pairs, the level-0 copy's source column and byte order -- comes from a host-built table (orb_plan.cpp group_table:
  // three aligned 16-byte loads), so the kernel does no reflection, minimum or selector arithmetic.
  const int gi = (int)(e - Y0 * (unsigned)G);
  const int X0 = 4 * gi - SD_EDGE;   // first pixel of the group; negative / beyond w - 1: border
  const int4* gt = (const int4*)(coef + L.cg) + 3 * gi;
  // wave-uniform base pointers + 32-bit per-lane offsets: loads and stores use the SGPR-base addressing form
  uint8_t* dstb = pyr + (size_t)frame * pyr_frame_bytes + L.off;
  const uint32_t dst_x = (uint32_t)(X0 + SD_EDGE);
  uint32_t packed[PYR_RPT];
  bool live[PYR_RPT];
  int Yr[PYR_RPT], PY[PYR_RPT];   // interior (source-side) row, padded (destination) row
#pragma unroll
  for (int r = 0; r < PYR_RPT; r++) {
    if (mirror) {   // interior rows only: the border rows are second stores of their REFLECT_101 source rows (below)
      Yr[r] = (int)Y0 + r * Hh;
      live[r] = Yr[r] < L.h;
      if (!live[r]) Yr[r] = (int)Y0;   // duplicate work, not stored
      PY[r] = Yr[r] + SD_EDGE;
    } else {
      PY[r] = (int)Y0 + r * Hh;
      live[r] = PY[r] < L.prows;
      if (!live[r]) PY[r] = (int)Y0;   // duplicate work, not stored
      Yr[r] = reflect101(PY[r] - SD_EDGE, L.h);
    }
  }
  if (level == 0) {
#pragma unroll
    for (int r = 0; r < PYR_RPT; r++) {
      // k_pyr_resize runs for level 0 only when base pointer and strides are 4-byte aligned (pipeline_body)
      const uint8_t* s = src0 + (size_t)frame * src_frame_stride;
      // (computed, not read from the group table: the copy is latency-bound and the table would be one more dependent round trip)
      int Xr[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (L.w >= 64) {   // single reflection, branch-free (columns beyond the padded width, whose result is dropped, are clamped)
          int x = X0 + k;
          x = x < 0 ? -x : x;
          x = x >= L.w ? 2 * L.w - 2 - x : x;
          Xr[k] = max(x, 0);
        } else {
          Xr[k] = reflect101(X0 + k, L.w);
        }
      }
      const int Xmin = min(min(Xr[0], Xr[1]), min(Xr[2], Xr[3]));   // the four sources are Xmin ... Xmin + 3 in some order
      const uint32_t sel = (uint32_t)(Xr[0] - Xmin) | (uint32_t)(Xr[1] - Xmin) << 8 | (uint32_t)(Xr[2] - Xmin) << 16 | (uint32_t)(Xr[3] - Xmin) << 24;
      const uint32_t o = (uint32_t)(__mul24(Yr[r], src_stride) + Xmin);
      const uint32_t* q = (const uint32_t*)(s + (o & ~3u));
      // (the second dword is not fetched where it would start beyond the row's last pixel: nothing in it is selected, and the
      // last row of the last frame has nothing behind it)
      const uint32_t last = (uint32_t)(__mul24(Yr[r], src_stride) + L.w - 1);
      const uint32_t q1 = q[((o & 3u) != 0u && (o & ~3u) + 4u <= last) ? 1 : 0];
      packed[r] = __builtin_amdgcn_perm(0u, __builtin_amdgcn_alignbyte(q1, q[0], o & 3u), sel);
    }
  } else {
    const int4* rt = (const int4*)(coef + L.cr);   // per interior row: byte offsets of its two source rows, vertical coefficient pair
    const int4 t0 = gt[0], t1 = gt[1], t2 = gt[2];
    const int base = t0.x;
    // v_perm selector per output pixel: {left source byte, 0, right source byte, 0} of the row's 8-byte window = the two
    // taps as a u16 pair, which v_dot2_u32_u16 multiplies with the packed coefficient pair ab[k] in one instruction
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint32_t selP[4] = {(uint32_t)t0.y, (uint32_t)t0.z, (uint32_t)t0.w, (uint32_t)t1.x};
    const uint32_t ab[4] = {(uint32_t)t1.y, (uint32_t)t1.z, (uint32_t)t1.w, (uint32_t)t2.x};
    const uint8_t* sb = pyr + (size_t)frame * pyr_frame_bytes + S.off;   // 64-byte aligned (level offsets and row pitches are)
    uint32_t u[PYR_RPT][3], v[PYR_RPT][3], bbv[PYR_RPT];
    unsigned shv[PYR_RPT];
#pragma unroll
    for (int r = 0; r < PYR_RPT; r++) {   // all loads of both rows first
      const int4 tr = rt[Yr[r]];
      bbv[r] = (uint32_t)tr.z;
      const uint32_t a0 = (uint32_t)(tr.x + base), a1 = (uint32_t)(tr.y + base);
      shv[r] = a0 & 3u;   // same for both source rows: the row pitch is a multiple of 64
      const uint32_t* q0 = (const uint32_t*)(sb + (a0 & ~3u));
      const uint32_t* q1 = (const uint32_t*)(sb + (a1 & ~3u));
#pragma unroll
      for (int k = 0; k < 3; k++) {
        u[r][k] = q0[k];
        v[r][k] = q1[k];
      }
    }
#pragma unroll
    for (int r = 0; r < PYR_RPT; r++) {
      const int b0 = (int)(bbv[r] & 0xffff), b1 = (int)(bbv[r] >> 16);
      const unsigned sh = shv[r];
      // 8 source bytes starting at sx[0], per row
      const uint32_t r0lo = __builtin_amdgcn_alignbyte(u[r][1], u[r][0], sh), r0hi = __builtin_amdgcn_alignbyte(u[r][2], u[r][1], sh);
      const uint32_t r1lo = __builtin_amdgcn_alignbyte(v[r][1], v[r][0], sh), r1hi = __builtin_amdgcn_alignbyte(v[r][2], v[r][1], sh);
      uint32_t pk = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const us2 w = __builtin_bit_cast(us2, ab[k]);
        const int h0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(r0hi, r0lo, selP[k])), w, 0u, false);
        const int h1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(r1hi, r1lo, selP[k])), w, 0u, false);
        // (b * (h >> 4)) >> 16 == mulhi(b << 12, h & ~15): one 32-bit multiply per source row (b <= 2048, h < 2^20)
        // no saturation needed: a0 + a1 = b0 + b1 = 2048, so h <= 255 * 2048 and the sum below is at most (2048 * 32640 >> 16) + 2 = 1022
        const uint32_t ov = (__umulhi((uint32_t)b0 << 12, (uint32_t)h0 & ~15u) + __umulhi((uint32_t)b1 << 12, (uint32_t)h1 & ~15u) + 2u) >> 2;
        pk |= ov << (8 * k);
      }
      packed[r] = pk;
    }
  }
  // r3: the top / bottom REFLECT_101 border rows are COPIES of interior rows of the same level (copyMakeBorder after resize,
  // src/ORBextractor.cc:693-696): padded row 19 - Y holds row Y (1 <= Y <= 19), padded row 2 h + 17 - Y holds row Y
  // (h - 20 <= Y <= h - 2).  With `mirror` a thread's rows are interior rows and the thread that has just computed a pixel group
  // of such a row stores it a second time, instead of 38 more rows of full resize work per level (9 % of level 1, 22 % of level 7).
  // (The same for the border COLUMNS -- byte-reversed second stores by the threads at both ends of a row -- was built and lost:
  // nearly every wave holds a row end, and its 30-40 extra partial-lane store instructions cost more than the edge workgroups.)
  // columns beyond the padded width (the last group of a row when w + 38 is not a multiple of 4) stay zero
  const int nvalid = L.w + 2 * SD_EDGE - (int)dst_x;
  const uint32_t keep = nvalid >= 4 ? 0xffffffffu : (1u << (8 * nvalid)) - 1u;
#pragma unroll
  for (int r = 0; r < PYR_RPT; r++) {
    if (!live[r]) continue;
    packed[r] &= keep;
    *(uint32_t*)(dstb + (dst_x + (uint32_t)__mul24(PY[r], L.pstride))) = packed[r];
    if (mirror) {
      const int Yi = Yr[r];
      const int prow2 = Yi >= 1 && Yi <= SD_EDGE ? SD_EDGE - Yi : (Yi >= L.h - 20 && Yi <= L.h - 2 ? 2 * L.h + 17 - Yi : -1);
      if (prow2 >= 0) *(uint32_t*)(dstb + (dst_x + (uint32_t)__mul24(prow2, L.pstride))) = packed[r];
    }
  }
}

// one launch per level
__global__ __launch_bounds__(256) void k_pyr_split(const LevelGeom L, const LevelGeom S, size_t pyr_frame_bytes, int level,
                                                   const int32_t* __restrict__ coef, const uint8_t* __restrict__ src0, int src_stride,
                                                   size_t src_frame_stride, uint8_t* __restrict__ pyr, int G, unsigned magicG, int Hh,
                                                   int mirror) {
  unsigned uframe, bx;
  xcd_frame_block(uframe, bx);
  pyr_resize_body(L, S, pyr_frame_bytes, level, coef, src0, src_stride, src_frame_stride, pyr, G, magicG, Hh, (int)uframe,
                  bx * 256 + threadIdx.x, mirror);
}

// ------------------------------------------------------------------------------------------
// k_fast_cells: one workgroup per (grid cell, frame).  cv::FAST(cellImage, kps, th, true):
// corner test (>= 9 contiguous ring pixels brighter than v+t or darker than v-t), score =
// max over the 16 nine-arcs of min |v - p| minus 1, strict-greater 3x3 NMS in which pixels
// outside the cell's detection zone count as score 0, survivors emitted in
// raster order (the order cv::FAST produces and retainBest's tie-breaking depends on).
// The cell (+3 px ring halo) is staged once through LDS in aligned 4-byte words; scores live in
// a byte map in LDS; ordered emission uses wave ballots over wave-contiguous pixel ranges.
// ------------------------------------------------------------------------------------------
// FAST-9/16 on one pixel whose 7x7 neighbourhood is in LDS: ring differences d[k] = v - p[k]
// (ring offsets (dx,dy) clockwise from (0,3): SURVEY App. A1).
__device__ __forceinline__ void fast_ring(const uint8_t* __restrict__ c, int tp, int d[16]) {
  const int v = c[0];
  d[0] = v - c[3 * tp];
  d[1] = v - c[3 * tp + 1];
  d[2] = v - c[2 * tp + 2];
  d[3] = v - c[tp + 3];
  d[4] = v - c[3];
  d[5] = v - c[-tp + 3];
  d[6] = v - c[-2 * tp + 2];
  d[7] = v - c[-3 * tp + 1];
  d[8] = v - c[-3 * tp];
  d[9] = v - c[-3 * tp - 1];
  d[10] = v - c[-2 * tp - 2];
  d[11] = v - c[-tp - 3];
  d[12] = v - c[-3];
  d[13] = v - c[tp - 3];
  d[14] = v - c[2 * tp - 2];
  d[15] = v - c[3 * tp - 1];
}

// corner test: >= 9 contiguous ring pixels darker than v - th or brighter than v + th
__device__ __forceinline__ bool fast_is_corner(const uint8_t* __restrict__ c, int tp, int th) {
  int d[16];
  fast_ring(c, tp, d);
  unsigned dark = 0, bright = 0;   // dark: p < v - th  <=> d > th ; bright: p > v + th <=> d < -th
#pragma unroll
  for (int k = 0; k < 16; k++) {
    dark |= (unsigned)(d[k] > th) << k;
    bright |= (unsigned)(d[k] < -th) << k;
  }
  unsigned md = dark | (dark << 16), mb = bright | (bright << 16);
  unsigned rd = md & (md >> 1);
  rd &= rd >> 2;
  rd &= rd >> 4;
  rd &= md >> 8;
  unsigned rb = mb & (mb >> 1);
  rb &= rb >> 2;
  rb &= rb >> 4;
  rb &= mb >> 8;
  return ((rd | rb) & 0xffffu) != 0;
}

// cornerScore<16>: max over the 16 nine-arcs of min(d) / min(-d), minus 1 (>= th for a corner)
__device__ __forceinline__ int fast_corner_score(const uint8_t* __restrict__ c, int tp) {
  int d[16];
  fast_ring(c, tp, d);
  int m2[16], m4[16], M2[16], M4[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    m2[k] = min(d[k], d[(k + 1) & 15]);
    M2[k] = max(d[k], d[(k + 1) & 15]);
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    m4[k] = min(m2[k], m2[(k + 2) & 15]);
    M4[k] = max(M2[k], M2[(k + 2) & 15]);
  }
  int a = -255, b = 255;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    int m9 = min(min(m4[k], m4[(k + 4) & 15]), d[(k + 8) & 15]);
    int M9 = max(max(M4[k], M4[(k + 4) & 15]), d[(k + 8) & 15]);
    a = max(a, m9);
    b = min(b, M9);
  }
  return max(a, -b) - 1;
}

typedef unsigned short fu16;
__device__ __forceinline__ fu16 fmin16(fu16 a, fu16 b) { return a < b ? a : b; }
__device__ __forceinline__ fu16 fmax16(fu16 a, fu16 b) { return a > b ? a : b; }

// Corner test and cornerScore in one: with A = min over the 16 nine-arcs of max(p) and
// B = max over the arcs of min(p), a 9-arc darker than v - th exists iff v - A > th, a brighter one iff
// B - v > th, and cornerScore = max(v - A, B - v) - 1 (the same quantity as fast_corner_score: min / max
// of the differences d = v - p over an arc are v - max(p) / v - min(p)).  Sliding 9-windows on the ring
// are built from 3-windows with v_max3 / v_min3: 2 x (16 + 16 + 8) instructions.  Returns the score;
// the pixel is a corner iff score >= th.
__device__ __forceinline__ int fast_ring_score(const uint8_t* __restrict__ c, int tp) {
  int p[16];
  const int v = c[0];
  p[0] = c[3 * tp]; p[1] = c[3 * tp + 1]; p[2] = c[2 * tp + 2]; p[3] = c[tp + 3];
  p[4] = c[3]; p[5] = c[-tp + 3]; p[6] = c[-2 * tp + 2]; p[7] = c[-3 * tp + 1];
  p[8] = c[-3 * tp]; p[9] = c[-3 * tp - 1]; p[10] = c[-2 * tp - 2]; p[11] = c[-tp - 3];
  p[12] = c[-3]; p[13] = c[tp - 3]; p[14] = c[2 * tp - 2]; p[15] = c[3 * tp - 1];
  int x3[16], n3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    x3[k] = max(max(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
    n3[k] = min(min(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
  }
  // (16-bit min / max for the final reductions would issue at 2.2 instead of 4.1 cycles, but the compiler then loses the
  // v_max3 / v_min3 merges across the two stages, or fuses pairs into v_min3_u16 / v_max3_u16 at 8.1 cycles: 365 cycles per
  // 64 pixels either way -- measured, profiles/r03_valu_microbench*.json)
  int A = 255, B = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int x9 = max(max(x3[k], x3[(k + 3) & 15]), x3[(k + 6) & 15]);
    const int n9 = min(min(n3[k], n3[(k + 3) & 15]), n3[(k + 6) & 15]);
    A = min(A, x9);
    B = max(B, n9);
  }
  return max(v - A, B - v) - 1;
}

#ifdef SD_PNP_PROF   // stage timers of k_fast_cells (tools/prof_select.py --fast): cycles of thread 0 per phase
// one record per workgroup, plain stores (hot-address atomics from 150 k workgroups back up the memory pipeline the tile
// loads go through and inflate the very phase being measured); summed on the host
#define FPROF_FRAMES 256
#define FPROF_CELLS 192
__device__ unsigned long long g_fast_prof_wg[(size_t)FPROF_FRAMES * FPROF_CELLS * 8];
// per-workgroup sums in LDS, one global atomic per phase at the end of the kernel: an atomic per phase and strip would sit in
// vmcnt and be waited for by the next tile load (the first version of these timers charged that wait to "stage")
#define FPROF_DECL                                   \
  __shared__ unsigned long long s_fprof[8];          \
  if (threadIdx.x < 8) s_fprof[threadIdx.x] = 0;     \
  long long _pt = clock64()
#define FPROF(i)                                                                 \
  do {                                                                           \
    long long _n = clock64();                                                    \
    if (threadIdx.x == 0) s_fprof[i] += (unsigned long long)(_n - _pt);          \
    _pt = _n;                                                                    \
  } while (0)
#define FPROF_FLUSH                                                                                                   \
  do {                                                                                                                \
    if (threadIdx.x == 0 && frame < FPROF_FRAMES && cell < FPROF_CELLS)                                          \
      for (int _i = 0; _i < 8; _i++) g_fast_prof_wg[((size_t)frame * FPROF_CELLS + cell) * 8 + _i] = s_fprof[_i]; \
  } while (0)
#else
#define FPROF_DECL
#define FPROF(i)
#define FPROF_FLUSH
#endif

// Structure of one strip (a cell is one strip unless it is too large for LDS):
//   A. every wave owns a contiguous band of rows; per 64-px row segment a 4-read compass test
//      (a 9-arc always contains two ADJACENT compass points of the same polarity) rejects most
//      pixels; survivors are appended, in raster order, to the wave's LDS queue (ballot prefix);
//   B. the queue is processed densely (all lanes busy): one pass computes cornerScore from sliding
//      min / max windows on the ring (fast_ring_score); score >= th is the corner test; corners are
//      re-compacted in place (still raster order) and their scores go to the LDS score map;
//   C. after a block barrier, NMS on the queued corners against the LDS score map, then ordered
//      emission (wave bands are contiguous in raster order, so per-wave counts give offsets).
extern "C" __device__ __attribute__((const)) int __ockl_wfred_add_i32(int);

#ifndef FAST_STAGE_DEPTH
#define FAST_STAGE_DEPTH 12   // tile dwords in flight per thread while staging (256 threads x 12 x 4 B = 12 KB per round trip)
#endif

// 8 waves per SIMD: the kernel needed 65 VGPRs, one over the 64-register step; held to 64 it gains a resident wave per SIMD
// and the whole extraction 5 % (169 k -> 177 k frames/s ORB-only)
// Everything the workgroup needs besides its CellGeom arrives as kernel arguments (no dependent cell -> level -> plan loads):
// img0 / frame_stride / src / edge describe the levels' rows -- the padded pyramid (edge = SD_EDGE), or, for level 0, the
// caller's frames themselves (edge = 0; 4-byte aligned base and strides): FAST only touches interior pixels (the zones
// start SD_EDGE - 3 px inside), so level 0 need not wait for the padded copy of the frame.  One launch covers the cells
// cell0 .. cell0 + gridDim.x - 1, of one level or of several consecutive ones (the small levels go together: each of their
// launches was mostly ramp-up and tail).
struct FastSrc {   // where the rows of a level start inside a frame's block, and their pitch (by value: one launch may span levels)
  uint32_t off[SD_MAX_LEVELS];
  int pstride[SD_MAX_LEVELS];
};

__global__ __launch_bounds__(256, 8) void k_fast_cells(const CellGeom* __restrict__ cells, const uint8_t* __restrict__ img0,
                                                    size_t frame_stride, const FastSrc src, int edge, uint32_t* __restrict__ cand,
                                                    uint32_t cand_per_frame, int32_t* __restrict__ cell_count, int ncells_total,
                                                    int cell0, int th) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int wcnt[4];
  unsigned uframe, ucell;
  xcd_frame_block(uframe, ucell);
  const int cell = (int)ucell + cell0;   // launched per level: the cells of a level are contiguous
  const CellGeom C = cells[cell];
  const int frame = (int)uframe;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform: the band limits and loop counters derived from it stay scalar
  if (C.zw <= 0 || C.zh <= 0) {
    if (tid == 0) cell_count[(size_t)frame * ncells_total + cell] = 0;
    return;
  }
  const uint8_t* img = img0 + (size_t)frame * frame_stride + src.off[C.level];
  const int pstride = src.pstride[C.level];
  uint32_t* out = cand + (size_t)frame * cand_per_frame + C.cand_off;
  const int zw = C.zw, zh = C.zh, S = C.strip_rows;
  const int xs = C.zx0 - 3 + edge;   // x of tile column 0 in the source rows (before alignment)
  const int xa = xs & ~3, sh = xs - xa;
  const int TPW = (sh + zw + 6 + 3) >> 2;   // tile pitch in 4-byte words
  const int TP = TPW * 4;
  const int SP = (zw + 2 + 3) & ~3;         // score-map pitch (bytes)
  const int RPW = (S + 2 + 3) >> 2;         // score rows per wave (upper bound)
  const int QCAP = RPW * zw;                // queue entries per wave (worst case: every pixel)
  uint8_t* tile = smem;
  uint8_t* sc = smem + (size_t)TP * (S + 2 + 6);
  uint16_t* queue = (uint16_t*)(sc + (size_t)SP * (S + 2 + 2)) + (size_t)wave * QCAP;
  // the wave's queue as a scalar byte offset into the dynamic LDS (phase A stores through SGPR base + lane offset)
  const uint32_t qbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)((size_t)TP * (S + 2 + 6) + (size_t)SP * (S + 2 + 2) + (size_t)wave * QCAP * 2));
  const unsigned magic = 0xFFFFFFFFu / (unsigned)zw + 1u;   // q / zw == umulhi(q, magic) for q < 2^16, zw < 2^12
  const unsigned magic_tpw = 0xFFFFFFFFu / (unsigned)TPW + 1u;   // tile dword index / TPW (indices < 2^16: the tile is < 64 KB)
  const unsigned long long lt = lanemask_lt();
  int total = 0;
  FPROF_DECL;

  for (int r0 = 0; r0 < zh; r0 += S) {
    const int r1 = min(r0 + S, zh);
    const int sr0 = max(r0 - 1, 0), sr1 = min(r1 + 1, zh);   // zone rows whose scores are needed
    const int nsr = sr1 - sr0;
    const int npr = nsr + 6;
    // ---- stage pixels (aligned words) and clear the score map.  The tile's dwords are numbered row-major and dealt to the
    // 256 threads FAST_STAGE_DEPTH at a time, all loads before the first LDS store: a tile of up to 12 KB is ONE global round
    // trip (the strip's life is mostly this wait: with 8 rows per wave and batch it was three)
    {
      const uint8_t* g = img + (size_t)(C.zy0 + sr0 - 3 + edge) * pstride + xa;
      const int ndw = __mul24(TPW, npr);
      int t0 = tid;
      asm volatile("" : "+v"(t0));   // opaque per strip: otherwise the row / column of every slot is hoisted out of the strip loop and spilled
      for (int i0 = t0; i0 < ndw; i0 += 256 * FAST_STAGE_DEPTH) {
        uint32_t v[FAST_STAGE_DEPTH];
#pragma unroll
        for (int j = 0; j < FAST_STAGE_DEPTH; j++) {
          const int i = i0 + 256 * j;
          const int row = (int)__umulhi((unsigned)i, magic_tpw), wc = i - __mul24(row, TPW);
          v[j] = i < ndw ? *(const uint32_t*)(g + (uint32_t)(__mul24(row, pstride) + wc * 4)) : 0u;
        }
#pragma unroll
        for (int j = 0; j < FAST_STAGE_DEPTH; j++) {
          const int i = i0 + 256 * j;
          if (i < ndw) ((uint32_t*)tile)[i] = v[j];
        }
      }
      FPROF(7);   // kernel / strip start -> this wave's tile words are in LDS
      const int nsc = ((nsr + 2) * SP) >> 2;
      for (int i = tid; i < nsc; i += 256) ((uint32_t*)sc)[i] = 0;
    }
    __syncthreads();
    FPROF(0);
    // ---- A: compass quick test, ordered queue per wave
    const int rpw = (nsr + 3) >> 2;
    const int y_lo = min(wave * rpw, nsr), y_hi = min(y_lo + rpw, nsr);
    int qn = 0;
    // r3: the wave's band of rows is walked as ONE flat range of pixel indices i = y * zw + x, 64 at a time, instead of row by
    // row: zones are 47...121 pixels wide, so row-wise chunks ran at 57-98 % lane use (74 % on average: a third more chunks, and
    // this phase is bound by LDS-instruction issue as much as by the vector ALU).  Costs one multiply-high and one multiply-add per
    // chunk (row = i / zw by magic number, LDS address = i + row * (TP - zw) + const); the queue entry IS i.
    {
      const int i_lo = __mul24(y_lo, zw), i_hi = __mul24(y_hi, zw);
      const int pad = TP - zw;
      const uint8_t* tile0 = tile + 3 * TP + 3 + sh;
#ifdef FAST_SKIP_A
      for (int i0 = i_hi; i0 < i_hi; i0 += 64) {
#else
      for (int i0 = i_lo; i0 < i_hi; i0 += 64) {
#endif
        // straight-line code on 16-bit min / max.  The lanes beyond the band read a clamped position and are masked out
        // of the ballot on the scalar unit; with D = max(min(p0, p8), min(p4, p12)) and Bt = min(max(p0, p8), max(p4, p12))
        // the compass condition ((k0 | k8) & (k4 | k12)) | ((b0 | b8) & (b4 | b12)), k = p < v - th, b = p > v + th, is
        // max(v - D, Bt - v) > th: 9 full-rate 16-bit instructions and ONE compare.
        const int i = min(i0 + lane, i_hi - 1);
        const int y = (int)__umulhi((unsigned)i, magic);
        const uint8_t* c = tile0 + (i + __mul24(y, pad));
        const fu16 v = c[0], p0 = c[3 * TP], p4 = c[3], p8 = c[-3 * TP], p12 = c[-3];
        const fu16 D = fmax16(fmin16(p0, p8), fmin16(p4, p12));
        const fu16 Bt = fmin16(fmax16(p0, p8), fmax16(p4, p12));
        const short da = (short)(v - D), db = (short)(Bt - v);
        bool pass = (da > db ? da : db) > (short)th;
        if (i_hi - i0 < 64) pass = pass && (i0 + lane < i_hi);   // wave-uniform branch: only the band's last chunk pays this compare
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (pass) *(uint16_t*)(smem + (qbase + 2u * (uint32_t)(qn + rank))) = (uint16_t)i;
        qn += __popcll(m);
      }
    }
    FPROF(1);
#ifdef FAST_SKIP_B   // instruction-budget experiments only (tools/fast_budget.sh): results are wrong by design
    qn = 0;
#endif
    // ---- B: ring test + score of the queued pixels in one pass (dense); corners re-compacted in place
    int cn = 0;
    for (int e0 = 0; e0 < qn; e0 += 64) {
      const int e = e0 + lane;
      bool corner = false;
      unsigned q = 0;
      if (e < qn) {
        q = queue[e];
        const int y = (int)__umulhi(q, magic), x = (int)q - __mul24(y, zw);
        const int sc_v = fast_ring_score(tile + __mul24(y + 3, TP) + x + 3 + sh, TP);
        corner = sc_v >= th;
        if (corner && sc_v > 0) sc[__mul24(y + 1, SP) + x + 1] = (uint8_t)sc_v;
      }
      const unsigned long long m = __builtin_amdgcn_ballot_w64(corner);   // all reads of this chunk precede the writes (cn <= e0)
      if (corner) queue[cn + __popcll(m & lt)] = (uint16_t)q;
      cn += __popcll(m);
    }
    qn = cn;
#ifdef FAST_SKIP_C
    qn = 0;
#endif
    FPROF(2);
    FPROF(3);
    __syncthreads();
    FPROF(4);
    // ---- C: NMS + ordered emission
    {
      unsigned long long bits = 0;
      int cnt = 0;
      const int nj = (qn + 63) >> 6;
      for (int j = 0; j < nj; j++) {
        const int e = j * 64 + lane;
        bool keep = false;
        if (e < qn) {
          const unsigned q = queue[e];
          const int y = (int)__umulhi(q, magic), x = (int)q - __mul24(y, zw);
          const int yz = sr0 + y;
          // r3: the nine scores are read FIRST and reduced with max3 (strictly greater than all eight neighbours = greater than their
          // maximum).  Written as a short-circuit && chain this was nine dependent LDS round trips per chunk, each behind a branch
          // (ds_read_u8 -> s_waitcnt -> v_cmp -> s_and_saveexec); rows of the halo are inside the score map too, so nothing is conditional.
          const uint8_t* p = sc + __mul24(y + 1, SP) + x + 1;
          const int s = p[0];
          const int n0 = p[-1], n1 = p[1], n2 = p[-SP - 1], n3 = p[-SP], n4 = p[-SP + 1], n5 = p[SP - 1], n6 = p[SP], n7 = p[SP + 1];
          const int nm = max(max(max(n0, n1), n2), max(max(max(n3, n4), n5), max(n6, n7)));
          keep = (s > 0) & (s > nm) & (yz >= r0) & (yz < r1);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        cnt += __popcll(m);
        bits |= (unsigned long long)keep << j;
      }
      if (lane == 0) wcnt[wave] = cnt;
      __syncthreads();
      int base = total;
      for (int w = 0; w < wave; w++) base += wcnt[w];
      const int strip_total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      for (int j = 0; j < nj; j++) {
        const bool keep = (bits >> j) & 1ull;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
          const unsigned q = queue[j * 64 + lane];
          const int y = (int)__umulhi(q, magic), x = (int)q - __mul24(y, zw);
          const int s = sc[__mul24(y + 1, SP) + x + 1];
          const unsigned pos = (unsigned)(base + __popcll(m & lt));
          if (pos < C.cap)
            out[pos] = ((uint32_t)s << 24) | ((uint32_t)(C.zy0 + sr0 + y) << 12) | (uint32_t)(C.zx0 + x);
        }
        base += __popcll(m);
      }
      total += strip_total;
    }
    FPROF(5);
    __syncthreads();
    FPROF(6);
  }
  if (tid == 0) cell_count[(size_t)frame * ncells_total + cell] = min(total, (int)C.cap);
  FPROF_FLUSH;
}

// ------------------------------------------------------------------------------------------
// Selection (src/ORBextractor.cc:541-605), three steps, one launch each (k_select_quota, k_select_cells [+ k_select_bigcells],
// k_select_final):
//   1. quota redistribution loop            :541-575 (serial per level)
//   2. per-cell retainBest + resize         :586-588   one WAVE per cell: the cell's candidates are staged into LDS, trimmed
//      with the wave-parallel introselect replay (introselect_wave.h) and the survivors written straight to their slot of the
//      level list (:591-597; the kept counts, hence the offsets, are known after step 1)
//   3. level-wide retainBest + resize       :601-604   one wave per level, same replay, list in LDS
// Cells with more candidates than the LDS buffers, or levels whose list exceeds the LDS list, fall back to the serial replay
// in HBM (never seen on VGA / 720p frames; kept for correctness, tests/test_orb_gpu.py::test_selection_paths).
// r1 / early r2 ran the three steps in ONE kernel (8 waves per (level, frame)): those waves sat idle through the serial quota
// loop and the one-wave level step, and a level with 30 cells needed 4 rounds of 8 waves: 0.52 ms against 0.28 ms split.
// ------------------------------------------------------------------------------------------
#define SEL_WAVES 8         // big-cell buffer = SEL_WAVES x the per-geometry cell capacity (what the single kernel's 8 waves held)
#define SEL_LIST_CAP 1536   // level list (LDS)
#define SEL_MAX_CELLS 512
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// ------------------------------------------------------------------------------------------
//   k_select_quota   one wave per frame, lane l runs level l's quota loop (counts staged through LDS); writes the kept count
//                    and list offset of every cell and the list length of every level
//   k_select_cells   one wave per (cell, frame): retainBest of the cell in LDS, survivors to their slot of the level list in
//                    HBM; k_select_bigcells: the (rare) cells with more candidates than that kernel's small buffer
//   k_select_final   one wave per (level, frame): level-wide retainBest (list through LDS), selected keys + count
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_select_quota(const OrbPlan* __restrict__ P, const CellGeom* __restrict__ cells,
                                                     const int32_t* __restrict__ cell_count, int32_t* __restrict__ cell_keep,
                                                     int32_t* __restrict__ cell_off, int32_t* __restrict__ lvl_m) {
  extern __shared__ __attribute__((aligned(16))) int s_q[];   // [4][ncells]: total, evaluated / bNoMore, retain, offset
  const int frame = blockIdx.x, lane = threadIdx.x, nc = P->ncells;
  int* s_total = s_q;
  int* s_flag = s_q + nc;
  int* s_retain = s_q + 2 * nc;
  int* s_offv = s_q + 3 * nc;
  const int32_t* cc = cell_count + (size_t)frame * nc;
  for (int c = lane; c < nc; c += 64) {
    s_total[c] = cc[c];
    s_flag[c] = cells[c].evaluated;
  }
  sdsel::wave_fence();
  if (lane < P->nlevels) {
    const LevelGeom& L = P->lv[lane];
    int M = 0;
    if (L.ncells > 0 && L.quota > 0) {
      const int nC = L.ncells, c0 = L.cell0, nfc = L.nfeaturesCell;
      int nNoMore = 0, nToDistribute = 0;
      for (int c = c0; c < c0 + nC; c++) {
        const int nKeys = s_total[c];
        if (!s_flag[c]) {                // cell not evaluated: the reference `continue`s (nToRetain=0, bNoMore=false)
          s_retain[c] = 0;
          s_flag[c] = 0;                 // from here on: the bNoMore flag
          continue;
        }
        if (nKeys > nfc) {
          s_retain[c] = nfc;
          s_flag[c] = 0;
        } else {
          s_retain[c] = nKeys;
          nToDistribute += nfc - nKeys;
          s_flag[c] = 1;
          nNoMore++;
        }
      }
      while (nToDistribute > 0 && nNoMore < nC) {
        // nfeaturesCell + ceil((float)nToDistribute/(nCells-nNoMore))
        const int nNew = nfc + (int)ceilf((float)nToDistribute / (float)(nC - nNoMore));
        nToDistribute = 0;
        for (int c = c0; c < c0 + nC; c++) {
          if (!s_flag[c]) {
            if (s_total[c] > nNew) {
              s_retain[c] = nNew;
            } else {
              s_retain[c] = s_total[c];
              nToDistribute += nNew - s_total[c];
              s_flag[c] = 1;
              nNoMore++;
            }
          }
        }
      }
      for (int c = c0; c < c0 + nC; c++) {   // kept counts and their offsets in the level list
        const int k = min(s_total[c], s_retain[c]);
        s_retain[c] = k;
        s_offv[c] = M;
        M += k;
      }
    } else {
      for (int c = L.cell0; c < L.cell0 + max(L.ncells, 0); c++) { s_retain[c] = 0; s_offv[c] = 0; }
    }
    lvl_m[(size_t)frame * P->nlevels + lane] = M;
  }
  sdsel::wave_fence();
  for (int c = lane; c < nc; c += 64) {
    cell_keep[(size_t)frame * nc + c] = s_retain[c];
    cell_off[(size_t)frame * nc + c] = s_offv[c];
  }
}

__global__ __launch_bounds__(64) void k_select_cells(const OrbPlan* __restrict__ P, const CellGeom* __restrict__ cells,
                                                     uint32_t* __restrict__ cand, const int32_t* __restrict__ cell_count,
                                                     const int32_t* __restrict__ cell_keep, const int32_t* __restrict__ cell_off,
                                                     uint32_t* __restrict__ lvl_scratch, int buf_cap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_cellbuf[];   // [buf_cap] candidates | u16 [2 * WAVE_SEL_CAP] stop tables
  const int c = blockIdx.x, frame = blockIdx.y, lane = threadIdx.x, nc = P->ncells;
  const int keep = cell_keep[(size_t)frame * nc + c];
  if (keep <= 0) return;
  const int n = cell_count[(size_t)frame * nc + c], o = cell_off[(size_t)frame * nc + c];
  const CellGeom cg = cells[c];
  uint32_t* src = cand + (size_t)frame * P->cand_per_frame + cg.cand_off;
  uint32_t* glist = lvl_scratch + (size_t)frame * P->cand_per_frame + P->lv[cg.level].cand_off + o;
  if (n > buf_cap) return;   // k_select_bigcells
  lds_u32* buf = (lds_u32*)s_cellbuf;
  for (int i = lane; i < n; i += 64) buf[i] = src[i];
  sdsel::wave_fence();
  if (n > keep) sdsel::wave_nth_element(buf, n, keep, (sdsel::lds_u16*)(s_cellbuf + buf_cap));
  for (int i = lane; i < keep; i += 64) glist[i] = buf[i];
}

// cells with more candidates than k_select_cells' buffer (dense texture, noise): one wave per (level, frame) walks its level's
// cells and trims those, one at a time, in a buffer of big_cap entries; beyond that the serial replay in HBM (lane 0).
// On ordinary frames every workgroup finds nothing to do and leaves after one load.
__global__ __launch_bounds__(64) void k_select_bigcells(const OrbPlan* __restrict__ P, const CellGeom* __restrict__ cells,
                                                        uint32_t* __restrict__ cand, const int32_t* __restrict__ cell_count,
                                                        const int32_t* __restrict__ cell_keep, const int32_t* __restrict__ cell_off,
                                                        uint32_t* __restrict__ lvl_scratch, int small_cap, int big_cap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_cellbuf[];   // [big_cap] candidates | u16 [2 * WAVE_SEL_CAP] stop tables
  const int level = blockIdx.x, frame = blockIdx.y, lane = threadIdx.x, nc = P->ncells;
  const LevelGeom& L = P->lv[level];
  if (L.ncells <= 0 || L.quota <= 0) return;
  for (int c0 = 0; c0 < L.ncells; c0 += 64) {
    const int cl = L.cell0 + c0 + lane;
    const bool mine = c0 + lane < L.ncells && cell_count[(size_t)frame * nc + cl] > small_cap && cell_keep[(size_t)frame * nc + cl] > 0;
    unsigned long long todo = __ballot(mine);
    while (todo) {
      const int c = L.cell0 + c0 + __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int n = cell_count[(size_t)frame * nc + c], keep = cell_keep[(size_t)frame * nc + c], o = cell_off[(size_t)frame * nc + c];
      uint32_t* src = cand + (size_t)frame * P->cand_per_frame + cells[c].cand_off;
      uint32_t* glist = lvl_scratch + (size_t)frame * P->cand_per_frame + L.cand_off + o;
      if (n <= big_cap) {
        lds_u32* buf = (lds_u32*)s_cellbuf;
        for (int i = lane; i < n; i += 64) buf[i] = src[i];
        sdsel::wave_fence();
        if (n > keep) sdsel::wave_nth_element(buf, n, keep, (sdsel::lds_u16*)(s_cellbuf + big_cap));
        for (int i = lane; i < keep; i += 64) glist[i] = buf[i];
        sdsel::wave_fence();   // the buffer is reused by the next cell
      } else if (lane == 0) {
        if (n > keep) sdsel::nth_element(src, n, keep);
        for (int i = 0; i < keep; i++) glist[i] = src[i];
      }
    }
  }
}

__global__ __launch_bounds__(64) void k_select_final(const OrbPlan* __restrict__ P, const int32_t* __restrict__ lvl_m,
                                                     uint32_t* __restrict__ lvl_scratch, uint32_t* __restrict__ sel,
                                                     int32_t* __restrict__ sel_count) {
  __shared__ uint32_t s_list[SEL_LIST_CAP];
  __shared__ uint16_t s_tmp[2 * WAVE_SEL_CAP];
  const int level = blockIdx.x, frame = blockIdx.y, lane = threadIdx.x;
  const LevelGeom& L = P->lv[level];
  int32_t* out_n = sel_count + (size_t)frame * P->nlevels + level;
  if (L.ncells <= 0 || L.quota <= 0) {
    if (lane == 0) *out_n = 0;
    return;
  }
  const int M = lvl_m[(size_t)frame * P->nlevels + level];
  uint32_t* glist = lvl_scratch + (size_t)frame * P->cand_per_frame + L.cand_off;
  uint32_t* dst = sel + (size_t)frame * P->nsel + L.sel_off;
  int Mout = M;
  if (M > L.quota) {
    Mout = L.quota;
    if (M <= SEL_LIST_CAP) {
      for (int i = lane; i < M; i += 64) s_list[i] = glist[i];
      sdsel::wave_fence();
      sdsel::wave_nth_element((lds_u32*)s_list, M, L.quota, (sdsel::lds_u16*)s_tmp);
      for (int i = lane; i < Mout; i += 64) dst[i] = s_list[i];
    } else {
      if (lane == 0) sdsel::nth_element(glist, M, L.quota);
      sdsel::wave_fence();
      for (int i = lane; i < Mout; i += 64) dst[i] = glist[i];
    }
  } else {
    for (int i = lane; i < Mout; i += 64) dst[i] = glist[i];
  }
  if (lane == 0) *out_n = Mout;
}

// ------------------------------------------------------------------------------------------
// k_blur: GaussianBlur(7x7, sigma 2, REFLECT_101) in OpenCV's 8-bit fixed point: integer taps
// round(g*256) = {18,34,49,55,49,34,18}, row pass in int32, column pass (sum + 2^15) >> 16,
// saturated.  The padded pyramid already holds the REFLECT_101 border, so no border logic.
// Register sliding window, no LDS: a thread owns 4 adjacent output columns of a 32-row band,
// reads each input row as 3 aligned dwords (12 bytes cover the 4+6 taps), keeps the last 7
// row-pass results in registers (as pairs of vertically adjacent rows) and emits one packed 4-byte
// store per row.  The taps run on the integer dot-product instructions (v_dot4_u32_u8 for the
// row pass over bytes, v_dot2_u32_u16 for the column pass over the 16-bit row results): 997
// vector instructions per 128 outputs instead of 1570 with mul24 / mad chains, same integers.  Only levels that
// own keypoints are blurred by the reference (src/ORBextractor.cc:655-660); here every level is
// (the blurred image is not an output), which frees the stage from waiting for the selection.
// ------------------------------------------------------------------------------------------
#define BLUR_RB 32                  // output rows per wave
#define SD_BLUR_SHIFT 1             // column offset of the blurred levels against the pyramid's layout (see k_blur)
#ifndef BLUR_PF
#define BLUR_PF 8                   // input rows in flight per thread (r3, with the 16-lane groups: 8 rows ORB-only 287.8 k, 6 rows 284.4 k, 4 rows 281.1 k)
#endif

__global__ __launch_bounds__(256) void k_blur(const OrbPlan* __restrict__ P, const BlurTile* __restrict__ tiles,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                              const int32_t* __restrict__ sel_count) {
  unsigned uframe, utile;
  xcd_frame_block(uframe, utile);
  const BlurTile T = tiles[utile];
  const int frame = (int)uframe, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  (void)sel_count;   // every level is blurred: the stage then depends on the pyramid only and overlaps FAST/selection
  const LevelGeom L = P->lv[T.level];
  // r3: the level is cut into 64-column strips x 32-row bands and the (strip, band) pairs are dealt FLAT to 16-lane groups (strip
  // fastest: the four groups of a wave read 256 contiguous columns wherever the level is that wide).  With a whole wave per 256
  // columns the levels' widths (640, 533, 444, 370, 309, 257, 214, 179) left a quarter of all lanes without a column.
  const int pr = T.p0 + wave * 4 + (lane >> 4);
  // (a single strip has no 32-bit magic number: 2^32 / 1 does not fit)
  const int band = T.nstrips == 1 ? pr : (int)__umulhi((unsigned)pr, T.magic), strip = pr - band * T.nstrips;
  const int x0 = strip * 64 + (lane & 15) * 4;
  const int y0 = band * BLUR_RB;
  if (x0 >= L.w || y0 >= L.h) return;
  const size_t fo = (size_t)frame * P->pyr_frame_bytes + L.off;
  // input bytes for outputs x0..x0+3: padded columns x0+16 .. x0+27 (4-byte aligned)
  const uint8_t* src = pyr + fo + (size_t)(y0 + SD_EDGE - 3) * L.pstride + (x0 + SD_EDGE - 3);
  // the blurred level sits ONE column to the right of the pyramid's layout (interior from padded column 20: its rows have the
  // room, and nothing reads its border): a lane's four output pixels are then an aligned dword
  uint8_t* dst = blur + fo + (size_t)(y0 + SD_EDGE) * L.pstride + (x0 + SD_EDGE + SD_BLUR_SHIFT);
  const int nrows = min(BLUR_RB, L.h - y0);
  const bool full = x0 + 3 < L.w;
  // Row pass: out[k] = sum_i tap[i] * byte[k + i] as two v_dot4_u32_u8 over byte windows cut out of the three dwords with
  // v_alignbyte (<= 255 * 257 = 65535: fits 16 bits).  Column pass: vertically adjacent row results packed in pairs
  // P[r] = R[r] | R[r+1] << 16, so an output is three v_dot2_u32_u16 and one mad; (s + 2^15) >> 16 and the saturation to 255
  // are one v_perm (high halves of two sums) + v_sat_pk_u8_i16 per pixel pair.
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const uint32_t T0 = 18u | 34u << 8 | 49u << 16 | 55u << 24, T1 = 49u | 34u << 8 | 18u << 16;
  const us2 T01 = {18, 34}, T23 = {49, 55}, T45 = {49, 34};
  uint32_t PP[6][4], prevR[4];
  // Input rows are fetched BLUR_PF rows ahead of their use (the row index is clamped to the band's last input row instead of
  // being branched around, so the loads are straight-line code): without this every row was load -> s_waitcnt vmcnt(0) ->
  // compute, 38 dependent global round trips per wave.
  uint32_t pf[BLUR_PF][3];
  const int last_in = nrows + 5;
  const uint32_t srow = (uint32_t)L.pstride;
#pragma unroll
  for (int d = 0; d < BLUR_PF; d++) {
    const uint32_t* rp = (const uint32_t*)(src + (size_t)((uint32_t)min(d, last_in) * srow));
    pf[d][0] = rp[0]; pf[d][1] = rp[1]; pf[d][2] = rp[2];
  }
#pragma unroll
  for (int r = 0; r < BLUR_RB + 6; r++) {
    const uint32_t w0 = pf[r % BLUR_PF][0], w1 = pf[r % BLUR_PF][1], w2 = pf[r % BLUR_PF][2];
    if (r + BLUR_PF < BLUR_RB + 6) {
      const uint32_t* rp = (const uint32_t*)(src + (size_t)((uint32_t)min(r + BLUR_PF, last_in) * srow));
      pf[r % BLUR_PF][0] = rp[0]; pf[r % BLUR_PF][1] = rp[1]; pf[r % BLUR_PF][2] = rp[2];
    }
    if (r < nrows + 6) {
      uint32_t R[4];
      R[0] = __builtin_amdgcn_udot4(w0, T0, __builtin_amdgcn_udot4(w1, T1, 0u, false), false);
#pragma unroll
      for (int k = 1; k < 4; k++)
        R[k] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, k), T0,
                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, k), T1, 0u, false), false);
      if (r >= 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) PP[(r - 1) % 6][k] = prevR[k] | (R[k] << 16);
      }
      if (r >= 6) {
        // output row r-6 = taps over row results r-6 .. r = PP[r-6], PP[r-4], PP[r-2] and R[r]
        uint32_t sum[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint32_t a = (uint32_t)__mul24(18, (int)R[k]) + (1u << 15);
          a = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, PP[(r - 2) % 6][k]), T45, a, false);
          a = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, PP[(r - 4) % 6][k]), T23, a, false);
          sum[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, PP[(r - 6) % 6][k]), T01, a, false);
        }
        uint32_t h01 = __builtin_amdgcn_perm(sum[1], sum[0], 0x07060302u), h23 = __builtin_amdgcn_perm(sum[3], sum[2], 0x07060302u);
        uint32_t p01, p23;
        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(p01) : "v"(h01));   // values 0..257 as i16 -> u8, saturated
        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(p23) : "v"(h23));
        const uint32_t packed = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
        uint8_t* o = dst + (size_t)(r - 6) * L.pstride;
        if (full) {
          *(uint32_t*)o = packed;
        } else {
          for (int k = 0; k < 4; k++)
            if (x0 + k < L.w) o[k] = (uint8_t)(packed >> (8 * k));
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) prevR[k] = R[k];
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_orient_desc: one wavefront per selected keypoint.
//   IC_Angle: integer moments over the r=15 disc (umax table), fastAtan2 polynomial in f32.
//   rBRIEF : a = cosf(angle*pi/180), b = sinf(..) (host-libm-exact, sd_sincosf.h), 256 tests
//            t0 < t1 on the blurred level at cvRound-rotated offsets; lane i evaluates tests
//            i, i+64, i+128, i+192 so each __ballot is 8 descriptor bytes.
//   Output keypoint: pt scaled by mvScaleFactor[level] AFTER description (:669-674).
// Compiled with -ffp-contract=off: every float op below is a single IEEE operation.
// ------------------------------------------------------------------------------------------
__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};
// umax[v] of the r = 15 disc (src/ORBextractor.cc:441-456); compile-time so that the unrolled row loop compares against literals
static constexpr int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

#ifndef DESC_KPW
#define DESC_KPW 2   // keypoints per wave
#endif
#define DESC_PATCH_LD 3                       // 8-byte loads per lane and keypoint: 3 x 64 >= 37 rows x 5 dword pairs
#define DESC_PATCH_DW (2 * DESC_PATCH_LD * 64)
__global__ __launch_bounds__(256) void k_orient_desc(const OrbPlan* __restrict__ P, const uint8_t* __restrict__ pyr,
                                                     const uint8_t* __restrict__ blur,
                                                     const uint32_t* __restrict__ sel,
                                                     const int32_t* __restrict__ sel_count,
                                                     sd_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                     int32_t* __restrict__ nout, int cap, int n_frames, int bpf) {
  // XCD-aware block order: workgroups are handed to the 8 XCDs round-robin, and each XCD has its own
  // 4 MB L2.  Block b works on frame 8 * (b / (8 * bpf)) + (b % 8), so all workgroups of a frame run on
  // ONE XCD and the frame's two pyramids (2.8 MB at VGA) stay L2-resident while its ~1000 overlapping
  // 31 x 31 patches are read (measured: HBM fetch 6.1 GB -> see DESIGN.md section 6).
  const int bx = blockIdx.x;
  const int xcd = bx & 7, t = bx >> 3;
  const int frame = (t / bpf) * 8 + xcd, blk = t % bpf;
  if (frame >= n_frames) return;
  const int lane = threadIdx.x & 63;
  // Two keypoints per wave, interleaved: the kernel is a chain of dependent gathers (key -> 31 x 31 patch -> angle ->
  // 512 sample points), so a wave with two independent chains in flight keeps twice the loads outstanding.
  // Everything that is the same for the whole wave (keypoint position, level geometry, patch base pointers, angle, cos, sin) is
  // made scalar with readfirstlane / readlane: the gathers then use the SGPR-base + 32-bit-offset addressing form (no 64-bit
  // vector address arithmetic), and the angle / cos / sin of BOTH keypoints are evaluated once, in the two halves of the wave.
  const int g0 = __builtin_amdgcn_readfirstlane((blk * 4 + (int)(threadIdx.x >> 6)) * DESC_KPW);   // output slots g0 .. g0 + DESC_KPW - 1
  const int32_t* sc = sel_count + (size_t)frame * P->nlevels;
  int level[DESC_KPW], idx[DESC_KPW], acc = 0;
#pragma unroll
  for (int q = 0; q < DESC_KPW; q++) { level[q] = -1; idx[q] = 0; }
  for (int l = 0; l < P->nlevels; l++) {
    const int n = sc[l];
#pragma unroll
    for (int q = 0; q < DESC_KPW; q++)
      if (level[q] < 0 && g0 + q < acc + n) {
        level[q] = l;
        idx[q] = g0 + q - acc;
      }
    acc += n;
  }
  if (blk == 0 && threadIdx.x == 0) nout[frame] = min(acc, cap);
  bool live[DESC_KPW];
  int X[DESC_KPW], Y[DESC_KPW], resp[DESC_KPW], step[DESC_KPW];
  const uint8_t* pbase[DESC_KPW];   // pixel (X - 19, Y - 19) of the level: every offset below is non-negative (the border is 19 px)
  const uint8_t* bbase[DESC_KPW];   // same position in the blurred level
  float scale[DESC_KPW], kpsize[DESC_KPW];
#pragma unroll
  for (int q = 0; q < DESC_KPW; q++) {
    live[q] = level[q] >= 0 && g0 + q < cap;
    const int lv = live[q] ? level[q] : 0;
    const LevelGeom& L = P->lv[lv];
    const uint32_t key = live[q] ? sel[(size_t)frame * P->nsel + L.sel_off + idx[q]] : 0u;
    X[q] = live[q] ? (int)(key & 0xfff) : SD_EDGE;     // dead slot: a harmless in-range position
    Y[q] = live[q] ? (int)((key >> 12) & 0xfff) : SD_EDGE;
    resp[q] = key >> 24;
    step[q] = L.pstride;
    scale[q] = L.scale;
    kpsize[q] = L.kpsize;
    const size_t fo = (size_t)frame * P->pyr_frame_bytes + L.off + (size_t)(Y[q] + SD_EDGE - 19) * L.pstride + X[q] + SD_EDGE - 19;
    pbase[q] = pyr + fo;
    bbase[q] = blur + fo + SD_BLUR_SHIFT;
  }
  if (!live[0]) return;   // slots are filled in order: no first keypoint, no second

  // ---- the blurred 37 x 37 patch of each keypoint (the rotated test points reach +-18) goes to LDS as 37 rows of ten aligned
  // dwords, fetched NOW, beside the IC_Angle rows: the 512 sample reads of a keypoint were byte gathers over ~30 cache lines
  // per instruction that could only start once the angle was known
  __shared__ __attribute__((aligned(16))) uint32_t s_patch[4][DESC_KPW][DESC_PATCH_DW];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint2 pdw[DESC_KPW][DESC_PATCH_LD];
  int psh[DESC_KPW];
#pragma unroll
  for (int q = 0; q < DESC_KPW; q++) {
    const uint8_t* a = bbase[q] + step[q] + 1;            // pixel (X - 18, Y - 18)
    psh[q] = (int)((uintptr_t)a & 3);
    const uint8_t* a4 = a - psh[q];
#pragma unroll
    for (int k = 0; k < DESC_PATCH_LD; k++) {
      const int e = min(k * 64 + lane, 37 * 5 - 1);
      const int row = (int)(((unsigned)e * 13108u) >> 16);   // e / 5, e < 192
      const int d = e - row * 5;
      pdw[q][k] = *(const uint2*)(a4 + (uint32_t)(__mul24(row, step[q]) + 8 * d));   // 4-byte aligned: one global_load_dwordx2
    }
  }

  // ---- IC_Angle: m10 = sum u * I, m01 = sum v * I over the disc.  Lane = (column u, half): the lower half of the wave walks
  // the rows +v, the upper half the rows -v; per row one load per keypoint, u * (sum of the column) and +-(sum of v * I) at
  // the end (integer sums: any order is exact).
  int m10[DESC_KPW], m01[DESC_KPW];
  {
    const int u = (lane & 31) - 15;
    const int half = lane >> 5;   // 0: rows +v, 1: rows -v
    const bool act = (lane & 31) < 31;
    const int au = abs(u);
    uint32_t off[DESC_KPW];
    int dstep[DESC_KPW], colsum[DESC_KPW], vsum[DESC_KPW];
#pragma unroll
    for (int q = 0; q < DESC_KPW; q++) {
      off[q] = (uint32_t)(19 * step[q] + 19 + u);
      dstep[q] = half ? -step[q] : step[q];
      colsum[q] = vsum[q] = 0;
    }
    // All 2 x 16 row loads are issued first, unconditionally (every position of the 31 x 31 square around a keypoint lies
    // inside the padded level; lane 31 / 63 reads column +16, also inside): straight-line code, 32 loads in flight.  With the
    // disc test as a branch around each row the generated code was load -> s_waitcnt vmcnt(0) -> add, sixteen dependent
    // round trips.  The disc is applied afterwards: umax[] is non-increasing, so row v counts iff v <= vlim(|u|).
    int vals[16][DESC_KPW];
#pragma unroll
    for (int v = 0; v <= 15; v++) {
#pragma unroll
      for (int q = 0; q < DESC_KPW; q++) {
        vals[v][q] = pbase[q][off[q]];
        off[q] += (uint32_t)dstep[q];
      }
    }
#pragma unroll
    for (int q = 0; q < DESC_KPW; q++)
#pragma unroll
      for (int k = 0; k < DESC_PATCH_LD; k++) ((uint2*)s_patch[wv][q])[k * 64 + lane] = pdw[q][k];   // pair e = dwords 2 e, 2 e + 1 of the patch
    int vlim = -1;   // largest row index of this lane's column inside the disc (-1: lane outside)
#pragma unroll
    for (int v = 0; v <= 15; v++) vlim += (act && au <= kUmax[v]) ? 1 : 0;
#pragma unroll
    for (int v = 0; v <= 15; v++) {
      const bool in = v <= vlim && !(half && v == 0);
#pragma unroll
      for (int q = 0; q < DESC_KPW; q++) {
        const int val = in ? vals[v][q] : 0;
        colsum[q] += val;
        vsum[q] += v * val;
      }
    }
#pragma unroll
    for (int q = 0; q < DESC_KPW; q++) {   // DPP reduction of the device library
      m10[q] = __ockl_wfred_add_i32(u * colsum[q]);
      m01[q] = __ockl_wfred_add_i32(half ? -vsum[q] : vsum[q]);
    }
  }
  // ---- angle, cos, sin: keypoint 0 in lanes 0..31, keypoint 1 in lanes 32..63, one evaluation
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  float angle[DESC_KPW], ca[DESC_KPW], sb[DESC_KPW];
  {
    static_assert(DESC_KPW == 2, "the angle evaluation splits the wave in two");
    const float my = (float)(lane < 32 ? m01[0] : m01[1]), mx = (float)(lane < 32 ? m10[0] : m10[1]);
    const float ang = fast_atan2_deg(my, mx);
    const float arad = ang * factorPI;
    const float c = sdsc::cosf_glibc(arad), sn = sdsc::sinf_glibc(arad);
#pragma unroll
    for (int q = 0; q < DESC_KPW; q++) {
      angle[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ang), 32 * q));
      ca[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 32 * q));
      sb[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sn), 32 * q));
    }
  }
  // ---- steered rBRIEF on the blurred level
  unsigned long long words[DESC_KPW][4];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // this wave's own patch stores before its sample reads
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int t = j * 64 + lane;
    const char4 pt = *(const char4*)(&c_pattern[t * 4]);
    const float x0 = (float)pt.x, y0 = (float)pt.y, x1 = (float)pt.z, y1 = (float)pt.w;
    int t0[DESC_KPW], t1[DESC_KPW];
#pragma unroll
    for (int q = 0; q < DESC_KPW; q++) {
      const float a = ca[q], b = sb[q];
      // center[cvRound(x*b + y*a)*step + cvRound(x*a - y*b)]
      const int r0 = __float2int_rn(x0 * b + y0 * a), q0 = __float2int_rn(x0 * a - y0 * b);
      const int r1 = __float2int_rn(x1 * b + y1 * a), q1 = __float2int_rn(x1 * a - y1 * b);
      const uint8_t* pb = (const uint8_t*)s_patch[wv][q] + (18 * 40 + 18 + psh[q]);
      t0[q] = pb[__mul24(r0, 40) + q0];
      t1[q] = pb[__mul24(r1, 40) + q1];
    }
#pragma unroll
    for (int q = 0; q < DESC_KPW; q++) words[q][j] = __builtin_amdgcn_ballot_w64(t0[q] < t1[q]);
  }
#pragma unroll
  for (int q = 0; q < DESC_KPW; q++) {
    if (!live[q]) continue;
    const int g = g0 + q;
    if (lane < 4) {
      unsigned long long w = lane == 0 ? words[q][0] : lane == 1 ? words[q][1] : lane == 2 ? words[q][2] : words[q][3];
      *(unsigned long long*)(desc + ((size_t)frame * cap + g) * 32 + lane * 8) = w;
    }
    if (lane == 0) {
      sd_keypoint k;
      float fx = (float)X[q], fy = (float)Y[q];
      if (level[q] != 0) {
        fx = fx * scale[q];
        fy = fy * scale[q];
      }
      k.x = fx;
      k.y = fy;
      k.size = kpsize[q];
      k.angle = angle[q];
      k.response = (float)resp[q];
      k.octave = level[q];
      k.class_id = -1;
      kps[(size_t)frame * cap + g] = k;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_undistort: Frame::UndistortKeyPoints (src/Frame.cc:335-366) = cv::undistortPoints(pts, pts, K,
// dist, Mat(), K), OpenCV 3.2: normalise, 5 fixed-point iterations of the radial/tangential
// model in fp64, re-project with P = K, narrow to f32.  One thread per keypoint.
// ------------------------------------------------------------------------------------------
struct DistParams { double fx, fy, cx, cy, k[12]; };

__global__ void k_undistort(const sd_keypoint* __restrict__ kps, sd_keypoint* __restrict__ kps_un, const int32_t* __restrict__ nout,
                            int cap, DistParams D) {
  const int frame = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nout[frame]) return;
  sd_keypoint kp = kps[(size_t)frame * cap + i];
  const double ifx = 1. / D.fx, ify = 1. / D.fy;
  double x = kp.x, y = kp.y;
  x = (x - D.cx) * ifx;
  y = (y - D.cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = (1 + ((D.k[7] * r2 + D.k[6]) * r2 + D.k[5]) * r2) / (1 + ((D.k[4] * r2 + D.k[1]) * r2 + D.k[0]) * r2);
    const double deltaX = 2 * D.k[2] * x * y + D.k[3] * (r2 + 2 * x * x) + D.k[8] * r2 + D.k[9] * r2 * r2;
    const double deltaY = D.k[2] * (r2 + 2 * y * y) + 2 * D.k[3] * x * y + D.k[10] * r2 + D.k[11] * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  const double xx = D.fx * x + 0.0 * y + D.cx;
  const double yy = 0.0 * x + D.fy * y + D.cy;
  const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
  kp.x = (float)(xx * ww);
  kp.y = (float)(yy * ww);
  kps_un[(size_t)frame * cap + i] = kp;
}

}  // namespace sd

// ==========================================================================================
// host side: handle + C ABI
// ==========================================================================================
using namespace sd;

#include "orb_internal.h"
static const char* kStageNames[ST_COUNT] = {"pyramid", "fast_nms", "select", "blur", "orient_desc"};

static int free_geom(sd_orb* h) {
  void* ptrs[] = {h->d_cells, h->d_tiles, h->d_coef, h->pyr_set[0], h->pyr_set[1], h->d_blur, h->d_cand, h->d_scratch,
                  h->d_cell_count, h->d_sel, h->d_sel_count, h->d_cell_keep, h->d_cell_off, h->d_lvl_m};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  h->pyr_set[0] = h->pyr_set[1] = nullptr;
  h->d_cells = nullptr; h->d_tiles = nullptr; h->d_coef = nullptr; h->d_pyr = nullptr; h->d_blur = nullptr;
  h->d_cand = nullptr; h->d_scratch = nullptr; h->d_cell_count = nullptr; h->d_sel = nullptr; h->d_sel_count = nullptr;
  h->d_cell_keep = nullptr; h->d_cell_off = nullptr; h->d_lvl_m = nullptr;
  return SD_OK;
}

static void select_set(sd_orb* h, int sidx) {
  h->set = sidx;
  h->d_pyr = h->pyr_set[sidx];
  h->d_kps = h->kps_set[sidx];
  h->d_kps_un = h->kps_un_set[sidx];
  h->d_desc = h->desc_set[sidx];
  h->d_nout = h->nout_set[sidx];
}

static int wait_trackers(sd_orb* h) {   // host-side: nothing may still read any output set
  for (int i = 0; i < 2; i++)
    if (h->set_busy[i]) {
      SD_HIP_CHECK(hipEventSynchronize(h->ev_set_free[i]));
      h->set_busy[i] = false;
    }
  return SD_OK;
}

static void drop_graphs(sd_orb* h);

// Geometry (re)build.  Everything that can fail -- planning, allocation, upload -- works on a LOCAL plan and the handle is
// marked "no geometry" first, so a failure (a frame too small to plan, an allocation that does not fit) leaves a handle
// that rebuilds from scratch on its next call instead of one whose host plan no longer matches its device buffers.
static int build_geometry(sd_orb* h, const HostPlan& hp) {
  const size_t B = h->max_batch;
  const size_t slack = 4096;
  SD_HIP_CHECK(hipMalloc(&h->d_cells, std::max<size_t>(hp.cells.size(), 1) * sizeof(CellGeom)));
  SD_HIP_CHECK(hipMalloc(&h->d_tiles, std::max<size_t>(hp.blur_tiles.size(), 1) * sizeof(BlurTile)));
  SD_HIP_CHECK(hipMalloc(&h->d_coef, std::max<size_t>(hp.coef.size(), 1) * sizeof(int32_t)));
  for (int i = 0; i < h->nsets; i++) {
    SD_HIP_CHECK(hipMalloc(&h->pyr_set[i], hp.plan.pyr_frame_bytes * B + slack));
    SD_HIP_CHECK(hipMemsetAsync(h->pyr_set[i], 0, hp.plan.pyr_frame_bytes * B + slack, h->stream));
  }
  select_set(h, 0);
  SD_HIP_CHECK(hipMalloc(&h->d_blur, hp.plan.pyr_frame_bytes * B + slack));
  SD_HIP_CHECK(hipMalloc(&h->d_cand, std::max<size_t>(hp.plan.cand_per_frame, 1) * B * 4));
  SD_HIP_CHECK(hipMalloc(&h->d_scratch, std::max<size_t>(hp.plan.cand_per_frame, 1) * B * 4));
  SD_HIP_CHECK(hipMalloc(&h->d_cell_count, std::max<size_t>(hp.plan.ncells, 1) * B * 4));
  SD_HIP_CHECK(hipMalloc(&h->d_sel, std::max<size_t>(hp.plan.nsel, 1) * B * 4));
  SD_HIP_CHECK(hipMalloc(&h->d_sel_count, (size_t)h->nlevels * B * 4));
  SD_HIP_CHECK(hipMalloc(&h->d_cell_keep, std::max<size_t>(hp.plan.ncells, 1) * B * 4));
  SD_HIP_CHECK(hipMalloc(&h->d_cell_off, std::max<size_t>(hp.plan.ncells, 1) * B * 4));
  SD_HIP_CHECK(hipMalloc(&h->d_lvl_m, (size_t)h->nlevels * B * 4));
  SD_HIP_CHECK(hipMemsetAsync(h->d_blur, 0, hp.plan.pyr_frame_bytes * B + slack, h->stream));
  if (!hp.cells.empty())
    SD_HIP_CHECK(hipMemcpyAsync(h->d_cells, hp.cells.data(), hp.cells.size() * sizeof(CellGeom), hipMemcpyHostToDevice, h->stream));
  if (!hp.blur_tiles.empty())
    SD_HIP_CHECK(hipMemcpyAsync(h->d_tiles, hp.blur_tiles.data(), hp.blur_tiles.size() * sizeof(BlurTile), hipMemcpyHostToDevice, h->stream));
  if (!hp.coef.empty())
    SD_HIP_CHECK(hipMemcpyAsync(h->d_coef, hp.coef.data(), hp.coef.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  SD_HIP_CHECK(hipMemcpyAsync(h->d_plan, &hp.plan, sizeof(OrbPlan), hipMemcpyHostToDevice, h->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  return SD_OK;
}

static int ensure_geometry(sd_orb* h, int w, int hgt) {
  if (h->have_geom && h->cur_w == w && h->cur_h == hgt) return SD_OK;
  SD_REQUIRE(w <= h->max_w && hgt <= h->max_h, SD_ERR_CAPACITY, "frame larger than the handle's max_w x max_h");
  const char* why = "";
  HostPlan np = h->hp;   // carries the size-independent tables (scale factors, quotas, umax, pattern)
  if (!plan_geometry(h->nfeatures, h->nlevels, h->thFAST, w, hgt, np, &why)) {
    set_error(std::string("unsupported geometry: ") + why);
    return SD_ERR_INVALID_ARG;   // the handle keeps its previous, still consistent geometry
  }
  SD_REQUIRE(np.max_cells_per_level <= SEL_MAX_CELLS && (size_t)np.plan.ncells * 16 <= 60 * 1024, SD_ERR_INVALID_ARG,
             "too many grid cells (k_select_quota keeps four ints per cell in LDS)");
  SD_HIP_CHECK(hipSetDevice(h->device));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  { int rcw = wait_trackers(h); if (rcw != SD_OK) return rcw; }
  drop_graphs(h);
  h->select_recorded = false;
  h->have_geom = false;
  h->cur_w = h->cur_h = 0;
  h->last_frames = 0;
  free_geom(h);
  const int rc = build_geometry(h, np);
  if (rc != SD_OK) {
    free_geom(h);   // partial allocations; have_geom stays false: the next call rebuilds everything
    return rc;
  }
  h->hp = std::move(np);
  h->have_geom = true;
  h->cur_w = w;
  h->cur_h = hgt;
  return SD_OK;
}

// The kernels of one extraction on the handle's three streams (main: pyramid chain, select, descriptors; fast: FAST per
// level; aux: blur), forked and joined with events only -- also what gets captured into a hipGraph.
static int pipeline_body(sd_orb* h, const uint8_t* d_imgs, int n, int stride, size_t frame_stride, bool prof, hipEvent_t* ev,
                         bool frames_ready, bool capturing) {
  const HostPlan& hp = h->hp;
  const OrbPlan& P = hp.plan;
  hipStream_t s = h->stream;
  bool fast_started = false;
  // The previous call's selection (main stream) read d_cand / d_cell_count: FAST must not overwrite them before it is done.
  // r3: FAST of level 0 reads the caller's frames and needs nothing else of THIS call, so with frames that are already on the
  // device (frames_ready: the device-input entry point) it is ordered behind the previous SELECTION only (ev_select_done)
  // and runs beside the previous batch's descriptor kernel, a gather-latency-bound kernel that leaves the vector ALUs idle.
  // Otherwise (host frames copied on this stream, a captured graph, option off, first call) behind everything queued so far.
  const bool fast_early = frames_ready && !capturing && h->select_recorded && opt(OPT_FAST0_EARLY) != 0;
  if (fast_early) {
    SD_HIP_CHECK(hipStreamWaitEvent(h->fast_stream, h->ev_select_done, 0));
    if (h->user_fence_live) SD_HIP_CHECK(hipStreamWaitEvent(h->fast_stream, h->ev_user_fence[1], 0));   // the upload of these frames
  }
  SD_HIP_CHECK(hipEventRecord(h->ev_body_start, s));
  if (!fast_early) SD_HIP_CHECK(hipStreamWaitEvent(h->fast_stream, h->ev_body_start, 0));
  // r3: with two output sets (a tracker is attached) the pyramid of this call goes into the set the previous call does NOT read,
  // so the resize chain, too, may run beside the previous call's descriptor kernel -- on the auxiliary stream in front of the
  // blur, which follows it anyway (a FIFTH stream lands on the hardware queue of the FAST stream and the two serialise: HIP
  // spreads a process's streams over four hardware queues) -- behind the previous selection, the upload of the frames and the
  // tracker's last read of that set (launch_pipeline); the FAST launches of levels 1... then follow level 0 without waiting for
  // the descriptors to end (they were idle for 0.6 ms per step).  OFF by default: bit-exact (tests run it), but the full step with
  // the PnP solve loses 8 % (172 k vs 194.6 k frames/s, three alternating runs) -- the tracking kernels of the previous batch
  // then share the machine with two FAST launches instead of one; with TrackWithMotionModel it is even (187 k both).
  const bool pyr_early = fast_early && h->nsets == 2 && opt(OPT_PYR_EARLY) != 0;
  hipStream_t ps = pyr_early ? h->aux_stream : s;
  if (prof && !pyr_early) SD_HIP_CHECK(hipEventRecord(ev[0], s));
  if (pyr_early) {
    SD_HIP_CHECK(hipStreamWaitEvent(ps, h->ev_select_done, 0));
    if (h->user_fence_live) SD_HIP_CHECK(hipStreamWaitEvent(ps, h->ev_user_fence[1], 0));
    if (prof) SD_HIP_CHECK(hipEventRecord(ev[0], ps));   // (the pyramid stage is timed on the stream it runs on)
  }
  const bool src_aligned = (((uintptr_t)d_imgs | (uintptr_t)stride | (uintptr_t)frame_stride) & 3) == 0;
  // FAST of level 0 reads the frames themselves when they are 4-byte aligned: it starts at once, beside the resize chain
  const bool fast0_direct = src_aligned && P.lv[0].ncells > 0 && opt(OPT_FAST0_FROM_FRAMES) != 0;
  const int ring_slot = h->ev_calls % sd_orb::kRing;
  int nfp = 0;   // FAST launches timed so far
  auto fast_pair = [&](bool begin) {
    if (prof && h->evf_ready && nfp < sd_orb::kFastPairs) (void)hipEventRecord(h->evf[ring_slot][2 * nfp + (begin ? 0 : 1)], h->fast_stream);
    if (!begin) nfp++;
  };
  if (fast0_direct) {
    if (prof) SD_HIP_CHECK(hipEventRecord(ev[8], h->fast_stream));
    fast_started = true;
    FastSrc fs0;
    memset(&fs0, 0, sizeof(fs0));
    fs0.pstride[0] = stride;
    fast_pair(true);
    hipLaunchKernelGGL(k_fast_cells, dim3(P.lv[0].ncells, n), dim3(256), hp.fast_lds_level[0], h->fast_stream, h->d_cells, d_imgs,
                       frame_stride, fs0, 0, h->d_cand, P.cand_per_frame, h->d_cell_count, P.ncells, P.lv[0].cell0, P.thFAST);
    fast_pair(false);
  }
  FastSrc fsrc;
  memset(&fsrc, 0, sizeof(fsrc));
  for (int l = 0; l < P.nlevels; l++) {
    fsrc.off[l] = P.lv[l].off;
    fsrc.pstride[l] = P.lv[l].pstride;
  }
  const int merge_from = hp.fast_merge_from;   // orb_plan.cpp
  for (int l = 0; l < P.nlevels; l++) {
    const LevelGeom& L = P.lv[l];
    const LevelGeom& S = P.lv[l > 0 ? l - 1 : 0];
    const bool split = L.w >= 16 && L.fast_resize != 0 && (l > 0 || src_aligned);
    if (!split) {   // generic single-pass kernel (exact-2x INTER_AREA levels, odd source alignment, tiny levels)
      dim3 grid((L.pstride + 255) / 256, (L.prows + 4 * PYR_ROWS - 1) / (4 * PYR_ROWS), n), block(64, 4, 1);
      hipLaunchKernelGGL(k_pyr_level, grid, block, 0, ps, L, S, (size_t)P.pyr_frame_bytes, l, d_imgs, stride, frame_stride, h->d_pyr);
    } else {
    auto magic = [](unsigned d) { return (unsigned)(0xFFFFFFFFull / d) + 1u; };   // e / d == umulhi(e, magic) for e < 2^31 / d
    const int G = (L.w + 2 * SD_EDGE + 3) / 4;   // 4-pixel groups of a padded row
    const bool mirror = L.h >= 48;   // border rows as second stores of their source rows (single reflections)
    const int Hh = ((mirror ? L.h : L.prows) + PYR_RPT - 1) / PYR_RPT;   // (padded) rows per row-slot of a thread
    const unsigned n_resize = (unsigned)(((size_t)Hh * G + 255) / 256);
    hipLaunchKernelGGL(k_pyr_split, dim3(n_resize, n), dim3(256), 0, ps, L, S, (size_t)P.pyr_frame_bytes, l, h->d_coef, d_imgs,
                       stride, frame_stride, h->d_pyr, G, magic((unsigned)G), Hh, mirror ? 1 : 0);
    }
    // FAST of this level starts now, on its own stream
    // levels >= merge_from share ONE launch after the last level is complete (cells of consecutive levels are contiguous)
    const bool merged = l >= merge_from;
    int ncl = L.ncells;
    size_t lds = hp.fast_lds_level[l];
    if (merged) {
      if (l != P.nlevels - 1) continue;
      ncl = 0;
      lds = 0;
      for (int m = merge_from; m < P.nlevels; m++) {
        ncl += P.lv[m].ncells;
        lds = std::max(lds, hp.fast_lds_level[m]);
      }
    }
    const int first = merged ? merge_from : l;
    if (ncl > 0 && !(l == 0 && fast0_direct)) {
      SD_HIP_CHECK(hipEventRecord(h->ev_level[l], ps));
      SD_HIP_CHECK(hipStreamWaitEvent(h->fast_stream, h->ev_level[l], 0));
      if (prof && !fast_started) SD_HIP_CHECK(hipEventRecord(ev[8], h->fast_stream));
      fast_started = true;
      fast_pair(true);
      hipLaunchKernelGGL(k_fast_cells, dim3(ncl, n), dim3(256), lds, h->fast_stream, h->d_cells, (const uint8_t*)h->d_pyr,
                         (size_t)P.pyr_frame_bytes, fsrc, SD_EDGE, h->d_cand, P.cand_per_frame, h->d_cell_count, P.ncells, P.lv[first].cell0,
                         P.thFAST);
      fast_pair(false);
    }
  }
  if (prof) SD_HIP_CHECK(hipEventRecord(ev[1], ps));
  // blur on the auxiliary stream, beside FAST + selection (d_blur exists once: behind the previous call's descriptors)
  SD_HIP_CHECK(hipEventRecord(h->ev_pyr_done, ps));
  SD_HIP_CHECK(hipStreamWaitEvent(h->aux_stream, h->ev_pyr_done, 0));
  if (pyr_early) SD_HIP_CHECK(hipStreamWaitEvent(h->aux_stream, h->ev_body_start, 0));
  if (prof) SD_HIP_CHECK(hipEventRecord(ev[3], h->aux_stream));
  hipLaunchKernelGGL(k_blur, dim3((unsigned)hp.blur_tiles.size(), n), dim3(256), 0, h->aux_stream, h->d_plan, h->d_tiles, h->d_pyr,
                     h->d_blur, h->d_sel_count);
  if (prof) SD_HIP_CHECK(hipEventRecord(ev[6], h->aux_stream));
  SD_HIP_CHECK(hipEventRecord(h->ev_blur_done, h->aux_stream));
  if (prof) {
    if (!fast_started) SD_HIP_CHECK(hipEventRecord(ev[8], h->fast_stream));
    SD_HIP_CHECK(hipEventRecord(ev[2], h->fast_stream));
    h->evf_n[ring_slot] = h->evf_ready ? nfp : 0;
  }
  // ev_fast_done stands for "pyramid AND FAST complete" (a tracker's ImageAlign waits for it alone, track.hip wait_inputs): a
  // level without grid cells launches no FAST, so the FAST stream has not necessarily waited for that level's resize
  // (few features: no cells at all on the merged small levels) -- order it behind the whole pyramid explicitly
  SD_HIP_CHECK(hipStreamWaitEvent(h->fast_stream, h->ev_pyr_done, 0));
  SD_HIP_CHECK(hipEventRecord(h->ev_fast_done, h->fast_stream));
  SD_HIP_CHECK(hipStreamWaitEvent(s, h->ev_fast_done, 0));
  if (prof) SD_HIP_CHECK(hipEventRecord(ev[9], s));
  const int sel_cap = hp.max_cell_pixels > 12000 ? 1024 : 512;
  if (P.ncells > 0) {
    hipLaunchKernelGGL(k_select_quota, dim3(n), dim3(64), (size_t)P.ncells * 16, s, h->d_plan, h->d_cells, h->d_cell_count, h->d_cell_keep,
                       h->d_cell_off, h->d_lvl_m);
    // small buffer = 2 x sel_cap entries (5 KB at VGA: 32 one-wave workgroups per CU; 4 x was 0.385 ms, 2 x and 1 x 0.275 ms);
    // big buffer = SEL_WAVES x sel_cap
    int small_cap = 2 * sel_cap, big_cap = SEL_WAVES * sel_cap;
    if (const int e = opt(OPT_SELECT_SMALL_CAP)) small_cap = std::max(1, std::min(e, small_cap));   // tests: force the other paths
    if (const int e = opt(OPT_SELECT_BIG_CAP)) big_cap = std::max(small_cap, std::min(e, big_cap));
    hipLaunchKernelGGL(k_select_cells, dim3(P.ncells, n), dim3(64), (size_t)small_cap * 4 + 2 * WAVE_SEL_CAP * 2, s, h->d_plan, h->d_cells,
                       h->d_cand, h->d_cell_count, h->d_cell_keep, h->d_cell_off, h->d_scratch, small_cap);
    hipLaunchKernelGGL(k_select_bigcells, dim3(P.nlevels, n), dim3(64), (size_t)big_cap * 4 + 2 * WAVE_SEL_CAP * 2, s, h->d_plan, h->d_cells,
                       h->d_cand, h->d_cell_count, h->d_cell_keep, h->d_cell_off, h->d_scratch, small_cap, big_cap);
  }
  // (also without any grid cell -- nfeatures so small that every level's levelCols is 0: the per-level counts the descriptor
  // kernel reads must still be written, as zeros)
  hipLaunchKernelGGL(k_select_final, dim3(P.nlevels, n), dim3(64), 0, s, h->d_plan, h->d_lvl_m, h->d_scratch, h->d_sel, h->d_sel_count);
  if (!capturing) {   // d_cand / d_cell_count are free again: the next call's level-0 FAST may start (see the top of this function)
    SD_HIP_CHECK(hipEventRecord(h->ev_select_done, s));
    h->select_recorded = true;
  }
  if (prof) SD_HIP_CHECK(hipEventRecord(ev[7], s));
  SD_HIP_CHECK(hipStreamWaitEvent(s, h->ev_blur_done, 0));
  if (prof) SD_HIP_CHECK(hipEventRecord(ev[4], s));
  const int cap = std::max(P.nsel, 1);
  {
    const int bpf = (cap + 4 * DESC_KPW - 1) / (4 * DESC_KPW);   // 4 waves x DESC_KPW keypoints per workgroup
    hipLaunchKernelGGL(k_orient_desc, dim3((unsigned)(((n + 7) / 8) * 8 * bpf)), dim3(256), 0, s, h->d_plan, h->d_pyr, h->d_blur,
                       h->d_sel, h->d_sel_count, h->d_kps, h->d_desc, h->d_nout, cap, n, bpf);
  }
  if (h->have_dist) {   // mvKeysUn != mvKeys only when k1 != 0 (src/Frame.cc:336-339)
    DistParams D;
    D.fx = h->dist_K[0]; D.fy = h->dist_K[1]; D.cx = h->dist_K[2]; D.cy = h->dist_K[3];
    for (int i = 0; i < 12; i++) D.k[i] = i < 5 ? (double)h->dist[i] : 0.0;
    hipLaunchKernelGGL(k_undistort, dim3((cap + 255) / 256, n), dim3(256), 0, s, h->d_kps, h->d_kps_un, h->d_nout, cap, D);
  }
  if (prof) SD_HIP_CHECK(hipEventRecord(ev[5], s));
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

static void drop_graphs(sd_orb* h) {
  for (auto& g : h->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    g = sd_orb::GraphEntry();
  }
}

static int launch_pipeline(sd_orb* h, const uint8_t* d_imgs, int n, int stride, size_t frame_stride, bool frames_ready) {
  hipStream_t s = h->stream;
  const bool prof = h->profiling;
  hipEvent_t* ev = h->ev[h->ev_calls % sd_orb::kRing];
  // next output set; a tracker may still be reading its previous contents on another stream
  select_set(h, (h->set + 1) % h->nsets);
  if (h->set_busy[h->set]) {
    SD_HIP_CHECK(hipStreamWaitEvent(s, h->ev_set_free[h->set], 0));
    if (h->nsets == 2) SD_HIP_CHECK(hipStreamWaitEvent(h->aux_stream, h->ev_set_free[h->set], 0));   // (an early pyramid, pipeline_body)
    h->set_busy[h->set] = false;
  }
  // hipGraph replay is opt-in (option "extract.use_graph"): measured on ROCm 7.2 / MI355X the single-frame call takes 0.43 ms through
  // the graph against 0.37 ms with direct launches (tools/exp_pcie.py), so direct launches stay the default
  const bool use_graph = opt(OPT_USE_GRAPH) != 0;
  int rc = SD_OK;
  if (prof || !use_graph) {
    rc = pipeline_body(h, d_imgs, n, stride, frame_stride, prof, ev, frames_ready, false);
  } else {
    sd_orb::GraphEntry* ge = nullptr;
    float dv[9] = {h->dist_K[0], h->dist_K[1], h->dist_K[2], h->dist_K[3], h->dist[0], h->dist[1], h->dist[2], h->dist[3], h->dist[4]};
    for (auto& g : h->graphs)
      if (g.exec && g.imgs == d_imgs && g.n == n && g.stride == stride && g.frame_stride == frame_stride && g.set == h->set &&
          g.dist == h->have_dist && (!g.dist || memcmp(g.distv, dv, sizeof(dv)) == 0))
        ge = &g;
    if (!ge) {
      hipGraph_t graph = nullptr;
      SD_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      rc = pipeline_body(h, d_imgs, n, stride, frame_stride, false, ev, false, true);
      h->select_recorded = false;   // the graph's selection is not an event record a later call could wait for
      hipError_t e = hipStreamEndCapture(s, &graph);
      if (rc == SD_OK && e != hipSuccess) {
        set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        rc = SD_ERR_HIP;
      }
      if (rc == SD_OK) {
        sd_orb::GraphEntry& slot = h->graphs[h->graph_next++ % 4];
        if (slot.exec) (void)hipGraphExecDestroy(slot.exec);
        slot = sd_orb::GraphEntry();
        e = hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
          set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
          rc = SD_ERR_HIP;
        } else {
          slot.imgs = d_imgs; slot.n = n; slot.stride = stride; slot.frame_stride = frame_stride; slot.set = h->set;
          slot.dist = h->have_dist;
          memcpy(slot.distv, dv, sizeof(dv));
          ge = &slot;
        }
      }
      if (graph) (void)hipGraphDestroy(graph);
    }
    if (rc == SD_OK) SD_HIP_CHECK(hipGraphLaunch(ge->exec, s));
  }
  if (rc != SD_OK) return rc;
  // inside a captured graph ev_pyr_done is a graph node, not an event record a later hipStreamWaitEvent could see
  h->pyr_event_live = prof || !use_graph;
  SD_HIP_CHECK(hipEventRecord(h->ev_extract_done, s));
  h->extract_recorded = true;
  h->extract_serial++;
  if (prof) h->ev_calls++;
  h->last_frames = n;
  return SD_OK;
}


namespace sd {
// Second output set for a handle whose frames a tracker consumes on its own stream.
int orb_enable_double_buffer(sd_orb* h) {
  if (h->nsets == 2) return SD_OK;
  SD_HIP_CHECK(hipSetDevice(h->device));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  int nsel = 0;
  for (int q : h->hp.quota) nsel += q;
  const size_t cap = std::max(nsel, 1), B = h->max_batch;
  SD_HIP_CHECK(hipMalloc(&h->kps_set[1], cap * B * sizeof(sd_keypoint)));
  SD_HIP_CHECK(hipMalloc(&h->kps_un_set[1], cap * B * sizeof(sd_keypoint)));
  SD_HIP_CHECK(hipMalloc(&h->desc_set[1], cap * B * 32));
  SD_HIP_CHECK(hipMalloc(&h->nout_set[1], B * 4));
  SD_HIP_CHECK(hipMemset(h->nout_set[1], 0, B * 4));
  if (h->have_geom) {
    const size_t bytes = h->hp.plan.pyr_frame_bytes * B + 4096;
    SD_HIP_CHECK(hipMalloc(&h->pyr_set[1], bytes));
    SD_HIP_CHECK(hipMemset(h->pyr_set[1], 0, bytes));
  }
  h->nsets = 2;
  return SD_OK;
}
}  // namespace sd

namespace sd {
int read_sel_prof(unsigned long long* out64, int reset) {   // out64[8 * i + 7] = FAST phase i (the other slots: 0)
#ifdef SD_PNP_PROF
  SD_HIP_CHECK(hipDeviceSynchronize());
  for (int i = 0; i < 64; i++) out64[i] = 0;   // the selection kernels carry no timers since the split (a rocprofv3 trace times them)
  unsigned long long f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  {
    std::vector<unsigned long long> wg((size_t)FPROF_FRAMES * FPROF_CELLS * 8);
    SD_HIP_CHECK(hipMemcpyFromSymbol(wg.data(), HIP_SYMBOL(g_fast_prof_wg), wg.size() * 8));
    for (size_t i = 0; i < wg.size(); i++) f[i & 7] += wg[i];
    if (reset) {
      std::fill(wg.begin(), wg.end(), 0ull);
      SD_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fast_prof_wg), wg.data(), wg.size() * 8));
    }
  }
  for (int l = 0; l < 8; l++) out64[l * 8 + 7] = f[l];   // slot 7 of every level row carries FAST phase l
  return SD_OK;
#else
  set_error("library built without -DSD_PNP_PROF");
  return SD_ERR_INVALID_ARG;
#endif
}
}  // namespace sd

extern "C" {

const char* sd_last_error(void) { return g_err.c_str(); }
const char* sd_version(void) { return "sdslam_hip 0.1 (gfx950)"; }

int sd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sd_orb_create(int nfeatures, float scale_factor, int nlevels, int th_fast, int max_w, int max_h, int max_batch,
                  int device, sd_orb** out) {
  SD_REQUIRE(out != nullptr, SD_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  SD_REQUIRE(nfeatures > 0 && nlevels >= 1 && nlevels <= SD_MAX_LEVELS && scale_factor > 1.0f, SD_ERR_INVALID_ARG,
             "bad extractor parameters");
  SD_REQUIRE(max_w >= 1 && max_h >= 1 && max_w <= SD_MAX_DIM && max_h <= SD_MAX_DIM && max_batch >= 1, SD_ERR_INVALID_ARG,
             "bad capacity parameters");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device visible: the HIP path is the only implementation (no CPU fallback)");
    return SD_ERR_NO_DEVICE;
  }
  SD_REQUIRE(device >= 0 && device < ndev, SD_ERR_INVALID_ARG, "device index out of range");
  SD_HIP_CHECK(hipSetDevice(device));
  sd_orb* h = new sd_orb();
  h->nfeatures = nfeatures;
  h->scaleFactor = scale_factor;
  h->nlevels = nlevels;
  h->thFAST = th_fast;
  h->max_w = max_w;
  h->max_h = max_h;
  h->max_batch = max_batch;
  h->device = device;
  plan_tables(nfeatures, scale_factor, nlevels, h->hp);
  int nsel = 0;
  for (int q : h->hp.quota) nsel += q;
  const size_t cap = std::max(nsel, 1);
  hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&h->d_plan, sizeof(OrbPlan));
  if (e == hipSuccess) e = hipMalloc(&h->d_img, (size_t)max_w * max_h * max_batch);
  if (e == hipSuccess) e = hipMalloc(&h->kps_set[0], cap * max_batch * sizeof(sd_keypoint));
  if (e == hipSuccess) e = hipMalloc(&h->kps_un_set[0], cap * max_batch * sizeof(sd_keypoint));
  if (e == hipSuccess) e = hipMalloc(&h->desc_set[0], cap * max_batch * 32);
  if (e == hipSuccess) e = hipMalloc(&h->nout_set[0], (size_t)max_batch * 4);
  if (e == hipSuccess) e = hipMemset(h->nout_set[0], 0, (size_t)max_batch * 4);
  for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&h->ev_set_free[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_extract_done, hipEventDisableTiming);
  if (e == hipSuccess) select_set(h, 0);
  for (int r = 0; r < sd_orb::kRing && e == hipSuccess; r++)
    for (int i = 0; i < 10 && e == hipSuccess; i++) e = hipEventCreate(&h->ev[r][i]);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->fast_stream, hipStreamNonBlocking);
  for (int i = 0; i < SD_MAX_LEVELS && e == hipSuccess; i++) e = hipEventCreateWithFlags(&h->ev_level[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fast_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_select_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_body_start, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_pyr_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_blur_done, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_error(std::string("sd_orb_create: ") + hipGetErrorString(e));
    sd_orb_destroy(h);
    return SD_ERR_HIP;
  }
  h->stream = h->own_stream;
  *out = h;
  return SD_OK;
}

void sd_orb_destroy(sd_orb* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  (void)wait_trackers(h);
  drop_graphs(h);
  free_geom(h);
  void* ptrs[] = {h->d_plan, h->d_img, h->kps_set[0], h->kps_set[1], h->kps_un_set[0], h->kps_un_set[1], h->desc_set[0],
                  h->desc_set[1], h->nout_set[0], h->nout_set[1]};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (int i = 0; i < 2; i++)
    if (h->ev_set_free[i]) (void)hipEventDestroy(h->ev_set_free[i]);
  if (h->ev_extract_done) (void)hipEventDestroy(h->ev_extract_done);
  for (int r = 0; r < sd_orb::kRing; r++) {
    for (int i = 0; i < 10; i++)
      if (h->ev[r][i]) (void)hipEventDestroy(h->ev[r][i]);
    for (int i = 0; i < 2 * sd_orb::kFastPairs; i++)
      if (h->evf[r][i]) (void)hipEventDestroy(h->evf[r][i]);
  }
  if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
  if (h->fast_stream) { (void)hipStreamSynchronize(h->fast_stream); (void)hipStreamDestroy(h->fast_stream); }
  for (int i = 0; i < SD_MAX_LEVELS; i++)
    if (h->ev_level[i]) (void)hipEventDestroy(h->ev_level[i]);
  if (h->ev_fast_done) (void)hipEventDestroy(h->ev_fast_done);
  if (h->ev_select_done) (void)hipEventDestroy(h->ev_select_done);
  if (h->ev_body_start) (void)hipEventDestroy(h->ev_body_start);
  if (h->ev_pyr_done) (void)hipEventDestroy(h->ev_pyr_done);
  if (h->ev_blur_done) (void)hipEventDestroy(h->ev_blur_done);
  for (int i = 0; i < 2; i++)
    if (h->ev_user_fence[i]) (void)hipEventDestroy(h->ev_user_fence[i]);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

int sd_orb_levels(const sd_orb* h) { return h ? h->nlevels : 0; }

// Host-only (no GPU needed): geometry the extractor would use for a w x h frame.
int sd_orb_plan_info(int nfeatures, float scale_factor, int nlevels, int th_fast, int w, int hgt, int32_t* level_info,
                     int32_t* cell_zones, int cell_cap, int32_t* n_cells, uint64_t* bytes_per_frame) {
  SD_REQUIRE(level_info && n_cells, SD_ERR_INVALID_ARG, "NULL argument");
  SD_REQUIRE(nfeatures > 0 && nlevels >= 1 && nlevels <= SD_MAX_LEVELS && scale_factor > 1.0f, SD_ERR_INVALID_ARG,
             "bad extractor parameters");
  HostPlan hp;
  plan_tables(nfeatures, scale_factor, nlevels, hp);
  const char* why = "";
  if (!plan_geometry(nfeatures, nlevels, th_fast, w, hgt, hp, &why)) {
    set_error(std::string("unsupported geometry: ") + why);
    return SD_ERR_INVALID_ARG;
  }
  for (int l = 0; l < nlevels; l++) {
    const LevelGeom& L = hp.plan.lv[l];
    int32_t* o = level_info + 8 * l;
    o[0] = L.w; o[1] = L.h; o[2] = L.quota; o[3] = L.cols; o[4] = L.rows; o[5] = L.cellW; o[6] = L.cellH; o[7] = L.nfeaturesCell;
  }
  *n_cells = hp.plan.ncells;
  if (cell_zones) {
    SD_REQUIRE(cell_cap >= hp.plan.ncells, SD_ERR_CAPACITY, "cell_cap too small");
    for (int c = 0; c < hp.plan.ncells; c++) {
      const CellGeom& C = hp.cells[c];
      int32_t* o = cell_zones + 6 * c;
      o[0] = C.level; o[1] = C.zx0; o[2] = C.zy0; o[3] = C.zw; o[4] = C.zh; o[5] = C.evaluated;
    }
  }
  if (bytes_per_frame) *bytes_per_frame = hp.plan.pyr_frame_bytes * 2 + (uint64_t)hp.plan.cand_per_frame * 8;
  return SD_OK;
}

int sd_orb_scale_tables(const sd_orb* h, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  for (int i = 0; i < h->nlevels; i++) {
    if (sf) sf[i] = h->hp.sf[i];
    if (inv_sf) inv_sf[i] = h->hp.inv_sf[i];
    if (sigma2) sigma2[i] = h->hp.sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = h->hp.inv_sigma2[i];
  }
  return SD_OK;
}

int sd_orb_features_per_level(const sd_orb* h, int32_t* quota) {
  SD_REQUIRE(h && quota, SD_ERR_INVALID_ARG, "NULL argument");
  for (int i = 0; i < h->nlevels; i++) quota[i] = h->hp.quota[i];
  return SD_OK;
}

int sd_orb_extract_batch_device(sd_orb* h, const void* d_imgs, int n_frames, int w, int hgt, int stride,
                                size_t frame_stride) {
  SD_REQUIRE(h && d_imgs, SD_ERR_INVALID_ARG, "NULL argument");
  SD_REQUIRE(n_frames >= 1 && n_frames <= h->max_batch, SD_ERR_CAPACITY, "n_frames exceeds max_batch");
  SD_REQUIRE(w >= 1 && hgt >= 1 && stride >= w && frame_stride >= (size_t)stride * (hgt - 1) + w, SD_ERR_INVALID_ARG,
             "bad image shape/stride");
  SD_HIP_CHECK(hipSetDevice(h->device));
  int rc = ensure_geometry(h, w, hgt);
  if (rc != SD_OK) return rc;
  // frames_ready: the caller's frames are complete on the device (or ordered by sd_orb_stream_fence); the host-input entry
  // points copy them on the extraction stream and call launch_pipeline themselves
  return launch_pipeline(h, (const uint8_t*)d_imgs, n_frames, stride, frame_stride, !h->staging_input);
}

int sd_orb_download(sd_orb* h, int frame0, int n_frames, sd_keypoint* kps_out, uint8_t* desc_out, int cap_per_frame,
                    int32_t* n_out) {
  SD_REQUIRE(h && n_out, SD_ERR_INVALID_ARG, "NULL argument");
  SD_REQUIRE(frame0 >= 0 && n_frames >= 1 && frame0 + n_frames <= h->last_frames, SD_ERR_INVALID_ARG,
             "frame range outside the last batch");
  SD_HIP_CHECK(hipSetDevice(h->device));
  const int cap = std::max(h->hp.plan.nsel, 1);
  SD_HIP_CHECK(hipMemcpyAsync(n_out, h->d_nout + frame0, (size_t)n_frames * 4, hipMemcpyDeviceToHost, h->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int f = 0; f < n_frames; f++)
    SD_REQUIRE(n_out[f] <= cap_per_frame || (!kps_out && !desc_out), SD_ERR_CAPACITY, "cap_per_frame smaller than keypoint count");
  if (cap_per_frame == cap && n_frames > 8) {
    // same row pitch on both sides: two bulk copies instead of 2 x n_frames small ones (entries beyond n_out[f] are
    // whatever the device rows hold; callers must not read them)
    if (kps_out)
      SD_HIP_CHECK(hipMemcpyAsync(kps_out, h->d_kps + (size_t)frame0 * cap, (size_t)n_frames * cap * sizeof(sd_keypoint),
                                  hipMemcpyDeviceToHost, h->stream));
    if (desc_out)
      SD_HIP_CHECK(hipMemcpyAsync(desc_out, h->d_desc + (size_t)frame0 * cap * 32, (size_t)n_frames * cap * 32, hipMemcpyDeviceToHost,
                                  h->stream));
    SD_HIP_CHECK(hipStreamSynchronize(h->stream));
    return SD_OK;
  }
  for (int f = 0; f < n_frames; f++) {
    int n = n_out[f];
    if (n <= 0) continue;
    if (kps_out)
      SD_HIP_CHECK(hipMemcpyAsync(kps_out + (size_t)f * cap_per_frame, h->d_kps + (size_t)(frame0 + f) * cap,
                                  (size_t)n * sizeof(sd_keypoint), hipMemcpyDeviceToHost, h->stream));
    if (desc_out)
      SD_HIP_CHECK(hipMemcpyAsync(desc_out + (size_t)f * cap_per_frame * 32, h->d_desc + (size_t)(frame0 + f) * cap * 32,
                                  (size_t)n * 32, hipMemcpyDeviceToHost, h->stream));
  }
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  return SD_OK;
}

int sd_orb_extract_batch(sd_orb* h, const uint8_t* imgs, int n_frames, int w, int hgt, int stride, size_t frame_stride,
                         sd_keypoint* kps_out, uint8_t* desc_out, int cap_per_frame, int32_t* n_out) {
  SD_REQUIRE(h && n_out, SD_ERR_INVALID_ARG, "NULL argument");
  if (w <= 0 || hgt <= 0 || !imgs) {   // _image.empty(): return silently (src/ORBextractor.cc:622-623)
    for (int f = 0; f < n_frames; f++) n_out[f] = 0;
    return SD_OK;
  }
  SD_REQUIRE(n_frames >= 1 && n_frames <= h->max_batch, SD_ERR_CAPACITY, "n_frames exceeds max_batch");
  SD_REQUIRE(w <= h->max_w && hgt <= h->max_h, SD_ERR_CAPACITY, "frame larger than the handle's max_w x max_h");
  SD_REQUIRE(stride >= w, SD_ERR_INVALID_ARG, "stride < width");
  SD_HIP_CHECK(hipSetDevice(h->device));
  // pack rows tightly into the staging buffer (one copy when the frames already are tightly packed)
  if (stride == w && (n_frames == 1 || frame_stride == (size_t)w * hgt)) {
    SD_HIP_CHECK(hipMemcpyAsync(h->d_img, imgs, (size_t)n_frames * w * hgt, hipMemcpyHostToDevice, h->stream));
  } else {
    for (int f = 0; f < n_frames; f++)
      SD_HIP_CHECK(hipMemcpy2DAsync(h->d_img + (size_t)f * w * hgt, w, imgs + (size_t)f * frame_stride, stride, w,
                                    (size_t)hgt, hipMemcpyHostToDevice, h->stream));
  }
  h->staging_input = true;    // the frames reach d_img by a copy queued on the extraction stream just above
  int rc = sd_orb_extract_batch_device(h, h->d_img, n_frames, w, hgt, w, (size_t)w * hgt);
  h->staging_input = false;
  if (rc != SD_OK) return rc;
  return sd_orb_download(h, 0, n_frames, kps_out, desc_out, cap_per_frame, n_out);
}

int sd_orb_extract(sd_orb* h, const uint8_t* img, int w, int hgt, int stride, sd_keypoint* kps_out, uint8_t* desc_out,
                   int cap, int* n_out) {
  SD_REQUIRE(n_out, SD_ERR_INVALID_ARG, "n_out is NULL");
  int32_t n = 0;
  int rc = sd_orb_extract_batch(h, img, 1, w, hgt, stride, (size_t)stride * (hgt > 0 ? hgt : 0), kps_out, desc_out, cap, &n);
  *n_out = n;
  return rc;
}

int sd_orb_set_distortion(sd_orb* h, float fx, float fy, float cx, float cy, float k1, float k2, float p1, float p2, float k3) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  SD_REQUIRE(fx > 0 && fy > 0, SD_ERR_INVALID_ARG, "bad camera matrix");
  h->dist_K[0] = fx; h->dist_K[1] = fy; h->dist_K[2] = cx; h->dist_K[3] = cy;
  h->dist[0] = k1; h->dist[1] = k2; h->dist[2] = p1; h->dist[3] = p2; h->dist[4] = k3;
  h->have_dist = (k1 != 0.0f);   // mDistCoef.at<float>(0) == 0.0 -> mvKeysUn = mvKeys
  return SD_OK;
}

int sd_orb_download_undistorted(sd_orb* h, int frame0, int n_frames, sd_keypoint* kps_un_out, int cap_per_frame) {
  SD_REQUIRE(h && kps_un_out, SD_ERR_INVALID_ARG, "NULL argument");
  SD_REQUIRE(frame0 >= 0 && n_frames >= 1 && frame0 + n_frames <= h->last_frames, SD_ERR_INVALID_ARG, "frame range outside the last batch");
  SD_HIP_CHECK(hipSetDevice(h->device));
  const int cap = std::max(h->hp.plan.nsel, 1);
  std::vector<int32_t> n(n_frames);
  SD_HIP_CHECK(hipMemcpyAsync(n.data(), h->d_nout + frame0, (size_t)n_frames * 4, hipMemcpyDeviceToHost, h->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  const sd_keypoint* src = h->have_dist ? h->d_kps_un : h->d_kps;
  for (int f = 0; f < n_frames; f++) {
    SD_REQUIRE(n[f] <= cap_per_frame, SD_ERR_CAPACITY, "cap_per_frame smaller than keypoint count");
    if (n[f] > 0)
      SD_HIP_CHECK(hipMemcpyAsync(kps_un_out + (size_t)f * cap_per_frame, src + (size_t)(frame0 + f) * cap, (size_t)n[f] * sizeof(sd_keypoint),
                                  hipMemcpyDeviceToHost, h->stream));
  }
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  return SD_OK;
}

int sd_orb_level_info(const sd_orb* h, int level, int* w, int* hgt) {
  SD_REQUIRE(h && h->have_geom && level >= 0 && level < h->nlevels, SD_ERR_INVALID_ARG, "no geometry / bad level");
  if (w) *w = h->hp.plan.lv[level].w;
  if (hgt) *hgt = h->hp.plan.lv[level].h;
  return SD_OK;
}

static int copy_level(sd_orb* h, const uint8_t* base, int frame, int level, int padded, uint8_t* out, int out_stride) {
  SD_REQUIRE(h && out && h->have_geom && level >= 0 && level < h->nlevels && frame >= 0 && frame < h->last_frames,
             SD_ERR_INVALID_ARG, "bad frame/level");
  const LevelGeom& L = h->hp.plan.lv[level];
  SD_HIP_CHECK(hipSetDevice(h->device));
  const uint8_t* src = base + (size_t)frame * h->hp.plan.pyr_frame_bytes + L.off;
  int wc = L.w, hc = L.h;
  if (padded) {
    wc += 2 * SD_EDGE;
    hc += 2 * SD_EDGE;
  } else {
    src += (size_t)SD_EDGE * L.pstride + SD_EDGE;
  }
  SD_REQUIRE(out_stride >= wc, SD_ERR_INVALID_ARG, "out_stride too small");
  SD_HIP_CHECK(hipMemcpy2DAsync(out, out_stride, src, L.pstride, wc, hc, hipMemcpyDeviceToHost, h->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  return SD_OK;
}

int sd_orb_level_copy(sd_orb* h, int frame, int level, int padded, uint8_t* out, int out_stride) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  return copy_level(h, h->d_pyr, frame, level, padded, out, out_stride);
}

int sd_orb_debug_blurred(sd_orb* h, int frame, int level, uint8_t* out, int out_stride) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  return copy_level(h, h->d_blur + SD_BLUR_SHIFT, frame, level, 0, out, out_stride);
}

int sd_orb_debug_cell_counts(sd_orb* h, int frame, int level, int32_t* out, int cap, int* n_cells) {
  SD_REQUIRE(h && out && n_cells && h->have_geom && level >= 0 && level < h->nlevels && frame >= 0 && frame < h->last_frames,
             SD_ERR_INVALID_ARG, "bad frame/level");
  const LevelGeom& L = h->hp.plan.lv[level];
  *n_cells = L.ncells;
  SD_REQUIRE(cap >= L.ncells, SD_ERR_CAPACITY, "cap too small");
  if (L.ncells == 0) return SD_OK;
  SD_HIP_CHECK(hipSetDevice(h->device));
  SD_HIP_CHECK(hipMemcpyAsync(out, h->d_cell_count + (size_t)frame * h->hp.plan.ncells + L.cell0, (size_t)L.ncells * 4,
                              hipMemcpyDeviceToHost, h->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  return SD_OK;
}

int sd_orb_debug_level_keys(sd_orb* h, int frame, int level, uint32_t* keys_out, int cap, int* n) {
  SD_REQUIRE(h && keys_out && n && h->have_geom && level >= 0 && level < h->nlevels && frame >= 0 && frame < h->last_frames,
             SD_ERR_INVALID_ARG, "bad frame/level");
  const LevelGeom& L = h->hp.plan.lv[level];
  SD_HIP_CHECK(hipSetDevice(h->device));
  int32_t cnt = 0;
  SD_HIP_CHECK(hipMemcpyAsync(&cnt, h->d_sel_count + (size_t)frame * h->nlevels + level, 4, hipMemcpyDeviceToHost, h->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  *n = cnt;
  SD_REQUIRE(cap >= cnt, SD_ERR_CAPACITY, "cap too small");
  if (cnt > 0) {
    SD_HIP_CHECK(hipMemcpyAsync(keys_out, h->d_sel + (size_t)frame * h->hp.plan.nsel + L.sel_off, (size_t)cnt * 4,
                                hipMemcpyDeviceToHost, h->stream));
    SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  }
  return SD_OK;
}

int sd_orb_set_stream(sd_orb* h, void* hip_stream) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  SD_HIP_CHECK(hipSetDevice(h->device));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return SD_OK;
}

// Ordering against a caller's HIP stream (an upload stream that fills the frames of the next batch while this one is being
// processed).  direction 0: `hip_stream` waits for everything queued on the extraction stream so far (the frames of the
// extractions queued so far have been consumed when it proceeds); 1: the extraction stream -- and with it the FAST stream,
// which is ordered behind it at the start of every extraction -- waits for everything queued on `hip_stream` so far.
int sd_orb_stream_fence(sd_orb* h, void* hip_stream, int direction) {
  SD_REQUIRE(h && (direction == 0 || direction == 1), SD_ERR_INVALID_ARG, "bad arguments");
  SD_HIP_CHECK(hipSetDevice(h->device));
  hipStream_t ext = (hipStream_t)hip_stream;
  for (int i = 0; i < 2; i++)
    if (!h->ev_user_fence[i]) SD_HIP_CHECK(hipEventCreateWithFlags(&h->ev_user_fence[i], hipEventDisableTiming));
  if (direction == 0) {
    SD_HIP_CHECK(hipEventRecord(h->ev_user_fence[0], h->stream));
    SD_HIP_CHECK(hipStreamWaitEvent(ext, h->ev_user_fence[0], 0));
  } else {
    SD_HIP_CHECK(hipEventRecord(h->ev_user_fence[1], ext));
    SD_HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_user_fence[1], 0));
    h->user_fence_live = true;   // an early level-0 FAST launch (pipeline_body) waits for it as well
  }
  return SD_OK;
}

int sd_orb_sync(sd_orb* h) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  SD_HIP_CHECK(hipSetDevice(h->device));
  SD_HIP_CHECK(hipStreamSynchronize(h->aux_stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  return SD_OK;
}

int sd_orb_set_profiling(sd_orb* h, int on) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  if (on && !h->evf_ready) {
    SD_HIP_CHECK(hipSetDevice(h->device));
    for (int r = 0; r < sd_orb::kRing; r++)
      for (int i = 0; i < 2 * sd_orb::kFastPairs; i++) SD_HIP_CHECK(hipEventCreate(&h->evf[r][i]));
    h->evf_ready = true;
  }
  h->profiling = on != 0;
  h->ev_calls = 0;
  return SD_OK;
}

int sd_orb_num_stages(void) { return ST_COUNT; }
const char* sd_orb_stage_name(int stage) { return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : ""; }

int sd_orb_stage_ms(sd_orb* h, float* ms_out, int cap) {
  SD_REQUIRE(h && ms_out && cap >= ST_COUNT, SD_ERR_INVALID_ARG, "bad arguments");
  SD_REQUIRE(h->profiling && h->ev_calls > 0, SD_ERR_INVALID_ARG, "profiling is off or no call recorded");
  SD_HIP_CHECK(hipSetDevice(h->device));
  SD_HIP_CHECK(hipStreamSynchronize(h->stream));
  const int n = std::min(h->ev_calls, (int)sd_orb::kRing);
  for (int i = 0; i < ST_COUNT; i++) ms_out[i] = 0.f;
  static const int kBegin[ST_COUNT] = {0, 8, 9, 3, 4}, kEnd[ST_COUNT] = {1, 2, 7, 6, 5};
  SD_HIP_CHECK(hipStreamSynchronize(h->aux_stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->fast_stream));
  for (int r = 0; r < n; r++) {
    const int slot = (h->ev_calls - 1 - r) % sd_orb::kRing;
    for (int i = 0; i < ST_COUNT; i++) {
      float ms = 0;
      if (i == ST_FAST && h->evf_n[slot] > 0) {   // sum of the k_fast_cells launches (what a kernel trace of the same run adds up to)
        for (int k = 0; k < h->evf_n[slot]; k++) {
          float one = 0;
          SD_HIP_CHECK(hipEventElapsedTime(&one, h->evf[slot][2 * k], h->evf[slot][2 * k + 1]));
          ms += one;
        }
      } else {
        SD_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[slot][kBegin[i]], h->ev[slot][kEnd[i]]));
      }
      ms_out[i] += ms / n;
    }
  }
  return SD_OK;
}

int sd_orb_stage_bytes(const sd_orb* h, double* bytes_out, int cap) {
  SD_REQUIRE(h && bytes_out && cap >= ST_COUNT && h->have_geom, SD_ERR_INVALID_ARG, "bad arguments / no geometry yet");
  for (int i = 0; i < ST_COUNT; i++) bytes_out[i] = h->hp.stage_bytes[i];
  return SD_OK;
}

int sd_dev_alloc(size_t bytes, void** out) {
  SD_REQUIRE(out, SD_ERR_INVALID_ARG, "out is NULL");
  SD_HIP_CHECK(hipMalloc(out, bytes));
  return SD_OK;
}
int sd_dev_free(void* p) {
  SD_HIP_CHECK(hipFree(p));
  return SD_OK;
}
// Page-locked host memory for frames / results: with it the host-buffer entry points copy at the PCIe rate
// (pageable buffers go through the driver's staging copies, measured 6 GB/s on the test box).
int sd_host_alloc(size_t bytes, void** out) {
  SD_REQUIRE(out, SD_ERR_INVALID_ARG, "out is NULL");
  SD_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return SD_OK;
}
int sd_host_free(void* p) {
  SD_HIP_CHECK(hipHostFree(p));
  return SD_OK;
}
int sd_dev_upload(void* dst, const void* src, size_t bytes) {
  SD_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return SD_OK;
}
int sd_dev_download(void* dst, const void* src, size_t bytes) {
  SD_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return SD_OK;
}

// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1459-1473): 256-bit Hamming distance
int sd_hamming(const uint8_t* a32, const uint8_t* b32) {
  int dist = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t x, y;
    memcpy(&x, a32 + 8 * i, 8);
    memcpy(&y, b32 + 8 * i, 8);
    dist += __builtin_popcountll(x ^ y);
  }
  return dist;
}

}  // extern "C"
