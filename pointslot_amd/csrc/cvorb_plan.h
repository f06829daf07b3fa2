// Device-side descriptors of the object-feature detector (cvorb_kernels.hip / cvorb_host.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CV_BORDER 23          // max(edgeThreshold 19, descPatchSize ceil(15 sqrt 2) = 22, HARRIS_BLOCK_SIZE / 2) + 1 (orb.cpp)
#define CV_MAX_LEVELS 8

struct CvLevelDev {
  int32_t w, h, stride;       // level size; padded plane row stride
  float scale;                // layerScale[level] = (float)pow(scaleFactor, level)
  uint8_t* pad;               // (h + 46) x stride, origin = padded pixel (-23, -23)
  uint8_t* blur;              // padded like `pad`: the blurred level inside, the unblurred border outside (OpenCV blurs in place)
  uint8_t* mask;              // h x w, tight; nullptr when the call has no mask
  uint8_t* score;             // h x w FAST score plane
  int32_t* rowcnt; int32_t* rowoff;
  float4* cand;               // keypoints after the mask / border filters, raster order: (x, y, FAST score, Harris response)
};
struct CvPlanDev {
  CvLevelDev lv[CV_MAX_LEVELS];
  int32_t nlevels;
  int32_t umax[17];
};
struct CvSel { int32_t x, y, level; float response; };   // a selected keypoint (level coordinates)
struct ps_keypoint_pod { float x, y, size, angle, response; int32_t octave, class_id; };

// ---- batched, device-resident form (cvb_* kernels): nimg images with object masks, work restricted to the 32 x 32 tiles of every
// level's padded plane that can influence a keypoint under the mask ----
#define CVB_TILE 32
#define CVB_CAND_CAP 2048      // FAST keypoints under the mask per (image, level); more is reported, never truncated silently
#define CVB_MAX_TILES 4096     // tiles of one level's padded plane (worklist entries carry the tile in 12 bits)
struct CvbLevel {
  int32_t w, h, stride;        // level size, padded-plane row stride
  int32_t tw, th, tile_off;    // tiles of the padded plane; offset of the level in the per-image tile arrays
  int32_t cw, ch, cell_off;    // 8 x 8 cells of the padded plane; offset of the level in the per-image cell arrays
  int32_t quota;               // nfeaturesPerLevel
  float scale;
  size_t o_pad, o_blur, o_mask, o_score;   // byte offsets inside one image's arena
  const int4* xtab; const int4* ytab;      // INTER_LINEAR_EXACT tables from level l - 1 (nullptr at level 0)
  double sx, sy;               // the tables' source step per destination pixel (1 / ((double)dsize / ssize)): cvb_resize evaluates the entries itself
  int32_t dminx, dmaxx, dminy, dmaxy;      // destination indices below dmin take source 0, from dmax on source ssize - 1 (weight 1)
};
struct CvbPlan {
  CvbLevel lv[CV_MAX_LEVELS];
  int32_t nlevels, edge, fast_th, w0, h0;
  int32_t ocw, och;            // level-0 occupancy cells (8 x 8 image pixels)
  int32_t cell_total, cell_max; // cells of all levels / of the largest level
  int32_t tile_total;
  int32_t umax[17];
  int32_t kq[4];
  uint8_t* arena; size_t arena_pitch;      // per image: the planes of all levels
  const uint8_t* imgs; int32_t img_stride; size_t img_pitch;       // the call's inputs (level 0's source, level 0's mask)
  const uint8_t* masks; int32_t mask_stride; size_t mask_pitch;
  uint8_t* occ;                // [nimg][och][ocw]: the object mask has a non-zero pixel in that cell
  uint8_t* kpmap;              // [nimg][cell_total]: cells of every level in which a keypoint is possible
  uint32_t* wl; int32_t* wl_count; int32_t wl_cap;   // [3 (planes, FAST, blur)][CV_MAX_LEVELS][wl_cap] entries (image << 12 | tile)
  float4* cand; int32_t* ncand;                      // [nimg][nlevels][CVB_CAND_CAP], [nimg][nlevels]
  CvSel* sel; int32_t* nsel;                         // [nimg][nlevels][CVB_CAND_CAP], [nimg][nlevels]
  ps_keypoint_pod* kps; uint8_t* desc; int32_t* count; int32_t* overflow;   // [nimg][ocap], [nimg][ocap][32], [nimg], [nimg]
  int32_t ocap;
  uint8_t* dump;               // 1 KB nobody reads: where a tile kernel's lanes that have no output row send their (unconditional) store
};
