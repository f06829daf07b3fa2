// Geometry plan of the ORB pipeline for one image size; built on the host (orb_host.hip), passed
// by value to every kernel.  All offsets are bytes (or elements where noted) inside one image's
// arena; image b lives at arena_base + b * arena_bytes.
#pragma once
#include <stdint.h>

#define PS_ORB_MAX_LEVELS 8
#define PS_EDGE 19          // EDGE_THRESHOLD, /root/reference/src/ORBextractor.cc:74
#define PS_MINB 16          // minBorderX/Y = EDGE_THRESHOLD - 3, ORBextractor.cc:773-774
#define PS_QT_NCAP 2048     // largest node capacity of the quadtree kernel (per-level quota + 4 must fit; 149 KB of LDS)
#define PS_FAST_WIN 66      // max FAST cell window edge (cell + 6): a cell is at most 60 px (ceil(width / floor(width / 30)))

struct OrbLevel {
  int32_t w, h;             // level image size (ORBextractor.cc:1112)
  int32_t stride;           // padded plane row stride in bytes (multiple of 64)
  int32_t bstride;          // blurred plane row stride in bytes (multiple of 64)
  uint32_t plane_off;       // padded plane origin (its pixel (-19,-19))
  uint32_t blur_off;        // blurred plane origin (pixel (0,0))
  int32_t n_cols, n_rows;   // FAST cell grid (ORBextractor.cc:784-787)
  int32_t w_cell, h_cell;
  int32_t cell_base;        // index of the level's first cell in the per-image cell arrays
  int32_t cell_cap;         // candidate slots per cell
  uint32_t cand_off;        // element offset (u32) of the level's candidate slots
  uint32_t key_off;         // element offset (u32) of the level's quadtree key scratch
  int32_t key_cap;
  int32_t quota;            // mnFeaturesPerLevel
  int32_t n_ini;            // quadtree: initial node count, hX (ORBextractor.cc:543-545)
  float h_x;
  int32_t sel_off;          // element offset of the level's selected-keypoint slots
  int32_t sel_cap;
  float scale;              // mvScaleFactor[level]
  float inv_scale;          // mvInvScaleFactor[level]
  float kp_size;            // (float)(int)(31 * scale), ORBextractor.cc:839
  uint32_t xtab_off, ytab_off;  // element offsets into the resize tables (level >= 1)
  int32_t blur_blk_base;        // first block of this level in the single orb_blur launch
  int32_t border_blk_base;      // same for orb_border
};

struct OrbPlan {
  OrbLevel lv[PS_ORB_MAX_LEVELS];
  int32_t nlevels;
  int32_t img_w, img_h;
  int32_t n_cells;          // cells per image, all levels
  int32_t sel_total;        // selected-keypoint slots per image, all levels
  int32_t kp_cap;           // output keypoints per image
  int32_t ini_th, min_th;
  int32_t blur_blocks;      // grid.x of orb_blur
  int32_t border_blocks;    // grid.x of orb_border
  uint64_t arena_bytes;     // per image
  uint64_t cellcnt_off;     // int32[n_cells]
  uint64_t cand_base;       // u32 slots
  uint64_t key_base;        // u32 x 2 per key: [xy][node_score]
  uint64_t sel_base;        // u32[sel_total]
  uint64_t selcnt_off;      // int32[PS_ORB_MAX_LEVELS]
  uint64_t ncand_off;       // int32[PS_ORB_MAX_LEVELS] (diagnostic: candidates per level)
  uint32_t celltab_off;     // in int4 units into the plan's table buffer: u32[n_cells] = level | cell column << 4 | cell row << 18 (orb_fast_cells)
};

// one stereo pair for the stereo matcher (orb_stereo.hip): raw device pointers so that the left and the right image may
// live in the same batch (images 2k / 2k+1 of one handle) or in two handles (the reference's two ORBextractor objects)
struct StPair {
  const uint8_t* arena_l; const uint8_t* arena_r;
  const void* kps_l; const uint8_t* desc_l; const int32_t* cnt_l;
  const void* kps_r; const uint8_t* desc_r; const int32_t* cnt_r;
  float* u_right; float* depth; int32_t* sad; int32_t* kept;
  uint8_t* scratch;         // PS_ST_SCRATCH bytes: row-bucket table of the right keypoints (st_bucket -> st_match)
};
// stereo matcher scratch per pair: bucket offsets (8-row buckets of the right keypoints' y), bucketed indices, compact right
// keypoint records {x, minr | maxr << 12 | octave << 24}
#define PS_ST_MAXB 512
#define PS_ST_CAP 4096
#define PS_ST_OFF_BYTES ((PS_ST_MAXB + 4) * 4)
#define PS_ST_SCRATCH (PS_ST_OFF_BYTES + PS_ST_CAP * 4 + PS_ST_CAP * 8)
