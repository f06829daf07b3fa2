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
