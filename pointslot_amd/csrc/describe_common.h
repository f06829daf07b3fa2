// Shared by the two descriptor kernels (orb_describe: ORBextractor.cc:77-147; cvb_describe: the cv::ORB stand-in of Frame.cc:2623-2665):
// the rBRIEF pattern in the kernels' lane layout, the intensity-centroid item table, sin / cos on [0, 2 pi], DPP row sums.
// Included inside each translation unit's anonymous namespace.
#pragma once

constexpr int8_t k_pattern[1024] = {
#include "orb_pattern.inc"
};
// the pattern for orb_describe's lane layout: lane l of a 16-lane group owns tests 16 l .. 16 l + 15 (two descriptor bytes);
// word [jj][l][t] = test 16 l + 4 jj + t as four bytes (x0, y0, x1, y1).  The workgroup copies it to LDS, where lane l
// reads four 16-byte rows (consecutive lanes, consecutive rows: no bank conflicts).
// r05: the four coordinates are stored as FP8 (OCP E4M3: every integer up to 15 in magnitude is exact; the pattern's are -13 .. 12), so
// that ONE v_cvt_pk_f32_fp8 turns a point's (x, y) into the packed float pair the steering arithmetic takes - it was a sign-extending
// byte conversion per coordinate.
struct PatTab { uint32_t w[4][16][4]; };
constexpr uint32_t fp8_e4m3_of_small_int(int v) {
  const uint32_t sgn = v < 0 ? 0x80u : 0u;
  const int a = v < 0 ? -v : v;
  if (a == 0) return 0u;
  int e = 0;
  while ((a >> (e + 1)) != 0) e++;                 // floor(log2 a), a <= 15
  const uint32_t m = ((uint32_t)a << 3 >> e) & 7u; // the three bits behind the leading one (a has at most four significant bits)
  return sgn | ((uint32_t)(e + 7) << 3) | m;
}
constexpr PatTab make_pattab() {
  PatTab t{};
  for (int l = 0; l < 16; l++)
    for (int j = 0; j < 16; j++) {
      const int8_t* p = k_pattern + 4 * (16 * l + j);
      t.w[j >> 2][l][j & 3] = fp8_e4m3_of_small_int(p[0]) | (fp8_e4m3_of_small_int(p[1]) << 8) | (fp8_e4m3_of_small_int(p[2]) << 16) | (fp8_e4m3_of_small_int(p[3]) << 24);
    }
  return t;
}
constexpr bool pattern_fits_fp8() {
  for (int i = 0; i < 1024; i++) if (k_pattern[i] > 15 || k_pattern[i] < -15) return false;
  return true;
}
static_assert(pattern_fits_fp8(), "the pattern table stores coordinates as E4M3: integers beyond 15 are not exact");
__constant__ PatTab c_pattab = make_pattab();


// cos and sin of x in [0, 2 pi] in double precision (error ~1e-16, i.e. the same float after narrowing as a correctly rounded
// libm except for one argument in ~1e8): two-term Cody-Waite reduction by pi/2 and the classic degree-13/12 kernels on
// [-pi/4, pi/4], evaluated with explicit FMAs.  A third of the instructions of the general-purpose library routines, which carry
// a Payne-Hanek path and double-double arithmetic for arguments this kernel never sees.
__device__ __forceinline__ void sincos_0_2pi(double x, double& sn, double& cs, const double* C) {
  const double kd = __builtin_rint(x * C[0]);                             // x * 2 / pi
  const int k = (int)kd;
  double r = __builtin_fma(-kd, C[1], x);
  r = __builtin_fma(-kd, C[2], r);
  const double z = r * r;
  double ps = C[3];
  ps = __builtin_fma(ps, z, C[4]);
  ps = __builtin_fma(ps, z, C[5]);
  ps = __builtin_fma(ps, z, C[6]);
  ps = __builtin_fma(ps, z, C[7]);
  ps = __builtin_fma(ps, z, C[8]);
  const double s0 = __builtin_fma(r * z, ps, r);
  double pc = C[9];
  pc = __builtin_fma(pc, z, C[10]);
  pc = __builtin_fma(pc, z, C[11]);
  pc = __builtin_fma(pc, z, C[12]);
  pc = __builtin_fma(pc, z, C[13]);
  pc = __builtin_fma(pc, z, C[14]);
  const double c0 = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const double ss = (k & 1) ? c0 : s0, cc = (k & 1) ? s0 : c0;
  sn = (k & 2) ? -ss : ss;
  cs = ((k + 1) & 2) ? -cc : cc;
}

// IC_Angle item table for a 16-lane group: lane l = 4 r + c takes, in step j, the eight pixels of row v = 4 j + r - 15 at columns
// u0 .. u0 + 7, u0 = 8 c - 15, of the 31 x 31 patch; the entry is the byte mask (0xFF) of the columns inside the disc
// (|u| <= umax[|v|], ORBextractor.cc:452-468; row 16 and column 16 do not exist: zero).  Copied to LDS by the workgroup.
struct IcTab { uint2 v[8][16]; };
constexpr IcTab make_ictab() {
  IcTab tb{};
  const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  for (int j = 0; j < 8; j++)
    for (int l = 0; l < 16; l++) {
      const int v = 4 * j + (l >> 2) - 15, u0 = 8 * (l & 3) - 15;
      uint32_t m[2] = {0, 0};
      if (v <= 15) {
        const int um_row = umax[v < 0 ? -v : v];
        for (int k = 0; k < 8; k++) {
          const int u = u0 + k, au = u < 0 ? -u : u;
          if (au <= um_row) m[k >> 2] |= 0xFFu << (8 * (k & 3));
        }
      }
      tb.v[j][l].x = m[0]; tb.v[j][l].y = m[1];
    }
  return tb;
}
__constant__ IcTab c_ictab = make_ictab();

// coefficients of sincos_0_2pi.  (The compiler turns them into 64-bit literals - two vector moves per use, 24 in all.  r05: behind an
// opaque pointer they become scalar loads AT their use - a scalar-memory round trip in the middle of the keypoint's dependent chain:
// orb_describe 0.139 -> 0.149 ms per 128 images; requested early and pinned, they push the kernel into scratch: 0.246 ms.)
__constant__ double c_sincos[16] = {0.63661977236758134308, 1.57079632673412561417e+00, 6.07710050650619224932e-11,
                                    1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
                                    -1.98412698295895385996e-4, 8.33333333332211858878e-3, -1.66666666666666307295e-1,
                                    -1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
                                    2.48015872888517045348e-5, -1.38888888888730564116e-3, 4.16666666666665929218e-2, 0.0};

// sum over the 16 lanes of a DPP row, left in every lane of the row: xor-1, xor-2, half-row mirror, row mirror
__device__ __forceinline__ int row_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
  return v;
}

typedef float ds_f2 __attribute__((ext_vector_type(2)));
