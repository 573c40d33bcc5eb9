// Probe: operand layout and block-scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands.
// Hypothesis H1: lane l holds A[row l&15][k = 32 (l>>4) + j] in byte j (0..31) of its 8 dwords, B[k = 32 (l>>4) + j][col l&15]
// likewise; the scale operand's selected byte is an E8M0 factor applied to that lane's 32 k-values; C/D as the bf16 form.
// Alternative H2: bytes 0..15 <-> k = 16 (l>>4) + j, bytes 16..31 <-> k = 64 + 16 (l>>4) + (j-16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c, unsigned* pk) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  c[l] = acc;
  // packing order of v_cvt_pk_fp8_f32: (1.0, 2.0) into the low word, (-0.5, 448) into the high word
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(1.0f, 2.0f, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(-0.5f, 1000.0f, w, true);
  if (l == 0) pk[0] = (unsigned)w;
}

static unsigned char to_e4m3(float v) {   // exact for the small values used here (|v| = n/8 * 2^e)
  if (v == 0.f) return 0;
  unsigned char s = v < 0 ? 0x80 : 0; v = fabsf(v);
  int e; float m = frexpf(v, &e);   // v = m 2^e, m in [0.5,1)
  int E = e - 1 + 7; float frac = m * 2.f - 1.f;   // 1.frac
  int M = (int)lrintf(frac * 8.f);
  if (E <= 0) { M = (int)lrintf(v / ldexpf(1.f, -9)); E = 0; }
  return s | (unsigned char)(E << 3) | (unsigned char)M;
}

int main() {
  float A[16][128], B[128][16];
  srand(7);
  const float vals[] = {0.f, 0.25f, -0.25f, 0.5f, -0.5f, 1.f, -1.f, 1.5f, -1.5f, 2.f, -2.f, 3.f, -3.f, 0.75f, -0.75f, 4.f};
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) A[i][kk] = vals[rand() % 16];
  for (int kk = 0; kk < 128; ++kk) for (int j = 0; j < 16; ++j) B[kk][j] = vals[rand() % 16];
  int sA[64], sB[64];
  const bool unit = getenv("UNIT_SCALES") != nullptr;
  for (int l = 0; l < 64; ++l) { sA[l] = unit ? 127 : ((127 + (l % 3) - 1) | (0x55 << 8)); sB[l] = unit ? 127 : ((127 - (l % 2)) | (0x33 << 16)); }
  for (int hyp = 1; hyp <= 2; ++hyp) {
    unsigned char ab[64][32], bb[64][32];
    auto kof = [&](int g, int j) { return hyp == 1 ? 32 * g + j : (j < 16 ? 16 * g + j : 64 + 16 * g + (j - 16)); };
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
      ab[l][j] = to_e4m3(A[l & 15][kof(l >> 4, j)]);
      bb[l][j] = to_e4m3(B[kof(l >> 4, j)][l & 15]);
    }
    i32x8 *da, *db; int *dsa, *dsb; f32x4* dc; unsigned* dpk;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 64 * 16); hipMalloc(&dpk, 4);
    hipMemcpy(da, ab, 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, bb, 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sA, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, dpk);
    float C[64][4]; unsigned pk;
    hipMemcpy(C, dc, 64 * 16, hipMemcpyDeviceToHost); hipMemcpy(&pk, dpk, 4, hipMemcpyDeviceToHost);
    // expected: scales per (row, k-block g) from lane (row, g) of A; per (col, g) from lane (col, g) of B; C layout col = l&15, row = 4 (l>>4) + r
    double worst = 0, worst_noscale = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      const int col = l & 15, row = 4 * (l >> 4) + r;
      double e = 0, e0 = 0;
      for (int g = 0; g < 4; ++g) {
        double part = 0;
        for (int j = 0; j < 32; ++j) part += (double)A[row][kof(g, j)] * B[kof(g, j)][col];
        const double fa = ldexp(1.0, (sA[row + 16 * g] & 0xff) - 127), fb = ldexp(1.0, (sB[col + 16 * g] & 0xff) - 127);
        e += part * fa * fb; e0 += part;
      }
      worst = fmax(worst, fabs(e - C[l][r])); worst_noscale = fmax(worst_noscale, fabs(e0 - C[l][r]));
    }
    printf("hypothesis H%d: max |C - expected| with per-lane block scales %.4g, ignoring scales %.4g   (C[0][0..3] = %g %g %g %g)\n", hyp, worst,
           worst_noscale, C[0][0], C[0][1], C[0][2], C[0][3]);
    if (hyp == 1) printf("cvt_pk_fp8_f32: word = 0x%08x  (expect 0x7eb04038 if a -> low byte, saturating to 448 = 0x7e)\n", pk);
  }
  return 0;
}
