// pmf_dev.h -- device-side helpers shared by the gfx950 kernels of libpymf_hip.
//
// MFMA used throughout: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate, exact
// fmaf-chain numerics).  Operand maps (one VGPR each):
//   A: lane l holds A[i = l & 15][k = l >> 4]
//   B: lane l holds B[k = l >> 4][j = l & 15]
//   C/D (4 VGPRs): reg r of lane l is D[row = 4 * (l >> 4) + r][col = l & 15]
// A contraction's k order is free as long as A and B agree, which lets every
// fragment be fetched with one 16-byte LDS read that feeds 4 consecutive MFMAs
// (k-step e of a group takes element e of both operands' float4).
#pragma once
#include <hip/hip_runtime.h>

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: launch helpers remember it per device.
constexpr int PMF_MAX_DEVICES = 64;
static inline int pmf_current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= PMF_MAX_DEVICES) d = 0;
  return d;
}
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PMF_EPS_DEN 1e-9f   // added to denominators only (pymf/nmf.py:124,130; snmf.py:89)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// a / b for a >= 0 and b a positive NORMAL float (b = den + 1e-9 here): v_rcp_f32 (1 ulp)
// refined by one Newton step on the quotient, q = q0 + r * fma(-b, q0, a).  The result is
// the correctly rounded quotient except for rare last-bit ties -- the same class of
// difference as a different BLAS summation order -- at 4 VALU ops instead of the 10 of the
// IEEE div expansion (which only adds scaling for denormal / overflowing operands).
__device__ __forceinline__ float pmf_div(float a, float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float q0 = a * r;
  const float e = fmaf(-b, q0, a);
  return fmaf(e, r, q0);
}

// LDS image of a [rows][64] f32 panel: 256-byte rows (exactly the 64 banks),
// 16-byte chunk c of row r stored at chunk (c ^ (r & 15)).  A fragment read --
// 16 lanes on 16 different rows, 4 k-groups on chunks 4t..4t+3 -- is then
// conflict-free for ds_read_b128, and a row read (16 lanes on 16 consecutive
// floats of one row) stays inside one aligned 64-byte group.
__device__ __forceinline__ int swz_off(int row, int chunk) {   // float index
  return row * 64 + ((chunk ^ (row & 15)) << 2);
}

__device__ __forceinline__ f32x4 lds_read4(const float* base, int row, int chunk) {
  return *reinterpret_cast<const f32x4*>(base + swz_off(row, chunk));
}

__device__ __forceinline__ void lds_write4(float* base, int row, int chunk, f32x4 v) {
  *reinterpret_cast<f32x4*>(base + swz_off(row, chunk)) = v;
}

// 1 / d in float64 without the IEEE division sequence: v_rcp_f64 (about 26 good bits) and two Newton
// steps; the result is within an ulp or two of the correctly rounded quotient.  For reciprocals on a
// serial critical path (pivots).
__device__ __forceinline__ double pmf_rcp_f64(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = fma(fma(-d, x, 1.0), x, x);
  x = fma(fma(-d, x, 1.0), x, x);
  return x;
}

// float64 MFMA (v_mfma_f64_16x16x4_f64) and a wave-uniform lane read of a double: shared by pmf_inv.h and the NMFALS kernels
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 mfma_f64(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double readlane_f64(double v, int srclane) {   // srclane wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, returned in every lane.  Butterfly inside each row of 16 lanes on DPP
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: register-to-register, a few cycles
// each -- __shfl_xor goes through the LDS crossbar, ~100 cycles a step), then the four row sums are
// read out as scalars.  Fixed order: deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_mov_f64<0xB1>(v);      // lane ^ 1
  v += dpp_mov_f64<0x4E>(v);      // lane ^ 2
  v += dpp_mov_f64<0x141>(v);     // i <-> 7 - i   (the other quad of the 8)
  v += dpp_mov_f64<0x140>(v);     // i <-> 15 - i  (the other half of the row)
  double r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    r[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * q),
                            __builtin_amdgcn_readlane(__double2loint(v), 16 * q));
  return (r[0] + r[1]) + (r[2] + r[3]);
}

// Counter-based U[0,1) generator for the synthetic fills (splitmix64 finaliser).
__device__ __forceinline__ float u01_from(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
