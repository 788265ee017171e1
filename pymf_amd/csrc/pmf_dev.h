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

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Counter-based U[0,1) generator for the synthetic fills (splitmix64 finaliser).
__device__ __forceinline__ float u01_from(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
