// Shared pieces of the split-operand (bf16 x 3) kernels: edge_fwd_x3.hip, edge_bwd_x3.hip.
// An fp32 value x is split EXACTLY into three bf16 pieces, x = h + m + l:  h = rne_bf16(x), m = rne_bf16(x - h),
// l = rne_bf16(x - h - m) (both subtractions are exact in fp32; tests/test_host.py restates this on the host).
// A product a*b is taken as  al*bh + ah*bl + am*bm + am*bh + ah*bm + ah*bh  (smallest first), every piece
// product exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; the dropped al*bm + am*bl + al*bl is
// below 2^-23 |a||b|.
#pragma once
#include <hip/hip_runtime.h>

namespace ng {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// v_cvt_pk_bf16_f32 (round to nearest even) through the compiler's own conversion, NOT inline asm: the hazard
// recognizer does not apply the MFMA-related wait states (XDL write -> VALU, SrcC read -> VALU write, ...) to
// instructions hidden inside an asm statement, and once the scheduler interleaved the asm conversions with an MFMA
// chain the edge forward kernel returned run-to-run different e (round 2: ~16 % of the edges off by up to 6e-6 in the
// training variant; tools/dbg_det.py).  The vector fptrunc lowers to the same single instruction on gfx950.
typedef float f32x2_cvt __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_cvt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cvt{lo, hi}, bf16x2_cvt));
}

// (x0, x1) -> packed bf16 pieces; piece p of x0 in the low half, of x1 in the high half
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(s0, s1);
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the six piece products of (A triple) x (B triple), smallest terms first
__device__ __forceinline__ f32x16 mma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
  acc = mfma_bf16(a[2], b[0], acc);
  acc = mfma_bf16(a[0], b[2], acc);
  acc = mfma_bf16(a[1], b[1], acc);
  acc = mfma_bf16(a[1], b[0], acc);
  acc = mfma_bf16(a[0], b[1], acc);
  acc = mfma_bf16(a[0], b[0], acc);
  return acc;
}

// the same six piece products for TWO accumulators that share the B triple (a0 / a1: two A triples); the two chains
// alternate, so consecutive MFMAs never depend on each other (a chain of dependent 32x32x16 MFMAs issues slower than
// the pipe's 32-cycle cadence: SQ_WAIT_INST_ANY 41-49 % in the GEMM that used mma6 per accumulator)
__device__ __forceinline__ void mma6_2a(const u32x4 (&a0)[3], const u32x4 (&a1)[3], const u32x4 (&b)[3], f32x16& acc0,
                                        f32x16& acc1) {
  acc0 = mfma_bf16(a0[2], b[0], acc0); acc1 = mfma_bf16(a1[2], b[0], acc1);
  acc0 = mfma_bf16(a0[0], b[2], acc0); acc1 = mfma_bf16(a1[0], b[2], acc1);
  acc0 = mfma_bf16(a0[1], b[1], acc0); acc1 = mfma_bf16(a1[1], b[1], acc1);
  acc0 = mfma_bf16(a0[1], b[0], acc0); acc1 = mfma_bf16(a1[1], b[0], acc1);
  acc0 = mfma_bf16(a0[0], b[1], acc0); acc1 = mfma_bf16(a1[0], b[1], acc1);
  acc0 = mfma_bf16(a0[0], b[0], acc0); acc1 = mfma_bf16(a1[0], b[0], acc1);
}

}  // namespace ng
