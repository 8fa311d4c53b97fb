// Shared pieces of the two-piece fp16 split-operand kernels (edge_fwd_h2.hip, edge_bwd_h2.hip, gemm_h2 paths).
// An fp32 value x is written as x = h + l + r with h = rne_f16(x), l = rne_f16(x - h) (the subtraction is exact in
// fp32), |r| <= 2^-22 |x| (rms 2^-23 |x|: two 11-bit significands with a signed residual cover 22-23 bits; host
// restatement in tests/test_host.py), and a
// product a*b is taken as  al*bh + ah*bl + ah*bh  (smallest first), every piece product exact in the fp32 accumulator
// of v_mfma_f32_32x32x16_f16 (11 x 11 significand bits).  Dropped: al*bl <= 2^-22 |a||b| and the residuals — the size
// of an fp32 rounding of the product.  Three matrix instructions per fp32 multiply; the exact three-piece bf16 split
// these kernels used until late round 2 needed six, and two conversions / subtractions per element instead of one.
// Range: fp16 pieces need |x| < 65504 (activations: unscaled; gradients: scaled by a power of two per launch, see
// edge_bwd_h2.hip); below 2^-14 the l piece is a subnormal the matrix pipe honours (tools/ubench/mfma_f16.hip), so the
// absolute representation error never exceeds max(2^-25, 2^-22 |x|).
#pragma once
#include <hip/hip_runtime.h>

namespace ng {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_cvt __attribute__((ext_vector_type(2)));
typedef float f32x2_cvt __attribute__((ext_vector_type(2)));

// two floats -> packed fp16, round to nearest even, through the compiler's own vector conversion and NOT inline asm:
// LLVM's hazard recognizer does not apply the MFMA-related wait states (XDL write -> VALU, SrcC read -> VALU write, ...)
// to instructions hidden inside an asm statement; when the scheduler interleaved asm conversions with an MFMA chain
// (round 2, bf16 version of these kernels) ~16 % of the edges came out different from run to run by up to 6e-6
// (tests/test_gpu_determinism.py).
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cvt{lo, hi}, f16x2_cvt));
}

// -1 as an fp16 value the optimizer cannot see through (an SGPR; the empty asm emits no instruction)
__device__ __forceinline__ _Float16 h2_minus_one() {
  int b = 0xBC00;
  asm volatile("" : "+s"(b));
  return __builtin_bit_cast(_Float16, (short)b);
}

// (x0, x1) -> packed fp16 pieces; piece of x0 in the low half, of x1 in the high half.
// The residual x - (float)h is taken as fma((float)h, (float)(-1 as fp16), x): with BOTH multiplicands extended from fp16
// the compiler selects v_fma_mix_f32 (fp16 sources by op_sel, fp32 addend) — one instruction per element instead of
// v_cvt_f32_f16 + v_sub_f32; the difference is exact in fp32, so the single rounding changes nothing (same bits).  Four
// instructions per pair instead of six.  v_fma_mix_f32 issues beside the 16-bit MFMA like any plain VALU instruction
// (tools/ubench/mfma_fill.hip `fmamix`: 34.1 ... 36.1 cycles per MFMA with 1 ... 5 of them per gap), unlike the packed fp32
// ops its VOP3P encoding shares.  (A visible -1.0f is folded back into a subtraction, an opaque FLOAT -1 makes the
// compiler convert and use v_pk_fma_f32: round 2.)
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l) {
  h = cvt_pk_f16(x0, x1);
  const f16x2_cvt hv = __builtin_bit_cast(f16x2_cvt, h);
  const float m1 = (float)h2_minus_one();
  const float r0 = __builtin_fmaf((float)hv[0], m1, x0), r1 = __builtin_fmaf((float)hv[1], m1, x1);
  l = cvt_pk_f16(r0, r1);
}

__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// the three piece products of (A pair) x (B pair), smallest terms first; index 0 = h, 1 = l
__device__ __forceinline__ f32x16 mma3(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16 acc) {
  acc = mfma_f16(a[1], b[0], acc);
  acc = mfma_f16(a[0], b[1], acc);
  acc = mfma_f16(a[0], b[0], acc);
  return acc;
}

// the same for TWO accumulators that share the B pair; the chains alternate so that consecutive MFMAs are independent
__device__ __forceinline__ void mma3_2a(const u32x4 (&a0)[2], const u32x4 (&a1)[2], const u32x4 (&b)[2], f32x16& acc0,
                                        f32x16& acc1) {
  acc0 = mfma_f16(a0[1], b[0], acc0); acc1 = mfma_f16(a1[1], b[0], acc1);
  acc0 = mfma_f16(a0[0], b[1], acc0); acc1 = mfma_f16(a1[0], b[1], acc1);
  acc0 = mfma_f16(a0[0], b[0], acc0); acc1 = mfma_f16(a1[0], b[0], acc1);
}


// ---- LDS-DMA through an asm statement (round 5).  For a `__builtin_amdgcn_raw_ptr_buffer_load_lds` in flight hipcc puts
// `s_waitcnt vmcnt(N)` in front of every later LDS read it cannot prove disjoint from the copy's destination — the round
// trip that was meant to run beside a matrix interval comes back at that interval's first operand read
// (profiles/r05e_gw_ab.txt, r05f_dma_ab.txt).  The statement writes M0 itself and restores it; completion is the KERNEL's
// business: an explicit `s_waitcnt vmcnt(N)` and a barrier in front of the first read of the copied data.  hipcc's waits
// for its own loads stay safe (vector memory operations complete in order: they can only wait for more than they need).
// -DNG_DMA_BUILTIN keeps the builtin (A/B builds).
typedef int dma_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_i4 dma_rsrc(const void* p, unsigned bytes) {
  const uint64_t b = (uint64_t)p;
  return dma_i4{__builtin_amdgcn_readfirstlane((int)(unsigned)b), __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu)),
                __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
// one 1-KB piece: lane l's 16 bytes from (voff + soff) land at lds_dst + 16 l
__device__ __forceinline__ void lds_dma16(dma_i4 rs, const void* lds_dst, int voff, int soff) {
#ifdef NG_DMA_BUILTIN
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)(unsigned)rs[1] << 32) | (unsigned)rs[0]), 0, rs[2], rs[3]);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#else
  unsigned keep;
  const int dst = __builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_dst);
  const int so = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rs), "s"(so) : "memory");
#endif
}

}  // namespace ng
