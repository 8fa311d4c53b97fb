// Fused FC block (atom_feature_size == 64): all layers of nmrgnn/model.py:191-196 in one persistent
// kernel per direction.
//   forward   for l < L-1:  x_{l+1} = act(x_l W_l + b_l) + x_l        (F -> F, residual)
//             g = act(x_{L-1} W_{L-1} + b_{L-1})                        (F -> F/2)
//   backward  the same chain reversed, with the activation output recovered as s_l = x_{l+1} - x_l, so the
//             tape holds only the layer inputs x_0 .. x_{L-1} and g (no separate activation copies).
// The layer-by-layer path (tall_gemm / tall_tn) moves every activation through HBM two or three times
// per layer and is HBM-bound at ~40 % of peak; here a 64-row tile stays in LDS across the layers and
// the weights of ALL layers are register-resident MFMA fragments (16 VGPRs per layer and wave).
//
// Forward: 256 threads = 4 waves, two workgroups per CU.  Wave w owns output columns 16w..16w+15 of every
// hidden layer (v_mfma_f32_16x16x4_f32, A = W^T fragment, B = activation rows via ds_read_b128); the
// bias is the initial accumulator value; tiles ping-pong between two LDS buffers; the next tile's rows
// are requested one tile ahead.
#include <algorithm>

#include "mfma_gemm.cuh"
#include "ng_internal.h"
#include "pack_bodies.cuh"
#include "edge_fused.h"   // NG_LDS_BARRIER
#include "reduce.cuh"

namespace ng {

namespace {

constexpr int FC_F = 64;
constexpr int FC_H = 32;        // width of the last layer
constexpr int FC_TM = 64;       // rows per tile
constexpr int FC_LD = 68;       // LDS row stride
constexpr int FC_MAXL = 6;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FcPtrs {
  const float* W[FC_MAXL];
  const float* b[FC_MAXL];
  float* y[FC_MAXL];            // forward: y[l] = x_{l+1} for l < L-1 (nullptr: not kept)
};

// forward fragments:  Wf[((l*4 + ct)*4 + T)*64 + lane][u] = W_l[k = 16T + 4(lane>>4) + u][n = 16ct + (lane&15)]
// backward fragments: Wb[((l*4 + kt)*4 + T)*64 + lane][u] = W_l[k = 16kt + (lane&15)][n = 16T + 4(lane>>4) + u]
// (columns n >= n_out of the last layer are zero; packed by pack_bodies.cuh: PK_FC)
static_assert(FC_F == pk::FC_Fd && FC_H == pk::FC_Hd && FC_MAXL == 6, "pack_bodies.cuh");

struct FcFwdArgs {
  int64_t N;
  int act;
  const float* x;          // [N][64]
  const float* Wf;         // packed forward fragments
  FcPtrs p;
  float* g;                // [N][32]
  float* dummy;            // >= 64 floats: where rows >= N store
  const unsigned* WfH;     // packed piece fragments (pack_bodies.cuh: fc_h2)
  RangeGuard guard;        // word == nullptr: unguarded (fp32 body only)
  const unsigned* wflag;   // flag word of an image kept over calls (== wflag_ver: its weights left the piece range), or nullptr
  unsigned wflag_ver;
};

__device__ __forceinline__ float4 act4(int act, const f32x4& a) {
  float4 v = make_float4(a[0], a[1], a[2], a[3]);
#ifdef FC_ABL_NOACT      // timing experiment: what the activation costs
  return v;
#endif
  if (act == NG_ACT_SOFTPLUS) {
    v.x = softplus_f(v.x); v.y = softplus_f(v.y); v.z = softplus_f(v.z); v.w = softplus_f(v.w);
  } else if (act != NG_ACT_NONE) {
    v.x = act_apply(act, v.x); v.y = act_apply(act, v.y); v.z = act_apply(act, v.z); v.w = act_apply(act, v.w);
  }
  return v;
}

// one 16-row x 16-column unit pair: rows 16*rt0.. and 16*(rt0+1).., same column tile (shared fragments)
__device__ __forceinline__ void fc_unit2(const float* wf, const float* __restrict__ Xin, int rt0, int a16,
                                         int g4, const float4& bias, f32x4& acc0, f32x4& acc1) {
  const float* x0 = Xin + (16 * rt0 + a16) * FC_LD + 4 * g4;
  const float* x1 = x0 + 16 * FC_LD;
  float4 xa[4], xb[4];
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    xa[T] = *reinterpret_cast<const float4*>(x0 + 16 * T);
    xb[T] = *reinterpret_cast<const float4*>(x1 + 16 * T);
  }
  acc0 = f32x4{bias.x, bias.y, bias.z, bias.w};
  acc1 = acc0;
#ifdef FC_ABL_NOMFMA     // timing experiment: the matrix products removed (operands still read)
  acc0[0] += xa[0].x + xa[1].y + xa[2].z + xa[3].w + wf[0]; acc1[0] += xb[0].x + xb[1].y + xb[2].z + xb[3].w;
  return;
#endif
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 0], xa[T].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 0], xb[T].x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 1], xa[T].y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 1], xb[T].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 2], xa[T].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 2], xb[T].z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 3], xa[T].w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * T + 3], xb[T].w, acc1, 0, 0, 0);
  }
}

// One 64-row tile through all layers with the f32-input MFMAs; X0 holds the tile's rows (fp32) on entry.
// WREG: the fragments of every layer are register-resident (the fp32 kernel); else they are fetched per layer from the
// image (the piece kernel's repair path: a tile with an activation beyond the fp16 range is redone here).
template <int NL, bool WREG>
__device__ __forceinline__ void fc_fwd_tile_f32(const FcFwdArgs& a, int64_t row0, float* X0, float* X1,
                                                const float* __restrict__ sB, const float (*wfr)[16], int wave, int lane) {
  const int a16 = lane & 15, g4 = lane >> 4;
  float* Xin = X0;
  float* Xout = X1;
#pragma unroll
  for (int l = 0; l < NL - 1; ++l) {
    float wl[16];
    if (!WREG) {
      const float4* p = reinterpret_cast<const float4*>(a.Wf) + ((l * 4 + wave) * 4) * 64 + lane;
#pragma unroll
      for (int T = 0; T < 4; ++T) { const float4 v = p[T * 64]; wl[4 * T] = v.x; wl[4 * T + 1] = v.y; wl[4 * T + 2] = v.z; wl[4 * T + 3] = v.w; }
    }
    const float* wf = WREG ? wfr[l] : wl;
    const int col = 16 * wave + 4 * g4;
    const float4 bias = *reinterpret_cast<const float4*>(sB + l * FC_F + col);
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
      f32x4 acc0, acc1;
      fc_unit2(wf, Xin, 2 * rp, a16, g4, bias, acc0, acc1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 16 * (2 * rp + h) + a16;
        const float4 s = act4(a.act, h ? acc1 : acc0);
        const float4 xo = *reinterpret_cast<const float4*>(Xin + r * FC_LD + col);
        const float4 y = make_float4(s.x + xo.x, s.y + xo.y, s.z + xo.z, s.w + xo.w);
        *reinterpret_cast<float4*>(Xout + r * FC_LD + col) = y;
#ifndef FC_ABL_NOSTORE
        if (a.p.y[l])
#else
        if (a.p.y[l] && a.N < 0)
#endif
        {
          const int64_t row = row0 + r;
          *reinterpret_cast<float4*>(row < a.N ? a.p.y[l] + row * FC_F + col : a.dummy + col) = y;
        }
      }
    }
    NG_LDS_BARRIER();
    float* tmp = Xin; Xin = Xout; Xout = tmp;
  }
  {   // last layer: column tile wave & 1, row tiles 2*(wave >> 1), +1
    float wl[16];
    if (!WREG) {
      const float4* p = reinterpret_cast<const float4*>(a.Wf) + (((NL - 1) * 4 + (wave & 1)) * 4) * 64 + lane;
#pragma unroll
      for (int T = 0; T < 4; ++T) { const float4 v = p[T * 64]; wl[4 * T] = v.x; wl[4 * T + 1] = v.y; wl[4 * T + 2] = v.z; wl[4 * T + 3] = v.w; }
    }
    const float* wf = WREG ? wfr[NL - 1] : wl;
    const int col = 16 * (wave & 1) + 4 * g4;
    const float4 bias = *reinterpret_cast<const float4*>(sB + (NL - 1) * FC_F + col);
    f32x4 acc0, acc1;
    fc_unit2(wf, Xin, 2 * (wave >> 1), a16, g4, bias, acc0, acc1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 16 * (2 * (wave >> 1) + h) + a16;
      const float4 s = act4(a.act, h ? acc1 : acc0);
      const int64_t row = row0 + r;
      *reinterpret_cast<float4*>(row < a.N ? a.g + row * FC_H + col : a.dummy + col) = s;
    }
  }
}

// rows of tile `TILE` in the coalesced staging layout: thread -> rows (tid >> 4) + 16 i, float4 column tid & 15
#define FC_FETCH(TILE)                                                                                   \
  do {                                                                                                   \
    const int64_t r_ = (TILE) * FC_TM + (tid >> 4);                                                      \
    const float4* x4_ = reinterpret_cast<const float4*>(a.x) + (tid & 15);                               \
    px0 = x4_[(r_ < a.N ? r_ : a.N - 1) * 16];                                                           \
    px1 = x4_[(r_ + 16 < a.N ? r_ + 16 : a.N - 1) * 16];                                                \
    px2 = x4_[(r_ + 32 < a.N ? r_ + 32 : a.N - 1) * 16];                                                \
    px3 = x4_[(r_ + 48 < a.N ? r_ + 48 : a.N - 1) * 16];                                                \
  } while (0)

template <int NL>
__device__ __forceinline__ void fc_fwd_body_f32(const FcFwdArgs& a, float* X0, float* X1, float* sB) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
  for (int l = 0; l < NL; ++l)      // static layer index: a runtime one would put the pointer table in scratch
    if (tid < FC_F) sB[l * FC_F + tid] = (l < NL - 1 || tid < FC_H) ? a.p.b[l][tid] : 0.f;
  // weight fragments of every layer: hidden layers column tile `wave`, last layer column tile wave & 1
  float wf[NL][16];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int ct = l < NL - 1 ? wave : (wave & 1);
    const float4* p = reinterpret_cast<const float4*>(a.Wf) + ((l * 4 + ct) * 4) * 64 + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      const float4 v = p[T * 64];
      wf[l][4 * T + 0] = v.x; wf[l][4 * T + 1] = v.y; wf[l][4 * T + 2] = v.z; wf[l][4 * T + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(wf[l][i]));
  }
  __syncthreads();

  const int64_t ntiles = (a.N + FC_TM - 1) / FC_TM;
  // next tile's rows, kept in four named registers (an array captured by a lambda ended up in scratch)
  float4 px0, px1, px2, px3;
  FC_FETCH((int64_t)blockIdx.x);
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    {
      float* d_ = X0 + (tid >> 4) * FC_LD + 4 * (tid & 15);
      *reinterpret_cast<float4*>(d_) = px0;
      *reinterpret_cast<float4*>(d_ + 16 * FC_LD) = px1;
      *reinterpret_cast<float4*>(d_ + 32 * FC_LD) = px2;
      *reinterpret_cast<float4*>(d_ + 48 * FC_LD) = px3;
    }
    NG_LDS_BARRIER();
    FC_FETCH(tile + gridDim.x < ntiles ? tile + gridDim.x : tile);
    fc_fwd_tile_f32<NL, true>(a, tile * FC_TM, X0, X1, sB, wf, wave, lane);
    NG_LDS_BARRIER();      // the next tile overwrites X0 (and X1 after its first layer)
  }
}

// The repair path of the piece body: the tile's rows again from HBM, then the fp32 layers with per-layer
// fragment loads.  Called by the whole workgroup (uniform) for a tile that flagged an activation beyond the fp16 range.
template <int NL>
__device__ __forceinline__ void fc_fwd_repair(const FcFwdArgs& a, int64_t tile, float* X0, float* X1, const float* sB) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float4 px0, px1, px2, px3;
  FC_FETCH(tile);
  float* d_ = X0 + (tid >> 4) * FC_LD + 4 * (tid & 15);
  *reinterpret_cast<float4*>(d_) = px0;
  *reinterpret_cast<float4*>(d_ + 16 * FC_LD) = px1;
  *reinterpret_cast<float4*>(d_ + 32 * FC_LD) = px2;
  *reinterpret_cast<float4*>(d_ + 48 * FC_LD) = px3;
  __syncthreads();
  fc_fwd_tile_f32<NL, false>(a, tile * FC_TM, X0, X1, sB, nullptr, wave, lane);
  __syncthreads();
}

// ---- the piece body (round 4): every product as three v_mfma_f32_16x16x32_f16 on two-fp16-piece operands -----------
// The f32-input MFMAs were 24 us of the forward's 71 and 44 of the backward's 112 (-DFC_ABL_NOMFMA*): 64 instructions
// of 32 cycles per layer, tile and wave, and an f32-input MFMA stream blocks the SIMD's issue.  Here a layer is 24
// instructions of 16 cycles.  Operands: weights as pieces of 2^8 W (PK_FC image, fc_h2; out of range -> flag word ->
// the fp32 body for the whole launch), activations UNSCALED (h2_common.cuh: absolute piece error max(2^-25, 2^-22 |x|))
// in two row-major planes [64][64 + 8] fp16 that ping-pong between layers.  The lane that produces y[row][4 columns] in
// layer l produces the same rows and columns in layer l + 1, so the residual input stays in registers (layer 0 reads
// it from the staged fp32 tile).  An activation at or beyond 65504 (or non-finite) flags the TILE; after its last layer
// the workgroup redoes a flagged tile with the fp32 layers (fc_fwd_repair): no second launch, nothing assumed about
// the operands' range.
constexpr int FC_ROWB = (FC_F + 8) * 2;        // bytes per plane row: rows 36 banks apart, ds_read_b128 of 16 rows conflict-free
constexpr int FC_PLANE = FC_TM * FC_ROWB;      // bytes per piece plane
// activations enter the planes times 2^4: an l piece is an fp16 subnormal (absolute quantum 2^-24) below |x| ~ 2^-3
// unscaled — 2^-7 with the factor, so the absolute piece error is max(2^-29, 2^-22 |x|); the range bound is 65504 / 16
constexpr float FC_XS = 16.0f, FC_XMAX = 65504.0f / 16.0f;

__device__ __forceinline__ bool fc_out_of_range(const float4& v) {
  return !(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) < FC_XMAX);
}

template <int NL, int ACT = -1>
__device__ __forceinline__ void fc_fwd_body_h2(const FcFwdArgs& a, float* X0, char* planes, float* sB, int* s_bad) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a16 = lane & 15, g4 = lane >> 4;
#pragma unroll
  for (int l = 0; l < NL; ++l)
    if (tid < FC_F) sB[l * FC_F + tid] = (l < NL - 1 || tid < FC_H) ? a.p.b[l][tid] : 0.f;
  if (tid == 0) s_bad[0] = 0;
  // piece fragments of every layer (A operands): hidden layers column tile `wave`, last layer column tile wave & 1
  u32x4 wh[NL][2], wl[NL][2];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int ct = l < NL - 1 ? wave : (wave & 1);
    const u32x4* p = reinterpret_cast<const u32x4*>(a.WfH) + (size_t)((l * 4 + ct) * 2) * 2 * 64 + lane;
#pragma unroll
    for (int T = 0; T < 2; ++T) { wh[l][T] = p[(2 * T) * 64]; wl[l][T] = p[(2 * T + 1) * 64]; }
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(wh[l][T][j])); asm volatile("" : "+v"(wl[l][T][j])); }
  }
  __syncthreads();

  const int64_t ntiles = (a.N + FC_TM - 1) / FC_TM;
  float4 px0, px1, px2, px3;
  FC_FETCH((int64_t)blockIdx.x);
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * FC_TM;
    bool bad = false;
    {   // the tile's rows: fp32 (layer 0's residual) and pieces (layer 0's operand)
      float* d_ = X0 + (tid >> 4) * FC_LD + 4 * (tid & 15);
      char* q_ = planes + (tid >> 4) * FC_ROWB + 8 * (tid & 15);
#define FC_PUT(PX, I)                                                                                     \
      do {                                                                                                \
        *reinterpret_cast<float4*>(d_ + 16 * (I) * FC_LD) = PX;                                           \
        unsigned h0_, l0_, h1_, l1_;                                                                      \
        split2_pair(FC_XS * PX.x, FC_XS * PX.y, h0_, l0_); split2_pair(FC_XS * PX.z, FC_XS * PX.w, h1_, l1_);     \
        *reinterpret_cast<u32x2*>(q_ + 16 * (I) * FC_ROWB) = u32x2{h0_, h1_};                             \
        *reinterpret_cast<u32x2*>(q_ + 16 * (I) * FC_ROWB + FC_PLANE) = u32x2{l0_, l1_};                  \
        bad |= fc_out_of_range(PX);                                                                       \
      } while (0)
      FC_PUT(px0, 0); FC_PUT(px1, 1); FC_PUT(px2, 2); FC_PUT(px3, 3);
#undef FC_PUT
    }
    NG_LDS_BARRIER();
    FC_FETCH(tile + gridDim.x < ntiles ? tile + gridDim.x : tile);
    char* Pin = planes;
    char* Pout = planes + 2 * FC_PLANE;
    float4 yprev[4];
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
      const int col = 16 * wave + 4 * g4;
      const float4 bias = *reinterpret_cast<const float4*>(sB + l * FC_F + col);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int r = 16 * rt + a16;
        const char* xrow = Pin + r * FC_ROWB + 16 * g4;
        const u32x4 xh0 = *reinterpret_cast<const u32x4*>(xrow), xl0 = *reinterpret_cast<const u32x4*>(xrow + FC_PLANE);
        const u32x4 xh1 = *reinterpret_cast<const u32x4*>(xrow + 64), xl1 = *reinterpret_cast<const u32x4*>(xrow + 64 + FC_PLANE);
        // acc0 collects the small products, acc1 the leading ones (and the bias, times the weights' 2^8)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc1 = {(256.0f * FC_XS) * bias.x, (256.0f * FC_XS) * bias.y, (256.0f * FC_XS) * bias.z, (256.0f * FC_XS) * bias.w};
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[l][0]), __builtin_bit_cast(f16x8, xh0), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[l][0]), __builtin_bit_cast(f16x8, xh0), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[l][0]), __builtin_bit_cast(f16x8, xl0), acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[l][1]), __builtin_bit_cast(f16x8, xh1), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[l][1]), __builtin_bit_cast(f16x8, xh1), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[l][1]), __builtin_bit_cast(f16x8, xl1), acc0, 0, 0, 0);
        const f32x4 v = (acc0 + acc1) * (1.0f / (256.0f * FC_XS));
        const float4 s = act4(ACT >= 0 ? ACT : a.act, v);
        const float4 xo = l == 0 ? *reinterpret_cast<const float4*>(X0 + r * FC_LD + col) : yprev[rt];
        const float4 y = make_float4(s.x + xo.x, s.y + xo.y, s.z + xo.z, s.w + xo.w);
        yprev[rt] = y;
        if (a.p.y[l]) {
          const int64_t row = row0 + r;
          *reinterpret_cast<float4*>(row < a.N ? a.p.y[l] + row * FC_F + col : a.dummy + col) = y;
        }
        bad |= fc_out_of_range(y);
        unsigned h0, l0, h1, l1;
        split2_pair(FC_XS * y.x, FC_XS * y.y, h0, l0); split2_pair(FC_XS * y.z, FC_XS * y.w, h1, l1);
        char* q = Pout + r * FC_ROWB + 2 * col;
        *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(q + FC_PLANE) = u32x2{l0, l1};
      }
      NG_LDS_BARRIER();
      char* tmp = Pin; Pin = Pout; Pout = tmp;
    }
    {   // last layer: column tile wave & 1, row tiles 2*(wave >> 1), +1
      constexpr int l = NL - 1;
      const int col = 16 * (wave & 1) + 4 * g4;
      const float4 bias = *reinterpret_cast<const float4*>(sB + l * FC_F + col);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 16 * (2 * (wave >> 1) + h) + a16;
        const char* xrow = Pin + r * FC_ROWB + 16 * g4;
        const u32x4 xh0 = *reinterpret_cast<const u32x4*>(xrow), xl0 = *reinterpret_cast<const u32x4*>(xrow + FC_PLANE);
        const u32x4 xh1 = *reinterpret_cast<const u32x4*>(xrow + 64), xl1 = *reinterpret_cast<const u32x4*>(xrow + 64 + FC_PLANE);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc1 = {(256.0f * FC_XS) * bias.x, (256.0f * FC_XS) * bias.y, (256.0f * FC_XS) * bias.z, (256.0f * FC_XS) * bias.w};
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[l][0]), __builtin_bit_cast(f16x8, xh0), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[l][0]), __builtin_bit_cast(f16x8, xh0), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[l][0]), __builtin_bit_cast(f16x8, xl0), acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[l][1]), __builtin_bit_cast(f16x8, xh1), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[l][1]), __builtin_bit_cast(f16x8, xh1), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[l][1]), __builtin_bit_cast(f16x8, xl1), acc0, 0, 0, 0);
        const f32x4 v = (acc0 + acc1) * (1.0f / (256.0f * FC_XS));
        const float4 s = act4(ACT >= 0 ? ACT : a.act, v);
        const int64_t row = row0 + r;
        *reinterpret_cast<float4*>(row < a.N ? a.g + row * FC_H + col : a.dummy + col) = s;
      }
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) s_bad[0] = 1;
    __syncthreads();      // (also: the flagged tile's stores are out before the repair's stores to the same rows)
    if (s_bad[0]) {       // uniform over the workgroup
      fc_fwd_repair<NL>(a, tile, X0, reinterpret_cast<float*>(planes), sB);
      if (tid == 0) s_bad[0] = 0;
      __syncthreads();
    }
  }
}

// both bodies in one kernel: the choice is uniform over the launch (weights beyond the piece range: the pack launch of this
// call raised the guard, or the image kept over calls has its flag word set) and costs no second launch
template <int NL, bool H2>
__global__ __launch_bounds__(256, 2) void fc_fwd_kernel(FcFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  float* X0 = fsm;                                              // [64][68] fp32
  char* planes = reinterpret_cast<char*>(X0 + FC_TM * FC_LD);   // [2 buffers][2 planes][64][72] fp16   (fp32 body: X1)
  float* sB = reinterpret_cast<float*>(planes + 4 * FC_PLANE);  // [NL][64]
  int* s_bad = reinterpret_cast<int*>(sB + NL * FC_F);
  if (H2 && !(a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag))))
  {
    // the activation as a compile-time constant for the reference's default (round 6): read from the arguments, every use was a
    // scalar branch around the other activations' code — a basic-block boundary between the MFMAs and the elementwise work of
    // every unit (forward 52.7 -> 46 us)
    if (a.act == NG_ACT_SOFTPLUS) fc_fwd_body_h2<NL, NG_ACT_SOFTPLUS>(a, X0, planes, sB, s_bad);
    else fc_fwd_body_h2<NL, -1>(a, X0, planes, sB, s_bad);
  }
  else
    fc_fwd_body_f32<NL>(a, X0, reinterpret_cast<float*>(planes), sB);
}

static size_t fc_fwd_lds_bytes(int L) { return (size_t)FC_TM * FC_LD * 4 + 4 * FC_PLANE + (size_t)L * FC_F * 4 + 16; }

// ---- backward --------------------------------------------------------------------------------------------
// 512 threads = 8 waves, one persistent workgroup per CU, five LDS tiles [64][68]:
//   D0/D1  upstream gradient of the current layer (ping-pong),  P  dP = dY * act'(s),
//   XA/XB  layer input x_l and layer output y_l = x_{l+1} (the output of layer l is the input of l+1, so
//          walking the layers downwards only x_{l-1} has to come from HBM: one 64x64 tile per layer).
// Per layer:  [VALU] P = D * act'(y - x), bias-gradient partial            | barrier
//             [MFMA] dW_l += x^T P   (wave: k-tile w&3, n-tiles 2(w>>2), +1; K-strided ds_read_b32)
//                    dX   = P W_l^T (+ D)  (wave: k-tile w&3, row tiles 2(w>>2), +1) -> other D buffer
//             next layer's x tile: registers -> the LDS buffer y_l no longer needs   | barrier
// Weight gradients live in registers for the whole launch (8 VGPRs per layer and wave); one partial per
// workgroup, summed by reduce_z (deterministic).
struct FcBwdArgs {
  int64_t N;
  int act;
  const float* x[FC_MAXL];   // x_l, layer inputs [N][64]
  const float* g;            // [N][32] block output
  const float* dg;           // [N][32] upstream gradient
  const float* Wb;           // packed backward fragments
  float* dx;                 // [N][64] gradient w.r.t. x_0
  float* partial;            // [grid][part_stride]
  int part_stride;
  float* dummy;
  const unsigned* WbH;       // packed piece fragments (pack_bodies.cuh: fc_h2)
  RangeGuard guard;          // as in FcFwdArgs
  const unsigned* wflag;
  unsigned wflag_ver;
};

typedef short gs16x4 __attribute__((ext_vector_type(4)));
template <int S>
__device__ __forceinline__ float fc_ror(float v) {      // row_ror:S within the 16 lanes of a DPP row
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (S & 15), 0xf, 0xf, false));
}
__device__ __forceinline__ int fc_wave_min(int v) {     // valid in lane 63
  const int big = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false));
  return v;
}

__host__ __device__ inline int fc_part_floats(int L) { return L * FC_F * FC_F + L * FC_F; }

// rows of the tile contracted by MFMA k-step T (0..15), lane group g: conflict-free for row strides == 4 mod 16
__device__ __forceinline__ int kstep_row(int T, int g) { return (T & 3) + 4 * g + 16 * (T >> 2); }

template <int NL, bool H2, int ACT = -1>      // ACT: compile-time activation (softplus) or -1 = from the arguments, as in the forward
__device__ __forceinline__ void fc_bwd_body(const FcBwdArgs& a) {
  const int act_ = ACT >= 0 ? ACT : a.act;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* D0 = smem;
  float* D1 = D0 + FC_TM * FC_LD;
  float* Pt = D1 + FC_TM * FC_LD;                                   // fp32 body: dP tile;  piece body: its two fp16 planes
  char* planes = reinterpret_cast<char*>(Pt);                       // [2][64][72] fp16
  float* XA = reinterpret_cast<float*>(planes + 2 * FC_PLANE);
  float* XB = XA + FC_TM * FC_LD;
  float* s_rs = XB + FC_TM * FC_LD;                                 // [64] 2^-8 / S per row of the dP planes
  int* s_sb = reinterpret_cast<int*>(s_rs + FC_TM);                 // [64] biased exponent of S
  int* s_wmin = s_sb + FC_TM;                                       // [8]  per-wave minimum of it over rows that are not all zero
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a16 = lane & 15, g4 = lane >> 4;
  const int kt = wave & 3, pr = wave >> 2;      // k-tile, pair index (n-tiles / row tiles 2pr, 2pr+1)

  float wb[H2 ? 1 : NL][16];
  u32x4 wbh[H2 ? NL : 1][2], wbl[H2 ? NL : 1][2];      // piece body: A operands of the dX product (fc_h2: WbH)
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    if (H2) {
      const u32x4* p = reinterpret_cast<const u32x4*>(a.WbH) + (size_t)((l * 4 + kt) * 2) * 2 * 64 + lane;
#pragma unroll
      for (int T = 0; T < 2; ++T) { wbh[H2 ? l : 0][T] = p[(2 * T) * 64]; wbl[H2 ? l : 0][T] = p[(2 * T + 1) * 64]; }
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(wbh[H2 ? l : 0][T][j])); asm volatile("" : "+v"(wbl[H2 ? l : 0][T][j])); }
    } else {
      const float4* p = reinterpret_cast<const float4*>(a.Wb) + ((l * 4 + kt) * 4) * 64 + lane;
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        const float4 v = p[T * 64];
        wb[H2 ? 0 : l][4 * T + 0] = v.x; wb[H2 ? 0 : l][4 * T + 1] = v.y; wb[H2 ? 0 : l][4 * T + 2] = v.z; wb[H2 ? 0 : l][4 * T + 3] = v.w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(wb[H2 ? 0 : l][i]));
    }
  }
  f32x4 accW[NL][2];
  // bias-gradient sums of this thread's (row lane, column chunk): registers in the fp32 body; the piece body keeps them
  // in LDS (same owner, same order of additions: same bits) — its four layers of piece fragments left no room (68 B/lane
  // of scratch with them in registers)
  float4 accb[H2 ? 1 : NL];
  float* s_accb = reinterpret_cast<float*>(s_wmin + 8);            // [32][NL*64] (piece body)
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    accW[l][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    accW[l][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (H2) *reinterpret_cast<float4*>(s_accb + (tid >> 4) * (NL * FC_F) + l * FC_F + 4 * (tid & 15)) = f4zero();
    else accb[H2 ? 0 : l] = f4zero();
  }

  // elementwise ownership: thread -> rows er, er + 32, column chunk ec (float4)
  const int er = tid >> 4, ec = tid & 15;
  const int64_t ntiles = (a.N + FC_TM - 1) / FC_TM;
  float4 px[2];                 // next x tile (or dg | g at the head of a tile)
  auto fetch_x = [&](const float* src, int64_t row0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t row = row0 + er + 32 * i;
      px[i] = reinterpret_cast<const float4*>(src)[(row < a.N ? row : a.N - 1) * 16 + ec];
      if (row >= a.N) px[i] = f4zero();
    }
  };
  auto fetch_head = [&](int64_t row0) {     // columns 0..31: dg, columns 32..63: g
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t row = row0 + er + 32 * i;
      const float* src = ec < 8 ? a.dg : a.g;
      px[i] = reinterpret_cast<const float4*>(src)[(row < a.N ? row : a.N - 1) * 8 + (ec & 7)];
      if (row >= a.N) px[i] = f4zero();
    }
  };
  auto put = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(buf + (er + 32 * i) * FC_LD + 4 * ec) = px[i];
  };
  float4 py[2];                 // x_{NL-1} of the NEXT tile, requested a whole tile ahead
  auto fetch_top = [&](int64_t row0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t row = row0 + er + 32 * i;
      py[i] = reinterpret_cast<const float4*>(a.x[NL - 1])[(row < a.N ? row : a.N - 1) * 16 + ec];
      if (row >= a.N) py[i] = f4zero();
    }
  };

  fetch_head((int64_t)blockIdx.x * FC_TM);
  fetch_top((int64_t)blockIdx.x * FC_TM);
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * FC_TM;
    // head: dg -> D0[:, 0:32], g -> XB[:, 0:32] (y of the last layer); then x_{NL-1} -> XA
    {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* dst = ec < 8 ? D0 : XB;
        *reinterpret_cast<float4*>(dst + (er + 32 * i) * FC_LD + 4 * (ec & 7)) = px[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(XA + (er + 32 * i) * FC_LD + 4 * ec) = py[i];
    fetch_top(tile + gridDim.x < ntiles ? (tile + gridDim.x) * FC_TM : row0);
    if (NL >= 2) fetch_x(a.x[NL - 2], row0);
    NG_LDS_BARRIER();
    float* Dc = D0;
    float* Dn = D1;
    float* Xt = XA;      // x_l
    float* Yt = XB;      // y_l
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
      const bool last = l == NL - 1;
      // ---- P = D * act'(s);  s = y - x (hidden) or g (last);  bias-gradient partial
      // piece body: the row goes into the two fp16 planes scaled by a power of two S taken from the row's own max |P| (the
      // 16 lanes of a DPP row hold the whole row; upstream gradients span decades between labelled and unlabelled atoms):
      // S max in [2^13, 2^14); s_rs[row] = 2^-8 / S for the dX epilogue, s_sb[row] the exponent for the dW product
      int wm = 253;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = (er + 32 * i) * FC_LD + 4 * ec;
        float4 p = f4zero();
        if (!last || ec < 8) {
          const float4 d = *reinterpret_cast<const float4*>(Dc + o);
          const float4 y = *reinterpret_cast<const float4*>(Yt + o);
          float4 s = y;
          if (!last) {
            const float4 x = *reinterpret_cast<const float4*>(Xt + o);
            s.x -= x.x; s.y -= x.y; s.z -= x.z; s.w -= x.w;
          }
          p = d;
#ifdef FC_ABL_NOACT_BWD
          if (act_ != NG_ACT_NONE && a.N < 0) {
#else
          if (act_ != NG_ACT_NONE) {
#endif
            p.x *= act_grad_from_out(act_, s.x); p.y *= act_grad_from_out(act_, s.y);
            p.z *= act_grad_from_out(act_, s.z); p.w *= act_grad_from_out(act_, s.w);
          }
        }
        if (H2) {
          float4* sa = reinterpret_cast<float4*>(s_accb + er * (NL * FC_F) + l * FC_F + 4 * ec);
          float4 t = *sa;
          t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
          *sa = t;
        } else {
          accb[H2 ? 0 : l].x += p.x; accb[H2 ? 0 : l].y += p.y; accb[H2 ? 0 : l].z += p.z; accb[H2 ? 0 : l].w += p.w;
        }
#ifdef FC_ABL_H2_NOP
        if (H2 && a.N < 0) {
#else
        if (H2) {
#endif
          float m = fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fmaxf(fabsf(p.z), fabsf(p.w)));
          m = fmaxf(m, fc_ror<8>(m)); m = fmaxf(m, fc_ror<4>(m)); m = fmaxf(m, fc_ror<2>(m)); m = fmaxf(m, fc_ror<1>(m));
          const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
          const int sb = (ef == 0 || ef == 255) ? 127 : min(267 - ef, 253);      // zero / non-finite rows: S = 1
          const float S = __builtin_bit_cast(float, sb << 23);
          const int row = er + 32 * i;
          if (ec == 0) { s_rs[row] = __builtin_bit_cast(float, (254 - sb) << 23) * (1.0f / 256.0f); s_sb[row] = sb; }
          wm = min(wm, ef == 0 ? 253 : sb);
          unsigned h0, l0, h1, l1;
          split2_pair(S * p.x, S * p.y, h0, l0); split2_pair(S * p.z, S * p.w, h1, l1);
          char* q = planes + row * FC_ROWB + 8 * ec;
          *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
          *reinterpret_cast<u32x2*>(q + FC_PLANE) = u32x2{l0, l1};
        } else {
          *reinterpret_cast<float4*>(Pt + o) = p;
        }
      }
      if (H2) {
        wm = fc_wave_min(wm);
        if (lane == 63) s_wmin[wave] = wm;
      }
      NG_LDS_BARRIER();
      // ---- dW_l[k][n] += sum_rows x[row][k] P[row][n]: D rows i = n, cols j = k
      if (H2) {
        // dW on the fp16 pipe: D[n][k] += sum_rows P[row][n] x[row][k], the contraction over the tile's rows in two 32-deep
        // steps.  The P rows carry different scales S_r, so the x operand takes the inverse: x'[r] = x[r] * S_ref / S_r with
        // S_ref the smallest S of the tile (its largest row; ratio <= 1), the step's product goes into a fresh accumulator and
        // is added to the running sums times 1 / S_ref (mp_win_bwd.hip, node kernel: the same scheme).
#ifdef FC_ABL_H2_NODW
        if (a.N < 0) {
#else
        if (!last || pr == 0) {
#endif
          int sbref = s_wmin[0];
#pragma unroll
          for (int i = 1; i < 8; ++i) sbref = min(sbref, s_wmin[i]);
          const float inv_ref = __builtin_bit_cast(float, (254 - sbref) << 23);
#pragma unroll
          for (int step = 0; step < 2; ++step) {
            float hv[8];
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
              const int row = 32 * step + 8 * g4 + tt;
              const int sba = s_sb[row];
              // all-zero rows keep S = 1 (sba may lie below the reference): their P pieces are exact zeros, any finite x' does
              const float ratio = sba < sbref ? 1.0f : __builtin_bit_cast(float, max(sbref - sba + 127, 0) << 23);
              hv[tt] = Xt[row * FC_LD + 16 * kt + a16] * ratio;
            }
            // the layer input is a forward quantity of any size: feature column k = 16 kt + a16 (this lane's 8 rows here, the
            // rest of the step's 32 in the lanes 16 / 32 / 48 further on) gets a power-of-two scale when its largest entry
            // reaches 2^15, and this lane's outputs (all of column k) take the inverse.  1 (bit-neutral) for ordinary inputs.
            float hm = fmaxf(fmaxf(fmaxf(fabsf(hv[0]), fabsf(hv[1])), fmaxf(fabsf(hv[2]), fabsf(hv[3]))),
                             fmaxf(fmaxf(fabsf(hv[4]), fabsf(hv[5])), fmaxf(fabsf(hv[6]), fabsf(hv[7]))));
            float osc = inv_ref;
            if (__builtin_amdgcn_ballot_w64(hm >= 32768.0f) != 0) {      // wave-uniform, never taken for ordinary activations
              hm = fmaxf(hm, __shfl_xor(hm, 16));
              hm = fmaxf(hm, __shfl_xor(hm, 32));
              const int hef = (__builtin_bit_cast(int, hm) >> 23) & 255;
              const bool hbig = hef >= 127 + 15 && hef != 255;
              const float hs = hbig ? __builtin_bit_cast(float, (268 - hef) << 23) : 1.0f;
              const float hsi = hbig ? __builtin_bit_cast(float, (hef - 14) << 23) : 1.0f;
#pragma unroll
              for (int tt = 0; tt < 8; ++tt) hv[tt] *= hs;
              osc = hsi * inv_ref;
            }
            unsigned h0, l0, h1, l1, h2, l2, h3, l3;
            split2_pair(hv[0], hv[1], h0, l0); split2_pair(hv[2], hv[3], h1, l1);
            split2_pair(hv[4], hv[5], h2, l2); split2_pair(hv[6], hv[7], h3, l3);
            const u32x4 xh = {h0, h1, h2, h3}, xl = {l0, l1, l2, l3};
            // A operand: column n = 16 (2 pr + j) + a16 of the dP planes over the step's rows 8 g4 .. + 7: two transposing
            // reads of four rows per plane
            const char* bp = planes + (32 * step + 8 * g4 + (a16 >> 2)) * FC_ROWB + (32 * pr + 4 * (a16 & 3)) * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const char* q0 = bp + 32 * j;
              const gs16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)q0);
              const gs16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + 4 * FC_ROWB));
              const gs16x4 w0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + FC_PLANE));
              const gs16x4 w1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(q0 + FC_PLANE + 4 * FC_ROWB));
              const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
              const u32x2 c0 = __builtin_bit_cast(u32x2, w0), c1 = __builtin_bit_cast(u32x2, w1);
              const u32x4 ph = {a0[0], a0[1], a1[0], a1[1]}, pl = {c0[0], c0[1], c1[0], c1[1]};
              f32x4 at = {0.f, 0.f, 0.f, 0.f};
              at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, pl), __builtin_bit_cast(f16x8, xh), at, 0, 0, 0);
              at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ph), __builtin_bit_cast(f16x8, xl), at, 0, 0, 0);
              at = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ph), __builtin_bit_cast(f16x8, xh), at, 0, 0, 0);
              accW[l][j] += at * osc;
            }
          }
        }
      } else if (!last || pr == 0) {
        const float* xb = Xt + 16 * kt + a16;
        const float* p0 = Pt + 16 * (2 * pr) + a16;
#pragma unroll
        for (int T = 0; T < 16; ++T) {
          const int ro = kstep_row(T, g4) * FC_LD;
          const float bv = xb[ro];
          const float a0 = p0[ro];
          const float a1 = p0[ro + 16];
#ifdef FC_ABL_NOMFMA_BWD
          accW[l][0][0] += a0 * bv; accW[l][1][0] += a1 * bv;
          continue;
#endif
          accW[l][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, accW[l][0], 0, 0, 0);
          accW[l][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, accW[l][1], 0, 0, 0);
        }
      }
      // ---- dX[row][k] = sum_n P[row][n] W_l[k][n] (+ D[row][k] for the residual layers) -> Dn
      {
        const float* q0 = Pt + (16 * (2 * pr) + a16) * FC_LD + 4 * g4;
        const float* q1 = q0 + 16 * FC_LD;
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        if (H2) {
          // piece body: per row tile three MFMAs per 32-wide step (the last layer's dP has 32 columns: one step); the small
          // products and the leading one in separate accumulators, the row's 2^-8 / S at the end
#ifdef FC_ABL_H2_NODX
          if (a.N < 0)
#endif
#pragma unroll
          for (int h = 0; h < 2; ++h) {      // one row tile after the other: eight operand registers live, not sixteen
            const char* xr = planes + (16 * (2 * pr + h) + a16) * FC_ROWB + 16 * g4;
            f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int T = 0; T < 2; ++T) {
              if (T == 0 || !last) {
                const u32x4 ah = *reinterpret_cast<const u32x4*>(xr + 64 * T), al = *reinterpret_cast<const u32x4*>(xr + 64 * T + FC_PLANE);
                const f16x8 WH = __builtin_bit_cast(f16x8, wbh[H2 ? l : 0][T]), WL = __builtin_bit_cast(f16x8, wbl[H2 ? l : 0][T]);
                cs = __builtin_amdgcn_mfma_f32_16x16x32_f16(WL, __builtin_bit_cast(f16x8, ah), cs, 0, 0, 0);
                cl = __builtin_amdgcn_mfma_f32_16x16x32_f16(WH, __builtin_bit_cast(f16x8, ah), cl, 0, 0, 0);
                cs = __builtin_amdgcn_mfma_f32_16x16x32_f16(WH, __builtin_bit_cast(f16x8, al), cs, 0, 0, 0);
              }
            }
            const f32x4 r = (cs + cl) * s_rs[16 * (2 * pr + h) + a16];
            if (h == 0) c0 = r; else c1 = r;
          }
        }
        const int nT = H2 ? 0 : (last ? 2 : 4);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
          if (T < nT) {
            const float4 xa = *reinterpret_cast<const float4*>(q0 + 16 * T);
            const float4 xb4 = *reinterpret_cast<const float4*>(q1 + 16 * T);
#ifdef FC_ABL_NOMFMA_BWD
            c0[0] += xa.x + xa.y + xa.z + xa.w + wb[H2 ? 0 : l][4 * T]; c1[0] += xb4.x + xb4.y + xb4.z + xb4.w;
            continue;
#endif
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[H2 ? 0 : l][4 * T + 0], xa.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[H2 ? 0 : l][4 * T + 0], xb4.x, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[H2 ? 0 : l][4 * T + 1], xa.y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[H2 ? 0 : l][4 * T + 1], xb4.y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[H2 ? 0 : l][4 * T + 2], xa.z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[H2 ? 0 : l][4 * T + 2], xb4.z, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[H2 ? 0 : l][4 * T + 3], xa.w, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[H2 ? 0 : l][4 * T + 3], xb4.w, c1, 0, 0, 0);
          }
        }
        const int col = 16 * kt + 4 * g4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = 16 * (2 * pr + h) + a16;
          const f32x4& c = h ? c1 : c0;
          float4 v = make_float4(c[0], c[1], c[2], c[3]);
          if (!last) {
            const float4 d = *reinterpret_cast<const float4*>(Dc + r * FC_LD + col);
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
          }
          if (l == 0) {
            const int64_t row = row0 + r;
            *reinterpret_cast<float4*>(row < a.N ? a.dx + row * FC_F + col : a.dummy + col) = v;
          } else {
            *reinterpret_cast<float4*>(Dn + r * FC_LD + col) = v;
          }
        }
      }
      // ---- next layer's input tile: registers -> the buffer y_l no longer needs; request the one after
      if (l >= 1) {
        put(Yt);
        if (l >= 2) fetch_x(a.x[l - 2], row0);
        else fetch_head(tile + gridDim.x < ntiles ? (tile + gridDim.x) * FC_TM : row0);
      }
      NG_LDS_BARRIER();
      float* t1 = Dc; Dc = Dn; Dn = t1;
      float* t2 = Yt; Yt = Xt; Xt = t2;      // y_{l-1} = x_l ;  x_{l-1} sits in the old y buffer
    }
  }

  // ---- this workgroup's partial: dW_l [64][nout] then db_l [nout]
  float* part = a.partial + (int64_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int nout = l == NL - 1 ? FC_H : FC_F;
    const int k = 16 * kt + a16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = 16 * (2 * pr + j) + 4 * g4;
      if (n < nout)
        *reinterpret_cast<float4*>(part + l * FC_F * FC_F + k * nout + n) =
            make_float4(accW[l][j][0], accW[l][j][1], accW[l][j][2], accW[l][j][3]);
    }
  }
  // bias gradients: sum over the 32 row lanes through LDS (tiles are free now)
  __syncthreads();
  float* red = H2 ? s_accb : smem;     // [32][NL*64]
  if (!H2) {
#pragma unroll
    for (int l = 0; l < NL; ++l) *reinterpret_cast<float4*>(red + er * (NL * FC_F) + l * FC_F + 4 * ec) = accb[H2 ? 0 : l];
  }
  __syncthreads();
  for (int it = tid; it < NL * FC_F; it += 512) {
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += red[r * (NL * FC_F) + it];
    part[NL * FC_F * FC_F + it] = s;
  }
}


// both bodies in one kernel (see fc_fwd_kernel): weights beyond the piece range select the fp32 body for the launch; the
// gradient operands of the piece body carry per-row scales, its x operand a per-column one — range-safe by construction
template <int NL, bool H2>
__global__ __launch_bounds__(512, 1) void fc_bwd_kernel(FcBwdArgs a) {
  if (H2 && !(a.guard.word && (range_guard_raised(a.guard) || wimage_flag_raised(a.wflag)))) {
    if (a.act == NG_ACT_SOFTPLUS) fc_bwd_body<NL, true, NG_ACT_SOFTPLUS>(a);
    else fc_bwd_body<NL, true>(a);
  } else {
    fc_bwd_body<NL, false>(a);
  }
}

static size_t fc_bwd_lds_bytes(int L) { return (size_t)4 * FC_TM * FC_LD * 4 + 2 * FC_PLANE + (FC_TM + FC_TM + 8) * 4 + (size_t)32 * L * FC_F * 4; }

}  // namespace

bool fc_fused_supported(int F, int L) {
  if (sw().fc_layered) return false;
  return F == FC_F && L >= 2 && L <= FC_MAXL;
}

// the fused BACKWARD keeps every layer's dW block in registers: four layers fit (the reference's default, model.py:33), five
// spilled 32 B and six 192 B per lane — those depths take the layer-by-layer backward below (same tape: the layer inputs)
bool fc_fused_bwd_supported(int F, int L) { return fc_fused_supported(F, L) && L <= 4; }

// image: fp32 forward + backward fragments, a dummy row, the flag word, then the fp16 piece fragments of both directions
size_t fc_fused_pack_floats(int L) { return (size_t)4 * L * FC_F * FC_F + 64 + 16; }

static bool fc_h2_on() { return !sw().gemm_math_fp32; }

struct FcImage {
  float* Wf; float* Wb; float* dummy;
  unsigned* WfH; unsigned* WbH;
  const unsigned* wflag;      // flag word of a cached image (nullptr: the call's own pack launch raises the guard instead)
};

// the block's packed weights: the cached image of the weights when the cache is on (shared by the forward and the backward of
// a step, refreshed behind Adam), else `scratch`.  A guarded call's pack launch checks the weights against the piece range.
static bool fc_packed(ng_ctx* ctx, hipStream_t st, int L, const float* const* W, float* scratch, RangeGuard guard, FcImage* im,
                      int* rc) {
  bool have = false;
  float* ws = (float*)cached_image(ctx, W[0], 5, fc_fused_pack_floats(L) * 4, &have);
  const bool cached = ws != nullptr;
  if (!ws) ws = scratch;
  *rc = NG_OK;
  if (!ws) return false;      // not cacheable and the caller brought no scratch: it calls again with one
  const size_t LF = (size_t)L * FC_F * FC_F;
  im->Wf = ws; im->Wb = ws + LF; im->dummy = ws + 2 * LF;
  unsigned* flag = reinterpret_cast<unsigned*>(ws + 2 * LF + 64);
  im->WfH = reinterpret_cast<unsigned*>(ws + 2 * LF + 80); im->WbH = im->WfH + LF;
  im->wflag = cached ? flag : nullptr;
  if (!have) {
    PackJob j;
    j.kind = PK_FC; j.blocks = 32; j.i0 = L;
    for (int l = 0; l < L; ++l) j.src[l] = W[l];
    j.dst[0] = im->Wf; j.dst[1] = im->Wb; j.dst[2] = (float*)im->WfH; j.dst[3] = (float*)im->WbH;
    j.flag = cached ? flag : nullptr; j.guard = guard;
    *rc = pack_launch(ctx, st, j);
    if (*rc == NG_OK && cached) cache_set_job(ctx, W[0], 5, j);
  }
  return true;
}

int fc_fused_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int L, int act, const float* x, const float* const* W,
                 const float* const* b, float* const* y, float* g) {
  if (N == 0) return NG_OK;
  const bool h2 = fc_h2_on() && L <= 5;      // (six layers of piece fragments spill: the fp32 body)
  RangeGuard guard{nullptr, 0};
  if (h2) {
    guard = range_guard_begin(ctx);
    if (!guard.word) return NG_ERR_NOMEM;
  }
  float* scratch = ctx->wcache ? nullptr : (float*)workspace(ctx, fc_fused_pack_floats(L) * 4);
  int rc = NG_OK;
  FcImage im{};
  if (!fc_packed(ctx, st, L, W, scratch, guard, &im, &rc)) {      // (cache on, but this source is not cacheable)
    scratch = (float*)workspace(ctx, fc_fused_pack_floats(L) * 4);
    if (!scratch) return NG_ERR_NOMEM;
    (void)fc_packed(ctx, st, L, W, scratch, guard, &im, &rc);
  }
  if (rc) return rc;
  FcFwdArgs a{};
  a.N = N; a.act = act; a.x = x; a.Wf = im.Wf; a.g = g; a.dummy = im.dummy;
  a.WfH = im.WfH; a.guard = guard; a.wflag = im.wflag; a.wflag_ver = pack_flag_version(ctx);
  for (int l = 0; l < L; ++l) {
    a.p.W[l] = W[l]; a.p.b[l] = b[l];
    a.p.y[l] = (y && l < L - 1) ? y[l] : nullptr;
  }
  const int64_t ntiles = cdiv(N, FC_TM);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu * 2);
  const size_t lds = fc_fwd_lds_bytes(L);
  ProfScope ps(ctx, st, "fc_fused_fwd");
#define NG_FCF(LL)                                                                                           \
  do {                                                                                                       \
    if (h2) hipLaunchKernelGGL((fc_fwd_kernel<LL, true>), dim3(grid), dim3(256), lds, st, a);                \
    else hipLaunchKernelGGL((fc_fwd_kernel<LL, false>), dim3(grid), dim3(256), lds, st, a);                  \
  } while (0)
  switch (L) {
    case 2: NG_FCF(2); break;
    case 3: NG_FCF(3); break;
    case 4: NG_FCF(4); break;
    case 5: NG_FCF(5); break;
    case 6: NG_FCF(6); break;
  }
#undef NG_FCF
  NG_HIP(ctx, hipGetLastError());
  return NG_OK;
}

int fc_fused_bwd(ng_ctx* ctx, hipStream_t st, int64_t N, int L, int act, const float* const* x, const float* g,
                 const float* const* W, const float* dg, float* dx, float* const* dW, float* const* db) {
  const int part = (fc_part_floats(L) + 3) / 4 * 4;
  const int64_t ntiles = std::max<int64_t>(cdiv(N, FC_TM), 1);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu);
  float* ws = (float*)workspace(ctx, (fc_fused_pack_floats(L) + (size_t)(grid + 1) * part) * 4);
  if (!ws) return NG_ERR_NOMEM;
  float* partial = ws + fc_fused_pack_floats(L);
  float* summed = partial + (size_t)grid * part;
  if (float* dp = deferred_partials(ctx, (size_t)grid * part)) partial = dp;
  const bool h2 = fc_h2_on();
  RangeGuard guard{nullptr, 0};
  if (h2) {
    guard = range_guard_begin(ctx);
    if (!guard.word) return NG_ERR_NOMEM;
  }
  int rc = NG_OK;
  FcImage im{};
  (void)fc_packed(ctx, st, L, W, ws, guard, &im, &rc);
  if (rc) return rc;
  FcBwdArgs a{};
  a.N = N; a.act = act; a.g = g; a.dg = dg; a.Wb = im.Wb; a.dx = dx; a.partial = partial; a.part_stride = part;
  a.dummy = im.dummy;
  a.WbH = im.WbH; a.guard = guard; a.wflag = im.wflag; a.wflag_ver = pack_flag_version(ctx);
  for (int l = 0; l < L; ++l) a.x[l] = x[l];
  const size_t lds = fc_bwd_lds_bytes(L);
  {
    ProfScope ps(ctx, st, "fc_fused_bwd");
#define NG_FCB(LL)                                                                                           \
  do {                                                                                                       \
    if (h2) hipLaunchKernelGGL((fc_bwd_kernel<LL, true>), dim3(grid), dim3(512), lds, st, a);                \
    else hipLaunchKernelGGL((fc_bwd_kernel<LL, false>), dim3(grid), dim3(512), lds, st, a);                  \
  } while (0)
    switch (L) {
      case 2: NG_FCB(2); break;
      case 3: NG_FCB(3); break;
      case 4: NG_FCB(4); break;
      default: return fail(ctx, NG_ERR_UNSUPPORTED, "fc_fused_bwd: more than four layers take the layered backward");
    }
#undef NG_FCB
    NG_HIP(ctx, hipGetLastError());
  }
  ProfScope ps(ctx, st, "reduce_partials");
  ReduceSegs sg{};
  sg.n = 2 * L;
  for (int l = 0; l < L; ++l) {
    const int nout = l == L - 1 ? FC_H : FC_F;
    sg.begin[l] = l * FC_F * FC_F; sg.len[l] = FC_F * nout; sg.dst[l] = dW[l];
    sg.begin[L + l] = L * FC_F * FC_F + l * FC_F; sg.len[L + l] = nout; sg.dst[L + l] = db[l];
  }
  (void)summed;
  return reduce_seg_or_defer(ctx, st, partial, grid, fc_part_floats(L), part, sg);
}

}  // namespace ng

// ---- C ABI ---------------------------------------------------------------------------------------------
namespace ng {
// dP[i][n] = dY[i][n] * act'(s[i][n]),  s = a[i][n] - (b ? b[i][n] : 0)  (the activation OUTPUT of the layer);
// partial[blk][n] = column sums of dP over the block's rows (bias gradient, two-stage deterministic)
__global__ __launch_bounds__(256) void fc_dp_kernel(int64_t N, int No, int64_t rows_per_block, int act,
                                                    const float* __restrict__ dY, const float* __restrict__ a,
                                                    const float* __restrict__ b, float* __restrict__ dP,
                                                    float* __restrict__ partial, float* __restrict__ blockmax) {
  extern __shared__ __attribute__((aligned(16))) float fc_red[];     // [RL][No]
  const int c4n = No / 4;
  const int RL = 256 / c4n;
  const int q = threadIdx.x % c4n, r = threadIdx.x / c4n;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = std::min<int64_t>(r0 + rows_per_block, N);
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  float amax = 0.f;      // max |dP| of the block: the fp16 GEMMs' power-of-two gradient scale (gemm_h2.hip)
  if (r < RL) {
    for (int64_t i = r0 + r; i < r1; i += RL) {
      const int64_t o = i * c4n + q;
      float4 d = reinterpret_cast<const float4*>(dY)[o];
      float4 sv = reinterpret_cast<const float4*>(a)[o];
      if (b) {
        const float4 bv = reinterpret_cast<const float4*>(b)[o];
        sv.x -= bv.x; sv.y -= bv.y; sv.z -= bv.z; sv.w -= bv.w;
      }
      if (act != NG_ACT_NONE) {
        d.x *= act_grad_from_out(act, sv.x); d.y *= act_grad_from_out(act, sv.y);
        d.z *= act_grad_from_out(act, sv.z); d.w *= act_grad_from_out(act, sv.w);
      }
      reinterpret_cast<float4*>(dP)[o] = d;
      cs.x += d.x; cs.y += d.y; cs.z += d.z; cs.w += d.w;
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(d.x), fabsf(d.y))), fmaxf(fabsf(d.z), fabsf(d.w)));
    }
    *reinterpret_cast<float4*>(fc_red + r * No + 4 * q) = cs;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < No; it += 256) {
    float t = 0.f;
    for (int rr = 0; rr < RL; ++rr) t += fc_red[rr * No + it];
    partial[(int64_t)blockIdx.x * No + it] = t;
  }
  if (blockmax) block_max_store(amax, blockmax);
}
}  // namespace ng

extern "C" int ng_fc_block_fwd(ng_ctx* ctx, void* stream, int64_t N, int F, int L, int act, const float* x,
                               const float* const* W, const float* const* b, float* const* y, float* g) {
  using namespace ng;
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, L >= 2, "fc_block: at least two layers");
  NG_REQUIRE(ctx, F % 8 == 0, "fc_block: F % 8");
  NG_REQUIRE(ctx, y, "fc_block: y (layer outputs) required");
  if (fc_fused_supported(F, L)) return fc_fused_fwd(ctx, st, N, L, act, x, W, b, y, g);
  const float* cur = x;
  for (int l = 0; l < L - 1; ++l) {
    const int rc = dense_fwd(ctx, st, N, F, F, act, cur, W[l], b[l], nullptr, cur, y[l], nullptr);
    if (rc) return rc;
    cur = y[l];
  }
  return dense_fwd(ctx, st, N, F, F / 2, act, cur, W[L - 1], b[L - 1], nullptr, nullptr, g, nullptr);
}

extern "C" int ng_add_scaled(ng_ctx*, void*, int64_t, const float*, const float*, float, float*);
extern "C" int ng_dense_bwd(ng_ctx*, void*, int64_t, int, int, int, int, const float*, const float*, const float*,
                            const float*, float*, float*, float*);

extern "C" int64_t ng_fc_block_scratch_floats(int64_t N, int F, int L) {
  return ng::fc_fused_bwd_supported(F, L) ? 0 : 3 * N * (int64_t)F;
}

extern "C" int ng_fc_block_bwd(ng_ctx* ctx, void* stream, int64_t N, int F, int L, int act, const float* const* x,
                               const float* g, const float* const* W, const float* dg, float* dx,
                               float* const* dW, float* const* db, float* scratch) {
  using namespace ng;
  if (!ctx) return NG_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  NG_REQUIRE(ctx, L >= 2, "fc_block: at least two layers");
  if (N > 0 && fc_fused_bwd_supported(F, L)) return fc_fused_bwd(ctx, st, N, L, act, x, g, W, dg, dx, dW, db);
  // layer by layer.  Per layer ONE pass forms dP = dY * act'(s) (s = x_{l+1} - x_l rebuilt on the fly, or g for the last
  // layer) together with the bias gradient's column sums; the dX and dW GEMMs then read dP (before: a pass that wrote
  // s, then three consumers that each re-read dY and s).  scratch: [3][N][F] = dP | dX ping | dX pong
  NG_REQUIRE(ctx, scratch || N == 0, "fc_block_bwd: scratch [3*N*F] required for this feature size");
  NG_REQUIRE(ctx, F % 8 == 0 && F <= 1024, "fc_block_bwd: F % 8 == 0, F <= 1024");
  float* dP = scratch;
  float* d0 = scratch + N * F;
  float* d1 = d0 + N * F;
  const float* cur = dg;
  float* nxt = d0;
  for (int l = L - 1; l >= 0; --l) {
    const int No = l == L - 1 ? F / 2 : F;
    const bool resid = l < L - 1;
    if (N == 0) {
      NG_HIP(ctx, hipMemsetAsync(dW[l], 0, (size_t)F * No * 4, st));
      NG_HIP(ctx, hipMemsetAsync(db[l], 0, (size_t)No * 4, st));
      continue;
    }
    const int c4n = No / 4, rl = 256 / c4n > 0 ? 256 / c4n : 1;
    const int nb = (int)std::min<int64_t>(cdiv(N, rl), (int64_t)ctx->num_cu * 4);
    const int64_t rows = cdiv(cdiv(N, nb), rl) * rl;
    const int nblk = (int)cdiv(N, rows);
    float* partial = deferred_partials(ctx, (size_t)nblk * No);
    if (!partial) partial = (float*)aux_workspace(ctx, (size_t)nblk * No * 4);
    if (!partial) return NG_ERR_NOMEM;
    const float* gsc = nullptr;         // the same dP feeds both products: one scale, from the dP kernel's block maxima
    {
      ProfScope ps(ctx, st, "fc_dP");
      int cap = 0;
      float* bmax = dense_grad_uses_h2(N, F, No) ? gemm_grad_blockmax(ctx, &cap) : nullptr;
      hipLaunchKernelGGL(fc_dp_kernel, dim3(nblk), dim3(256), (size_t)rl * No * 4, st, N, No, rows, act, cur,
                         resid ? x[l + 1] : g, resid ? x[l] : nullptr, dP, partial, bmax);
      NG_HIP(ctx, hipGetLastError());
      const int rcr = reduce_or_defer(ctx, st, partial, nblk, (int64_t)No, db[l]);
      if (rcr) return rcr;
      if (bmax) {
        const int rcs = gemm_grad_scale_from_blocks(ctx, st, nblk, &gsc);
        if (rcs) return rcs;
      }
    }
    float* dwscr = (float*)workspace(ctx, dense_dw_scratch_floats(ctx, N, F, No, false) * sizeof(float));
    if (!dwscr) return NG_ERR_NOMEM;
    int rc = dense_dw(ctx, st, N, F, No, NG_ACT_NONE, x[l], dP, nullptr, nullptr, dW[l], nullptr, 0, 0, 0, dwscr, "dense_dw", gsc);
    if (rc) return rc;
    float* out = l == 0 ? dx : nxt;
    rc = dense_dx(ctx, st, N, F, No, NG_ACT_NONE, dP, nullptr, nullptr, W[l], resid ? cur : nullptr, out, "dense_dx", gsc);
    if (rc) return rc;
    cur = out;
    nxt = out == d0 ? d1 : d0;
  }
  return NG_OK;
}
