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
__device__ __forceinline__ void fc_unit2(const float (&wf)[16], const float* __restrict__ Xin, int rt0, int a16,
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

template <int NL>
__global__ __launch_bounds__(256, 2) void fc_fwd_kernel(FcFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float X0[FC_TM * FC_LD];
  __shared__ __attribute__((aligned(16))) float X1[FC_TM * FC_LD];
  __shared__ __attribute__((aligned(16))) float sB[NL * FC_F];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a16 = lane & 15, g4 = lane >> 4;
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
#define FC_FETCH(TILE)                                                                                   \
  do {                                                                                                   \
    const int64_t r_ = (TILE) * FC_TM + (tid >> 4);                                                      \
    const float4* x4_ = reinterpret_cast<const float4*>(a.x) + (tid & 15);                               \
    px0 = x4_[(r_ < a.N ? r_ : a.N - 1) * 16];                                                           \
    px1 = x4_[(r_ + 16 < a.N ? r_ + 16 : a.N - 1) * 16];                                                \
    px2 = x4_[(r_ + 32 < a.N ? r_ + 32 : a.N - 1) * 16];                                                \
    px3 = x4_[(r_ + 48 < a.N ? r_ + 48 : a.N - 1) * 16];                                                \
  } while (0)
  FC_FETCH((int64_t)blockIdx.x);
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * FC_TM;
    {
      float* d_ = X0 + (tid >> 4) * FC_LD + 4 * (tid & 15);
      *reinterpret_cast<float4*>(d_) = px0;
      *reinterpret_cast<float4*>(d_ + 16 * FC_LD) = px1;
      *reinterpret_cast<float4*>(d_ + 32 * FC_LD) = px2;
      *reinterpret_cast<float4*>(d_ + 48 * FC_LD) = px3;
    }
    NG_LDS_BARRIER();
    FC_FETCH(tile + gridDim.x < ntiles ? tile + gridDim.x : tile);
    float* Xin = X0;
    float* Xout = X1;
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
      const int col = 16 * wave + 4 * g4;
      const float4 bias = *reinterpret_cast<const float4*>(sB + l * FC_F + col);
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        f32x4 acc0, acc1;
        fc_unit2(wf[l], Xin, 2 * rp, a16, g4, bias, acc0, acc1);
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
      const int col = 16 * (wave & 1) + 4 * g4;
      const float4 bias = *reinterpret_cast<const float4*>(sB + (NL - 1) * FC_F + col);
      f32x4 acc0, acc1;
      fc_unit2(wf[NL - 1], Xin, 2 * (wave >> 1), a16, g4, bias, acc0, acc1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 16 * (2 * (wave >> 1) + h) + a16;
        const float4 s = act4(a.act, h ? acc1 : acc0);
        const int64_t row = row0 + r;
        *reinterpret_cast<float4*>(row < a.N ? a.g + row * FC_H + col : a.dummy + col) = s;
      }
    }
    NG_LDS_BARRIER();      // the next tile overwrites X0 (and X1 after its first layer)
  }
}

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
};

__host__ __device__ inline int fc_part_floats(int L) { return L * FC_F * FC_F + L * FC_F; }

// rows of the tile contracted by MFMA k-step T (0..15), lane group g: conflict-free for row strides == 4 mod 16
__device__ __forceinline__ int kstep_row(int T, int g) { return (T & 3) + 4 * g + 16 * (T >> 2); }

template <int NL>
__global__ __launch_bounds__(512, 1) void fc_bwd_kernel(FcBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* D0 = smem;
  float* D1 = D0 + FC_TM * FC_LD;
  float* Pt = D1 + FC_TM * FC_LD;
  float* XA = Pt + FC_TM * FC_LD;
  float* XB = XA + FC_TM * FC_LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a16 = lane & 15, g4 = lane >> 4;
  const int kt = wave & 3, pr = wave >> 2;      // k-tile, pair index (n-tiles / row tiles 2pr, 2pr+1)

  float wb[NL][16];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const float4* p = reinterpret_cast<const float4*>(a.Wb) + ((l * 4 + kt) * 4) * 64 + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      const float4 v = p[T * 64];
      wb[l][4 * T + 0] = v.x; wb[l][4 * T + 1] = v.y; wb[l][4 * T + 2] = v.z; wb[l][4 * T + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(wb[l][i]));
  }
  f32x4 accW[NL][2];
  float4 accb[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    accW[l][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    accW[l][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    accb[l] = f4zero();
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
          if (a.act != NG_ACT_NONE) {
            p.x *= act_grad_from_out(a.act, s.x); p.y *= act_grad_from_out(a.act, s.y);
            p.z *= act_grad_from_out(a.act, s.z); p.w *= act_grad_from_out(a.act, s.w);
          }
        }
        *reinterpret_cast<float4*>(Pt + o) = p;
        accb[l].x += p.x; accb[l].y += p.y; accb[l].z += p.z; accb[l].w += p.w;
      }
      NG_LDS_BARRIER();
      // ---- dW_l[k][n] += sum_rows x[row][k] P[row][n]: D rows i = n, cols j = k
      if (!last || pr == 0) {
        const float* xb = Xt + 16 * kt + a16;
        const float* p0 = Pt + 16 * (2 * pr) + a16;
#pragma unroll
        for (int T = 0; T < 16; ++T) {
          const int ro = kstep_row(T, g4) * FC_LD;
          const float bv = xb[ro];
          const float a0 = p0[ro];
          const float a1 = p0[ro + 16];
          accW[l][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, accW[l][0], 0, 0, 0);
          accW[l][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, accW[l][1], 0, 0, 0);
        }
      }
      // ---- dX[row][k] = sum_n P[row][n] W_l[k][n] (+ D[row][k] for the residual layers) -> Dn
      {
        const float* q0 = Pt + (16 * (2 * pr) + a16) * FC_LD + 4 * g4;
        const float* q1 = q0 + 16 * FC_LD;
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        const int nT = last ? 2 : 4;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
          if (T < nT) {
            const float4 xa = *reinterpret_cast<const float4*>(q0 + 16 * T);
            const float4 xb4 = *reinterpret_cast<const float4*>(q1 + 16 * T);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[l][4 * T + 0], xa.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[l][4 * T + 0], xb4.x, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[l][4 * T + 1], xa.y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[l][4 * T + 1], xb4.y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[l][4 * T + 2], xa.z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[l][4 * T + 2], xb4.z, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[l][4 * T + 3], xa.w, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[l][4 * T + 3], xb4.w, c1, 0, 0, 0);
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
  float* red = smem;     // [32][NL*64]
#pragma unroll
  for (int l = 0; l < NL; ++l) *reinterpret_cast<float4*>(red + er * (NL * FC_F) + l * FC_F + 4 * ec) = accb[l];
  __syncthreads();
  for (int it = tid; it < NL * FC_F; it += 512) {
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += red[r * (NL * FC_F) + it];
    part[NL * FC_F * FC_F + it] = s;
  }
}

}  // namespace

bool fc_fused_supported(int F, int L) {
  if (sw().fc_layered) return false;
  return F == FC_F && L >= 2 && L <= FC_MAXL;
}

// the fused BACKWARD keeps every layer's dW block in registers: four layers fit (the reference's default, model.py:33), five
// spilled 32 B and six 192 B per lane — those depths take the layer-by-layer backward below (same tape: the layer inputs)
bool fc_fused_bwd_supported(int F, int L) { return fc_fused_supported(F, L) && L <= 4; }

size_t fc_fused_pack_floats(int L) { return (size_t)2 * L * FC_F * FC_F + 64; }

// the block's packed weights (forward + backward fragments + a dummy row): the cached image of the weights when the cache is
// on (shared by the forward and the backward of a step, refreshed behind Adam), else `scratch`
static float* fc_packed(ng_ctx* ctx, hipStream_t st, int L, const float* const* W, float* scratch, int* rc) {
  bool have = false;
  float* ws = (float*)cached_image(ctx, W[0], 5, fc_fused_pack_floats(L) * 4, &have);
  const bool cached = ws != nullptr;
  if (!ws) ws = scratch;
  *rc = NG_OK;
  if (!ws) return nullptr;      // not cacheable and the caller brought no scratch: it calls again with one
  if (!have) {
    PackJob j;
    j.kind = PK_FC; j.blocks = 32; j.i0 = L;
    for (int l = 0; l < L; ++l) j.src[l] = W[l];
    j.dst[0] = ws; j.dst[1] = ws + (size_t)L * FC_F * FC_F;
    *rc = pack_launch(ctx, st, j);
    if (*rc == NG_OK && cached) cache_set_job(ctx, W[0], 5, j);
  }
  return ws;
}

int fc_fused_fwd(ng_ctx* ctx, hipStream_t st, int64_t N, int L, int act, const float* x, const float* const* W,
                 const float* const* b, float* const* y, float* g) {
  if (N == 0) return NG_OK;
  float* scratch = ctx->wcache ? nullptr : (float*)workspace(ctx, fc_fused_pack_floats(L) * 4);
  int rc = NG_OK;
  float* ws = fc_packed(ctx, st, L, W, scratch, &rc);
  if (!ws) {      // (cache on, but this source is not cacheable)
    scratch = (float*)workspace(ctx, fc_fused_pack_floats(L) * 4);
    if (!scratch) return NG_ERR_NOMEM;
    ws = fc_packed(ctx, st, L, W, scratch, &rc);
  }
  if (rc) return rc;
  float* Wf = ws;
  FcFwdArgs a{};
  a.N = N; a.act = act; a.x = x; a.Wf = Wf; a.g = g; a.dummy = ws + (size_t)2 * L * FC_F * FC_F;
  for (int l = 0; l < L; ++l) {
    a.p.W[l] = W[l]; a.p.b[l] = b[l];
    a.p.y[l] = (y && l < L - 1) ? y[l] : nullptr;
  }
  const int64_t ntiles = cdiv(N, FC_TM);
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cu * 2);
  ProfScope ps(ctx, st, "fc_fused_fwd");
  switch (L) {
    case 2: hipLaunchKernelGGL((fc_fwd_kernel<2>), dim3(grid), dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL((fc_fwd_kernel<3>), dim3(grid), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((fc_fwd_kernel<4>), dim3(grid), dim3(256), 0, st, a); break;
    case 5: hipLaunchKernelGGL((fc_fwd_kernel<5>), dim3(grid), dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL((fc_fwd_kernel<6>), dim3(grid), dim3(256), 0, st, a); break;
  }
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
  int rc = NG_OK;
  float* pkd = fc_packed(ctx, st, L, W, ws, &rc);
  if (rc) return rc;
  float* Wb = pkd + (size_t)L * FC_F * FC_F;
  float* dummy = pkd + (size_t)2 * L * FC_F * FC_F;
  FcBwdArgs a{};
  a.N = N; a.act = act; a.g = g; a.dg = dg; a.Wb = Wb; a.dx = dx; a.partial = partial; a.part_stride = part;
  a.dummy = dummy;
  for (int l = 0; l < L; ++l) a.x[l] = x[l];
  const size_t lds = (size_t)5 * FC_TM * FC_LD * 4;
  {
    ProfScope ps(ctx, st, "fc_fused_bwd");
    switch (L) {
      case 2: hipLaunchKernelGGL((fc_bwd_kernel<2>), dim3(grid), dim3(512), lds, st, a); break;
      case 3: hipLaunchKernelGGL((fc_bwd_kernel<3>), dim3(grid), dim3(512), lds, st, a); break;
      case 4: hipLaunchKernelGGL((fc_bwd_kernel<4>), dim3(grid), dim3(512), lds, st, a); break;
      default: return fail(ctx, NG_ERR_UNSUPPORTED, "fc_fused_bwd: more than four layers take the layered backward");
    }
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
