// Weight-image packers as device functions (round 4).  Every kernel of the training step that multiplies by a weight
// matrix reads a PACKED copy of it — MFMA fragment order, fp16 piece planes — that used to be rebuilt by its own small
// launch in front of every use: fourteen launches of ~4.5 us per step although the weights change exactly once per step,
// inside the Adam kernel.  The packers are now bodies that ONE kernel (repack.hip) runs, either for a single image
// (first use, or the cache is off: pack_launch) or for every image the context has registered, in one launch behind
// ng_adam_step (repack_all).  A body works on block `bid` of `nb` blocks of 256 threads.
//
// Layouts are those documented at the consumers (edge_fwd_h2.hip, edge_bwd_h2.hip, edge_fused.hip, mp_win.hip,
// mp_win_bwd.hip, fc_fused.hip); the constants below are asserted equal to theirs where they include this file.
#pragma once
#include "h2_common.cuh"
#include "ng_internal.h"

namespace ng {

// (PackKind / PackJob: ng_common.h)

constexpr int PKB = 256;      // threads per block of every packer

namespace pk {

constexpr int FHd = 128;                       // edge hidden width (edge_fused.h: FH)
constexpr int H2_NCHUNKd = 7, H2_CHUNKd = 32 * 1024;
constexpr float WSCALE = 256.0f;               // 2^8: weight pieces are taken from 2^8 W
constexpr int WFd = 64;                        // window kernels' feature width
constexpr int FC_Fd = 64, FC_Hd = 32;

__host__ __device__ inline int h2_feat(int t, int s, int hf) { return (t & 3) + 16 * s + 8 * (t >> 2) + 4 * hf; }

// ---- PK_EDGE_H2 (edge_fwd_h2.hip): chunk c (0..5): layer c>>1, output blocks 2*(c&1) + {0,1}; chunk 6: output layer
//   fragment ((bo_l*4 + bi)*2 + s)*2 + p, 1 KB each, lane-linear 16 B per lane:
//   lane (row i = l&31, k-slot t) = piece_p( 2^8 W[k = 32 bi + h2_feat(t, s, l>>5)][n = 32 bo + i] )
__device__ __forceinline__ void edge_h2_img(int bid, int nb, int tid, const float* __restrict__ W0, const float* __restrict__ W1,
                                            const float* __restrict__ W2, const float* __restrict__ Wo, int E,
                                            unsigned* __restrict__ img) {
  for (int idx = bid * PKB + tid; idx < H2_NCHUNKd * 16 * 64; idx += nb * PKB) {   // (chunk, bo_l, bi, s, lane)
    const int lane = idx & 63, s = (idx >> 6) & 1, bi = (idx >> 7) & 3, bo_l = (idx >> 9) & 1, c = idx >> 10;
    const int i = lane & 31, hf = lane >> 5;
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = 32 * bi + h2_feat(t, s, hf);
      if (c < 6) {
        const float* W = (c >> 1) == 0 ? W0 : ((c >> 1) == 1 ? W1 : W2);
        v[t] = WSCALE * W[k * FHd + 32 * (2 * (c & 1) + bo_l) + i];
      } else {
        v[t] = (bo_l == 0 && i < E) ? WSCALE * Wo[k * E + i] : 0.f;
      }
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
    const int frag = ((bo_l * 4 + bi) * 2 + s) * 2;
    unsigned* dst = img + (size_t)c * (H2_CHUNKd / 4) + (size_t)frag * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { dst[j] = h[j]; dst[256 + j] = l[j]; }
  }
}

// ---- PK_EDGE_WT (edge_bwd_h2.hip): W^T fragments of the dZ GEMMs: [2 layers (W2, W3)][4 k-slabs][8 k-steps][2 pieces][1 KB]
//   lane (row k = 32 zk + (l&31), k-slot t) = piece_p( 2^8 W[k][n = 16 ks + 8 (l>>5) + t] )
// Behind the fragments (byte 64 K of the image): {nW2, nW3, nWo} = the largest absolute row sums of W2, W3, Wo — the growth bounds
// of the backward's power-of-two gradient scale (edge_bwd_h2.hip: Ranges), formed here because the weights are at hand;
// the kernel combines them with max|de| itself (round 4: one launch less per step than a separate scale kernel).
__device__ __forceinline__ void edge_wt_img(int bid, int nb, int tid, const float* __restrict__ W2, const float* __restrict__ W3,
                                            const float* __restrict__ Wo, int E, unsigned* __restrict__ img) {
  if (bid == nb - 1 && Wo) {      // the last block (it has no fragment work when nb = 17)
    __shared__ float red[3][PKB];
    float r2 = 0.f, r3 = 0.f, ro = 0.f;
    if (tid < FHd) {
      for (int n = 0; n < FHd; ++n) { r2 += fabsf(W2[tid * FHd + n]); r3 += fabsf(W3[tid * FHd + n]); }
      for (int n = 0; n < E; ++n) ro += fabsf(Wo[tid * E + n]);
    }
    red[0][tid] = r2; red[1][tid] = r3; red[2][tid] = ro;
    __syncthreads();
    for (int s = PKB / 2; s > 0; s >>= 1) {
      if (tid < s)
#pragma unroll
        for (int j = 0; j < 3; ++j) red[j][tid] = fmaxf(red[j][tid], red[j][tid + s]);
      __syncthreads();
    }
    if (tid < 3) reinterpret_cast<float*>(img)[(2 * 4 * 8 * 2 * 1024) / 4 + tid] = red[tid][0];
    return;
  }
  if (Wo) nb -= 1;
  for (int idx = bid * PKB + tid; idx < 2 * 4 * 8 * 64; idx += nb * PKB) {   // (L, zk, ks, lane)
    const int lane = idx & 63, ks = (idx >> 6) & 7, zk = (idx >> 9) & 3, L = idx >> 11;
    const float* W = L == 0 ? W2 : W3;
    const int k = 32 * zk + (lane & 31), n0 = 16 * ks + 8 * (lane >> 5);
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      split2_pair(WSCALE * W[k * FHd + n0 + 2 * j], WSCALE * W[k * FHd + n0 + 2 * j + 1], h[j], l[j]);
    unsigned* dst = img + (size_t)(((L * 4 + zk) * 8 + ks) * 2) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { dst[j] = h[j]; dst[256 + j] = l[j]; }
  }
}

// ---- PK_EDGE_F32 (edge_fused.hip):
// Wpk[layer][w][t][lane][s] = W[layer][k = 8t + 4*(lane>>5) + s][n = 32w + (lane&31)]   (forward)
// WpkT[layer][w][t][lane][s] = W[layer][k = 32w + (lane&31)][n = 8t + 4*(lane>>5) + s]  (dX = dP W^T)
__device__ __forceinline__ void edge_f32_frag(int bid, int nb, int tid, const float* __restrict__ W0, const float* __restrict__ W1,
                                              const float* __restrict__ W2, float* __restrict__ Wpk, float* __restrict__ WpkT) {
  const int per_layer = FHd * FHd;
  for (int idx = bid * PKB + tid; idx < 3 * per_layer; idx += nb * PKB) {
    const int layer = idx / per_layer;
    int r = idx % per_layer;
    const int s = r & 3; r >>= 2;
    const int lane = r & 63; r >>= 6;
    const int t = r & 15; r >>= 4;
    const int w = r;
    const float* Wl = layer == 0 ? W0 : (layer == 1 ? W1 : W2);
    const int ka = 8 * t + 4 * (lane >> 5) + s, na = 32 * w + (lane & 31);
    if (Wpk) Wpk[idx] = Wl[ka * FHd + na];
    if (WpkT) WpkT[idx] = Wl[na * FHd + ka];
  }
}

// ---- window kernels (mp_win.hip): weight fragments for v_mfma_f32_16x16x4_f32 (A operand: row i = lane & 15, k = lane >> 4)
// out[((ct*NT + T)*64 + lane)*4 + u] = Wsrc(k = 16T + 4(lane>>4) + u, o = 16ct + (lane&15))
// mode 0 (forward):       Wsrc(k = n*64 + l, o = m) = w[l][m][n]
// mode 1 (back to nodes): Wsrc(k = n*64 + m, o = l) = w[l][m][n]
// mode 2 (dA = dP Wp^T):  Wsrc(k = m, o = n*64 + l) = w[l][m][n]
__device__ __forceinline__ void mpw_f32(int bid, int nb, int tid, int E, int mode, const float* __restrict__ w, float* __restrict__ out) {
  const int KF = E * WFd;
  const int kdim = mode == 2 ? WFd : KF;
  const int odim = mode == 2 ? KF : WFd;
  const int NT = kdim / 16;
  const int total = kdim * odim;
  for (int idx = bid * PKB + tid; idx < total; idx += nb * PKB) {
    int r = idx;
    const int u = r & 3; r >>= 2;
    const int lane = r & 63; r >>= 6;
    const int T = r % NT, ct = r / NT;
    const int k = 16 * T + 4 * (lane >> 4) + u;
    const int o = 16 * ct + (lane & 15);
    float v;
    if (mode == 0) {
      v = w[((k % WFd) * WFd + o) * E + k / WFd];
    } else if (mode == 1) {
      v = w[(o * WFd + (k % WFd)) * E + k / WFd];
    } else {
      v = w[((o % WFd) * WFd + k) * E + o / WFd];
    }
    out[idx] = v;
  }
}

__device__ __forceinline__ bool mpw_out_of_range(float w256) { return !(fabsf(w256) < 65504.0f); }

// fragments for v_mfma_f32_16x16x32_f16 with two-piece operands, generic over the source index map:
// out[(((ct*NT2 + T)*2 + p)*64 + lane)*4 + j] = fp16 pair (t = 2j, 2j+1) of piece p of 2^8 Wsrc(k = 32T + 8(lane>>4) + t,
// o = 16ct + (lane&15));  MODE as in mpw_f32 (0: NT2 = KF/32, 4 column tiles; 1: the same shape; 2: NT2 = 2, KF/16 tiles)
template <int MODE>
__device__ __forceinline__ bool mpw_h2(int bid, int nb, int tid, int E, const float* __restrict__ w, unsigned* __restrict__ out) {
  const int KF = E * WFd;
  const int NT2 = MODE == 2 ? 2 : KF / 32;
  const int nct = MODE == 2 ? KF / 16 : 4;
  bool bad = false;
  for (int idx = bid * PKB + tid; idx < nct * NT2 * 64; idx += nb * PKB) {   // (ct, T, lane)
    const int lane = idx & 63, T = (idx >> 6) % NT2, ct = (idx >> 6) / NT2;
    const int o = 16 * ct + (lane & 15);
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k0 = 32 * T + 8 * (lane >> 4) + 2 * j, k1 = k0 + 1;
      float v0, v1;
      if (MODE == 0) {
        v0 = WSCALE * w[((k0 % WFd) * WFd + o) * E + k0 / WFd]; v1 = WSCALE * w[((k1 % WFd) * WFd + o) * E + k1 / WFd];
      } else if (MODE == 1) {
        v0 = WSCALE * w[(o * WFd + (k0 % WFd)) * E + k0 / WFd]; v1 = WSCALE * w[(o * WFd + (k1 % WFd)) * E + k1 / WFd];
      } else {
        v0 = WSCALE * w[((o % WFd) * WFd + k0) * E + o / WFd]; v1 = WSCALE * w[((o % WFd) * WFd + k1) * E + o / WFd];
      }
      bad |= mpw_out_of_range(v0) || mpw_out_of_range(v1);
      split2_pair(v0, v1, h[j], l[j]);
    }
    unsigned* d = out + ((size_t)((ct * NT2 + T) * 2) * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = h[j]; d[256 + j] = l[j]; }
  }
  return bad;
}

// ---- PK_FC (fc_fused.hip)
// forward fragments:  Wf[((l*4 + ct)*4 + T)*64 + lane][u] = W_l[k = 16T + 4(lane>>4) + u][n = 16ct + (lane&15)]
// backward fragments: Wb[((l*4 + kt)*4 + T)*64 + lane][u] = W_l[k = 16kt + (lane&15)][n = 16T + 4(lane>>4) + u]
// (columns n >= n_out of the last layer are zero)
__device__ __forceinline__ void fc_frag(int bid, int nb, int tid, int L, const float* const (&W)[6], float* __restrict__ Wf,
                                        float* __restrict__ Wb) {
  const int per_layer = FC_Fd * FC_Fd;
  for (int idx = bid * PKB + tid; idx < L * per_layer; idx += nb * PKB) {
    const int l = idx / per_layer;
    int r = idx % per_layer;
    const int u = r & 3; r >>= 2;
    const int lane = r & 63; r >>= 6;
    const int T = r & 3; r >>= 2;
    const int ct = r;
    const int nout = l == L - 1 ? FC_Hd : FC_Fd;
    const float* Wl = W[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) Wl = l == q ? W[q] : Wl;      // (no run-time index into the argument array)
    {
      const int k = 16 * T + 4 * (lane >> 4) + u, n = 16 * ct + (lane & 15);
      Wf[idx] = n < nout ? Wl[k * nout + n] : 0.f;
    }
    {
      const int k = 16 * ct + (lane & 15), n = 16 * T + 4 * (lane >> 4) + u;
      Wb[idx] = n < nout ? Wl[k * nout + n] : 0.f;
    }
  }
}

// fp16 piece fragments of the FC block for v_mfma_f32_16x16x32_f16 (round 4; fc_fused.hip: the H2 bodies), 2^8 W_l:
//   WfH[((((l*4 + ct)*2 + T)*2 + p)*64 + lane)*4 + j] = pair (k = 32T + 8(lane>>4) + 2j, +1) of piece p, n = 16ct + (lane&15)
//   WbH[((((l*4 + kt)*2 + T)*2 + p)*64 + lane)*4 + j] = pair (n = 32T + 8(lane>>4) + 2j, +1) of piece p, k = 16kt + (lane&15)
// (columns n >= n_out of the last layer are zero).  Returns true when a weight leaves the piece range.
__device__ __forceinline__ bool fc_h2(int bid, int nb, int tid, int L, const float* const (&W)[6], unsigned* __restrict__ WfH,
                                      unsigned* __restrict__ WbH) {
  bool bad = false;
  for (int idx = bid * PKB + tid; idx < L * 512; idx += nb * PKB) {      // (l, ct, T, lane)
    const int lane = idx & 63, T = (idx >> 6) & 1, ct = (idx >> 7) & 3, l = idx >> 9;
    const int nout = l == L - 1 ? FC_Hd : FC_Fd;
    const float* Wl = W[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) Wl = l == q ? W[q] : Wl;
    const int i16 = 16 * ct + (lane & 15), s0 = 32 * T + 8 * (lane >> 4);
    unsigned fh[4], fl[4], bh[4], bl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sa = s0 + 2 * j, sb = sa + 1;
      // forward: k = slot, n = i16;  backward: k = i16, n = slot
      const float f0 = i16 < nout ? WSCALE * Wl[sa * nout + i16] : 0.f, f1 = i16 < nout ? WSCALE * Wl[sb * nout + i16] : 0.f;
      const float b0 = sa < nout ? WSCALE * Wl[i16 * nout + sa] : 0.f, b1 = sb < nout ? WSCALE * Wl[i16 * nout + sb] : 0.f;
      bad |= mpw_out_of_range(f0) || mpw_out_of_range(f1) || mpw_out_of_range(b0) || mpw_out_of_range(b1);
      split2_pair(f0, f1, fh[j], fl[j]);
      split2_pair(b0, b1, bh[j], bl[j]);
    }
    const size_t o = ((size_t)(((l * 4 + ct) * 2 + T) * 2) * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { WfH[o + j] = fh[j]; WfH[o + 256 + j] = fl[j]; WbH[o + j] = bh[j]; WbH[o + 256 + j] = bl[j]; }
  }
  return bad;
}

// ---- PK_GG (gemm_h2.hip: mp_gg_kernel): fp16 piece fragments of an MPLayer weight w[l][m][n] in the k-step order of the
// gather-GEMM.  Step s = c32 * E + n covers the 32 gathered columns c = 32 c32 .. + 31 of edge feature n; output column o:
//   mode 0 (forward update):  value(c, o) = w[l = c][m = o][n]        (A[i][(n, l)] = sum_j e_n h[nbr_j][l])
//   mode 1 (pull, backward):  value(c, o) = w[l = o][m = c][n]        (B[i][(n, m)] = sum_in e_n dP[src][m])
// img[s][nb][ks][p][lane][8 fp16]: lane (o = 32 nb + (l&31), k-slot t) = piece_p( 2^8 value(32 c32 + 16 ks + 8 (l>>5) + t, o) )
__device__ __forceinline__ void gg_img(int bid, int nb_, int tid, int E, int F, int mode, const float* __restrict__ w,
                                       unsigned* __restrict__ img) {
  const int steps = (F / 32) * E, NB = F / 32;
  for (int idx = bid * PKB + tid; idx < steps * NB * 2 * 64; idx += nb_ * PKB) {   // (s, nb, ks, lane)
    const int lane = idx & 63, ks = (idx >> 6) & 1;
    const int nb = (idx >> 7) % NB, s = (idx >> 7) / NB;
    const int n = s % E, c32 = s / E;
    const int o = 32 * nb + (lane & 31), c0 = 32 * c32 + 16 * ks + 8 * (lane >> 5);
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ca = c0 + 2 * j, cb = ca + 1;
      const float va = mode == 0 ? w[((int64_t)ca * F + o) * E + n] : w[((int64_t)o * F + ca) * E + n];
      const float vb = mode == 0 ? w[((int64_t)cb * F + o) * E + n] : w[((int64_t)o * F + cb) * E + n];
      split2_pair(WSCALE * va, WSCALE * vb, h[j], l[j]);
    }
    unsigned* dst = img + (size_t)s * (NB * 2 * 2 * 256) + ((nb * 2 + ks) * 2) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { dst[j] = h[j]; dst[256 + j] = l[j]; }
  }
}

// ---- PK_GX (gemm_h2.hip: gx_launch): fp16 piece image of the weight operand of a [M][K] x [K][N] product,
//   image[(ct * KT + kt)][nb][ks][p][lane][8 fp16]:  lane (row n = BN ct + 32 nb + (l&31), k-slot t) =
//     piece_p( 2^8 B[k = 32 kt + 16 ks + 8 (l>>5) + t][n] ),   BN = 64 nbw
// mode 0: B[k][n] = W[k][n];  mode 1: W is stored [N][K] (dX = dP W^T);
// mode 2 / 3: the same two with W = Wp[ne F + l][m] = w[l][m][ne], the GEMM form of an MPLayer weight (PK_MP_PLAIN), read from w
// itself: the image does not wait for the plain copy, so both are rebuilt in the one launch behind the weight update
constexpr int GXP_BK = 32;
constexpr float GXP_WSCALE = 256.0f;
__device__ __forceinline__ float gx_src(const float* __restrict__ W, int mode, int K, int N, int k, int n) {
  switch (mode) {
    case 0: return W[(int64_t)k * N + n];
    case 1: return W[(int64_t)n * K + k];
    case 2: { const int F = N, E = K / N; return W[((int64_t)(k % F) * F + n) * E + k / F]; }        // Wp [K = E F][N = F]
    default: { const int F = K, E = N / K; return W[((int64_t)(n % F) * F + k) * E + n / F]; }       // Wp stored [N = E F][K = F]
  }
}
__device__ __forceinline__ void gx_img(int bid, int nblk, int tid, int K, int N, int mode, int nbw, const float* __restrict__ W,
                                       unsigned* __restrict__ img) {
  const int KT = K / GXP_BK;
  const int BN = 64 * nbw, NB = 2 * nbw;           // columns / 32-column blocks per tile
  const int64_t total = (int64_t)(N / BN) * KT * NB * 2 * 64;
  for (int64_t idx = (int64_t)bid * PKB + tid; idx < total; idx += (int64_t)nblk * PKB) {    // (ct, kt, nb, ks, lane)
    const int lane = idx & 63, ks = (idx >> 6) & 1;
    const int nb = (int)((idx >> 7) % NB);
    const int kt = (int)(((idx >> 7) / NB) % KT), ct = (int)(((idx >> 7) / NB) / KT);
    const int n = BN * ct + 32 * nb + (lane & 31), k0 = GXP_BK * kt + 16 * ks + 8 * (lane >> 5);
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      split2_pair(GXP_WSCALE * gx_src(W, mode, K, N, k0 + 2 * j, n), GXP_WSCALE * gx_src(W, mode, K, N, k0 + 2 * j + 1, n), h[j], l[j]);
    unsigned* dst = img + ((int64_t)(ct * KT + kt) * (NB * 2 * 2 * 256)) + ((nb * 2 + ks) * 2) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { dst[j] = h[j]; dst[256 + j] = l[j]; }
  }
}

// ---- PK_MP_PLAIN (node_ops.hip / mp_csr.hip): w[l][m][n] (reference layout, n fastest) -> Wp[k = n F + l][m]
__device__ __forceinline__ void mp_plain(int bid, int nblk, int tid, int F, int E, const float* __restrict__ w, float* __restrict__ Wp) {
  const int total = F * F * E;
  for (int idx = bid * PKB + tid; idx < total; idx += nblk * PKB) {
    const int k = idx / F, m = idx % F;
    const int n = k / F, l = k % F;
    Wp[idx] = w[((int64_t)l * F + m) * E + n];
  }
}

}  // namespace pk

// one block of one job
__device__ __forceinline__ void pack_job_block(const PackJob& j, int bid, int tid, unsigned ver) {
  bool bad = false;
  switch (j.kind) {
    case PK_EDGE_H2:
      pk::edge_h2_img(bid, j.blocks, tid, j.src[0], j.src[1], j.src[2], j.src[3], j.i0, (unsigned*)j.dst[0]);
      break;
    case PK_EDGE_WT:
      pk::edge_wt_img(bid, j.blocks, tid, j.src[0], j.src[1], j.src[2], j.i0, (unsigned*)j.dst[0]);
      break;
    case PK_EDGE_F32:
      pk::edge_f32_frag(bid, j.blocks, tid, j.src[0], j.src[1], j.src[2], (float*)j.dst[0], (float*)j.dst[1]);
      break;
    case PK_MPW_FWD: {
      // blocks: [0, 24) piece image, [24, 48) the f32 image (when there is one)
      if (bid < 24) bad = pk::mpw_h2<0>(bid, 24, tid, j.i0, j.src[0], (unsigned*)j.dst[0]);
      else pk::mpw_f32(bid - 24, 24, tid, j.i0, 0, j.src[0], (float*)j.dst[1]);
    } break;
    case PK_MPW_BWD: {
      // blocks: [0, 24) T pieces (they see every weight), [24, 48) N pieces, [48, 72) f32 T, [72, 96) f32 N
      const int part = bid / 24, b = bid % 24;
      if (part == 0) bad = pk::mpw_h2<2>(b, 24, tid, j.i0, j.src[0], (unsigned*)j.dst[0]);
      else if (part == 1) (void)pk::mpw_h2<1>(b, 24, tid, j.i0, j.src[0], (unsigned*)j.dst[1]);
      else if (part == 2) pk::mpw_f32(b, 24, tid, j.i0, 2, j.src[0], (float*)j.dst[2]);
      else pk::mpw_f32(b, 24, tid, j.i0, 1, j.src[0], (float*)j.dst[3]);
    } break;
    case PK_MPW_F32:
      pk::mpw_f32(bid, j.blocks, tid, j.i0, j.i1, j.src[0], (float*)j.dst[0]);
      break;
    case PK_FC: {
      const float* const W[6] = {j.src[0], j.src[1], j.src[2], j.src[3], j.src[4], j.src[5]};
      pk::fc_frag(bid, j.blocks, tid, j.i0, W, (float*)j.dst[0], (float*)j.dst[1]);
      if (j.dst[2]) bad = pk::fc_h2(bid, j.blocks, tid, j.i0, W, (unsigned*)j.dst[2], (unsigned*)j.dst[3]);
    } break;
    case PK_GG:
      pk::gg_img(bid, j.blocks, tid, j.i0, j.i1 & 0xFFFF, j.i1 >> 16, j.src[0], (unsigned*)j.dst[0]);
      break;
    case PK_GX:
      pk::gx_img(bid, j.blocks, tid, j.i0, j.i1 & 0xFFFFF, (j.i1 >> 20) & 15, j.i1 >> 24, j.src[0], (unsigned*)j.dst[0]);
      break;
    case PK_MP_PLAIN:
      pk::mp_plain(bid, j.blocks, tid, j.i0, j.i1, j.src[0], (float*)j.dst[0]);
      break;
    default: break;
  }
  // flag word pair of an image kept over calls: [0] = the version at which a weight was last found outside the piece range,
  // [1] = the version of the image itself (written by the job's first block, every time).  A consumer takes its fp32 body
  // when the two are equal (ng_internal.h: wimage_flag_raised) — both words come from memory, so a launch recorded in a HIP
  // graph (whose `ver` argument is frozen at capture) and its recorded consumers still agree at every replay (round-5
  // advisor finding: the consumers compared against a version passed as a kernel argument).
  if (j.flag && bid == 0 && tid == 0) j.flag[1] = ver;
  if (bad) {
    if (j.flag) *j.flag = ver;
    if (j.guard.word) range_guard_raise(j.guard, true);
  }
}

}  // namespace ng
