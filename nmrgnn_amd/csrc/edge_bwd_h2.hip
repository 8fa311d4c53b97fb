// Fused persistent backward of the edge path on the fp16 matrix pipe with two-piece split fp32 operands
// (h2_common.cuh: three piece products per multiply).  Default; same math, phases, workgroup partials and reduction
// kernel as edge_fused_bwd.hip (f32-input MFMA, NG_EDGE_MATH=fp32).  Backward of nmrgnn/model.py:251-261, SURVEY App. B.
// (Until late round 2: exact three-piece bf16 split, six piece products, images 152 KB.)
//
//   A:  dE = S*m*de ;  G3 = (dE Wo^T) * s'(Z3)   [VALU]      dWo += Z3^T dE, dbo += sum dE   [VALU, fp32]
//   B:  dW3 += Z2^T G3 ; db3 += colsum G3 ; dZ2 = G3 W3^T ; G2 = dZ2 * s'(Z2)
//   C:  dW2 += Z1^T G2 ; db2 += colsum G2 ; dZ1 = G2 W2^T ; G1 = dZ1 * s'(Z1)
//   D:  R = m*rbf(d) recomputed ; dW1 += R^T G1 ; db1 += colsum G1
//
// Ranges.  fp16 pieces hold |x| < 65504 and resolve absolute steps of 2^-25 (subnormal l pieces are honoured by the
// MFMA).  Activations (Z, R: O(1)) are split unscaled.  Gradients can be arbitrarily small (loss scaling, 1/G), so the
// whole backward — linear in dE — runs on S*dE with S a power of two chosen per call (kernel prologue; until round 4 a launch of its own) from
//   max|de| * max(nWo, nWo nW3, nWo nW3 nW2),  n. = largest absolute row sum of the matrix a gradient passes through,
// the worst case any G entry can reach: S puts that bound at 2^15, so no piece can overflow, and typical entries
// (orders of magnitude below the bound, still >> 2^-14) keep full two-piece precision.  The partials are multiplied
// by 1/S when they are written.  W^T pieces are taken from 2^8 W (the l piece of a typical weight would otherwise be
// subnormal); the dZ epilogue multiplies by 2^-8.
//
// Mapping.  512 threads = 8 waves (2 per SIMD), one persistent workgroup per CU, 64-edge tiles.
// Every GEMM operand that comes from activations lives in LDS as an fp16-piece IMAGE [2 pieces][64 edges][136]:
// four images (Z-type x 2, G ping, G pong) = 134 KB; with the second Z-type image a tile needs FOUR barriers
// (after the staging writes, after phase A, after phase B, after phase C) instead of the seven of the
// three-image schedule.  Elementwise work is done ONCE per element in the accumulator
// layout of the dZ GEMM (lane = edge row, 16 columns of the wave's 32-column slab): the lane that loads Z_l from
// HBM in that layout splits it into the image AND keeps the fp32 values for s'(Z_l) in its own epilogue.
//   dZ GEMM (wave: k-slab zk = w&3, edge half zrt = w>>2): A = W^T pieces streamed from a fragment-ordered image
//     in L2 (buffer loads), B = G pieces read as rows of the image (ds_read_b128), 24 MFMAs.
//   dW GEMM (wave: k-slab w>>1, n-slabs 2(w&1)+{0,1}; contraction over the tile's 64 edges): both operands are
//     COLUMNS of an image; ds_read_b64_tr_b16 delivers a lane 4 consecutive edges of its column (two reads = one
//     8-edge MFMA operand), 24 MFMAs per layer.  The three 128x128 accumulators stay in registers (96 VGPRs).
//   Bias gradients ride on the dW GEMM: G^T x ones in one extra accumulator (hx_dw_gemm).
#include "edge_bwd_h2.cuh"

namespace ng {


#ifdef HX_STAMP
#define HX_T(k)                                                                              \
  do {                                                                                       \
    if (lane == 0 && (wave & 3) == 0 && titer >= 2 && titer < 6)                             \
      sStamp[((wave >> 2) * 4 + (titer - 2)) * 16 + (k)] = __builtin_readcyclecounter();     \
  } while (0)
#else
#define HX_T(k)
#endif

template <bool LIVE>
__global__ __launch_bounds__(HX_THREADS, 1) void edge_bwd_h2_kernel(EdgeBwdH2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_hx[];
  // images: Z-type x at smem + x * HX_IMG_Z, G-type x at smem + 2 * HX_IMG_Z + x * HX_IMG_G (roles alternate per tile)
  float* sWo4 = reinterpret_cast<float*>(smem_hx + HX_IMGS);      // [128][4]
  float* sdE = sWo4 + FH * 4;         // [64][4]
  float* sCen = sdE + FTM * 4;        // [128]
#ifdef HX_STAMP
  unsigned long long* sStamp = reinterpret_cast<unsigned long long*>(sCen + FH);   // [2][4][16]
  int titer = -1;
#endif

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kslab = wave >> 1, nsl0 = 2 * (wave & 1);   // dW blocks
  const int zk = wave & 3, zrt = wave >> 2;             // dZ block / elementwise ownership
  const int cn = tid & 127, rq = tid >> 7;              // dWo ownership
  const int row = 32 * zrt + l31;                       // this lane's edge row in the tile
  const int col0 = 32 * zk + 4 * half;                  // its columns: col0 + 8q + j
  const int prz = hx_prow_z(row), prg = hx_prow_g(row); // where that row lives in the images
  const int E = a.E;
  // a NEGATIVE live row count: nothing to do and nothing to write — the reduction behind this launch then writes zero gradients
  // without reading the partials (the edge-function table's guard is down, edge_table.hip)
  if (LIVE && __builtin_amdgcn_readfirstlane(*a.n_live) < 0) return;
  // gradients run scaled by a power of two S (header: Ranges), formed here by every workgroup from the block maxima of |de| and the
  // row-sum bounds the W^T pack left behind the fragments: max is exact and order-free, so S — and every bit of the result —
  // is the same in every workgroup and for every launch geometry (round 4: this was a one-block launch of its own)
  float gscale, ginv;
  {
    float* sred = reinterpret_cast<float*>(smem_hx);
    float m = 0.f;
    if (a.n_blockmax >= 0) {
      for (int i = tid; i < a.n_blockmax; i += HX_THREADS) m = fmaxf(m, a.blockmax[i]);
    } else {        // molecule-sized call: max |de| directly (order-free: the value of the two-stage form)
      const int n = -a.n_blockmax;
      const float4* de4 = reinterpret_cast<const float4*>(a.blockmax);
      for (int i = tid; i < n >> 2; i += HX_THREADS) {
        const float4 v = de4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
      for (int i = (n & ~3) + tid; i < n; i += HX_THREADS) m = fmaxf(m, fabsf(a.blockmax[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) sred[wave] = m;
    __syncthreads();
    m = sred[0];
#pragma unroll
    for (int w8 = 1; w8 < HX_THREADS / 64; ++w8) m = fmaxf(m, sred[w8]);
    const float* wb = reinterpret_cast<const float*>(a.wt_img + 2 * 4 * 8 * 2 * 1024);      // {nW2, nW3, nWo}
    const float b3 = m * wb[2], b2 = b3 * wb[1], b1 = b2 * wb[0];
    const float bound = fmaxf(b3, fmaxf(b2, b1));
    int ex = 0;
    // a NaN / inf gradient keeps S = 1 and propagates; an all-zero one too
    if (bound > 0.f && bound < 3.0e38f) {
      int eb;
      (void)frexpf(bound, &eb);              // bound = f * 2^eb, f in [0.5, 1)  ->  bound <= 2^eb
      ex = 15 - eb;
      ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
    }
    gscale = ldexpf(1.0f, ex);
    ginv = ldexpf(1.0f, -ex);
    __syncthreads();
  }
  // rows of this launch (LIVE: the segment's share of the compacted live rows, a device scalar)
  const int64_t n_edges = LIVE ? std::max<int64_t>(0, std::min<int64_t>(a.n_edges, (int64_t)*a.n_live - a.row_base)) : a.n_edges;
  // tape offsets of this wave's block for the tile starting at ROW0 (see hx_load_z)
#define HX_ZFULL(ROW0) (a.tape_blocked && (ROW0) + 32 * zrt + 32 <= n_edges)
#define HX_ZOFF(ROW0, GI) (HX_ZFULL(ROW0) ? (int)(((ROW0) / 32 + zrt) * 16384 + (zk * 256 + lane) * 16) : (GI) * (FH * 4) + col0 * 4)
#define HX_ZQ(ROW0) (HX_ZFULL(ROW0) ? 1024 : 32)

  // Wo for the G3 product, stored as column PAIRS [c/2][n][c&1]: a 16-B read delivers (Wo[c][n], Wo[c+1][n]) side by
  // side for two n, the operand shape of v_pk_fma_f32 (row-major [c][4] cost six v_mov per column pair to rearrange)
  for (int t = tid; t < FH * 4; t += HX_THREADS) {
    const int c = t >> 2, n = t & 3;
    sWo4[((c >> 1) * 4 + n) * 2 + (c & 1)] = n < E ? a.Wo[c * E + n] : 0.f;
  }
  if (tid < FH) sCen[tid] = a.centers[tid];

  f32x16 accW[3][2];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[l][j][r] = 0.f;
  f32x16 accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = 0.f;
  float accWo[4] = {0.f, 0.f, 0.f, 0.f};
  float accbo[4] = {0.f, 0.f, 0.f, 0.f};     // sum of this lane's row of dE over the tiles (8 lanes hold each row)

  const int64_t ntiles = (n_edges + FTM - 1) / FTM;
  // one buffer resource per array (offsets are 32-bit: n_edges * 512 B < 4 GB, checked by the host)
  const unsigned zbytes = (unsigned)(n_edges * FH * 4);
  const __amdgpu_buffer_rsrc_t rsZ1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z_save), 0, zbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsZ2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z_save + a.z_layer_stride), 0, zbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsZ3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z_save + 2 * a.z_layer_stride), 0, zbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsDs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d_src), 0, (unsigned)(n_edges * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsDn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d_eff), 0, (unsigned)(n_edges * 4), 0x00020000);
  // LIVE: de is the caller's whole [n_slots][E] array, addressed by slot (n_slots * E * 4 < 2^32: checked by the host)
  const __amdgpu_buffer_rsrc_t rsDe = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.de), 0, LIVE ? 0xFFFFFFFFu : (unsigned)(n_edges * a.E * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsPm = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.perm), 0, LIVE ? (unsigned)(n_edges * 4) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wt_img), 0, 2 * 4 * 8 * 2 * 1024, 0x00020000);
  __syncthreads();

  // per-tile inputs of this lane's row, requested one tile ahead
  float z3r[16], z2r[16], pf_ds = 1.f, pf_dn, pf_de[4];
  // LIVE: the slot of this lane's row, requested TWO tiles ahead (by the prefetch of the tile before): as an address of
  // the de loads it must be in a register when they are issued — a load that waits for another load inside the prefetch
  // would stall in front of the GEMM the prefetch hides under
  int pf_slot = 0;
  auto load_slot = [&](int64_t row0) {
    const int gi = (int)std::max<int64_t>(std::min<int64_t>(row0 + row, n_edges - 1), 0);
    pf_slot = (int)__builtin_amdgcn_raw_buffer_load_b32(rsPm, gi * 4, 0, 0);
  };
  // Nothing here may USE a loaded value (no select, no conversion): a use inside this block makes the compiler
  // wait for the HBM loads right behind their issue, in front of the GEMM they are meant to hide under.  Indices are
  // clamped, the masks are applied at the point of use in the next iteration.
  auto prefetch = [&](int64_t row0, int64_t row0_after) {
    const int64_t gr = std::min<int64_t>(row0 + row, n_edges - 1);
    const int gi = (int)gr;
    hx_load_z(z3r, rsZ3, HX_ZOFF(row0, gi), HX_ZQ(row0));
    if (!LIVE) pf_ds = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsDs, gi * 4, 0, 0));
    pf_dn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsDn, gi * 4, 0, 0));
    const int ge = LIVE ? pf_slot : gi;
#pragma unroll
    for (int n = 0; n < 4; ++n)
      pf_de[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsDe, (ge * E + std::min(n, E - 1)) * 4, 0, 0));
    if (LIVE) load_slot(row0_after);
  };
  // Z2 of the next tile: its registers are free only after phase B's epilogue
  auto prefetch_z2 = [&](int64_t row0) {
    const int gi = (int)std::min<int64_t>(row0 + row, n_edges - 1);
    hx_load_z(z2r, rsZ2, HX_ZOFF(row0, gi), HX_ZQ(row0));
  };
  // first row of the tile `steps` grid strides after `tile`, clamped to the last tile
  auto row0_of = [&](int64_t tile, int steps) { return std::min<int64_t>(tile + (int64_t)steps * gridDim.x, ntiles - 1) * FTM; };
  if ((int64_t)blockIdx.x < ntiles) {
    if (LIVE) load_slot((int64_t)blockIdx.x * FTM);      // (the one place the slot load is waited for: once per launch)
    prefetch((int64_t)blockIdx.x * FTM, row0_of(blockIdx.x, 1));
    prefetch_z2((int64_t)blockIdx.x * FTM);
  }

  // ---- the head of a tile ("phase A": mask, dE, fp32 Z3 staging, dWo / dbo on the VALU, G3, the Z2 image) in three
  // steps separated by barriers.  For every tile but a workgroup's first it runs INSIDE phase D of the tile before
  // (round 3): phase D is a bare dW GEMM — matrix pipe and LDS reads, hardly any VALU — and this head is VALU / LDS
  // work without a single MFMA (4.8k of the 19.6k cycles of a tile when it stood alone).  Images alternate per tile:
  // tile t keeps {G3, G1} in G[gp] and {Z2, R} in Z[zp], {G2} in G[gp^1], {Z1} in Z[zp^1]; during its phase D the
  // buffers G[gp^1] / Z[zp^1] are dead and take the next tile's G3 and staging -> Z2.
  float dEm[4];
  // step 1: this lane's row of the tile whose inputs sit in pf_* / z3r; `live` = 0 on the pass past the last tile
  auto head1 = [&](int64_t row0, bool live, char* Zn, bool& on_o, float& dn_o) {
    on_o = live && pf_ds > 0.f && row0 + row < n_edges;
    dn_o = pf_dn;
#pragma unroll
    for (int n = 0; n < 4; ++n) dEm[n] = (on_o && n < E) ? gscale * pf_de[n] : 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) accbo[n] += dEm[n];
    float* stg = reinterpret_cast<float*>(Zn);        // fp32 [64][132]: exactly the footprint of a Z-type image
    if (zk == 0 && half == 0) *reinterpret_cast<float4*>(sdE + 4 * row) = make_float4(dEm[0], dEm[1], dEm[2], dEm[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(stg + row * HX_STG + col0 + 8 * q) =
          make_float4(z3r[4 * q + 0], z3r[4 * q + 1], z3r[4 * q + 2], z3r[4 * q + 3]);
  };
  // step 2 (after a barrier): dWo from the staging in Zn, G3 pieces -> Gn.  (The Z2 pieces follow in step 3, once every
  // reader of the staging is past the next barrier: z2r has to stay in registers for s'(Z2) anyway, G3 does not —
  // with the staging in the G buffer the sixteen G3 values were live across the second wave's dW GEMM and got spilled.)
  auto head2 = [&](char* Gn, const char* Zn) {
    const float* stg = reinterpret_cast<const float*>(Zn);
    // dWo[k][n] += sum_rows Z3[row][k] dE[row][n]   (thread: k = cn, rows 16rq..16rq+15), fp32 on the VALU
#pragma unroll 4
    for (int r = 16 * rq; r < 16 * rq + 16; ++r) {
      const float z = stg[r * HX_STG + cn];
      const float4 d = *reinterpret_cast<const float4*>(sdE + 4 * r);
      accWo[0] += z * d.x; accWo[1] += z * d.y; accWo[2] += z * d.z; accWo[3] += z * d.w;
    }
    // G3 = (dE Wo^T) * s'(Z3)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const float4 w01 = *reinterpret_cast<const float4*>(sWo4 + 4 * (col0 + 8 * q + j));      // (x0 x1 y0 y1)
        const float4 w23 = *reinterpret_cast<const float4*>(sWo4 + 4 * (col0 + 8 * q + j) + 4);  // (z0 z1 w0 w1)
        float p0 = w01.x * dEm[0], p1 = w01.y * dEm[0];
        p0 = fmaf(w01.z, dEm[1], p0); p1 = fmaf(w01.w, dEm[1], p1);
        p0 = fmaf(w23.x, dEm[2], p0); p1 = fmaf(w23.y, dEm[2], p1);
        p0 = fmaf(w23.z, dEm[3], p0); p1 = fmaf(w23.w, dEm[3], p1);
        z3r[4 * q + j] = hx_sprime(p0, z3r[4 * q + j]);
        z3r[4 * q + j + 1] = hx_sprime(p1, z3r[4 * q + j + 1]);
      }
    hx_img_write<HX_ROWG>(Gn, prg, col0, z3r);       // G3 pieces
  };

  bool on = false;
  float dn = 0.f;
  u32x4 w0[2];
  // Z1 of a tile is requested during the phase D of the tile BEFORE it (its registers are the ones the Z3 values leave
  // when the head has turned them into the G3 image): requested at the tile head, as until round 3, the HBM loads
  // sat in front of the W^T fragment loads of phase B's dZ GEMM — memory returns in order, so the waves that start the
  // phase with that GEMM waited an HBM round trip for L2 hits.
  float z1r[16];
  auto load_z1 = [&](int64_t row0) {
    const int gi = (int)std::min<int64_t>(row0 + row, n_edges - 1);
    hx_load_z(z1r, rsZ1, HX_ZOFF(row0, gi), HX_ZQ(row0));
  };
  if ((int64_t)blockIdx.x < ntiles) {      // the first tile's head stands alone
    head1((int64_t)blockIdx.x * FTM, true, smem_hx, on, dn);
    NG_LDS_BARRIER();
    head2(smem_hx + 2 * HX_IMG_Z, smem_hx);
    load_z1((int64_t)blockIdx.x * FTM);
    NG_LDS_BARRIER();
    hx_img_write<HX_ROWZ>(smem_hx, prz, col0, z2r);  // Z2 pieces over the staging
    hx_wload(w0, wrs, lane * 16, ((1 * 4 + zk) * 8) * 2 * 1024, 0);
    NG_LDS_BARRIER();
  }

  // The image roles alternate per tile; the loop is unrolled by two with COMPILE-TIME roles (runtime image bases cost an
  // address VGPR per access site: 21 spills in a kernel that has no register to spare).
  auto tile_body = [&](int64_t tile, auto parity) {
    constexpr int gp = decltype(parity)::value, zp = gp;
#ifdef HX_STAMP
    ++titer;
#endif
    HX_T(0);
    const bool has_next = tile + gridDim.x < ntiles;
    const int64_t row0n = std::min<int64_t>(tile + gridDim.x, ntiles - 1) * FTM;
    char* IZ = smem_hx + zp * HX_IMG_Z;                             // Z2 -> R
    char* IZb = smem_hx + (zp ^ 1) * HX_IMG_Z;                      // Z1 ; next tile's Z2
    char* GA = smem_hx + 2 * HX_IMG_Z + gp * HX_IMG_G;              // G3 -> G1
    char* GB = smem_hx + 2 * HX_IMG_Z + (gp ^ 1) * HX_IMG_G;        // G2 ; next tile's Z3 staging -> G3
    // ------------------------------------------------------------------ phase B (layer 3)
    // the two waves of a SIMD (zrt = 0 / 1) take the two independent GEMMs of the phase in opposite order, so that
    // one wave's epilogue (VALU: s', split) runs beside the other's MFMAs
    if (zrt == 0) hx_dw_gemm(accW[2], accB, 4, IZ, GA, kslab, nsl0, lane);
    HX_T(1);
    {
      float g[16];
      // the wave whose epilogue comes NEXT gets the matrix pipe first (the arbiter otherwise favours the partner,
      // which then runs both of its GEMMs back to back and both epilogues end up side by side)
      if (zrt != 0) __builtin_amdgcn_s_setprio(2);
      // the Z1 pieces (IZb: nobody reads it in this phase) are written inside the product, a quarter row every other k-step
      hx_dz_gemm(g, GA, prg, wrs, w0, 1, zk, lane, [&](int ks) {
        if ((ks & 1) == 0) {
          const int q = ks >> 1;
          hx_img_write_q<HX_ROWZ>(IZb, prz, col0, q, z1r[4 * q], z1r[4 * q + 1], z1r[4 * q + 2], z1r[4 * q + 3]);
        }
      });
      if (zrt != 0) __builtin_amdgcn_s_setprio(0);
      HX_T(2);
#pragma unroll
      for (int r = 0; r < 16; r += 2) hx_sprime2(g[r], g[r + 1], z2r[r], z2r[r + 1]);
      hx_img_write<HX_ROWG>(GB, prg, col0, g);       // G2
    }
    hx_wload(w0, wrs, lane * 16, ((0 * 4 + zk) * 8) * 2 * 1024, 0);
    HX_T(3);
    if (zrt != 0) hx_dw_gemm(accW[2], accB, 4, IZ, GA, kslab, nsl0, lane);
    HX_T(4);
    NG_LDS_BARRIER();      // G2, Z1 images complete; every reader of G3 (GA) and Z2 (IZ) is done
    HX_T(5);
    // ------------------------------------------------------------------ phase C (layer 2)
    if (zrt == 0) hx_dw_gemm(accW[1], accB, 2, IZb, GB, kslab, nsl0, lane);
    HX_T(6);
    {
      float g[16];
      if (zrt != 0) __builtin_amdgcn_s_setprio(2);
      // R = m * rbf(d_eff) -> IZ (free since the barrier above) inside the product, a quarter row every other k-step
      // (masked rows: d = 1e19 -> exp2(-inf) = exact 0, as in the forward)
      const float dm = on ? dn : 1.0e19f;
      hx_dz_gemm(g, GB, prg, wrs, w0, 0, zk, lane, [&](int ks) {
        if ((ks & 1) == 0) {
          const int q = ks >> 1;
          const float4 mu = *reinterpret_cast<const float4*>(sCen + col0 + 8 * q);
          const float u0 = dm - mu.x, u1 = dm - mu.y, u2 = dm - mu.z, u3 = dm - mu.w;
          hx_img_write_q<HX_ROWZ>(IZ, prz, col0, q, __builtin_amdgcn_exp2f(u0 * u0 * a.neg_inv_gap_log2e),
                                  __builtin_amdgcn_exp2f(u1 * u1 * a.neg_inv_gap_log2e),
                                  __builtin_amdgcn_exp2f(u2 * u2 * a.neg_inv_gap_log2e),
                                  __builtin_amdgcn_exp2f(u3 * u3 * a.neg_inv_gap_log2e));
        }
      });
      if (zrt != 0) __builtin_amdgcn_s_setprio(0);
      // next tile's Z3 / d / dE (registers dead since the head).  Issued BEHIND the last W^T fragment loads of the tile:
      // memory returns in order, a fragment load queued behind these HBM loads would wait for all of them.
      prefetch(row0n, row0_of(tile, 2));
      HX_T(7);
#pragma unroll
      for (int r = 0; r < 16; r += 2) hx_sprime2(g[r], g[r + 1], z1r[r], z1r[r + 1]);
      hx_img_write<HX_ROWG>(GA, prg, col0, g);       // G1
    }
    HX_T(8);
    if (zrt != 0) hx_dw_gemm(accW[1], accB, 2, IZb, GB, kslab, nsl0, lane);
    HX_T(9);
    NG_LDS_BARRIER();      // G1, R images complete; every reader of G2 (GB) and Z1 (IZb) is done
    HX_T(10);
    // ------------------------------------------------------------------ phase D (layer 1) with the NEXT tile's head
    prefetch_z2(row0n);    // unconditional (clamped): no branch around the loads
    head1(row0n, has_next, IZb, on, dn);     // the loop-carried mask / distance now belong to the next tile (R is built)
    HX_T(11);
    NG_LDS_BARRIER();      // staging complete
    HX_T(12);
    if (zrt == 0) hx_dw_gemm(accW[0], accB, 0, IZ, GA, kslab, nsl0, lane);
    head2(GB, IZb);
    load_z1(row0n);
    HX_T(13);
    if (zrt != 0) hx_dw_gemm(accW[0], accB, 0, IZ, GA, kslab, nsl0, lane);
    HX_T(14);
    NG_LDS_BARRIER();      // every reader of the staging is done; every reader of G1 / R too
    hx_img_write<HX_ROWZ>(IZb, prz, col0, z2r);      // next tile's Z2 pieces over the staging
    hx_wload(w0, wrs, lane * 16, ((1 * 4 + zk) * 8) * 2 * 1024, 0);
    NG_LDS_BARRIER();
    HX_T(15);
  };
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += 2 * (int64_t)gridDim.x) {
    tile_body(tile, std::integral_constant<int, 0>());
    if (tile + gridDim.x >= ntiles) break;
    tile_body(tile + gridDim.x, std::integral_constant<int, 1>());
  }

#ifdef HX_STAMP
  __syncthreads();
  if (blockIdx.x == 3 && tid < 128) a.stamps[tid] = sStamp[tid];
  __syncthreads();
#endif
  __syncthreads();
  // ---------------------------------------------------------------------- write this workgroup's partial
  float* part = a.partial + (int64_t)blockIdx.x * a.part_stride;
  {
    // range guard (ng_internal.h): an activation or weight beyond the fp16 range has turned into NaN in the products
    float chk = 0.f;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) chk += fabsf(accW[l][j][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) chk += fabsf(accB[r]);
#pragma unroll
    for (int n = 0; n < 4; ++n) chk += fabsf(accWo[n]) + fabsf(accbo[n]);
    range_guard_raise(a.guard, not_finite(chk * ginv));
  }
  {
    const int k = kslab * 32 + l31;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = (nsl0 + j) * 32 + 8 * q + 4 * half;
          *reinterpret_cast<float4*>(part + l * FH * FH + k * FH + n) =
              make_float4(ginv * accW[l][j][4 * q + 0], ginv * accW[l][j][4 * q + 1], ginv * accW[l][j][4 * q + 2],
                          ginv * accW[l][j][4 * q + 3]);
        }
  }
  // bias sums of the two edge halves, dWo / dbo of the four row quarters: summed through LDS (IZ is free now)
  float* red = reinterpret_cast<float*>(smem_hx);
  float* dbw = reinterpret_cast<float*>(smem_hx + 2 * HX_IMG_Z);     // [4 k-slab waves][3 layers][128]
  const int red_stride = 3 * FH + FH * E + E;
  {
    const int c = l31 / 5;     // this lane's column group; its first column carries the sums
    if (c < 6 && l31 == 5 * c && (c & 1) >= 0) {
      const int layer = c >> 1, j = c & 1;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dbw[(kslab * 3 + layer) * FH + 32 * (nsl0 + j) + (r & 3) + 8 * (r >> 2) + 4 * half] = ginv * accB[r];
    }
  }
  // dbo: the lanes that own a row (zk == 0, lower half) publish their sums; fixed-order sum over the 64 rows below
  if (zk == 0 && half == 0) *reinterpret_cast<float4*>(sdE + 4 * row) = make_float4(accbo[0], accbo[1], accbo[2], accbo[3]);
  __syncthreads();
#pragma unroll
  for (int l = 0; l < 3; ++l) red[rq * red_stride + l * FH + cn] = dbw[(rq * 3 + l) * FH + cn];
  for (int n = 0; n < E; ++n) red[rq * red_stride + 3 * FH + cn * E + n] = ginv * accWo[n];
  if (cn < E) {
    float sbo = 0.f;
    if (rq == 0)
      for (int r = 0; r < FTM; ++r) sbo += sdE[4 * r + cn];
    red[rq * red_stride + 3 * FH + FH * E + cn] = ginv * sbo;
  }
  __syncthreads();
  for (int t = tid; t < red_stride; t += HX_THREADS)
    part[3 * FH * FH + t] = red[t] + red[red_stride + t] + red[2 * red_stride + t] + red[3 * red_stride + t];
}

// ---- the power-of-two gradient scale (header: Ranges).  Stage 1: per-block max |de|; stage 2 (one block): the
// largest absolute row sums of Wo, W3, W2 and S = 2^floor(log2(2^15 / bound)).  max is exact and order-free, so the
// scale — and with it every bit of the result — does not depend on the launch geometry.
constexpr int HX_SCALE_BLOCKS = 1024;
constexpr int64_t HX_DIRECT_MAX = 32768;      // values of de a workgroup scans itself (edge_bwd_h2_launch)
__global__ __launch_bounds__(256) void hx_absmax_kernel(const float* __restrict__ de, int64_t n, float* __restrict__ blockmax,
                                                        const int32_t* __restrict__ n_live) {
  __shared__ float red[4];
  if (n_live && *n_live < 0) return;      // the launch behind this one has nothing to do (edge-function table: guard down)
  float m = 0.f;
  // float4 body when the tensor starts 16-byte aligned (a torch allocation does), the rest element by element
  const int64_t n4 = (reinterpret_cast<uintptr_t>(de) & 15) == 0 ? n >> 2 : 0;
  const float4* de4 = reinterpret_cast<const float4*>(de);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = de4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(de[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Buffer offsets inside the kernel are 32-bit (one resource per array, 512 B of tape per edge), so one LAUNCH covers
// at most HX_SEG_EDGES edges; longer edge lists run as several launches over consecutive segments (a multiple of the
// 64-edge tile and of the 32-edge tape group), each with its own rows of the partial buffer.
constexpr int64_t HX_SEG_EDGES = ((int64_t)1 << 23) - 256;
int edge_bwd_h2_segments(int64_t n_edges) { return (int)std::max<int64_t>(1, cdiv(n_edges, HX_SEG_EDGES)); }

bool edge_bwd_h2_supported(int E, int64_t n_edges) { (void)n_edges; return E >= 1 && E <= 4; }

bool edge_tape_blocked(int E, int64_t n_edges) {
  return edge_split_enabled() && !sw().edge_bwd_math_fp32 && edge_bwd_h2_supported(E, n_edges);
}
constexpr size_t HX_WT_BYTES = (size_t)2 * 4 * 8 * 2 * 1024;       // the fragments; {nW2, nW3, nWo} follow

size_t edge_bwd_h2_ws_bytes() { return HX_WT_BYTES + 16 + (size_t)(HX_SCALE_BLOCKS + 2) * 4; }

// wt_img: edge_bwd_h2_ws_bytes() of scratch; partial: [edge_bwd_h2_segments(n_edges) * grid][part_stride]
int edge_bwd_h2_launch(ng_ctx* ctx, hipStream_t st, int64_t n_edges, int E, const float* d_src, const float* d_eff,
                       const float* centers, float gap, const float* const* W, const float* z_save, const float* de,
                       char* wt_img, float* partial, int part_stride, int grid, int tape_blocked, RangeGuard guard,
                       LiveEdges live) {
  float* blockmax = reinterpret_cast<float*>(wt_img + HX_WT_BYTES + 16);
  // the W^T image: cached while the weights are frozen / refreshed behind Adam, else in the caller's scratch
  int nb_max = 0;
  bool have_wt = false;
  char* wimg = (char*)cached_image(ctx, W[1], 13, HX_WT_BYTES + 16, &have_wt);
  const bool cached_wt = wimg != nullptr;
  if (!wimg) wimg = wt_img;
  {
    ProfScope ps(ctx, st, "edge_bwd_h2_prep");
    if (!have_wt) {
      PackJob j;
      j.kind = PK_EDGE_WT; j.blocks = 17; j.src[0] = W[1]; j.src[1] = W[2]; j.src[2] = W[3]; j.i0 = E; j.dst[0] = wimg;
      if (int rc = pack_launch(ctx, st, j)) return rc;
      if (cached_wt) cache_set_job(ctx, W[1], 13, j);
    }
    if (n_edges * E <= HX_DIRECT_MAX && (reinterpret_cast<uintptr_t>(de) & 15) == 0) {
      // one graph per call: the few workgroups read the 48 KB of de themselves instead of waiting for a launch that does
      nb_max = -(int)(n_edges * E);
      blockmax = const_cast<float*>(de);
    } else {
      nb_max = (int)std::min<int64_t>(HX_SCALE_BLOCKS, cdiv(n_edges * E, 1024));
      hipLaunchKernelGGL(hx_absmax_kernel, dim3(nb_max), dim3(256), 0, st, de, n_edges * E, blockmax, live.n_live);
      NG_HIP(ctx, hipGetLastError());
    }
  }
  const int nseg = edge_bwd_h2_segments(n_edges);
  for (int sg = 0; sg < nseg; ++sg) {
    const int64_t e0 = (int64_t)sg * HX_SEG_EDGES;
    EdgeBwdH2Args a;
    a.n_edges = std::min<int64_t>(HX_SEG_EDGES, n_edges - e0); a.d_src = d_src + e0; a.d_eff = d_eff + e0; a.centers = centers;
    a.neg_inv_gap_log2e = (float)(-1.4426950408889634 / (double)gap);
    a.wt_img = wimg; a.blockmax = blockmax; a.n_blockmax = nb_max; a.Wo = W[3]; a.z_save = z_save + e0 * FH; a.z_layer_stride = n_edges * FH;
    a.de = live.perm ? de : de + e0 * E;
    a.perm = live.perm ? live.perm + e0 : nullptr; a.n_live = live.n_live; a.row_base = e0;
    a.partial = partial + (size_t)sg * grid * part_stride; a.part_stride = part_stride; a.E = E; a.tape_blocked = tape_blocked;
    a.stamps = nullptr;
    a.guard = guard;
#ifdef HX_STAMP
    static unsigned long long* dbg = nullptr;
    if (!dbg) { hipMalloc(&dbg, 4096); hipMemset(dbg, 0, 4096); }
    a.stamps = dbg;
#endif
    ProfScope ps(ctx, st, "edge_bwd_h2");
    if (live.perm)
      hipLaunchKernelGGL(edge_bwd_h2_kernel<true>, dim3(grid), dim3(HX_THREADS), HX_LDS_BYTES, st, a);
    else
      hipLaunchKernelGGL(edge_bwd_h2_kernel<false>, dim3(grid), dim3(HX_THREADS), HX_LDS_BYTES, st, a);
    NG_HIP(ctx, hipGetLastError());
    if (sg + 1 < nseg) continue;
#ifdef HX_STAMP
    {
      static int calls = 0;
      if (++calls == 3) {
        unsigned long long h[256];
        hipStreamSynchronize(st);
        hipMemcpy(h, a.stamps, 2048, hipMemcpyDeviceToHost);
        const int ngrp = 2;      // stamped waves: 0, 4
        for (int w = 0; w < ngrp; ++w)
          for (int t = 0; t < 4; ++t) {
            printf("wave %d tile %d:", 4 * w, t);
            for (int k = 1; k < 16; ++k) printf(" %5lld", (long long)(h[(w * 4 + t) * 16 + k] - h[(w * 4 + t) * 16 + k - 1]));
            if (t < 3) printf(" | next %5lld", (long long)(h[(w * 4 + t + 1) * 16] - h[(w * 4 + t) * 16 + 15]));
            printf("  t0 %lld\n", (long long)(h[(w * 4 + t) * 16] - h[0]));
          }
      }
    }
#endif
  }
  return NG_OK;
}

}  // namespace ng
