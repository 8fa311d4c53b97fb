// Fused persistent backward of the edge path, SIXTEEN waves with split roles (round 5; the default of the split-operand
// backward, NG_EDGE_BWD=w8 selects the eight-wave kernel of edge_bwd_h2.hip for A/B).  Same math, same fp16-piece images,
// same tape, same W^T fragments, same workgroup partials as edge_bwd_h2.hip; backward of nmrgnn/model.py:251-261.
//
// Why.  The eight-wave kernel is bound by the SUM of four partially overlapping resources (matrix pipe 8.4k cycles per
// 64-edge tile and SIMD, LDS 9k, VALU issue 6k, L1 5.5k: a tile takes 18.5k): its two waves per SIMD sit in one barrier
// domain and run the same code, so VALU / LDS-store phases of one wave meet the same phases of the other.  Here a SIMD
// holds FOUR waves (1024 threads, <= 128 VGPRs) of two kinds that never run the same code:
//   Z waves (0..7; wave = k-slab zk x edge half zrt, as the eight waves of the old kernel): everything that touches the
//     tape — the head (mask, dE, G3 = (dE Wo^T) s'(Z3), Z2 image, dWo), the two dZ GEMMs with the Z1 / R images built
//     inside them, s' epilogues, all image writes.  48 MFMAs + 4 per tile, ~600 VALU instructions.
//   M waves (8..15; wave = k-slab x n-slab pair): the three dW GEMMs (transposing reads + 72 MFMAs per tile), nothing
//     else; the 96 accumulator registers live only here.
// The role branch is the OUTERMOST control flow of the kernel (two complete tile loops with the same barrier sequence):
// live ranges are per program point, not per wave — a branch inside a common loop would keep the M waves' accumulators
// allocated through the Z waves' code.
//
// Schedule (three barriers per tile; images alternate per tile as in the old kernel, X = parity, Y = other):
//   interval   Z waves (tile t)                              M waves
//   I1         head: G3 -> G[X], Z2 -> Z[X], dWo             dW1(t-1): R, G1 of the tile before (Z[Y], G[Y])
//   I2         dZ2 = G3 W3^T (+ Z1 -> Z[Y]), G2 -> G[Y]      dW3(t): Z2, G3 (Z[X], G[X])
//   I3         dZ1 = G2 W2^T (+ R -> Z[X]),  G1 -> G[X]      dW2(t): Z1, G2 (Z[Y], G[Y])
// A Z wave needs nothing from another wave inside an interval; the matrix pipe sees 48 + 48 MFMAs per SIMD in I2 / I3.
//
// What changed against the old kernel besides the roles:
//   dWo = Z3^T dE without the fp32 staging tile (it had no buffer left, and cost two barriers): the Z wave turns its
//     32 x 32 block of Z3 into the accumulator layout of a 32x32 MFMA with lane = COLUMN by multiplying its pieces with a
//     0/1 selection matrix (4 MFMAs, exact: a piece times 1), and sums D[edge] * dE[edge][n] over its 16 registers; the
//     dE rows come from a wave-private LDS copy (no cross-wave dependency).  Z3 enters as h + l (2^-22 relative).
//   bias gradients: column sums of the G fragments the M waves read anyway (v_dot2c_f32_f16 with ones, fp32
//     accumulate) added to a per-(k-slab) LDS row by the one wave that owns it (program order: deterministic).
#include "edge_bwd_h2.cuh"

#if defined(RS_ABL_NOZIMG) || defined(RS_ABL_NOMREAD) || defined(RS_ABL_NOFILL) || defined(RS_ABL_NOW) || defined(RS_ABL_NOG) || defined(RS_ABL_NOM)
#define RS_ABL 1      // timing experiments (results are wrong): no range guard, so that the fp32 fallback does not run
#define range_guard_raise(g, bad) ((void)(bad))
#endif

namespace ng {

constexpr int RS_THREADS = 1024;
// LDS behind the four images: Wo pairs [128][4] | centres [128] | dE copies [8 Z waves][32][4] | selection operands
// [2 steps][64 lanes] x 16 B | bias rows [4 k-slabs][3 layers][128] | scratch [16]
constexpr int RS_MISC_FLOATS = FH * 4 + FH + 8 * 32 * 4 + 2 * 64 * 4 + 4 * 3 * FH + 16;
#ifdef HX_STAMP
constexpr int RS_LDS_BYTES = HX_IMGS + RS_MISC_FLOATS * 4 + 2048;
#else
constexpr int RS_LDS_BYTES = HX_IMGS + RS_MISC_FLOATS * 4;      // 152,128 of 163,840
#endif

typedef _Float16 rs_h2 __attribute__((ext_vector_type(2)));

// sum of the eight fp16 values of a fragment register set, fp32 accumulate.  (The element goes through a scalar first:
// __builtin_bit_cast applied DIRECTLY to an element of an ext_vector — bit_cast(T, v[i]) — reads element 0 for every i
// with hipcc 7.2; found here as a bias gradient that was 8 x too large.)
__device__ __forceinline__ float rs_sum8(u32x4 v, float s) {
  const rs_h2 one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned w = v[i];
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(rs_h2, w), one, s, false);
  }
  return s;
}

// acc[j][n][k] += sum_edges G[e][n] Zin[e][k] as hx_dw_gemm; the bias gradient (column sums of G over the k-step this
// wave owns among the four that share its n-slabs) goes to dbrow[0..63] = the wave's two n-slabs of its k-slab's LDS row
__device__ __forceinline__ void rs_dw_gemm(f32x16 (&acc)[2], float* __restrict__ dbrow, const char* __restrict__ imgZ,
                                           const char* __restrict__ imgG, int kslab, int nsl0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const char* zb = imgZ + (4 * (g >> 1) + 8 * (i >> 2)) * HX_ROWZ + (16 * (g & 1) + 4 * (i & 3)) * 2 + 64 * kslab;
  const char* g0 = imgG + (4 * (i >> 2) + 2 * (g >> 1)) * HX_ROWG + (16 * (g & 1) + 4 * (i & 3)) * 2 + 64 * nsl0;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    // Sixteen fragment registers instead of the eight-wave kernel's twenty-four: the Z operand and the G operand of the
    // first n-slab, three MFMAs, then the second n-slab's G operand into the first one's registers.  The M waves have
    // time to spare (they wait for the Z waves most of an interval) and no registers: 96 accumulators + these.
    u32x4 b[2], a[2];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#ifdef RS_ABL_NOMREAD
      b[p] = u32x4{(unsigned)lane, 1u, 2u, 3u}; a[p] = u32x4{(unsigned)lane, 5u, 6u, 7u};
      asm volatile("" : "+v"(b[p]), "+v"(a[p]));
#else
      b[p] = hx_tr_frag(zb + p * HX_PIECE_Z + ((ks & 1) + 32 * (ks >> 1)) * HX_ROWZ, 2 * HX_ROWZ);
      a[p] = hx_tr_frag(g0 + p * HX_PIECE_G + 16 * ks * HX_ROWG, HX_ROWG);
#endif
    }
    acc[0] = mma3(a, b, acc[0]);
#ifndef RS_NO_DB
    if (ks == kslab) s0 = rs_sum8(a[0], rs_sum8(a[1], 0.f));
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifdef RS_ABL_NOMREAD
    asm volatile("" : "+v"(a[0]), "+v"(a[1]));
#else
#pragma unroll
    for (int p = 0; p < 2; ++p) a[p] = hx_tr_frag(g0 + 64 + p * HX_PIECE_G + 16 * ks * HX_ROWG, HX_ROWG);
#endif
    acc[1] = mma3(a, b, acc[1]);
#ifndef RS_NO_DB
    if (ks == kslab) {
      s1 = rs_sum8(a[0], rs_sum8(a[1], 0.f));
      // lanes l and l + 32 hold the two 8-edge halves of the k-step for the same column.  ONE half exchange does both
      // sums: v_permlane32_swap(vdst = s0, src = s1) leaves {s0[l], s1[l-32]} in vdst and {s0[l+32], s1[l]} in src, so
      // vdst + src is the whole-column sum of n-slab 0 in lanes 0..31 and of n-slab 1 in lanes 32..63 = dbrow[lane]
      const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s0), __builtin_bit_cast(unsigned, s1), false, false);
      const unsigned ra = r[0], rb = r[1];
      atomicAdd(dbrow + lane, __builtin_bit_cast(float, ra) + __builtin_bit_cast(float, rb));
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}

// fp16 1.0 the optimizer cannot see through (cf. h2_minus_one): keeps the h + l below on v_fma_mix_f32
__device__ __forceinline__ _Float16 rs_one16() {
  int b = 0x3C00;
  asm volatile("" : "+s"(b));
  return __builtin_bit_cast(_Float16, (short)b);
}

// this lane's 16 values (columns col0 + 8q + j of its row) read back from a Z-type piece image as h + l: the s' epilogues
// take Z from the image their own wave wrote instead of holding the tape values in registers for an interval (the Z waves
// carry a tile's worth of tape loads in flight and have no room for a second copy).  |h + l - z| <= 2^-22 |z|.
// One v_fma_mix_f32 per element (three fp16 sources) + 8 ds_read_b64.
__device__ __forceinline__ void rs_img_read_z(const char* __restrict__ img, int row, int col0, float (&z)[16]) {
  const float one = (float)rs_one16();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const char* p = img + row * HX_ROWZ + (col0 + 8 * q) * 2;
    const u32x2 h = *reinterpret_cast<const u32x2*>(p), l = *reinterpret_cast<const u32x2*>(p + HX_PIECE_Z);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned hw = h[i], lw = l[i];
      const f16x2_cvt hv = __builtin_bit_cast(f16x2_cvt, hw), lv = __builtin_bit_cast(f16x2_cvt, lw);
      z[4 * q + 2 * i] = __builtin_fmaf((float)hv[0], one, (float)lv[0]);
      z[4 * q + 2 * i + 1] = __builtin_fmaf((float)hv[1], one, (float)lv[1]);
    }
  }
}

// hx_dz_gemm with the W^T fragments RS_WDEPTH - 1 steps ahead (the eight-wave kernel: three; here four waves per SIMD
// cover the L2 latency and the Z waves have no registers for a fourth set)
#ifndef RS_WDEPTH
#define RS_WDEPTH 3
#endif
#ifndef RS_DE_ROWS
#define RS_DE_ROWS 4      // dE rows per group of the dWo sums (edge_bwd_rs_kernel, head)
#endif
template <class F>
__device__ __forceinline__ void rs_dz_gemm(float (&out)[16], const char* __restrict__ imgG, int prow_g,
                                           __amdgpu_buffer_rsrc_t wrs, const u32x4 (&w0)[2], int L, int zk, int lane,
                                           F&& fill) {
  constexpr int D = RS_WDEPTH;
  const int half = lane >> 5;
  const char* gb = imgG + prow_g * HX_ROWG + 16 * half;
  const int wvo = lane * 16;
  int wso = ((L * 4 + zk) * 8) * 2 * 1024;
  asm volatile("" : "+s"(wso));      // (see hx_dz_gemm)
  f32x16 acc0;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
  u32x4 wa[D][2], b[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) { wa[0][p] = w0[p]; b[0][p] = *reinterpret_cast<const u32x4*>(gb + p * HX_PIECE_G); }
#pragma unroll
#ifndef RS_ABL_NOW
  for (int i = 1; i < D - 1; ++i) hx_wload(wa[i], wrs, wvo, wso, i);
#else
  for (int i = 1; i < D - 1; ++i) { wa[i][0] = w0[0]; wa[i][1] = w0[1]; }
#endif
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
#ifndef RS_ABL_NOW
    if (ks + D - 1 < 8) hx_wload(wa[(ks + D - 1) % D], wrs, wvo, wso, ks + D - 1);
#else
    if (ks + D - 1 < 8) { wa[(ks + D - 1) % D][0] = w0[0]; wa[(ks + D - 1) % D][1] = w0[1]; }
#endif
    if (ks < 7) {
#pragma unroll
#ifndef RS_ABL_NOG
      for (int p = 0; p < 2; ++p) b[(ks + 1) & 1][p] = *reinterpret_cast<const u32x4*>(gb + 32 * (ks + 1) + p * HX_PIECE_G);
#else
      for (int p = 0; p < 2; ++p) { b[(ks + 1) & 1][p] = b[ks & 1][p]; asm volatile("" : "+v"(b[(ks + 1) & 1][p])); }
#endif
    }
    acc0 = mma3(wa[ks % D], b[ks & 1], acc0);
#ifndef RS_ABL_NOFILL
    fill(ks);
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) out[r] = acc0[r] * HX_WINV;     // the W^T pieces carry 2^8
}

// Lane geometry of a Z wave, derived from an OPAQUE copy of the lane id at the head of every interval: anything the
// optimizer can trace to the plain lane id it hoists out of the tile loop as an invariant (image offsets, table addresses,
// tape offsets: ~20 registers the Z waves do not have — they came back as scratch reloads, and a scratch reload waits for
// every tape load queued in front of it).  Recomputed, the values live for one interval; ~15 VALU instructions each time.
struct RsGeo { int lane, half, l31, row, col0, prz, prg; };
__device__ __forceinline__ RsGeo rs_geo(int lane_in, int zk, int zrt) {
  RsGeo g;
  g.lane = lane_in;
  asm volatile("" : "+v"(g.lane));
  g.half = g.lane >> 5; g.l31 = g.lane & 31;
  g.row = 32 * zrt + g.l31;
  g.col0 = 32 * zk + 4 * g.half;
  g.prz = hx_prow_z(g.row); g.prg = hx_prow_g(g.row);
  return g;
}

#ifdef HX_STAMP
#define RS_T(k)                                                                              \
  do {                                                                                       \
    if (lane == 0 && (wave & 3) == 0 && titer >= 2 && titer < 6)                             \
      sStamp[((wave >> 2) * 4 + (titer - 2)) * 16 + (k)] = __builtin_readcyclecounter();     \
  } while (0)
#else
#define RS_T(k)
#endif

template <bool LIVE>
__global__ __launch_bounds__(RS_THREADS) void edge_bwd_rs_kernel(EdgeBwdH2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_rs[];
  float* sWo4 = reinterpret_cast<float*>(smem_rs + HX_IMGS);      // [128][4], column pairs (see edge_bwd_h2.hip)
  float* sCen = sWo4 + FH * 4;                                    // [128]
  float* sdEw = sCen + FH;                                        // [8][32][4]
  u32x4* sSel = reinterpret_cast<u32x4*>(sdEw + 8 * 32 * 4);      // [2][64]
  float* dbw = reinterpret_cast<float*>(sSel + 2 * 64);           // [4][3][128]
  float* sred = dbw + 4 * 3 * FH;                                 // [16]
#ifdef HX_STAMP
  unsigned long long* sStamp = reinterpret_cast<unsigned long long*>(sred + 16);   // [4][4][16]
  int titer = -1;
#endif

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int E = a.E;
  // the power-of-two gradient scale: as in edge_bwd_h2.hip (max is exact and order-free: the same S in every workgroup)
  float gscale, ginv;
  {
    float m = 0.f;
    for (int i = tid; i < a.n_blockmax; i += RS_THREADS) m = fmaxf(m, a.blockmax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) sred[wave] = m;
    __syncthreads();
    m = sred[0];
#pragma unroll
    for (int w = 1; w < RS_THREADS / 64; ++w) m = fmaxf(m, sred[w]);
    const float* wb = reinterpret_cast<const float*>(a.wt_img + 2 * 4 * 8 * 2 * 1024);      // {nW2, nW3, nWo}
    const float b3 = m * wb[2], b2 = b3 * wb[1], b1 = b2 * wb[0];
    const float bound = fmaxf(b3, fmaxf(b2, b1));
    int ex = 0;
    if (bound > 0.f && bound < 3.0e38f) {
      int eb;
      (void)frexpf(bound, &eb);
      ex = 15 - eb;
      ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
    }
    // (wave-uniform: kept in scalar registers — the M waves have no vector register to spare for them)
    gscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ldexpf(1.0f, ex))));
    ginv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ldexpf(1.0f, -ex))));
  }
  const int64_t n_edges = LIVE ? std::max<int64_t>(0, std::min<int64_t>(a.n_edges, (int64_t)*a.n_live - a.row_base)) : a.n_edges;
  const int64_t ntiles = (n_edges + FTM - 1) / FTM;

  for (int t = tid; t < FH * 4; t += RS_THREADS) {
    const int c = t >> 2, n = t & 3;
    sWo4[((c >> 1) * 4 + n) * 2 + (c & 1)] = n < E ? a.Wo[c * E + n] : 0.f;
  }
  if (tid < FH) sCen[tid] = a.centers[tid];
  for (int t = tid; t < 4 * 3 * FH; t += RS_THREADS) dbw[t] = 0.f;
  if (tid < 128) {
    // selection operand of step s for lane (column jc, k-slot group hb): k-slot t of the Z3 fragment of step s holds
    // column 4 hb + 8 (2 s + (t >> 2)) + (t & 3) of the wave's 32-column slab (hx_img_write's pairing)
    const int s = tid >> 6, ln = tid & 63, jc = ln & 31, hb = ln >> 5;
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t0 = 2 * i, t1 = 2 * i + 1;
      const int c0 = 4 * hb + 8 * (2 * s + (t0 >> 2)) + (t0 & 3), c1 = 4 * hb + 8 * (2 * s + (t1 >> 2)) + (t1 & 3);
      w[i] = (c0 == jc ? 0x3C00u : 0u) | (c1 == jc ? 0x3C000000u : 0u);
    }
    sSel[tid] = u32x4{w[0], w[1], w[2], w[3]};
  }
  __syncthreads();

  // image bases by parity p: Z-type p at smem + p * HX_IMG_Z, G-type p at smem + 2 * HX_IMG_Z + p * HX_IMG_G
#define RS_Z(p) (smem_rs + (p) * HX_IMG_Z)
#define RS_G(p) (smem_rs + 2 * HX_IMG_Z + (p) * HX_IMG_G)
  float* part = a.partial + (int64_t)blockIdx.x * a.part_stride;
  float* red = reinterpret_cast<float*>(smem_rs);                  // end of kernel: [4][red_stride] over the images
  const int red_stride = 3 * FH + FH * E + E;
  float* sBo = reinterpret_cast<float*>(smem_rs + 2 * HX_IMG_Z);    // end of kernel: [64][4] row sums of dE

#ifdef RS_NO_Z
  if (false) {
#else
  if (wave < 8) {
#endif
    // =============================================================================================== Z waves
    __builtin_amdgcn_s_setprio(2);      // the Z waves carry the critical path; the M waves fill the matrix pipe behind them
    const int zk = wave & 3, zrt = wave >> 2;
    float* sdE = sdEw + wave * 128;
#define HX_ZFULL(ROW0) (a.tape_blocked && (ROW0) + 32 * zrt + 32 <= ne)
#define HX_ZOFF(GE, ROW0, GI) (HX_ZFULL(ROW0) ? ((ROW0) / 32 + zrt) * 16384 + (zk * 256 + (GE).lane) * 16 : (GI) * (FH * 4) + (GE).col0 * 4)
#define HX_ZQ(ROW0) (HX_ZFULL(ROW0) ? 1024 : 32)
    const unsigned zbytes = (unsigned)(n_edges * FH * 4);
    const __amdgpu_buffer_rsrc_t rsZ1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z_save), 0, zbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z_save + a.z_layer_stride), 0, zbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z_save + 2 * a.z_layer_stride), 0, zbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsDs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d_src), 0, (unsigned)(n_edges * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsDn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d_eff), 0, (unsigned)(n_edges * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsDe = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.de), 0, LIVE ? 0xFFFFFFFFu : (unsigned)(n_edges * a.E * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsPm = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.perm), 0, LIVE ? (unsigned)(n_edges * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wt_img), 0, 2 * 4 * 8 * 2 * 1024, 0x00020000);

    // (32-bit indices: a launch covers at most HX_SEG_EDGES = 2^23 - 256 rows; 64-bit row arithmetic costs register pairs)
    const int ne = (int)n_edges, nt = (int)ntiles, gstep = (int)gridDim.x;
    float accWo[4] = {0.f, 0.f, 0.f, 0.f};     // dWo[32 zk + l31][n] over this lane half's edges
    float accbo = 0.f;                         // sum over the tiles of dE[this lane's row][zk] (8 lanes hold a row: each keeps one column)
    float z3r[16], z2r[16], z1r[16], pf_ds = 1.f, pf_dn = 0.f, pf_de[4];
    int pf_slot = 0;
    // (the prefetch discipline of edge_bwd_h2.hip: nothing in these blocks may USE a loaded value)
    auto load_slot = [&](const RsGeo& ge_, int row0) {
      const int gi = std::max(std::min(row0 + ge_.row, ne - 1), 0);
      pf_slot = (int)__builtin_amdgcn_raw_buffer_load_b32(rsPm, gi * 4, 0, 0);
    };
    // Loads of a wave return in order and a tile's tape (96 KB per CU) needs most of the tile's time to arrive at the CU's
    // share of the HBM rate, so a tile's worth of requests is in flight all the time and their placement among the W^T
    // fragment loads decides who waits: Z3 / d / dE of the NEXT tile are requested inside I2's product right behind its
    // last fragment load (Z3's registers are free since the head), Z2 of the next tile the same way inside I3's product,
    // Z1 of THIS tile in the head behind the dWo block (no fragment load follows until the head's last instruction).
    // Each set dies at its image write: the s' epilogues read Z back from the image (rs_img_read_z).
    auto prefetch_a = [&](const RsGeo& ge_, int row0, int row0_after) {
      const int gi = std::min(row0 + ge_.row, ne - 1);
      hx_load_z(z3r, rsZ3, HX_ZOFF(ge_, row0, gi), HX_ZQ(row0));
      if (!LIVE) pf_ds = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsDs, gi * 4, 0, 0));
      pf_dn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsDn, gi * 4, 0, 0));
      const int ge = LIVE ? pf_slot : gi;
#pragma unroll
      for (int n = 0; n < 4; ++n)
        pf_de[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsDe, (ge * E + std::min(n, E - 1)) * 4, 0, 0));
      if (LIVE) load_slot(ge_, row0_after);
    };
    auto prefetch_z2 = [&](const RsGeo& ge_, int row0) {
      const int gi = std::min(row0 + ge_.row, ne - 1);
      hx_load_z(z2r, rsZ2, HX_ZOFF(ge_, row0, gi), HX_ZQ(row0));
    };
    auto load_z1 = [&](const RsGeo& ge_, int row0) {
      const int gi = std::min(row0 + ge_.row, ne - 1);
      hx_load_z(z1r, rsZ1, HX_ZOFF(ge_, row0, gi), HX_ZQ(row0));
    };
    auto row0_of = [&](int tile, int steps) { return std::min(tile + steps * gstep, nt - 1) * FTM; };
    if ((int)blockIdx.x < nt) {
      const RsGeo g0 = rs_geo(lane, zk, zrt);
      if (LIVE) load_slot(g0, (int)blockIdx.x * FTM);
      prefetch_a(g0, (int)blockIdx.x * FTM, row0_of(blockIdx.x, 1));
      prefetch_z2(g0, (int)blockIdx.x * FTM);
    }

    auto ztile = [&](int tile, auto parity) {
      constexpr int p = decltype(parity)::value;
#ifdef HX_STAMP
      ++titer;
#endif
      RS_T(0);
      const int row0 = tile * FTM;
      const int row0n = std::min(tile + gstep, nt - 1) * FTM;
      // ------------------------------------------------------------------ I1: the head
      const RsGeo gh = rs_geo(lane, zk, zrt);
      const bool on = pf_ds > 0.f && row0 + gh.row < ne;
      const float dn = pf_dn;
      float dEm[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) dEm[n] = (on && n < E) ? gscale * pf_de[n] : 0.f;
      accbo += zk == 0 ? dEm[0] : zk == 1 ? dEm[1] : zk == 2 ? dEm[2] : dEm[3];
      if (gh.half == 0) *reinterpret_cast<float4*>(sdE + 4 * gh.l31) = make_float4(dEm[0], dEm[1], dEm[2], dEm[3]);
      RS_T(1);
      {
        // dWo: this wave's Z3 block with gh.lane = column (header), then 16 edges x 4 outputs of FMAs
        f32x16 D;
#pragma unroll
        for (int r = 0; r < 16; ++r) D[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {      // (one k-step of pieces at a time: eight registers less at the kernel's fullest point)
          u32x4 zh, zl;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned h, l;
            split2_pair(z3r[8 * s + 2 * i], z3r[8 * s + 2 * i + 1], h, l);
            zh[i] = h; zl[i] = l;
          }
          const u32x4 sel = sSel[64 * s + gh.lane];
          D = mfma_f16(zl, sel, D);
          D = mfma_f16(zh, sel, D);
        }
        // Four rows at a time, each group's address tied to the previous group's sums through an empty asm: left to
        // themselves the sixteen dE row reads are all issued first and their 64 registers stay live until the MFMA result
        // arrives (sched_barrier alone does not help: the FMAs had been moved away from their loads before scheduling)
        int dro = 64 * gh.half;
#pragma unroll
        for (int rb = 0; rb < 16; rb += RS_DE_ROWS) {
          asm volatile("" : "+v"(dro), "+v"(accWo[0]), "+v"(accWo[1]), "+v"(accWo[2]), "+v"(accWo[3]));
          const char* dp = reinterpret_cast<const char*>(sdE) + dro;
#pragma unroll
          for (int r = rb; r < rb + RS_DE_ROWS; ++r) {
            const float4 d = *reinterpret_cast<const float4*>(dp + 16 * ((r & 3) + 8 * (r >> 2)));
            accWo[0] = fmaf(D[r], d.x, accWo[0]); accWo[1] = fmaf(D[r], d.y, accWo[1]);
            accWo[2] = fmaf(D[r], d.z, accWo[2]); accWo[3] = fmaf(D[r], d.w, accWo[3]);
          }
        }
        asm volatile("" : "+v"(accWo[0]), "+v"(accWo[1]), "+v"(accWo[2]), "+v"(accWo[3]));
      }
      RS_T(2);
      __builtin_amdgcn_sched_barrier(0);
      load_z1(gh, row0);      // (behind the dWo block: its sixteen registers are claimed from here on)
      __builtin_amdgcn_sched_barrier(0);
      // G3 = (dE Wo^T) * s'(Z3)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const float4 w01 = *reinterpret_cast<const float4*>(sWo4 + 4 * (gh.col0 + 8 * q + j));      // (x0 x1 y0 y1)
          const float4 w23 = *reinterpret_cast<const float4*>(sWo4 + 4 * (gh.col0 + 8 * q + j) + 4);  // (z0 z1 w0 w1)
          float p0 = w01.x * dEm[0], p1 = w01.y * dEm[0];
          p0 = fmaf(w01.z, dEm[1], p0); p1 = fmaf(w01.w, dEm[1], p1);
          p0 = fmaf(w23.x, dEm[2], p0); p1 = fmaf(w23.y, dEm[2], p1);
          p0 = fmaf(w23.z, dEm[3], p0); p1 = fmaf(w23.w, dEm[3], p1);
          z3r[4 * q + j] = hx_sprime(p0, z3r[4 * q + j]);
          z3r[4 * q + j + 1] = hx_sprime(p1, z3r[4 * q + j + 1]);
        }
      RS_T(3);
      hx_img_write<HX_ROWG>(RS_G(p), gh.prg, gh.col0, z3r);       // G3
      RS_T(4);
#ifdef RS_ABL_NOZIMG
      for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(z2r[r]));
#else
      hx_img_write<HX_ROWZ>(RS_Z(p), gh.prz, gh.col0, z2r);       // Z2
#endif
      u32x4 w0[2];
      hx_wload(w0, wrs, gh.lane * 16, ((1 * 4 + zk) * 8) * 2 * 1024, 0);
      RS_T(5);
      NG_LDS_BARRIER();
      RS_T(6);
      // ------------------------------------------------------------------ I2: layer 3 -> 2
      const RsGeo gu = rs_geo(lane, zk, zrt);
      {
        float g[16];
        rs_dz_gemm(g, RS_G(p), gu.prg, wrs, w0, 1, zk, gu.lane, [&](int ks) {
          if ((ks & 1) == 0) {
            const int q = ks >> 1;
#ifdef RS_ABL_NOZIMG
            asm volatile("" :: "v"(z1r[4 * q]), "v"(z1r[4 * q + 1]), "v"(z1r[4 * q + 2]), "v"(z1r[4 * q + 3]));
#else
            hx_img_write_q<HX_ROWZ>(RS_Z(p ^ 1), gu.prz, gu.col0, q, z1r[4 * q], z1r[4 * q + 1], z1r[4 * q + 2], z1r[4 * q + 3]);
#endif
          }
          if (ks == 8 - RS_WDEPTH) prefetch_a(gu, row0n, row0_of(tile, 2));      // behind the product's last fragment load
        });
        RS_T(7);
        {
          float zz[16];
          rs_img_read_z(RS_Z(p), gu.prz, gu.col0, zz);       // Z2 (this gu.lane's own writes of the head)
#pragma unroll
          for (int r = 0; r < 16; r += 2) hx_sprime2(g[r], g[r + 1], zz[r], zz[r + 1]);
        }
        hx_img_write<HX_ROWG>(RS_G(p ^ 1), gu.prg, gu.col0, g);       // G2
      }
      hx_wload(w0, wrs, gu.lane * 16, ((0 * 4 + zk) * 8) * 2 * 1024, 0);
      RS_T(8);
      NG_LDS_BARRIER();
      RS_T(9);
      // ------------------------------------------------------------------ I3: layer 2 -> 1
      const RsGeo gv = rs_geo(lane, zk, zrt);
      {
        float g[16];
        const float dm = on ? dn : 1.0e19f;      // masked rows: exp2(-inf) = exact 0, as in the forward
        rs_dz_gemm(g, RS_G(p ^ 1), gv.prg, wrs, w0, 0, zk, gv.lane, [&](int ks) {
          if ((ks & 1) == 0) {
            const int q = ks >> 1;
            const float4 mu = *reinterpret_cast<const float4*>(sCen + gv.col0 + 8 * q);
            const float u0 = dm - mu.x, u1 = dm - mu.y, u2 = dm - mu.z, u3 = dm - mu.w;
            hx_img_write_q<HX_ROWZ>(RS_Z(p), gv.prz, gv.col0, q, __builtin_amdgcn_exp2f(u0 * u0 * a.neg_inv_gap_log2e),
                                    __builtin_amdgcn_exp2f(u1 * u1 * a.neg_inv_gap_log2e),
                                    __builtin_amdgcn_exp2f(u2 * u2 * a.neg_inv_gap_log2e),
                                    __builtin_amdgcn_exp2f(u3 * u3 * a.neg_inv_gap_log2e));
          }
          if (ks == 8 - RS_WDEPTH) prefetch_z2(gv, row0n);
        });
        RS_T(10);
        {
          float zz[16];
          rs_img_read_z(RS_Z(p ^ 1), gv.prz, gv.col0, zz);   // Z1 (written inside I2's product)
#pragma unroll
          for (int r = 0; r < 16; r += 2) hx_sprime2(g[r], g[r + 1], zz[r], zz[r + 1]);
        }
        hx_img_write<HX_ROWG>(RS_G(p), gv.prg, gv.col0, g);       // G1
      }
      RS_T(11);
      NG_LDS_BARRIER();
      RS_T(12);
    };
#pragma unroll 1
    for (int tile = blockIdx.x; tile < nt; tile += 2 * gstep) {
      ztile(tile, std::integral_constant<int, 0>());
      // (no `break` between the halves: a second loop exit with the accumulators live made the allocator spill them — the M loop)
      if (tile + gstep < nt) ztile(tile + gstep, std::integral_constant<int, 1>());
    }
    NG_LDS_BARRIER();      // the M waves' last dW GEMM has read its images
    {
      float chk = fabsf(accbo);
#pragma unroll
      for (int n = 0; n < 4; ++n) chk += fabsf(accWo[n]);
      range_guard_raise(a.guard, not_finite(chk * ginv));
    }
    // dWo of the four (edge half, lane half) groups -> red[rq][3 FH + col * E + n]; dbo rows -> sBo
    {
      const int rq = 2 * zrt + half, cn = 32 * zk + l31, row = 32 * zrt + l31;
      for (int n = 0; n < E; ++n) red[rq * red_stride + 3 * FH + cn * E + n] = ginv * accWo[n];
      if (half == 0) sBo[4 * row + zk] = accbo;
    }
#undef HX_ZFULL
#undef HX_ZOFF
#undef HX_ZQ
#ifdef RS_NO_M
  } else if (false) {
#else
  } else {
#endif
    // =============================================================================================== M waves
    const int mw = wave - 8, kslab = mw >> 1, nsl0 = 2 * (mw & 1);
    f32x16 accW[3][2];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[l][j][r] = 0.f;
    float* dbk = dbw + kslab * 3 * FH + 32 * nsl0;      // + layer * FH
    // The loop runs one interval behind the schedule in the header: its body is tile t's I2, I3 and the NEXT tile's I1
    // (dW1 of tile t), so that there is no first / last special case with 96 live accumulator registers around it; the
    // lone barrier in front is the end of the first tile's I1, the last one of the loop the Z waves' closing barrier.
    auto mtile = [&](auto parity) {
      constexpr int p = decltype(parity)::value;
#ifdef HX_STAMP
      ++titer;
#endif
      RS_T(0);
#ifndef RS_ABL_NOM
      rs_dw_gemm(accW[2], dbk + 2 * FH, RS_Z(p), RS_G(p), kslab, nsl0, lane);                   // dW3
#endif
      RS_T(1);
      NG_LDS_BARRIER();
      RS_T(2);
#ifndef RS_ABL_NOM
      rs_dw_gemm(accW[1], dbk + FH, RS_Z(p ^ 1), RS_G(p ^ 1), kslab, nsl0, lane);               // dW2
#endif
      RS_T(3);
      NG_LDS_BARRIER();
      RS_T(4);
#ifndef RS_ABL_NOM
      rs_dw_gemm(accW[0], dbk, RS_Z(p), RS_G(p), kslab, nsl0, lane);                            // dW1
#endif
      RS_T(5);
      NG_LDS_BARRIER();
      RS_T(6);
    };
    NG_LDS_BARRIER();
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += 2 * (int64_t)gridDim.x) {
      mtile(std::integral_constant<int, 0>());
      // (an `if`, not a `break`: a second loop exit with 96 live accumulator registers costs 450 B/lane of spills)
      if (tile + gridDim.x < ntiles) mtile(std::integral_constant<int, 1>());
    }
    {
      float chk = 0.f;
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) chk += fabsf(accW[l][j][r]);
      range_guard_raise(a.guard, not_finite(chk * ginv));
    }
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
#ifdef HX_STAMP
  __syncthreads();
  if (blockIdx.x == 3 && tid < 256) a.stamps[tid] = sStamp[tid];
#endif
  // ---------------------------------------------------------------------- the small gradients of this workgroup's partial
  // bias rows of the four k-slab waves (LDS, accumulated over the tiles), dWo of the four row groups, dbo
  if (tid < 512) {
    const int cn = tid & 127, rq = tid >> 7;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const float v = ginv * dbw[(rq * 3 + l) * FH + cn];
      range_guard_raise(a.guard, not_finite(v));
      red[rq * red_stride + l * FH + cn] = v;
    }
  }
  __syncthreads();
  if (tid < E) {
    float sbo = 0.f;
    for (int r = 0; r < FTM; ++r) sbo += sBo[4 * r + tid];
    red[3 * FH + FH * E + tid] = ginv * sbo;
    for (int rq = 1; rq < 4; ++rq) red[rq * red_stride + 3 * FH + FH * E + tid] = 0.f;
  }
  __syncthreads();
  for (int t = tid; t < red_stride; t += RS_THREADS)
    part[3 * FH * FH + t] = red[t] + red[red_stride + t] + red[2 * red_stride + t] + red[3 * red_stride + t];
}

void edge_bwd_rs_run(hipStream_t st, int grid, const EdgeBwdH2Args& a, bool live) {
  if (live)
    hipLaunchKernelGGL(edge_bwd_rs_kernel<true>, dim3(grid), dim3(RS_THREADS), RS_LDS_BYTES, st, a);
  else
    hipLaunchKernelGGL(edge_bwd_rs_kernel<false>, dim3(grid), dim3(RS_THREADS), RS_LDS_BYTES, st, a);
}

}  // namespace ng
