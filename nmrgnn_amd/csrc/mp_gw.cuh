// Window gather-GEMM (round 5): the MPLayer update and the node-side pull of its backward at the reference's default
// width (atom_feature_size = 256, nmrgnn/model.py:22; semantics nmrgnn/layers.py:26-46) with NEITHER operand of the
// product in HBM — included by gemm_h2.hip (GgArgs, the PK_GG weight image, the fp32 repair kernel and the epilogue are
// mp_gg_kernel's).  What round 4's gather-GEMM measured (profiles/r04_gg_ab.txt, r05e_gw_ab.txt): four producer waves
// per CU set the pace even with the L2 loads AND the matrix products compiled out (386 us per launch) — one wave per
// SIMD issuing LDS record reads, index arithmetic, FMAs, splits and ring writes back to back.  This form has no roles:
//   * tile = 256 rows x 256 output columns, 256 threads = ONE wave per SIMD with the whole 512-register file: lane
//     (r = lane & 31, half = lane >> 5) of wave w OWNS rows m0 + 64 w + 32 b + r (b = 0, 1) for the whole kernel: the
//     gathered sums of a 16-column k-step are formed in its registers and split in place into the B operand of
//     v_mfma_f32_32x32x16_f16 (lane = row, 8 k-slots per half) — the gathered operand never touches LDS — and its
//     accumulators hold the two rows' 256 outputs (2 x 8 blocks x 16 registers), so the epilogue (row scale, activation,
//     residual) is the lane's own rows as well.  (Eight waves of 32 rows, two per SIMD at 256 registers each: the 128
//     accumulators beside sums, operands and weight fragments spilled 1 KB per lane.);
//   * the tile's list entries are staged ONCE in LDS as 16-byte records {window offset of the source, e_0..e_2} (64 KB);
//   * the gather reads an LDS WINDOW: the 64-byte column slice of the 320 source rows around the tile (molecule batches:
//     a row's sources lie in its own graph), brought in by LDS-DMA one k-step ahead, double-buffered.  A tile whose
//     sources span more rows (whole proteins, lists across graphs) takes the same code with buffer loads from L2;
//   * the weight operand is the shared one: the 16 KB of piece fragments of a (k-step, n) pair (PK_GG image, L2-resident)
//     come in by LDS-DMA two steps ahead into a three-slot ring (slot = n); every wave reads all of them (A operand:
//     lane = output column) and uses each fragment for both of its row blocks; 786 KB per 256 rows = 402 MB per launch
//     through the DMA path;
//   * one step = one n of one k-step, ONE barrier: 48 MFMAs per wave in eight groups of six, and after each group ONE
//     list entry (of both rows) of the NEXT k-step's gather — the matrix pipe works on the group while the wave issues
//     the entry's LDS reads and FMAs.
// Ranges as in mp_gg_kernel: forward operands are split unscaled (|A| >= 65504 raises the guard); the pull's sums are
// multiplied by S 2^-x (S of dP, 2^x >= max_n sum |e_n| of the row) and the epilogue multiplies the row by 2^x / S.  A
// raised guard is answered by mp_gg_repair_kernel.
// (LDS-DMA goes through h2_common.cuh's asm statement: lds_dma16)
typedef dma_i4 gw_i4;
__device__ __forceinline__ gw_i4 gw_rsrc(const void* p, unsigned bytes) { return dma_rsrc(p, bytes); }
__device__ __forceinline__ void gw_dma(gw_i4 rs, const void* lds_dst, int voff, int soff) { lds_dma16(rs, lds_dst, voff, soff); }

#ifdef GW_STAMP
__device__ unsigned long long gw_stamps[64];
#define GW_T(i) do { if (blockIdx.x == 300 && threadIdx.x == 0) gw_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define GW_T(i) do { } while (0)
#endif
typedef float gw_f2 __attribute__((ext_vector_type(2)));
typedef float gw_f4 __attribute__((ext_vector_type(4)));
constexpr int GW_THREADS = 256, GW_BM = 256, GW_WROWS = 320;
constexpr int GW_WINB = (GW_WROWS + 1) * 64;                             // one window buffer: 16 columns of 320 rows + one row of zeros
constexpr int GW_WSLOT = 16 * 1024;                                      // W piece fragments of one (k-step, n)
// staged records: entries 0..15 of the tile's rows as [entry][row] (the lanes of a wave read consecutive 16-byte
// records: packed row by row, rows of ~16 entries put every lane of a read on the same four banks — the record reads
// alone made a slot of the forward 980 cycles), entries 16.. packed row by row behind them; then the zero record
constexpr int GW_CAPA = 16 * GW_BM, GW_CAPB = 480, GW_ZERO = GW_CAPA + GW_CAPB;
constexpr int GW_LDS = 2 * GW_WINB + 3 * GW_WSLOT + (GW_ZERO + 1) * 16 + 128;     // 163,664 of 163,840 B
static_assert(GW_LDS <= 160 * 1024, "LDS");
constexpr int GW_OOB = 0x7ffffe00;                                      // non-window form: an offset past the end of the gathered array reads as zeros

// SL: groups per step that carry a slot of the gather (3 SL slots per k-step beside matrix work; 6 for lists of at most 16
// entries, else 8; longer lists finish in slots of their own)
template <int LK, int E, bool GRAD, int SL>
__global__ __launch_bounds__(GW_THREADS, 1) void mp_gw_kernel(GgArgs a) {
  static_assert(E == 3, "ring slot = n");
  extern __shared__ __attribute__((aligned(16))) char smem_gw[];
  char* const win = smem_gw;                                             // [2][GW_WROWS][64 B]
  char* const wring = smem_gw + 2 * GW_WINB;                             // [3][8 column blocks][2 pieces][1 KB]
  float4* const srec = reinterpret_cast<float4*>(smem_gw + 2 * GW_WINB + 3 * GW_WSLOT);   // [GW_ZERO + 1]
  int* const ctl = reinterpret_cast<int*>(srec + GW_ZERO + 1);          // per wave: [4] lowest, [4] highest source, [4] tail entries
  constexpr int WCHUNK = gx_wchunk(4);                                   // bytes of the image per (32-column slab, n)
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int F = a.F, NK = F / 16;
  const int64_t m0 = (int64_t)blockIdx.x * GW_BM;

  GW_T(0);
  // ---- the tile's entries into LDS: thread r stages row m0 + r (raw source index first), lowest / highest source on the way
  const int64_t srow = m0 + tid;
  const bool sval = srow < a.M;
  const int64_t smc = sval ? srow : a.M - 1;
  int64_t sp0;
  int sdeg;
  if (LK == GG_PADDED) { sp0 = smc * a.Kpad; sdeg = sval ? a.Kpad : 0; }
  else { sp0 = a.ptr[smc]; sdeg = sval ? (int)(a.ptr[smc + 1] - sp0) : 0; }
  // where the row's entries 16.. go: exclusive scan of max(deg - 16, 0) over the tile's rows
  const int sex = max(sdeg - 16, 0);
  int sinc = sex;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(sinc, o); if (lane >= o) sinc += v; }
  if (lane == 63) ctl[8 + wave] = sinc;
  NG_LDS_BARRIER();
  int sob = sinc - sex;
#pragma unroll
  for (int i = 0; i < 3; ++i) if (i < wave) sob += ctl[8 + i];
  const bool overflow = ctl[8] + ctl[9] + ctl[10] + ctl[11] > GW_CAPB;
  // slot of entry j of tile row r whose tail starts at ob; GW_ZERO: not staged
  auto phys = [&](int r, int j, int ob) __attribute__((always_inline)) {
    return j < 16 ? j * GW_BM + r : (ob + j - 16 < GW_CAPB ? GW_CAPA + ob + j - 16 : GW_ZERO);
  };
  int lo = 0x7fffffff, hi = -1;
  if (LK == GG_PADDED && E == 3 && a.Kpad == 16) {
    // the common padded form: a row's 16 sources (64 B) and 48 weights (192 B) as sixteen 16-byte loads instead of 64 scalar ones
    if (sdeg > 0) {
      int4 nl4[4];
      float4 ew4[12];
      const int4* pn = reinterpret_cast<const int4*>(a.idx + sp0);
      const float4* pe = reinterpret_cast<const float4*>(a.ew + sp0 * 3);
#pragma unroll
      for (int u = 0; u < 4; ++u) nl4[u] = pn[u];
#pragma unroll
      for (int u = 0; u < 12; ++u) ew4[u] = pe[u];
      const int* nli = reinterpret_cast<const int*>(nl4);
      const float* ewf = reinterpret_cast<const float*>(ew4);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int sidx = nli[j];
        lo = min(lo, sidx); hi = max(hi, sidx);
        srec[j * GW_BM + tid] = make_float4(__builtin_bit_cast(float, sidx), ewf[3 * j], ewf[3 * j + 1], ewf[3 * j + 2]);
      }
    }
  } else
  for (int j0 = 0; j0 < sdeg; j0 += 4) {       // four entries in flight
    float4 e4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e4[u] = gg_entry<LK, E>(a, sp0 + std::min(j0 + u, sdeg - 1));
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + u < sdeg) {
        const int sidx = __builtin_bit_cast(int, e4[u].x);
        lo = min(lo, sidx); hi = max(hi, sidx);
        const int q = phys(tid, j0 + u, sob);
        if (q != GW_ZERO) srec[q] = e4[u];
      }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
  if (lane == 0) { ctl[wave] = lo; ctl[4 + wave] = hi; }
  NG_LDS_BARRIER();
  int tlo = min(min(ctl[0], ctl[1]), min(ctl[2], ctl[3])), thi = max(max(ctl[4], ctl[5]), max(ctl[6], ctl[7]));
  tlo = __builtin_amdgcn_readfirstlane(tlo); thi = __builtin_amdgcn_readfirstlane(thi);
  const bool none = thi < tlo;
  const int wlo = none ? 0 : tlo;
  const bool winok = !a.no_window && (none || thi - tlo < GW_WROWS);
  // the zero record (past a row's end): weights 0 and a source that reads as zeros whatever the array holds
  if (tid == 0) srec[GW_ZERO] = make_float4(__builtin_bit_cast(float, winok ? GW_WROWS * 64 : GW_OOB), 0.f, 0.f, 0.f);
  if (tid < 32) {
    *reinterpret_cast<float*>(win + GW_WROWS * 64 + (tid & 15) * 4 + (tid >> 4) * GW_WINB) = 0.f;
  }
  // source index -> byte offset: into a window buffer, or into the gathered array (host: M * F * 4 < 2^31)
  for (int j = 0; j < sdeg; ++j) {
    const int q = phys(tid, j, sob);
    if (q != GW_ZERO) {
      const int sidx = __builtin_bit_cast(int, srec[q].x);
      srec[q].x = __builtin_bit_cast(float, winok ? ((sidx - wlo) * 64) | ((((sidx - wlo) >> 2) & 3) << 4) : sidx * (F * 4));
    }
  }
  GW_T(1);
  // ---- this lane's two rows (row b was staged by the lane pair's half b)
  int64_t mrow[2];
  bool valid[2];
  int rloc[2], deg[2], ob[2];     // tile-local row, entries, start of its tail in region B
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    rloc[b] = 64 * wave + 32 * b + l31;
    mrow[b] = m0 + rloc[b];
    valid[b] = mrow[b] < a.M;
    deg[b] = __shfl(sdeg, l31 + 32 * b);
    ob[b] = __shfl(sob, l31 + 32 * b);
  }
  int dmax = max(deg[0], deg[1]);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) dmax = max(dmax, __shfl_xor(dmax, o));
  const int wdmax = __builtin_amdgcn_readfirstlane(dmax);
  // global position of entry j of row b (for the entries that were not staged)
  int64_t gp0[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int64_t mcb = valid[b] ? mrow[b] : a.M - 1;
    gp0[b] = LK == GG_PADDED ? mcb * a.Kpad : (int64_t)a.ptr[mcb];
  }
  NG_LDS_BARRIER();
  float pscale[2] = {1.0f, 1.0f}, sfac[2] = {1.0f, 1.0f};
  if (GRAD) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float se0 = 0.f, se1 = 0.f, se2 = 0.f;
      for (int j0 = 0; j0 < deg[b]; j0 += 4) {
        float4 rc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rc[u] = srec[j0 + u < deg[b] ? phys(rloc[b], j0 + u, ob[b]) : GW_ZERO];
#pragma unroll
        for (int u = 0; u < 4; ++u) { se0 += fabsf(rc[u].y); se1 += fabsf(rc[u].z); se2 += fabsf(rc[u].w); }
      }
      if (overflow)
        for (int j = 16; j < deg[b]; ++j)
          if (phys(rloc[b], j, ob[b]) == GW_ZERO) {
            const float4 rc = gg_entry<LK, E>(a, gp0[b] + j);
            se0 += fabsf(rc.y); se1 += fabsf(rc.z); se2 += fabsf(rc.w);
          }
      const float sm = fmaxf(se0, fmaxf(se1, se2));
      int ex = 0;
      if (sm > 1.0f && sm < 3.0e38f) (void)frexpf(sm, &ex);        // sm = f * 2^ex, f in [0.5, 1)
      pscale[b] = a.gscale[0] * ldexpf(1.0f, -ex);
      sfac[b] = ldexpf(1.0f, ex);
    }
  }

  const __amdgpu_buffer_rsrc_t rsG =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.G), 0, (unsigned)(a.M * a.F * 4), 0x00020000);
  GW_T(2);
  const gw_i4 dmaG = gw_rsrc(a.G, (unsigned)(a.M * a.F * 4));
  const gw_i4 dmaW = gw_rsrc(a.Wimg, (unsigned)((int64_t)(F / 32) * E * WCHUNK));

  // W fragments of step (kk, n) -> ring slot n: 16 instructions of 1 KB, 4 per wave
  auto req_w = [&](int kk, int n) __attribute__((always_inline)) {
#ifdef GW_ABL_NOWDMA
    if (kk > 0) return;
#endif
    const int u = kk >> 1, ks = kk & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = wave * 4 + q;                        // (nb, p) = (f >> 1, f & 1)
      const int goff = (u * E + n) * WCHUNK + (((f >> 1) * 2 + ks) * 2 + (f & 1)) * 1024;
      gw_dma(dmaW, wring + n * GW_WSLOT + f * 1024, lane * 16, goff);
    }
  };
  // window of k-step kk -> win[kk & 1]: rows wlo .. wlo + 319, 64 bytes each; 20 instructions of 1 KB (16 rows), 5 per wave.
  // The row is part of the LANE offset, which the buffer bounds check covers: rows past the end of the array read as zeros.
  auto req_win = [&](int kk) __attribute__((always_inline)) {
#ifdef GW_ABL_NOWINDMA
    if (kk > 1) return;
#endif
    char* dst = win + (kk & 1) * GW_WINB;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int i = wave * 5 + q;
      // 16-byte piece q of window row r sits at position q ^ ((r >> 2) & 3): rows 4 apart share their banks (64-byte rows), the
      // swizzle spreads the same piece of such rows over four positions (the gather's row reads: 8-way -> 2-way conflicts)
      const unsigned vo = (unsigned)(wlo + 16 * i + (lane >> 2)) * (unsigned)(F * 4) + (unsigned)((((lane & 3) ^ ((lane >> 4) & 3)) * 16) + kk * 64);
      gw_dma(dmaG, dst + i * 1024, (int)vo, 0);
    }
  };
  static_assert(GW_WROWS == 4 * 5 * 16, "req_win");

  f32x16 acc[2][8];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][i][r] = 0.f;

  auto run = [&](auto win_tag) __attribute__((always_inline)) {
    constexpr bool WIN = decltype(win_tag)::value;
    // one value row (this lane's 32 bytes of the slice) times the three weights
    // (plain v_fma_f32: beside MFMAs a packed-f32 instruction costs more than the two it replaces — MI355X guide)
    auto axpy = [&](float (&s)[E][8], const float4& rc, const gw_f4& v0, const gw_f4& v1) __attribute__((always_inline)) {
      const float wn[3] = {rc.y, rc.z, rc.w};
#pragma unroll
      for (int n = 0; n < E; ++n)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[n][t] = fmaf(wn[n], v0[t], s[n][t]);
          s[n][4 + t] = fmaf(wn[n], v1[t], s[n][4 + t]);
        }
    };
    // The gather of one k-step as a three-stage stream over "slots" t = 0, 1, ...: slot t requests the records of entry t
    // of both rows, requests the value rows of entry t - 1 (its records arrived a slot ago) and adds the products of entry
    // t - 2 (its value rows arrived a slot ago).  A slot is branch-free — past a row's end the zero record — and shares
    // ONE basic block with a group of six MFMAs, which the scheduler is told to spread over the slot's ~65 vector
    // instructions: an in-order wave that issues six MFMAs back to back sits 192 cycles in front of the matrix pipe and
    // then leaves it idle for the 400 cycles of the entry (measured: 590 per group + entry, profiles/r05e_gw_ab.txt).
    // Buffers rotate by the slot number, which is a compile-time constant everywhere (24 slots per k-step).
    float4 rec[3][2];          // [t % 3][row]
    gw_f4 val[2][2][2];        // [t % 2][row][16-byte half]
    auto stream_reset = [&]() __attribute__((always_inline)) {
      const float4 z = srec[GW_ZERO];
#pragma unroll
      for (int i = 0; i < 3; ++i) { rec[i][0] = z; rec[i][1] = z; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b) { val[i][b][0] = gw_f4{0.f, 0.f, 0.f, 0.f}; val[i][b][1] = gw_f4{0.f, 0.f, 0.f, 0.f}; }
    };
    // T: the slot's number modulo 6 (buffer rotation), t: its number (entry index)
    // requests of slot t: value rows of entry t - 1, records of entry t
    auto slot_loads = [&](auto Tc, int t, int kk, bool on) __attribute__((always_inline)) {
      constexpr int T = decltype(Tc)::value;
      const char* wb = win + (kk & 1) * GW_WINB;
      const int so = __builtin_amdgcn_readfirstlane(kk * 64);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int off = __builtin_bit_cast(int, rec[(T + 2) % 3][b].x);
        if (WIN) {      // pieces 2 half and 2 half + 1 of the row, at their swizzled positions
          const int o0 = off ^ (half << 5);
          val[T % 2][b][0] = *reinterpret_cast<const gw_f4*>(wb + o0);
          val[T % 2][b][1] = *reinterpret_cast<const gw_f4*>(wb + (o0 ^ 16));
        } else {
          val[T % 2][b][0] = __builtin_bit_cast(gw_f4, __builtin_amdgcn_raw_buffer_load_b128(rsG, off + 32 * half, so, 0));
          val[T % 2][b][1] = __builtin_bit_cast(gw_f4, __builtin_amdgcn_raw_buffer_load_b128(rsG, off + 32 * half + 16, so, 0));
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) rec[T % 3][b] = srec[(on && t < deg[b]) ? phys(rloc[b], t, ob[b]) : GW_ZERO];
    };
    // products of slot t = entry t - 2, part (b, n): eight v_fma_f32 — as long as one MFMA
    auto slot_fma = [&](auto Tc, int b, int n, float (&s)[2][E][8]) __attribute__((always_inline)) {
      constexpr int T = decltype(Tc)::value;
      const float4& rc = rec[(T + 1) % 3][b];
      const float w1 = n == 0 ? rc.y : (n == 1 ? rc.z : rc.w);
#ifdef GW_PK
      const gw_f2 ww = {w1, w1};
#pragma unroll
      for (int hq = 0; hq < 2; ++hq)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          gw_f2 p = {s[b][n][4 * hq + 2 * u], s[b][n][4 * hq + 2 * u + 1]};
          p = __builtin_elementwise_fma(ww, gw_f2{val[(T + 1) % 2][b][hq][2 * u], val[(T + 1) % 2][b][hq][2 * u + 1]}, p);
          s[b][n][4 * hq + 2 * u] = p[0]; s[b][n][4 * hq + 2 * u + 1] = p[1];
        }
#else
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[b][n][t] = fmaf(w1, val[(T + 1) % 2][b][0][t], s[b][n][t]);
        s[b][n][4 + t] = fmaf(w1, val[(T + 1) % 2][b][1][t], s[b][n][4 + t]);
      }
#endif
    };
    auto slot = [&](auto Tc, int t, int kk, float (&s)[2][E][8], bool on = true) __attribute__((always_inline)) {
      slot_loads(Tc, t, kk, on);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int n = 0; n < E; ++n) slot_fma(Tc, b, n, s);
    };
    // slots t0 .. (a multiple of six at a time) without matrix work: the first k-step's gather and the tail of a long list
    auto stream_plain = [&](int t0, int t1, int kk, float (&s)[2][E][8]) __attribute__((always_inline)) {
#pragma unroll 1
      for (int t = t0; t < t1; t += 6) {
        slot(std::integral_constant<int, 0>{}, t, kk, s);
        slot(std::integral_constant<int, 1>{}, t + 1, kk, s);
        slot(std::integral_constant<int, 2>{}, t + 2, kk, s);
        slot(std::integral_constant<int, 3>{}, t + 3, kk, s);
        slot(std::integral_constant<int, 4>{}, t + 4, kk, s);
        slot(std::integral_constant<int, 5>{}, t + 5, kk, s);
      }
    };
    auto spill_entries = [&](int kk, int j0, int j1, float (&s)[2][E][8]) __attribute__((always_inline)) {
      if (!overflow) return;        // a tile whose rows' tails (entries 16..) exceed GW_CAPB: the rest from memory
      const char* wbs = win + (kk & 1) * GW_WINB;
      const int so = __builtin_amdgcn_readfirstlane(kk * 64);
#pragma unroll 1
      for (int j = max(j0, 16); j < j1; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (j < deg[b] && phys(rloc[b], j, ob[b]) == GW_ZERO) {
            const float4 e4 = gg_entry<LK, E>(a, gp0[b] + j);
            const int sr = __builtin_bit_cast(int, e4.x);
            gw_f4 v0, v1;
            if (WIN) {
              const int o0 = (((sr - wlo) * 64) | ((((sr - wlo) >> 2) & 3) << 4)) ^ (half << 5);
              v0 = *reinterpret_cast<const gw_f4*>(wbs + o0);
              v1 = *reinterpret_cast<const gw_f4*>(wbs + (o0 ^ 16));
            } else {
              v0 = __builtin_bit_cast(gw_f4, __builtin_amdgcn_raw_buffer_load_b128(rsG, sr * (F * 4) + 32 * half, so, 0));
              v1 = __builtin_bit_cast(gw_f4, __builtin_amdgcn_raw_buffer_load_b128(rsG, sr * (F * 4) + 32 * half + 16, so, 0));
            }
            axpy(s[b], e4, v0, v1);
          }
    };
    auto zero = [&](float (&s)[2][E][8]) __attribute__((always_inline)) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int n = 0; n < E; ++n)
#pragma unroll
          for (int t = 0; t < 8; ++t) s[b][n][t] = 0.f;
    };
    // the sums as the B operand: 8 k-slots of this lane's row, two fp16 pieces
    auto split = [&](const float (&s)[2][E][8], u32x4 (&x)[2][E][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int n = 0; n < E; ++n) {
          unsigned h[4], l[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float x0 = s[b][n][2 * t], x1 = s[b][n][2 * t + 1];
            split2_pair(GRAD ? pscale[b] * x0 : x0, GRAD ? pscale[b] * x1 : x1, h[t], l[t]);
          }
          x[b][n][0] = u32x4{h[0], h[1], h[2], h[3]};
          x[b][n][1] = u32x4{l[0], l[1], l[2], l[3]};
        }
    };
    // forward, training: the aggregate as a by-product (the backward's dw = A^T dP reads it); two lanes write 64 contiguous bytes
    auto store_a = [&](int kk, const float (&s)[2][E][8]) __attribute__((always_inline)) {
      if (GRAD || !a.A_out) return;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (!valid[b]) continue;
        float* ao = a.A_out + mrow[b] * ((int64_t)E * F) + 16 * kk + 8 * half;
#pragma unroll
        for (int n = 0; n < E; ++n) {
          __builtin_nontemporal_store(gw_f4{s[b][n][0], s[b][n][1], s[b][n][2], s[b][n][3]}, reinterpret_cast<gw_f4*>(ao + (int64_t)n * F));
          __builtin_nontemporal_store(gw_f4{s[b][n][4], s[b][n][5], s[b][n][6], s[b][n][7]}, reinterpret_cast<gw_f4*>(ao + (int64_t)n * F + 4));
        }
      }
    };

    float sums[2][E][8];
    u32x4 X[2][E][2];
    req_w(0, 0);
    req_w(0, 1);
    if (WIN) { req_win(0); if (NK > 1) req_win(1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NG_LDS_BARRIER();
    GW_T(3);
    zero(sums);
    const int nslots = wdmax + 2;            // slots a gather needs (the stream is three stages deep)
    stream_reset();
    stream_plain(0, nslots, 0, sums);
    spill_entries(0, 0, wdmax, sums);
    split(sums, X);
    NG_LDS_BARRIER();      // window(2) goes where window(0) is: not before every wave has finished the gather above
    GW_T(4);
#pragma unroll 1
    for (int kk = 0; kk < NK; ++kk) {
      if (kk < 4) GW_T(8 + 4 * kk);
      // top: X = operand of k-step kk, its sums still in `sums`; W(kk, 0) landed, W(kk, 1) requested; window(kk + 1) landed
      store_a(kk, sums);
      zero(sums);
#pragma unroll
      for (int n = 0; n < E; ++n) {
        // two steps ahead; the slot of W(kk, 2) / W(kk + 1, n - 1) was read in the previous step
        if (n == 0) req_w(kk, 2);
        else if (kk + 1 < NK) req_w(kk + 1, n - 1);
        if (WIN && n == 0 && kk + 2 < NK) req_win(kk + 2);
        // the next k-step's gather streams through the groups of six MFMAs: slot SL n + j beside group j < SL, always (no
        // branch around it: the accumulators of a two-sided branch end up in two register sets); past the lists' ends and
        // in the last k-step the slots work on the zero record
        const bool more = kk + 1 < NK;
        if (n == 0) stream_reset();
        const char* wst = wring + n * GW_WSLOT + lane * 16;
        // group j = output column block j of BOTH row blocks: one fragment pair (8 registers) feeds six MFMAs on two
        // accumulators; block j + 1 is read while block j multiplies
        u32x4 wa[2][2];              // [buffer][piece]
        auto wread = [&](int j, u32x4 (&w)[2]) __attribute__((always_inline)) {
#pragma unroll
          for (int p = 0; p < 2; ++p) w[p] = *reinterpret_cast<const u32x4*>(wst + (j * 2 + p) * 1024);
        };
        // MFMA i of a group: pieces (l h, h l, h h) on the accumulators of row blocks 0 and 1 alternately
        auto mma1 = [&](const u32x4 (&w)[2], int j, int i) __attribute__((always_inline)) {
#ifndef GW_ABL_NOMFMA
          const int b = i & 1, pw = i < 2 ? 1 : 0, px = (i >> 1) == 1 ? 1 : 0;
          acc[b][j] = mfma_f16(w[pw], X[b][n][px], acc[b][j]);
#endif
        };
        wread(0, wa[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          __builtin_amdgcn_sched_barrier(0);
          if (j + 1 < 8) wread(j + 1, wa[(j + 1) & 1]);
          const int t = SL * n + j;
#ifdef GW_ABL_NOGATHER
          constexpr bool gather_on = false;
#else
          constexpr bool gather_on = true;
#endif
          if (gather_on && j < SL) {
            // this slot's requests, then six times (one MFMA, eight FMAs of entry t - 2), each pinned: left to itself the
            // scheduler pulls several slots' worth of loads to the front and spills ~1 KB per lane
            const int T6 = t % 6;       // a constant after unrolling: the rotation index folds
            auto with_T = [&](auto Tc) __attribute__((always_inline)) {
              slot_loads(Tc, t, kk + 1, more);
#pragma unroll
              for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                mma1(wa[j & 1], j, i);
                slot_fma(Tc, i / 3, i % 3, sums);
                {   // pin the eight sums HERE: the FMAs are pure arithmetic whose results are needed only at the k-step's end,
                    // and instruction selection otherwise emits all of a step's FMAs behind its last group (eight slots' value
                    // rows and records live at once: 1.2 KB of scratch per lane)
                  float (&q)[8] = sums[i / 3][i % 3];
                  asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));
                }
              }
            };
            if (T6 == 0) with_T(std::integral_constant<int, 0>{});
            if (T6 == 1) with_T(std::integral_constant<int, 1>{});
            if (T6 == 2) with_T(std::integral_constant<int, 2>{});
            if (T6 == 3) with_T(std::integral_constant<int, 3>{});
            if (T6 == 4) with_T(std::integral_constant<int, 4>{});
            if (T6 == 5) with_T(std::integral_constant<int, 5>{});
          } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) mma1(wa[j & 1], j, i);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (n == E - 1) {
#ifndef GW_ABL_NOGATHER
          if (more && nslots > 3 * SL) stream_plain(3 * SL, nslots, kk + 1, sums);
#endif
          if (kk + 1 < NK) spill_entries(kk + 1, 0, wdmax, sums);
        }
        if (n == E - 1 && kk + 1 < NK) split(sums, X);
        // everything but this step's own requests has landed (vector memory operations complete in order)
        if (n == 0) {
          if (WIN && kk + 2 < NK) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (kk + 1 < NK) {
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        NG_LDS_BARRIER();
        if (kk < 4) GW_T(9 + 4 * kk + n);
      }
    }
  };
  if (winok) run(std::true_type{});
  else run(std::false_type{});

  // ---- epilogue.  The lane holds, for ITS rows, columns 32 i + 8 q + 4 half + (0..3) of block i: row scale, activation and
  // range check happen in that layout; the 64-column chunk of the wave's 32 rows then turns through a wave-private piece of
  // LDS (the loop's last barrier has released all of it) so that 16 lanes read / write 256 contiguous bytes of one row
  // (row-per-lane 16-byte accesses made the residual loads and the stores 56-78k of a tile's 390k cycles: store-issue-bound)
  GW_T(5);
  constexpr int EPROW = 272;                                  // 64 columns + 16 B: b128 rows conflict-free
  char* const ep = smem_gw + wave * (32 * EPROW);
  const int rr = lane >> 4, cc = lane & 15;
  // a row block's residual rows are all requested before its first chunk is turned (one memory round trip per row block
  // instead of one per 64-column chunk: the registers of the loop's sums and operands are free here)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    gw_f4 r[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + 64 * wave + 32 * b + rr + 4 * i;
        r[c][i] = gw_f4{0.f, 0.f, 0.f, 0.f};
        if (a.R && m < a.M) r[c][i] = *reinterpret_cast<const gw_f4*>(a.R + m * F + 64 * c + 4 * cc);
      }
    const float rs = GX_WINV * (GRAD ? a.gscale[1] * sfac[b] : 1.0f) * ((a.rowscale && valid[b]) ? a.rowscale[mrow[b]] : 1.0f);
    const int64_t mb = m0 + 64 * wave + 32 * b;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      {   // range guard (ng_internal.h): inf - inf of an out-of-range piece arrives here as NaN
        float chk = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) chk += fabsf(acc[b][2 * c][t]) + fabsf(acc[b][2 * c + 1][t]);
        range_guard_raise(a.guard, valid[b] && not_finite(chk * rs));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gw_f4 v;
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = acc[b][2 * c + j][4 * q + t] * rs;
          if (a.act == NG_ACT_SOFTPLUS) {           // the branch once per value group, the common activation without the switch
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = act_apply(NG_ACT_SOFTPLUS, v[t]);
          } else if (a.act != NG_ACT_NONE) {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = act_apply(a.act, v[t]);
          }
          *reinterpret_cast<gw_f4*>(ep + l31 * EPROW + (32 * j + 8 * q + 4 * half) * 4) = v;
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const gw_f4 v = *reinterpret_cast<const gw_f4*>(ep + (rr + 4 * i) * EPROW + cc * 16);
        const int64_t m = mb + rr + 4 * i;
        if (m < a.M) {
          if (a.S) *reinterpret_cast<gw_f4*>(a.S + m * F + 64 * c + 4 * cc) = v;
          *reinterpret_cast<gw_f4*>(a.Y + m * F + 64 * c + 4 * cc) = v + r[c][i];
        }
      }
    }
  }
  GW_T(6);
}
