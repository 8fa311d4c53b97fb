// NOT COMPILED — kept as text with its measurement (tools/variants/README.md).
// Wave-autonomous form of the FC block's forward piece body (round 6): drop-in for fc_fwd_body_h2 in nmrgnn_amd/csrc/fc_fused.hip
// (same FcFwdArgs, same launch; the biases move behind 57 KB of re-ordered fragments: see the BODY / fc_fwd_lds_bytes lines at the
// end).  60 tests green (test_gpu_fc_block, test_gpu_parity, test_gpu_determinism).  Measured on the bench step, same box, against
// the column-tile body with the activation as a compile-time constant: 46.5-48.6 us against 45.3-46.6 — no barrier, no LDS round
// trip of the activations, and no gain: the kernel's time is its 1,088 VALU + 84 MFMA instructions per 16 rows and four layers,
// issued by two waves per SIMD, in either form.
//
// ---- the wave-autonomous piece body (round 6) ---------------------------------------------------------------------------
// The piece body above hands every layer's output to the next through LDS planes and a workgroup barrier, and its four waves
// own COLUMN tiles: a 64-row tile is a chain of four barrier-to-barrier intervals of ~3 us each with two workgroups per CU to hide
// them (52 us for 131,072 rows whose arithmetic is ~20).  Here a wave owns 16 ROWS through all layers: with the transposed product
// (A = rows of W^T, B = the wave's rows) lane (row j = lane & 15, group g = lane >> 4) receives, of output tile T, the features
// in D rows 4g .. 4g + 3 — so the A operand's rows are taken in the order  D row rho of tile T <- feature 32 (T >> 1) + 8 (rho >> 2)
// + 4 (T & 1) + (rho & 3):  tiles 2s, 2s + 1 then leave the lane with features 32 s + 8 g + 0 .. 7, exactly the eight k-slots the
// next layer's B operand wants from it in k-step s.  Activations never leave the registers, there is no barrier behind the one that
// follows the staging of the fragments (the PK_FC piece image re-ordered into LDS, 56 KB for four layers), the residual input is the
// register the output replaces.  Range: every row is scaled by a power of two taken from its own maximum (the four lanes of a row
// meet in two shuffles) before it is split, so no finite activation can leave the fp16 range and there is no repair path; weights
// beyond the piece range select the fp32 body for the launch, as before.
// ACT: the activation as a compile-time constant (softplus, the reference's default: straight-line code, nothing between the
// MFMAs and the elementwise work but the scheduler), or -1: read from the arguments (a scalar branch per use)
template <int NL, int ACT>
__device__ __forceinline__ void fc_fwd_body_wave(const FcFwdArgs& a, char* smem, float* sB) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a16 = lane & 15, g4 = lane >> 4;
#pragma unroll
  for (int l = 0; l < NL; ++l)
    if (tid < FC_F) sB[l * FC_F + tid] = (l < NL - 1 || tid < FC_H) ? a.p.b[l][tid] : 0.f;
  u32x4* ws = reinterpret_cast<u32x4*>(smem);      // [(l, T, s, piece)][64 lanes]; hidden layers 1024 entries each, the last 512
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.WfH);
    constexpr int TOT = (NL - 1) * 1024 + 512;
    for (int e = tid; e < TOT; e += 256) {
      const int l = e < (NL - 1) * 1024 ? e >> 10 : NL - 1;
      const int r = e - l * 1024;
      const int ln = r & 63, piece = (r >> 6) & 1, s = (r >> 7) & 1, T = r >> 8;
      const int i = ln & 15, kg = ln >> 4;
      const int ct = 2 * (T >> 1) + (i >> 3), ip = 8 * ((i >> 2) & 1) + 4 * (T & 1) + (i & 3);
      ws[e] = src[(((l * 4 + ct) * 2 + s) * 2 + piece) * 64 + ip + 16 * kg];
    }
  }
  __syncthreads();

  const int64_t ntiles = (a.N + FC_TM - 1) / FC_TM;
  const float4* x4 = reinterpret_cast<const float4*>(a.x);
  auto rowc = [&](int64_t tile) { return std::min<int64_t>(tile * FC_TM + 16 * wave + a16, a.N - 1); };
  float4 nx0, nx1, nx2, nx3;
  {
    const float4* p = x4 + rowc(blockIdx.x) * 16 + 2 * g4;
    nx0 = p[0]; nx1 = p[1]; nx2 = p[8]; nx3 = p[9];
  }
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row = tile * FC_TM + 16 * wave + a16;
    const bool valid = row < a.N;
    float x[2][8] = {{nx0.x, nx0.y, nx0.z, nx0.w, nx1.x, nx1.y, nx1.z, nx1.w}, {nx2.x, nx2.y, nx2.z, nx2.w, nx3.x, nx3.y, nx3.z, nx3.w}};
    {   // the wave's rows of the workgroup's next tile
      const float4* p = x4 + rowc(tile + gridDim.x < ntiles ? tile + gridDim.x : tile) * 16 + 2 * g4;
      nx0 = p[0]; nx1 = p[1]; nx2 = p[8]; nx3 = p[9];
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      constexpr int dummy_ = 0; (void)dummy_;
      const bool last = l == NL - 1;
      // row scale S = 2^(140 - e), 2^(e - 127) <= max |x| < 2^(e - 126): S max |x| in [2^13, 2^14); zero / non-finite rows: 1
      float m = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 8; ++t) m = fmaxf(m, fabsf(x[s][t]));
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const int ef = (__builtin_bit_cast(int, m) >> 23) & 255;
      const int sb = (ef == 0 || ef == 255 || !(m == m)) ? 127 : min(267 - ef, 253);
      const float S = __builtin_bit_cast(float, sb << 23);
      const float back = __builtin_bit_cast(float, (254 - sb) << 23) * (1.0f / 256.0f);      // 1 / (S 2^8)
      u32x4 bh[2], bl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned h_, l_;
          split2_pair(S * x[s][2 * j], S * x[s][2 * j + 1], h_, l_);
          bh[s][j] = h_; bl[s][j] = l_;
        }
      const u32x4* wl_ = ws + l * 1024 + lane;
      const float bscale = 256.0f * S;
#pragma unroll
      for (int T = 0; T < (last ? 2 : 4); ++T) {
        const float4 bias = *reinterpret_cast<const float4*>(sB + l * FC_F + 32 * (T >> 1) + 8 * g4 + 4 * (T & 1));
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc1 = {bscale * bias.x, bscale * bias.y, bscale * bias.z, bscale * bias.w};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const u32x4 wh = wl_[((T * 2 + s) * 2 + 0) * 64], wlo = wl_[((T * 2 + s) * 2 + 1) * 64];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wlo), __builtin_bit_cast(f16x8, bh[s]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, bh[s]), acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, bl[s]), acc0, 0, 0, 0);
        }
        const f32x4 v = (acc0 + acc1) * back;
        const float4 sv = act4(ACT >= 0 ? ACT : a.act, v);
        if (!last) {
          float* xo = &x[T >> 1][4 * (T & 1)];
          xo[0] += sv.x; xo[1] += sv.y; xo[2] += sv.z; xo[3] += sv.w;       // the residual input is the slot the output takes
        } else if (valid) {
          *reinterpret_cast<float4*>(a.g + row * FC_H + 8 * g4 + 4 * T) = sv;
        }
      }
      if (!last && a.p.y[l] && valid) {
        float4* yo = reinterpret_cast<float4*>(a.p.y[l] + row * FC_F + 8 * g4);
        yo[0] = make_float4(x[0][0], x[0][1], x[0][2], x[0][3]); yo[1] = make_float4(x[0][4], x[0][5], x[0][6], x[0][7]);
        yo[8] = make_float4(x[1][0], x[1][1], x[1][2], x[1][3]); yo[9] = make_float4(x[1][4], x[1][5], x[1][6], x[1][7]);
      }
    }
  }
}


// in fc_fwd_kernel:   constexpr size_t BODY = max(FC_TM * FC_LD * 4 + 4 * FC_PLANE, ((NL - 1) * 1024 + 512) * 16);  sB = fsm + BODY;
//                     if (a.act == NG_ACT_SOFTPLUS) fc_fwd_body_wave<NL, NG_ACT_SOFTPLUS>(a, fsm, sB); else fc_fwd_body_wave<NL, -1>(a, fsm, sB);
// fc_fwd_lds_bytes(L) = max(64 * 68 * 4 + 4 * FC_PLANE, ((L - 1) * 1024 + 512) * 16) + L * 64 * 4 + 16
