// libmgacbam.so, C ABI (include/mgacbam.h): MGAMaskHead
#include "host.cuh"
#include "head.cuh"

// ------------------------------------------------------------------------------------------------
// MGAMaskHead
// ------------------------------------------------------------------------------------------------
static int head_check_shape(int B, int C, int H, int W, int hidden) {
  if (B < 1 || C < 1 || H < 1 || W < 1 || hidden < 1 || hidden > 1024 || C > 8192)
    return fail(MGACBAM_E_SHAPE, "mask head: bad shape B=%d C=%d H=%d W=%d hidden=%d", B, C, H, W, hidden);
  if (static_cast<long long>(H) * W > (1ll << 28) || static_cast<long long>(B) * std::max(C, hidden) * H * W > (1ll << 40))
    return fail(MGACBAM_E_SHAPE, "mask head: tensor too large B=%d C=%d H=%d W=%d", B, C, H, W);
  // the 3x3 kernels work on runs of pixels with a halo of W+1 either side (k_head_out: inside a run of at most 1024 pixels; k_head_bwd_act:
  // an LDS plane of run + halo): rows wider than that are refused here, not at launch; one sample's z block must fit a buffer descriptor
  { int ppt, opx, per;
    if (!head_out_shape(hidden, H * W, W, ppt, opx, per) || W > 500)
      return fail(MGACBAM_E_SHAPE, "mask head: image rows of W=%d do not fit the 3x3 kernels' pixel runs (W <= 500)", W);
    if (static_cast<long long>(hidden) * H * W >= (1ll << 30))
      return fail(MGACBAM_E_SHAPE, "mask head: hidden * H * W = %lld does not fit a buffer descriptor (< 2^30)", static_cast<long long>(hidden) * H * W); }
  return 0;
}
struct HeadTiling { int vec, hidp, cp, tile_px, tps, nwg, gx_tile_px, gx_tps, fw_kw, gx_kw, fw_mtw, gx_mtw, nwg_out, out_ppt, out_px, out_per, nwg1, act_ppt, act_hl, ncb, nshare, gw2; };
// wave arrangement of k_head_gemm (head.cuh): MW waves along M for `mtiles` 16-output tiles, KW waves along K when K is long (a chain of
// K/4 dependent steps otherwise), the rest along pixels
static void head_waves(int mtiles, int mtw, int K, int& pw, int& kw) {
  int mw = std::min(4, (mtiles + mtw - 1) / mtw);
  if (mw == 3) mw = 4;
  const int rest = 4 / mw;
  kw = K >= 128 ? rest : 1;
  pw = rest / kw;
}
static HeadTiling head_tiling(int B, int C, int H, int W, int hidden) {
  HeadTiling t;
  const int HW = H * W;
  t.vec = (HW % 4 == 0) ? 4 : 1;
  t.hidp = (hidden + 15) & ~15; t.cp = (C + 15) & ~15;
  int pw;
  t.fw_mtw = t.hidp / 16 > 8 ? 4 : 2;                               // forward: two accumulator tiles per wave (120 VGPRs, 4 workgroups per CU, every level of a YOLOv8n/s call in ONE launch); hidden > 128: four
  t.gx_mtw = 2;                                                     // gx: its B operand (g_a, z: E/4 each) is cheap to re-read; light workgroups
  head_waves(t.hidp / 16, t.fw_mtw, C, pw, t.fw_kw);
  t.tile_px = pw * 16 * t.vec;
  t.tps = (HW + t.tile_px - 1) / t.tile_px;
  t.nwg = B * t.tps;
  head_waves(t.cp / 16, t.gx_mtw, hidden, pw, t.gx_kw);
  t.gx_tile_px = pw * 16 * t.vec;
  t.gx_tps = (HW + t.gx_tile_px - 1) / t.gx_tile_px;
  head_out_shape(hidden, HW, W, t.out_ppt, t.out_px, t.out_per);
  t.nwg_out = B * t.out_per;
  t.act_ppt = HW >= 2048 ? 4 : (HW >= 512 ? 2 : 1);             // pixels per thread of k_head_bwd_act (amortises its per-channel reductions)
  t.nwg1 = B * ((HW + kBlock * t.act_ppt - 1) / (kBlock * t.act_ppt));
  t.act_hl = kBlock * t.act_ppt + 2 * (W + 1);
  t.ncb = (C + kHeadCB - 1) / kHeadCB;
  // pixel shares of k_head_bwd_gw: as many workgroups as ~4 MB of dW1 partials allow (32..256), and no more than there are pairs of pixel chunks
  const long long per_share = static_cast<long long>(t.ncb) * t.hidp * kHeadCB * 4;
  long long ns = (4ll << 20) / per_share;
  const long long chunks = static_cast<long long>(B) * ((HW + 4 * t.vec - 1) / (4 * t.vec));
  ns = std::min(ns, (chunks + 7) / 8);
  t.nshare = static_cast<int>(std::max(32ll, std::min(256ll, ns)));
  t.gw2 = (t.vec == 4 && t.hidp <= 64) ? 1 : 0;                     // k_head_bwd_gw2 (operands through LDS): a share = every nshare-th 64-pixel chunk
  if (t.gw2) {
    const long long chunks64 = static_cast<long long>(B) * ((HW + kHeadGwPx - 1) / kHeadGwPx);
    const long long cap = std::max(1ll, std::min(512ll, (4ll << 20) / per_share));
    t.nshare = static_cast<int>(std::max(1ll, std::min(cap, chunks64 / std::max(1, knobs().head_gw_div))));   // (measured: 2-4 chunks per workgroup and 512-2048 shares all within 1 %; 6+ chunks, one resident round: +12 %)
  }
  return t;
}
struct HeadCtxLayout { size_t z, mean, rstd, par, part, total; };
static HeadCtxLayout head_ctx_layout(int B, int C, int H, int W, int hidden) {
  const HeadTiling t = head_tiling(B, C, H, W, hidden);
  HeadCtxLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o = align16(o + n * 4); return at; };
  L.z = take(static_cast<size_t>(B) * hidden * H * W);
  L.mean = take(t.hidp); L.rstd = take(t.hidp);
  L.par = take(static_cast<size_t>(t.hidp) * kHeadPar);
  L.part = take(static_cast<size_t>(t.nwg) * 2 * t.hidp);
  L.total = o;
  return L;
}
struct HeadScratchLayout { size_t ga, part1, kst, gwpart, total; };
static HeadScratchLayout head_scratch_layout(int B, int C, int H, int W, int hidden) {
  const HeadTiling t = head_tiling(B, C, H, W, hidden);
  HeadScratchLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o = align16(o + n * 4); return at; };
  L.ga = take(static_cast<size_t>(B) * hidden * H * W);
  L.part1 = take(static_cast<size_t>(t.nwg1) * t.hidp * kHeadNStat);
  L.kst = take(5 * static_cast<size_t>(t.hidp));
  L.gwpart = take(static_cast<size_t>(t.ncb) * t.nshare * t.hidp * kHeadCB);
  L.total = o;
  return L;
}
extern "C" size_t mgahead_ctx_bytes(int B, int C, int H, int W, int hidden) {
  if (head_check_shape(B, C, H, W, hidden)) return 0;
  return head_ctx_layout(B, C, H, W, hidden).total;
}
extern "C" size_t mgahead_bwd_scratch_bytes(int B, int C, int H, int W, int hidden) {
  if (head_check_shape(B, C, H, W, hidden)) return 0;
  return head_scratch_layout(B, C, H, W, hidden).total;
}
static int head_common(const mgahead_params_t& P, int B, int C, int H, int W, int dtype, void* ctx, HeadArgs& A, Sig& sig) {
  if (!P.w1 || !P.bn_weight || !P.bn_bias || !P.running_mean || !P.running_var || !P.wh || !P.bh)
    return fail(MGACBAM_E_NULL, "mask head: NULL parameter pointer");
  if (int e = head_check_shape(B, C, H, W, P.hidden)) return e;
  if (dtype < MGACBAM_F32 || dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "mask head: dtype %d", dtype);
  if (!(P.eps > 0.f) || !(P.momentum >= 0.f && P.momentum <= 1.f)) return fail(MGACBAM_E_SHAPE, "mask head: eps=%g momentum=%g", P.eps, P.momentum);
  const HeadTiling t = head_tiling(B, C, H, W, P.hidden);
  memset(&A, 0, sizeof(A));
  A.p = HeadPtrs{P.w1, P.bn_weight, P.bn_bias, P.running_mean, P.running_var, reinterpret_cast<long long*>(P.num_batches_tracked), P.wh, P.bh};
  A.g.B = B; A.g.C = C; A.g.hid = P.hidden; A.g.H = H; A.g.W = W; A.g.HW = H * W; A.g.hidp = t.hidp; A.g.cp = t.cp;
  A.g.eps = P.eps; A.g.momentum = P.momentum; A.g.training = P.training ? 1 : 0;
  const HeadCtxLayout L = head_ctx_layout(B, C, H, W, P.hidden);
  char* cp = static_cast<char*>(ctx);
  A.c = HeadCtx{reinterpret_cast<float*>(cp + L.z), reinterpret_cast<float*>(cp + L.mean), reinterpret_cast<float*>(cp + L.rstd),
                reinterpret_cast<float*>(cp + L.par), reinterpret_cast<float*>(cp + L.part)};
  A.tile_px = t.tile_px; A.tiles_per_sample = t.tps; A.nwg = t.nwg;
  A.gx_tile_px = t.gx_tile_px; A.gx_tiles_per_sample = t.gx_tps; A.fw_kw = t.fw_kw; A.gx_kw = t.gx_kw; A.fw_mtw = t.fw_mtw; A.gx_mtw = t.gx_mtw;
  A.trace = knobs().trace; A.trace_base = 0;
  A.nwg_out = t.nwg_out; A.out_ppt = t.out_ppt; A.out_px = t.out_px; A.out_per = t.out_per; A.nwg1 = t.nwg1; A.act_ppt = t.act_ppt; A.act_hl_max = t.act_hl; A.ncb = t.ncb; A.nshare = t.nshare; A.gw2 = t.gw2;
  sig = Sig{dtype, t.vec, 0, 0, 0, 0};
  return 0;
}
static size_t head_gemm_smem(const HeadArgs* lv, int n) {
  size_t m = 0;
  for (int l = 0; l < n; ++l) m = std::max(m, (static_cast<size_t>(kHeadLdsX) + 8 * lv[l].g.hidp) * sizeof(float));
  return m;
}
template <typename Fn>
static int head_fill(Group<HeadArgs>& G, const HeadArgs* lv, int n, Fn blocks_of) {
  int tot = 0;
  for (int l = 0; l < n; ++l) { G.start[l] = tot; tot += blocks_of(lv[l]); }
  G.start[n] = tot;
  return tot;
}
static int head_forward_group(HeadArgs* lv, int n, const Sig& sig, hipStream_t st) {
  Group<HeadArgs> G;
  G.n = n;
  for (int l = 0; l < n; ++l) G.lv[l] = lv[l];
  {
    // one launch per accumulator template (two tiles per wave: hidden <= 128, i.e. every level of the n/s models; four beyond): each level
    // alone is latency-bound, so levels that share a launch overlap each other
    const size_t smem = head_gemm_smem(lv, n);
    for (int pass = 0; pass < 2; ++pass) {
      Group<HeadArgs> Gm;
      Gm.n = 0;
      int grid = 0, mtw = 1;
      for (int l = 0; l < n; ++l)
        if ((lv[l].fw_mtw <= 2) == (pass == 0)) {
          Gm.lv[Gm.n] = lv[l]; Gm.start[Gm.n] = grid; grid += lv[l].nwg; ++Gm.n;
          mtw = std::max(mtw, lv[l].fw_mtw);
        }
      if (!Gm.n) continue;
      Gm.start[Gm.n] = grid;
      for (int l = 0; l < Gm.n; ++l) Gm.lv[l].trace_base = pass * 8192;
      for (int l = 0; l < Gm.n; ++l) {                            // the wave arrangement follows the template the level runs under
        int pw;
        head_waves(Gm.lv[l].g.hidp / 16, mtw, Gm.lv[l].g.C, pw, Gm.lv[l].fw_kw);
        if (pw * 16 * sig.vec != Gm.lv[l].tile_px) return fail(MGACBAM_E_SHAPE, "mask head: inconsistent tiling");   // (cannot happen: see head_tiling)
      }
#define CALL_HP3(Tt, Vv, Mm) LAUNCH((k_head_gemm<Tt, Vv, false, Mm>), grid, smem, st, Gm)
#define CALL_HP(Tt, Vv) { if (mtw <= 2) { CALL_HP3(Tt, Vv, 2); } else { CALL_HP3(Tt, Vv, 4); } }
      if (sig.dtype == MGACBAM_F32) { if (sig.vec == 4) { CALL_HP(float, 4); } else { CALL_HP(float, 1); } }
      else if (sig.dtype == MGACBAM_F16) { if (sig.vec == 4) { CALL_HP(__half, 4); } else { CALL_HP(__half, 1); } }
      else { if (sig.vec == 4) { CALL_HP(bf16_t, 4); } else { CALL_HP(bf16_t, 1); } }
#undef CALL_HP
#undef CALL_HP3
      if (int e = launch_status("k_head_gemm<fwd>")) return e;
    }
  }
  {
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.g.hid; });
    LAUNCH(k_head_stats, grid, 0, st, G);
    if (int e = launch_status("k_head_stats")) return e;
  }
  {
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.nwg_out; });
    const size_t smem = 0;
#define CALL_HO(Tt) { if (sig.vec == 4) { LAUNCH((k_head_out<Tt, 4>), grid, smem, st, G); } else { LAUNCH((k_head_out<Tt, 1>), grid, smem, st, G); } }
    switch (sig.lf32 ? MGACBAM_F32 : sig.dtype) {              // (the kernel's element type is the LOGITS' type: z is fp32)
      case MGACBAM_F32: CALL_HO(float); break;
      case MGACBAM_F16: CALL_HO(__half); break;
      default: CALL_HO(bf16_t); break;
    }
#undef CALL_HO
    if (int e = launch_status("k_head_out")) return e;
  }
  return 0;
}
extern "C" int mgahead_forward(const mgahead_fwd_level_t* levels, int n_levels, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  HeadArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    const mgahead_fwd_level_t& L = levels[l];
    if (!L.x || !L.logits || !L.ctx) return fail(MGACBAM_E_NULL, "mask head forward: x / logits / ctx is NULL");
    if (int e = head_common(L.p, L.B, L.C, L.H, L.W, L.dtype, L.ctx, args[l], sigs[l])) return e;
    if (int e = check_capacity("mask head forward", "ctx", head_ctx_layout(L.B, L.C, L.H, L.W, L.p.hidden).total, L.ctx_bytes)) return e;
    const size_t need = sigs[l].vec * elem_size(L.dtype);
    if (!aligned_to(L.x, need) || !aligned_to(L.ctx, 16)) return fail(MGACBAM_E_ALIGN, "mask head forward: x must be %zu-byte aligned, ctx 16-byte", need);
    args[l].x = L.x; args[l].logits = L.logits;
    sigs[l].lf32 = (L.flags & MGAHEAD_LOGITS_F32) ? 1 : 0;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int e = for_each_group(args, sigs, n_levels, [&](HeadArgs* g, int m, const Sig& s) { return head_forward_group(g, m, s, st); })) return e;
  g_err[0] = 0;
  return 0;
}
static int head_backward_group(HeadArgs* lv, int n, const Sig& sig, hipStream_t st) {
  Group<HeadArgs> G;
  G.n = n;
  for (int l = 0; l < n; ++l) G.lv[l] = lv[l];
  {
    int hl = 0;
    for (int l = 0; l < n; ++l) hl = std::max(hl, lv[l].act_hl_max);
    for (int l = 0; l < n; ++l) { lv[l].act_hl_max = hl; G.lv[l].act_hl_max = hl; }      // the reduction scratch sits behind the launch's longest run
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.nwg1 * ((a.g.hid + kHeadJC - 1) / kHeadJC); });
    const size_t smem = (static_cast<size_t>(hl) + 16 * kHeadJC * kHeadNStat) * sizeof(float);
    switch (sig.lf32 ? MGACBAM_F32 : sig.dtype) {              // (the kernel's element type is g_logits' type)
      case MGACBAM_F32: LAUNCH(k_head_bwd_act<float>, grid, smem, st, G); break;
      case MGACBAM_F16: LAUNCH(k_head_bwd_act<__half>, grid, smem, st, G); break;
      default: LAUNCH(k_head_bwd_act<bf16_t>, grid, smem, st, G); break;
    }
    if (int e = launch_status("k_head_bwd_act")) return e;
  }
  {
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.g.hid; });
    LAUNCH(k_head_bwd_fin, grid, 0, st, G);
    if (int e = launch_status("k_head_bwd_fin")) return e;
  }
  {
    for (int l = 0; l < n; ++l) G.lv[l].trace_base = 16384;
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.g.B * a.gx_tiles_per_sample; });
    const size_t smem = head_gemm_smem(lv, n);
#define CALL_HX(Tt, Vv) LAUNCH((k_head_gemm<Tt, Vv, true, 2>), grid, smem, st, G)
    if (sig.dtype == MGACBAM_F32) { if (sig.vec == 4) { CALL_HX(float, 4); } else { CALL_HX(float, 1); } }
    else if (sig.dtype == MGACBAM_F16) { if (sig.vec == 4) { CALL_HX(__half, 4); } else { CALL_HX(__half, 1); } }
    else { if (sig.vec == 4) { CALL_HX(bf16_t, 4); } else { CALL_HX(bf16_t, 1); } }
#undef CALL_HX
    if (int e = launch_status("k_head_gemm<gx>")) return e;
  }
  {                                                               // dW1 partials
    for (int pass = 0; pass < 2; ++pass) {                         // pass 0: the levels that take the LDS-staged form, pass 1: the rest
      Group<HeadArgs> Gw;
      Gw.n = 0;
      int grid = 0;
      size_t smem = 0;
      for (int l = 0; l < n; ++l)
        if ((lv[l].gw2 != 0) == (pass == 0)) {
          Gw.lv[Gw.n] = G.lv[l]; Gw.start[Gw.n] = grid; grid += lv[l].ncb * lv[l].nshare; ++Gw.n;
          const size_t hp = static_cast<size_t>(lv[l].g.hidp);
          smem = std::max(smem, (pass == 0 ? 5 * hp + (kHeadCB + hp) * kHeadGwPitch : 5 * hp + 1024) * sizeof(float));
        }
      if (!Gw.n) continue;
      Gw.start[Gw.n] = grid;
      if (pass == 0) {
        switch (sig.dtype) {
          case MGACBAM_F32: LAUNCH(k_head_bwd_gw2<float>, grid, smem, st, Gw); break;
          case MGACBAM_F16: LAUNCH(k_head_bwd_gw2<__half>, grid, smem, st, Gw); break;
          default: LAUNCH(k_head_bwd_gw2<bf16_t>, grid, smem, st, Gw); break;
        }
        if (int e = launch_status("k_head_bwd_gw2")) return e;
        continue;
      }
#define CALL_HW(Tt, Vv) LAUNCH((k_head_bwd_gw<Tt, Vv>), grid, smem, st, Gw)
      if (sig.dtype == MGACBAM_F32) { if (sig.vec == 4) { CALL_HW(float, 4); } else { CALL_HW(float, 1); } }
      else if (sig.dtype == MGACBAM_F16) { if (sig.vec == 4) { CALL_HW(__half, 4); } else { CALL_HW(__half, 1); } }
      else { if (sig.vec == 4) { CALL_HW(bf16_t, 4); } else { CALL_HW(bf16_t, 1); } }
#undef CALL_HW
      if (int e = launch_status("k_head_bwd_gw")) return e;
    }
  }
  {
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return (a.g.hid * a.g.C + kHeadGwfOut - 1) / kHeadGwfOut; });
    LAUNCH(k_head_bwd_gwf, grid, 0, st, G);
    if (int e = launch_status("k_head_bwd_gwf")) return e;
  }
  return 0;
}
extern "C" int mgahead_backward(const mgahead_bwd_level_t* levels, int n_levels, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  HeadArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    const mgahead_bwd_level_t& L = levels[l];
    if (!L.x || !L.g_logits || !L.ctx || !L.scratch || !L.gx) return fail(MGACBAM_E_NULL, "mask head backward: x / g_logits / ctx / scratch / gx is NULL");
    if (!L.gw1 || !L.gbn_weight || !L.gbn_bias || !L.gwh || !L.gbh) return fail(MGACBAM_E_NULL, "mask head backward: NULL parameter-gradient pointer");
    if (int e = head_common(L.p, L.B, L.C, L.H, L.W, L.dtype, const_cast<void*>(L.ctx), args[l], sigs[l])) return e;
    const size_t need = sigs[l].vec * elem_size(L.dtype);
    if (!aligned_to(L.x, need) || !aligned_to(L.gx, need) || !aligned_to(L.ctx, 16) || !aligned_to(L.scratch, 16))
      return fail(MGACBAM_E_ALIGN, "mask head backward: x/gx must be %zu-byte aligned, ctx/scratch 16-byte", need);
    HeadArgs& A = args[l];
    A.x = L.x; A.gl = L.g_logits; A.gx = L.gx;
    A.gw1 = L.gw1; A.ggamma = L.gbn_weight; A.gbeta = L.gbn_bias; A.gwh = L.gwh; A.gbh = L.gbh;
    A.accum_gx = (L.flags & MGAHEAD_BWD_ACCUM_GX) ? 1 : 0;
    sigs[l].lf32 = (L.flags & MGAHEAD_LOGITS_F32) ? 1 : 0;
    A.gl2 = L.g_logits2;
    const HeadScratchLayout SL = head_scratch_layout(L.B, L.C, L.H, L.W, L.p.hidden);
    if (int e = check_capacity("mask head backward", "ctx", head_ctx_layout(L.B, L.C, L.H, L.W, L.p.hidden).total, L.ctx_bytes)) return e;
    if (int e = check_capacity("mask head backward", "scratch", SL.total, L.scratch_bytes)) return e;
    char* sp = static_cast<char*>(L.scratch);
    A.s = HeadScratch{reinterpret_cast<float*>(sp + SL.ga), reinterpret_cast<float*>(sp + SL.part1), reinterpret_cast<float*>(sp + SL.kst),
                      reinterpret_cast<float*>(sp + SL.gwpart)};
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int e = for_each_group(args, sigs, n_levels, [&](HeadArgs* g, int m, const Sig& s) { return head_backward_group(g, m, s, st); })) return e;
  g_err[0] = 0;
  return 0;
}

