// libmgacbam.so, C ABI (include/mgacbam.h): MaskCBAM backward (mgacbam_backward[_stages])
#include "host.cuh"
#include "bwd.cuh"

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
static int backward_args(const mgacbam_bwd_level_t& L, BwdArgs& A, Sig& sig) {
  if (!L.x || !L.gy || !L.ctx || !L.scratch || !L.gx) return fail(MGACBAM_E_NULL, "backward: x / gy / ctx / scratch / gx is NULL");
  if (!L.gw1 || !L.gb1 || !L.gw2 || !L.gb2 || !L.gwsa || !L.gbeta) return fail(MGACBAM_E_NULL, "backward: NULL parameter-gradient pointer");
  if (L.gmask && !L.mask) return fail(MGACBAM_E_NULL, "backward: gmask requested but mask is NULL");
  if (int e = check_params(L.p)) return e;
  if (int e = check_shape(L.B, L.C, L.H, L.W, L.p.hidden, L.p.k)) return e;
  if (L.dtype < MGACBAM_F32 || L.dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "backward: dtype %d", L.dtype);
  const int VEC = vec_of(L.H, L.W, L.dtype);
  const size_t need = VEC * elem_size(L.dtype);
  if (!aligned_to(L.x, need) || !aligned_to(L.gy, need) || !aligned_to(L.gx, need) || !aligned_to(L.ctx, 16) ||
      !aligned_to(L.scratch, 16) || (L.gmask && !aligned_to(L.gmask, 16)))
    return fail(MGACBAM_E_ALIGN, "backward: x/gy/gx must be %zu-byte aligned, ctx/scratch 16-byte", need);
  A.x = L.x; A.mask = L.mask; A.gy = L.gy; A.gx = L.gx; A.gmask = L.gmask;
  A.gw1 = L.gw1; A.gb1 = L.gb1; A.gw2 = L.gw2; A.gb2 = L.gb2; A.gwsa = L.gwsa; A.gbeta = L.gbeta;
  A.c = ctx_ptrs(const_cast<void*>(L.ctx), L.B, L.C, L.H, L.W, L.p.hidden);
  A.p = make_params(L.p);
  A.g = make_geo(L.B, L.C, L.H, L.W, L.p);
  A.t = choose_tune(L.B, L.C, L.H, L.W, L.p.k, L.dtype);
  const ScratchLayout SL = scratch_layout(L.B, L.C, L.H, L.W, L.p.hidden, L.p.k, L.dtype);
  {
    mgacbam_ctx_layout_t CL;
    ctx_layout(L.B, L.C, L.H, L.W, L.p.hidden, &CL);
    if (int e = check_capacity("backward", "ctx", static_cast<size_t>(CL.total), L.ctx_bytes)) return e;
    if (int e = check_capacity("backward", "scratch", SL.total, L.scratch_bytes)) return e;
  }
  char* sp = static_cast<char*>(L.scratch);
  A.s.A_part = reinterpret_cast<float*>(sp + SL.A_part);
  A.s.gpre = reinterpret_cast<float*>(sp + SL.gpre); A.s.gplanes = reinterpret_cast<float*>(sp + SL.gplanes);
  A.s.gwsa_part = reinterpret_cast<float*>(sp + SL.gwsa_part);
  A.s.gz = reinterpret_cast<float*>(sp + SL.gz); A.s.gbq = reinterpret_cast<float*>(sp + SL.gbq);
  A.s.gh_avg = reinterpret_cast<float*>(sp + SL.gh_avg); A.s.gh_mx = reinterpret_cast<float*>(sp + SL.gh_mx);
  A.s.pgh = reinterpret_cast<float*>(sp + SL.pgh);
  A.nt = chan_tiles(A.t, A.g.H, A.g.W, VEC);
  A.nconv = A.g.B * conv_tiles(A.t, A.g.H, A.g.W);
  A.nwsa = A.g.B * wsa_tiles(A.t, A.g.H, A.g.W);
  A.nrole = A.nwsa;
  A.wsa_tail = 0;
  A.npg = params_blocks(A.g);
  A.ncg = 0;
  A.nflag = static_cast<int>(sync_flags(static_cast<size_t>(L.H) * L.W));
  A.bflag0 = L.B * A.nflag + 4 + L.B;
  A.cflag0 = A.bflag0 + L.B * A.nflag;
  A.mbflag0 = A.cflag0 + L.B * A.nflag; A.mcflag0 = A.mbflag0 + L.B * A.nflag;
  A.wflag0 = A.mcflag0 + L.B * A.nflag; A.sflag0 = A.wflag0 + L.B * A.nflag;
  A.merged = 0;
  A.vec = VEC;
  { const Knobs kn = knobs(); A.trace = kn.trace; A.spin_limit = kn.spin_limit; }
  const int proj = (L.flags & MGACBAM_BWD_HAVE_PROJ) && L.gmask != nullptr;
  A.g.proj_h = (proj && L.p.hidden <= MGACBAM_PROJ_MAX_HIDDEN) ? L.p.hidden : 0;
  sig = Sig{L.dtype, VEC, L.mask != nullptr, L.p.k, L.gmask != nullptr, proj};
  return 0;
}

static int backward_group(BwdArgs* lv, int n, const Sig& sig, int stages, hipStream_t st) {
  Group<BwdArgs> G;
  G.n = n;
  // k_bwd_reduce2 re-reads three planes (g_planes x 2, cidx) per channel group: 4 channels per row halve that share of its loads
  // (config 4: 88 -> 80 us) whenever the grid still fills the chip; k_pool (one mask plane per group) measured slower with 4
  const int r2max = knobs().r2_cpt == 1 || knobs().r2_cpt == 2 || knobs().r2_cpt == 4 ? knobs().r2_cpt : 4;
  const int cpt = group_cpt(lv, n, r2max);
  for (int l = 0; l < n; ++l) {
    lv[l].t.pool_cpt = cpt;
    const int cpb = (kBlock / lv[l].t.pool_tx) * cpt;
    lv[l].ncg = (lv[l].g.C + cpb - 1) / cpb;
    G.lv[l] = lv[l];
  }
  auto fill = [&](auto blocks_of) { int tot = 0; for (int l = 0; l < n; ++l) { G.start[l] = tot; tot += blocks_of(lv[l]); } G.start[n] = tot; return tot; };

  // MGACBAM_BWD_FOLD: transposed conv as trailing role workgroups of the k_bwd_reduce1 launch (whole backward in this call, a tile at
  // least one image row and at least kSyncPx pixels -- one flag per tile in ctx.sync -- and few tiles per conv window)
  // (the conv tiles are the LAST workgroups of the launch and wait only for lower-numbered producers, which never wait themselves:
  //  progress does not depend on residency; the span bound is a speed heuristic)
  bool fold = (stages & MGACBAM_BWD_FOLD) && (stages & MGACBAM_BWD_REDUCE1) && (stages & MGACBAM_BWD_CONVT) && knobs().bwd_fold;
  for (int l = 0; l < n && fold; ++l) {
    const int TP = lv[l].t.chan_tx * sig.vec;
    fold = TP >= kSyncPx && TP >= lv[l].g.W && 8 * (((lv[l].t.conv_th + lv[l].g.k) * lv[l].g.W + TP - 1) / TP + 1) <= 512 &&
           lv[l].nconv <= lv[l].g.B * lv[l].nflag;                 // one flag per conv tile fits the region reserved in ctx.sync
  }
  // k_bwd_r12: the folded launch AND k_bwd_reduce2 with its dWsa roles as one launch (bwd.cuh), when both stages are in this call
  const bool fuse_early = (stages & MGACBAM_BWD_FUSE) != 0;
  const bool tail_early = fuse_early && (stages & MGACBAM_BWD_APPLY) && (stages & MGACBAM_BWD_PARAMGRAD) && (stages & MGACBAM_BWD_WSA) && sig.k == 7 &&
                          knobs().wsa_tail;
  bool merge = fold && fuse_early && (stages & MGACBAM_BWD_REDUCE2) && (stages & MGACBAM_BWD_WSA) && sig.k == 7 && !tail_early && knobs().bwd_merge &&
               !knobs().wsa_fat;
  for (int l = 0; l < n && merge; ++l) merge = lv[l].nwsa <= lv[l].g.B * lv[l].nflag && lv[l].ncg <= lv[l].g.C && lv[l].nconv % lv[l].g.B == 0;
  if (merge) {
    R12Group R;
    R.g.n = n;
    size_t smem = 0;
    for (int l = 0; l < n; ++l) {
      lv[l].merged = 1; lv[l].bflag0 = lv[l].mbflag0; lv[l].cflag0 = lv[l].mcflag0; lv[l].nrole = lv[l].nwsa;
      R.g.lv[l] = lv[l];
      smem = std::max({smem, reduce1_smem(lv[l].g, sig.vec), convT_smem(lv[l].t, sig.k), wsa_smem(lv[l].t, sig.k),
                       (64 + static_cast<size_t>(std::max(kPghLds, kBlock / lv[l].t.pool_tx))) * sizeof(float)});
    }
    int tot = 0;
    for (int p = 0; p < 4; ++p) {
      for (int l = 0; l < n; ++l) {
        R.seg[p][l] = tot;
        tot += p == 0 ? xcd_grid(lv[l].g.B, lv[l].nt) : p == 1 ? pad8(lv[l].nconv) : p == 2 ? pad8(lv[l].nwsa) : sweep_blocks(lv[l], lv[l].t.pool_tx, cpt);
      }
      R.seg[p][n] = tot;
    }
    const int grid = tot;
#define CALL_R12B(CPTV) LAUNCH((k_bwd_r12<TT, VV, CPTV>), grid, smem, st, R)
#define CALL_R12(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(cpt, CALL_R12B); }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_R12);
#undef CALL_R12
#undef CALL_R12B
    if (int e = launch_status("k_bwd_r12")) return e;
    for (int l = 0; l < n; ++l) G.lv[l] = lv[l];                  // (the later launches of this call see the same level state)
  }
  if (fold && !merge) {
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max({smem, reduce1_smem(lv[l].g, sig.vec), convT_smem(lv[l].t, sig.k)});
    const int grid = fill([&](const BwdArgs& a) { return xcd_grid(a.g.B, a.nt) + pad8(a.nconv); });
#define CALL_R1F(Tt, Vv) if (sig.k == 7) LAUNCH((k_bwd_reduce1_fold<Tt, Vv, 7>), grid, smem, st, G); else LAUNCH((k_bwd_reduce1_fold<Tt, Vv, 0>), grid, smem, st, G)
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_R1F);
#undef CALL_R1F
    if (int e = launch_status("k_bwd_reduce1_fold")) return e;
  }
  if ((stages & MGACBAM_BWD_REDUCE1) && !fold) {  // 1. per-(b,c) and per-pixel reductions of gy*x
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, reduce1_smem(lv[l].g, sig.vec));
    const int grid = fill([&](const BwdArgs& a) { return xcd_grid(a.g.B, a.nt); });
#define CALL_R1(Tt, Vv) LAUNCH((k_bwd_reduce1<Tt, Vv>), grid, smem, st, G)
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_R1);
#undef CALL_R1
    if (int e = launch_status("k_bwd_reduce1")) return e;
  }
  if ((stages & MGACBAM_BWD_CONVT) && !fold) {  // 2. transposed conv
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, convT_smem(lv[l].t, sig.k));
    const int grid = fill([&](const BwdArgs& a) { return a.nconv; });
    switch (sig.k) {
      case 3: LAUNCH(k_bwd_convT<3>, grid, smem, st, G); break;
      case 5: LAUNCH(k_bwd_convT<5>, grid, smem, st, G); break;
      case 7: LAUNCH(k_bwd_convT<7>, grid, smem, st, G); break;
      default: LAUNCH(k_bwd_convT<0>, grid, smem, st, G); break;
    }
    if (int e = launch_status("k_bwd_convT")) return e;
  }
  const bool fuse = (stages & MGACBAM_BWD_FUSE) != 0;
  const bool fuse_pg = fuse && (stages & MGACBAM_BWD_APPLY) && (stages & MGACBAM_BWD_PARAMGRAD);
  // dWsa tile partials: leading roles of k_bwd_reduce2; or (opt-in knob MGACBAM_WSA_TAIL, measured slower: bwd.cuh) the LAST workgroups
  // of the k_bwd_apply launch when ctx.sync is the zero-filled hand-off state (MGACBAM_BWD_FOLD's contract: the arrival counters live there)
  // (a staged caller asks for it by passing MGACBAM_BWD_WSA with the APPLY call instead of the REDUCE2 call: bench.py's per-kernel timing)
  const bool wsa_tail = fuse_pg && (stages & MGACBAM_BWD_FOLD) && (stages & MGACBAM_BWD_WSA) && sig.k == 7 && knobs().wsa_tail;
  const bool fuse_wsa = fuse && (stages & MGACBAM_BWD_REDUCE2) && (stages & MGACBAM_BWD_WSA) && sig.k == 7 && !wsa_tail;
  if (wsa_tail) for (int l = 0; l < n; ++l) { lv[l].wsa_tail = 1; G.lv[l].wsa_tail = 1; }
  if ((stages & MGACBAM_BWD_REDUCE2) && !merge) {  // 3. rest of g_ca (needs g_planes), g_z [+ dWsa partials as role workgroups]
    size_t smem = 0;
    for (int l = 0; l < n; ++l) {
      smem = std::max(smem, (64 + static_cast<size_t>(std::max(kPghLds, kBlock / lv[l].t.pool_tx))) * sizeof(float));
      if (fuse_wsa) smem = std::max(smem, wsa_smem(lv[l].t, sig.k));
    }
    if (fuse_wsa && knobs().wsa_fat) {
      // experiment (MGACBAM_WSA_FAT=1, off by default): as many dWsa roles as the streaming workgroups leave slots idle (config 2: 1792 of
      // 2048), each working through several tiles, instead of one thin role per tile.  Measured at configs 2 and 3: no change (26.9-27.2 us
      // either way) -- what the roles add to the launch (5.5 us over the role-free kernel) is not slot displacement
      int slots = 0;
#define RES_R22(CPTV) slots = resident_workgroups(k_bwd_reduce2<TT, VV, CPTV, true>, smem)
#define RES_R2(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(cpt, RES_R22); }
      DISPATCH_T_VEC(sig.dtype, sig.vec, RES_R2);
#undef RES_R2
#undef RES_R22
      long long streaming = 0, tiles = 0;
      for (int l = 0; l < n; ++l) {
        const int cpb = (kBlock / lv[l].t.pool_tx) * cpt;
        streaming += static_cast<long long>(lv[l].g.B) * ((lv[l].g.C + cpb - 1) / cpb);
        tiles += lv[l].nwsa;
      }
      const long long idle = slots - streaming;
      if (idle >= 32 && idle < tiles) {
        for (int l = 0; l < n; ++l) {
          lv[l].nrole = static_cast<int>(std::max(1ll, std::min<long long>(lv[l].nwsa, idle * lv[l].nwsa / tiles)));
          G.lv[l].nrole = lv[l].nrole;
        }
      }
    }
    const int grid = fill([&](const BwdArgs& a) { return (fuse_wsa ? pad8(a.nrole) : 0) + sweep_blocks(a, a.t.pool_tx, cpt); });
#define CALL_R22(CPTV) if (fuse_wsa) LAUNCH((k_bwd_reduce2<TT, VV, CPTV, true>), grid, smem, st, G); else LAUNCH((k_bwd_reduce2<TT, VV, CPTV, false>), grid, smem, st, G)
#define CALL_R2(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(cpt, CALL_R22); }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_R2);
#undef CALL_R2
#undef CALL_R22
    if (int e = launch_status("k_bwd_reduce2")) return e;
  }
  if ((stages & MGACBAM_BWD_WSA) && !fuse_wsa && !wsa_tail && !merge) {  // 4. dWsa tile partials (depends on stage 1 only)
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, wsa_smem(lv[l].t, sig.k));
    const int grid = fill([&](const BwdArgs& a) { return a.nwsa; });
    switch (sig.k) {
      case 3: LAUNCH(k_bwd_wsa<3>, grid, smem, st, G); break;
      case 5: LAUNCH(k_bwd_wsa<5>, grid, smem, st, G); break;
      case 7: LAUNCH(k_bwd_wsa<7>, grid, smem, st, G); break;
      default: LAUNCH(k_bwd_wsa<0>, grid, smem, st, G); break;
    }
    if (int e = launch_status("k_bwd_wsa")) return e;
  }
  if ((stages & MGACBAM_BWD_PARAMGRAD) && !fuse_pg) {  // 5. every parameter gradient
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, params_smem(lv[l].g));
    const int grid = fill([&](const BwdArgs& a) { return a.npg; });
    LAUNCH(k_bwd_params, grid, smem, st, G);
    if (int e = launch_status("k_bwd_params")) return e;
  }
  if (stages & MGACBAM_BWD_APPLY) {  // 6. gx (+ gmask) [+ parameter gradients as role workgroups]
    size_t smem = 0;
    for (int l = 0; l < n; ++l) {
      smem = std::max(smem, bwd_apply_smem(lv[l].g, sig.vec));
      if (fuse_pg) smem = std::max(smem, params_smem(lv[l].g));
      if (wsa_tail) smem = std::max(smem, wsa_smem(lv[l].t, sig.k));
    }
    const int grid = fill([&](const BwdArgs& a) { return (fuse_pg ? pad8(a.npg) : 0) + xcd_grid(a.g.B, a.nt) + (wsa_tail ? a.nwsa + (3 * a.g.k * a.g.k + 3) / 4 : 0); });
#define CALL_AP2(GM) if (fuse_pg) LAUNCH((k_bwd_apply<TT, VV, GM, true>), grid, smem, st, G); else LAUNCH((k_bwd_apply<TT, VV, GM, false>), grid, smem, st, G)
#define CALL_AP(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; if (sig.gmask) { CALL_AP2(true); } else { CALL_AP2(false); } }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_AP);
#undef CALL_AP
#undef CALL_AP2
    if (int e = launch_status("k_bwd_apply")) return e;
  }
  return 0;
}

extern "C" int mgacbam_backward_stages(const mgacbam_bwd_level_t* levels, int n_levels, int stages, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  hipStream_t st = static_cast<hipStream_t>(stream);
  BwdArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l)
    if (int e = backward_args(levels[l], args[l], sigs[l])) return e;
  if (int e = for_each_group(args, sigs, n_levels, [&](BwdArgs* g, int m, const Sig& s) { return backward_group(g, m, s, stages, st); })) return e;
  g_err[0] = 0;
  return 0;
}
extern "C" int mgacbam_backward(const mgacbam_bwd_level_t* levels, int n_levels, void* stream) {
  return mgacbam_backward_stages(levels, n_levels, MGACBAM_BWD_ALL, stream);
}

