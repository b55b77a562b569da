// libmgacbam.so, C ABI (include/mgacbam.h): MaskCBAM forward (mgacbam_forward[_stages])
#include "host.cuh"
#include "fwd.cuh"

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static int forward_args(const mgacbam_fwd_level_t& L, FwdArgs& A, Sig& sig) {
  if (!L.x || !L.y || !L.ctx) return fail(MGACBAM_E_NULL, "forward: x / y / ctx is NULL");
  if (int e = check_params(L.p)) return e;
  if (int e = check_shape(L.B, L.C, L.H, L.W, L.p.hidden, L.p.k)) return e;
  if (L.dtype < MGACBAM_F32 || L.dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "forward: dtype %d", L.dtype);
  const int VEC = vec_of(L.H, L.W, L.dtype);
  const size_t need = VEC * elem_size(L.dtype);
  if (!aligned_to(L.x, need) || !aligned_to(L.y, need) || !aligned_to(L.ctx, 16) || (L.mask && !aligned_to(L.mask, 16)))
    return fail(MGACBAM_E_ALIGN, "forward: x/y must be %zu-byte aligned, ctx 16-byte, mask %d-byte", need, VEC * 4);
  {
    mgacbam_ctx_layout_t CL;
    ctx_layout(L.B, L.C, L.H, L.W, L.p.hidden, &CL);
    if (int e = check_capacity("forward", "ctx", static_cast<size_t>(CL.total), L.ctx_bytes)) return e;
  }
  A.x = L.x; A.mask = L.mask; A.y = L.y; A.fused = 0;
  { const Knobs kn = knobs(); A.trace = kn.trace; A.spin_limit = kn.spin_limit; A.fault = kn.fault; }
  A.nflag = static_cast<int>(sync_flags(static_cast<size_t>(L.H) * L.W));
  A.c = ctx_ptrs(L.ctx, L.B, L.C, L.H, L.W, L.p.hidden);
  A.p = make_params(L.p);
  A.g = make_geo(L.B, L.C, L.H, L.W, L.p);
  A.t = choose_tune(L.B, L.C, L.H, L.W, L.p.k, L.dtype);
  const int proj = (L.flags & MGACBAM_FWD_SAVE_PROJ) && L.mask != nullptr;
  A.g.proj_h = (proj && L.p.hidden <= MGACBAM_PROJ_MAX_HIDDEN) ? L.p.hidden : 0;
  sig = Sig{L.dtype, VEC, L.mask != nullptr, L.p.k, 0, proj};
  // fp16 / bf16: k_gate reads 16 bytes per lane (8 elements, kept packed in the registers) whatever vector width the other kernels
  // use -- twice the pixels per tile, half the workgroups: at YOLOv8n sizes the grid then runs as ONE resident round
  sig.gvec = VEC;
  if (L.dtype != MGACBAM_F32 && VEC == 4 && knobs().gate_h8 && (static_cast<long long>(L.H) * L.W) % 8 == 0) {
    Tune t8 = A.t;
    gate_geometry(L.C, L.H, L.W, L.p.k, 8, t8);
    if (t8.gate_tx > 0) { sig.gvec = 8; A.t = t8; }
  }
  return 0;
}

static int forward_group(FwdArgs* lv, int n, const Sig& sig, int stages, hipStream_t st) {
  Group<FwdArgs> G;
  G.n = n;
  const int pool_cpt = group_cpt(lv, n);
  for (int l = 0; l < n; ++l) { lv[l].t.pool_cpt = pool_cpt; G.lv[l] = lv[l]; }
  auto fill = [&](auto blocks_of) { int tot = 0; for (int l = 0; l < n; ++l) { G.start[l] = tot; tot += blocks_of(lv[l]); } G.start[n] = tot; return tot; };

  // MGACBAM_FWD_FUSE: stages 2 + 3 become ONE x-resident launch (k_gate) when every level of the group is eligible
  const int gvec = sig.gvec;                   // per level (forward_args), uniform over the group by construction
  bool gate = (stages & MGACBAM_FWD_FUSE) && (stages & MGACBAM_FWD_CHAN) && (stages & MGACBAM_FWD_APPLY) && !sig.proj && sig.vec <= 4 && gvec > 0;
  for (int l = 0; l < n && gate; ++l) gate = lv[l].t.gate_tx > 0;
  size_t gsmem = 0;
  if (gate) {
    // residency precondition of the in-launch hand-off, from the DEVICE (CU count x occupancy of the chosen instantiation): the
    // 8*span + 1 workgroups a tile's wait spans must be co-resident; half of the budget is left to whatever else runs on the chip
    int span = 0;
    for (int l = 0; l < n; ++l) { gsmem = std::max(gsmem, gate_smem(lv[l].g, lv[l].t, gvec)); span = std::max(span, lv[l].t.gate_span); }
    int resident = 0;
#define RES_GATE(Tt, Vv) resident = (sig.k == 7) ? resident_workgroups(k_gate<Tt, Vv, 7>, gsmem) : resident_workgroups(k_gate<Tt, Vv, 0>, gsmem)
    DISPATCH_T_VEC(sig.dtype, gvec, RES_GATE);
#undef RES_GATE
    gate = 2 * (8 * span + 1) <= resident;
  }
  if (gate) for (int l = 0; l < n; ++l) { lv[l].fused = 1; G.lv[l].fused = 1; }

  if (stages & MGACBAM_FWD_POOL) {  // 1. pooling
    const int grid = fill([&](const FwdArgs& a) { return sweep_blocks(a, a.t.pool_tx, pool_cpt); });
#define CALL_POOL2(CPTV) if (sig.has_mask) LAUNCH((k_pool<TT, VV, CPTV, true>), grid, 0, st, G); else LAUNCH((k_pool<TT, VV, CPTV, false>), grid, 0, st, G)
#define CALL_POOL(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(pool_cpt, CALL_POOL2); }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_POOL);
#undef CALL_POOL
#undef CALL_POOL2
    if (int e = launch_status("k_pool")) return e;
  }
  if (gate) {
    const size_t smem = gsmem;
    GateGroup GG;
    const int tiles = fill([&](const FwdArgs& a) { return xcd_grid(a.g.B, gate_tiles(a.t, a.g.H, a.g.W, gvec)); });
    GG.g = G;
    GG.nrole = 0;
    for (int l = 0; l < n; ++l) { GG.rstart[l] = GG.nrole; GG.nrole += lv[l].g.B; }
    GG.rstart[n] = GG.nrole;
    const int grid = GG.nrole + tiles;
#define CALL_GATE(Tt, Vv) if (sig.k == 7) LAUNCH((k_gate<Tt, Vv, 7>), grid, smem, st, GG); else LAUNCH((k_gate<Tt, Vv, 0>), grid, smem, st, GG)
    DISPATCH_T_VEC(sig.dtype, gvec, CALL_GATE);
#undef CALL_GATE
    return launch_status("k_gate");
  }
  if (stages & MGACBAM_FWD_CHAN) {  // 2. shared MLP + channel gate (prologue, or a launch of its own), channel max / mean planes
    // With C*hidden large the MLP prologue keeps every k_chan workgroup from streaming for 15-20 us; one tiny launch per step is cheaper
    bool split_mlp = false;
    for (int l = 0; l < n; ++l) split_mlp |= static_cast<long long>(lv[l].g.C) * lv[l].g.hidden >= 8192;
    { const int f = knobs().split_mlp; if (f == 0) split_mlp = false; else if (f == 1) split_mlp = true; }
    if (split_mlp) {
      size_t msmem = 0;
      for (int l = 0; l < n; ++l) msmem = std::max(msmem, (3 * static_cast<size_t>(lv[l].g.C) + 2 * lv[l].g.hidden) * sizeof(float));
      const int mgrid = fill([&](const FwdArgs& a) { return a.g.B; });
      LAUNCH(k_mlp, mgrid, msmem, st, G);
      if (int e = launch_status("k_mlp")) return e;
    }
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, chan_smem(lv[l].g, sig.vec, sig.proj));
    const int grid = fill([&](const FwdArgs& a) { return xcd_grid(a.g.B, (a.g.HW / sig.vec + a.t.chanf_tx - 1) / a.t.chanf_tx); });
#define CALL_CHAN(Tt, Vv)                                                                                         \
    if (split_mlp) { if (sig.proj) LAUNCH((k_chan<Tt, Vv, true, true>), grid, smem, st, G); else LAUNCH((k_chan<Tt, Vv, false, true>), grid, smem, st, G); } \
    else { if (sig.proj) LAUNCH((k_chan<Tt, Vv, true>), grid, smem, st, G); else LAUNCH((k_chan<Tt, Vv, false>), grid, smem, st, G); }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_CHAN);
#undef CALL_CHAN
    if (int e = launch_status("k_chan")) return e;
  }
  if (stages & MGACBAM_FWD_APPLY) {  // 3. k x k conv + spatial gate (prologue), both gates + alpha residual
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, apply_smem(lv[l].g, lv[l].t, sig.vec));
    const int grid = fill([&](const FwdArgs& a) { return xcd_grid(a.g.B, chan_tiles(a.t, a.g.H, a.g.W, sig.vec)); });
#define CALL_APPLY(Tt, Vv)                                                    \
    switch (sig.k) {                                                          \
      case 3: LAUNCH((k_apply<Tt, Vv, 3>), grid, smem, st, G); break;         \
      case 5: LAUNCH((k_apply<Tt, Vv, 5>), grid, smem, st, G); break;         \
      case 7: LAUNCH((k_apply<Tt, Vv, 7>), grid, smem, st, G); break;         \
      default: LAUNCH((k_apply<Tt, Vv, 0>), grid, smem, st, G); break;        \
    }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_APPLY);
#undef CALL_APPLY
    if (int e = launch_status("k_apply")) return e;
  }
  return 0;
}

extern "C" int mgacbam_forward_stages(const mgacbam_fwd_level_t* levels, int n_levels, int stages, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  hipStream_t st = static_cast<hipStream_t>(stream);
  FwdArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l)
    if (int e = forward_args(levels[l], args[l], sigs[l])) return e;
  if (int e = for_each_group(args, sigs, n_levels, [&](FwdArgs* g, int m, const Sig& s) { return forward_group(g, m, s, stages, st); })) return e;
  g_err[0] = 0;
  return 0;
}
extern "C" int mgacbam_forward(const mgacbam_fwd_level_t* levels, int n_levels, void* stream) {
  return mgacbam_forward_stages(levels, n_levels, MGACBAM_FWD_ALL, stream);
}

