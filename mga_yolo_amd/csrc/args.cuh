// Kernel argument blocks (passed by value) and the ctx / scratch layouts shared by host and device code.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace mgacbam {

struct Geo {
  int B, C, H, W, HW;
  int hidden, k;
  int use_sigmoid;
  float thr, eps;
  int proj_h;      // > 0: W1-projection planes of x are saved (forward) / available (backward); = hidden (<= kProjMax)
};
constexpr int kProjMax = 4;   // MGACBAM_PROJ_MAX_HIDDEN
#ifndef MGACBAM_GATE_R
#define MGACBAM_GATE_R 16
#endif
constexpr int kGateR = MGACBAM_GATE_R;   // channels a thread of the x-resident kernels keeps in registers (x VEC pixels each)
constexpr int kSyncPx = 16;   // a hand-off tile is at least this many pixels: ctx.sync holds ceil(HW/kSyncPx)+1 flags per sample

// saved statistics (device pointers into the caller's ctx buffer) -- see mgacbam_ctx_layout_t
struct CtxPtrs {
  float* S; float* use; float* den;
  float* avg; float* mx; float* mavg;
  int* valid; int* amax;
  float* h_avg; float* h_mx; float* ca;
  float* planes; int* cidx; float* sa;
  float* proj;     // (B, hidden, HW) when hidden <= kProjMax
  int* sync;       // in-launch hand-off state, all generation counters (zero-filled once by the caller, never reset): [B][nflag] k_gate tile flags,
                   // 4 status words ([0] = time-out), [B] per-sample ca flags, [B][nflag] k_bwd_reduce1 tile flags, [B][nflag] folded conv-tile flags
};

struct ParamPtrs { const float* w1; const float* b1; const float* w2; const float* b2; const float* wsa; const float* beta; };

// launch geometry chosen on the host (api.hip: choose_*); TX = lanes along H*W, power of two
struct Tune {
  int pool_tx;     // sweep kernels (k_pool, k_bwd_reduce2, k_eca_*): rows of TX lanes sweep H*W, TY = 256/TX rows x CPT channels
  int pool_cpt;
  int chan_tx;     // tile kernels (k_chan, k_apply, k_bwd_reduce1, k_bwd_apply, k_eca_bwd): one H*W vector per lane, TY = 256/TX channel slices
  int chanf_tx;    // k_chan alone (its tiles are independent of every other kernel's): TX lanes along H*W
  int conv_twq;    // backward conv tiles: TWQ quads (4 px) wide, TH rows
  int conv_th;
  int wsa_th;      // k_bwd_wsa tiles: fewer rows so 4 staged planes stay under ~16 KB of LDS (role workgroups)
  int apply_rows;  // k_apply: upper bound of the plane rows (tile rows + halo) staged per workgroup
  int nt_stores;   // 1: y / gx are written with non-temporal stores
  int gate_rows;   // k_gate: upper bound of the plane rows (tile rows + halo) staged per workgroup
  int gate_tx;     // k_gate (x-resident chan+apply): TX lanes along H*W, TY = 256/TX slices of kGateR channels each; 0 = not eligible
  int gate_span;   // k_gate: tiles a k x k window reaches on either side (a tile waits for workgroups up to 8*span ids away)
};

struct FwdArgs {
  const void* x; const float* mask; void* y;
  CtxPtrs c; ParamPtrs p; Geo g; Tune t;
  int nflag;       // flags per sample in c.sync (host: sync_flags(HW))
  int fused;       // MGACBAM_FWD_FUSE path selected for this level's group
  unsigned spin_limit;   // bound of every hand-off spin (knob MGACBAM_SPIN_LIMIT, default 2^20 ~ 1 s)
  int fault;       // fault injection for tests (knob MGACBAM_FAULT=1): sample 0's role workgroup never publishes ca
  long long* trace;   // MGACBAM_TRACE builds only: per-workgroup phase timestamps (tools/trace_gate.py), else nullptr
};

// transient backward buffers (device pointers into the caller's scratch buffer)
struct ScratchPtrs {
  float* A_part;    // (B,nt,2,C) per hw-tile partials: [0] A = sum gy*x*sa, [1] D = sum gy*(v-x) (k_bwd_reduce1 -> k_bwd_reduce2)
  float* gpre;      // (B,HW)    dL/d(conv output)
  float* gplanes;   // (B,3,HW)  dL/d(planes)
  float* gwsa_part; // (nconv, 3*k*k)
  float* gz;        // (B,C)     dL/dz
  float* gbq;       // (B,C)     D = sum_hw gy*(v - x)  (for dL/dbeta)
  float* gh_avg;    // (B,hidden)
  float* gh_mx;     // (B,hidden)
  float* pgh;       // (B,ncg,hidden) per-channel-group partials of W2^T g_z (k_bwd_reduce2 -> k_bwd_apply prologue)
};

struct BwdArgs {
  const void* x; const float* mask; const void* gy; void* gx; float* gmask;
  float* gw1; float* gb1; float* gw2; float* gb2; float* gwsa; float* gbeta;
  CtxPtrs c; ParamPtrs p; ScratchPtrs s; Geo g; Tune t;
  int nt;       // hw tiles of k_bwd_reduce1
  int nconv;    // k_bwd_convT tiles of this level (B * tiles_y * tiles_x)
  int nwsa;     // k_bwd_wsa tiles of this level: one dWsa partial each
  int nrole;    // role workgroups of k_bwd_reduce2 that work through those tiles (<= nwsa)
  int npg;      // parameter-gradient workgroups of this level (k_bwd_params roles)
  int ncg;      // channel groups per sample of k_bwd_reduce2
  int nflag;    // flags per sample in c.sync
  int bflag0;   // first backward hand-off flag in c.sync (ints): [B][nflag], one per k_bwd_reduce1 tile
  int cflag0;   // first flag of the folded transposed-conv tiles in c.sync: [nconv], generation counters like the tile flags
  int wflag0;   // ... of the dWsa tile roles when they ride in the merged launch: [nwsa]
  int sflag0;   // ... of k_bwd_reduce2's sweep workgroups in the merged launch: [B][C] (a sample's channel groups are its first ncg entries)
  int mbflag0, mcflag0;   // the merged launch's own tile / conv-tile flags (the host puts them into bflag0 / cflag0 for that launch)
  int merged;   // 1: k_bwd_reduce1 tiles, transposed-conv tiles, dWsa tiles and k_bwd_reduce2 sweeps are ONE launch (k_bwd_r12)
  unsigned spin_limit;
  int vec;      // elements per lane of the tile kernels (TP = chan_tx * vec pixels per tile)
  int wsa_tail; // 1: the dWsa tile partials and their sums are the LAST workgroups of the k_bwd_apply launch (arrival counters = status
                // words 1, 2 of c.sync, 0 between calls); 0: k_bwd_reduce2 roles / own launch
  long long* trace;   // MGACBAM_TRACE builds only (tools/trace_gate.py), else nullptr
};

static inline size_t align16(size_t v) { return (v + 15) & ~size_t(15); }

// One launch covers up to kGroupMax pyramid levels: workgroup ids [start[l], start[l+1]) belong to level l.
// P3+P4+P5 of YOLOv8n are 52+26+13 MB -- alone, P5 cannot fill 256 CUs and every extra launch costs ~2.5 us of
// dependent-kernel boundary, so each stage is ONE launch whose grid is the concatenation of the levels' grids, the level
// whose workgroups run longest FIRST (api: for_each_group).  Interleaving the levels round-robin instead was measured and
// is worse (k_bwd_reduce1 30 -> 43 us): the long, latency-bound P5 workgroups then also sit in the last round = the tail.
constexpr int kGroupMax = 4;
template <typename Args>
struct Group {
  int n;
  int start[kGroupMax + 1];
  Args lv[kGroupMax];
};

}  // namespace mgacbam
