// Execution of a Program: forward and hand-derived reverse runs over the op list (no autograd), op by op on the HIP kernels.
#include <cstdint>
#include <string>
#include "engine_internal.h"

namespace ddi {

// ---------------------------------------------------------------------------------------------------
// execution
// ---------------------------------------------------------------------------------------------------
void fill_conv(ConvGemmParams& p, const Ctx& c) {
  memset(&p, 0, sizeof p);
  p.partial = (float*)c.scratch_partial;
  p.alpha = 1.f;
}

// fp32 programs (the guide network): conv forward / dgrad on guide_f32.hip
void conv_f32_geometry(ConvF32Params& p, const ConvW* w, bool bwd) {
  const PackedConv& sh = bwd ? w->sb : w->sf;
  p.w = bwd ? w->wf_bwd : w->wf_fwd; p.taptab = bwd ? w->tap_bwd : w->tap_fwd;
  p.cin = sh.cin; p.ntaps = sh.ntaps; p.N = sh.N; p.K = sh.K;
  p.groups = w->groups;
  const int gi = w->Cin / w->groups, go = w->Cout / w->groups;
  p.cpg_in = bwd ? go : gi; p.cpg_out = bwd ? gi : go;
}

void run_conv_f32_fwd(const Program& P, const Op& op, const Ctx& c) {
  const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
  const ConvW* w = op.cw;
  ConvF32Params p; memset(&p, 0, sizeof p);
  conv_f32_geometry(p, w, false);
  p.x = act_f32(c, x); p.x_ld = x.ld; p.y = act_f32(c, y); p.y_ld = y.ld;
  p.B = x.B; p.H = x.H; p.W = x.W; p.Ho = y.H; p.Wo = y.W; p.stride = op.stride; p.M = y.rows;
  if (w->bias) { p.flags |= CF_BIAS; p.bias = w->bias; }
  if (op.res >= 0) { p.flags |= CF_RES; p.res = act_f32(c, P.t[op.res]); p.res_ld = P.t[op.res].ld; }
  if (op.relu == 1) p.flags |= CF_RELU;
  if (op.relu == 2) p.flags |= CF_RELU6;
  HIPCHK(launch_conv_f32(p, c.s));
}

void run_conv_f32_bwd(const Program& P, const Op& op, const Ctx& c) {
  const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
  const ConvW* w = op.cw;
  float* gy = grad_f32(c, y);
  // ReLU mask from the fp32 forward output (y > 0), then the residual fan-out, then the dgrad GEMM
  if (op.relu) HIPCHK(launch_mask_f32(gy, y.ld, act_f32(c, y), y.ld, gy, y.ld, y.rows, rup(y.C, 4), op.relu == 2 ? 6.f : 0.f, c.s));
  if (op.res >= 0 && P.t[op.res].grad) {
    const Tn& r = P.t[op.res];
    if (op.res_acc) HIPCHK(launch_add_f32(grad_f32(c, r), r.ld, gy, y.ld, grad_f32(c, r), r.ld, r.rows, rup(r.C, 4), c.s));
    else if (!op.res_alias) HIPCHK(launch_copy_f32(gy, y.ld, grad_f32(c, r), r.ld, r.rows, rup(r.C, 4), c.s));
  }
  if (!x.grad) return;
  ConvF32Params p; memset(&p, 0, sizeof p);
  conv_f32_geometry(p, w, true);
  p.x = gy; p.x_ld = y.ld;
  p.B = y.B; p.H = y.H; p.W = y.W; p.Ho = x.H; p.Wo = x.W; p.M = x.rows; p.stride = 1;
  if (op.stride == 2) { p.shift = 1; p.parity = 1; }
  float* gx = grad_f32(c, x);
  p.y = gx; p.y_ld = x.ld;
  if (op.x_acc) { p.flags |= CF_RES; p.res = gx; p.res_ld = x.ld; }
  HIPCHK(launch_conv_f32(p, c.s));
}

// forward launch parameters of a bf16 convolution / linear op (everything but the run-time fusion flags); `gn_in` >= 0: read the RAW
// input of the GroupNorm op in front of it (CF_GNFOLD) instead of that op's output
static void fwd_conv_params(const Program& P, const Op& op, const Ctx& c, ConvGemmParams& p, int gn_in = -1) {
  const Tn& x = P.t[gn_in >= 0 ? gn_in : op.x_fwd >= 0 ? op.x_fwd : op.x]; const Tn& y = P.t[op.y];
  fill_conv(p, c);
  const ConvW* w = op.cw;
  p.x = act_ptr(c, x); p.x_ld = x.ld; p.w = w->w_fwd; p.taptab = w->tap_fwd;
  p.y = act_raw(c, y); p.y_ld = y.ld;
  p.B = x.B; p.H = x.H; p.W = x.W; p.Ho = y.H; p.Wo = y.W; p.stride = op.stride; p.shift = op.up; p.parity = 0;
  p.cin = w->sf.cin; p.ntaps = w->sf.ntaps; p.M = y.rows; p.N = w->sf.N; p.K = w->sf.K;
  int flags = 0;
  if (op.use_table && w->bias_table) { flags |= CF_BIAS; p.bias = w->bias_table + (size_t)c.step_index * w->Cout; }
  else if (w->bias) { flags |= CF_BIAS; p.bias = w->bias; }
  if (op.res >= 0) { flags |= CF_RES; p.res = act_ptr(c, P.t[op.res]); p.res_ld = P.t[op.res].ld; }
  if (op.relu) flags |= CF_RELU;
  if (op.out_f32) flags |= CF_OUT_F32;
  if (w->geglu) {
    flags |= CF_GEGLU;
    if (op.raw >= 0 && c.stash) { flags |= CF_GEGLU_RAW; p.raw = act_ptr(c, P.t[op.raw]); p.raw_ld = P.t[op.raw].ld; }
  }
  if (op.ln_fold) { flags |= CF_LNFOLD; p.ln_stats = (const float*)(c.act + op.ln_stats_off); p.ln_c1 = w->ln_c1; }
  p.flags = flags;
}

void run_fwd(const Program& P, const Ctx& c, int op_begin, int op_end) {
  if (op_end < 0) op_end = (int)P.ops.size();
  if (c.prof) c.prof->new_run();
  for (int i = op_begin; i < op_end; ++i) {
    const Op& op = P.ops[i];
    const int fam = (op.kind == OP_CONV && !P.f32) ? Profiler::CONV : op.kind == OP_ATTN ? Profiler::ATTN
                    : (op.kind == OP_GN || op.kind == OP_LN) ? Profiler::NORM : Profiler::OTHER;
    if (c.prof) {
      if (op.kind == OP_CONV) c.prof->begin(fam, op.flops, c.s, P.t[op.y].rows, op.cw->sf.N, op.cw->sf.K, 0);
      else if (op.kind == OP_ATTN) c.prof->begin(fam, op.flops, c.s, op.Nq, op.Nk, op.D, 0);
      else c.prof->begin(fam, 0.0, c.s, P.t[op.x].rows, P.t[op.x].C, 0, 0);
    }
    switch (op.kind) {
      case OP_CONV: {
        if (P.f32) { run_conv_f32_fwd(P, op, c); if (c.flops) *c.flops += op.flops; break; }
        // a GroupNorm(+SiLU) folded into this convolution (decided when the GroupNorm op ran: P.gn_folded): read its raw input
        const int gfold = (op.gn_from >= 0 && P.gn_folded[op.gn_from]) ? P.ops[op.gn_from].x : -1;
        const Tn& x = P.t[gfold >= 0 ? gfold : op.x_fwd >= 0 ? op.x_fwd : op.x]; const Tn& y = P.t[op.y];
        const ConvW* w = op.cw;
        ConvGemmParams p;
        fwd_conv_params(P, op, c, p, gfold);
        if (gfold >= 0) { p.flags |= CF_GNFOLD; p.gn_coef = c.gn_coef; p.gn_silu = P.ops[op.gn_from].silu; }
        if (op.rowstat_emit) {
          // LayerNorm row partials for the op that follows: only when the kernel the launcher picks for this shape has the form
          p.rowpart = c.rowpart; p.rowpart_ld = op.rowstat_ld;
          int spans = 0;
          // (an op that is also a GroupNorm-partials producer keeps CF_STATS: one statistics epilogue per launch, and the consumer of the
          // row partials falls back to its own statistics pass through row_spans == 0)
          if (!op.part && c.rowpart && conv_gemm_can_emit_rowstats(p, c.partial_cap, &spans) && spans <= op.rowstat_ld) { p.flags |= CF_ROWSTATS; P.row_spans[i] = spans; }
          else { p.rowpart = nullptr; P.row_spans[i] = 0; }
        }
        if (op.use_table && c.img_bias > 0 && w->bias_table_img) {
          // SDXL text_time conditioning: the time-embedding bias differs per image -> one launch per image of the batch (B x H x W
          // rows each; at 128x128 / 64x64 / 32x32 latents an image still fills the chip), each with its own row of the bias table
          if (x.B != c.img_bias || op.stride != 1 || op.up) throw std::runtime_error("per-image bias: unexpected conv geometry");
          const size_t xrows = (size_t)x.H * x.W, yrows = (size_t)y.H * y.W;
          ConvGemmParams q = p;
          q.B = 1; q.M = (int)yrows;
          bool emit = false;
          if (op.part) {
            q.stats = (float*)(c.act + op.part_off); q.stats_ld = op.part_ld;
            emit = !(yrows & 63) && conv_gemm_can_emit_stats(q, c.partial_cap);
            if (emit) q.flags |= CF_STATS; else q.stats = nullptr;
            P.emitted[i] = emit ? 1 : 0;
          }
          for (int bi = 0; bi < x.B; ++bi) {
            ConvGemmParams r = q;
            r.x = p.x + bi * xrows * x.ld;
            r.y = (char*)p.y + bi * yrows * y.ld * 2;
            r.bias = w->bias_table_img + ((size_t)c.step_index * c.img_bias + bi) * w->Cout;
            if (emit) r.stats = q.stats + (size_t)bi * (yrows / 64) * op.part_ld * 2;
            HIPCHK(launch_conv_gemm(r, c.partial_cap, c.s));
          }
          if (c.flops) *c.flops += op.flops;
          break;
        }
        if (op.part) {
          p.stats = (float*)(c.act + op.part_off); p.stats_ld = op.part_ld;
          const bool emit = conv_gemm_can_emit_stats(p, c.partial_cap);
          if (emit) p.flags |= CF_STATS; else p.stats = nullptr;
          P.emitted[i] = emit ? 1 : 0;
        }
        HIPCHK(launch_conv_gemm(p, c.partial_cap, c.s));
        if (c.flops) *c.flops += op.flops;
      } break;
      case OP_GN: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        GroupNormParams p; memset(&p, 0, sizeof p);
        const float* chan_part = nullptr;
        if (op.part) {
          bool all = true;
          for (int pi : op.producers) all = all && P.emitted[pi];
          if (all) chan_part = (const float*)(c.act + op.part_off);
        }
        if (getenv("DD_GN_REPORT")) {
          static int n_total = 0, n_fused = 0;
          ++n_total; n_fused += chan_part ? 1 : 0;
          if (n_total % 200 == 0) fprintf(stderr, "[gn] %d of %d GroupNorm forwards took their statistics from the producing convolutions\n", n_fused, n_total);
        }
        p.chan_part = chan_part; p.part_ld = op.part_ld;
        p.x = act_ptr(c, x); p.x_ld = x.ld; p.y = act_ptr(c, y); p.y_ld = y.ld;
        p.gamma = op.nw->gamma; p.beta = op.nw->beta; p.stats = (float*)(c.act + op.stats_off); p.scratch = c.gn_scratch;
        p.B = x.B; p.HW = x.H * x.W; p.C = x.C; p.G = op.G; p.eps = op.eps; p.silu = op.silu;
        P.gn_folded[i] = 0;
        if (op.gn_into >= 0 && c.gn_coef && !(c.img_bias > 0 && P.ops[op.gn_into].use_table && P.ops[op.gn_into].cw->bias_table_img)) {
          // the only reader of this GroupNorm is the 3x3 convolution right behind it: when the launcher will run that convolution on
          // the halo-resident kernel, the normalisation is applied to its staged input tile (CF_GNFOLD) and only the per-(image,
          // channel) affine is produced here -- no pass over the tensor
          ConvGemmParams q;
          fwd_conv_params(P, P.ops[op.gn_into], c, q, op.x);
          if (conv_gemm_can_fold_gn(q)) { p.coef = c.gn_coef; p.y = nullptr; P.gn_folded[i] = 1; }
        }
        HIPCHK(launch_groupnorm_fwd(p, c.s));
      } break;
      case OP_LN: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        LayerNormParams p; memset(&p, 0, sizeof p);
        p.x = act_ptr(c, x); p.x_ld = x.ld; p.y = act_ptr(c, y); p.y_ld = y.ld;
        p.gamma = op.nw->gamma; p.beta = op.nw->beta; p.stats = (float*)(c.act + op.stats_off);
        p.M = x.rows; p.C = x.C; p.eps = op.eps;
        if (op.ln_fold) {
          // folded into the linear that follows: statistics only, from the producing GEMM's row partials when it emitted them
          p.y = nullptr;
          if (op.rowstat_from >= 0 && P.row_spans[op.rowstat_from] > 0) { p.rowpart = c.rowpart; p.rowpart_ld = op.rowstat_ld; p.spans = P.row_spans[op.rowstat_from]; }
        }
        HIPCHK(launch_layernorm_fwd(p, c.s));
      } break;
      case OP_ATTN: {
        const Tn& q = P.t[op.q]; const Tn& y = P.t[op.y];
        AttnParams p; memset(&p, 0, sizeof p);
        p.q = act_ptr(c, q); p.ldq = q.ld;
        if (op.cross_slot >= 0) {
          p.k = (*c.cross_kv)[op.cross_slot].first; p.v = (*c.cross_kv)[op.cross_slot].second; p.ldk = p.ldv = q.C;
        } else {
          p.k = act_ptr(c, P.t[op.k]); p.v = act_ptr(c, P.t[op.v]); p.ldk = P.t[op.k].ld; p.ldv = P.t[op.v].ld;
        }
        p.o = act_ptr(c, y); p.ldo = y.ld; p.lse = (float*)(c.act + op.stats_off);
        p.B = q.B; p.H = op.heads; p.Nq = op.Nq; p.Nk = op.Nk; p.D = op.D; p.scale = op.q_prescaled ? 0.6931471805599453f : 1.f / sqrtf((float)op.D); p.q_prescaled = op.q_prescaled;
        p.causal = op.causal;
        p.pv_fp8 = op.pv_fp8;      // BASELINE configs[4]: fp8 P.V for d = 64 heads (SDXL), opt-in per engine (dd_config.unet_attn_fp8)
        if (op.cross_slot < 0 && attention_gemm_supported(p) && c.tap1x1 && attention_gemm_workspace(p.Nq, p.Nk, p.D, 0) <= c.tmp_cap)
          HIPCHK(launch_attention_gemm_fwd(p, c.scratch_tmp, c.tmp_cap, c.tap1x1, (float*)c.scratch_partial, c.partial_cap, c.s));
        else
          HIPCHK(launch_attention_fwd(p, c.s));
        if (c.flops) *c.flops += op.flops;
      } break;
      case OP_CONCAT: {
        if (op.fused) break;
        const Tn& a = P.t[op.x]; const Tn& b = P.t[op.x2]; const Tn& y = P.t[op.y];
        HIPCHK(launch_copy_bf16(act_ptr(c, a), a.ld, act_ptr(c, y), y.ld, y.rows, a.C, c.s));
        HIPCHK(launch_copy_bf16(act_ptr(c, b), b.ld, act_ptr(c, y) + a.C, y.ld, y.rows, b.C, c.s));
      } break;
      case OP_MAXPOOL: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (P.f32) HIPCHK(launch_maxpool3x3s2_f32(act_f32(c, x), act_f32(c, y), x.B, x.H, x.W, x.ld, c.s));
        else HIPCHK(launch_maxpool3x3s2(act_ptr(c, x), act_ptr(c, y), x.B, x.H, x.W, x.C, c.s));
      } break;
      case OP_ACT: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        HIPCHK(launch_act_bf16(act_ptr(c, x), x.ld, act_ptr(c, y), y.ld, x.rows, x.C, op.act_kind, c.s));
      } break;
      case OP_PATCHIFY: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        HIPCHK(launch_patchify(act_f32(c, x), x.ld, act_ptr(c, y), x.B, x.H, op.patch, x.C, c.s));
      } break;
      case OP_VITEMBED: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        HIPCHK(launch_vit_embed(act_ptr(c, x), x.ld, op.nw->gamma, op.nw->beta, act_ptr(c, y), y.ld, x.B, x.H, x.C, c.s));
      } break;
      case OP_SELECT: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        HIPCHK(launch_select_rows(act_ptr(c, x), x.ld, act_ptr(c, y), y.ld, x.B, op.sel_stride, x.C, c.s));
      } break;
      case OP_DUP: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        HIPCHK(launch_copy_bf16(act_ptr(c, x), x.ld, act_ptr(c, y), y.ld, x.rows, rup(x.C, 8), c.s));
        HIPCHK(launch_copy_bf16(act_ptr(c, x), x.ld, act_ptr(c, y) + (size_t)x.rows * y.ld, y.ld, x.rows, rup(x.C, 8), c.s));
      } break;
      case OP_GAP: break;
    }
    if (c.prof) c.prof->end(c.s);
  }
}

// DD_GRAD_CHECK=1: every gradient access of run_bwd is checked against the interval plan_grad_memory packed the slab by
bool ddi::g_grad_check = getenv("DD_GRAD_CHECK") != nullptr && atoi(getenv("DD_GRAD_CHECK")) != 0;
static thread_local int t_bwd_op = INT32_MIN;     // op whose backward is running (INT32_MIN: outside run_bwd)
void ddi::grad_access_check(const Tn& t) {
  if (t_bwd_op == INT32_MIN) return;
  if (t_bwd_op < t.glo || t_bwd_op > t.ghi)
    throw std::runtime_error("gradient plan violated: the backward of op " + std::to_string(t_bwd_op) + " touches a gradient that is only live in [" +
                             std::to_string(t.glo) + ", " + std::to_string(t.ghi) + "]");
}

void run_bwd(const Program& P, const Ctx& c) {
  if (c.prof) c.prof->new_run();
  struct OpScope { ~OpScope() { t_bwd_op = INT32_MIN; } } _scope;
  for (int i = (int)P.ops.size() - 1; i >= 0; --i) {
    const Op& op = P.ops[i];
    t_bwd_op = i;
    const int fam = (op.kind == OP_CONV && !P.f32) ? Profiler::CONV : op.kind == OP_ATTN ? Profiler::ATTN
                    : (op.kind == OP_GN || op.kind == OP_LN) ? Profiler::NORM : Profiler::OTHER;
    if (c.prof) {
      if (op.kind == OP_CONV) c.prof->begin(fam, op.flops, c.s, P.t[op.x].rows << (2 * op.up), op.cw->sb.N, op.cw->sb.K, 1);
      else if (op.kind == OP_ATTN) c.prof->begin(fam, op.flops * (op.cross_slot >= 0 ? 1.5 : 2.5), c.s, op.Nq, op.Nk, op.D, 1);
      else c.prof->begin(fam, 0.0, c.s, P.t[op.x].rows, P.t[op.x].C, 0, 1);
    }
    struct EndGuard { const Ctx& c; ~EndGuard() { if (c.prof) c.prof->end(c.s); } } _guard{c};
    switch (op.kind) {
      case OP_CONV: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad && !(op.res >= 0 && P.t[op.res].grad)) break;
        if (P.f32) { run_conv_f32_bwd(P, op, c); if (x.grad && c.flops) *c.flops += op.flops; break; }
        bf16_t* gy = grad_ptr(c, y);
        const ConvW* w = op.cw;
        if (op.relu) HIPCHK(launch_mask_bf16(gy, y.ld, act_ptr(c, y), y.ld, gy, y.ld, y.rows, y.C, c.s));
        if (op.res >= 0 && P.t[op.res].grad) {
          const Tn& r = P.t[op.res];
          if (op.res_acc) HIPCHK(launch_add_bf16(grad_ptr(c, r), r.ld, gy, y.ld, grad_ptr(c, r), r.ld, r.rows, r.C, c.s));
          else if (!op.res_alias) HIPCHK(launch_copy_bf16(gy, y.ld, grad_ptr(c, r), r.ld, r.rows, r.C, c.s));
        }
        if (!x.grad) break;
        const bf16_t* gin = gy; int gin_ld = y.ld;
        char* tmp = c.scratch_tmp;
        if (w->geglu) {
          bf16_t* draw = (bf16_t*)tmp; tmp += rup_sz((size_t)y.rows * w->Cout * 2, 256);
          HIPCHK(launch_geglu_bwd(act_ptr(c, P.t[op.raw]), P.t[op.raw].ld, gy, y.ld, draw, w->Cout, y.rows, w->Cout / 2, c.s));
          gin = draw; gin_ld = w->Cout;
        }
        ConvGemmParams p; fill_conv(p, c);
        p.x = gin; p.x_ld = gin_ld; p.w = w->w_bwd; p.taptab = w->tap_bwd;
        p.B = y.B; p.H = y.H; p.W = y.W;
        p.cin = w->sb.cin; p.ntaps = w->sb.ntaps; p.N = w->sb.N; p.K = w->sb.K;
        p.stride = 1;
        if (op.stride == 2) { p.shift = 1; p.parity = 1; }
        const int Hl = x.H << op.up, Wl = x.W << op.up;
        p.Ho = Hl; p.Wo = Wl; p.M = x.B * Hl * Wl;
        bf16_t* gx = grad_ptr(c, x);
        if (op.up) {
          bf16_t* hi = (bf16_t*)tmp;
          const int ldh = rup(w->Cin, 8);
          p.y = hi; p.y_ld = ldh; p.flags = 0;
          HIPCHK(launch_conv_gemm(p, c.partial_cap, c.s));
          HIPCHK(launch_sumpool2x2(hi, ldh, gx, x.ld, x.B, x.H, x.W, rup(x.C, 8), op.x_acc ? 1 : 0, c.s));
        } else {
          p.y = gx; p.y_ld = x.ld; p.flags = 0;
          if (op.x_acc) { p.flags |= CF_RES; p.res = gx; p.res_ld = x.ld; }
          HIPCHK(launch_conv_gemm(p, c.partial_cap, c.s));
        }
        if (c.flops) *c.flops += op.flops;
      } break;
      case OP_GN: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad) break;
        GroupNormParams p; memset(&p, 0, sizeof p);
        p.x = act_ptr(c, x); p.x_ld = x.ld;
        p.gamma = op.nw->gamma; p.beta = op.nw->beta; p.stats = (float*)(c.act + op.stats_off); p.scratch = c.gn_scratch;
        p.B = x.B; p.HW = x.H * x.W; p.C = x.C; p.G = op.G; p.eps = op.eps; p.silu = op.silu;
        p.dy = grad_ptr(c, y); p.dy_ld = y.ld; p.dx = grad_ptr(c, x); p.dx_ld = x.ld; p.accumulate = op.x_acc;
        HIPCHK(launch_groupnorm_bwd(p, c.s));
      } break;
      case OP_LN: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad) break;
        LayerNormParams p; memset(&p, 0, sizeof p);
        p.x = act_ptr(c, x); p.x_ld = x.ld; p.gamma = op.nw->gamma; p.beta = op.nw->beta;
        p.stats = (float*)(c.act + op.stats_off); p.M = x.rows; p.C = x.C; p.eps = op.eps;
        p.dy = grad_ptr(c, y); p.dy_ld = y.ld; p.dx = grad_ptr(c, x); p.dx_ld = x.ld; p.accumulate = op.x_acc;
        HIPCHK(launch_layernorm_bwd(p, c.s));
      } break;
      case OP_ATTN: {
        const Tn& q = P.t[op.q]; const Tn& y = P.t[op.y];
        if (!q.grad) break;
        AttnParams p; memset(&p, 0, sizeof p);
        p.q = act_ptr(c, q); p.ldq = q.ld;
        if (op.cross_slot >= 0) {
          p.k = (*c.cross_kv)[op.cross_slot].first; p.v = (*c.cross_kv)[op.cross_slot].second; p.ldk = p.ldv = q.C;
        } else {
          const Tn& k = P.t[op.k]; const Tn& v = P.t[op.v];
          p.k = act_ptr(c, k); p.v = act_ptr(c, v); p.ldk = k.ld; p.ldv = v.ld;
          p.dk = grad_ptr(c, k); p.dv = grad_ptr(c, v); p.lddk = k.ld; p.lddv = v.ld;
        }
        p.o = act_ptr(c, y); p.ldo = y.ld; p.lse = (float*)(c.act + op.stats_off);
        p.delta = p.lse + (size_t)q.B * op.heads * op.Nq;
        p.B = q.B; p.H = op.heads; p.Nq = op.Nq; p.Nk = op.Nk; p.D = op.D; p.scale = op.q_prescaled ? 0.6931471805599453f : 1.f / sqrtf((float)op.D); p.q_prescaled = op.q_prescaled;
        p.d_o = grad_ptr(c, y); p.lddo = y.ld; p.dq = grad_ptr(c, q); p.lddq = q.ld;
        if (op.cross_slot < 0 && attention_gemm_supported(p) && c.tap1x1 && attention_gemm_workspace(p.Nq, p.Nk, p.D, 1) <= c.tmp_cap)
          HIPCHK(launch_attention_gemm_bwd(p, c.scratch_tmp, c.tmp_cap, c.tap1x1, (float*)c.scratch_partial, c.partial_cap, c.s));
        else
          HIPCHK(launch_attention_bwd(p, c.s));
        if (c.flops) *c.flops += op.flops * (op.cross_slot >= 0 ? 1.5 : 2.5);
      } break;
      case OP_CONCAT: {
        if (op.fused) break;
        const Tn& a = P.t[op.x]; const Tn& b = P.t[op.x2]; const Tn& y = P.t[op.y];
        bf16_t* gy = grad_ptr(c, y);
        if (a.grad) {
          if (op.x_acc) HIPCHK(launch_add_bf16(grad_ptr(c, a), a.ld, gy, y.ld, grad_ptr(c, a), a.ld, a.rows, a.C, c.s));
          else HIPCHK(launch_copy_bf16(gy, y.ld, grad_ptr(c, a), a.ld, a.rows, a.C, c.s));
        }
        if (b.grad) {
          if (op.x2_acc) HIPCHK(launch_add_bf16(grad_ptr(c, b), b.ld, gy + a.C, y.ld, grad_ptr(c, b), b.ld, b.rows, b.C, c.s));
          else HIPCHK(launch_copy_bf16(gy + a.C, y.ld, grad_ptr(c, b), b.ld, b.rows, b.C, c.s));
        }
      } break;
      case OP_MAXPOOL: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad) break;
        if (op.x_acc) throw std::runtime_error("maxpool backward accumulate unsupported");
        if (P.f32) HIPCHK(launch_maxpool3x3s2_bwd_f32(act_f32(c, x), grad_f32(c, y), grad_f32(c, x), x.B, x.H, x.W, x.ld, c.s));
        else HIPCHK(launch_maxpool3x3s2_bwd(act_ptr(c, x), grad_ptr(c, y), grad_ptr(c, x), x.B, x.H, x.W, x.C, c.s));
      } break;
      case OP_ACT: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad) break;
        HIPCHK(launch_act_bwd_bf16(act_ptr(c, x), x.ld, grad_ptr(c, y), y.ld, grad_ptr(c, x), x.ld, x.rows, x.C, op.act_kind, op.x_acc ? 1 : 0, c.s));
      } break;
      case OP_PATCHIFY: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad) break;
        if (op.x_acc || !x.gf32) throw std::runtime_error("patchify backward: the image gradient must be an fp32 first write");
        HIPCHK(launch_patchify_bwd(grad_ptr(c, y), grad_f32(c, x), x.ld, x.B, x.H, op.patch, x.C, c.s));
      } break;
      case OP_VITEMBED: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad) break;
        if (op.x_acc) throw std::runtime_error("vit_embed backward accumulate unsupported");
        HIPCHK(launch_vit_embed_bwd(grad_ptr(c, y), y.ld, grad_ptr(c, x), x.ld, x.B, x.H, x.C, c.s));
      } break;
      case OP_DUP: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad) break;
        bf16_t* gy = grad_ptr(c, y); bf16_t* gx = grad_ptr(c, x);
        const bf16_t* hi = gy + (size_t)x.rows * y.ld;
        if (op.x_acc) {
          HIPCHK(launch_add_bf16(gx, x.ld, gy, y.ld, gx, x.ld, x.rows, rup(x.C, 8), c.s));
          HIPCHK(launch_add_bf16(gx, x.ld, hi, y.ld, gx, x.ld, x.rows, rup(x.C, 8), c.s));
        } else {
          HIPCHK(launch_add_bf16(gy, y.ld, hi, y.ld, gx, x.ld, x.rows, rup(x.C, 8), c.s));
        }
      } break;
      case OP_SELECT: {
        const Tn& x = P.t[op.x]; const Tn& y = P.t[op.y];
        if (!x.grad) break;
        HIPCHK(launch_select_rows_bwd(grad_ptr(c, y), y.ld, grad_ptr(c, x), x.ld, x.B, op.sel_stride, x.C, op.x_acc ? 1 : 0, c.s));
      } break;
      case OP_GAP: break;
    }
  }
}


}  // namespace ddi
