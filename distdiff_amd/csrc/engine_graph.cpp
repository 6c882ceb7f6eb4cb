// The static op graphs of the expansion engine: builder helpers, backward / transient / statistics-fusion plans and the model builders
// (SD-1.x / SDXL UNet, AutoencoderKL decoder and encoder, CLIP text tower, the guide networks).  SURVEY.md section 8a rows A2, A4, A7.
#include "engine_internal.h"

namespace ddi {

// ---------------------------------------------------------------------------------------------------
// program builder helpers
// ---------------------------------------------------------------------------------------------------
struct Builder {
  dd_engine* E;
  Program& P;
  int full_batch = 0;      // UNet: 2B when the CFG halves share their prefix (the program starts on B images), else 0
  Builder(dd_engine* e, Program& p) : E(e), P(p) {}

  // y = epilogue(conv(x)) ; returns y (allocated unless `y_into` >= 0)
  int conv(int x, ConvW* w, int stride = 1, int up = 0, int res = -1, int relu = 0, int out_f32 = 0, int use_table = 0,
           int y_into = -1, bool keep_raw = true) {
    const Tn& tx = P.t[x];
    const int Hl = tx.H << up, Wl = tx.W << up;
    const int Ho = (Hl + 2 * w->pad + w->pad_br - w->KH) / stride + 1, Wo = (Wl + 2 * w->pad + w->pad_br - w->KW) / stride + 1;
    const int Cy = w->geglu ? w->Cout / 2 : w->Cout;
    // the output of a GEGLU projection only feeds ff.net.2 (its VJP reads the stashed pre-activations): transient
    int y = y_into >= 0 ? y_into : (w->geglu && !out_f32) ? P.transient(tx.B, Ho, Wo, Cy) : P.tensor(tx.B, Ho, Wo, Cy, true, out_f32 != 0);
    Op op; op.kind = OP_CONV; op.x = x; op.y = y; op.res = res; op.cw = w; op.stride = stride; op.up = up; op.relu = relu;
    op.out_f32 = out_f32; op.use_table = use_table;
    if (w->geglu && P.want_grad && keep_raw) op.raw = P.tensor(tx.B, Ho, Wo, w->Cout, false);
    const size_t M = (size_t)tx.B * Ho * Wo;
    op.flops = 2.0 * M * w->Cout * (w->Cin / w->groups) * w->KH * w->KW;
    if (w->f32) { P.ops.push_back(op); return y; }   // fp32 kernel: no split-K, no scratch
    const int split = conv_gemm_pick_split((int)M, w->sf.N, w->sf.K);
    P.scratch_partial = std::max(P.scratch_partial, (size_t)split * M * w->sf.N * 4);
    if (P.want_grad) {
      const size_t Mb = (size_t)tx.B * Hl * Wl;  // dgrad output rows (high-res when upsample is fused)
      const int sb = conv_gemm_pick_split((int)Mb, w->sb.N, w->sb.K);
      P.scratch_partial = std::max(P.scratch_partial, (size_t)sb * Mb * w->sb.N * 4);
      size_t tmp = 0;
      if (up) tmp += rup_sz(Mb * rup(w->Cin, 8) * 2, 256);
      if (w->geglu) tmp += rup_sz(M * w->Cout * 2, 256);
      P.scratch_tmp = std::max(P.scratch_tmp, tmp);
    }
    P.ops.push_back(op);
    return y;
  }
  int gn(int x, NormW* w, int G, float eps, int silu) {
    const Tn& tx = P.t[x];
    int y = P.transient(tx.B, tx.H, tx.W, tx.C);      // consumed by the convolution that follows, never by a reverse program
    Op op; op.kind = OP_GN; op.x = x; op.y = y; op.nw = w; op.G = G; op.eps = eps; op.silu = silu;
    op.stats_off = P.fp32_block((size_t)tx.B * G * 2);
    P.ops.push_back(op);
    return y;
  }
  // The LayerNorm and the linear just built (the last two ops) become one: the linear's forward packing already holds gamma o W
  // (make_conv(..., ln)), it reads the LayerNorm's input with CF_LNFOLD and the LayerNorm op only produces (mean, rstd) -- taken from
  // the row partials of the GEMM that produced its input when that GEMM is the op right in front of it (CF_ROWSTATS).
  void fold_ln() {
    const int ci = (int)P.ops.size() - 1, li = ci - 1;
    if (li < 0 || P.ops[ci].kind != OP_CONV || P.ops[li].kind != OP_LN || P.ops[ci].x != P.ops[li].y || !P.ops[ci].cw->ln_c1)
      throw std::runtime_error("fold_ln: expected LayerNorm -> linear");
    Op& c = P.ops[ci]; Op& l = P.ops[li];
    c.ln_fold = true; c.x_fwd = l.x; c.ln_stats_off = l.stats_off;
    l.ln_fold = true;
    if (li >= 1 && !getenv("DD_NO_LN_ROWSTATS")) {
      Op& pr = P.ops[li - 1];
      const Tn& tx = P.t[l.x];
      if (pr.kind == OP_CONV && pr.y == l.x && !pr.cw->geglu && !pr.cw->f32 && !pr.out_f32 && !pr.relu && pr.cw->KH == 1 && pr.cw->KW == 1 &&
          pr.stride == 1 && !pr.up && tx.parent == l.x && !pr.part) {
        l.rowstat_from = li - 1;
        pr.rowstat_emit = true;
        pr.rowstat_ld = l.rowstat_ld = (tx.C + 39) / 40;      // narrowest span any kernel form writes (gemm_ws.hip: 48 + 32 columns)
        P.scratch_rowpart = std::max(P.scratch_rowpart, (size_t)tx.rows * pr.rowstat_ld * 8);
      }
    }
  }
  // keep: the output outlives the next operation (a residual stream base, a program output)
  int ln(int x, NormW* w, float eps, bool keep = false) {
    const Tn& tx = P.t[x];
    int y = keep ? P.tensor(tx.B, tx.H, tx.W, tx.C) : P.transient(tx.B, tx.H, tx.W, tx.C);
    Op op; op.kind = OP_LN; op.x = x; op.y = y; op.nw = w; op.eps = eps;
    op.stats_off = P.fp32_block((size_t)tx.rows * 2);
    P.ops.push_back(op);
    return y;
  }
  // self attention: q,k,v are views ; cross attention: k,v come from slot (constant, no grad)
  int attn(int q, int k, int v, int heads, int Nq, int Nk, int cross_slot, int causal = 0, int q_prescaled = 0) {
    const Tn& tq = P.t[q];
    int y = P.tensor(tq.B, tq.H, tq.W, tq.C);
    Op op; op.kind = OP_ATTN; op.q = q; op.k = k; op.v = v; op.y = y; op.heads = heads; op.D = tq.C / heads; op.Nq = Nq; op.Nk = Nk;
    op.cross_slot = cross_slot; op.causal = causal; op.q_prescaled = q_prescaled;
    // wide heads (AutoencoderKL mid block) run through GEMMs on a materialised score matrix: scratch for up to 8 images per launch
    // (single-head layers: grouped GEMMs over the images, attention_gemm.hip), one image otherwise
    if (op.D >= 256 && cross_slot < 0 && !causal)
      P.scratch_tmp = std::max(P.scratch_tmp, attention_gemm_workspace(Nq, Nk, op.D, P.want_grad ? 1 : 0) * (size_t)(heads == 1 ? std::min(tq.B, 8) : 1));
    op.stats_off = P.fp32_block((size_t)tq.B * heads * Nq * 2);  // lse + delta
    op.flops = 4.0 * tq.B * heads * (double)Nq * Nk * op.D;
    P.ops.push_back(op);
    return y;
  }
  // torch.cat([a, b], dim=1) of the UNet's skip connections.  No copy: the output buffer is allocated here and BOTH operands are re-homed
  // into it as column views (row stride = Ca + Cb), so their producers -- convolutions running long before, on the down path, for the
  // skip -- write straight into it and every other consumer reads the view through its row stride; the same holds for the gradients
  // (the split of the backward pass disappears too).  Possible because every kernel takes row strides for inputs and outputs.
  int concat(int a, int b) {
    const int y = P.tensor(P.t[a].B, P.t[a].H, P.t[a].W, P.t[a].C + P.t[b].C);
    Op op; op.kind = OP_CONCAT; op.x = a; op.x2 = b; op.y = y;
    auto movable = [&](int id) {
      const Tn& t = P.t[id];
      if (t.parent != id || t.f32 || P.f32 || t.transient || (t.C & 7)) return false;
      for (size_t k = 0; k < P.t.size(); ++k)
        if ((int)k != id && P.t[k].parent == id) return false;     // it has views of its own (their offsets would go stale)
      return true;
    };
    if (a != b && movable(a) && movable(b) && !getenv("DD_NO_CONCAT_FUSION")) {
      const Tn ty = P.t[y];
      int c0 = 0;
      for (int id : {a, b}) {
        Tn& t = P.t[id];
        t.off = ty.off + (size_t)c0 * 2; t.goff = ty.goff + (size_t)c0 * 2; t.ld = ty.ld; t.parent = y;
        c0 += t.C;
      }
      op.fused = true;
    }
    P.ops.push_back(op);
    return y;
  }
  int act(int x, int kind) {   // text encoder MLP (forward only), ViT guide MLP (with backward)
    const Tn& tx = P.t[x];
    int y = P.transient(tx.B, tx.H, tx.W, tx.C);
    Op op; op.kind = OP_ACT; op.x = x; op.y = y; op.act_kind = kind;
    P.ops.push_back(op);
    return y;
  }
  // ViT guide: image [B,S,S,3] fp32 -> patch rows [B*(S/p)^2, 3*p*p]
  int patchify(int x, int p) {
    const Tn& tx = P.t[x];
    const int g = tx.H / p;
    int y = P.tensor(tx.B, g * g, 1, tx.C * p * p);
    Op op; op.kind = OP_PATCHIFY; op.x = x; op.y = y; op.patch = p;
    P.ops.push_back(op);
    return y;
  }
  // class token + positional embedding (nw->gamma = class_embedding [W], nw->beta = positional_embedding [(np+1)*W])
  int vit_embed(int x, NormW* emb) {
    const Tn& tx = P.t[x];
    int y = P.tensor(tx.B, tx.H + 1, 1, tx.C);
    Op op; op.kind = OP_VITEMBED; op.x = x; op.y = y; op.nw = emb;
    P.ops.push_back(op);
    return y;
  }
  // cat[x, x] along the batch: the point where the two classifier-free-guidance halves stop being identical (build_unet)
  int dup(int x) {
    const Tn& tx = P.t[x];
    int y = P.tensor(2 * tx.B, tx.H, tx.W, tx.C);
    Op op; op.kind = OP_DUP; op.x = x; op.y = y;
    P.ops.push_back(op);
    return y;
  }
  int select_first(int x) {     // the class token of every image
    const Tn& tx = P.t[x];
    int y = P.tensor(tx.B, 1, 1, tx.C);
    Op op; op.kind = OP_SELECT; op.x = x; op.y = y; op.sel_stride = tx.H * tx.W;
    P.ops.push_back(op);
    return y;
  }
  int maxpool(int x) {
    const Tn& tx = P.t[x];
    int y = P.tensor(tx.B, tx.H / 2, tx.W / 2, tx.C);
    Op op; op.kind = OP_MAXPOOL; op.x = x; op.y = y;
    P.ops.push_back(op);
    return y;
  }
};

void plan_grad_memory(Program& P);
// decide, for every op input, whether its gradient contribution is the first write (assign) or an accumulation
void plan_backward(Program& P) {
  std::vector<int> state(P.t.size(), 0);       // per parent: 0 none, 1 partial (views), 2 full
  std::vector<char> vwritten(P.t.size(), 0);   // per view
  auto mark = [&](int id) -> bool {            // returns accumulate?
    const Tn& t = P.t[id];
    if (!t.grad) return false;
    const bool is_view = (t.parent != id);
    if (!is_view) {
      if (state[id] == 1) throw std::runtime_error("backward plan: whole-tensor gradient write after a partial view write");
      const bool acc = state[id] == 2;
      state[id] = 2;
      return acc;
    }
    if (state[t.parent] == 2 || vwritten[id]) return true;
    vwritten[id] = 1;
    state[t.parent] = 1;
    return false;
  };
  for (int i = (int)P.ops.size() - 1; i >= 0; --i) {
    Op& op = P.ops[i];
    switch (op.kind) {
      case OP_CONV:
        if (op.res >= 0) {
          op.res_acc = mark(op.res);
          // y = conv(x) + res: g(res) (+)= g(y).  When that is the FIRST contribution to g(res) and both are plain tensors of the same
          // shape, g(res) simply takes over g(y)'s buffer (g(y) is dead once this op's backward has run; later contributions
          // accumulate into it) instead of being copied.
          Tn& r = P.t[op.res]; const Tn& y = P.t[op.y];
          if (!op.res_acc && r.grad && r.parent == op.res && y.parent == op.y && r.ld == y.ld && r.C == y.C && r.rows == y.rows &&
              !getenv("DD_NO_GRAD_ALIAS")) {
            bool has_views = false;
            for (size_t k = 0; k < P.t.size(); ++k)
              if ((int)k != op.res && P.t[k].parent == op.res) has_views = true;
            if (!has_views) { r.goff = y.goff; op.res_alias = true; }
          }
        }
        op.x_acc = mark(op.x);
        break;
      case OP_GN: case OP_LN: case OP_MAXPOOL: case OP_GAP:
        op.x_acc = mark(op.x);
        break;
      case OP_ACT: case OP_PATCHIFY: case OP_VITEMBED: case OP_SELECT: case OP_DUP:
        op.x_acc = mark(op.x);
        break;
      case OP_ATTN:
        mark(op.q);
        if (op.cross_slot < 0) { mark(op.k); mark(op.v); }
        break;
      case OP_CONCAT:
        if (op.fused) break;       // the operands' gradients are column views of the output's gradient: nothing to move
        op.x_acc = mark(op.x);
        op.x2_acc = mark(op.x2);
        break;
    }
  }
  if (getenv("DD_PLAN_REPORT")) {
    int nres = 0, nacc = 0, nalias = 0, ncopy = 0;
    for (const Op& op : P.ops)
      if (op.kind == OP_CONV && op.res >= 0 && P.t[op.res].grad) { ++nres; if (op.res_acc) ++nacc; else if (op.res_alias) ++nalias; else ++ncopy; }
    fprintf(stderr, "[plan] residual gradients: %d convolutions with a residual, %d accumulate (add kernel), %d alias the output gradient, %d copy\n", nres, nacc, nalias, ncopy);
  }
  plan_grad_memory(P);
}

// Gradient buffers by liveness.  Builder::tensor gives every tensor with a gradient its own range of the gradient slab; but a reverse
// program touches a gradient only between its first contribution (the backward of the LAST forward consumer) and the backward of its
// producer, so -- skip connections aside -- a handful of them are alive at any time.  After plan_backward has fixed the buffer classes
// (a base tensor, its column views, residual gradients aliased onto an output gradient), every class gets the interval of op indices
// in which the reverse run reads or writes it and the classes are packed first-fit: two classes share bytes only when their intervals are
// disjoint.  Gradients that are read or written OUTSIDE run_bwd (program inputs / outputs: the sampler drivers seed and collect them)
// and tensors with padding columns (which some kernels never write and others read: they must stay zero) keep a range of their own.
// 32 images of the bench workload: the gradient slab shrinks from 65.9 GB to 13.2 GB (UNet 38.7 -> 1.9, VAE decoder 64.4 -> 13.0, guide 1.47 ->
// 0.22 GB); results are bit-identical (tests/test_engine_gpu.py::test_gradient_slab_by_liveness_is_bitwise_identical).
void plan_grad_memory(Program& P) {
  if (!P.want_grad || getenv("DD_NO_GRAD_REUSE")) return;
  const int nt = (int)P.t.size(), nops = (int)P.ops.size();
  std::vector<int> cls(nt);
  for (int i = 0; i < nt; ++i) cls[i] = P.t[i].parent;                 // views -> their base (one level by construction)
  std::function<int(int)> find = [&](int i) { while (cls[i] != i) { cls[i] = cls[cls[i]]; i = cls[i]; } return i; };
  for (int i = 0; i < nt; ++i) if (P.t[P.t[i].parent].parent != P.t[i].parent) throw std::runtime_error("grad plan: nested views");
  for (const Op& op : P.ops)
    if (op.kind == OP_CONV && op.res_alias) cls[find(op.res)] = find(op.y);
  const int LO = -1, HI = nops;
  std::vector<int> lo(nt, HI + 1), hi(nt, LO - 1);
  std::vector<size_t> size(nt, 0);
  std::vector<char> has_prod(nt, 0), has_cons(nt, 0);
  auto touch = [&](int id, int at) {
    if (id < 0 || !P.t[id].grad) return;
    const int c = find(id);
    lo[c] = std::min(lo[c], at); hi[c] = std::max(hi[c], at);
  };
  for (int i = 0; i < nops; ++i) {
    const Op& op = P.ops[i];
    if (op.y >= 0) { has_prod[op.y] = 1; touch(op.y, i); }
    for (int id : {op.x, op.res, op.q, op.k, op.v, op.x2})
      if (id >= 0) { touch(id, i); if (op.kind != OP_GAP) has_cons[id] = 1; }
  }
  for (int i = 0; i < nt; ++i) {
    const Tn& t = P.t[i];
    if (!t.grad) continue;
    const int c = find(i);
    if (t.parent == i) {
      const size_t bytes = rup_sz((size_t)t.rows * t.ld * (t.gf32 ? 4 : 2), 256);
      if (size[c] && size[c] != bytes) throw std::runtime_error("grad plan: aliased gradients of different sizes");
      size[c] = bytes;
      if (find(i) != i && t.goff != P.t[c].goff) throw std::runtime_error("grad plan: aliased gradients at different offsets");
    }
    const bool padded = t.C != t.ld && t.parent == i;
    bool produced = has_prod[i], consumed = has_cons[i];
    if (t.parent != i) { produced = produced || has_prod[t.parent]; }
    if (!produced) touch(i, LO);               // a program input: its gradient is collected after the run
    if (!consumed && t.parent == i) {          // a program output (or the input of the pooling the driver differentiates itself): seeded before the run
      bool view_consumed = false;
      for (int k = 0; k < nt; ++k) if (k != i && P.t[k].parent == i && has_cons[k]) view_consumed = true;
      if (!view_consumed) touch(i, HI);
    }
    if (padded) { touch(i, LO); touch(i, HI); }
  }
  // relative offsets of every tensor inside its class, taken before anything moves
  std::vector<size_t> rel(nt, 0);
  for (int i = 0; i < nt; ++i) if (P.t[i].grad) rel[i] = P.t[i].goff - P.t[find(i)].goff;
  std::vector<int> order;
  for (int i = 0; i < nt; ++i) if (P.t[i].grad && find(i) == i && size[i]) order.push_back(i);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return hi[a] != hi[b] ? hi[a] > hi[b] : size[a] > size[b]; });   // reverse-run order
  struct Blk { size_t off, size; int lo, hi; };
  std::vector<Blk> placed;
  std::vector<size_t> newoff(nt, 0);
  size_t total = 0;
  for (int c : order) {
    if (lo[c] > hi[c]) { lo[c] = LO; hi[c] = HI; }     // never touched by an op: keep it out of everybody's way
    std::vector<std::pair<size_t, size_t>> busy;       // ranges of the classes alive at the same time
    for (const Blk& b : placed) if (!(b.hi < lo[c] || b.lo > hi[c])) busy.push_back({b.off, b.off + b.size});
    std::sort(busy.begin(), busy.end());
    size_t at = 0;
    for (auto& r : busy) { if (at + size[c] <= r.first) break; at = std::max(at, r.second); }
    newoff[c] = at;
    placed.push_back({at, size[c], lo[c], hi[c]});
    total = std::max(total, at + size[c]);
  }
  for (int i = 0; i < nt; ++i) if (P.t[i].grad) { P.t[i].goff = newoff[find(i)] + rel[i]; P.t[i].glo = lo[find(i)]; P.t[i].ghi = hi[find(i)]; }
  // consistency of the packing itself: two classes that are alive at the same time never share a byte
  for (size_t a = 0; a < placed.size(); ++a)
    for (size_t b = a + 1; b < placed.size(); ++b)
      if (!(placed[a].hi < placed[b].lo || placed[a].lo > placed[b].hi) && placed[a].off < placed[b].off + placed[b].size && placed[b].off < placed[a].off + placed[a].size)
        throw std::runtime_error("grad plan: overlapping live ranges");
  if (getenv("DD_PLAN_REPORT")) fprintf(stderr, "[plan] gradient slab: %.3f GB one range per tensor -> %.3f GB by liveness (%zu classes)\n", P.grad_bytes / 1e9, total / 1e9, order.size());
  P.grad_bytes = total;
}

// Transient tensors (Tn::transient) share two ping-pong buffers: legal only if every reader of one runs before the next tensor that
// takes the same buffer is produced, and no reverse program reads it.  Checked once per program at build time.
void check_transients(const Program& P) {
  auto reads = [&](const Op& o, int id) {
    auto is = [&](int t) { return t >= 0 && (t == id || P.t[t].parent == id); };
    return is(o.x) || is(o.res) || is(o.q) || is(o.k) || is(o.v) || is(o.x2);
  };
  for (size_t id = 0; id < P.t.size(); ++id) {
    if (!P.t[id].transient) continue;
    int prod = -1, next_same_slot = (int)P.ops.size();
    for (size_t oi = 0; oi < P.ops.size(); ++oi)
      if (P.ops[oi].y == (int)id) prod = (int)oi;
    if (prod < 0) throw std::runtime_error("transient tensor without a producer");
    for (size_t oi = prod + 1; oi < P.ops.size(); ++oi) {
      const int y = P.ops[oi].y;
      if (y >= 0 && P.t[y].transient && P.t[y].tr_slot == P.t[id].tr_slot) { next_same_slot = (int)oi; break; }
    }
    for (size_t oi = 0; oi < P.ops.size(); ++oi) {
      const Op& o = P.ops[oi];
      if (!reads(o, (int)id)) continue;
      if ((int)oi <= prod || (int)oi > next_same_slot) throw std::runtime_error("transient tensor is read after its buffer was reused");
      // activations a reverse program reads must be stashed, not transient
      const bool bwd_reads = (o.kind == OP_GN || o.kind == OP_LN || o.kind == OP_MAXPOOL || o.kind == OP_ACT || o.kind == OP_ATTN) ||
                             (o.kind == OP_CONV && o.res == (int)id && false);
      if (P.want_grad && bwd_reads) throw std::runtime_error("transient tensor is an input a reverse program reads");
    }
    if (P.want_grad)
      for (const Op& o : P.ops)
        if (o.y == (int)id && ((o.kind == OP_CONV && o.relu) || o.kind == OP_ATTN)) throw std::runtime_error("transient tensor is an output a reverse program reads");
  }
}

// GroupNorm statistics without a pass over the tensor: when every producer of a GroupNorm's input (possibly several convolutions
// writing column ranges of one concat buffer) is an implicit-GEMM convolution, those convolutions emit per-(64-row block, channel)
// partial (mean, M2) from their epilogue registers and the GroupNorm merges them.  Whether a convolution can do that depends on the
// kernel the launcher picks for its shape (conv_gemm_can_emit_stats), so the final decision is taken per run; this pass only sets up
// the buffers and the producer lists.
void plan_gn_stats(Program& P) {
  check_transients(P);
  P.emitted.assign(P.ops.size(), 0);
  P.row_spans.assign(P.ops.size(), 0);
  P.gn_folded.assign(P.ops.size(), 0);
  // GroupNorm -> 3x3 convolution pairs (ResnetBlock2D: norm1 -> conv1, norm2 -> conv2; conv_norm_out -> conv_out): candidates for
  // CF_GNFOLD.  The GroupNorm's output is a transient read by that convolution only (check_transients), so nothing else needs it.
  // Built, bit-identical to the separate kernels (tests/test_kernels_gpu.py::test_groupnorm_applied_by_the_halo_convolution) and
  // OFF by default: the apply runs once per tile, 64-channel chunk and n-tile while the matrix pipe waits (norm family 206 -> 155 ms,
  // conv family 1615 -> 1673 ms per 32-image step, same device, tools/ab_gn.sh): DD_GN_APPLY_FUSION=1 switches it on.
  // DD_GN_NARROW_FUSION=1: only conv_norm_out -> conv_out (N <= 4, conv_halo_kernel<1, 2>).  Measured on the bench step, same box:
  // conv_out on the narrow halo form 2098 -> 2089 ms per step; with its GroupNorm folded in as well 2095 ms (norm -3.5 ms, conv +8 ms:
  // that form is bound by its exposed halo refill and the apply in LDS lengthens exactly that) -- off as well.
  const bool fold_all = getenv("DD_GN_APPLY_FUSION") && atoi(getenv("DD_GN_APPLY_FUSION"));
  const bool fold_narrow = getenv("DD_GN_NARROW_FUSION") && atoi(getenv("DD_GN_NARROW_FUSION"));
  if (!P.f32 && (fold_all || fold_narrow))
    for (size_t gi = 0; gi + 1 < P.ops.size(); ++gi) {
      Op& g = P.ops[gi]; Op& cv = P.ops[gi + 1];
      if (g.kind != OP_GN || cv.kind != OP_CONV || cv.x != g.y || cv.x_fwd >= 0 || !P.t[g.y].transient) continue;
      if (cv.cw->KH != 3 || cv.cw->KW != 3 || cv.stride != 1 || cv.up || cv.cw->f32 || cv.cw->geglu) continue;
      if (!fold_all && cv.cw->Cout > 4) continue;
      g.gn_into = (int)gi + 1; cv.gn_from = (int)gi;
    }
  if (P.f32 || getenv("DD_NO_GN_FUSION")) return;
  std::unordered_map<int, size_t> root_part;
  for (size_t gi = 0; gi < P.ops.size(); ++gi) {
    if (P.ops[gi].kind != OP_GN) continue;
    const Tn x = P.t[P.ops[gi].x];
    const int root = x.parent;
    const Tn rt = P.t[root];
    if (x.f32 || rt.f32 || (x.rows & 63) || ((x.H * x.W) & 63)) continue;
    const int coff = (int)((x.off - rt.off) / 2);
    std::vector<int> prod;
    int covered = 0;
    bool ok = true;
    for (size_t oi = 0; oi < gi && ok; ++oi) {
      const Op& o = P.ops[oi];
      if (o.y < 0 || (o.kind == OP_CONCAT && o.fused)) continue;
      const Tn& ty = P.t[o.y];
      if (ty.parent != root) continue;
      const int yc = (int)((ty.off - rt.off) / 2);
      if (yc + ty.C <= coff || yc >= coff + x.C) continue;
      if (o.kind != OP_CONV || o.cw->geglu || o.cw->f32 || o.out_f32 || yc < coff || yc + ty.C > coff + x.C) { ok = false; break; }
      prod.push_back((int)oi);
      covered += ty.C;
    }
    if (!ok || covered != x.C) continue;
    if (!root_part.count(root)) root_part[root] = P.fp32_block((size_t)(rt.rows / 64) * rt.C * 2);
    Op& g = P.ops[gi];
    g.part = true; g.part_off = root_part[root] + (size_t)coff * 8; g.part_ld = rt.C; g.producers = prod;
    for (int oi : prod) {
      Op& o = P.ops[oi];
      const int yc = (int)((P.t[o.y].off - rt.off) / 2);
      o.part = true; o.part_off = root_part[root] + (size_t)yc * 8; o.part_ld = rt.C;
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// model builders (topology: SURVEY.md section 8a rows A2, A4, A7)
// ---------------------------------------------------------------------------------------------------
int build_resnet(Builder& b, const std::string& model, const std::string& p, int x, int G, float eps, bool temb) {
  dd_engine* E = b.E;
  NormW* n1 = make_norm(E, model, p + ".norm1");
  ConvW* c1 = make_conv(E, model, p + ".conv1", 1);
  NormW* n2 = make_norm(E, model, p + ".norm2");
  ConvW* c2 = make_conv(E, model, p + ".conv2", 1);
  if (temb) {
    // per-timestep effective bias table is filled by dd_set_schedule from time_emb_proj
    const HostTensor& tw = E->get(model, p + ".time_emb_proj.weight");
    const HostTensor& tb = E->get(model, p + ".time_emb_proj.bias");
    c1->temb_w = (float*)E->wupload(tw.data.data(), tw.numel() * 4);
    c1->temb_b = (float*)E->wupload(tb.data.data(), tb.numel() * 4);
    E->temb_convs.push_back(c1);
  }
  int h = b.gn(x, n1, G, eps, 1);
  h = b.conv(h, c1, 1, 0, -1, 0, 0, temb ? 1 : 0);
  h = b.gn(h, n2, G, eps, 1);
  int sc = x;
  if (E->has(model, p + ".conv_shortcut.weight")) sc = b.conv(x, make_conv(E, model, p + ".conv_shortcut", 0));
  return b.conv(h, c2, 1, 0, sc);
}

// diffusers Transformer2DModel: GroupNorm -> proj_in -> `depth` BasicTransformerBlocks -> proj_out + residual.  proj_in / proj_out are
// 1x1 convolutions (SD-1.x) or nn.Linear (SDXL, use_linear_projection): both are [C, C] GEMMs on NHWC rows here.
int build_transformer(Builder& b, const std::string& p, int x, int heads, int G, int depth) {
  dd_engine* E = b.E;
  Program& P = b.P;
  const std::string m = "unet";
  const int C = P.t[x].C, HW = P.t[x].H * P.t[x].W;
  int h = b.gn(x, make_norm(E, m, p + ".norm"), G, 1e-6f, 0);
  h = b.conv(h, make_conv(E, m, p + ".proj_in", 0));
  char tb[32];
  for (int d = 0; d < depth; ++d) {
    snprintf(tb, sizeof tb, ".transformer_blocks.%d", d);
    const std::string t = p + tb;
    // self attention (fused QKV projection, no bias)
    // the three LayerNorms of a block are folded into the linear each of them feeds (Builder::fold_ln; DD_NO_LN_FOLD=1 builds the plain graph)
    const bool fold = ln_fold_enabled();
    // the softmax scale 1/sqrt(d) * log2(e) lives in the to_q rows: the attention forward exponentiates the MFMA results as they are
    const int qs = attn_prescale() ? 1 : 0;
    const float qscale = qs ? 1.4426950408889634f / sqrtf((float)(C / heads)) : 1.f;
    int n = b.ln(h, make_norm(E, m, t + ".norm1"), 1e-5f);
    int qkv = b.conv(n, make_conv_cat(E, m, {t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"}, false, fold ? t + ".norm1" : "",
                                      qs ? C : 0, qscale));
    if (fold) b.fold_ln();
    int q = P.view(qkv, 0, C), k = P.view(qkv, C, C), v = P.view(qkv, 2 * C, C);
    const int fp8 = E->cfg.unet_attn_fp8 && C / heads == 64;      // BASELINE configs[4]: fp8 P.V for the d = 64 heads (SDXL)
    int a = b.attn(q, k, v, heads, HW, HW, -1, 0, qs);
    P.ops.back().pv_fp8 = fp8;
    h = b.conv(a, make_conv(E, m, t + ".attn1.to_out.0", 0), 1, 0, h);
    // cross attention: K,V of the text embeddings are computed once per prompt (dd_set_prompt)
    n = b.ln(h, make_norm(E, m, t + ".norm2"), 1e-5f);
    int q2 = b.conv(n, make_conv(E, m, t + ".attn2.to_q", 0, false, false, fold ? t + ".norm2" : "", qs ? C : 0, qscale));
    if (fold) b.fold_ln();
    if (P.t[q2].B == E->cfg.max_batch && b.full_batch == 2 * E->cfg.max_batch) {
      // Up to here the unconditional and the conditional half of the CFG batch were the SAME computation (same latents, same timestep;
      // only the text differs): it ran once on B images.  The first cross-attention is where they part: cat[q, q], cat[h, h].
      q2 = b.dup(q2);
      h = b.dup(h);
      x = b.dup(x);
    }
    dd_engine::CrossSlot slot;
    slot.wk = make_conv(E, m, t + ".attn2.to_k", 0, false, false);
    slot.wv = make_conv(E, m, t + ".attn2.to_v", 0, false, false);
    slot.C = C;
    E->cross_slots.push_back(slot);
    a = b.attn(q2, -1, -1, heads, HW, E->cfg.text_len, (int)E->cross_slots.size() - 1, 0, qs);
    P.ops.back().pv_fp8 = fp8;
    h = b.conv(a, make_conv(E, m, t + ".attn2.to_out.0", 0), 1, 0, h);
    // GEGLU feed-forward
    n = b.ln(h, make_norm(E, m, t + ".norm3"), 1e-5f);
    int ff = b.conv(n, make_conv(E, m, t + ".ff.net.0.proj", 0, true, true, fold ? t + ".norm3" : ""));
    if (fold) b.fold_ln();
    h = b.conv(ff, make_conv(E, m, t + ".ff.net.2", 0), 1, 0, h);
  }
  return b.conv(h, make_conv(E, m, p + ".proj_out", 0), 1, 0, x);
}

void build_unet(dd_engine* E) {
  const dd_config& c = E->cfg;
  Program& P = E->unet;
  P.want_grad = c.enable_grad != 0;
  Builder b(E, P);
  const int B2 = 2 * c.max_batch, L = c.latent_size, G = c.unet_groups, nl = c.unet_levels;
  const float eps = c.unet_eps;
  const std::string m = "unet";
  auto heads_of = [&](int lev) { return c.unet_level_heads[lev] > 0 ? c.unet_level_heads[lev] : c.unet_num_heads; };
  auto depth_of = [&](int lev) { return c.unet_transformer_depth[lev] > 0 ? c.unet_transformer_depth[lev] : 1; };
  // classifier-free guidance runs the UNet on cat[z, z] (generate_data.py:110-112): until the first cross-attention the two halves are
  // bit-for-bit the same computation, so the program starts on B images and duplicates at that point (build_transformer).  Not with
  // SDXL's text_time conditioning, whose time embedding already differs per half.
  const bool share = c.unet_add_time_dim == 0 && !getenv("DD_NO_CFG_SHARE");
  b.full_batch = share ? B2 : 0;
  E->unet_in = P.tensor(share ? c.max_batch : B2, L, L, c.unet_in_channels);
  int h = b.conv(E->unet_in, make_conv(E, m, "conv_in", 1));
  std::vector<int> skips{h};
  char buf[128];
  for (int i = 0; i < nl; ++i) {
    for (int j = 0; j < c.unet_layers_per_block; ++j) {
      snprintf(buf, sizeof buf, "down_blocks.%d.resnets.%d", i, j);
      h = build_resnet(b, m, buf, h, G, eps, true);
      if (c.unet_down_attn[i]) { snprintf(buf, sizeof buf, "down_blocks.%d.attentions.%d", i, j); h = build_transformer(b, buf, h, heads_of(i), G, depth_of(i)); }
      skips.push_back(h);
    }
    if (i < nl - 1) {
      snprintf(buf, sizeof buf, "down_blocks.%d.downsamplers.0.conv", i);
      h = b.conv(h, make_conv(E, m, buf, 1), 2);
      skips.push_back(h);
    }
  }
  h = build_resnet(b, m, "mid_block.resnets.0", h, G, eps, true);
  h = build_transformer(b, "mid_block.attentions.0", h, heads_of(nl - 1), G, depth_of(nl - 1));
  h = build_resnet(b, m, "mid_block.resnets.1", h, G, eps, true);
  for (int i = 0; i < nl; ++i) {
    for (int j = 0; j < c.unet_layers_per_block + 1; ++j) {
      int sk = skips.back(); skips.pop_back();
      if (P.t[sk].B != P.t[h].B) sk = b.dup(sk);      // a skip from the shared CFG prefix
      h = b.concat(h, sk);
      snprintf(buf, sizeof buf, "up_blocks.%d.resnets.%d", i, j);
      h = build_resnet(b, m, buf, h, G, eps, true);
      if (c.unet_up_attn[i]) { snprintf(buf, sizeof buf, "up_blocks.%d.attentions.%d", i, j); h = build_transformer(b, buf, h, heads_of(nl - 1 - i), G, depth_of(nl - 1 - i)); }
    }
    if (i < nl - 1) {
      snprintf(buf, sizeof buf, "up_blocks.%d.upsamplers.0.conv", i);
      h = b.conv(h, make_conv(E, m, buf, 1), 1, 1);
    }
  }
  if (P.t[h].B != B2) throw std::runtime_error("UNet without any cross-attention: the CFG halves never part");
  h = b.gn(h, make_norm(E, m, "conv_norm_out"), G, eps, 1);
  E->unet_out = b.conv(h, make_conv(E, m, "conv_out", 1), 1, 0, -1, 0, 1);
  if (P.want_grad) plan_backward(P);
  plan_gn_stats(P);
}

void build_vae(dd_engine* E) {
  const dd_config& c = E->cfg;
  Program& P = E->vae;
  P.want_grad = c.enable_grad != 0;
  Builder b(E, P);
  const int B = c.max_batch, L = c.latent_size, G = c.vae_groups, nl = c.vae_levels;
  const float eps = c.vae_eps;
  const std::string m = "vae";
  E->vae_in = P.tensor(B, L, L, c.vae_latent_channels);
  int h = b.conv(E->vae_in, make_conv(E, m, "post_quant_conv", 0));
  h = b.conv(h, make_conv(E, m, "decoder.conv_in", 1));
  h = build_resnet(b, m, "decoder.mid_block.resnets.0", h, G, eps, false);
  {
    const std::string a = "decoder.mid_block.attentions.0";
    const int C = P.t[h].C, HW = P.t[h].H * P.t[h].W;
    int n = b.gn(h, make_norm(E, m, a + ".group_norm"), G, eps, 0);
    int qkv = b.conv(n, make_conv_cat(E, m, {a + ".to_q", a + ".to_k", a + ".to_v"}, true));
    int q = P.view(qkv, 0, C), k = P.view(qkv, C, C), v = P.view(qkv, 2 * C, C);
    int o = b.attn(q, k, v, 1, HW, HW, -1);
    h = b.conv(o, make_conv(E, m, a + ".to_out.0", 0), 1, 0, h);
  }
  h = build_resnet(b, m, "decoder.mid_block.resnets.1", h, G, eps, false);
  char buf[128];
  for (int i = 0; i < nl; ++i) {
    for (int j = 0; j < c.vae_layers_per_block + 1; ++j) {
      snprintf(buf, sizeof buf, "decoder.up_blocks.%d.resnets.%d", i, j);
      h = build_resnet(b, m, buf, h, G, eps, false);
    }
    if (i < nl - 1) {
      snprintf(buf, sizeof buf, "decoder.up_blocks.%d.upsamplers.0.conv", i);
      h = b.conv(h, make_conv(E, m, buf, 1), 1, 1);
    }
  }
  h = b.gn(h, make_norm(E, m, "decoder.conv_norm_out"), G, eps, 1);
  // the image leaves the decoder in fp32 (no bf16 rounding in front of the guide's ReLU masks or the uint8 quantisation)
  E->vae_out = b.conv(h, make_conv(E, m, "decoder.conv_out", 1), 1, 0, -1, 0, /*out_f32=*/1);
  if (P.want_grad) plan_backward(P);
  plan_gn_stats(P);
}

// f-2: AutoencoderKL.encode (dataloader.py:808) -- Encoder: conv_in, DownEncoderBlock2D x levels (resnets +
// stride-2 conv with F.pad (0,1,0,1)), mid Res-Attn-Res, GN+SiLU+conv_out, quant_conv -> moments (mean | logvar) fp32.
// Forward only; built when the state dict carries encoder.* keys.
void build_vae_encoder(dd_engine* E) {
  const dd_config& c = E->cfg;
  Program& P = E->venc;
  P.want_grad = false;
  Builder b(E, P);
  const int B = c.max_batch, S = c.latent_size << (c.vae_levels - 1), G = c.vae_groups, nl = c.vae_levels;
  const float eps = c.vae_eps;
  const std::string m = "vae";
  E->venc_in = P.tensor(B, S, S, c.vae_out_channels);
  int h = b.conv(E->venc_in, make_conv(E, m, "encoder.conv_in", 1));
  char buf[128];
  for (int i = 0; i < nl; ++i) {
    for (int j = 0; j < c.vae_layers_per_block; ++j) {
      snprintf(buf, sizeof buf, "encoder.down_blocks.%d.resnets.%d", i, j);
      h = build_resnet(b, m, buf, h, G, eps, false);
    }
    if (i < nl - 1) {
      snprintf(buf, sizeof buf, "encoder.down_blocks.%d.downsamplers.0.conv", i);
      ConvW* w = make_conv(E, m, buf, 0);
      w->pad_br = 1;
      h = b.conv(h, w, 2);
    }
  }
  h = build_resnet(b, m, "encoder.mid_block.resnets.0", h, G, eps, false);
  {
    const std::string a = "encoder.mid_block.attentions.0";
    const int C = P.t[h].C, HW = P.t[h].H * P.t[h].W;
    int n = b.gn(h, make_norm(E, m, a + ".group_norm"), G, eps, 0);
    int qkv = b.conv(n, make_conv_cat(E, m, {a + ".to_q", a + ".to_k", a + ".to_v"}, true));
    int q = P.view(qkv, 0, C), k = P.view(qkv, C, C), v = P.view(qkv, 2 * C, C);
    int o = b.attn(q, k, v, 1, HW, HW, -1);
    h = b.conv(o, make_conv(E, m, a + ".to_out.0", 0), 1, 0, h);
  }
  h = build_resnet(b, m, "encoder.mid_block.resnets.1", h, G, eps, false);
  h = b.gn(h, make_norm(E, m, "encoder.conv_norm_out"), G, eps, 1);
  h = b.conv(h, make_conv(E, m, "encoder.conv_out", 1));
  E->venc_out = b.conv(h, make_conv(E, m, "quant_conv", 0), 1, 0, -1, 0, /*out_f32=*/1);
  plan_gn_stats(P);
}

// f-2: CLIPTextModel (transformers; dataloader.py:633-646 `text_encoder(input_ids)[0]`): token + position embeddings, pre-LN
// transformer layers with causal self-attention and a quick_gelu (or erf-GELU) MLP, final LayerNorm.  Forward only.
// which = 1: the second tower of an SDXL-style model ("text2", CLIPTextModelWithProjection).  text_hidden_layer = -2: the program output
// is hidden_states[-2] (the input of the last layer, no final LayerNorm) as StableDiffusionXLPipeline.encode_prompt reads both towers;
// the second tower also keeps final_layer_norm(last layer) for its pooled head (dd_text_encode_tower).
void build_text_encoder(dd_engine* E, int which) {
  const dd_config& c = E->cfg;
  Program& P = which ? E->text2 : E->text;
  P.want_grad = false;
  Builder b(E, P);
  const std::string m = which ? "text2" : "text", tm = "text_model.";
  const HostTensor& tok = E->get(m, tm + "embeddings.token_embedding.weight");
  const HostTensor& pos = E->get(m, tm + "embeddings.position_embedding.weight");
  const int vocab = (int)tok.shape[0], C = (int)tok.shape[1];
  if ((int)pos.shape[0] < c.text_len) throw std::runtime_error("text encoder has fewer positions than text_len");
  const bool two = E->has("text2", tm + "embeddings.token_embedding.weight");
  if (!two && C != c.unet_cross_dim) throw std::runtime_error("text encoder width != UNet cross_attention_dim");
  if (c.text_hidden_layer != 0 && c.text_hidden_layer != -2) throw std::runtime_error("text_hidden_layer must be 0 (last_hidden_state) or -2");
  const int heads = which ? (c.text2_heads > 0 ? c.text2_heads : 20) : (c.text_heads > 0 ? c.text_heads : 12);
  if (C % heads) throw std::runtime_error("text hidden size is not divisible by its head count");
  float* te = (float*)E->wupload(tok.data.data(), tok.numel() * 4);
  float* pe = (float*)E->wupload(pos.data.data(), pos.numel() * 4);
  const int Bt = 2 * c.max_batch, T = c.text_len;
  const float eps0 = which ? c.text2_eps : c.text_eps, eps = eps0 > 0.f ? eps0 : 1e-5f;
  const int act = which ? c.text2_act : c.text_act;
  if (!E->text_ids) { E->text_batch = Bt; E->text_ids = (int*)E->dmalloc((size_t)Bt * T * 4); }
  int x = P.tensor(Bt, T, 1, C);
  const int in = x;
  int layers = 0;
  char buf[160];
  for (;; ++layers) {
    snprintf(buf, sizeof buf, "%sencoder.layers.%d.layer_norm1.weight", tm.c_str(), layers);
    if (!E->has(m, buf)) break;
  }
  if (layers < 2 && c.text_hidden_layer == -2) throw std::runtime_error("text_hidden_layer = -2 needs at least two layers");
  int penultimate = -1;
  for (int l = 0; l < layers; ++l) {
    if (l == layers - 1) penultimate = x;
    snprintf(buf, sizeof buf, "%sencoder.layers.%d", tm.c_str(), l);
    const std::string p = buf;
    int h = b.ln(x, make_norm(E, m, p + ".layer_norm1"), eps);
    int qkv = b.conv(h, make_conv_cat(E, m, {p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, true));
    int q = P.view(qkv, 0, C), k = P.view(qkv, C, C), v = P.view(qkv, 2 * C, C);
    int o = b.attn(q, k, v, heads, T, T, -1, /*causal=*/1);
    x = b.conv(o, make_conv(E, m, p + ".self_attn.out_proj", 0), 1, 0, x);
    h = b.ln(x, make_norm(E, m, p + ".layer_norm2"), eps);
    h = b.conv(h, make_conv(E, m, p + ".mlp.fc1", 0));
    h = b.act(h, act);
    x = b.conv(h, make_conv(E, m, p + ".mlp.fc2", 0), 1, 0, x);
  }
  const int fin = b.ln(x, make_norm(E, m, tm + "final_layer_norm"), eps, /*keep=*/true);
  const int out = c.text_hidden_layer == -2 ? penultimate : fin;
  if (which) {
    E->text2_in = in; E->text2_out = out; E->text2_final = fin;
    E->tok_emb2 = te; E->pos_emb2 = pe; E->text2_vocab = vocab; E->text2_hidden = C;
    if (E->has(m, "text_projection.weight")) {
      const HostTensor& pw = E->get(m, "text_projection.weight");
      if ((int)pw.shape[1] != C) throw std::runtime_error("text2 text_projection width != hidden size");
      E->text2_proj = (int)pw.shape[0];
      E->text2_proj_w = (float*)E->wupload(pw.data.data(), pw.numel() * 4);
    }
    if (E->text_hidden + C != c.unet_cross_dim) throw std::runtime_error("text + text2 widths != UNet cross_attention_dim");
  } else {
    E->text_in = in; E->text_out = out;
    E->tok_emb = te; E->pos_emb = pe; E->text_vocab = vocab; E->text_hidden = C;
  }
  check_transients(P);
}

void build_guide(dd_engine* E) {
  const dd_config& c = E->cfg;
  Program& P = E->guide;
  P.want_grad = c.enable_grad != 0;
  P.f32 = true;   // exact fp32 forward, masks and VJP (guide_f32.hip): the energy gradient goes through this network's ReLU masks
  Builder b(E, P);
  const std::string m = "guide";
  const int B = c.max_batch, S = c.guide_input_size;
  const float eps = c.guide_bn_eps;
  E->guide_in = P.tensor(B, S, S, 3);
  // timm ResNet family (model_utils.py:47-79): widths, groups (ResNeXt) and the bottleneck width (Wide-ResNet) come from the
  // weight shapes of the state dict
  int h = b.conv(E->guide_in, make_conv_bn(E, m, "conv1", "bn1", 3, eps, 3), 2, 0, -1, 1);
  h = b.maxpool(h);
  char buf[128];
  for (int li = 0; li < c.guide_stages; ++li)
    for (int bi = 0; bi < c.guide_blocks[li]; ++bi) {
      const int stride = (bi == 0 && li > 0) ? 2 : 1;
      snprintf(buf, sizeof buf, "layer%d.%d", li + 1, bi);
      const std::string p = buf;
      int o = b.conv(h, make_conv_bn(E, m, p + ".conv1", p + ".bn1", 0, eps, P.t[h].C), 1, 0, -1, 1);
      o = b.conv(o, make_conv_bn(E, m, p + ".conv2", p + ".bn2", 1, eps, P.t[o].C), stride, 0, -1, 1);
      int sc = h;
      if (E->has(m, p + ".downsample.0.weight"))
        sc = b.conv(h, make_conv_bn(E, m, p + ".downsample.0", p + ".downsample.1", 0, eps, P.t[h].C), stride);
      h = b.conv(o, make_conv_bn(E, m, p + ".conv3", p + ".bn3", 0, eps, P.t[o].C), 1, 0, sc, 1);
    }
  E->guide_feat = h;
  if (P.want_grad) plan_backward(P);
  plan_gn_stats(P);
}

// timm mobilenetv2_100 (model_utils.py:64-71) forward_features: conv_stem/bn1/ReLU6 -> blocks (stage 0: DepthwiseSeparableConv = conv_dw, bn1,
// ReLU6, conv_pw, bn2; later stages: InvertedResidual = conv_pw, bn1, ReLU6, conv_dw (stride), bn2, ReLU6, conv_pwl, bn3, + x when the
// stride is 1 and the channel count is unchanged) -> conv_head, bn2, ReLU6.  Exact fp32 like the ResNets (ReLU6 masks); depthwise
// convolutions are grouped convolutions with one channel per group (block-diagonal packing, K-steps outside the groups skipped).
void build_guide_mbv2(dd_engine* E) {
  const dd_config& c = E->cfg;
  Program& P = E->guide;
  P.want_grad = c.enable_grad != 0;
  P.f32 = true;
  Builder b(E, P);
  const std::string m = "guide";
  const int B = c.max_batch, S = c.guide_input_size;
  const float eps = c.guide_bn_eps;
  E->guide_in = P.tensor(B, S, S, 3);
  int h = b.conv(E->guide_in, make_conv_bn(E, m, "conv_stem", "bn1", 1, eps, 3), 2, 0, -1, /*relu6=*/2);
  char buf[128];
  for (int s = 0; s < c.guide_stages; ++s)
    for (int bi = 0; bi < c.guide_blocks[s]; ++bi) {
      const int stride = bi == 0 ? c.guide_strides[s] : 1;
      snprintf(buf, sizeof buf, "blocks.%d.%d", s, bi);
      const std::string p = buf;
      const int x = h;
      int o;
      if (E->has(m, p + ".conv_pwl.weight")) {
        o = b.conv(x, make_conv_bn(E, m, p + ".conv_pw", p + ".bn1", 0, eps, P.t[x].C), 1, 0, -1, 2);
        o = b.conv(o, make_conv_bn(E, m, p + ".conv_dw", p + ".bn2", 1, eps, P.t[o].C), stride, 0, -1, 2);
        const HostTensor& wl = E->get(m, p + ".conv_pwl.weight");
        const bool skip = stride == 1 && (int)wl.shape[0] == P.t[x].C;
        h = b.conv(o, make_conv_bn(E, m, p + ".conv_pwl", p + ".bn3", 0, eps, P.t[o].C), 1, 0, skip ? x : -1, 0);
      } else {
        o = b.conv(x, make_conv_bn(E, m, p + ".conv_dw", p + ".bn1", 1, eps, P.t[x].C), stride, 0, -1, 2);
        const HostTensor& wp = E->get(m, p + ".conv_pw.weight");
        const bool skip = stride == 1 && (int)wp.shape[0] == P.t[x].C;
        h = b.conv(o, make_conv_bn(E, m, p + ".conv_pw", p + ".bn2", 0, eps, P.t[o].C), 1, 0, skip ? x : -1, 0);
      }
    }
  h = b.conv(h, make_conv_bn(E, m, "conv_head", "bn2", 0, eps, P.t[h].C), 1, 0, -1, 2);
  E->guide_feat = h;
  if (P.t[h].C != guide_feat_dim_decl(c)) throw std::runtime_error("mobilenetv2 guide: conv_head width != guide_feature_dim");
  if (P.want_grad) plan_backward(P);
  plan_gn_stats(P);
}

// open_clip VisionTransformer (the image tower behind `image_encoder.encode_image` when --arch open_clip_vit_b32, the reference's default
// guide; model_utils.py:80-87): conv1 (stride = kernel = patch, no bias) -> [class_embedding; patches] + positional_embedding -> ln_pre ->
// residual attention blocks (ln_1 -> nn.MultiheadAttention (fused in_proj) -> +x ; ln_2 -> c_fc -> GELU -> c_proj -> +x) -> ln_post on
// the class token -> @ proj.  No ReLU / max-pool masks: bf16 MFMA like the UNet.  Width, depth, MLP size and output dim come from the
// state dict; heads / patch / activation from dd_config.
void build_guide_vit(dd_engine* E) {
  const dd_config& c = E->cfg;
  Program& P = E->guide;
  P.want_grad = c.enable_grad != 0;
  Builder b(E, P);
  const std::string m = "guide", v = "visual.";
  const int B = c.max_batch, S = c.guide_input_size, p = c.guide_vit_patch;
  if (p < 1 || S % p) throw std::runtime_error("ViT guide: guide_input_size must be a multiple of the patch size");
  const HostTensor& w1 = E->get(m, v + "conv1.weight");
  const int W = (int)w1.shape[0];
  const int heads = c.guide_vit_heads > 0 ? c.guide_vit_heads : W / 64;
  if ((int)w1.shape[2] != p || (int)w1.shape[3] != p || W % heads) throw std::runtime_error("ViT guide: conv1 / heads do not match the config");
  const int np = (S / p) * (S / p);
  E->guide_in = P.tensor(B, S, S, 3, true, /*f32_act=*/true, /*f32_grad=*/true);
  int h = b.patchify(E->guide_in, p);
  // conv1 as a linear over the flattened patch: weight [W, 3*p*p], k = (c, iy, ix) = the memory order of conv1.weight
  h = b.conv(h, make_conv_raw(E, w1.data.data(), nullptr, false, W, 3 * p * p, 1, 1, 0, false, c.enable_grad != 0));
  {
    auto nw = std::make_unique<NormW>();
    const HostTensor& ce = E->get(m, v + "class_embedding");
    const HostTensor& pe = E->get(m, v + "positional_embedding");
    if ((int)ce.numel() != W || (int)pe.numel() != (np + 1) * W) throw std::runtime_error("ViT guide: embedding shapes do not match the input size");
    nw->C = W;
    nw->gamma = (float*)E->wupload(ce.data.data(), ce.numel() * 4);
    nw->beta = (float*)E->wupload(pe.data.data(), pe.numel() * 4);
    E->norms.push_back(std::move(nw));
    h = b.vit_embed(h, E->norms.back().get());
  }
  const float eps = 1e-5f;
  int x = b.ln(h, make_norm(E, m, v + "ln_pre"), eps, /*keep=*/true);   // the residual stream
  char buf[160];
  const int N = np + 1;
  for (int l = 0;; ++l) {
    snprintf(buf, sizeof buf, "%stransformer.resblocks.%d", v.c_str(), l);
    const std::string r = buf;
    if (!E->has(m, r + ".ln_1.weight")) break;
    int n = b.ln(x, make_norm(E, m, r + ".ln_1"), eps);
    const HostTensor& iw = E->get(m, r + ".attn.in_proj_weight");
    const HostTensor& ib = E->get(m, r + ".attn.in_proj_bias");
    int qkv = b.conv(n, make_conv_raw(E, iw.data.data(), ib.data.data(), true, 3 * W, W, 1, 1, 0, false, c.enable_grad != 0));
    int q = P.view(qkv, 0, W), k = P.view(qkv, W, W), vv = P.view(qkv, 2 * W, W);
    int a = b.attn(q, k, vv, heads, N, N, -1);
    x = b.conv(a, make_conv(E, m, r + ".attn.out_proj", 0), 1, 0, x);
    n = b.ln(x, make_norm(E, m, r + ".ln_2"), eps);
    int f = b.conv(n, make_conv(E, m, r + ".mlp.c_fc", 0));
    f = b.act(f, c.guide_vit_act);
    x = b.conv(f, make_conv(E, m, r + ".mlp.c_proj", 0), 1, 0, x);
  }
  int cls = b.select_first(x);
  cls = b.ln(cls, make_norm(E, m, v + "ln_post"), eps);
  // pooled @ proj: proj is [W, D]; as a linear layer its weight is proj^T [D, W]
  const HostTensor& pr = E->get(m, v + "proj");
  const int D = (int)pr.shape[1];
  std::vector<float> pt;
  if (!E->shape_only) {
    pt.resize((size_t)D * W);
    for (int d = 0; d < D; ++d)
      for (int k = 0; k < W; ++k) pt[(size_t)d * W + k] = pr.data[(size_t)k * D + d];
  }
  E->guide_feat = b.conv(cls, make_conv_raw(E, pt.data(), nullptr, false, D, W, 1, 1, 0, false, c.enable_grad != 0), 1, 0, -1, 0, /*out_f32=*/1);
  if (D != c.guide_feature_dim) throw std::runtime_error("ViT guide: projection dim != guide_feature_dim");
  if (P.want_grad) plan_backward(P);
  plan_gn_stats(P);
}


}  // namespace ddi
