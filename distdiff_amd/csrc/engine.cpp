// The expansion engine: a static op-graph executor for the SD-1.x UNet, the AutoencoderKL decoder and the
// ResNet-50 guide, with a hand-derived reverse program (input-gradients only, no autograd) composed from
// per-op VJPs, plus the guided DDIM sampler drivers behind the C ABI of include/distdiff_hip.h.
//
// Reference functions replaced: generate_data.py:109-121 (denoise_one_step), :687-732 (transform_guidance),
// :735-767 (direct_guidance), :1161-1228 (loop + decode); diffusers / timm module forwards as listed in
// SURVEY.md section 8a. Activations are NHWC bf16 matrices; every forward op stashes what its VJP needs
// (288 GB HBM makes a full stash viable), the reverse program is built once at finalize time.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/distdiff_hip.h"
#include "../../include/distdiff_hip_ops.h"
#include "kernels.h"

namespace {

#define HIPCHK(x)                                                                                     \
  do {                                                                                                \
    hipError_t _e = (x);                                                                              \
    if (_e != hipSuccess) {                                                                           \
      char _b[512];                                                                                   \
      snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
      throw std::runtime_error(_b);                                                                   \
    }                                                                                                 \
  } while (0)

inline int rup(int v, int m) { return (v + m - 1) / m * m; }
inline size_t rup_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

template <class T>
T* dev_upload(const std::vector<T>& h) {
  T* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, std::max<size_t>(h.size() * sizeof(T), 16)));
  if (!h.empty()) HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

// ---------------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------------
struct ConvW {
  int Cout = 0, Cin = 0, KH = 1, KW = 1, pad = 0;
  int pad_br = 0;              // extra zero rows/cols at the bottom/right only (AutoencoderKL encoder downsample: F.pad (0,1,0,1))
  bool geglu = false;
  int groups = 1;              // grouped convolution (ResNeXt guide); fp32 programs only
  bool f32 = false;            // fp32 packing for the guide program (guide_f32.hip): wf_* / sf / sb describe the fp32 matrices
  float* wf_fwd = nullptr; float* wf_bwd = nullptr;
  PackedConv sf{}, sb{};
  bf16_t* w_fwd = nullptr; int* tap_fwd = nullptr;
  bf16_t* w_bwd = nullptr; int* tap_bwd = nullptr;
  float* ln_c1 = nullptr;      // LayerNorm folded into this linear (CF_LNFOLD): column sums of the folded, bf16-rounded weights (packed order)
  float* bias = nullptr;       // [Cout] (packed order for GEGLU) or null
  float* bias_table = nullptr; // [n_steps][Cout] per-timestep effective bias (resnet conv1 + time_emb_proj)
  float* bias_table_img = nullptr;   // SDXL text_time conditioning: [n_steps][2B][Cout], filled by dd_set_added_cond
  // fp32 copies kept for the time-embedding tables
  float* temb_w = nullptr; float* temb_b = nullptr;
};
struct NormW { float* gamma = nullptr; float* beta = nullptr; int C = 0; };

// ---------------------------------------------------------------------------------------------------
// op graph
// ---------------------------------------------------------------------------------------------------
struct Tn {              // activation tensor or channel view
  size_t off = 0;        // byte offset in the activation slab
  size_t goff = 0;       // byte offset in the gradient slab
  int parent = -1;       // gradient-tracking parent (self for base tensors)
  int rows = 0, C = 0, ld = 0, B = 0, H = 0, W = 0;
  bool f32 = false, grad = false;
  bool gf32 = false;     // the gradient of this tensor is fp32 (always in fp32 programs; the image input of the ViT guide)
  // Transient activation: only read by the operation that follows its producer and never by a reverse program (no weight gradients
  // are computed, so the INPUT of a convolution is dead after it ran: GroupNorm / LayerNorm / GEGLU / activation outputs).  It lives in
  // one of two ping-pong buffers shared by all instances instead of the per-instance stash slab.
  bool transient = false; int tr_slot = 0;
};

enum OpKind { OP_CONV, OP_GN, OP_LN, OP_ATTN, OP_CONCAT, OP_MAXPOOL, OP_GAP, OP_ACT, OP_PATCHIFY, OP_VITEMBED, OP_SELECT, OP_DUP };

struct Op {
  OpKind kind;
  int x = -1, y = -1, res = -1, raw = -1, q = -1, k = -1, v = -1, x2 = -1;
  ConvW* cw = nullptr;
  NormW* nw = nullptr;
  int stride = 1, up = 0, relu = 0, out_f32 = 0, use_table = 0;
  int G = 0, silu = 0; float eps = 0;
  int heads = 0, D = 0, Nq = 0, Nk = 0, cross_slot = -1;
  int causal = 0, act_kind = 0;
  int patch = 0, sel_stride = 0;   // OP_PATCHIFY: patch size; OP_SELECT: row stride (tokens per image)
  size_t stats_off = 0;  // fp32 stats / lse in the activation slab
  bool fused = false;    // OP_CONCAT: both operands live inside the output buffer (column views): no copy, forward or backward
  // backward plan
  bool x_acc = false, res_acc = false, x2_acc = false;
  bool res_alias = false;   // the residual's gradient buffer IS this op's output-gradient buffer (first write: no copy kernel)
  // GroupNorm statistics from the producing convolutions (plan_gn_stats): conv ops emit per-(64-row block, channel) partials into
  // the fp32 block at part_off (CF_STATS), the GroupNorm op merges them instead of reading the tensor once more
  bool part = false; size_t part_off = 0; int part_ld = 0;
  std::vector<int> producers;
  // LayerNorm folded into the linear that follows (plan_ln_fold): the OP_LN only produces (mean, rstd) -- from the row partials of
  // the producing GEMM (rowstat_from = its op index, CF_ROWSTATS) when that GEMM can emit them, else from one read of the tensor --
  // and the OP_CONV reads the LayerNorm's INPUT (x_fwd) with CF_LNFOLD; the backward plan is untouched (x stays the LayerNorm output)
  bool ln_fold = false; int x_fwd = -1; size_t ln_stats_off = 0;
  int rowstat_from = -1; bool rowstat_emit = false; int rowstat_ld = 0;
  double flops = 0;
};

struct Program {
  std::vector<Tn> t;
  std::vector<Op> ops;
  size_t act_bytes = 0, grad_bytes = 0;
  size_t scratch_partial = 0, scratch_tmp = 0;  // shared scratch requirements (bytes)
  bool want_grad = false;
  bool f32 = false;      // every activation AND gradient of this program is fp32 (the guide network, guide_f32.hip)
  mutable std::vector<char> emitted;   // per op, per forward run: this convolution did emit its GroupNorm partials
  mutable std::vector<int> row_spans;  // per op, per forward run: column spans of the LayerNorm row partials this GEMM emitted (0 = none)
  size_t scratch_rowpart = 0;          // bytes of the shared row-partial buffer (producer GEMM -> LayerNorm statistics, adjacent ops)
  size_t tr_max = 0;     // bytes of one transient ping-pong buffer
  int tr_count = 0;
  int transient(int B, int H, int W, int C, bool grad = true) {
    if (f32 || getenv("DD_NO_TRANSIENT")) return tensor(B, H, W, C, grad);
    const size_t save = act_bytes;
    const int id = tensor(B, H, W, C, grad);
    tr_max = std::max(tr_max, act_bytes - save);
    act_bytes = save;                    // give the stash bytes back: the tensor lives in the transient buffers
    t[id].off = 0; t[id].transient = true; t[id].tr_slot = tr_count++ & 1;
    return id;
  }

  int tensor(int B, int H, int W, int C, bool grad = true, bool f32_act = false, bool f32_grad = false) {
    Tn n;
    n.B = B; n.H = H; n.W = W; n.rows = B * H * W; n.C = C; n.ld = f32 ? rup(C, 4) : rup(C, 8); n.f32 = f32_act || f32;
    n.grad = grad && want_grad;
    n.gf32 = f32 || f32_grad;
    n.off = act_bytes;
    act_bytes += rup_sz((size_t)n.rows * n.ld * (n.f32 ? 4 : 2), 256);
    if (n.grad) { n.goff = grad_bytes; grad_bytes += rup_sz((size_t)n.rows * n.ld * (n.gf32 ? 4 : 2), 256); }
    n.parent = (int)t.size();
    t.push_back(n);
    return (int)t.size() - 1;
  }
  int view(int base, int c0, int C) {
    Tn n = t[base];
    n.off += (size_t)c0 * 2; n.goff += (size_t)c0 * 2; n.C = C; n.parent = t[base].parent;
    t.push_back(n);
    return (int)t.size() - 1;
  }
  size_t fp32_block(size_t count) {
    const size_t o = act_bytes;
    act_bytes += rup_sz(count * 4, 256);
    return o;
  }
};

struct Profiler {   // HIP-event timing of every op, by kernel family (dd_profile_*)
  enum { CONV = 0, ATTN = 1, NORM = 2, OTHER = 3, NFAM = 4 };
  bool on = false;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Rec { int fam; double flops; size_t e0, e1; int M, N, K, bwd; };
  std::vector<Rec> recs;
  bool chain = false;    // the last event recorded is the end of the previous op of the same run: it doubles as this op's start
  hipEvent_t get() {
    if (used == pool.size()) { hipEvent_t e; hipEventCreate(&e); pool.push_back(e); }
    return pool[used++];
  }
  void new_run() { chain = false; }   // other launches may sit between two program runs: the next op records its own start
  void begin(int fam, double flops, hipStream_t s, int M = 0, int N = 0, int K = 0, int bwd = 0) {
    if (!on) return;
    Rec r; r.fam = fam; r.flops = flops; r.M = M; r.N = N; r.K = K; r.bwd = bwd;
    if (chain) r.e0 = used - 1;                      // one event per op boundary: half the recording overhead inside the timed step
    else { r.e0 = used; hipEventRecord(get(), s); }
    r.e1 = 0;
    recs.push_back(r);
  }
  void end(hipStream_t s) {
    if (!on) return;
    recs.back().e1 = used;
    hipEventRecord(get(), s);
    chain = true;
  }
};

struct Ctx {  // per-call execution context
  char* act = nullptr;   // activation slab of the instance being run
  char* grad = nullptr;  // shared gradient slab
  char* scratch_partial = nullptr; size_t partial_cap = 0;
  char* scratch_tmp = nullptr;
  float* gn_scratch = nullptr;
  float* rowpart = nullptr;      // LayerNorm row partials of the GEMM that ran last (CF_ROWSTATS)
  const int* tap1x1 = nullptr;   // device int: the 1x1 tap, for GEMMs issued outside a ConvW (wide-head attention)
  size_t tmp_cap = 0;
  int step_index = 0;
  int B = 0;             // live batch of this call (<= built batch)
  hipStream_t s = nullptr;
  const std::vector<std::pair<bf16_t*, bf16_t*>>* cross_kv = nullptr;  // per cross-attention slot
  double* flops = nullptr;
  Profiler* prof = nullptr;
  bool stash = true;     // false on plain (no-VJP) steps: skip stores that only the reverse program reads
  char* tr = nullptr; size_t tr_stride = 0;   // transient ping-pong buffers of the program being run
  int img_bias = 0;      // > 0: the time-embedding bias is per image (SDXL added conditioning): number of images (2B) of the tables
};

inline char* act_raw(const Ctx& c, const Tn& t) { return t.transient ? c.tr + (size_t)t.tr_slot * c.tr_stride : c.act + t.off; }
inline bf16_t* act_ptr(const Ctx& c, const Tn& t) { return (bf16_t*)act_raw(c, t); }
inline bf16_t* grad_ptr(const Ctx& c, const Tn& t) { return (bf16_t*)(c.grad + t.goff); }
inline float* act_f32(const Ctx& c, const Tn& t) { return (float*)act_raw(c, t); }
inline float* grad_f32(const Ctx& c, const Tn& t) { return (float*)(c.grad + t.goff); }

}  // namespace

struct dd_engine {
  dd_config cfg{};
  std::string err;
  std::unordered_map<std::string, HostTensor> raw;  // "model/key" -> fp32 host copy until finalize
  bool finalized = false;

  std::vector<std::unique_ptr<ConvW>> convs;
  std::vector<std::unique_ptr<NormW>> norms;
  std::vector<void*> dev_allocs;

  Program unet, vae, guide;
  int unet_in = -1, unet_out = -1, vae_in = -1, vae_out = -1, guide_in = -1, guide_feat = -1;
  // f-2: the stage before the loop (built when the weights are present)
  Program venc, text;
  int venc_in = -1, venc_out = -1, text_in = -1, text_out = -1;
  char* venc_slab = nullptr; char* text_slab = nullptr;
  float* tok_emb = nullptr; float* pos_emb = nullptr; int text_vocab = 0, text_hidden = 0, text_batch = 0;
  int* text_ids = nullptr;
  struct CrossSlot { ConvW* wk; ConvW* wv; int C; };
  std::vector<CrossSlot> cross_slots;
  std::vector<std::pair<bf16_t*, bf16_t*>> cross_kv;  // device K,V [2B*text_len, C] per slot
  bf16_t* ctx_bf16 = nullptr;                         // [2B*text_len, cross_dim]
  std::vector<ConvW*> temb_convs;                     // resnet conv1's with time_emb_proj
  float* temb_w1 = nullptr; float* temb_b1 = nullptr; float* temb_w2 = nullptr; float* temb_b2 = nullptr;
  // SDXL text_time conditioning
  float* add_w1 = nullptr; float* add_b1 = nullptr; float* add_w2 = nullptr; float* add_b2 = nullptr;
  float* d_emb = nullptr;          // [n_steps][TE] time_embedding(t) of the current schedule
  bool added_cond_set = false;

  // schedule
  std::vector<int> timesteps;
  float* coef_table = nullptr;   // [n][8]: guidance_scale, sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev), -, -, -
  dd_sampler_params sp{};
  // prototypes
  float* Pc = nullptr; float* Pg = nullptr; int pC = 0, pK = 0, pD = 0;

  // instance slabs: [0 .. P-1]; each holds UNet | VAE | guide activations + small fp32 state
  struct Inst {
    char* unet = nullptr; char* vae = nullptr; char* guide = nullptr;
    float* eps2 = nullptr;  // view into unet slab (conv_out fp32 output)
    float* z_in = nullptr; float* z_next = nullptr; float* x0 = nullptr; float* feat = nullptr; float* gfeat = nullptr;
  };
  std::vector<Inst> inst;
  char* grad_slab = nullptr;   // shared by the three programs (max of their grad sizes)
  char* tr_slab = nullptr;     // two transient ping-pong buffers, shared by every program and instance (they run one after another)
  char* scratch_partial = nullptr; size_t partial_cap = 0;
  char* scratch_tmp = nullptr; size_t tmp_cap = 0;
  float* gn_scratch = nullptr;
  float* rowpart = nullptr;
  int* tap1x1 = nullptr;
  float* f32_tmp[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [B,4,L,L] fp32 temporaries
  float* img_tmp = nullptr;    // [B,3,8L,8L] fp32
  float* score_tmp = nullptr;
  float* sample_w = nullptr; bool sample_w_set = false;   // per-image energy weights (dd_set_sample_weights); default 1/B
  float* image_scores = nullptr;                         // per-image energies of the last guidance call
  const float* image_override = nullptr; int image_override_count = 0;   // dd_debug_set_images: parity tests evaluate the guide at given images
  size_t total_bytes = 0;
  double flops = 0;
  Profiler prof;
  // hipGraph replay of the plain denoise step (~700 launches): one captured graph per (timestep index, latent buffers); the bias
  // tables / DDIM coefficients of a step and every workspace pointer are static, so a step is the same launch sequence every time
  struct StepGraph { hipGraphExec_t exec = nullptr; double flops = 0; int seen = 0; };
  std::unordered_map<std::string, StepGraph> step_graphs;
  hipStream_t gstream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
  bool graphs_ok = true;

  void dfree(void* p) {
    if (!p) return;
    for (size_t i = 0; i < dev_allocs.size(); ++i)
      if (dev_allocs[i] == p) { dev_allocs[i] = dev_allocs.back(); dev_allocs.pop_back(); break; }
    hipFree(p);
  }
  std::vector<void*> sched_allocs;   // tables of the current schedule (replaced by the next dd_set_schedule)
  // packed weights: every weight-derived device buffer in creation order (dd_packed_bytes / dd_export_packed / dd_import_packed)
  int declared = 0;
  bool shape_only = false;           // tensors were declared (shapes only): buffers are allocated, their content arrives by import
  std::vector<std::pair<char*, size_t>> packed;
  void* wupload(const void* host, size_t bytes) {
    void* d = dmalloc(bytes, false);
    packed.push_back({(char*)d, bytes});
    if (!shape_only && bytes) {
      if (!host) throw std::runtime_error("internal: weight upload without host data");
      hipError_t e = hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) throw std::runtime_error(std::string("weight upload failed: ") + hipGetErrorString(e));
    }
    return d;
  }
  void* dmalloc(size_t bytes, bool zero = true) {
    void* p = nullptr;
    bytes = std::max<size_t>(bytes, 256);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) throw std::runtime_error("hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
    if (zero) HIPCHK(hipMemset(p, 0, bytes));
    dev_allocs.push_back(p);
    total_bytes += bytes;
    return p;
  }
  const HostTensor& get(const std::string& model, const std::string& key) {
    auto it = raw.find(model + "/" + key);
    if (it == raw.end()) throw std::runtime_error("missing weight " + model + "/" + key);
    return it->second;
  }
  bool has(const std::string& model, const std::string& key) { return raw.count(model + "/" + key) != 0; }
};

namespace {

inline int guide_feat_dim_decl(const dd_config& c) {
  return c.guide_feature_dim > 0 ? c.guide_feature_dim : c.guide_planes[c.guide_stages - 1] * c.guide_expansion;
}

// ---------------------------------------------------------------------------------------------------
// weight construction
// ---------------------------------------------------------------------------------------------------
// Every device buffer that holds weight-derived data goes through E->wupload(): it is registered, in creation order, in the engine's
// packed-weight list (dd_packed_bytes / dd_export_packed / dd_import_packed), and an engine that only knows the tensor SHAPES
// (dd_declare_tensor: a rank that will receive the packed buffers over RCCL) allocates it without packing anything on the host.

// fp32 packing (guide program): w is [Cout][Cin/groups][KH][KW]
ConvW* make_conv_f32(dd_engine* E, const float* w, const float* bias, int Cout, int Cin, int KH, int KW, int pad, int groups,
                     bool need_bwd) {
  auto cw = std::make_unique<ConvW>();
  cw->Cout = Cout; cw->Cin = Cin; cw->KH = KH; cw->KW = KW; cw->pad = pad; cw->groups = groups; cw->f32 = true;
  for (int mode = 0; mode < (need_bwd ? 2 : 1); ++mode) {
    PackedConv& sh = mode ? cw->sb : cw->sf;
    sh = pack_conv_shape_f32(Cout, Cin, KH, KW, mode, groups);
    std::vector<float> wp;
    std::vector<int> tt;
    if (!E->shape_only) {
      wp.resize((size_t)sh.N * sh.K); tt.resize(sh.ntaps);
      pack_conv_weight_f32(w, Cout, Cin, KH, KW, pad, mode, groups, wp.data(), tt.data());
    }
    float* d = (float*)E->wupload(wp.data(), (size_t)sh.N * sh.K * 4);
    int* t = (int*)E->wupload(tt.data(), (size_t)sh.ntaps * 4);
    if (mode) { cw->wf_bwd = d; cw->tap_bwd = t; } else { cw->wf_fwd = d; cw->tap_fwd = t; }
  }
  if (bias || E->shape_only) cw->bias = (float*)E->wupload(bias, (size_t)Cout * 4);
  E->convs.push_back(std::move(cw));
  return E->convs.back().get();
}

// has_bias is passed explicitly: a shape-only engine has no host data to look at.
// ln_gamma / ln_beta (host, [Cin]; `fold` tells a shape-only engine): the LayerNorm in front of this linear is folded into it --
//   LN(x) W^T + b = rstd * (x (gamma o W)^T - mean * c1) + (b + W beta),  c1[n] = sum_k (gamma o W)[n, k]
// (kernels.h CF_LNFOLD): the FORWARD packing holds gamma o W, the bias W beta + b, ln_c1 the column sums of the bf16-rounded folded
// weights (the rank-1 correction must cancel what the MFMAs actually accumulate); the input-gradient packing keeps the plain W, the
// LayerNorm's own backward multiplies by gamma as before.
ConvW* make_conv_raw(dd_engine* E, const float* w, const float* bias, bool has_bias, int Cout, int Cin, int KH, int KW, int pad,
                     bool geglu, bool need_bwd, bool fold = false, const float* ln_gamma = nullptr, const float* ln_beta = nullptr) {
  auto cw = std::make_unique<ConvW>();
  cw->Cout = Cout; cw->Cin = Cin; cw->KH = KH; cw->KW = KW; cw->pad = pad; cw->geglu = geglu;
  if (fold && (KH != 1 || KW != 1)) throw std::runtime_error("LayerNorm folding needs a linear layer");
  std::vector<float> wfold, bfold, c1;
  if (fold && !E->shape_only) {
    wfold.resize((size_t)Cout * Cin); bfold.assign(Cout, 0.f); c1.assign(Cout, 0.f);
    for (int n = 0; n < Cout; ++n) {
      double sb = bias ? bias[n] : 0.0, sc = 0.0;
      for (int k = 0; k < Cin; ++k) {
        const float wf = w[(size_t)n * Cin + k] * ln_gamma[k];
        wfold[(size_t)n * Cin + k] = wf;
        sb += (double)w[(size_t)n * Cin + k] * ln_beta[k];
        sc += host_bf2f(host_f2bf(wf));
      }
      bfold[n] = (float)sb; c1[n] = (float)sc;
    }
  }
  for (int mode = 0; mode < (need_bwd ? 2 : 1); ++mode) {
    PackedConv& sh = mode ? cw->sb : cw->sf;
    sh = pack_conv_shape(Cout, Cin, KH, KW, mode);
    std::vector<bf16_t> wp;
    std::vector<int> tt;
    if (!E->shape_only) {
      wp.resize((size_t)sh.N * sh.K); tt.resize(sh.ntaps);
      pack_conv_weight((fold && mode == 0) ? wfold.data() : w, Cout, Cin, KH, KW, pad, mode, geglu, wp.data(), tt.data());
    }
    bf16_t* d = (bf16_t*)E->wupload(wp.data(), (size_t)sh.N * sh.K * 2);
    int* t = (int*)E->wupload(tt.data(), (size_t)sh.ntaps * 4);
    if (mode) { cw->w_bwd = d; cw->tap_bwd = t; } else { cw->w_fwd = d; cw->tap_fwd = t; }
  }
  if (has_bias || fold) {
    std::vector<float> b, c;
    if (!E->shape_only) {
      b.resize(Cout);
      const float* src = fold ? bfold.data() : bias;
      for (int n = 0; n < Cout; ++n) b[n] = src[geglu ? geglu_perm(n, Cout / 2) : n];
      if (fold) { c.resize(Cout); for (int n = 0; n < Cout; ++n) c[n] = c1[geglu ? geglu_perm(n, Cout / 2) : n]; }
    }
    cw->bias = (float*)E->wupload(b.data(), (size_t)Cout * 4);
    if (fold) cw->ln_c1 = (float*)E->wupload(c.data(), (size_t)Cout * 4);
  }
  E->convs.push_back(std::move(cw));
  return E->convs.back().get();
}

inline bool ln_fold_enabled() { static const bool on = !getenv("DD_NO_LN_FOLD"); return on; }

// ln: prefix of the LayerNorm to fold into this linear ("" = none)
ConvW* make_conv(dd_engine* E, const std::string& model, const std::string& prefix, int pad, bool geglu = false,
                 bool has_bias = true, const std::string& ln = "") {
  const HostTensor& w = E->get(model, prefix + ".weight");
  const int Cout = (int)w.shape[0], Cin = (int)w.shape[1];
  const int KH = w.shape.size() == 4 ? (int)w.shape[2] : 1, KW = w.shape.size() == 4 ? (int)w.shape[3] : 1;
  const bool hb = has_bias && E->has(model, prefix + ".bias");
  const float* b = hb ? E->get(model, prefix + ".bias").data.data() : nullptr;
  const bool fold = !ln.empty();
  return make_conv_raw(E, w.data.data(), b, hb, Cout, Cin, KH, KW, pad, geglu, E->cfg.enable_grad != 0, fold,
                       fold ? E->get(model, ln + ".weight").data.data() : nullptr, fold ? E->get(model, ln + ".bias").data.data() : nullptr);
}

// several linears sharing the input, concatenated along Cout (fused QKV)
ConvW* make_conv_cat(dd_engine* E, const std::string& model, const std::vector<std::string>& prefixes, bool with_bias,
                     const std::string& ln = "") {
  std::vector<float> w, b;
  int Cin = 0, Cout = 0;
  for (auto& p : prefixes) {
    const HostTensor& t = E->get(model, p + ".weight");
    Cin = (int)t.shape[1];
    Cout += (int)t.shape[0];
    w.insert(w.end(), t.data.begin(), t.data.end());
    if (with_bias) { const HostTensor& bb = E->get(model, p + ".bias"); b.insert(b.end(), bb.data.begin(), bb.data.end()); }
  }
  const bool fold = !ln.empty();
  return make_conv_raw(E, w.data(), with_bias ? b.data() : nullptr, with_bias, Cout, Cin, 1, 1, 0, false, E->cfg.enable_grad != 0, fold,
                       fold ? E->get(model, ln + ".weight").data.data() : nullptr, fold ? E->get(model, ln + ".bias").data.data() : nullptr);
}

// conv (no bias) followed by eval-mode BatchNorm, folded: w' = w * g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps)
// `cin_total` = channels of the input tensor: groups = cin_total / weight.shape[1] (ResNeXt, model_utils.py:56-63).  The guide
// program is fp32 (guide_f32.hip).
ConvW* make_conv_bn(dd_engine* E, const std::string& model, const std::string& conv, const std::string& bn, int pad, float eps,
                    int cin_total) {
  const HostTensor& w = E->get(model, conv + ".weight");
  const HostTensor& g = E->get(model, bn + ".weight");
  const HostTensor& be = E->get(model, bn + ".bias");
  const HostTensor& mu = E->get(model, bn + ".running_mean");
  const HostTensor& var = E->get(model, bn + ".running_var");
  const int Cout = (int)w.shape[0], Cg = (int)w.shape[1], KH = (int)w.shape[2], KW = (int)w.shape[3];
  if (Cg < 1 || cin_total % Cg || Cout % (cin_total / Cg))
    throw std::runtime_error("guide conv " + conv + ": weight shape does not divide the input channels (groups)");
  const int groups = cin_total / Cg;
  std::vector<float> wf, bf(Cout);
  if (!E->shape_only) {
    wf.resize(w.numel());
    const size_t per = (size_t)Cg * KH * KW;
    for (int n = 0; n < Cout; ++n) {
      const float sc = g.data[n] / sqrtf(var.data[n] + eps);
      for (size_t i = 0; i < per; ++i) wf[n * per + i] = w.data[n * per + i] * sc;
      bf[n] = be.data[n] - mu.data[n] * sc;
    }
  }
  return make_conv_f32(E, wf.data(), bf.data(), Cout, cin_total, KH, KW, pad, groups, E->cfg.enable_grad != 0);
}

NormW* make_norm(dd_engine* E, const std::string& model, const std::string& prefix) {
  auto nw = std::make_unique<NormW>();
  const HostTensor& g = E->get(model, prefix + ".weight");
  const HostTensor& b = E->get(model, prefix + ".bias");
  nw->C = (int)g.shape[0];
  nw->gamma = (float*)E->wupload(g.data.data(), (size_t)nw->C * 4);
  nw->beta = (float*)E->wupload(b.data.data(), (size_t)nw->C * 4);
  E->norms.push_back(std::move(nw));
  return E->norms.back().get();
}

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
        pr.rowstat_ld = l.rowstat_ld = (tx.C + 63) / 64;
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
  int attn(int q, int k, int v, int heads, int Nq, int Nk, int cross_slot, int causal = 0) {
    const Tn& tq = P.t[q];
    int y = P.tensor(tq.B, tq.H, tq.W, tq.C);
    Op op; op.kind = OP_ATTN; op.q = q; op.k = k; op.v = v; op.y = y; op.heads = heads; op.D = tq.C / heads; op.Nq = Nq; op.Nk = Nk;
    op.cross_slot = cross_slot; op.causal = causal;
    // wide heads (AutoencoderKL mid block) run through GEMMs on a materialised score matrix: per-image scratch (attention_gemm.hip)
    if (op.D >= 256 && cross_slot < 0 && !causal)
      P.scratch_tmp = std::max(P.scratch_tmp, attention_gemm_workspace(Nq, Nk, op.D, P.want_grad ? 1 : 0));
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

void run_fwd(const Program& P, const Ctx& c, int op_begin = 0, int op_end = -1) {
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
        const Tn& x = P.t[op.x_fwd >= 0 ? op.x_fwd : op.x]; const Tn& y = P.t[op.y];
        ConvGemmParams p; fill_conv(p, c);
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
        if (op.rowstat_emit) {
          // LayerNorm row partials for the op that follows: only when the kernel the launcher picks for this shape has the form
          p.rowpart = c.rowpart; p.rowpart_ld = op.rowstat_ld;
          int spans = 0;
          if (c.rowpart && conv_gemm_can_emit_rowstats(p, c.partial_cap, &spans) && spans <= op.rowstat_ld) { p.flags |= CF_ROWSTATS; P.row_spans[i] = spans; }
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
        p.B = q.B; p.H = op.heads; p.Nq = op.Nq; p.Nk = op.Nk; p.D = op.D; p.scale = 1.f / sqrtf((float)op.D);
        p.causal = op.causal;
        if (op.cross_slot < 0 && attention_gemm_supported(p) && c.tap1x1 && attention_gemm_workspace(p.Nq, p.Nk, p.D, 0) <= c.tmp_cap)
          HIPCHK(launch_attention_gemm_fwd(p, c.scratch_tmp, c.tap1x1, (float*)c.scratch_partial, c.partial_cap, c.s));
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

void run_bwd(const Program& P, const Ctx& c) {
  if (c.prof) c.prof->new_run();
  for (int i = (int)P.ops.size() - 1; i >= 0; --i) {
    const Op& op = P.ops[i];
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
        p.B = q.B; p.H = op.heads; p.Nq = op.Nq; p.Nk = op.Nk; p.D = op.D; p.scale = 1.f / sqrtf((float)op.D);
        p.d_o = grad_ptr(c, y); p.lddo = y.ld; p.dq = grad_ptr(c, q); p.lddq = q.ld;
        if (op.cross_slot < 0 && attention_gemm_supported(p) && c.tap1x1 && attention_gemm_workspace(p.Nq, p.Nk, p.D, 1) <= c.tmp_cap)
          HIPCHK(launch_attention_gemm_bwd(p, c.scratch_tmp, c.tap1x1, (float*)c.scratch_partial, c.partial_cap, c.s));
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
    int n = b.ln(h, make_norm(E, m, t + ".norm1"), 1e-5f);
    int qkv = b.conv(n, make_conv_cat(E, m, {t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"}, false, fold ? t + ".norm1" : ""));
    if (fold) b.fold_ln();
    int q = P.view(qkv, 0, C), k = P.view(qkv, C, C), v = P.view(qkv, 2 * C, C);
    int a = b.attn(q, k, v, heads, HW, HW, -1);
    h = b.conv(a, make_conv(E, m, t + ".attn1.to_out.0", 0), 1, 0, h);
    // cross attention: K,V of the text embeddings are computed once per prompt (dd_set_prompt)
    n = b.ln(h, make_norm(E, m, t + ".norm2"), 1e-5f);
    int q2 = b.conv(n, make_conv(E, m, t + ".attn2.to_q", 0, false, false, fold ? t + ".norm2" : ""));
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
    a = b.attn(q2, -1, -1, heads, HW, E->cfg.text_len, (int)E->cross_slots.size() - 1);
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
void build_text_encoder(dd_engine* E) {
  const dd_config& c = E->cfg;
  Program& P = E->text;
  P.want_grad = false;
  Builder b(E, P);
  const std::string m = "text", tm = "text_model.";
  const HostTensor& tok = E->get(m, tm + "embeddings.token_embedding.weight");
  const HostTensor& pos = E->get(m, tm + "embeddings.position_embedding.weight");
  E->text_vocab = (int)tok.shape[0]; E->text_hidden = (int)tok.shape[1];
  if ((int)pos.shape[0] < c.text_len) throw std::runtime_error("text encoder has fewer positions than text_len");
  if (E->text_hidden != c.unet_cross_dim) throw std::runtime_error("text encoder width != UNet cross_attention_dim");
  const int heads = c.text_heads > 0 ? c.text_heads : 12;
  if (E->text_hidden % heads) throw std::runtime_error("text hidden size is not divisible by text_heads");
  E->tok_emb = (float*)E->wupload(tok.data.data(), tok.numel() * 4);
  E->pos_emb = (float*)E->wupload(pos.data.data(), pos.numel() * 4);
  const int Bt = 2 * c.max_batch, T = c.text_len, C = E->text_hidden;
  const float eps = c.text_eps > 0.f ? c.text_eps : 1e-5f;
  E->text_batch = Bt;
  E->text_ids = (int*)E->dmalloc((size_t)Bt * T * 4);
  int x = P.tensor(Bt, T, 1, C);
  E->text_in = x;
  char buf[160];
  for (int l = 0;; ++l) {
    snprintf(buf, sizeof buf, "%sencoder.layers.%d", tm.c_str(), l);
    const std::string p = buf;
    if (!E->has(m, p + ".layer_norm1.weight")) break;
    int h = b.ln(x, make_norm(E, m, p + ".layer_norm1"), eps);
    int qkv = b.conv(h, make_conv_cat(E, m, {p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, true));
    int q = P.view(qkv, 0, C), k = P.view(qkv, C, C), v = P.view(qkv, 2 * C, C);
    int o = b.attn(q, k, v, heads, T, T, -1, /*causal=*/1);
    x = b.conv(o, make_conv(E, m, p + ".self_attn.out_proj", 0), 1, 0, x);
    h = b.ln(x, make_norm(E, m, p + ".layer_norm2"), eps);
    h = b.conv(h, make_conv(E, m, p + ".mlp.fc1", 0));
    h = b.act(h, c.text_act);
    x = b.conv(h, make_conv(E, m, p + ".mlp.fc2", 0), 1, 0, x);
  }
  E->text_out = b.ln(x, make_norm(E, m, tm + "final_layer_norm"), eps, /*keep=*/true);
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

inline int guide_feat_dim(const dd_config& c) {
  return c.guide_feature_dim > 0 ? c.guide_feature_dim : c.guide_planes[c.guide_stages - 1] * c.guide_expansion;
}

// ---------------------------------------------------------------------------------------------------
// sampler drivers
// ---------------------------------------------------------------------------------------------------
struct Run {
  dd_engine* E; hipStream_t s; int B;
  Ctx ctx(const Program& P, char* act) {
    Ctx c; c.act = act; c.grad = E->grad_slab; c.tr = E->tr_slab; c.tr_stride = P.tr_max; c.scratch_partial = E->scratch_partial; c.partial_cap = E->partial_cap;
    c.scratch_tmp = E->scratch_tmp; c.tmp_cap = E->tmp_cap; c.tap1x1 = E->tap1x1; c.gn_scratch = E->gn_scratch; c.rowpart = E->rowpart; c.s = s; c.B = B; c.cross_kv = &E->cross_kv; c.flops = &E->flops; c.prof = &E->prof;
    return c;
  }
};

void check_batch(dd_engine* E, int B) {
  if (B != E->cfg.max_batch) throw std::runtime_error("this engine was built for batch " + std::to_string(E->cfg.max_batch) +
                                                      " (static shapes); got B=" + std::to_string(B));
  if (!E->finalized) throw std::runtime_error("dd_finalize_weights has not been called");
}

// UNet forward on instance k: z fp32 NCHW -> eps2 fp32 NHWC [2B*HW, ld] inside the slab
void unet_fwd(dd_engine* E, int k, const float* z, int step_index, hipStream_t s, bool stash = true) {
  const dd_config& c = E->cfg;
  Run r{E, s, c.max_batch};
  Ctx ctx = r.ctx(E->unet, E->inst[k].unet);
  ctx.step_index = step_index;
  ctx.stash = stash;
  if (c.unet_add_time_dim > 0) {
    if (!E->added_cond_set) throw std::runtime_error("this UNet has text_time additional conditioning: call dd_set_added_cond first");
    ctx.img_bias = 2 * c.max_batch;
  }
  const Tn& in = E->unet.t[E->unet_in];
  HIPCHK(launch_nchw_f32_to_nhwc_bf16(z, act_ptr(ctx, in), c.max_batch, c.unet_in_channels, c.latent_size, c.latent_size, in.ld, in.ld,
                                      in.B == 2 * c.max_batch ? 1 : 0, 1.f, s));
  run_fwd(E->unet, ctx);
}

void vae_fwd(dd_engine* E, int k, const float* x0, hipStream_t s) {
  const dd_config& c = E->cfg;
  Run r{E, s, c.max_batch};
  Ctx ctx = r.ctx(E->vae, E->inst[k].vae);
  const Tn& in = E->vae.t[E->vae_in];
  HIPCHK(launch_nchw_f32_to_nhwc_bf16(x0, act_ptr(ctx, in), c.max_batch, c.vae_latent_channels, c.latent_size, c.latent_size, in.ld,
                                      in.ld, 0, 1.f / c.vae_scaling_factor, s));
  run_fwd(E->vae, ctx);
}

// features of a finished guide forward -> feats [B, D] fp32: global average pool of the last feature map (ResNets, model_utils.py:31-33)
// or the projected class token (ViT)
void guide_features_out(dd_engine* E, const Ctx& gc, float* feats, hipStream_t s, int use_max = 0) {
  const dd_config& c = E->cfg;
  const Tn& f = E->guide.t[E->guide_feat];
  if (c.guide_kind == 1) HIPCHK(hipMemcpy2DAsync(feats, (size_t)f.C * 4, act_f32(gc, f), (size_t)f.ld * 4, (size_t)f.C * 4, f.rows, hipMemcpyDeviceToDevice, s));
  else HIPCHK(launch_gap_f32(act_f32(gc, f), f.ld, feats, nullptr, c.max_batch, f.H * f.W, f.C, use_max, s));
}
// cotangent of the features [B, D] fp32 -> gradient of the guide's output tensor (reverse of guide_features_out)
void guide_features_grad_in(dd_engine* E, const Ctx& gc, const float* gfeat, hipStream_t s) {
  const dd_config& c = E->cfg;
  const Tn& f = E->guide.t[E->guide_feat];
  if (c.guide_kind == 1) {
    if (f.ld != f.C) throw std::runtime_error("ViT guide: feature dim must be a multiple of 8");
    HIPCHK(launch_f32_to_bf16(gfeat, grad_ptr(gc, f), (size_t)f.rows * f.C, s));
  } else {
    HIPCHK(launch_gap_bwd_f32(gfeat, grad_f32(gc, f), f.ld, c.max_batch, f.H * f.W, f.C, nullptr, s));
  }
}

// guide forward from the decoded image of instance k (bicubic -> guide network -> features) -> feat [B, D]
void guide_fwd_from_image(dd_engine* E, int k, hipStream_t s) {
  const dd_config& c = E->cfg;
  Run r{E, s, c.max_batch};
  Ctx gc = r.ctx(E->guide, E->inst[k].guide);
  Ctx vc = r.ctx(E->vae, E->inst[k].vae);
  const Tn& img = E->vae.t[E->vae_out];
  const Tn& gin = E->guide.t[E->guide_in];
  // the decoder's conv_out stores the image in fp32: image, bicubic resize and the whole guide stay fp32
  if (E->image_override) {  // parity tests: the guide (its ReLU / max-pool masks) is evaluated AT the given image, gradients flow as usual
    const size_t per = (size_t)c.max_batch * c.vae_out_channels * img.H * img.W;
    const float* src = E->image_override + per * std::min(k, E->image_override_count - 1);
    HIPCHK(launch_nchw_to_nhwc_f32(src, act_f32(vc, img), c.max_batch, c.vae_out_channels, img.H, img.W, img.ld, img.ld, s));
  }
  HIPCHK(launch_bicubic_f32(act_f32(vc, img), img.ld, act_f32(gc, gin), gin.ld, c.max_batch, img.H, img.W, gin.H, gin.W, 3, gin.ld, s));
  run_fwd(E->guide, gc);
  guide_features_out(E, gc, E->inst[k].feat, s);
}

// reverse of guide_fwd_from_image + vae_fwd: gfeat -> g_x0 (fp32 NCHW)
void guide_vae_bwd(dd_engine* E, int k, float* g_x0, hipStream_t s) {
  const dd_config& c = E->cfg;
  Run r{E, s, c.max_batch};
  Ctx gc = r.ctx(E->guide, E->inst[k].guide);
  Ctx vc = r.ctx(E->vae, E->inst[k].vae);
  const Tn& f = E->guide.t[E->guide_feat];
  // GAP^T; the ReLU mask of the last bottleneck is applied by that conv op's backward
  (void)f;
  guide_features_grad_in(E, gc, E->inst[k].gfeat, s);
  run_bwd(E->guide, gc);
  const Tn& gin = E->guide.t[E->guide_in];
  const Tn& img = E->vae.t[E->vae_out];
  // the guide (fp32) and VAE (bf16) gradient regions are disjoint parts of the shared gradient slab (see finalize)
  HIPCHK(launch_bicubic_bwd_f32(grad_f32(gc, gin), gin.ld, grad_ptr(vc, img), 1, img.ld, c.max_batch, img.H, img.W, gin.H, gin.W, 3, s));
  run_bwd(E->vae, vc);
  const Tn& vin = E->vae.t[E->vae_in];
  HIPCHK(launch_nhwc_to_nchw_f32(grad_ptr(vc, vin), 0, g_x0, c.max_batch, c.vae_latent_channels, c.latent_size, c.latent_size, vin.ld,
                                 1.f / c.vae_scaling_factor, 0.f, 0, 0.f, 0.f, s));
}

// one guided forward step on instance k: z_in -> (z_next, x0, feat) ; energy accumulates into score, writes gfeat
void guided_forward(dd_engine* E, int k, const float* z_in, int step_index, const int* targets, int normalize, float weight,
                    float* score, hipStream_t s) {
  const dd_config& c = E->cfg;
  auto& I = E->inst[k];
  const int HW = c.latent_size * c.latent_size;
  unet_fwd(E, k, z_in, step_index, s);
  const Tn& out = E->unet.t[E->unet_out];
  HIPCHK(launch_cfg_ddim((const float*)(I.unet + out.off), out.ld, z_in, I.z_next, I.x0, c.max_batch, c.unet_out_channels, HW,
                         E->coef_table + (size_t)step_index * 8, s));
  vae_fwd(E, k, I.x0, s);
  guide_fwd_from_image(E, k, s);
  HIPCHK(launch_energy(I.feat, E->Pc, E->Pg, targets, c.max_batch, E->pD, E->pK, E->sp.gs, E->sp.ls, E->sp.use_global, E->sp.use_local,
                       normalize, weight, E->sample_w_set ? E->sample_w : nullptr, score, E->image_scores, I.gfeat, s));
}

// reverse of guided_forward: given g_znext (may be null) returns g_z (fp32 NCHW) in g_z_out
void guided_backward(dd_engine* E, int k, int step_index, const float* g_znext, float* g_z_out, float* g_x0_tmp, hipStream_t s) {
  const dd_config& c = E->cfg;
  const int HW = c.latent_size * c.latent_size;
  Run r{E, s, c.max_batch};
  guide_vae_bwd(E, k, g_x0_tmp, s);
  Ctx uc = r.ctx(E->unet, E->inst[k].unet);
  uc.step_index = step_index;
  const Tn& out = E->unet.t[E->unet_out];
  HIPCHK(launch_cfg_ddim_bwd(g_x0_tmp, g_znext, grad_ptr(uc, out), out.ld, g_z_out, c.max_batch, c.unet_out_channels, HW,
                             E->coef_table + (size_t)step_index * 8, s));
  run_bwd(E->unet, uc);
  const Tn& in = E->unet.t[E->unet_in];
  HIPCHK(launch_dup_bwd(grad_ptr(uc, in), in.ld, g_z_out, c.max_batch, c.unet_in_channels, HW, 1, in.B == 2 * c.max_batch ? 2 : 1, s));
}

void set_config_defaults(dd_config& c) {
  if (c.max_guidance_period < 1) c.max_guidance_period = 1;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
#define DD_TRY(E, ...)                                    \
  try { __VA_ARGS__; return DD_OK; }                      \
  catch (const std::exception& ex) { (E)->err = ex.what(); return DD_ERR_HIP; }

extern "C" {

int dd_create(const dd_config* cfg, dd_engine** out) {
  if (!cfg || !out) return DD_ERR_ARG;
  if (cfg->unet_levels > DD_MAX_LEVELS || cfg->vae_levels > DD_MAX_LEVELS || cfg->guide_stages > DD_MAX_LEVELS || cfg->max_batch < 1)
    return DD_ERR_ARG;
  dd_engine* e = new dd_engine();
  e->cfg = *cfg;
  set_config_defaults(e->cfg);
  *out = e;
  return DD_OK;
}

void dd_destroy(dd_engine* e) {
  if (!e) return;
  for (auto& kv : e->step_graphs) if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
  if (e->gstream) hipStreamDestroy(e->gstream);
  if (e->ev_in) hipEventDestroy(e->ev_in);
  if (e->ev_out) hipEventDestroy(e->ev_out);
  for (void* p : e->dev_allocs) hipFree(p);
  delete e;
}

const char* dd_last_error(dd_engine* e) { return e ? e->err.c_str() : "null engine"; }

int dd_load_tensor(dd_engine* e, const char* model, const char* key, const float* data, int ndim, const int64_t* shape) {
  if (!e || !model || !key || !data || ndim < 1 || ndim > 4) return DD_ERR_ARG;
  if (e->finalized) { e->err = "dd_load_tensor after dd_finalize_weights"; return DD_ERR_STATE; }
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  e->raw[std::string(model) + "/" + key] = std::move(t);
  return DD_OK;
}

int dd_declare_tensor(dd_engine* e, const char* model, const char* key, int ndim, const int64_t* shape) {
  if (!e || !model || !key || ndim < 1 || ndim > 4 || !shape) return DD_ERR_ARG;
  if (e->finalized) { e->err = "dd_declare_tensor after dd_finalize_weights"; return DD_ERR_STATE; }
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  e->raw[std::string(model) + "/" + key] = std::move(t);
  e->declared++;
  return DD_OK;
}

size_t dd_packed_bytes(dd_engine* e) {
  if (!e) return 0;
  size_t n = 0;
  for (auto& p : e->packed) n += rup_sz(p.second, 256);
  return n;
}

// the packed weights as one virtual byte array (each buffer padded to 256 bytes): copy [offset, offset + bytes) to / from `buf`
static int packed_copy(dd_engine* E, char* buf, size_t offset, size_t bytes, bool to_engine, hipStream_t s) {
  DD_TRY(E, {
    if (!E->finalized) throw std::runtime_error("dd_finalize_weights has not been called");
    size_t pos = 0;
    const size_t end = offset + bytes;
    for (auto& p : E->packed) {
      const size_t lo = std::max(pos, offset), hi = std::min(pos + p.second, end);
      if (lo < hi) {
        char* dev = p.first + (lo - pos);
        char* b = buf + (lo - offset);
        HIPCHK(hipMemcpyAsync(to_engine ? dev : b, to_engine ? b : dev, hi - lo, hipMemcpyDeviceToDevice, s));
      }
      pos += rup_sz(p.second, 256);
      if (pos >= end) break;
    }
  });
}
int dd_export_packed(dd_engine* e, void* dst, size_t offset, size_t bytes, void* stream) {
  if (!e || !dst) return DD_ERR_ARG;
  return packed_copy(e, (char*)dst, offset, bytes, false, (hipStream_t)stream);
}
int dd_import_packed(dd_engine* e, const void* src, size_t offset, size_t bytes, void* stream) {
  if (!e || !src) return DD_ERR_ARG;
  return packed_copy(e, (char*)src, offset, bytes, true, (hipStream_t)stream);
}

int dd_finalize_weights(dd_engine* E) {
  if (!E) return DD_ERR_ARG;
  if (E->finalized) { E->err = "already finalized"; return DD_ERR_STATE; }
  if (E->declared && E->declared != (int)E->raw.size()) { E->err = "dd_declare_tensor and dd_load_tensor cannot be mixed"; return DD_ERR_STATE; }
  E->shape_only = E->declared > 0;
  DD_TRY(E, {
    const dd_config& c = E->cfg;
    build_unet(E);
    build_vae(E);
    if (c.guide_kind == 1) build_guide_vit(E);
    else if (c.guide_kind == 2) build_guide_mbv2(E);
    else build_guide(E);
    const bool have_venc = E->has("vae", "encoder.conv_in.weight");
    const bool have_text = E->has("text", "text_model.embeddings.token_embedding.weight");
    if (have_venc) build_vae_encoder(E);
    if (have_text) build_text_encoder(E);
    // time embedding MLP weights (fp32, setup-time only)
    auto up = [&](const char* key) {
      const HostTensor& t = E->get("unet", key);
      return (float*)E->wupload(t.data.data(), t.numel() * 4);
    };
    E->temb_w1 = up("time_embedding.linear_1.weight"); E->temb_b1 = up("time_embedding.linear_1.bias");
    E->temb_w2 = up("time_embedding.linear_2.weight"); E->temb_b2 = up("time_embedding.linear_2.bias");
    if (c.unet_add_time_dim > 0) {
      E->add_w1 = up("add_embedding.linear_1.weight"); E->add_b1 = up("add_embedding.linear_1.bias");
      E->add_w2 = up("add_embedding.linear_2.weight"); E->add_b2 = up("add_embedding.linear_2.bias");
    }
    E->raw.clear();
    // activation slabs
    const int P = c.enable_grad ? c.max_guidance_period : 1;
    const int B = c.max_batch, L = c.latent_size;
    const size_t zbytes = (size_t)B * std::max(c.unet_in_channels, c.vae_latent_channels) * L * L * 4;
    E->inst.resize(P);
    for (int k = 0; k < P; ++k) {
      auto& I = E->inst[k];
      I.unet = (char*)E->dmalloc(E->unet.act_bytes);
      if (k == 0 || c.enable_grad) {
        I.vae = (char*)E->dmalloc(E->vae.act_bytes);
        I.guide = (char*)E->dmalloc(E->guide.act_bytes);
      }
      I.z_in = (float*)E->dmalloc(zbytes); I.z_next = (float*)E->dmalloc(zbytes); I.x0 = (float*)E->dmalloc(zbytes);
      I.feat = (float*)E->dmalloc((size_t)B * guide_feat_dim(c) * 4);
      I.gfeat = (float*)E->dmalloc((size_t)B * guide_feat_dim(c) * 4);
    }
    E->tr_slab = (char*)E->dmalloc(2 * std::max({E->unet.tr_max, E->vae.tr_max, E->guide.tr_max, E->venc.tr_max, E->text.tr_max, (size_t)256}));
    if (c.enable_grad) {
      // UNet gradients alone; VAE and guide gradients live side by side (bicubic^T bridges them)
      const size_t g = std::max(E->unet.grad_bytes, E->vae.grad_bytes + E->guide.grad_bytes);
      E->grad_slab = (char*)E->dmalloc(g);
      for (auto& t : E->guide.t) t.goff += E->vae.grad_bytes;
    }
    // encoder programs run before the loop: the image encoder borrows the decoder slab of instance 0 when it fits
    if (have_venc) E->venc_slab = E->venc.act_bytes <= E->vae.act_bytes ? E->inst[0].vae : (char*)E->dmalloc(E->venc.act_bytes);
    if (have_text) E->text_slab = (char*)E->dmalloc(E->text.act_bytes);
    E->partial_cap = std::max({E->unet.scratch_partial, E->vae.scratch_partial, E->guide.scratch_partial, E->venc.scratch_partial,
                               E->text.scratch_partial, (size_t)1 << 20});
    E->scratch_partial = (char*)E->dmalloc(E->partial_cap, false);
    E->tmp_cap = std::max({E->unet.scratch_tmp, E->vae.scratch_tmp, E->guide.scratch_tmp, E->venc.scratch_tmp, E->text.scratch_tmp, (size_t)256});
    if (!getenv("DD_ATTN_FLASH_ONLY")) {   // A/B switch: keep the flash kernels for wide heads too
      const int tap = (32 << 6) | 32;
      E->tap1x1 = (int*)E->dmalloc(sizeof(int), false);
      HIPCHK(hipMemcpy(E->tap1x1, &tap, sizeof(int), hipMemcpyHostToDevice));
    }
    E->scratch_tmp = (char*)E->dmalloc(E->tmp_cap);
    const int maxG = std::max(c.unet_groups, c.vae_groups);
    E->gn_scratch = (float*)E->dmalloc(groupnorm_scratch_bytes(2 * B, maxG), false);
    E->rowpart = (float*)E->dmalloc(std::max(E->unet.scratch_rowpart, (size_t)256), false);
    for (auto& f : E->f32_tmp) f = (float*)E->dmalloc(zbytes);
    E->img_tmp = (float*)E->dmalloc((size_t)B * 3 * 64 * L * L * 4);
    E->score_tmp = (float*)E->dmalloc(256);
    E->sample_w = (float*)E->dmalloc((size_t)B * 4);
    E->image_scores = (float*)E->dmalloc((size_t)B * 4);
    // cross-attention K/V buffers
    E->ctx_bf16 = (bf16_t*)E->dmalloc((size_t)2 * B * c.text_len * rup(c.unet_cross_dim, 8) * 2);
    for (auto& sl : E->cross_slots) {
      bf16_t* k = (bf16_t*)E->dmalloc((size_t)2 * B * c.text_len * sl.C * 2);
      bf16_t* v = (bf16_t*)E->dmalloc((size_t)2 * B * c.text_len * sl.C * 2);
      E->cross_kv.push_back({k, v});
    }
    HIPCHK(hipDeviceSynchronize());
    E->finalized = true;
  });
}

int dd_set_schedule(dd_engine* E, const int* timesteps, int n, const float* alphas_cumprod, int num_train, float final_alpha,
                    const dd_sampler_params* sp) {
  if (!E || !timesteps || n < 1 || !alphas_cumprod || !sp) return DD_ERR_ARG;
  if (!E->finalized) { E->err = "finalize first"; return DD_ERR_STATE; }
  DD_TRY(E, {
    const dd_config& c = E->cfg;
    E->timesteps.assign(timesteps, timesteps + n);
    E->sp = *sp;
    HIPCHK(hipDeviceSynchronize());
    for (void* q : E->sched_allocs) E->dfree(q);
    E->sched_allocs.clear();
    std::vector<float> coef((size_t)n * 8, 0.f);
    const int ratio = num_train / n;
    for (int i = 0; i < n; ++i) {
      const int t = timesteps[i], prev = t - ratio;
      if (t < 0 || t >= num_train) throw std::runtime_error("timestep out of range");
      const double a = alphas_cumprod[t], ap = prev >= 0 ? alphas_cumprod[prev] : final_alpha;
      float* q = &coef[(size_t)i * 8];
      q[0] = sp->guidance_scale; q[1] = (float)sqrt(a); q[2] = (float)sqrt(1 - a); q[3] = (float)sqrt(ap); q[4] = (float)sqrt(1 - ap);
    }
    E->coef_table = (float*)E->dmalloc(coef.size() * 4, false);
    E->sched_allocs.push_back(E->coef_table);
    HIPCHK(hipMemcpy(E->coef_table, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    // sinusoidal timestep embedding (diffusers Timesteps: flip_sin_to_cos, freq_shift) on host, MLP + projections on GPU (fp32)
    const int C0 = c.unet_block_out_channels[0], half = C0 / 2, TE = C0 * 4;
    std::vector<float> sinus((size_t)n * C0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < half; ++j) {
        const float freq = expf(-logf(10000.f) * (float)j / ((float)half - c.unet_freq_shift));
        const float a = (float)timesteps[i] * freq;
        const float sn = sinf(a), cs = cosf(a);
        if (c.unet_flip_sin_to_cos) { sinus[(size_t)i * C0 + j] = cs; sinus[(size_t)i * C0 + half + j] = sn; }
        else { sinus[(size_t)i * C0 + j] = sn; sinus[(size_t)i * C0 + half + j] = cs; }
      }
    float* d_sin = (float*)E->dmalloc(sinus.size() * 4, false);
    HIPCHK(hipMemcpy(d_sin, sinus.data(), sinus.size() * 4, hipMemcpyHostToDevice));
    float* d_h1 = (float*)E->dmalloc((size_t)n * TE * 4, false);
    float* d_emb = (float*)E->dmalloc((size_t)n * TE * 4, false);
    E->sched_allocs.push_back(d_sin); E->sched_allocs.push_back(d_h1); E->sched_allocs.push_back(d_emb);
    E->d_emb = d_emb;
    E->added_cond_set = false;
    HIPCHK(launch_linear_f32(d_sin, E->temb_w1, E->temb_b1, d_h1, n, TE, C0, 0, nullptr));
    HIPCHK(launch_linear_f32(d_h1, E->temb_w2, E->temb_b2, d_emb, n, TE, TE, 1, nullptr));
    for (ConvW* cw : E->temb_convs) {
      cw->bias_table = (float*)E->dmalloc((size_t)n * cw->Cout * 4, false);
      E->sched_allocs.push_back(cw->bias_table);
      HIPCHK(launch_linear_f32(d_emb, cw->temb_w, cw->temb_b, cw->bias_table, n, cw->Cout, TE, 1, nullptr));
      if (cw->bias) HIPCHK(launch_add_rowvec_f32(cw->bias_table, cw->bias, n, cw->Cout, nullptr));   // + conv1.bias
      if (c.unet_add_time_dim > 0) {
        cw->bias_table_img = (float*)E->dmalloc((size_t)n * 2 * c.max_batch * cw->Cout * 4, false);
        E->sched_allocs.push_back(cw->bias_table_img);
      }
    }
    HIPCHK(hipDeviceSynchronize());
  });
}

int dd_set_prototypes(dd_engine* E, const float* Pc, const float* Pg, int C, int K, int D) {
  if (!E || C < 1 || D < 1) return DD_ERR_ARG;
  DD_TRY(E, {
    E->pC = C; E->pK = K; E->pD = D;
    HIPCHK(hipDeviceSynchronize());
    E->dfree(E->Pc); E->dfree(E->Pg);
    E->Pc = nullptr; E->Pg = nullptr;
    if (Pc) { E->Pc = (float*)E->dmalloc((size_t)C * D * 4, false); HIPCHK(hipMemcpy(E->Pc, Pc, (size_t)C * D * 4, hipMemcpyHostToDevice)); }
    if (Pg) { E->Pg = (float*)E->dmalloc((size_t)C * K * D * 4, false); HIPCHK(hipMemcpy(E->Pg, Pg, (size_t)C * K * D * 4, hipMemcpyHostToDevice)); }
  });
}

int dd_set_prompt(dd_engine* E, const float* embeds, int B, void* stream) {
  if (!E || !embeds) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int rows = 2 * B * c.text_len, ld = rup(c.unet_cross_dim, 8);
    // [rows, cross_dim] fp32 -> bf16 rows (treated as NCHW with H*W = 1 per row)
    HIPCHK(launch_nchw_f32_to_nhwc_bf16(embeds, E->ctx_bf16, rows, c.unet_cross_dim, 1, 1, ld, ld, 0, 1.f, s));
    for (size_t i = 0; i < E->cross_slots.size(); ++i) {
      auto& sl = E->cross_slots[i];
      for (int which = 0; which < 2; ++which) {
        ConvW* w = which ? sl.wv : sl.wk;
        ConvGemmParams p; memset(&p, 0, sizeof p);
        p.alpha = 1.f; p.partial = (float*)E->scratch_partial;
        p.x = E->ctx_bf16; p.x_ld = ld; p.w = w->w_fwd; p.taptab = w->tap_fwd;
        p.y = which ? E->cross_kv[i].second : E->cross_kv[i].first; p.y_ld = sl.C;
        p.B = 1; p.H = rows; p.W = 1; p.Ho = rows; p.Wo = 1; p.stride = 1;
        p.cin = w->sf.cin; p.ntaps = w->sf.ntaps; p.M = rows; p.N = w->sf.N; p.K = w->sf.K;
        HIPCHK(launch_conv_gemm(p, E->partial_cap, s));
      }
    }
  });
}

int dd_set_added_cond(dd_engine* E, const float* text_embeds, const float* time_ids, int B, void* stream) {
  if (!E || !text_embeds || !time_ids) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (c.unet_add_time_dim <= 0) throw std::runtime_error("this UNet has no additional conditioning (unet_add_time_dim == 0)");
    if (!E->d_emb) throw std::runtime_error("dd_set_schedule first");
    hipStream_t s = (hipStream_t)stream;
    const int nb = 2 * B, n = (int)E->timesteps.size(), TE = c.unet_block_out_channels[0] * 4;
    const int Dt = c.unet_add_text_dim, Ds = 6 * c.unet_add_time_dim, Din = Dt + Ds;
    // scratch (setup path: plain allocations, freed at the end)
    float* cat = nullptr; float* h1 = nullptr; float* aug = nullptr; float* X = nullptr;
    HIPCHK(hipMalloc((void**)&cat, (size_t)nb * Din * 4)); HIPCHK(hipMalloc((void**)&h1, (size_t)nb * TE * 4));
    HIPCHK(hipMalloc((void**)&aug, (size_t)nb * TE * 4)); HIPCHK(hipMalloc((void**)&X, (size_t)n * nb * TE * 4));
    // cat[text_embeds, add_time_proj(time_ids).reshape(B, -1)]
    HIPCHK(hipMemcpy2DAsync(cat, (size_t)Din * 4, text_embeds, (size_t)Dt * 4, (size_t)Dt * 4, nb, hipMemcpyDeviceToDevice, s));
    for (int k = 0; k < 6; ++k) {
      // time id k of every image -> columns [Dt + k*dim, Dt + (k+1)*dim): gather the k-th id with a strided copy first
      HIPCHK(hipMemcpy2DAsync(h1, 4, time_ids + k, 6 * 4, 4, nb, hipMemcpyDeviceToDevice, s));
      HIPCHK(launch_sinusoid_f32(h1, cat, nb, c.unet_add_time_dim, Din, Dt + k * c.unet_add_time_dim, 1, s));
    }
    HIPCHK(launch_linear_f32(cat, E->add_w1, E->add_b1, h1, nb, TE, Din, 0, s));
    HIPCHK(launch_linear_f32(h1, E->add_w2, E->add_b2, aug, nb, TE, TE, 1, s));
    HIPCHK(launch_add_outer_f32(E->d_emb, aug, X, n, nb, TE, s));                 // emb[step, image] = time_embedding(t) + aug_emb
    for (ConvW* cw : E->temb_convs) {
      HIPCHK(launch_linear_f32(X, cw->temb_w, cw->temb_b, cw->bias_table_img, n * nb, cw->Cout, TE, 1, s));
      if (cw->bias) HIPCHK(launch_add_rowvec_f32(cw->bias_table_img, cw->bias, n * nb, cw->Cout, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    hipFree(cat); hipFree(h1); hipFree(aug); hipFree(X);
    E->added_cond_set = true;
  });
}

int dd_add_noise(dd_engine* E, const float* x, const float* noise, float* out, int B, int step_index, void* stream) {
  if (!E || !x || !noise || !out) return DD_ERR_ARG;
  DD_TRY(E, {
    if (step_index < 0 || step_index >= (int)E->timesteps.size()) throw std::runtime_error("step_index out of range");
    const dd_config& c = E->cfg;
    // coef_table[step][1..2] = sqrt(a_t), sqrt(1-a_t)
    HIPCHK(launch_axpby(x, noise, out, (size_t)B * c.unet_in_channels * c.latent_size * c.latent_size,
                        E->coef_table + (size_t)step_index * 8 + 1, (hipStream_t)stream));
  });
}

int dd_unet_forward(dd_engine* E, const float* z, int step_index, float* eps2_out, int B, void* stream) {
  if (!E || !z || !eps2_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    unet_fwd(E, 0, z, step_index, s);
    const Tn& out = E->unet.t[E->unet_out];
    HIPCHK(launch_nhwc_to_nchw_f32(E->inst[0].unet + out.off, 1, eps2_out, 2 * B, c.unet_out_channels, c.latent_size, c.latent_size, out.ld,
                                   1.f, 0.f, 0, 0.f, 0.f, s));
  });
}

// the launch sequence of one plain step: UNet forward (no stash) + CFG + DDIM
static void denoise_step_enqueue(dd_engine* E, const float* z, int step_index, float* z_prev_out, float* x0_out, hipStream_t s) {
  const dd_config& c = E->cfg;
  unet_fwd(E, 0, z, step_index, s, /*stash=*/false);
  const Tn& out = E->unet.t[E->unet_out];
  HIPCHK(launch_cfg_ddim((const float*)(E->inst[0].unet + out.off), out.ld, z, z_prev_out, x0_out, c.max_batch, c.unet_out_channels,
                         c.latent_size * c.latent_size, E->coef_table + (size_t)step_index * 8, s));
}

int dd_denoise_step(dd_engine* E, const float* z, int step_index, float* z_prev_out, float* x0_out, int B, void* stream) {
  if (!E || !z || !z_prev_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    if (step_index < 0 || step_index >= (int)E->timesteps.size()) throw std::runtime_error("step_index out of range");
    hipStream_t s = (hipStream_t)stream;
    // Opt-in (DD_GRAPH=1): measured on MI355X the replay does not pay -- B = 16: 1485 vs 1481 ms per batch, B = 1: 313 vs 302 ms
    // (DESIGN.md section 6): even at B = 1 the ~6300 launches of an image are GPU-bound small kernels, not launch-bound.
    static const bool use_graph = getenv("DD_GRAPH") != nullptr && atoi(getenv("DD_GRAPH")) != 0;
    if (!use_graph || !E->graphs_ok || E->prof.on) { denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, s); return DD_OK; }
    // first sighting of a (step, buffers) key: plain launches (also runs every one-time hipFuncSetAttribute); second: capture on the
    // engine's own stream (the caller's may be the legacy default stream, which cannot be captured) and instantiate; then replay
    char key[96];
    snprintf(key, sizeof key, "%d/%p/%p/%p", step_index, (const void*)z, (void*)z_prev_out, (void*)x0_out);
    auto it = E->step_graphs.find(key);
    if (it == E->step_graphs.end()) {
      if (E->step_graphs.size() >= 256) { denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, s); return DD_OK; }
      dd_engine::StepGraph g;
      const double f0 = E->flops;
      denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, s);
      g.flops = E->flops - f0; g.seen = 1;
      E->step_graphs[key] = g;
      return DD_OK;
    }
    dd_engine::StepGraph& g = it->second;
    if (!E->gstream) {
      HIPCHK(hipStreamCreateWithFlags(&E->gstream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&E->ev_in, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&E->ev_out, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(E->ev_in, s));
    HIPCHK(hipStreamWaitEvent(E->gstream, E->ev_in, 0));
    if (!g.exec) {
      hipGraph_t graph = nullptr;
      const double f0 = E->flops;
      bool ok = hipStreamBeginCapture(E->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        try { denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, E->gstream); } catch (...) { ok = false; }
        ok = (hipStreamEndCapture(E->gstream, &graph) == hipSuccess) && ok && graph;
      }
      E->flops = f0;
      if (ok) ok = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) hipGraphDestroy(graph);
      if (!ok) {   // capture is an optimisation only: fall back to plain launches for good
        (void)hipGetLastError();
        g.exec = nullptr; E->graphs_ok = false;
        denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, s);
        return DD_OK;
      }
    }
    HIPCHK(hipGraphLaunch(g.exec, E->gstream));
    E->flops += g.flops;
    HIPCHK(hipEventRecord(E->ev_out, E->gstream));
    HIPCHK(hipStreamWaitEvent(s, E->ev_out, 0));
  });
}

int dd_decode(dd_engine* E, const float* z, float* image_out, int denormalize, int B, void* stream) {
  if (!E || !z || !image_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    vae_fwd(E, 0, z, s);
    const Tn& img = E->vae.t[E->vae_out];
    HIPCHK(launch_nhwc_to_nchw_f32(E->inst[0].vae + img.off, 1, image_out, B, c.vae_out_channels, img.H, img.W, img.ld,
                                   denormalize ? 0.5f : 1.f, denormalize ? 0.5f : 0.f, denormalize, 0.f, 1.f, s));
  });
}

int dd_vae_encode(dd_engine* E, const float* images, const float* noise, float* latents_out, float* moments_out, int B, void* stream) {
  if (!E || !images || !latents_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    if (!E->venc_slab) throw std::runtime_error("no VAE encoder weights were loaded (vae/encoder.* keys)");
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    Run r{E, s, B};
    Ctx ctx = r.ctx(E->venc, E->venc_slab);
    ctx.stash = false;
    const Tn& in = E->venc.t[E->venc_in];
    HIPCHK(launch_nchw_f32_to_nhwc_bf16(images, act_ptr(ctx, in), B, c.vae_out_channels, in.H, in.W, in.ld, in.ld, 0, 1.f, s));
    run_fwd(E->venc, ctx);
    const Tn& mo = E->venc.t[E->venc_out];
    HIPCHK(launch_vae_sample((const float*)(ctx.act + mo.off), mo.ld, noise, latents_out, moments_out, B, c.vae_latent_channels,
                             mo.H * mo.W, c.vae_scaling_factor, s));
  });
}

int dd_text_encode(dd_engine* E, const int* input_ids, float* embeds_out, int n, void* stream) {
  if (!E || !input_ids || !embeds_out) return DD_ERR_ARG;
  DD_TRY(E, {
    if (!E->finalized) throw std::runtime_error("dd_finalize_weights has not been called");
    if (!E->text_slab) throw std::runtime_error("no text encoder weights were loaded (text/text_model.* keys)");
    if (n < 1 || n > E->text_batch) throw std::runtime_error("dd_text_encode: n must be in [1, 2*max_batch]");
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int T = c.text_len, Cc = E->text_hidden;
    HIPCHK(hipMemsetAsync(E->text_ids, 0, (size_t)E->text_batch * T * 4, s));
    HIPCHK(hipMemcpyAsync(E->text_ids, input_ids, (size_t)n * T * 4, hipMemcpyDeviceToDevice, s));
    Run r{E, s, E->text_batch};
    Ctx ctx = r.ctx(E->text, E->text_slab);
    ctx.stash = false;
    const Tn& in = E->text.t[E->text_in];
    HIPCHK(launch_clip_embed(E->text_ids, E->tok_emb, E->pos_emb, act_ptr(ctx, in), in.ld, in.rows, T, Cc, E->text_vocab, s));
    run_fwd(E->text, ctx);
    const Tn& o = E->text.t[E->text_out];
    HIPCHK(launch_rows_bf16_to_f32(act_ptr(ctx, o), o.ld, embeds_out, n * T, Cc, s));
  });
}

int dd_guide_encode_pooled(dd_engine* E, const float* images, float* feats, int B, int use_max, void* stream);
int dd_guide_encode(dd_engine* E, const float* images, float* feats, int B, void* stream) {
  return dd_guide_encode_pooled(E, images, feats, B, 0, stream);
}
int dd_guide_encode_pooled(dd_engine* E, const float* images, float* feats, int B, int use_max, void* stream) {
  if (!E || !images || !feats) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    if (use_max && E->cfg.guide_kind == 1) throw std::runtime_error("the ViT guide has no spatial pooling (encode_image = the projected class token)");
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    Run r{E, s, B};
    Ctx gc = r.ctx(E->guide, E->inst[0].guide);
    const Tn& gin = E->guide.t[E->guide_in];
    HIPCHK(launch_nchw_to_nhwc_f32(images, act_f32(gc, gin), B, 3, gin.H, gin.W, gin.ld, gin.ld, s));
    run_fwd(E->guide, gc);
    guide_features_out(E, gc, feats, s, use_max);
  });
}

int dd_transform_guidance(dd_engine* E, const float* z, const int* targets, const float* ch_e, const float* ch_b, int first_step_index,
                          int P, float* z_out, float* score_out, float* grad_eb_out, int B, void* stream) {
  if (!E || !z || !targets || !ch_e || !ch_b || !z_out || P < 1) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    if (P > (int)E->inst.size()) throw std::runtime_error("guidance period exceeds max_guidance_period");
    if (first_step_index < 0 || first_step_index + P > (int)E->timesteps.size()) throw std::runtime_error("guide steps out of range");
    if ((E->sp.use_global && !E->Pc) || (E->sp.use_local && !E->Pg)) throw std::runtime_error("prototypes not set");
    hipStream_t s = (hipStream_t)stream;
    const int BC = B * c.unet_in_channels, HW = c.latent_size * c.latent_size;
    float* score = score_out ? score_out : E->score_tmp;
    HIPCHK(hipMemsetAsync(score, 0, sizeof(float), s));
    HIPCHK(hipMemsetAsync(E->image_scores, 0, (size_t)B * sizeof(float), s));
    // z0 = z*(1+e)+b  (generate_data.py:696)
    HIPCHK(launch_affine(z, ch_e, ch_b, E->inst[0].z_in, BC, HW, s));
    const float weight = 1.f / (float)E->sp.guidance_period;   // score / args.guidance_period (:719)
    for (int k = 0; k < P; ++k) {
      const float* zin = E->inst[k].z_in;
      guided_forward(E, k, zin, first_step_index + k, targets, 0, weight, score, s);
      if (k + 1 < P) HIPCHK(hipMemcpyAsync(E->inst[k + 1].z_in, E->inst[k].z_next, (size_t)BC * HW * 4, hipMemcpyDeviceToDevice, s));
    }
    float* g_next = nullptr;
    float* gbuf[2] = {E->f32_tmp[0], E->f32_tmp[1]};
    for (int k = P - 1; k >= 0; --k) {
      float* g_z = gbuf[k & 1];
      guided_backward(E, k, first_step_index + k, g_next, g_z, E->f32_tmp[2], s);
      g_next = g_z;
    }
    if (grad_eb_out) HIPCHK(hipMemcpyAsync(grad_eb_out, g_next, (size_t)BC * HW * 4, hipMemcpyDeviceToDevice, s));
    // e -= rho*ge ; b -= rho*gb ; z' = clamp(z*(1+e)+b, z-c, z+c)  (:721-728)
    HIPCHK(launch_transform_update(z, g_next, ch_e, ch_b, z_out, BC, HW, E->sp.rho, E->sp.constraint_value, s));
  });
}

int dd_direct_guidance(dd_engine* E, const float* z, const int* targets, int step_index, float* z_next_out, float* x0_out,
                       float* score_out, float* grad_z_out, int B, void* stream) {
  if (!E || !z || !targets || !z_next_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    if (step_index < 0 || step_index >= (int)E->timesteps.size()) throw std::runtime_error("step_index out of range");
    if ((E->sp.use_global && !E->Pc) || (E->sp.use_local && !E->Pg)) throw std::runtime_error("prototypes not set");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * c.unet_in_channels * c.latent_size * c.latent_size;
    float* score = score_out ? score_out : E->score_tmp;
    HIPCHK(hipMemsetAsync(score, 0, sizeof(float), s));
    HIPCHK(hipMemsetAsync(E->image_scores, 0, (size_t)B * sizeof(float), s));
    HIPCHK(hipMemcpyAsync(E->inst[0].z_in, z, n * 4, hipMemcpyDeviceToDevice, s));
    guided_forward(E, 0, E->inst[0].z_in, step_index, targets, 1, 1.f, score, s);
    float* g_z = E->f32_tmp[0];
    guided_backward(E, 0, step_index, nullptr, g_z, E->f32_tmp[2], s);
    if (grad_z_out) HIPCHK(hipMemcpyAsync(grad_z_out, g_z, n * 4, hipMemcpyDeviceToDevice, s));
    if (x0_out) HIPCHK(hipMemcpyAsync(x0_out, E->inst[0].x0, n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(launch_sub_scaled(E->inst[0].z_next, g_z, z_next_out, n, E->sp.rho, s));   // :762
  });
}

int dd_expand(dd_engine* E, const dd_expand_args* a, void* stream) {
  if (!E || !a || !a->image_latents || !a->noise || !a->z_out) return DD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)E->timesteps.size();
  if (a->start_index < 0 || a->start_index >= n) { E->err = "start_index out of range"; return DD_ERR_ARG; }
  int rc = dd_add_noise(E, a->image_latents, a->noise, E->f32_tmp[3], a->B, a->start_index, stream);
  if (rc) return rc;
  float* cur = E->f32_tmp[3];
  float* nxt = E->f32_tmp[4];
  for (int i = a->start_index; i < n; ++i) {
    if (a->guidance_type == 1 && a->guide_count > 0 && i == a->guide_first) {
      // transform guidance at t == guide_timesteps[0], then the step is executed again from the corrected latent (:1203-1207)
      rc = dd_transform_guidance(E, cur, a->targets, a->e, a->b, a->guide_first, a->guide_count, nxt, a->score_out, nullptr, a->B, stream);
      if (rc) return rc;
      std::swap(cur, nxt);
      rc = dd_denoise_step(E, cur, i, nxt, nullptr, a->B, stream);
    } else if (a->guidance_type == 2 && i >= a->guide_first && i < a->guide_first + a->guide_count) {
      rc = dd_direct_guidance(E, cur, a->targets, i, nxt, nullptr, a->score_out, nullptr, a->B, stream);
    } else {
      rc = dd_denoise_step(E, cur, i, nxt, nullptr, a->B, stream);
    }
    if (rc) return rc;
    std::swap(cur, nxt);
  }
  try {
    const dd_config& c = E->cfg;
    HIPCHK(hipMemcpyAsync(a->z_out, cur, (size_t)a->B * c.unet_in_channels * c.latent_size * c.latent_size * 4, hipMemcpyDeviceToDevice, s));
  } catch (const std::exception& ex) { E->err = ex.what(); return DD_ERR_HIP; }
  if (a->image_out) return dd_decode(E, cur, a->image_out, 1, a->B, stream);
  return DD_OK;
}

int dd_set_sample_weights(dd_engine* E, const float* w_host, int B) {
  if (!E) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    E->sample_w_set = w_host != nullptr;
    if (w_host) {
      HIPCHK(hipDeviceSynchronize());   // a previous guidance call may still be reading the weights
      HIPCHK(hipMemcpy(E->sample_w, w_host, (size_t)B * sizeof(float), hipMemcpyHostToDevice));
    }
  });
}

int dd_get_image_scores(dd_engine* E, float* scores_out, int B, void* stream) {
  if (!E || !scores_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    HIPCHK(hipMemcpyAsync(scores_out, E->image_scores, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  });
}

// ---- per-module VJP diagnostics (parity tests of the hand-derived reverse programs against torch.autograd) ----
int dd_unet_vjp(dd_engine* E, const float* z, int step_index, const float* g_eps2, float* g_z_out, int B, void* stream) {
  if (!E || !z || !g_eps2 || !g_z_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    hipStream_t s = (hipStream_t)stream;
    const int HW = c.latent_size * c.latent_size;
    unet_fwd(E, 0, z, step_index, s);
    Run r{E, s, B};
    Ctx uc = r.ctx(E->unet, E->inst[0].unet);
    uc.step_index = step_index;
    const Tn& out = E->unet.t[E->unet_out];
    HIPCHK(launch_nchw_f32_to_nhwc_bf16(g_eps2, grad_ptr(uc, out), 2 * B, c.unet_out_channels, c.latent_size, c.latent_size, out.ld, out.ld, 0,
                                        1.f, s));
    run_bwd(E->unet, uc);
    const Tn& in = E->unet.t[E->unet_in];
    HIPCHK(launch_dup_bwd(grad_ptr(uc, in), in.ld, g_z_out, B, c.unet_in_channels, HW, 0, in.B == 2 * B ? 2 : 1, s));
  });
}

int dd_decode_vjp(dd_engine* E, const float* z, const float* g_image, float* g_z_out, int B, void* stream) {
  if (!E || !z || !g_image || !g_z_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    hipStream_t s = (hipStream_t)stream;
    vae_fwd(E, 0, z, s);
    Run r{E, s, B};
    Ctx vc = r.ctx(E->vae, E->inst[0].vae);
    const Tn& img = E->vae.t[E->vae_out];
    HIPCHK(launch_nchw_f32_to_nhwc_bf16(g_image, grad_ptr(vc, img), B, c.vae_out_channels, img.H, img.W, img.ld, img.ld, 0, 1.f, s));
    run_bwd(E->vae, vc);
    const Tn& vin = E->vae.t[E->vae_in];
    HIPCHK(launch_nhwc_to_nchw_f32(grad_ptr(vc, vin), 0, g_z_out, B, c.vae_latent_channels, c.latent_size, c.latent_size, vin.ld,
                                   1.f / c.vae_scaling_factor, 0.f, 0, 0.f, 0.f, s));
  });
}

int dd_guide_vjp(dd_engine* E, const float* images, const float* g_feats, float* g_images_out, int B, void* stream) {
  if (!E || !images || !g_feats || !g_images_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    hipStream_t s = (hipStream_t)stream;
    Run r{E, s, B};
    Ctx gc = r.ctx(E->guide, E->inst[0].guide);
    const Tn& gin = E->guide.t[E->guide_in];
    HIPCHK(launch_nchw_to_nhwc_f32(images, act_f32(gc, gin), B, 3, gin.H, gin.W, gin.ld, gin.ld, s));
    run_fwd(E->guide, gc);
    guide_features_grad_in(E, gc, g_feats, s);
    run_bwd(E->guide, gc);
    HIPCHK(launch_nhwc_to_nchw_f32(grad_f32(gc, gin), 1, g_images_out, B, 3, gin.H, gin.W, gin.ld, 1.f, 0.f, 0, 0.f, 0.f, s));
  });
}

int dd_profile_enable(dd_engine* E, int on) {
  if (!E) return DD_ERR_ARG;
  E->prof.on = on != 0;
  E->prof.used = 0;
  E->prof.chain = false;
  E->prof.recs.clear();
  return DD_OK;
}
// out[fam*3 + {0,1,2}] = {total ms, algorithmic flops, launches(op count)} for fam in {conv_gemm, attention, norm, other}
int dd_profile_read(dd_engine* E, double* out12) {
  if (!E || !out12) return DD_ERR_ARG;
  DD_TRY(E, {
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < 12; ++i) out12[i] = 0;
    FILE* dump = getenv("DD_PROFILE_DUMP") ? fopen(getenv("DD_PROFILE_DUMP"), "w") : nullptr;
    if (dump) fprintf(dump, "fam,bwd,M,N,K,flops,ms\n");
    for (auto& r : E->prof.recs) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, E->prof.pool[r.e0], E->prof.pool[r.e1]));
      if (dump) fprintf(dump, "%d,%d,%d,%d,%d,%.0f,%.5f\n", r.fam, r.bwd, r.M, r.N, r.K, r.flops, ms);
      out12[r.fam * 3 + 0] += ms; out12[r.fam * 3 + 1] += r.flops; out12[r.fam * 3 + 2] += 1;
    }
    if (dump) fclose(dump);
    E->prof.used = 0; E->prof.chain = false; E->prof.recs.clear();
  });
}

// ---- debug introspection: copy activation / gradient of tensor `idx` of program `prog` (0 unet, 1 vae, 2 guide; + 16 * instance)
// to a HOST fp32 buffer [rows*ld]; info4 = {rows, C, ld, is_f32}. Synchronises.
int dd_debug_tensor(dd_engine* E, int prog_inst, int idx, int want_grad, float* host_out, int* info4) {
  const int prog = prog_inst & 15, k = prog_inst >> 4;   // bits 4.. select the instance (chained guided step) of an activation
  if (!E || prog < 0 || prog > 2 || k < 0 || k >= (int)E->inst.size()) return DD_ERR_ARG;
  DD_TRY(E, {
    Program& P = prog == 0 ? E->unet : prog == 1 ? E->vae : E->guide;
    if (idx < 0) idx += (int)P.t.size();   // negative: from the end (-1 = the program's output tensor)
    if (idx < 0 || idx >= (int)P.t.size()) throw std::runtime_error("tensor index out of range");
    const Tn& t = P.t[idx];
    if (info4) { info4[0] = t.rows; info4[1] = t.C; info4[2] = t.ld; info4[3] = t.f32 ? 1 : 0; }
    if (!host_out) return DD_OK;
    HIPCHK(hipDeviceSynchronize());
    const dd_engine::Inst& I = E->inst[k];
    char* slab = prog == 0 ? I.unet : prog == 1 ? I.vae : I.guide;
    if (!want_grad && !slab) throw std::runtime_error("this instance has no slab for that program");
    char* base = want_grad ? E->grad_slab + t.goff : t.transient ? E->tr_slab + (size_t)t.tr_slot * P.tr_max : slab + t.off;
    const size_t n = (size_t)t.rows * t.ld;
    if ((t.f32 && !want_grad) || (want_grad && t.gf32)) { HIPCHK(hipMemcpy(host_out, base, n * 4, hipMemcpyDeviceToHost)); }
    else {
      std::vector<bf16_t> tmp(n);
      HIPCHK(hipMemcpy(tmp.data(), base, n * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) host_out[i] = host_bf2f(tmp[i]);
    }
  });
}
// parity-test hook: image DEVICE fp32 [B,3,8L,8L] (not denormalised) replaces the decoder's output in front of the bicubic resize of
// every later guided forward (the gradient still flows through the decoder); NULL switches it off.  The caller keeps the buffer alive.
int dd_debug_set_images(dd_engine* E, const float* images, int count) {
  if (!E || count < 0) return DD_ERR_ARG;
  E->image_override = count > 0 ? images : nullptr;
  E->image_override_count = E->image_override ? count : 0;
  return DD_OK;
}
int dd_debug_set_image(dd_engine* E, const float* image) { return dd_debug_set_images(E, image, 1); }
int dd_debug_num_tensors(dd_engine* E, int prog) {
  if (!E || prog < 0 || prog > 2) return DD_ERR_ARG;
  return (int)(prog == 0 ? E->unet : prog == 1 ? E->vae : E->guide).t.size();
}

// output stage (generate_data.py:1227-1234): [B,3,H,W] fp32 in [0,1] -> uint8 HWC with torchvision.save_image's quantisation
int dd_image_to_u8(dd_engine* E, const float* image, uint8_t* out_hwc, int B, void* stream) {
  if (!E || !image || !out_hwc || B < 1) return DD_ERR_ARG;
  DD_TRY(E, {
    const dd_config& c = E->cfg;
    HIPCHK(launch_to_uint8(image, out_hwc, B, c.vae_out_channels, 8 * c.latent_size, 8 * c.latent_size, (hipStream_t)stream));
  });
}

size_t dd_workspace_bytes(dd_engine* e) { return e ? e->total_bytes : 0; }
double dd_flops_last(dd_engine* e) { if (!e) return 0; const double f = e->flops; e->flops = 0; return f; }

}  // extern "C"
