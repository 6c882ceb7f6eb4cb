// Internal types of the expansion engine (engine*.cpp): weights, the static op graph, execution context, the engine object.
// engine_weights.cpp builds packed weights, engine_graph.cpp the programs and their backward / fusion plans, engine_exec.cpp runs
// them, engine.cpp holds the sampler drivers and the C ABI of include/distdiff_hip.h.
#pragma once
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

struct dd_engine;

namespace ddi {

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
  // Interval of op indices in which the reverse run may touch this tensor's gradient (plan_grad_memory: the interval of its buffer
  // class; -1 / nops = also outside run_bwd: program inputs / outputs, padded tensors).  Defaults: always.  DD_GRAD_CHECK=1 makes every
  // gradient access of run_bwd verify it (a fused backward that wrote g(x) outside [producer, last consumer] would corrupt another
  // tensor's gradient silently once the slab is packed by liveness).
  int glo = -(1 << 30), ghi = 1 << 30;
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
  int pv_fp8 = 0;              // OP_ATTN: P.V on the fp8 MFMA (dd_config.unet_attn_fp8; d = 64 heads of the UNet only)
  int q_prescaled = 0;         // OP_ATTN: the to_q weights carry 1/sqrt(D) * log2(e) (attn_prescale(); the kernels then get scale = ln 2)
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
  // GroupNorm(+SiLU) applied by the 3x3 convolution that follows (plan_gn_fold; decided per run, CF_GNFOLD): OP_GN.gn_into = that
  // convolution's op index, OP_CONV.gn_from = the GroupNorm's
  int gn_into = -1, gn_from = -1;
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
  mutable std::vector<char> gn_folded; // per op, per forward run: this GroupNorm only produced its affine, the next convolution applies it
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
  float* gn_coef = nullptr;      // GroupNorm affine [B][C][2] of the op that ran last (CF_GNFOLD)
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
extern bool g_grad_check;                       // DD_GRAD_CHECK=1
void grad_access_check(const Tn& t);            // throws when run_bwd touches t's gradient outside [t.glo, t.ghi]
inline bf16_t* grad_ptr(const Ctx& c, const Tn& t) { if (g_grad_check) grad_access_check(t); return (bf16_t*)(c.grad + t.goff); }
inline float* act_f32(const Ctx& c, const Tn& t) { return (float*)act_raw(c, t); }
inline float* grad_f32(const Ctx& c, const Tn& t) { if (g_grad_check) grad_access_check(t); return (float*)(c.grad + t.goff); }

}  // namespace ddi

using namespace ddi;

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
  Program venc, text, text2;
  int venc_in = -1, venc_out = -1, text_in = -1, text_out = -1;
  char* venc_slab = nullptr; char* text_slab = nullptr;
  float* tok_emb = nullptr; float* pos_emb = nullptr; int text_vocab = 0, text_hidden = 0, text_batch = 0;
  int* text_ids = nullptr;
  // second tower of a two-tower model (SDXL): its own program / slab, plus the pooled head (final LayerNorm output + text_projection)
  int text2_in = -1, text2_out = -1, text2_final = -1;
  char* text2_slab = nullptr;
  float* tok_emb2 = nullptr; float* pos_emb2 = nullptr; int text2_vocab = 0, text2_hidden = 0, text2_proj = 0;
  float* text2_proj_w = nullptr;                      // [proj][hidden] fp32
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
  float* gn_coef = nullptr;
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

namespace ddi {

inline int guide_feat_dim_decl(const dd_config& c) {
  return c.guide_feature_dim > 0 ? c.guide_feature_dim : c.guide_planes[c.guide_stages - 1] * c.guide_expansion;
}

inline int guide_feat_dim(const dd_config& c) { return guide_feat_dim_decl(c); }
}  // namespace ddi

namespace ddi {
// engine_weights.cpp
ConvW* make_conv_f32(dd_engine* E, const float* w, const float* bias, int Cout, int Cin, int KH, int KW, int pad, int groups, bool need_bwd);
// qrows / qscale: output rows [0, qrows) (weights and bias, forward and input-gradient packings) are multiplied by qscale before the bf16
// rounding -- the attention query projection carrying 1/sqrt(D) * log2(e) (attn_prescale)
ConvW* make_conv_raw(dd_engine* E, const float* w, const float* bias, bool has_bias, int Cout, int Cin, int KH, int KW, int pad, bool geglu,
                     bool need_bwd, bool fold = false, const float* ln_gamma = nullptr, const float* ln_beta = nullptr, int qrows = 0,
                     float qscale = 1.f);
ConvW* make_conv(dd_engine* E, const std::string& model, const std::string& prefix, int pad, bool geglu = false, bool has_bias = true,
                 const std::string& ln = "", int qrows = 0, float qscale = 1.f);
ConvW* make_conv_cat(dd_engine* E, const std::string& model, const std::vector<std::string>& prefixes, bool with_bias, const std::string& ln = "",
                     int qrows = 0, float qscale = 1.f);
bool attn_prescale();       // fold the softmax scale into the to_q weights of the UNet's transformer blocks (DD_ATTN_PRESCALE=0: off)
ConvW* make_conv_bn(dd_engine* E, const std::string& model, const std::string& conv, const std::string& bn, int pad, float eps, int cin_total);
NormW* make_norm(dd_engine* E, const std::string& model, const std::string& prefix);
bool ln_fold_enabled();
// engine_graph.cpp
void build_unet(dd_engine* E);
void build_vae(dd_engine* E);
void build_vae_encoder(dd_engine* E);
void build_text_encoder(dd_engine* E, int which = 0);
void build_guide(dd_engine* E);
void build_guide_mbv2(dd_engine* E);
void build_guide_vit(dd_engine* E);
// engine_exec.cpp
void run_fwd(const Program& P, const Ctx& c, int op_begin = 0, int op_end = -1);
void run_bwd(const Program& P, const Ctx& c);
}  // namespace ddi
