// Whole-network plan for the Inception-v1 I3D of the reference
// (video_features_pytorch/models/I3D_doubled.py:149-380, KTH head of
// I3D_doubled_kth.py:302-308): explicit forward and backward-DATA launch lists over
// caller-owned arenas, the perturbation-mask search loop
// (FindMasksComparison_I3D_smth.py:193-214) and Grad-CAM (grad_cam_videos.py:64-142)
// driven entirely from the host side of this library with no device sync.
//
// No autograd: only dL/d(input) is propagated, activations are kept
// once as the ReLU gates, Inception branches write straight into their slice of the
// concatenated output (no torch.cat copy).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ivf_common.h"

namespace ivf {

struct ActBuf {
  int T, H, W, C;     // channels-last [B,T,H,W,C]
  bool relu_out;      // produced by Unit3D ReLU (or a concat of such)
  size_t act_off = 0, grad_off = 0;   // BYTE offsets into the workspace
  int esz = 4;                        // bytes per stored element: 4, or 2 (bf16 storage, IVF_MATH_BF16ACT; never the clip)
  // 1-bit ReLU gate record (1 byte per 8 channels, written by the forward epilogues of the producers): kept for
  // buffers whose gradient a convolution gates, so the backward reads 1/32 of the bytes of the fp32 activation
  bool need_gate = false;
  size_t gate_off = 0;                // byte offset into the workspace
  int consumers = 0;
  std::string name;
  size_t per_clip() const { return (size_t)T * H * W * C; }
};

struct ConvLayer {
  std::string name;
  int cin, cinp, cout;
  int k[3], s[3];
  bool has_bn;
  size_t wf_off = 0, wb_off = 0, scale_off = 0, shift_off = 0;  // float offsets in weights arena
  size_t wf_elems = 0, wb_elems = 0;
  ivf_conv3d_bwd_geom geom{};
  int pad[3] = {0, 0, 0};   // forward front pads for the planned input size
  // Inception 1x1x1 fusion (b0 / b1a / b2a share their input):
  //  * b1a and b2a run forward as ONE conv writing the [b1a | b2a] buffer: the group's packed
  //    weights / scale / shift live with b1a (grp_*), b2a's rows start at grp_row;
  //  * all three run backward as ONE GEMM over [dY_b0 | dT_b1a | dT_b2a]: the packed matrix
  //    lives with b0 (fus_*), each unit's columns start at fus_koff.
  int grp_owner = -1, grp_row = 0, grp_rows = 0;
  size_t grp_wf_off = 0, grp_scale_off = 0, grp_shift_off = 0, grp_wf_elems = 0;
  int fus_owner = -1, fus_koff = 0, fus_ktotal = 0;
  size_t fus_wb_off = 0, fus_wb_elems = 0;
};

struct Op {
  enum Type { CONV, POOL } type;
  int src, dst;
  int src_coff, cin;   // channel window read from src
  int dst_coff, cout;
  int conv = -1;
  int k[3], s[3], p[3];
  size_t idx_off = 0;  // pool arg-max bytes offset (in bytes, inside workspace)
  int var_fwd = IVF_CONV_AUTO, var_bwd = IVF_CONV_AUTO;   // tuned kernel variants
  bool fwd_group = false;   // forward: the fused b0|b1a|b2a GEMM (weights: convs[conv].grp_*): columns [0, n0) -> dst,
                            // columns [n0, cout) -> dst2 (one read of the module input for the three 1x1x1 units)
  bool fwd_skip = false;    // forward: nothing to do here (b0: covered by the fused GEMM)
  int dst2 = -1, n0 = 0;
  bool bwd_skip = false;    // backward: nothing to do here (covered by the fused GEMM below)
  bool bwd_fused = false;   // backward: the fused b0+b1a+b2a GEMM; second gradient source = src2
  int src2 = -1;
  double flops_bwd_per_clip = 0.0;
  double flops_per_clip = 0.0;  // algorithmic: 2 * out positions * Cout * taps * REAL Cin (same for bwd-data)
  // backward bookkeeping for grad(src)
  bool bwd_accumulate = false, bwd_mask = false;
  // Inception module this op belongs to (-1: trunk) and whether it runs on the side stream: the HBM-bound branch
  // (the 3x3x3 pool and b3b, both ways) overlaps the MFMA-bound 3x3x3 convs of the other branches
  int module = -1;
  bool side_fwd = false, side_bwd = false;
};

}  // namespace ivf

using namespace ivf;

struct ivf_i3d {
  ivf_i3d_config cfg;
  std::vector<ActBuf> bufs;
  std::vector<ConvLayer> convs;   // registration order, last = logits
  std::vector<Op> ops;
  int feat_buf = -1;
  int cam_buf = -1;      // Grad-CAM target endpoint of the pass in flight (-1: none): its gradient stays UNGATED
  size_t weights_floats = 0;
  size_t ws_bytes = 0;
  // misc workspace offsets (bytes)
  size_t off_logits, off_probs, off_pooled, off_dpooled, off_score, off_sig, off_terms, off_dreg, off_dsig,
      off_fbwd, off_target, off_cam, off_camw, off_mm, off_dfeat_raw, off_pair;
  float* warena = nullptr;
  char* ws = nullptr;
  std::vector<bool> loaded;
  // side stream + fork/join events (created on first use: the plan itself is built without a GPU)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // On by default (IVF_OVERLAP=0 / ivf_i3d_set_overlap(net, 0) turn it off): +1.2 % clips/s at B = 32 in the 6-pass
  // mode once the fork sits at module entry (round 2 forked at the first side op, behind b1b / b2b: +0.7 %, backward
  // only).  The branch it hides is HBM-bound and needs about half the CUs to hold its bandwidth (~24 GB/s per CU), while
  // a 3x3x3 conv workgroup needs a CU to itself (all of its LDS and registers), so the two mostly trade CUs instead of
  // sharing them.  Results are bit-identical either way (test_side_stream_overlap_is_bit_identical).
  bool overlap = true;

  // (typed float* for the C-ABI's sake: in an IVF_MATH_BF16ACT plan every buffer but the clip holds bf16 elements)
  float* act(int i) const { return (float*)(ws + bufs[i].act_off); }
  float* grad(int i) const { return (float*)(ws + bufs[i].grad_off); }
  bool act16() const { return cfg.math == IVF_MATH_BF16ACT; }
  unsigned char* gatebits(int i) const { return (unsigned char*)ws + bufs[i].gate_off; }
  template <class T>
  T* at(size_t off) const { return (T*)(ws + off); }
};

namespace ivf {

static const char* kInception[] = {"Mixed_3b", "Mixed_3c", "Mixed_4b", "Mixed_4c", "Mixed_4d",
                                   "Mixed_4e", "Mixed_4f", "Mixed_5b", "Mixed_5c"};
static const int kIncTable[9][7] = {
    {192, 64, 96, 128, 16, 32, 32},    {256, 128, 128, 192, 32, 96, 64},
    {480, 192, 96, 208, 16, 48, 64},   {512, 160, 112, 224, 24, 64, 64},
    {512, 128, 128, 256, 24, 64, 64},  {512, 112, 144, 288, 32, 64, 64},
    {528, 256, 160, 320, 32, 128, 128}, {832, 256, 160, 320, 32, 128, 128},
    {832, 384, 192, 384, 48, 128, 128}};

struct Builder {
  ivf_i3d* n;
  int add_buf(const std::string& name, int T, int H, int W, int C, bool relu) {
    ActBuf b;
    b.name = name; b.T = T; b.H = H; b.W = W; b.C = C; b.relu_out = relu;
    n->bufs.push_back(b);
    return (int)n->bufs.size() - 1;
  }
  int add_conv_layer(const std::string& name, int cin, int cout, int kt, int kh, int kw, int st, int sh,
                     int sw, bool bn) {
    ConvLayer c;
    c.name = name; c.cin = cin; c.cinp = (cin + 3) / 4 * 4; c.cout = cout;
    c.k[0] = kt; c.k[1] = kh; c.k[2] = kw; c.s[0] = st; c.s[1] = sh; c.s[2] = sw; c.has_bn = bn;
    n->convs.push_back(c);
    return (int)n->convs.size() - 1;
  }
  // conv op: src window -> dst window; dst buffer created by caller
  void conv_op(int layer, int src, int src_coff, int dst, int dst_coff) {
    ConvLayer& c = n->convs[layer];
    const ActBuf& s = n->bufs[src];
    Op o{};
    o.type = Op::CONV; o.src = src; o.dst = dst; o.src_coff = src_coff; o.cin = c.cinp;
    o.dst_coff = dst_coff; o.cout = c.cout; o.conv = layer;
    int dims[3] = {s.T, s.H, s.W};
    for (int d = 0; d < 3; ++d) {
      o.k[d] = c.k[d]; o.s[d] = c.s[d];
      int f, bk;
      same_pad(dims[d], c.k[d], c.s[d], &f, &bk);
      o.p[d] = f; c.pad[d] = f;
    }
    {
      const ActBuf& t = n->bufs[dst];
      o.flops_per_clip = 2.0 * t.T * t.H * t.W * c.cout * c.k[0] * c.k[1] * c.k[2] * c.cin;
    }
    n->ops.push_back(o);
    n->bufs[src].consumers++;
  }
  void pool_op(int src, int dst, int kt, int kh, int kw, int st, int sh, int sw) {
    const ActBuf& s = n->bufs[src];
    Op o{};
    o.type = Op::POOL; o.src = src; o.dst = dst; o.src_coff = 0; o.cin = s.C; o.dst_coff = 0; o.cout = s.C;
    int k[3] = {kt, kh, kw}, st3[3] = {st, sh, sw}, dims[3] = {s.T, s.H, s.W};
    for (int d = 0; d < 3; ++d) {
      o.k[d] = k[d]; o.s[d] = st3[d];
      int f, bk;
      same_pad(dims[d], k[d], st3[d], &f, &bk);
      o.p[d] = f;
    }
    n->ops.push_back(o);
    n->bufs[src].consumers++;
  }
};

static int build_plan(ivf_i3d* n) {
  const ivf_i3d_config& c = n->cfg;
  Builder b{n};
  // buffer 0: (perturbed) clip, channels-last padded to 4
  int x = b.add_buf("input", c.T, c.H, c.W, 4, false);
  auto unit = [&](const std::string& name, int src, int cin, int cout, int k, int st, int ss) {
    const ActBuf& s = n->bufs[src];
    int T = same_out(s.T, k, st), H = same_out(s.H, k, ss), W = same_out(s.W, k, ss);
    int dst = b.add_buf(name, T, H, W, cout, true);
    int L = b.add_conv_layer(name, cin, cout, k, k, k, st, ss, ss, true);
    b.conv_op(L, src, 0, dst, 0);
    return dst;
  };
  auto pool = [&](const std::string& name, int src, int kt, int ks, int st, int ss) {
    const ActBuf& s = n->bufs[src];
    int dst = b.add_buf(name, same_out(s.T, kt, st), same_out(s.H, ks, ss), same_out(s.W, ks, ss), s.C, false);
    b.pool_op(src, dst, kt, ks, ks, st, ss, ss);
    return dst;
  };
  x = unit("Conv3d_1a_7x7", x, c.C, 64, 7, c.stem_stride_t, 2);
  x = pool("MaxPool3d_2a_3x3", x, 1, 3, 1, 2);
  x = unit("Conv3d_2b_1x1", x, 64, 64, 1, 1, 1);
  x = unit("Conv3d_2c_3x3", x, 64, 192, 3, 1, 1);
  x = pool("MaxPool3d_3a_3x3", x, 1, 3, 1, 2);
  for (int m = 0; m < 9; ++m) {
    if (m == 2) x = pool("MaxPool3d_4a_3x3", x, 3, 3, c.pool4a_stride_t, 2);
    if (m == 7) x = pool("MaxPool3d_5a_2x2", x, 2, 2, c.pool5a_stride_t, 2);
    const int* t = kIncTable[m];
    std::string nm = kInception[m];
    const ActBuf s = n->bufs[x];
    if (s.C != t[0]) { set_error("plan: channel mismatch at %s", nm.c_str()); return IVF_ERR_BAD_ARG; }
    int ctot = t[1] + t[3] + t[5] + t[6];
    int y = b.add_buf(nm, s.T, s.H, s.W, ctot, true);
    int l0 = b.add_conv_layer(nm + ".b0", t[0], t[1], 1, 1, 1, 1, 1, 1, true);
    int l1a = b.add_conv_layer(nm + ".b1a", t[0], t[2], 1, 1, 1, 1, 1, 1, true);
    int l1b = b.add_conv_layer(nm + ".b1b", t[2], t[3], 3, 3, 3, 1, 1, 1, true);
    int l2a = b.add_conv_layer(nm + ".b2a", t[0], t[4], 1, 1, 1, 1, 1, 1, true);
    int l2b = b.add_conv_layer(nm + ".b2b", t[4], t[5], 3, 3, 3, 1, 1, 1, true);
    int l3b = b.add_conv_layer(nm + ".b3b", t[0], t[6], 1, 1, 1, 1, 1, 1, true);
    int t12 = b.add_buf(nm + ".b12a", s.T, s.H, s.W, t[2] + t[4], true);   // [b1a | b2a]
    int t3 = b.add_buf(nm + ".b3a", s.T, s.H, s.W, t[0], false);
    {
      ConvLayer& A = n->convs[l1a];
      ConvLayer& Bq = n->convs[l2a];
      const int gr = t[1] + t[2] + t[4];   // rows of the forward group [b0 | b1a | b2a]
      ConvLayer& Z = n->convs[l0];
      Z.grp_owner = l1a; Z.grp_row = 0; Z.grp_rows = gr;
      A.grp_owner = l1a; A.grp_row = t[1]; A.grp_rows = gr;
      Bq.grp_owner = l1a; Bq.grp_row = t[1] + t[2]; Bq.grp_rows = gr;
      const int kt = t[1] + t[2] + t[4];
      n->convs[l0].fus_owner = l0; n->convs[l0].fus_koff = 0; n->convs[l0].fus_ktotal = kt;
      A.fus_owner = l0; A.fus_koff = t[1]; A.fus_ktotal = kt;
      Bq.fus_owner = l0; Bq.fus_koff = t[1] + t[2]; Bq.fus_ktotal = kt;
    }
    const size_t first_op = n->ops.size();
    b.conv_op(l0, x, 0, y, 0);
    n->ops.back().fwd_skip = true;          // forward: part of the fused GEMM below
    n->ops.back().flops_per_clip = 0.0;     // (counted there)
    n->ops.back().bwd_fused = true;
    n->ops.back().src2 = t12;
    n->ops.back().flops_bwd_per_clip = 2.0 * s.T * s.H * s.W * (double)t[0] * (t[1] + t[2] + t[4]);
    b.conv_op(l1a, x, 0, y, 0);            // forward: the fused [b0 | b1a | b2a] GEMM, b0 -> y, [b1a | b2a] -> t12
    n->ops.back().fwd_group = true;
    n->ops.back().bwd_skip = true;
    n->ops.back().dst2 = t12;
    n->ops.back().n0 = t[1];
    n->ops.back().cout = t[1] + t[2] + t[4];
    n->ops.back().flops_per_clip = 2.0 * s.T * s.H * s.W * (double)t[0] * (t[1] + t[2] + t[4]);
    n->bufs[x].consumers--;               // its backward is part of the fused GEMM (counted via b0)
    b.conv_op(l1b, t12, 0, y, t[1]);
    b.conv_op(l2b, t12, t[2], y, t[1] + t[3]);
    b.pool_op(x, t3, 3, 3, 3, 1, 1, 1);
    n->ops.back().side_fwd = n->ops.back().side_bwd = true;
    b.conv_op(l3b, t3, 0, y, t[1] + t[3] + t[5]);
    n->ops.back().side_fwd = n->ops.back().side_bwd = true;
    for (size_t i = first_op; i < n->ops.size(); ++i) n->ops[i].module = m;
    x = y;
  }
  n->feat_buf = x;
  b.add_conv_layer("logits", 1024, c.num_classes, 1, 1, 1, 1, 1, 1, false);
  const ActBuf& f = n->bufs[x];
  if (f.T != c.head_kt || f.H != c.head_kh || f.W != c.head_kw) {
    set_error("i3d: head AvgPool3d window (%d,%d,%d) must cover the Mixed_5c map (%d,%d,%d); "
              "other windows give a [B,K,t] output the reference squeezes inconsistently (SURVEY F13)",
              c.head_kt, c.head_kh, c.head_kw, f.T, f.H, f.W);
    return IVF_ERR_UNSUPPORTED;
  }

  // backward flags: walk ops in reverse; per (buffer, channel window) the first writer of the
  // gradient overwrites, later ones accumulate, the last applies the ReLU gate of the buffer.
  {
    auto key = [](const Op& o) { return ((long long)o.src << 20) | o.src_coff; };
    std::vector<std::pair<long long, int>> writers;   // (key, count)
    auto find = [&](long long k) -> int& {
      for (auto& w : writers)
        if (w.first == k) return w.second;
      writers.push_back({k, 0});
      return writers.back().second;
    };
    for (const Op& o : n->ops)
      if (!o.bwd_skip) find(key(o))++;
    std::vector<std::pair<long long, int>> seen;
    auto seen_of = [&](long long k) -> int& {
      for (auto& w : seen)
        if (w.first == k) return w.second;
      seen.push_back({k, 0});
      return seen.back().second;
    };
    for (int i = (int)n->ops.size() - 1; i >= 0; --i) {
      Op& o = n->ops[i];
      if (o.bwd_skip) continue;
      int& sc = seen_of(key(o));
      o.bwd_accumulate = sc > 0;
      sc++;
      o.bwd_mask = (sc == find(key(o))) && n->bufs[o.src].relu_out;
    }
  }

  // ---- weights arena layout (floats)
  size_t w = 0;
  auto take = [&](size_t elems) { size_t o = w; w += (elems + 63) / 64 * 64; return o; };
  for (size_t i = 0; i < n->convs.size(); ++i) {
    ConvLayer& L = n->convs[i];
    int taps = L.k[0] * L.k[1] * L.k[2];
    if (L.name == "logits") {
      L.wf_elems = (size_t)L.cout * L.cin;
      L.wf_off = take(L.wf_elems);
      L.shift_off = take(L.cout);
      continue;
    }
    (void)taps;
    L.wf_elems = ivf_conv3d_pack_fwd_elems(L.cout, L.cinp, L.k[0], L.k[1], L.k[2], c.math);
    L.wf_off = take(L.wf_elems);
    L.wb_elems = ivf_conv3d_pack_bwd_elems(L.cout, L.cinp, L.k[0], L.k[1], L.k[2], L.s[0], L.s[1], L.s[2],
                                           L.pad[0], L.pad[1], L.pad[2], c.math);
    L.wb_off = take(L.wb_elems);
    L.scale_off = take(L.cout);
    L.shift_off = take(L.cout);
    if (L.grp_owner == (int)i) {
      L.grp_wf_elems = ivf_conv3d_pack_fwd_elems(L.grp_rows, L.cinp, 1, 1, 1, c.math);
      L.grp_wf_off = take(L.grp_wf_elems);
      L.grp_scale_off = take(L.grp_rows);
      L.grp_shift_off = take(L.grp_rows);
    }
    if (L.fus_owner == (int)i) {
      L.fus_wb_elems = ivf_conv3d_pack_bwd_fused1x1_elems(L.fus_ktotal, L.cinp, c.math);
      L.fus_wb_off = take(L.fus_wb_elems);
    }
  }
  n->weights_floats = w;

  // ---- workspace layout
  const size_t B = c.B;
  size_t bytes = 0;
  auto takeb = [&](size_t nbytes) { size_t o = bytes; bytes += align_up(nbytes, 256); return o; };
  for (auto& bf : n->bufs) {
    bf.esz = (n->act16() && bf.name != "input") ? 2 : 4;
    bf.act_off = takeb(B * bf.per_clip() * bf.esz);
    bf.grad_off = takeb(B * bf.per_clip() * bf.esz);
  }
  for (auto& o : n->ops)
    if (o.type == Op::POOL) {
      const ActBuf& d = n->bufs[o.dst];
      o.idx_off = takeb(B * d.per_clip());
    }
  static const bool no_gate_bits = getenv("IVF_NO_GATE_BITS") != nullptr;   // A/B switch for measurements
  for (const auto& o : n->ops)
    if (o.type == Op::CONV && !o.bwd_skip && o.bwd_mask && (n->bufs[o.src].C & 7) == 0 && !no_gate_bits)
      n->bufs[o.src].need_gate = true;
  // every producer of such a buffer must be able to record the bits: an implicit-GEMM / LDS-halo epilogue writing
  // an 8-aligned channel window (the stem's pix4 kernel and the pools cannot)
  for (const auto& o : n->ops) {
    if (o.type == Op::POOL) n->bufs[o.dst].need_gate = false;
    if (o.type == Op::CONV && ((o.dst_coff | o.cout | o.n0) & 7)) {
      n->bufs[o.dst].need_gate = false;
      if (o.dst2 >= 0) n->bufs[o.dst2].need_gate = false;
    }
    if (o.type == Op::CONV && o.cin == 4 && o.s[1] == 2) n->bufs[o.dst].need_gate = false;   // the stem (pix4)
  }
  for (auto& bf : n->bufs)
    if (bf.need_gate) bf.gate_off = takeb(B * bf.per_clip() / 8);
  const int K = c.num_classes, T = c.T;
  n->off_logits = takeb(B * K * 4);
  n->off_probs = takeb(B * K * 4);
  n->off_pooled = takeb(B * 1024 * 4);
  n->off_dpooled = takeb(B * 1024 * 4);
  n->off_score = takeb(B * 4);
  n->off_sig = takeb(B * T * 4);
  n->off_terms = takeb(B * 2 * 4);
  n->off_dreg = takeb(B * T * 4);
  n->off_dsig = takeb(B * T * 4);
  n->off_fbwd = takeb(ivf_freeze_bwd_workspace_bytes((int)B, T));
  n->off_target = takeb(B * 4);
  size_t max_pos = 0, max_t = 0;      // Grad-CAM on any endpoint (ivf_i3d_gradcam_layer)
  for (const auto& bf : n->bufs) {
    if (bf.name == "input" || bf.name.find('.') != std::string::npos) continue;
    max_pos = std::max(max_pos, (size_t)bf.T * bf.H * bf.W);
    max_t = std::max(max_t, (size_t)bf.T);
  }
  n->off_cam = takeb(B * max_pos * 4);
  n->off_camw = takeb(B * 1024 * 4);
  n->off_mm = takeb(B * max_t * 2 * 4);
  n->off_dfeat_raw = takeb(B * f.per_clip() * 4);
  n->off_pair = takeb(B * T * 12);
  n->ws_bytes = bytes;
  n->loaded.assign(n->convs.size(), false);
  return IVF_OK;
}

static void fill_conv_fwd(const ivf_i3d* n, const Op& o, int b, ivf_conv3d_desc* d) {
  const ActBuf& s = n->bufs[o.src];
  const ActBuf& t = n->bufs[o.dst];
  memset(d, 0, sizeof(*d));
  d->B = b; d->Ti = s.T; d->Hi = s.H; d->Wi = s.W; d->Cin = o.cin; d->in_ld = s.C; d->in_coff = o.src_coff;
  d->To = t.T; d->Ho = t.H; d->Wo = t.W; d->Cout = o.cout; d->out_ld = t.C; d->out_coff = o.dst_coff;
  d->kT = o.k[0]; d->kH = o.k[1]; d->kW = o.k[2]; d->sT = o.s[0]; d->sH = o.s[1]; d->sW = o.s[2];
  d->pT = o.p[0]; d->pH = o.p[1]; d->pW = o.p[2];
  d->relu = 1;
  d->math = n->cfg.math;
  d->variant = o.var_fwd;
  if (o.dst2 >= 0) {   // the fused [b0 | b1a | b2a] GEMM: o.cout spans the three units
    d->N0 = o.n0;
    d->out2 = n->act(o.dst2);
    d->out2_ld = n->bufs[o.dst2].C;
    d->out2_coff = 0;
    if (n->bufs[o.dst2].need_gate) {
      d->gate_out2 = n->gatebits(o.dst2);
      d->gate_out2_ld = n->bufs[o.dst2].C / 8;
    }
  }
  if (t.need_gate) {
    d->gate_out = n->gatebits(o.dst);
    d->gate_out_ld = t.C / 8;
    d->gate_out_coff = o.dst_coff;
  }
}

static void fill_conv_bwd(const ivf_i3d* n, const Op& o, int b, ivf_conv3d_desc* d) {
  const ActBuf& s = n->bufs[o.src];   // gradient written here
  const ActBuf& t = n->bufs[o.dst];   // gradient read from here
  const ConvLayer& L = n->convs[o.conv];
  memset(d, 0, sizeof(*d));
  d->B = b; d->Ti = t.T; d->Hi = t.H; d->Wi = t.W; d->Cin = o.cout; d->in_ld = t.C; d->in_coff = o.dst_coff;
  d->kT = L.geom.kT; d->kH = L.geom.kH; d->kW = L.geom.kW; d->sT = d->sH = d->sW = 1;
  d->pT = L.geom.pT; d->pH = L.geom.pH; d->pW = L.geom.pW;
  d->out_ld = s.C; d->out_coff = o.src_coff;
  d->accumulate = o.bwd_accumulate;
  d->mask_ld = s.C; d->mask_coff = o.src_coff;
  d->math = n->cfg.math;
  d->variant = o.var_bwd;
  if (o.bwd_fused) {
    // one GEMM over [dY_b0 | dT_b1a | dT_b2a] -> d(input of the module)
    const ActBuf& t12 = n->bufs[o.src2];
    d->Cin = L.fus_ktotal;
    d->K0 = o.cout;
    d->in2_ld = t12.C; d->in2_coff = 0;
    d->in2 = n->grad(o.src2);
    d->To = s.T; d->Ho = s.H; d->Wo = s.W;
    d->Cout = o.cin;
    return;
  }
  if (L.geom.d2s) {
    d->d2s = 1;
    d->bsT = o.s[0]; d->bsH = o.s[1]; d->bsW = o.s[2];
    d->To = cdiv(s.T, o.s[0]); d->Ho = cdiv(s.H, o.s[1]); d->Wo = cdiv(s.W, o.s[2]);
    d->Cout = L.geom.rows;
    d->dT = s.T; d->dH = s.H; d->dW = s.W; d->dC = o.cin;
  } else {
    d->To = s.T; d->Ho = s.H; d->Wo = s.W;
    d->Cout = o.cin;
  }
}

static void fill_pool(const ivf_i3d* n, const Op& o, int b, ivf_pool3d_desc* d) {
  const ActBuf& s = n->bufs[o.src];
  const ActBuf& t = n->bufs[o.dst];
  memset(d, 0, sizeof(*d));
  d->B = b; d->Ti = s.T; d->Hi = s.H; d->Wi = s.W; d->C = s.C; d->in_ld = s.C; d->in_coff = 0;
  d->To = t.T; d->Ho = t.H; d->Wo = t.W; d->out_ld = t.C; d->out_coff = 0;
  d->kT = o.k[0]; d->kH = o.k[1]; d->kW = o.k[2]; d->sT = o.s[0]; d->sH = o.s[1]; d->sW = o.s[2];
  d->pT = o.p[0]; d->pH = o.p[1]; d->pW = o.p[2];
  // a pool that owes its input gradient the ReLU gate marks dead windows in the forward
  // instead of re-reading the activation in the backward
  // (not for the Grad-CAM target: the hook of grad-cam.py:50-51 sees the gradient w.r.t. the endpoint's
  // OUTPUT, before its own ReLU gate, so the true arg-max must be kept even in dead windows)
  d->gate_nonpos = o.bwd_mask && o.src != n->cam_buf;
  d->act_bf16 = n->act16();
}

static int check_ready(const ivf_i3d* n, int b) {
  IVF_CHECK_ARG(n != nullptr, "i3d: null handle");
  IVF_CHECK_ARG(n->warena && n->ws, "i3d: ivf_i3d_bind has not been called");
  IVF_CHECK_ARG(b > 0 && b <= n->cfg.B, "i3d: batch %d outside [1,%d]", b, n->cfg.B);
  for (size_t i = 0; i < n->loaded.size(); ++i)
    IVF_CHECK_ARG(n->loaded[i], "i3d: weights of unit %s not loaded", n->convs[i].name.c_str());
  return IVF_OK;
}

// Fork/join of the side stream around the ops of one Inception module.  Every buffer of the plan has its own
// storage, so the only orderings to keep are the true dependencies: the side branch reads the module input (fork
// after everything before it on the caller's stream), the next consumer of the module output -- or of grad(input),
// which the pool's backward writes first and the fused 1x1x1 backward GEMM accumulates into -- waits for it (join).
struct SideLane {
  ivf_i3d* n;
  hipStream_t main;
  bool enabled, forked = false;
  int rc = IVF_OK;
  SideLane(ivf_i3d* net, hipStream_t s) : n(net), main(s) {
    enabled = net->overlap;
    if (enabled && !n->side) {
      if (hipStreamCreateWithFlags(&n->side, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&n->ev_join, hipEventDisableTiming) != hipSuccess) {
        set_error("i3d: could not create the side stream");
        rc = IVF_ERR_HIP;
        enabled = false;
      }
    }
  }
  // Fork at MODULE ENTRY: the side branch depends on the module input only, so the event is recorded before the
  // first main-lane op of the module is enqueued (recorded at the first side op it would sit behind b1b / b2b, which
  // come earlier in forward op order, and nothing would overlap).
  void fork() {
    if (!enabled || forked) return;
    if (hipEventRecord(n->ev_fork, main) != hipSuccess || hipStreamWaitEvent(n->side, n->ev_fork, 0) != hipSuccess) rc = IVF_ERR_HIP;
    forked = true;
  }
  hipStream_t side() {   // stream for a side-lane op
    if (!enabled) return main;
    fork();
    return n->side;
  }
  void join() {
    if (!forked) return;
    if (hipEventRecord(n->ev_join, n->side) != hipSuccess || hipStreamWaitEvent(main, n->ev_join, 0) != hipSuccess) rc = IVF_ERR_HIP;
    forked = false;
  }
};

static int run_forward(ivf_i3d* n, int b, float* logits, float* probs, hipStream_t s) {
  SideLane lane(n, s);
  IVF_PROPAGATE(lane.rc);
  int module = -1;
  for (const Op& o : n->ops) {
    if (o.module != module) {   // the module input must be complete: wait for the previous module's side branch
      lane.join();
      module = o.module;
      if (module >= 0) lane.fork();
    }
    if (o.fwd_skip) continue;
    hipStream_t st = o.side_fwd ? lane.side() : s;
    prof_set_site(o.type == Op::CONV ? 2 * (int)(&o - n->ops.data()) : -1);
    if (o.type == Op::CONV) {
      const ConvLayer& L = n->convs[o.conv];
      ivf_conv3d_desc d;
      fill_conv_fwd(n, o, b, &d);
      prof_set_flops(o.flops_per_clip * b);
      int rc;
      if (o.fwd_group)
        rc = ivf_conv3d(&d, n->act(o.src), n->warena + L.grp_wf_off, n->warena + L.grp_scale_off,
                        n->warena + L.grp_shift_off, nullptr, n->act(o.dst), st);
      else
        rc = ivf_conv3d(&d, n->act(o.src), n->warena + L.wf_off, n->warena + L.scale_off, n->warena + L.shift_off,
                        nullptr, n->act(o.dst), st);
      if (rc != IVF_OK) { lane.join(); return rc; }
    } else {
      ivf_pool3d_desc d;
      fill_pool(n, o, b, &d);
      int rc = ivf_maxpool3d_fwd(&d, n->act(o.src), n->act(o.dst), n->at<unsigned char>(o.idx_off), st);
      if (rc != IVF_OK) { lane.join(); return rc; }
    }
  }
  lane.join();
  IVF_PROPAGATE(lane.rc);
  const ActBuf& f = n->bufs[n->feat_buf];
  const ConvLayer& L = n->convs.back();
  float* lg = n->at<float>(n->off_logits);
  float* pr = n->at<float>(n->off_probs);
  IVF_PROPAGATE((n->act16() ? ivf_head_fwd_bf16 : (decltype(&ivf_head_fwd_bf16))ivf_head_fwd)(
      n->act(n->feat_buf), n->warena + L.wf_off, n->warena + L.shift_off, n->at<float>(n->off_pooled), lg, pr, b,
      f.T * f.H * f.W, f.C, n->cfg.num_classes, n->cfg.softmax, s));
  size_t nb = (size_t)b * n->cfg.num_classes * sizeof(float);
  if (logits) IVF_CHECK_HIP(hipMemcpyAsync(logits, lg, nb, hipMemcpyDeviceToDevice, s));
  if (probs) IVF_CHECK_HIP(hipMemcpyAsync(probs, pr, nb, hipMemcpyDeviceToDevice, s));
  return IVF_OK;
}

// backward-data from the head down to the channels-last input gradient
static int run_backward(ivf_i3d* n, int b, const int* target, const float* dout, float* score,
                        hipStream_t s) {
  const ActBuf& f = n->bufs[n->feat_buf];
  const ConvLayer& Lh = n->convs.back();
  IVF_PROPAGATE((n->act16() ? ivf_head_bwd_bf16 : (decltype(&ivf_head_bwd_bf16))ivf_head_bwd)(
      n->act(n->feat_buf), n->warena + Lh.wf_off, n->at<float>(n->off_probs), target, dout, score, nullptr,
      n->grad(n->feat_buf), b, f.T * f.H * f.W, f.C, n->cfg.num_classes, n->cfg.softmax, 1, s));
  SideLane lane(n, s);
  IVF_PROPAGATE(lane.rc);
  int module = -1;
  for (int i = (int)n->ops.size() - 1; i >= 0; --i) {
    const Op& o = n->ops[i];
    if (o.module != module) {   // grad(module output) must be complete
      lane.join();
      module = o.module;
      if (module >= 0) lane.fork();
    }
    if (n->cam_buf >= 0 && o.dst == n->cam_buf) break;   // Grad-CAM pass: the target's gradient is complete
    if (o.bwd_skip) continue;
    // the fused 1x1x1 backward GEMM accumulates into grad(input) after the pool's backward has written it
    if (o.bwd_fused) lane.join();
    hipStream_t st = o.side_bwd ? lane.side() : s;
    const float* gate = (o.bwd_mask && o.src != n->cam_buf) ? n->act(o.src) : nullptr;
    prof_set_site(o.type == Op::CONV ? 2 * i + 1 : -1);
    int rc;
    if (o.type == Op::CONV) {
      const ConvLayer& L = n->convs[o.conv];
      ivf_conv3d_desc d;
      fill_conv_bwd(n, o, b, &d);
      const float* fgate = gate;
      if (gate && n->bufs[o.src].need_gate) {   // the 1-bit record instead of the fp32 activation
        d.gate_in = n->gatebits(o.src);
        d.gate_in_ld = n->bufs[o.src].C / 8;
        d.gate_in_coff = o.src_coff;
        fgate = nullptr;
      }
      prof_set_flops((o.bwd_fused ? o.flops_bwd_per_clip : o.flops_per_clip) * b);
      rc = ivf_conv3d(&d, n->grad(o.dst), n->warena + (o.bwd_fused ? L.fus_wb_off : L.wb_off), nullptr, nullptr, fgate,
                      n->grad(o.src), st);
    } else {
      ivf_pool3d_desc d;
      fill_pool(n, o, b, &d);
      // sole writer of a ReLU output's gradient: gated through the arg-max record (fill_pool);
      // with other writers before it the accumulated sum still needs the explicit gate
      rc = ivf_maxpool3d_bwd(&d, n->grad(o.dst), n->at<unsigned char>(o.idx_off), n->grad(o.src),
                             o.bwd_accumulate ? gate : nullptr, o.bwd_accumulate, st);
    }
    if (rc != IVF_OK) { lane.join(); return rc; }
  }
  lane.join();
  IVF_PROPAGATE(lane.rc);
  return IVF_OK;
}

__global__ void ncthw_to_cl4_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int T,
                                    int HW) {
  size_t total = (size_t)B * T * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int px = i % HW;
    int t = (i / HW) % T;
    int b = i / ((size_t)HW * T);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) v[c] = x[((size_t)(b * C + c) * T + t) * HW + px];
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__global__ void cl4_to_ncthw_kernel(const float* __restrict__ y, float* __restrict__ x, int B, int C, int T,
                                    int HW) {
  size_t total = (size_t)B * C * T * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int px = i % HW;
    int t = (i / HW) % T;
    int c = (i / ((size_t)HW * T)) % C;
    int b = i / ((size_t)HW * T * C);
    x[i] = y[((size_t)(b * T + t) * HW + px) * 4 + c];
  }
}

static inline int grid_for(size_t total, int block = 256, int cap = 4096) {
  size_t g = (total + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g ? g : 1));
}

}  // namespace ivf

extern "C" int ivf_i3d_create(const ivf_i3d_config* cfg, ivf_i3d_t** out) {
  IVF_CHECK_ARG(cfg && out, "i3d_create: null pointer");
  IVF_CHECK_ARG(cfg->B > 0 && cfg->C > 0 && cfg->C <= 4 && cfg->T > 0 && cfg->T <= 64 && cfg->H > 0 &&
                    cfg->W > 0 && cfg->num_classes > 0,
                "i3d_create: bad geometry (C<=4, T<=64)");
  IVF_CHECK_ARG(cfg->stem_stride_t >= 1 && cfg->stem_stride_t <= 2 && cfg->pool4a_stride_t >= 1 &&
                    cfg->pool4a_stride_t <= 2 && cfg->pool5a_stride_t >= 1 && cfg->pool5a_stride_t <= 2,
                "i3d_create: temporal strides must be 1 or 2");
  IVF_CHECK_ARG(cfg->math >= IVF_MATH_FP32 && cfg->math <= IVF_MATH_BF16ACT, "i3d_create: unknown math mode");
  ivf_i3d* n = new ivf_i3d();
  n->cfg = *cfg;
  n->overlap = !(getenv("IVF_OVERLAP") != nullptr && getenv("IVF_OVERLAP")[0] == '0');   // on unless IVF_OVERLAP=0
  int rc = build_plan(n);
  if (rc != IVF_OK) {
    delete n;
    return rc;
  }
  *out = n;
  return IVF_OK;
}

extern "C" int ivf_i3d_set_overlap(ivf_i3d_t* net, int on) {
  IVF_CHECK_ARG(net, "i3d_set_overlap: null net");
  net->overlap = on != 0;
  return IVF_OK;
}

extern "C" void ivf_i3d_destroy(ivf_i3d_t* net) {
  if (!net) return;
  if (net->ev_fork) (void)hipEventDestroy(net->ev_fork);
  if (net->ev_join) (void)hipEventDestroy(net->ev_join);
  if (net->side) (void)hipStreamDestroy(net->side);
  delete net;
}
extern "C" size_t ivf_i3d_weights_bytes(const ivf_i3d_t* net) { return net ? net->weights_floats * 4 : 0; }
extern "C" size_t ivf_i3d_workspace_bytes(const ivf_i3d_t* net) { return net ? net->ws_bytes : 0; }

extern "C" int ivf_i3d_bind(ivf_i3d_t* net, void* weights_arena, void* workspace) {
  IVF_CHECK_ARG(net && weights_arena && workspace, "i3d_bind: null pointer");
  IVF_CHECK_ARG(((uintptr_t)weights_arena & 255) == 0 && ((uintptr_t)workspace & 255) == 0,
                "i3d_bind: arenas must be 256-byte aligned");
  net->warena = (float*)weights_arena;
  net->ws = (char*)workspace;
  return IVF_OK;
}

extern "C" int ivf_i3d_num_convs(const ivf_i3d_t* net) { return net ? (int)net->convs.size() : 0; }

extern "C" int ivf_i3d_conv_info(const ivf_i3d_t* net, int i, char* name64, int* cout, int* cin, int* kT,
                                 int* kH, int* kW, int* has_bn) {
  IVF_CHECK_ARG(net && i >= 0 && i < (int)net->convs.size(), "i3d_conv_info: bad index");
  const ConvLayer& L = net->convs[i];
  if (name64) { strncpy(name64, L.name.c_str(), 63); name64[63] = 0; }
  if (cout) *cout = L.cout;
  if (cin) *cin = L.cin;
  if (kT) *kT = L.k[0];
  if (kH) *kH = L.k[1];
  if (kW) *kW = L.k[2];
  if (has_bn) *has_bn = L.has_bn;
  return IVF_OK;
}

extern "C" int ivf_i3d_load_conv(ivf_i3d_t* net, int i, const float* w, const float* g, const float* be,
                                 const float* mu, const float* var, const float* bias, float eps,
                                 ivf_stream_t stream) {
  IVF_CHECK_ARG(net && net->warena, "i3d_load_conv: bind first");
  IVF_CHECK_ARG(i >= 0 && i < (int)net->convs.size() && w, "i3d_load_conv: bad index/null weight");
  ConvLayer& L = net->convs[i];
  hipStream_t s = (hipStream_t)stream;
  float* A = net->warena;
  if (L.name == "logits") {
    IVF_CHECK_ARG(bias, "i3d_load_conv: logits unit needs its bias");
    IVF_CHECK_HIP(hipMemcpyAsync(A + L.wf_off, w, L.wf_elems * 4, hipMemcpyDeviceToDevice, s));
    IVF_CHECK_HIP(hipMemcpyAsync(A + L.shift_off, bias, (size_t)L.cout * 4, hipMemcpyDeviceToDevice, s));
    net->loaded[i] = true;
    return IVF_OK;
  }
  IVF_CHECK_ARG(g && be && mu && var, "i3d_load_conv: unit %s needs BatchNorm tensors", L.name.c_str());
  IVF_PROPAGATE(ivf_bn_fold(g, be, mu, var, eps, A + L.scale_off, A + L.shift_off, L.cout, s));
  IVF_PROPAGATE(ivf_conv3d_pack_fwd(w, A + L.wf_off, L.cout, L.cin, L.cinp, L.k[0], L.k[1], L.k[2],
                                    net->cfg.math, s));
  IVF_PROPAGATE(ivf_conv3d_pack_bwd(w, A + L.scale_off, A + L.wb_off, L.cout, L.cin, L.cinp, L.k[0], L.k[1],
                                    L.k[2], L.s[0], L.s[1], L.s[2], L.pad[0], L.pad[1], L.pad[2], net->cfg.math,
                                    &L.geom, s));
  if (L.grp_owner >= 0) {
    const ConvLayer& G = net->convs[L.grp_owner];
    IVF_PROPAGATE(ivf_conv3d_pack_fwd_rows(w, A + G.grp_wf_off, L.cout, L.cin, L.cinp, 1, 1, 1, L.grp_row,
                                           G.grp_rows, net->cfg.math, s));
    IVF_CHECK_HIP(hipMemcpyAsync(A + G.grp_scale_off + L.grp_row, A + L.scale_off, (size_t)L.cout * 4,
                                 hipMemcpyDeviceToDevice, s));
    IVF_CHECK_HIP(hipMemcpyAsync(A + G.grp_shift_off + L.grp_row, A + L.shift_off, (size_t)L.cout * 4,
                                 hipMemcpyDeviceToDevice, s));
  }
  if (L.fus_owner >= 0) {
    const ConvLayer& F = net->convs[L.fus_owner];
    IVF_PROPAGATE(ivf_conv3d_pack_bwd_fused1x1(w, A + L.scale_off, A + F.fus_wb_off, L.cout, L.cin, L.cinp, L.fus_koff,
                                               L.fus_ktotal, net->cfg.math, s));
  }
  net->loaded[i] = true;
  return IVF_OK;
}

extern "C" float* ivf_i3d_input_buffer(ivf_i3d_t* net) { return (net && net->ws) ? net->act(0) : nullptr; }
extern "C" float* ivf_i3d_input_grad_buffer(ivf_i3d_t* net) { return (net && net->ws) ? net->grad(0) : nullptr; }

extern "C" int ivf_i3d_forward_staged(ivf_i3d_t* net, int b, float* logits, float* probs,
                                      ivf_stream_t stream) {
  IVF_PROPAGATE(check_ready(net, b));
  return run_forward(net, b, logits, probs, (hipStream_t)stream);
}

extern "C" int ivf_i3d_forward(ivf_i3d_t* net, const float* x, int b, float* logits, float* probs,
                               ivf_stream_t stream) {
  IVF_PROPAGATE(check_ready(net, b));
  IVF_CHECK_ARG(x, "i3d_forward: null clip");
  const ivf_i3d_config& c = net->cfg;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ncthw_to_cl4_kernel, dim3(grid_for((size_t)b * c.T * c.H * c.W)), dim3(256), 0, s, x,
                     net->act(0), b, c.C, c.T, c.H * c.W);
  IVF_CHECK_LAUNCH();
  return run_forward(net, b, logits, probs, s);
}

extern "C" int ivf_i3d_backward(ivf_i3d_t* net, int b, const int* target, const float* dout, float* score,
                                float* dx, ivf_stream_t stream) {
  IVF_PROPAGATE(check_ready(net, b));
  IVF_CHECK_ARG(target || dout, "i3d_backward: need target or dout");
  hipStream_t s = (hipStream_t)stream;
  IVF_PROPAGATE(run_backward(net, b, target, dout, score, s));
  if (dx) {
    const ivf_i3d_config& c = net->cfg;
    hipLaunchKernelGGL(cl4_to_ncthw_kernel, dim3(grid_for((size_t)b * c.C * c.T * c.H * c.W)), dim3(256), 0,
                       s, net->grad(0), dx, b, c.C, c.T, c.H * c.W);
    IVF_CHECK_LAUNCH();
  }
  return IVF_OK;
}

extern "C" int ivf_i3d_endpoint(const ivf_i3d_t* net, const char* name, float** ptr, int* T, int* H, int* W,
                                int* C, int* ld) {
  IVF_CHECK_ARG(net && name && net->ws, "i3d_endpoint: bad args / not bound");
  std::string want(name);
  bool want_grad = false;
  if (want.size() > 5 && want.compare(want.size() - 5, 5, ":grad") == 0) {
    want_grad = true;   // gradient buffer of the last backward (ReLU-gated where the endpoint is a ReLU output)
    want.resize(want.size() - 5);
  }
  for (size_t i = 0; i < net->bufs.size(); ++i) {
    const ActBuf& b = net->bufs[i];
    if (b.name == want) {
      if (ptr) *ptr = want_grad ? net->grad((int)i) : net->act((int)i);
      if (T) *T = b.T;
      if (H) *H = b.H;
      if (W) *W = b.W;
      if (C) *C = b.C;
      if (ld) *ld = b.C;
      return IVF_OK;
    }
  }
  set_error("i3d_endpoint: unknown endpoint '%s'", name);
  return IVF_ERR_BAD_ARG;
}

extern "C" int ivf_i3d_search(ivf_i3d_t* net, const float* x, int b, const int* target, float* raw_mask,
                              float* exp_avg, float* exp_avg_sq, float lam1, float lam2, float lr,
                              float beta1, float beta2, float eps, int N, int first_step, int mode,
                              float* traj, ivf_stream_t stream) {
  IVF_PROPAGATE(check_ready(net, b));
  IVF_CHECK_ARG(x && target && raw_mask && exp_avg && exp_avg_sq, "i3d_search: null pointer");
  IVF_CHECK_ARG(N >= 0 && first_step >= 1, "i3d_search: bad iteration counts");
  IVF_CHECK_ARG(mode == 0 || mode == 1, "i3d_search: mode must be 0 (freeze) or 1 (reverse)");
  const ivf_i3d_config& c = net->cfg;
  hipStream_t s = (hipStream_t)stream;
  const int T = c.T, HW = c.H * c.W;
  float* sig = net->at<float>(net->off_sig);
  float* terms = net->at<float>(net->off_terms);
  float* dreg = net->at<float>(net->off_dreg);
  float* dsig = net->at<float>(net->off_dsig);
  float* score = net->at<float>(net->off_score);
  int* partner = net->at<int>(net->off_pair);
  float* weight = (float*)(partner + (size_t)c.B * c.T);
  for (int it = 0; it < N; ++it) {
    prof_set_iteration(it);
    IVF_PROPAGATE(ivf_mask_reg(raw_mask, b, T, lam1, lam2, sig, terms, dreg, s));              // smth:198-200
    if (mode == 0) {
      IVF_PROPAGATE(ivf_freeze_fwd(x, sig, net->act(0), b, c.C, T, HW, 1, 4, s));              // smth:202
    } else {
      IVF_PROPAGATE(ivf_submask_pairs_batched(sig, b, T, 0.1f, partner, weight, s));
      IVF_PROPAGATE(ivf_reverse_fwd_batched(x, partner, weight, net->act(0), b, c.C, T, HW, 4, s));
    }
    IVF_PROPAGATE(run_forward(net, b, nullptr, nullptr, s));                                    // smth:202-205
    IVF_PROPAGATE(run_backward(net, b, target, nullptr, score, s));                             // smth:213
    if (mode == 0)
      IVF_PROPAGATE(ivf_freeze_bwd(x, sig, net->grad(0), dsig, nullptr, b, c.C, T, HW, 1, 4,
                                   net->at<void>(net->off_fbwd), s));
    else
      IVF_PROPAGATE(ivf_reverse_bwd(x, partner, net->grad(0), dsig, b, c.C, T, HW, 4,
                                    net->at<void>(net->off_fbwd), s));
    IVF_PROPAGATE(ivf_search_step(raw_mask, sig, dsig, dreg, terms, score, exp_avg, exp_avg_sq,
                                  traj ? traj + (size_t)it * b * 4 : nullptr, b, T, first_step + it, lr,
                                  beta1, beta2, eps, s));                                       // smth:207-214
  }
  prof_set_iteration(-1);   // sampling off outside the loop
  return IVF_OK;
}

extern "C" int ivf_i3d_perturbed_forward(ivf_i3d_t* net, const float* x, int b, const float* mask, int mode,
                                         float* probs, ivf_stream_t stream) {
  IVF_PROPAGATE(check_ready(net, b));
  IVF_CHECK_ARG(x && mask && (mode == 0 || mode == 1), "i3d_perturbed_forward: bad args");
  const ivf_i3d_config& c = net->cfg;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) {
    IVF_PROPAGATE(ivf_freeze_fwd(x, mask, net->act(0), b, c.C, c.T, c.H * c.W, 1, 4, s));
  } else {
    int* partner = net->at<int>(net->off_pair);
    float* weight = (float*)(partner + (size_t)c.B * c.T);
    IVF_PROPAGATE(ivf_submask_pairs_batched(mask, b, c.T, 0.1f, partner, weight, s));
    IVF_PROPAGATE(ivf_reverse_fwd_batched(x, partner, weight, net->act(0), b, c.C, c.T, c.H * c.W, 4, s));
  }
  return run_forward(net, b, nullptr, probs, s);
}

extern "C" int ivf_i3d_gradcam(ivf_i3d_t* net, const float* x, int b, const int* target, int per_frame,
                               int out_h, int out_w, float* cam, float* probs, ivf_stream_t stream) {
  IVF_PROPAGATE(check_ready(net, b));
  IVF_CHECK_ARG(x && target && cam && out_h > 0 && out_w > 0, "i3d_gradcam: bad args");
  const ivf_i3d_config& c = net->cfg;
  hipStream_t s = (hipStream_t)stream;
  IVF_PROPAGATE(ivf_i3d_forward(net, x, b, nullptr, probs, s));
  const ActBuf& f = net->bufs[net->feat_buf];
  const ConvLayer& Lh = net->convs.back();
  const int npos = f.T * f.H * f.W;
  float* draw = net->at<float>(net->off_dfeat_raw);
  // gradient of the (post-softmax) class score w.r.t. Mixed_5c, ungated (the hook of
  // pytorch-grad-cam/grad-cam.py:50-51 sees the raw gradient)
  IVF_PROPAGATE((net->act16() ? ivf_head_bwd_bf16 : (decltype(&ivf_head_bwd_bf16))ivf_head_bwd)(
      net->act(net->feat_buf), net->warena + Lh.wf_off, net->at<float>(net->off_probs), target, nullptr, nullptr, nullptr,
      draw, b, npos, f.C, c.num_classes, c.softmax, 0, s));
  float* wts = net->at<float>(net->off_camw);
  float* cm = net->at<float>(net->off_cam);
  IVF_PROPAGATE((net->act16() ? ivf_gradcam_reduce_bf16 : (decltype(&ivf_gradcam_reduce_bf16))ivf_gradcam_reduce)(
      net->act(net->feat_buf), draw, wts, cm, b, npos, f.C, s));
  IVF_CHECK_ARG(c.T / f.T >= 1, "i3d_gradcam: clip shorter than the feature map");
  return ivf_cam_resize_normalise(cm, cam, net->at<float>(net->off_mm), b, f.T, f.H, f.W, out_h, out_w,
                                  c.T / f.T, per_frame, s);
}

extern "C" int ivf_i3d_gradcam_layer(ivf_i3d_t* net, const float* x, int b, const int* target, const char* layer,
                                     int per_frame, int out_h, int out_w, float* cam, float* probs,
                                     ivf_stream_t stream) {
  IVF_PROPAGATE(check_ready(net, b));
  IVF_CHECK_ARG(x && target && layer && cam && out_h > 0 && out_w > 0, "i3d_gradcam_layer: bad args");
  int X = -1;
  for (size_t i = 1; i < net->bufs.size(); ++i)
    if (net->bufs[i].name == layer && net->bufs[i].name.find('.') == std::string::npos) X = (int)i;
  if (X < 0) {
    set_error("i3d_gradcam_layer: '%s' is not an endpoint of the model (Conv3d_1a_7x7 ... Mixed_5c)", layer);
    return IVF_ERR_BAD_ARG;
  }
  if (X == net->feat_buf) return ivf_i3d_gradcam(net, x, b, target, per_frame, out_h, out_w, cam, probs, stream);
  const ivf_i3d_config& c = net->cfg;
  hipStream_t s = (hipStream_t)stream;
  const ActBuf& t = net->bufs[X];
  IVF_CHECK_ARG(c.T / t.T >= 1, "i3d_gradcam_layer: clip shorter than the endpoint's map");
  net->cam_buf = X;
  int rc = ivf_i3d_forward(net, x, b, nullptr, probs, s);
  if (rc == IVF_OK) rc = run_backward(net, b, target, nullptr, nullptr, s);   // stops above X, X ungated
  net->cam_buf = -1;
  IVF_PROPAGATE(rc);
  const int npos = t.T * t.H * t.W;
  float* wts = net->at<float>(net->off_camw);
  float* cm = net->at<float>(net->off_cam);
  IVF_PROPAGATE((net->act16() ? ivf_gradcam_reduce_bf16 : (decltype(&ivf_gradcam_reduce_bf16))ivf_gradcam_reduce)(
      net->act(X), net->grad(X), wts, cm, b, npos, t.C, s));
  return ivf_cam_resize_normalise(cm, cam, net->at<float>(net->off_mm), b, t.T, t.H, t.W, out_h, out_w, c.T / t.T,
                                  per_frame, s);
}

/* bytes per stored activation / gradient element of the plan's endpoints: 4, or 2 (bf16) for IVF_MATH_BF16ACT */
extern "C" int ivf_i3d_act_elem_bytes(const ivf_i3d_t* net) { return (net && net->act16()) ? 2 : 4; }

extern "C" double ivf_i3d_conv_flops_per_clip(const ivf_i3d_t* net) {
  double f = 0.0;
  if (net)
    for (const Op& o : net->ops) f += o.flops_per_clip;
  return f;
}

extern "C" int ivf_i3d_num_sites(const ivf_i3d_t* net) { return net ? 2 * (int)net->ops.size() : 0; }

extern "C" int ivf_i3d_site_name(const ivf_i3d_t* net, int site, char* name64) {
  IVF_CHECK_ARG(net && name64 && site >= 0 && site < 2 * (int)net->ops.size(), "i3d_site_name: bad site");
  const Op& o = net->ops[site >> 1];
  IVF_CHECK_ARG(o.type == Op::CONV, "i3d_site_name: site %d is not a convolution", site);
  const ConvLayer& L = net->convs[o.conv];
  std::string nm = L.name;
  if (!(site & 1) && o.fwd_group) nm = L.name.substr(0, L.name.rfind('.')) + ".b0|b1a|b2a";
  if ((site & 1) && o.bwd_fused) nm = L.name.substr(0, L.name.rfind('.')) + ".b0|b1a|b2a";
  nm += (site & 1) ? " backward-data" : " forward";
  strncpy(name64, nm.c_str(), 63);
  name64[63] = 0;
  return IVF_OK;
}

extern "C" int ivf_i3d_num_conv_ops(const ivf_i3d_t* net) {
  int n = 0;
  if (net)
    for (const Op& o : net->ops) n += (o.type == Op::CONV);
  return n;
}

extern "C" int ivf_i3d_get_tuning(const ivf_i3d_t* net, int* v) {
  IVF_CHECK_ARG(net && v, "i3d_get_tuning: null pointer");
  int i = 0;
  for (const Op& o : net->ops)
    if (o.type == Op::CONV) { v[i++] = o.var_fwd; v[i++] = o.var_bwd; }
  return IVF_OK;
}

extern "C" int ivf_i3d_set_tuning(ivf_i3d_t* net, const int* v) {
  IVF_CHECK_ARG(net && v, "i3d_set_tuning: null pointer");
  int i = 0;
  for (Op& o : net->ops)
    if (o.type == Op::CONV) { o.var_fwd = v[i++]; o.var_bwd = v[i++]; }
  return IVF_OK;
}

extern "C" int ivf_i3d_autotune(ivf_i3d_t* net, int b, int reps, ivf_stream_t stream) {
  IVF_PROPAGATE(check_ready(net, b));
  IVF_CHECK_ARG(reps >= 1, "i3d_autotune: reps >= 1");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  IVF_CHECK_HIP(hipEventCreate(&e0));
  IVF_CHECK_HIP(hipEventCreate(&e1));
  int rc = IVF_OK;
  // IVF_TUNE_LOG=<file>: every candidate's time per layer and direction (dev measurement aid)
  FILE* tlog = getenv("IVF_TUNE_LOG") ? fopen(getenv("IVF_TUNE_LOG"), "a") : nullptr;
  for (Op& o : net->ops) {
    if (o.type != Op::CONV) continue;
    const ConvLayer& L = net->convs[o.conv];
    for (int dir = 0; dir < 2 && rc == IVF_OK; ++dir) {
      if (dir == 1 && o.bwd_skip) continue;
      if (dir == 0 && o.fwd_skip) continue;
      ivf_conv3d_desc d;
      int ids[96];
      int* slot = dir == 0 ? &o.var_fwd : &o.var_bwd;
      *slot = IVF_CONV_AUTO;
      if (dir == 0) fill_conv_fwd(net, o, b, &d); else fill_conv_bwd(net, o, b, &d);
      int nv = ivf_conv3d_variants(&d, ids, 96);
      float best = 1e30f;
      int best_id = IVF_CONV_AUTO;
      for (int k = 0; k < nv; ++k) {
        d.variant = ids[k];
        auto run = [&]() {
          if (dir == 0)
            return o.fwd_group ? ivf_conv3d(&d, net->act(o.src), net->warena + L.grp_wf_off,
                                            net->warena + L.grp_scale_off, net->warena + L.grp_shift_off, nullptr,
                                            net->act(o.dst), s)
                               : ivf_conv3d(&d, net->act(o.src), net->warena + L.wf_off, net->warena + L.scale_off,
                                            net->warena + L.shift_off, nullptr, net->act(o.dst), s);
          const bool bits = o.bwd_mask && net->bufs[o.src].need_gate;
          if (bits) {
            d.gate_in = net->gatebits(o.src);
            d.gate_in_ld = net->bufs[o.src].C / 8;
            d.gate_in_coff = o.src_coff;
          }
          return ivf_conv3d(&d, net->grad(o.dst), net->warena + (o.bwd_fused ? L.fus_wb_off : L.wb_off), nullptr,
                            nullptr, (o.bwd_mask && !bits) ? net->act(o.src) : nullptr, net->grad(o.src), s);
        };
        if (run() != IVF_OK) continue;          // variant not applicable to this shape
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) (void)run();
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = IVF_ERR_HIP; set_error("autotune: sync failed"); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (tlog)
          fprintf(tlog, "%s %s b=%d variant=%d ms=%.4f gflop=%.1f\n", L.name.c_str(), dir ? "bwd" : "fwd", b, ids[k],
                  ms / reps, (dir && o.bwd_fused ? o.flops_bwd_per_clip : o.flops_per_clip) * b / 1e9);
        // a later candidate must win by 1.5 %: keeps the choice between tied tile shapes stable from run to run
        if (ms < best * (best_id == IVF_CONV_AUTO ? 1.f : 0.985f)) { best = ms; best_id = ids[k]; }
      }
      *slot = best_id;
    }
    if (rc != IVF_OK) break;
  }
  if (tlog) fclose(tlog);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}
