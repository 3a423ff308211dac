// Python bindings of the gfx950 kernel suite and the step-program runtime.
//
// Tensors cross the boundary as raw device addresses (tensor.data_ptr()) and streams as raw
// hipStream_t handles (torch.cuda.current_stream().cuda_stream); shape checking and output
// allocation live in distributed_tensorflow_amd/ops. Every kernel takes a keyword dictionary
// that is copied field by field into the kernel's argument struct, so unspecified fields are
// zero / null.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "comm/p2p_allreduce.h"
#include "comm/update_log.h"
#include "comm/async_shards.h"
#include "comm/link_probe.h"
#include "kernels/conv.h"
#include "kernels/ops.h"
#include "runtime/program.h"

namespace py = pybind11;
using namespace dtf;

namespace {

template <typename T>
T* as_ptr(const py::dict& d, const char* key) {
  if (!d.contains(key)) return nullptr;
  py::object o = d[key];
  if (o.is_none()) return nullptr;
  return reinterpret_cast<T*>(o.cast<uint64_t>());
}

template <typename T>
void get(const py::dict& d, const char* key, T& out) {
  if (d.contains(key) && !d[key].is_none()) out = d[key].cast<T>();
}

BnRef as_bn(const py::dict& d, const char* key) {
  BnRef b{nullptr, nullptr, nullptr, 0.f, 0.f};
  if (!d.contains(key) || d[key].is_none()) return b;
  py::tuple t = d[key].cast<py::tuple>();
  b.sums = reinterpret_cast<const double*>(t[0].cast<uint64_t>());
  b.gamma = reinterpret_cast<const float*>(t[1].cast<uint64_t>());
  b.beta = reinterpret_cast<const float*>(t[2].cast<uint64_t>());
  b.inv_count = t[3].cast<float>();
  b.eps = t[4].cast<float>();
  return b;
}

hipStream_t as_stream(uint64_t s) { return reinterpret_cast<hipStream_t>(s); }

#define P(field, T) a.field = as_ptr<T>(d, #field)
#define V(field) get(d, #field, a.field)

}  // namespace

PYBIND11_MODULE(_hip, m) {
  m.doc() = "gfx950 HIP kernels and step-program runtime";

  m.def("device_count", []() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
  });

  // ---------------------------------------------------------------- convolution
  m.def("conv_fprop", [](const py::dict& d, int ks, uint64_t stream) {
    ConvFpropArgs a{};
    P(in, const float); P(wpk, const float); P(out, float);
    V(B); V(H); V(W); V(Cin); V(Ho); V(Wo); V(Cout);
    V(TH); V(TW); V(tiles_y); V(tiles_x);
    V(pad); V(stride); V(upsample); V(nt_total); V(debug_skip);
    V(in_mode); a.in_bn = as_bn(d, "in_bn");
    V(res_mode); P(res, const float); a.res_bn = as_bn(d, "res_bn");
    V(epi_mode); P(bias, const float); P(mask_src, const float); a.mask_bn = as_bn(d, "mask_bn");
    P(extra, const float); P(out_sums, double); P(dbg_stamps, unsigned long long);
    if (a.stride == 0) a.stride = 1;
    return conv_fprop(a, ks, as_stream(stream));
  });
  m.def("conv_fprop_split", [](const py::dict& d, uint64_t stream) {
    ConvFpropArgs a{};
    P(in, const float); P(wpk, const float); P(out, float);
    V(B); V(H); V(W); V(Cin); V(Ho); V(Wo); V(Cout); V(pad); V(stride); V(upsample);
    V(in_mode); a.in_bn = as_bn(d, "in_bn");
    V(res_mode); P(res, const float); a.res_bn = as_bn(d, "res_bn");
    V(epi_mode); P(mask_src, const float); a.mask_bn = as_bn(d, "mask_bn");
    P(extra, const float); P(out_sums, double);
    if (a.stride == 0) a.stride = 1;
    return conv_fprop_split(a, as_stream(stream));
  });
  m.def("pack_split_weights", [](uint64_t w, uint64_t out, int C, int flip, uint64_t stream) {
    PackArgs a{reinterpret_cast<const float*>(w), reinterpret_cast<float*>(out), 3, C, C, flip};
    pack_split_weights(a, as_stream(stream));
  });
  m.def("packed_split_floats", &packed_split_floats);
  m.def("pack_conv_weights", [](uint64_t w, uint64_t out, int ks, int Cin, int Cout, int flip, uint64_t stream) {
    PackArgs a{reinterpret_cast<const float*>(w), reinterpret_cast<float*>(out), ks, Cin, Cout, flip};
    pack_conv_weights(a, as_stream(stream));
  });
  m.def("packed_weight_floats", &packed_weight_floats);
  m.def("conv_wgrad", [](const py::dict& d, int ks, uint64_t stream) {
    ConvWgradArgs a{};
    P(in, const float); P(dout, const float); P(dw, float);
    V(B); V(H); V(W); V(Cin); V(Ho); V(Wo); V(Cout);
    V(TH); V(TW); V(tiles_y); V(tiles_x);
    V(pad); V(stride); V(in_mode); a.in_bn = as_bn(d, "in_bn");
    V(tiles_per_block); V(dw_slot_stride); V(co_tiles); V(debug_skip); P(dbg_stamps, unsigned long long); P(dw_part, float);
    if (a.stride == 0) a.stride = 1;
    return conv_wgrad(a, ks, as_stream(stream));
  });
  m.def("conv_bwd_fused", [](const py::dict& d, uint64_t stream) {
    ConvBwdFusedArgs a{};
    P(dy, const float); P(x, const float); P(wpk_t, const float); P(g, float); P(dw, float); P(dw_part, float);
    V(dw_slot_stride); a.bn = as_bn(d, "bn"); P(gsums, double); P(extra, const float);
    V(B); V(H); V(W); V(C);
    P(dy_x, const float); a.dy_bn = as_bn(d, "dy_bn"); P(dy_gsums, const double); P(dy_dgamma, float); P(dy_dbeta, float);
    P(dy_res, const float); P(dy_out, float); V(dy_res_mode);
    P(tile_counters, int); P(dbg_stamps, unsigned long long);
    const bool split = d.contains("split") && d["split"].cast<bool>();   // wpk_t holds the split-operand packing
    return split ? conv_bwd_split(a, as_stream(stream)) : conv_bwd_fused(a, as_stream(stream));
  });
  m.def("conv_bwd_fused_copies", [](const py::dict& d) {
    ConvBwdFusedArgs a{};
    P(dy, const float); P(x, const float); P(wpk_t, const float); P(g, float); P(dw, float);
    a.bn = as_bn(d, "bn"); V(B); V(H); V(W); V(C);
    P(dy_x, const float); a.dy_bn = as_bn(d, "dy_bn"); P(dy_gsums, const double);
    P(dy_res, const float); P(dy_out, float); V(dy_res_mode);
    const bool split = d.contains("split") && d["split"].cast<bool>();
    return split ? conv_bwd_split_copies(a) : conv_bwd_fused_copies(a);
  });
  m.def("conv_wgrad_band_copies", [](const py::dict& d, int ks) {
    ConvWgradArgs a{};
    V(B); V(H); V(W); V(Cin); V(Ho); V(Wo); V(Cout); V(pad); V(stride); V(tiles_per_block); V(co_tiles);
    if (a.stride == 0) a.stride = 1;
    return conv_wgrad_band_copies(a, ks);
  });
  m.def("conv_stem_wgrad_copies", [](const py::dict& d) {
    ConvStemArgs a{};
    V(B); P(dy_x, const float);
    return conv_stem_wgrad_copies(a);
  });
  m.def("part_reduce", [](uint64_t table, int n_entries, int total_items, uint64_t slots, long long slot_stride, uint64_t stream) {
    PartReduceArgs a{reinterpret_cast<const PartEntry*>(table), n_entries, reinterpret_cast<float*>(slots), slot_stride};
    part_reduce(a, total_items, as_stream(stream));
  });
  m.attr("PART_ENTRY_BYTES") = (int)sizeof(PartEntry);
  m.def("conv_bwd_fused_supported", &conv_bwd_fused_supported);
  m.def("conv_bwd_split_supported", &conv_bwd_split_supported);
  m.def("conv_stem_fwd", [](const py::dict& d, uint64_t stream) {
    ConvStemArgs a{};
    P(x, const float); P(w, const float); P(y, float); P(sums, double); V(B);
    conv_stem_fwd(a, as_stream(stream));
  });
  m.def("conv_stem_wgrad", [](const py::dict& d, uint64_t stream) {
    ConvStemArgs a{};
    P(x, const float); P(dy, const float); P(dw, float); P(dw_part, float); V(slot_stride); V(B);
    P(dy_x, const float); a.dy_bn = as_bn(d, "dy_bn"); P(dy_gsums, const double); P(dy_dgamma, float); P(dy_dbeta, float);
    conv_stem_wgrad(a, as_stream(stream));
  });
  m.def("conv_stem_supported", &conv_stem_supported);
  m.def("conv5x5_tile", [](const py::dict& d, uint64_t stream) {
    Conv5x5Args a{};
    P(x, const float); P(w, const float); P(bias, const float); P(y, float);
    V(B); V(Cin); V(Cout); V(flip); V(imgs);
    P(pool_dp, const float); P(pool_arg, const unsigned char); P(pool_p, const float);
    P(pool_out, float); P(pool_arg_out, unsigned char);
    const bool split = d.contains("split") && d["split"].cast<bool>();   // w holds the split-operand packing
    if (split) conv5x5_split(a, as_stream(stream));
    else conv5x5_tile(a, as_stream(stream));
  });
  m.def("conv5x5_pack", [](const py::dict& d, uint64_t stream) {
    Conv5x5PackArgs a{};
    P(w, const float); P(out, float); V(Cin); V(Cout);
    conv5x5_pack(a, as_stream(stream));
  });
  m.def("conv5x5_split_pack", [](const py::dict& d, uint64_t stream) {
    Conv5x5PackArgs a{};
    P(w, const float); P(out, float); V(Cin); V(Cout); V(flip);
    conv5x5_split_pack(a, as_stream(stream));
  });
  m.def("conv5x5_split_packed_floats", &conv5x5_split_packed_floats);
  m.def("conv5x5_tile_supported", &conv5x5_tile_supported);
  m.def("conv5x5_wgrad", [](const py::dict& d, uint64_t stream) {
    Conv5x5WgradArgs a{};
    P(x, const float); P(dy, const float); P(dw, float); V(slot_stride); V(B); P(partials, float);
    P(pool_dp, const float); P(pool_arg, const unsigned char); P(pool_p, const float);
    const bool split = d.contains("split") && d["split"].cast<bool>();
    if (split) conv5x5_wgrad_split(a, as_stream(stream));
    else conv5x5_wgrad(a, as_stream(stream));
  });
  m.def("conv5x5_wgrad_parts", &conv5x5_wgrad_parts);
  m.def("partial_sum", [](const py::dict& d, uint64_t stream) {
    PartialSumArgs a{};
    P(y, float); P(ws, const float); V(n); V(parts); V(stride);
    partial_sum(a, as_stream(stream));
  });
  m.def("conv1_pool_fwd", [](const py::dict& d, uint64_t stream) {
    Conv1PoolFwdArgs a{};
    P(x, const float); P(w, const float); P(bias, const float); P(p, float); P(arg, unsigned char); V(B);
    conv1_pool_fwd(a, as_stream(stream));
  });
  m.def("conv1_pool_bwd", [](const py::dict& d, uint64_t stream) {
    Conv1PoolBwdArgs a{};
    P(x, const float); P(dp, const float); P(arg, const unsigned char); P(p, const float); P(dw, float); P(dbias, float);
    V(slot_stride); V(B);
    conv1_pool_bwd(a, as_stream(stream));
  });
  m.attr("BWD_TILE_COUNTERS") = (int)kBwdTileCounters;
  m.def("pack_multi", [](uint64_t params, uint64_t packed, uint64_t table, int n_entries, int total_blocks,
                         uint64_t stream) {
    PackMultiArgs a{reinterpret_cast<const float*>(params), reinterpret_cast<float*>(packed),
                    reinterpret_cast<const PackEntry*>(table), n_entries};
    pack_multi(a, total_blocks, as_stream(stream));
  });
  m.attr("PACK_ENTRY_BYTES") = (int)sizeof(PackEntry);
  m.def("zero_multi", [](const std::vector<std::pair<uint64_t, uint64_t>>& bufs, uint64_t stream) {
    if (bufs.empty() || bufs.size() > (size_t)kZeroMultiMax) throw std::runtime_error("zero_multi: 1..8 buffers");
    ZeroMultiArgs a{};
    a.count = (int)bufs.size();
    for (int i = 0; i < a.count; ++i) {
      if (bufs[i].second & 15) throw std::runtime_error("zero_multi: sizes must be multiples of 16 bytes");
      a.ptr[i] = reinterpret_cast<void*>(bufs[i].first);
      a.chunks[i] = bufs[i].second / 16;
    }
    zero_multi(a, as_stream(stream));
  });

  // ---------------------------------------------------------------- batch norm
  m.def("bn_stats", [](const py::dict& d, uint64_t stream) {
    BnStatsArgs a{};
    P(x, const float); P(sums, double); V(n_rows); V(C);
    bn_stats(a, as_stream(stream));
  });
  m.def("bn_apply", [](const py::dict& d, uint64_t stream) {
    BnApplyArgs a{};
    P(x, const float); P(y, float); a.bn = as_bn(d, "bn"); V(n_rows); V(C); V(relu);
    bn_apply(a, as_stream(stream));
  });
  m.def("bn_relu_bwd_reduce", [](const py::dict& d, uint64_t stream) {
    BnReluBwdReduceArgs a{};
    P(dy, const float); P(x, const float); P(g, float); a.bn = as_bn(d, "bn"); P(gsums, double);
    V(n_rows); V(C); V(relu);
    bn_relu_bwd_reduce(a, as_stream(stream));
  });
  m.def("bn_bwd_apply", [](const py::dict& d, uint64_t stream) {
    BnBwdApplyArgs a{};
    P(g, const float); P(x, const float); P(dx, float); a.bn = as_bn(d, "bn"); P(gsums, const double);
    P(dgamma, float); P(dbeta, float); V(n_rows); V(C); V(res_mode); P(res, const float); V(H); V(W);
    bn_bwd_apply(a, as_stream(stream));
  });

  // ---------------------------------------------------------------- head / loss
  m.def("head_fwd", [](const py::dict& d, uint64_t stream) {
    HeadFwdArgs a{};
    P(x, const float); a.bn = as_bn(d, "bn"); P(fc_w, const float); P(fc_b, const float); P(labels, const int);
    P(feat, float); P(logits, float); P(dlogits, float); P(loss, float); P(correct, int);
    V(B); V(HW); V(C); V(K); V(slots);
    head_fwd(a, as_stream(stream));
  });
  m.def("head_bwd", [](const py::dict& d, uint64_t stream) {
    HeadBwdArgs a{};
    P(x, const float); a.bn = as_bn(d, "bn"); P(fc_w, const float); P(feat, const float); P(dlogits, const float);
    P(g, float); P(gsums, double); P(d_fc_w, float); P(d_fc_b, float);
    V(B); V(HW); V(C); V(K);
    head_bwd(a, as_stream(stream));
  });
  m.def("softmax_xent", [](const py::dict& d, uint64_t stream) {
    SoftmaxXentArgs a{};
    P(logits, const float); P(labels, const int); P(dlogits, float); P(loss, float); P(correct, int); V(B); V(K);
    softmax_xent(a, as_stream(stream));
  });
  m.def("fc_head", [](const py::dict& d, uint64_t stream) {
    FcHeadArgs a{};
    P(h_raw, const float); P(b1, const float); P(w2, const float); P(b2, const float); P(labels, const int);
    P(logits, float); P(loss, float); P(correct, int); P(dh, float); P(dw2, float); P(db2, float); P(db1, float);
    V(B); V(H); V(K); V(keep_prob); V(seed); V(offset); P(offset_dev, const unsigned long long); V(max_blocks);
    fc_head(a, as_stream(stream));
  });
  m.def("top1_count", [](const py::dict& d, uint64_t stream) {
    Top1Args a{};
    P(logits, const float); P(labels, const int); P(correct, int); V(B); V(K);
    top1_count(a, as_stream(stream));
  });

  // ---------------------------------------------------------------- optimizers / flat
  m.def("fused_adam", [](const py::dict& d, uint64_t stream) {
    FlatOptArgs a{};
    P(p, float); P(g, const float); P(m, float); P(v, float); P(hyper, const float); V(n);
    fused_adam(a, as_stream(stream));
  });
  m.def("fused_momentum_sgd", [](const py::dict& d, uint64_t stream) {
    FlatOptArgs a{};
    P(p, float); P(g, const float); P(m, float); P(hyper, const float); V(n);
    fused_momentum_sgd(a, as_stream(stream));
  });
  m.def("set_scalars", [](uint64_t f_dst, const std::vector<float>& f, uint64_t u_dst, const std::vector<uint64_t>& u,
                          uint64_t stream) {
    if (current_recorder()) throw std::runtime_error("set_scalars carries per-step values and cannot be recorded into a program");
    if (f.size() > (size_t)kSetScalarsMax || u.size() > (size_t)kSetScalarsMax) throw std::runtime_error("set_scalars: at most 8 values of each kind");
    SetScalarsArgs a{};
    a.f_dst = reinterpret_cast<float*>(f_dst);
    a.u_dst = reinterpret_cast<unsigned long long*>(u_dst);
    a.nf = (int)f.size();
    a.nu = (int)u.size();
    for (int i = 0; i < a.nf; ++i) a.f[i] = f[i];
    for (int i = 0; i < a.nu; ++i) a.u[i] = u[i];
    set_scalars(a, as_stream(stream));
  });
  m.def("flat_scale", [](const py::dict& d, uint64_t stream) {
    FlatScaleArgs a{};
    P(x, float); P(scale_dev, const float); a.scale = 1.f; V(scale); V(n);
    flat_scale(a, as_stream(stream));
  });
  m.def("flat_axpy", [](const py::dict& d, uint64_t stream) {
    AxpyArgs a{};
    P(y, float); P(x, const float); V(alpha); V(n);
    flat_axpy(a, as_stream(stream));
  });
  m.def("slot_reduce", [](const py::dict& d, uint64_t stream) {
    SlotReduceArgs a{};
    P(y, float); P(slots, float); V(n); V(slot_stride);
    slot_reduce(a, as_stream(stream));
  });
  // ---------------------------------------------------------------- peer-to-peer collective
  m.def("p2p_allreduce", [](const py::dict& d, int blocks, uint64_t stream) {
    P2PAllReduceArgs a{};
    auto xb = d["xbuf"].cast<std::vector<uint64_t>>();
    auto fl = d["flags"].cast<std::vector<uint64_t>>();
    auto w = d["weight"].cast<std::vector<float>>();
    V(rank); V(world); V(n); V(seq); V(timeout_ticks);
    if ((int)xb.size() != a.world || (int)fl.size() != a.world || (int)w.size() != a.world || a.world > kMaxPeers)
      throw std::runtime_error("p2p_allreduce: per-rank lists must have `world` entries");
    for (int p = 0; p < a.world; ++p) {
      a.xbuf[p] = reinterpret_cast<float*>(xb[p]);
      a.flags[p] = reinterpret_cast<uint32_t*>(fl[p]);
      a.weight[p] = w[p];
    }
    P(data, float); P(error, int);
    p2p_allreduce(a, blocks, as_stream(stream));
  });
  m.attr("P2P_FLAG_BLOCK_BYTES") = (uint64_t)kFlagBlockBytes;
  // ---------------------------------------------------------------- update log (non-blocking aggregation)
  {
    auto fill = [](const py::dict& d) {
      UlogArgs a{};
      auto ctl = d["ctl"].cast<std::vector<uint64_t>>();
      auto stage = d["stage"].cast<std::vector<uint64_t>>();
      auto red = d["red"].cast<std::vector<uint64_t>>();
      V(rank); V(world); V(n);
      if ((int)ctl.size() != a.world || (int)stage.size() != a.world || (int)red.size() != a.world || a.world > kUlogMaxPeers)
        throw std::runtime_error("update log: per-rank lists must have `world` (at most 8) entries");
      for (int p = 0; p < a.world; ++p) {
        a.ctl[p] = reinterpret_cast<unsigned long long*>(ctl[p]);
        a.stage[p] = reinterpret_cast<float*>(stage[p]);
        a.red[p] = reinterpret_cast<float*>(red[p]);
      }
      P(grads, const float); P(p, float); P(m, float); P(v, float); P(status, unsigned long long);
      V(mode); V(quorum); V(abort_at); V(scale); V(weight); V(max_version); V(max_staleness); V(opt);
      V(lr0); V(decay); V(decay_steps); V(beta1); V(beta2); V(eps); V(momentum); V(weight_decay); V(timeout_ticks);
      return a;
    };
    m.def("ulog_step", [fill](const py::dict& d, int blocks, uint64_t stream) { ulog_step(fill(d), blocks, as_stream(stream)); });
    m.def("ulog_close", [fill](const py::dict& d, int blocks, uint64_t stream) { ulog_close(fill(d), blocks, as_stream(stream)); });
    m.def("ulog_init_ctl", [](uint64_t ctl, long long u0, uint64_t stream) {
      ulog_init_ctl(reinterpret_cast<unsigned long long*>(ctl), u0, as_stream(stream));
    });
    m.def("ulog_mark_dead", [](uint64_t ctl, unsigned mask, uint64_t stream) {
      ulog_mark_dead(reinterpret_cast<unsigned long long*>(ctl), mask, as_stream(stream));
    });
    m.def("ulog_peek", [](uint64_t ctl, int first, int count) {
      std::vector<unsigned long long> out(count > 0 ? count : 0);
      ulog_peek(reinterpret_cast<const unsigned long long*>(ctl), first, count, out.data());
      return out;
    });
    m.def("ulog_poke", [](uint64_t ctl, int word, unsigned long long value) {
      ulog_poke(reinterpret_cast<unsigned long long*>(ctl), word, value);
    });
    m.def("ulog_poke_async", [](uint64_t ctl, int word, unsigned long long value, long long delay_ticks, uint64_t stream) {
      ulog_poke_async(reinterpret_cast<unsigned long long*>(ctl), word, value, delay_ticks, as_stream(stream));
    });
    // ---- asynchronous parameter shards
    auto afill = [](const py::dict& d) {
      AsyncArgs a{};
      auto ctl = d["ctl"].cast<std::vector<uint64_t>>();
      auto shard = d["shard"].cast<std::vector<uint64_t>>();
      auto inbox = d["inbox"].cast<std::vector<uint64_t>>();
      V(rank); V(world); V(shard_floats);
      if ((int)ctl.size() != a.world || (int)shard.size() != a.world || (int)inbox.size() != a.world || a.world > kAsyncMaxPeers)
        throw std::runtime_error("async shards: per-rank lists must have `world` (at most 8) entries");
      for (int p = 0; p < a.world; ++p) {
        a.ctl[p] = reinterpret_cast<unsigned long long*>(ctl[p]);
        a.shard[p] = reinterpret_cast<float*>(shard[p]);
        a.inbox[p] = reinterpret_cast<float*>(inbox[p]);
      }
      P(params, float); P(grads, const float); P(m, float); P(v, float); P(status, unsigned long long);
      V(tau); V(opt); V(lr0); V(decay); V(decay_steps); V(beta1); V(beta2); V(eps); V(momentum); V(weight_decay);
      V(timeout_ticks); V(push_wait_ticks);
      return a;
    };
    m.def("async_pull", [afill](const py::dict& d, int blocks, uint64_t stream) { async_pull(afill(d), blocks, as_stream(stream)); });
    m.def("async_push", [afill](const py::dict& d, int blocks, uint64_t stream) { async_push(afill(d), blocks, as_stream(stream)); });
    m.def("async_service", [afill](const py::dict& d, int blocks, uint64_t stream) { async_service(afill(d), blocks, as_stream(stream)); });
    m.attr("ASYNC_CTL_BYTES") = (uint64_t)kAsyncCtlBytes;
    m.attr("ASYNC_HIST_BINS") = (int)kAsyncHistBins;
    m.attr("AW_VERSION") = (int)AW_VERSION; m.attr("AW_ERROR") = (int)AW_ERROR; m.attr("AW_STATS") = (int)AW_STATS;
    m.attr("AW_DOORBELL") = (int)AW_DOORBELL;
    m.attr("ULOG_CTL_BYTES") = (uint64_t)kUlogCtlBytes;
    m.attr("ULOG_RING") = (int)kUlogRing;
    m.attr("ULOG_WORD_BYTES") = (int)(kUlogWordStride * sizeof(unsigned long long));
    m.attr("UW_OPEN") = (int)UW_OPEN; m.attr("UW_ABORT") = (int)UW_ABORT; m.attr("UW_VERSION") = (int)UW_VERSION;
    m.attr("UW_ERROR") = (int)UW_ERROR; m.attr("UW_APPLIED") = (int)UW_APPLIED; m.attr("UW_STATS") = (int)UW_STATS;
    m.attr("UW_HIST_TAG") = (int)UW_HIST_TAG; m.attr("UW_HIST_AUX") = (int)UW_HIST_AUX;
  }
  m.def("link_probe", [](const py::dict& d, uint64_t stream) {
    LinkProbeArgs a{};
    P(own_ctl, unsigned long long); P(peer_ctl, unsigned long long); P(own_buf, float); P(peer_buf, float);
    P(counter, unsigned long long); P(result, unsigned long long);
    V(payload_floats); V(iters); V(cas_iters); V(role); V(timeout_ticks);
    link_probe(a, as_stream(stream));
  });
  m.attr("LINK_PROBE_CTL_BYTES") = (uint64_t)kLinkProbeCtlBytes;
  m.attr("LP_COUNTER") = (int)LP_COUNTER;
  m.def("ipc_alloc", [](uint64_t bytes, bool fine) { return (uint64_t)ipc_alloc(bytes, fine); });
  m.def("ipc_free", [](uint64_t p) { ipc_free(reinterpret_cast<void*>(p)); });
  m.def("ipc_export", [](uint64_t p) { return py::bytes(ipc_export(reinterpret_cast<void*>(p))); });
  m.def("ipc_import", [](const py::bytes& h) { return (uint64_t)ipc_import(std::string(h)); });
  m.def("ipc_release", [](uint64_t p) { ipc_release(reinterpret_cast<void*>(p)); });
  m.def("memset", [](uint64_t dst, int value, uint64_t bytes, uint64_t stream) {
    memset_async(reinterpret_cast<void*>(dst), value, bytes, as_stream(stream));
  });
  m.def("memcpy_d2d", [](uint64_t dst, uint64_t src, uint64_t bytes, uint64_t stream) {
    memcpy_async(reinterpret_cast<void*>(dst), reinterpret_cast<const void*>(src), bytes, as_stream(stream));
  });

  // ---------------------------------------------------------------- elementwise & friends
  m.def("add_relu_fwd", [](const py::dict& d, uint64_t stream) {
    AddReluArgs a{};
    P(a, const float); P(b, const float); P(out, float); V(n); V(relu_first);
    add_relu_fwd(a, as_stream(stream));
  });
  m.def("add_relu_bwd", [](const py::dict& d, uint64_t stream) {
    AddReluBwdArgs a{};
    P(dout, const float); P(a, const float); P(b, const float); P(da, float); P(db, float); V(n); V(relu_first);
    add_relu_bwd(a, as_stream(stream));
  });
  m.def("relu_bwd", [](const py::dict& d, uint64_t stream) {
    ReluMaskArgs a{};
    P(dy, const float); P(y, const float); P(dx, float); V(n);
    relu_bwd(a, as_stream(stream));
  });
  m.def("shortcut_a_fwd", [](const py::dict& d, uint64_t stream) {
    ShortcutAArgs a{};
    P(x, const float); P(y, float); V(B); V(H); V(W); V(C);
    shortcut_a_fwd(a, as_stream(stream));
  });
  m.def("shortcut_a_bwd", [](const py::dict& d, uint64_t stream) {
    ShortcutABwdArgs a{};
    P(dy, const float); P(dx, float); V(B); V(H); V(W); V(C); V(accumulate);
    shortcut_a_bwd(a, as_stream(stream));
  });
  m.def("maxpool2x2_fwd", [](const py::dict& d, uint64_t stream) {
    MaxPoolArgs a{};
    P(x, const float); P(y, float); P(argmax, unsigned char); V(B); V(H); V(W); V(C);
    maxpool2x2_fwd(a, as_stream(stream));
  });
  m.def("maxpool2x2_bwd", [](const py::dict& d, uint64_t stream) {
    MaxPoolBwdArgs a{};
    P(dy, const float); P(argmax, const unsigned char); P(y, const float); P(dx, float);
    V(B); V(H); V(W); V(C); V(relu_mask);
    maxpool2x2_bwd(a, as_stream(stream));
  });
  m.def("dropout", [](const py::dict& d, uint64_t stream) {
    DropoutArgs a{};
    P(x, const float); P(y, float); V(n); V(keep_prob); V(seed); V(offset); P(offset_dev, const unsigned long long);
    dropout_apply(a, as_stream(stream));
  });
  m.def("gemm_f32", [](const py::dict& d, uint64_t stream) {
    GemmArgs a{};
    P(A, const float); P(B, const float); P(C, float); P(bias, const float);
    V(M); V(N); V(K); V(lda); V(ldb); V(ldc); V(trans_a); V(trans_b); V(relu); V(accumulate); V(variant); V(splits);
    gemm_f32(a, as_stream(stream));
  });
  m.def("col_sum", [](const py::dict& d, uint64_t stream) {
    ColSumArgs a{};
    P(x, const float); P(out, float); V(n_rows); V(C); P(relu_of, const float); P(masked, float);
    col_sum(a, as_stream(stream));
  });
  m.def("tf_histogram", [](const py::dict& d, uint64_t stream) {
    HistogramArgs a{};
    P(x, const float); V(n); P(limits, const double); V(nb); P(counts, unsigned long long); P(moments, double); P(minmax, int);
    tf_histogram(a, as_stream(stream));
  });
  m.def("zero_count", [](const py::dict& d, uint64_t stream) {
    ZeroFracArgs a{};
    P(x, const float); P(zeros, unsigned long long); V(n);
    zero_count(a, as_stream(stream));
  });
  m.def("crop_whiten", [](const py::dict& d, uint64_t stream) {
    CropWhitenArgs a{};
    P(src, const float); P(index, const int); P(offs, const int); P(out, float);
    V(B); V(HP); V(WP); V(H); V(W); V(C); V(whiten);
    crop_whiten(a, as_stream(stream));
  });

  // Short-circuit reference handed to every training kernel launched or recorded afterwards from
  // this thread (0, 0 disables it).
  m.def("set_abort_ref", [](uint64_t stop, uint64_t version) {
    if ((stop == 0) != (version == 0)) throw std::runtime_error("set_abort_ref: both words or neither");
    current_abort_ref().stop = reinterpret_cast<const unsigned long long*>(stop);
    current_abort_ref().version = reinterpret_cast<const unsigned long long*>(version);
  });

  // ---------------------------------------------------------------- step programs
  py::class_<Program>(m, "Program")
      .def(py::init<>())
      .def("begin_record", [](Program& p) {
        if (current_recorder()) throw std::runtime_error("a Program is already recording on this thread");
        current_recorder() = &p;
      })
      .def("end_record", [](Program& p) {
        if (current_recorder() != &p) throw std::runtime_error("this Program is not recording");
        current_recorder() = nullptr;
      })
      .def("mark_segment", &Program::mark_segment)
      .def("mark_phase", &Program::mark_phase)
      .def("phase_names", [](const Program& p) {
        std::vector<std::string> v;
        for (auto& ph : p.phases) v.push_back(ph.second);
        return v;
      })
      .def("run_traced", [](const Program& p, uint64_t stream, int seg, uint64_t base_event) {
        std::vector<Program::TraceEvent> evs;
        {
          py::gil_scoped_release rel;
          evs = p.run_traced(as_stream(stream), seg, reinterpret_cast<hipEvent_t>(base_event));
        }
        py::list out;
        for (auto& e : evs) out.append(py::make_tuple(e.name, e.lane, e.phase, e.ts_us, e.dur_us, e.blocks, e.threads));
        return out;
      }, py::arg("stream"), py::arg("seg") = -1, py::arg("base_event") = 0)
      .def("set_lane", &Program::set_lane)
      .def("sync_lanes", &Program::sync_lanes)
      .def_property_readonly("num_segments", &Program::num_segments)
      .def_property_readonly("num_records", [](const Program& p) { return p.records.size(); })
      .def("record_names", [](const Program& p) {
        std::vector<std::string> v;
        for (auto& r : p.records) v.push_back(r.name);
        return v;
      })
      .def("segment_bounds", [](const Program& p) { return p.seg_begin; })
      .def("run", [](const Program& p, uint64_t stream, int seg) {
        py::gil_scoped_release rel;
        p.run(as_stream(stream), seg);
      }, py::arg("stream"), py::arg("seg") = -1)
      .def("instantiate", [](Program& p, uint64_t stream, int seg) { p.instantiate(as_stream(stream), seg); },
           py::arg("stream"), py::arg("seg") = -1)
      .def("instantiated", &Program::instantiated, py::arg("seg") = -1)
      .def("replay", [](const Program& p, uint64_t stream, int seg) {
        py::gil_scoped_release rel;
        p.replay(as_stream(stream), seg);
      }, py::arg("stream"), py::arg("seg") = -1);
}
