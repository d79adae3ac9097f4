// PyTorch custom operators (namespace `psvae`) over the C ABI of libpsvae_hip.so: the operators of
// the training step's four kernel families - matrix product, message-passing aggregation (forward
// and backward), pair combination (forward and backward) and the fused link loss - registered with
// torch.library so that they are visible as torch.ops.psvae.* (profiler, graph tools) and called
// without a Python-side pointer marshalling layer.  Each operator checks device / dtype / row
// layout, takes PyTorch's current HIP stream, owns the per-stream workspace its entry point needs,
// and turns a non-zero status into a c10::Error carrying psvae_last_error().  Host code only: the
// kernels live in libpsvae_hip.so (include/psvae_hip.h).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <initializer_list>
#include <map>
#include <mutex>
#include <utility>

#include "../../include/psvae_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

// Every operator runs on ONE GPU, the device of its operands: the C ABI launches on the current
// device and keys its per-device state (the fixed scratch of the no-alloc entry point, the raised
// LDS limit of the tall-K kernel) by it, while the stream and the workspace below come from the
// tensors' device.  So each operator first checks that all its tensors share a device and makes
// that device current for the duration of the call.
const Tensor *p(const Tensor &t) { return &t; }
const Tensor *p(const OptTensor &t) { return t.has_value() && t->defined() ? &*t : nullptr; }

c10::Device same_device(const char *fn, std::initializer_list<const Tensor *> tensors) {
    const Tensor *first = nullptr;
    for (const Tensor *t : tensors) {
        if (t == nullptr) continue;
        TORCH_CHECK(t->is_cuda(), fn, ": psvae operators need GPU tensors (there is no CPU fallback)");
        if (first == nullptr) first = t;
        TORCH_CHECK(t->device() == first->device(), fn, ": operands on different devices (", first->device(), " and ",
                    t->device(), ")");
    }
    TORCH_CHECK(first != nullptr, fn, ": no tensor operand");
    return first->device();
}
#define PSVAE_ON_DEVICE(fn, ...) c10::DeviceGuard device_guard_(same_device(fn, {__VA_ARGS__}))

void *cur_stream(const Tensor &t) {
    TORCH_CHECK(t.is_cuda(), "psvae operators need GPU tensors (there is no CPU fallback)");
    return (void *)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

void check(int rc, const char *name) { TORCH_CHECK(rc == 0, name, " failed with status ", rc, ": ", psvae_last_error()); }

const float *f32(const Tensor &t, const char *what) {
    TORCH_CHECK(t.is_cuda(), what, ": psvae operators need GPU tensors (there is no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, what, ": expected float32, got ", t.scalar_type());
    return t.const_data_ptr<float>();
}
const float *f32(const OptTensor &t, const char *what) { return t.has_value() && t->defined() ? f32(*t, what) : nullptr; }
float *f32_mut(const Tensor &t, const char *what) { return const_cast<float *>(f32(t, what)); }
float *f32_mut(const OptTensor &t, const char *what) { return const_cast<float *>(f32(t, what)); }

const int32_t *i32(const Tensor &t, const char *what) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kInt && t.is_contiguous(), what, ": expected a contiguous int32 GPU tensor");
    return t.const_data_ptr<int32_t>();
}
const uint8_t *u8(const Tensor &t, const char *what) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kByte && t.is_contiguous(), what, ": expected a contiguous uint8 GPU tensor");
    return t.const_data_ptr<uint8_t>();
}

// row stride (in elements) of a 2-D tensor whose last dimension is contiguous
int64_t row_stride(const Tensor &t, const char *what) {
    TORCH_CHECK(t.dim() == 2 && (t.size(1) <= 1 || t.stride(1) == 1), what, ": need a 2-D tensor with contiguous rows");
    return t.size(0) > 1 ? t.stride(0) : std::max<int64_t>(t.stride(0), t.size(1));
}

// A scratch buffer per (device, stream), grown on demand.  Kernels that reuse it are ordered by the
// stream they are enqueued on; two streams never share one.
Tensor workspace(size_t nbytes, const Tensor &like, void *stream) {
    static std::mutex mu;
    static std::map<std::pair<int, void *>, Tensor> cache;
    std::lock_guard<std::mutex> lock(mu);
    Tensor &buf = cache[{like.device().index(), stream}];
    if (!buf.defined() || (size_t)buf.numel() < nbytes)
        buf = at::empty({(int64_t)std::max<size_t>(nbytes, 1u << 20)}, like.options().dtype(at::kByte));
    return buf;
}

// C (+)= act(A B + bias): see psvae_gemm_f32 (operands addressed through element strides)
void gemm(int64_t M, int64_t N, int64_t K, const Tensor &A, const OptTensor &amask, int64_t sa_m, int64_t sa_k,
          const Tensor &B, int64_t sb_k, int64_t sb_n, Tensor C, int64_t ldc, const OptTensor &bias, OptTensor rowsum,
          const OptTensor &cmask, bool relu, bool accumulate, int64_t split_k) {
    PSVAE_ON_DEVICE("psvae::gemm", p(A), p(amask), p(B), p(C), p(bias), p(rowsum), p(cmask));
    void *s = cur_stream(C);
    const size_t nbytes = psvae_gemm_f32_workspace_bytes((int)M, (int)N, (int)K, (int)split_k);
    void *ws = nullptr;
    Tensor hold;
    if (nbytes) hold = workspace(nbytes, C, s), ws = hold.data_ptr();
    check(psvae_gemm_f32((int)M, (int)N, (int)K, f32(A, "gemm A"), f32(amask, "gemm amask"), sa_m, sa_k, f32(B, "gemm B"), sb_k,
                         sb_n, f32_mut(C, "gemm C"), ldc, f32(bias, "gemm bias"), f32_mut(rowsum, "gemm rowsum"),
                         f32(cmask, "gemm cmask"), relu ? PSVAE_ACT_RELU : PSVAE_ACT_NONE, accumulate ? 1 : 0, (int)split_k, ws,
                         nbytes, s),
          "psvae_gemm_f32");
}

// the same product with the ReLU decisions as bits: see psvae_gemm_f32_bits (bit tensors are int64 storage)
void gemm_bits(int64_t M, int64_t N, int64_t K, const Tensor &A, int64_t sa_m, int64_t sa_k, const Tensor &B, int64_t sb_k,
               int64_t sb_n, Tensor C, int64_t ldc, const OptTensor &bias, bool relu, OptTensor relu_bits,
               const OptTensor &gate_bits) {
    PSVAE_ON_DEVICE("psvae::gemm_bits", p(A), p(B), p(C), p(bias), p(relu_bits), p(gate_bits));
    const char *fn = "psvae_gemm_f32_bits";
    const size_t words = psvae_gemm_f32_bits_words((int)M, (int)N);
    auto bits = [&](const OptTensor &t) -> uint64_t * {
        if (!t.has_value() || !t->defined()) return nullptr;
        TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kLong && t->is_contiguous() && (size_t)t->numel() >= words, fn,
                    ": a bit array is a contiguous int64 GPU tensor of at least ", words, " words");
        return reinterpret_cast<uint64_t *>(t->data_ptr<int64_t>());
    };
    check(psvae_gemm_f32_bits((int)M, (int)N, (int)K, f32(A, fn), sa_m, sa_k, f32(B, fn), sb_k, sb_n, f32_mut(C, fn), ldc,
                              f32(bias, fn), relu ? PSVAE_ACT_RELU : PSVAE_ACT_NONE, bits(relu_bits), bits(gate_bits),
                              cur_stream(C)), fn);
}

void gine_aggregate_fwd(const Tensor &x, const Tensor &table, const Tensor &row_ptr, const Tensor &col, const Tensor &cls,
                        Tensor out) {
    PSVAE_ON_DEVICE("psvae::gine_aggregate_fwd", p(x), p(table), p(row_ptr), p(col), p(cls), p(out));
    const char *fn = "psvae_gine_aggregate_fwd";
    check(psvae_gine_aggregate_fwd((int)x.size(0), (int)x.size(1), (int)table.size(0), f32(x, fn), (int)row_stride(x, fn),
                                   f32(table, fn), i32(row_ptr, fn), i32(col, fn), u8(cls, fn), f32_mut(out, fn),
                                   (int)row_stride(out, fn), cur_stream(x)),
          fn);
}

void gine_aggregate_bwd(const Tensor &x, const Tensor &table, const Tensor &g, const Tensor &t_row_ptr, const Tensor &t_col,
                        const Tensor &t_cls, const OptTensor &add, const OptTensor &seg_add, const OptTensor &node_graph,
                        Tensor g_x, Tensor g_table, bool accumulate_table) {
    PSVAE_ON_DEVICE("psvae::gine_aggregate_bwd", p(x), p(table), p(g), p(t_row_ptr), p(t_col), p(t_cls), p(add), p(seg_add),
                    p(node_graph), p(g_x), p(g_table));
    const char *fn = "psvae_gine_aggregate_bwd";
    const int V = (int)x.size(0), H = (int)x.size(1), C = (int)table.size(0);
    void *s = cur_stream(x);
    const size_t nbytes = psvae_gine_aggregate_bwd_workspace_bytes(V, H, C);
    void *ws = nullptr;
    Tensor hold;
    if (nbytes) hold = workspace(nbytes, x, s), ws = hold.data_ptr();
    const bool has_add = add.has_value() && add->defined();
    const bool has_seg = seg_add.has_value() && seg_add->defined();
    TORCH_CHECK(!has_seg || (node_graph.has_value() && node_graph->defined() && node_graph->numel() == V), fn,
                ": seg_add needs node_graph [V]");
    check(psvae_gine_aggregate_bwd_graph_add(V, H, C, f32(x, fn), (int)row_stride(x, fn), f32(table, fn), f32(g, fn),
                                             (int)row_stride(g, fn), i32(t_row_ptr, fn), i32(t_col, fn), u8(t_cls, fn),
                                             f32(add, fn), has_add ? (int)row_stride(*add, fn) : H, f32(seg_add, fn),
                                             has_seg ? (int)row_stride(*seg_add, fn) : H,
                                             has_seg ? i32(*node_graph, fn) : nullptr, f32_mut(g_x, fn),
                                             (int)row_stride(g_x, fn), f32_mut(g_table, fn), accumulate_table ? 1 : 0, ws, nbytes,
                                             s),
          fn);
}

// raw storage of the mask-bit array: one word per float4 column of a node, of the width the library reports
void *mask_ptr(const Tensor &mask, int64_t V, int64_t H, int64_t C, const char *fn) {
    const int64_t words = V * (H / 4);
    const bool narrow = words > 0 && psvae_gine_mask_bytes((int)V, (int)H, (int)C) == (size_t)words * 2;
    const auto want = narrow ? at::kShort : at::kInt;
    TORCH_CHECK(mask.is_cuda() && (words == 0 || mask.scalar_type() == want) && mask.is_contiguous() && mask.numel() >= words, fn,
                ": mask bits are a contiguous ", (narrow ? "int16" : "int32"), " GPU tensor of at least V * H / 4 words");
    return mask.data_ptr();
}

void gine_aggregate_fwd_bits(const Tensor &x, const Tensor &table, const Tensor &row_ptr, const Tensor &col, const Tensor &cls,
                             Tensor out, Tensor mask) {
    PSVAE_ON_DEVICE("psvae::gine_aggregate_fwd_bits", p(x), p(table), p(row_ptr), p(col), p(cls), p(out), p(mask));
    const char *fn = "psvae_gine_aggregate_fwd_bits";
    const int V = (int)x.size(0), H = (int)x.size(1), C = (int)table.size(0);
    check(psvae_gine_aggregate_fwd_bits(V, H, C, f32(x, fn), (int)row_stride(x, fn), f32(table, fn), i32(row_ptr, fn), i32(col, fn),
                                        u8(cls, fn), f32_mut(out, fn), (int)row_stride(out, fn), mask_ptr(mask, V, H, C, fn),
                                        cur_stream(x)),
          fn);
}

// slabs (optional, float [psvae_gine_aggregate_bwd_slabs(V, H, C), C, H] contiguous): the table-gradient partials
// are left there and g_table is not touched; gine_table_reduce sums the slabs of several calls at once.
void gine_aggregate_bwd_bits(const Tensor &mask, int64_t C, const Tensor &g, const Tensor &t_row_ptr, const Tensor &t_col,
                             const Tensor &t_cls, const OptTensor &add, const OptTensor &seg_add, const OptTensor &node_graph,
                             Tensor g_x, Tensor g_table, bool accumulate_table, const OptTensor &slabs) {
    PSVAE_ON_DEVICE("psvae::gine_aggregate_bwd_bits", p(mask), p(g), p(t_row_ptr), p(t_col), p(t_cls), p(add), p(seg_add),
                    p(node_graph), p(g_x), p(g_table), p(slabs));
    const char *fn = "psvae_gine_aggregate_bwd_bits";
    const int V = (int)g.size(0), H = (int)g.size(1);
    TORCH_CHECK(g_table.size(0) == C && g_table.size(1) == H && g_table.is_contiguous(), fn, ": g_table must be [C, H]");
    void *s = cur_stream(g);
    const size_t nbytes = psvae_gine_aggregate_bwd_workspace_bytes(V, H, (int)C);
    void *ws = nullptr;
    Tensor hold;
    const bool partials_only = slabs.has_value() && slabs->defined();
    if (partials_only) {
        TORCH_CHECK(slabs->is_cuda() && slabs->scalar_type() == at::kFloat && slabs->is_contiguous() &&
                        (size_t)slabs->numel() * sizeof(float) >= nbytes, fn, ": slabs must be contiguous float32 [slabs, C, H]");
        ws = slabs->data_ptr();
    } else if (nbytes) {
        hold = workspace(nbytes, g, s), ws = hold.data_ptr();
    }
    const bool has_add = add.has_value() && add->defined();
    const bool has_seg = seg_add.has_value() && seg_add->defined();
    TORCH_CHECK(!has_seg || (node_graph.has_value() && node_graph->defined() && node_graph->numel() == V), fn,
                ": seg_add needs node_graph [V]");
    check(psvae_gine_aggregate_bwd_bits(V, H, (int)C, mask_ptr(mask, V, H, C, fn), f32(g, fn), (int)row_stride(g, fn),
                                        i32(t_row_ptr, fn), i32(t_col, fn), u8(t_cls, fn), f32(add, fn),
                                        has_add ? (int)row_stride(*add, fn) : H, f32(seg_add, fn),
                                        has_seg ? (int)row_stride(*seg_add, fn) : H, has_seg ? i32(*node_graph, fn) : nullptr,
                                        f32_mut(g_x, fn), (int)row_stride(g_x, fn), f32_mut(g_table, fn),
                                        partials_only ? PSVAE_TABLE_PARTIALS_ONLY : (accumulate_table ? 1 : 0), ws, nbytes, s),
          fn);
}

void gine_table_reduce(const Tensor &slabs, Tensor g_table, bool accumulate) {
    PSVAE_ON_DEVICE("psvae::gine_table_reduce", p(slabs), p(g_table));
    const char *fn = "psvae_gine_table_reduce";
    TORCH_CHECK(slabs.dim() == 3 && slabs.is_contiguous() && g_table.dim() == 2 && g_table.is_contiguous() &&
                    slabs.size(1) == g_table.size(0) && slabs.size(2) == g_table.size(1), fn, ": slabs [n, C, H], g_table [C, H]");
    check(psvae_gine_table_reduce((int)slabs.size(0), (int)g_table.size(0), (int)g_table.size(1), f32(slabs, fn), f32_mut(g_table, fn),
                                  accumulate ? 1 : 0, cur_stream(slabs)),
          fn);
}

void graph_rows_gated(const Tensor &q, const Tensor &node_graph, const Tensor &gate_bits, Tensor out) {
    PSVAE_ON_DEVICE("psvae::graph_rows_gated", p(q), p(node_graph), p(gate_bits), p(out));
    const char *fn = "psvae_graph_rows_gated";
    const int V = (int)out.size(0), N = (int)out.size(1);
    TORCH_CHECK(q.dim() == 2 && q.size(1) == N && node_graph.numel() == V, fn, ": q must be [graphs, N], node_graph [V]");
    TORCH_CHECK(gate_bits.is_cuda() && gate_bits.scalar_type() == at::kLong && gate_bits.is_contiguous() &&
                    (size_t)gate_bits.numel() >= psvae_gemm_f32_bits_words(V, N), fn, ": gate_bits too short");
    check(psvae_graph_rows_gated(V, N, f32(q, fn), (int)row_stride(q, fn), i32(node_graph, fn),
                                 reinterpret_cast<const uint64_t *>(gate_bits.const_data_ptr<int64_t>()), f32_mut(out, fn),
                                 (int)row_stride(out, fn), cur_stream(out)), fn);
}

void pair_combine_fwd(const Tensor &a, const Tensor &b, const Tensor &c, const Tensor &src, const Tensor &dst,
                      const Tensor &graph, bool relu, Tensor out) {
    PSVAE_ON_DEVICE("psvae::pair_combine_fwd", p(a), p(b), p(c), p(src), p(dst), p(graph), p(out));
    const char *fn = "psvae_pair_combine_fwd";
    check(psvae_pair_combine_fwd((int)src.numel(), (int)out.size(1), f32(a, fn), (int)row_stride(a, fn), f32(b, fn),
                                 (int)row_stride(b, fn), f32(c, fn), (int)row_stride(c, fn), i32(src, fn), i32(dst, fn),
                                 i32(graph, fn), relu ? PSVAE_ACT_RELU : PSVAE_ACT_NONE, f32_mut(out, fn),
                                 (int)row_stride(out, fn), cur_stream(out)),
          fn);
}

void pair_combine_fwd_graphs(const Tensor &a, const Tensor &b, const Tensor &c, const Tensor &src, const Tensor &dst,
                             const Tensor &pair_ptr, const Tensor &node_begin, const Tensor &node_end, int64_t max_nodes, bool relu,
                             Tensor out) {
    PSVAE_ON_DEVICE("psvae::pair_combine_fwd_graphs", p(a), p(b), p(c), p(src), p(dst), p(pair_ptr), p(node_begin), p(node_end), p(out));
    const char *fn = "psvae_pair_combine_fwd_graphs";
    const int B = (int)pair_ptr.numel() - 1;
    TORCH_CHECK(B >= 0 && node_begin.numel() >= B && node_end.numel() >= B && c.size(0) >= B && out.size(0) == src.numel() &&
                    dst.numel() == src.numel(), fn, ": pair_ptr [B+1], node ranges and graph rows for B graphs, out [pairs, F]");
    check(psvae_pair_combine_fwd_graphs(B, (int)out.size(1), (int)max_nodes, f32(a, fn), (int)row_stride(a, fn), f32(b, fn),
                                        (int)row_stride(b, fn), f32(c, fn), (int)row_stride(c, fn), i32(src, fn), i32(dst, fn),
                                        i32(pair_ptr, fn), i32(node_begin, fn), i32(node_end, fn),
                                        relu ? PSVAE_ACT_RELU : PSVAE_ACT_NONE, f32_mut(out, fn), (int)row_stride(out, fn),
                                        cur_stream(out)), fn);
}

void pair_combine_bwd(const Tensor &g_out, const OptTensor &out, bool relu, const Tensor &src, const Tensor &dst,
                      const Tensor &pair_ptr, const Tensor &node_begin, const Tensor &node_end, Tensor g_ab, Tensor g_c) {
    PSVAE_ON_DEVICE("psvae::pair_combine_bwd", p(g_out), p(out), p(src), p(dst), p(pair_ptr), p(node_begin), p(node_end), p(g_ab), p(g_c));
    const char *fn = "psvae_pair_combine_bwd";
    const int B = (int)g_c.size(0), F = (int)g_c.size(1);
    const bool any = src.numel() > 0;
    const bool has_out = any && out.has_value() && out->defined();
    check(psvae_pair_combine_bwd(B, F, any ? f32(g_out, fn) : nullptr, any ? (int)row_stride(g_out, fn) : F,
                                 has_out ? f32(*out, fn) : nullptr, has_out ? (int)row_stride(*out, fn) : F,
                                 relu ? PSVAE_ACT_RELU : PSVAE_ACT_NONE, i32(src, fn), i32(dst, fn), i32(pair_ptr, fn),
                                 i32(node_begin, fn), i32(node_end, fn), f32_mut(g_ab, fn), (int)row_stride(g_ab, fn),
                                 f32_mut(g_c, fn), (int)row_stride(g_c, fn), cur_stream(g_c)),
          fn);
}

void link_ce_fwd(const Tensor &hid, const Tensor &W, const OptTensor &b, const Tensor &golden, const OptTensor &live,
                 Tensor logits, Tensor dlogits, Tensor loss, Tensor n_correct) {
    PSVAE_ON_DEVICE("psvae::link_ce_fwd", p(hid), p(W), p(b), p(golden), p(live), p(logits), p(dlogits), p(loss), p(n_correct));
    const char *fn = "psvae_link_ce_fwd";
    const int P = (int)hid.size(0), F = (int)hid.size(1);
    void *s = cur_stream(W);
    const size_t nbytes = psvae_link_ce_workspace_bytes(P);
    Tensor hold = workspace(nbytes, W, s);
    TORCH_CHECK(n_correct.is_cuda() && n_correct.scalar_type() == at::kInt, fn, ": n_correct must be an int32 GPU tensor");
    check(psvae_link_ce_fwd(P, F, (int)W.size(0), P ? f32(hid, fn) : nullptr, P ? (int)row_stride(hid, fn) : F, f32(W, fn),
                            f32(b, fn), P ? i32(golden, fn) : nullptr,
                            live.has_value() && live->defined() ? i32(*live, fn) : nullptr, f32_mut(logits, fn), f32_mut(dlogits, fn),
                            f32_mut(loss, fn), n_correct.data_ptr<int32_t>(), hold.data_ptr(), nbytes, s),
          fn);
}

void link_ce_bwd_input(const Tensor &dlogits, const Tensor &W, const Tensor &g_loss, Tensor g_hid) {
    PSVAE_ON_DEVICE("psvae::link_ce_bwd_input", p(dlogits), p(W), p(g_loss), p(g_hid));
    const char *fn = "psvae_link_ce_bwd_input";
    const int P = (int)dlogits.size(0);
    check(psvae_link_ce_bwd_input(P, (int)W.size(1), (int)dlogits.size(1), f32(dlogits, fn), f32(W, fn), f32(g_loss, fn),
                                  P ? f32_mut(g_hid, fn) : nullptr, P ? (int)row_stride(g_hid, fn) : (int)W.size(1),
                                  cur_stream(W)),
          fn);
}

// ---- the smaller entry points of the training step: same conventions, no workspace unless noted
const float *f32c(const Tensor &t, const char *what) {   // contiguous
    TORCH_CHECK(t.is_contiguous(), what, ": expected a contiguous tensor");
    return f32(t, what);
}
const float *f32c(const OptTensor &t, const char *what) { return t.has_value() && t->defined() ? f32c(*t, what) : nullptr; }

void segment_sum_fwd(const Tensor &x, const Tensor &seg_begin, const Tensor &seg_end, Tensor out) {
    PSVAE_ON_DEVICE("psvae::segment_sum_fwd", p(x), p(seg_begin), p(seg_end), p(out));
    const char *fn = "psvae_segment_sum_fwd";
    check(psvae_segment_sum_fwd((int)out.size(0), (int)x.size(1), f32(x, fn), (int)row_stride(x, fn), i32(seg_begin, fn),
                                i32(seg_end, fn), f32_mut(out, fn), (int)row_stride(out, fn), cur_stream(out)), fn);
}

void segment_sum_bwd(const Tensor &g_out, const Tensor &seg_begin, const Tensor &seg_end, Tensor g_x) {
    PSVAE_ON_DEVICE("psvae::segment_sum_bwd", p(g_out), p(seg_begin), p(seg_end), p(g_x));
    const char *fn = "psvae_segment_sum_bwd";
    check(psvae_segment_sum_bwd((int)g_out.size(0), (int)g_out.size(1), f32(g_out, fn), (int)row_stride(g_out, fn),
                                i32(seg_begin, fn), i32(seg_end, fn), f32_mut(g_x, fn), (int)row_stride(g_x, fn),
                                cur_stream(g_x)), fn);
}

void rsample_kl_fwd(const Tensor &mean, const Tensor &lv_pre, const Tensor &eps, Tensor z, Tensor kl) {
    PSVAE_ON_DEVICE("psvae::rsample_kl_fwd", p(mean), p(lv_pre), p(eps), p(z), p(kl));
    const char *fn = "psvae_rsample_kl_fwd";
    void *s = cur_stream(z);
    const size_t nbytes = psvae_rsample_kl_workspace_bytes((int)mean.size(0), (int)mean.size(1));
    void *ws = nullptr;
    Tensor hold;
    if (nbytes) hold = workspace(nbytes, z, s), ws = hold.data_ptr();
    check(psvae_rsample_kl_fwd((int)mean.size(0), (int)mean.size(1), f32c(mean, fn), f32c(lv_pre, fn), f32c(eps, fn),
                               const_cast<float *>(f32c(z, fn)), f32_mut(kl, fn), ws, nbytes, s), fn);
}

void rsample_kl_bwd(const Tensor &mean, const Tensor &lv_pre, const Tensor &eps, const OptTensor &g_z, const OptTensor &g_kl,
                    Tensor g_mean, Tensor g_lv_pre) {
    PSVAE_ON_DEVICE("psvae::rsample_kl_bwd", p(mean), p(lv_pre), p(eps), p(g_z), p(g_kl), p(g_mean), p(g_lv_pre));
    const char *fn = "psvae_rsample_kl_bwd";
    check(psvae_rsample_kl_bwd((int)mean.size(0), (int)mean.size(1), f32c(mean, fn), f32c(lv_pre, fn), f32c(eps, fn),
                               f32c(g_z, fn), f32(g_kl, fn), const_cast<float *>(f32c(g_mean, fn)),
                               const_cast<float *>(f32c(g_lv_pre, fn)), cur_stream(g_mean)), fn);
}

void token_ce_fwd(const Tensor &logits, const Tensor &gold, int64_t ignore_index, int64_t n_classes, Tensor dlogits, Tensor loss,
                  Tensor counts) {
    PSVAE_ON_DEVICE("psvae::token_ce_fwd", p(logits), p(gold), p(dlogits), p(loss), p(counts));
    const char *fn = "psvae_token_ce_fwd";
    const int R = (int)logits.size(0), V = (int)n_classes;
    TORCH_CHECK(V >= 1 && V <= logits.size(1) && dlogits.size(0) == R && dlogits.size(1) >= V, fn, ": n_classes / dlogits shape");
    void *s = cur_stream(loss);
    const size_t nbytes = psvae_token_ce_workspace_bytes(R);
    Tensor hold = workspace(nbytes, loss, s);
    TORCH_CHECK(counts.is_cuda() && counts.scalar_type() == at::kInt, fn, ": counts must be an int32 GPU tensor");
    check(psvae_token_ce_fwd(R, V, R ? f32(logits, fn) : nullptr, R ? row_stride(logits, fn) : V, R ? i32(gold, fn) : nullptr,
                             (int)ignore_index, R ? const_cast<float *>(f32c(dlogits, fn)) : nullptr,
                             R ? std::max<int64_t>(row_stride(dlogits, fn), dlogits.size(1)) : V, f32_mut(loss, fn),
                             counts.data_ptr<int32_t>(), hold.data_ptr(), nbytes, s), fn);
}

void embed_trans_fwd(const Tensor &x, const Tensor &xp, const Tensor &xq, int64_t n_atom, int64_t n_piece, int64_t n_pos,
                     const Tensor &T, Tensor out) {
    PSVAE_ON_DEVICE("psvae::embed_trans_fwd", p(x), p(xp), p(xq), p(T), p(out));
    const char *fn = "psvae_embed_trans_fwd";
    TORCH_CHECK(T.dim() == 2 && T.size(0) == n_atom + n_piece + n_pos && T.size(1) == out.size(1), fn,
                ": T must be a [n_atom + n_piece + n_pos, H] table");
    TORCH_CHECK(x.numel() == xp.numel() && x.numel() == xq.numel() && x.numel() == out.size(0), fn,
                ": id vectors and output rows differ in length");
    check(psvae_embed_trans_fwd((int)x.numel(), (int)out.size(1), i32(x, fn), i32(xp, fn), i32(xq, fn), (int)n_atom,
                                (int)n_piece, (int)n_pos, f32c(T, fn), f32_mut(out, fn), (int)row_stride(out, fn),
                                cur_stream(out)), fn);
}

void embed_atom_onehot(const Tensor &x, const Tensor &xp, const Tensor &xq, int64_t n_atom, int64_t n_piece, int64_t n_pos,
                       Tensor out) {
    PSVAE_ON_DEVICE("psvae::embed_atom_onehot", p(x), p(xp), p(xq), p(out));
    const char *fn = "psvae_embed_atom_onehot";
    check(psvae_embed_atom_onehot((int)x.numel(), i32(x, fn), i32(xp, fn), i32(xq, fn), (int)n_atom, (int)n_piece, (int)n_pos,
                                  const_cast<float *>(f32c(out, fn)), (int)out.size(1), cur_stream(out)), fn);
}

void key_sort(const Tensor &k0, const OptTensor &k1, const OptTensor &k2, int64_t n0, int64_t n1, int64_t n2, int64_t chunk,
              Tensor perm, Tensor row_off, Tensor unit_off) {
    const char *fn = "psvae_key_sort";
    PSVAE_ON_DEVICE("psvae::key_sort", p(k0), p(k1), p(k2), p(perm), p(row_off), p(unit_off));
    const bool h1 = k1.has_value() && k1->defined(), h2 = k2.has_value() && k2->defined();
    TORCH_CHECK(h1 || !h2, fn, ": the third id array needs the second");
    const int tables = 1 + (h1 ? 1 : 0) + (h2 ? 1 : 0);
    const int V = (int)k0.numel();
    const int64_t R = n0 + (h1 ? n1 : 0) + (h2 ? n2 : 0);
    TORCH_CHECK((!h1 || k1->numel() == V) && (!h2 || k2->numel() == V), fn, ": id arrays differ in length");
    TORCH_CHECK(perm.numel() == (int64_t)tables * V && row_off.numel() == R + 1 && unit_off.numel() == R + 1, fn,
                ": perm must hold tables * V entries, row_off / unit_off R + 1");
    void *s = cur_stream(perm);
    const size_t nbytes = psvae_key_sort_workspace_bytes(V, (int)R);
    void *ws = nullptr;
    Tensor hold;
    if (nbytes) hold = workspace(nbytes, perm, s), ws = hold.data_ptr();
    check(psvae_key_sort(V, tables, V ? i32(k0, fn) : nullptr, h1 && V ? i32(*k1, fn) : nullptr, h2 && V ? i32(*k2, fn) : nullptr,
                         (int)n0, (int)n1, (int)n2, (int)chunk, const_cast<int32_t *>(i32(perm, fn)),
                         const_cast<int32_t *>(i32(row_off, fn)), const_cast<int32_t *>(i32(unit_off, fn)), ws, nbytes, s), fn);
}

void keyed_row_sum(const Tensor &g, const Tensor &perm, const Tensor &row_off, const Tensor &unit_off, int64_t chunk, Tensor out) {
    const char *fn = "psvae_keyed_row_sum";
    PSVAE_ON_DEVICE("psvae::keyed_row_sum", p(g), p(perm), p(row_off), p(unit_off), p(out));
    const int R = (int)row_off.numel() - 1, H = (int)out.size(1);
    TORCH_CHECK(out.dim() == 2 && out.size(0) == R && g.dim() == 2 && g.size(1) == H && unit_off.numel() == R + 1, fn,
                ": out must be [rows of row_off, width of g]");
    const int64_t entries = perm.numel();
    void *s = cur_stream(out);
    const size_t nbytes = psvae_keyed_row_sum_workspace_bytes(entries, R, H, (int)chunk);
    Tensor hold = workspace(nbytes, out, s);
    check(psvae_keyed_row_sum(R, H, entries, (int)chunk, entries ? f32(g, fn) : nullptr, entries ? (int)row_stride(g, fn) : H,
                              i32(perm, fn), i32(row_off, fn), i32(unit_off, fn), f32_mut(out, fn), (int)row_stride(out, fn),
                              hold.data_ptr(), nbytes, s), fn);
}

void embedding_fwd(const Tensor &ids, const Tensor &table, Tensor out) {
    PSVAE_ON_DEVICE("psvae::embedding_fwd", p(ids), p(table), p(out));
    const char *fn = "psvae_embedding_fwd";
    check(psvae_embedding_fwd((int)ids.numel(), (int)table.size(1), (int)table.size(0), i32(ids, fn), f32c(table, fn),
                              const_cast<float *>(f32c(out, fn)), cur_stream(out)), fn);
}

void onehot(const Tensor &ids, Tensor out) {
    PSVAE_ON_DEVICE("psvae::onehot", p(ids), p(out));
    const char *fn = "psvae_onehot_f32";
    check(psvae_onehot_f32((int)ids.numel(), i32(ids, fn), const_cast<float *>(f32c(out, fn)), (int)out.size(1),
                           cur_stream(out)), fn);
}

void block_table(const Tensor &ea, const Tensor &ep, const Tensor &eq, Tensor Z) {
    PSVAE_ON_DEVICE("psvae::block_table", p(ea), p(ep), p(eq), p(Z));
    const char *fn = "psvae_block_table";
    TORCH_CHECK(ea.dim() == 2 && ep.dim() == 2 && eq.dim() == 2 && Z.dim() == 2 &&
                    Z.size(0) == ea.size(0) + ep.size(0) + eq.size(0) && Z.size(1) == ea.size(1) + ep.size(1) + eq.size(1) + 4,
                fn, ": Z must be [rows of the three tables, their widths + 4]");
    check(psvae_block_table((int)ea.size(0), (int)ea.size(1), (int)ep.size(0), (int)ep.size(1), (int)eq.size(0), (int)eq.size(1),
                            f32c(ea, fn), f32c(ep, fn), f32c(eq, fn), const_cast<float *>(f32c(Z, fn)), cur_stream(Z)), fn);
}

void mse_selected_fwd(const Tensor &pred, const Tensor &props, const Tensor &sel, Tensor diff, Tensor loss) {
    PSVAE_ON_DEVICE("psvae::mse_selected_fwd", p(pred), p(props), p(sel), p(diff), p(loss));
    const char *fn = "psvae_mse_selected_fwd";
    TORCH_CHECK(pred.dim() == 2 && props.dim() == 2 && props.size(0) == pred.size(0) && sel.numel() == pred.size(1) &&
                    sel.scalar_type() == at::kLong && sel.is_contiguous() && diff.numel() == pred.numel() && loss.numel() == 1,
                fn, ": pred [B,k], props [B,*], sel int64 [k], diff [B*k], loss scalar");
    check(psvae_mse_selected_fwd((int)pred.size(0), (int)pred.size(1), f32(pred, fn), (int)row_stride(pred, fn), f32(props, fn),
                                 (int)row_stride(props, fn), sel.const_data_ptr<int64_t>(), const_cast<float *>(f32c(diff, fn)),
                                 f32_mut(loss, fn), cur_stream(loss)), fn);
}

void mse_selected_bwd(const Tensor &diff, const Tensor &g, Tensor g_pred) {
    PSVAE_ON_DEVICE("psvae::mse_selected_bwd", p(diff), p(g), p(g_pred));
    const char *fn = "psvae_mse_selected_bwd";
    TORCH_CHECK(g.numel() == 1 && g_pred.numel() == diff.numel(), fn, ": g scalar, g_pred like diff");
    check(psvae_mse_selected_bwd((int)diff.numel(), f32c(diff, fn), f32(g, fn), const_cast<float *>(f32c(g_pred, fn)),
                                 cur_stream(g_pred)), fn);
}

void loss_mix_fwd(const Tensor &rec, const Tensor &pred, const Tensor &kl, double alpha, double one_minus_alpha, double beta,
                  const OptTensor &beta_dev, Tensor out) {
    PSVAE_ON_DEVICE("psvae::loss_mix_fwd", p(rec), p(pred), p(kl), p(beta_dev), p(out));
    const char *fn = "psvae_loss_mix_fwd";
    TORCH_CHECK(rec.numel() == 1 && pred.numel() == 1 && kl.numel() == 1 && out.numel() == 1, fn, ": scalars expected");
    check(psvae_loss_mix_fwd(f32(rec, fn), f32(pred, fn), f32(kl, fn), (float)alpha, (float)one_minus_alpha, (float)beta,
                             f32(beta_dev, fn), f32_mut(out, fn), cur_stream(out)), fn);
}

void loss_mix_bwd(const Tensor &g, double alpha, double one_minus_alpha, double beta, const OptTensor &beta_dev, Tensor g3) {
    PSVAE_ON_DEVICE("psvae::loss_mix_bwd", p(g), p(beta_dev), p(g3));
    const char *fn = "psvae_loss_mix_bwd";
    TORCH_CHECK(g.numel() == 1 && g3.numel() == 3 && g3.is_contiguous(), fn, ": g scalar, g3 [3]");
    check(psvae_loss_mix_bwd(f32(g, fn), (float)alpha, (float)one_minus_alpha, (float)beta, f32(beta_dev, fn), f32_mut(g3, fn),
                             cur_stream(g3)), fn);
}

void affine_merge_fwd(const Tensor &w_last, const OptTensor &b_last, const Tensor &w_out, const OptTensor &b_out, Tensor w_eff,
                      Tensor b_eff) {
    PSVAE_ON_DEVICE("psvae::affine_merge_fwd", p(w_last), p(b_last), p(w_out), p(b_out), p(w_eff), p(b_eff));
    const char *fn = "psvae_affine_merge_fwd";
    check(psvae_affine_merge_fwd((int)w_last.size(0), (int)w_last.size(1), (int)w_out.size(1), f32c(w_last, fn),
                                 f32c(b_last, fn), f32c(w_out, fn), f32c(b_out, fn), const_cast<float *>(f32c(w_eff, fn)),
                                 const_cast<float *>(f32c(b_eff, fn)), cur_stream(w_eff)), fn);
}

void affine_merge_bwd(const Tensor &w_last, const Tensor &w_out, const OptTensor &b_out, const Tensor &g_w_eff,
                      const Tensor &g_b_eff, Tensor g_w_last, Tensor g_w_out, OptTensor g_b_out) {
    PSVAE_ON_DEVICE("psvae::affine_merge_bwd", p(w_last), p(w_out), p(b_out), p(g_w_eff), p(g_b_eff), p(g_w_last), p(g_w_out), p(g_b_out));
    const char *fn = "psvae_affine_merge_bwd";
    check(psvae_affine_merge_bwd((int)w_last.size(0), (int)w_last.size(1), (int)w_out.size(1), f32c(w_last, fn),
                                 f32c(w_out, fn), f32c(b_out, fn), f32c(g_w_eff, fn), f32c(g_b_eff, fn),
                                 const_cast<float *>(f32c(g_w_last, fn)), const_cast<float *>(f32c(g_w_out, fn)),
                                 const_cast<float *>(f32c(g_b_out, fn)), cur_stream(g_w_last)), fn);
}

void colsum(const Tensor &x, const OptTensor &mask, Tensor out, bool accumulate) {
    PSVAE_ON_DEVICE("psvae::colsum", p(x), p(mask), p(out));
    const char *fn = "psvae_colsum_f32";
    const int M = (int)x.size(0), N = (int)x.size(1);
    const int64_t ld = row_stride(x, fn);
    TORCH_CHECK(!(mask.has_value() && mask->defined()) || row_stride(*mask, fn) == ld, fn, ": mask must have the row stride of x");
    void *s = cur_stream(out);
    const size_t nbytes = psvae_colsum_workspace_bytes(M, N);
    void *ws = nullptr;
    Tensor hold;
    if (nbytes) hold = workspace(nbytes, out, s), ws = hold.data_ptr();
    check(psvae_colsum_f32(M, N, f32(x, fn), f32(mask, fn), ld, f32_mut(out, fn), accumulate ? 1 : 0, ws, nbytes, s), fn);
}

void gru_seq_fwd(const Tensor &gi, const Tensor &h0, const Tensor &w_hh, const OptTensor &b_hh, Tensor hall, Tensor gates,
                 Tensor ghn) {
    PSVAE_ON_DEVICE("psvae::gru_seq_fwd", p(gi), p(h0), p(w_hh), p(b_hh), p(hall), p(gates), p(ghn));
    const char *fn = "psvae_gru_seq_fwd";
    check(psvae_gru_seq_fwd((int)gi.size(0), (int)gi.size(1), (int)gi.size(2) / 3, f32c(gi, fn), f32c(h0, fn), f32c(w_hh, fn),
                            f32c(b_hh, fn), const_cast<float *>(f32c(hall, fn)), const_cast<float *>(f32c(gates, fn)),
                            const_cast<float *>(f32c(ghn, fn)), cur_stream(hall)), fn);
}

void gru_seq_bwd(const Tensor &gates, const Tensor &ghn, const Tensor &hall, const Tensor &g_out, const Tensor &w_hh_t,
                 Tensor dgi, Tensor dgh, Tensor dh0) {
    PSVAE_ON_DEVICE("psvae::gru_seq_bwd", p(gates), p(ghn), p(hall), p(g_out), p(w_hh_t), p(dgi), p(dgh), p(dh0));
    const char *fn = "psvae_gru_seq_bwd";
    check(psvae_gru_seq_bwd((int)gates.size(0), (int)gates.size(1), (int)gates.size(2) / 3, f32c(gates, fn), f32c(ghn, fn),
                            f32c(hall, fn), f32c(g_out, fn), f32c(w_hh_t, fn), const_cast<float *>(f32c(dgi, fn)),
                            const_cast<float *>(f32c(dgh, fn)), const_cast<float *>(f32c(dh0, fn)), cur_stream(dgi)), fn);
}

}  // namespace

TORCH_LIBRARY(psvae, m) {
    m.def("gemm(int M, int N, int K, Tensor A, Tensor? amask, int sa_m, int sa_k, Tensor B, int sb_k, int sb_n, "
          "Tensor(a!) C, int ldc, Tensor? bias, Tensor(b!)? rowsum, Tensor? cmask, bool relu, bool accumulate, "
          "int split_k) -> ()", &gemm);
    m.def("gemm_bits(int M, int N, int K, Tensor A, int sa_m, int sa_k, Tensor B, int sb_k, int sb_n, Tensor(a!) C, int ldc, "
          "Tensor? bias, bool relu, Tensor(b!)? relu_bits, Tensor? gate_bits) -> ()", &gemm_bits);
    m.def("gine_aggregate_fwd(Tensor x, Tensor table, Tensor row_ptr, Tensor col, Tensor cls, Tensor(a!) out) -> ()",
          &gine_aggregate_fwd);
    m.def("gine_aggregate_bwd(Tensor x, Tensor table, Tensor g, Tensor t_row_ptr, Tensor t_col, Tensor t_cls, Tensor? add, "
          "Tensor? seg_add, Tensor? node_graph, Tensor(a!) g_x, Tensor(b!) g_table, bool accumulate_table) -> ()",
          &gine_aggregate_bwd);
    m.def("gine_aggregate_fwd_bits(Tensor x, Tensor table, Tensor row_ptr, Tensor col, Tensor cls, Tensor(a!) out, "
          "Tensor(b!) mask) -> ()", &gine_aggregate_fwd_bits);
    m.def("gine_aggregate_bwd_bits(Tensor mask, int C, Tensor g, Tensor t_row_ptr, Tensor t_col, Tensor t_cls, Tensor? add, "
          "Tensor? seg_add, Tensor? node_graph, Tensor(a!) g_x, Tensor(b!) g_table, bool accumulate_table, "
          "Tensor(c!)? slabs=None) -> ()",
          &gine_aggregate_bwd_bits);
    m.def("gine_table_reduce(Tensor slabs, Tensor(a!) g_table, bool accumulate) -> ()", &gine_table_reduce);
    m.def("graph_rows_gated(Tensor q, Tensor node_graph, Tensor gate_bits, Tensor(a!) out) -> ()", &graph_rows_gated);
    m.def("pair_combine_fwd_graphs(Tensor a, Tensor b, Tensor c, Tensor src, Tensor dst, Tensor pair_ptr, Tensor node_begin, "
          "Tensor node_end, int max_nodes, bool relu, Tensor(a!) out) -> ()", &pair_combine_fwd_graphs);
    m.def("pair_combine_fwd(Tensor a, Tensor b, Tensor c, Tensor src, Tensor dst, Tensor graph, bool relu, Tensor(a!) out) -> ()",
          &pair_combine_fwd);
    m.def("pair_combine_bwd(Tensor g_out, Tensor? out, bool relu, Tensor src, Tensor dst, Tensor pair_ptr, Tensor node_begin, "
          "Tensor node_end, Tensor(a!) g_ab, Tensor(b!) g_c) -> ()", &pair_combine_bwd);
    m.def("link_ce_fwd(Tensor hid, Tensor W, Tensor? b, Tensor golden, Tensor? live, Tensor(a!) logits, Tensor(b!) dlogits, "
          "Tensor(c!) loss, Tensor(d!) n_correct) -> ()", &link_ce_fwd);
    m.def("link_ce_bwd_input(Tensor dlogits, Tensor W, Tensor g_loss, Tensor(a!) g_hid) -> ()", &link_ce_bwd_input);
    m.def("segment_sum_fwd(Tensor x, Tensor seg_begin, Tensor seg_end, Tensor(a!) out) -> ()", &segment_sum_fwd);
    m.def("segment_sum_bwd(Tensor g_out, Tensor seg_begin, Tensor seg_end, Tensor(a!) g_x) -> ()", &segment_sum_bwd);
    m.def("rsample_kl_fwd(Tensor mean, Tensor lv_pre, Tensor eps, Tensor(a!) z, Tensor(b!) kl) -> ()", &rsample_kl_fwd);
    m.def("rsample_kl_bwd(Tensor mean, Tensor lv_pre, Tensor eps, Tensor? g_z, Tensor? g_kl, Tensor(a!) g_mean, "
          "Tensor(b!) g_lv_pre) -> ()", &rsample_kl_bwd);
    m.def("token_ce_fwd(Tensor logits, Tensor gold, int ignore_index, int n_classes, Tensor(a!) dlogits, Tensor(b!) loss, "
          "Tensor(c!) counts) -> ()",
          &token_ce_fwd);
    m.def("embed_trans_fwd(Tensor x, Tensor xp, Tensor xq, int n_atom, int n_piece, int n_pos, Tensor T, Tensor(a!) out) -> ()",
          &embed_trans_fwd);
    m.def("embed_atom_onehot(Tensor x, Tensor xp, Tensor xq, int n_atom, int n_piece, int n_pos, Tensor(a!) out) -> ()",
          &embed_atom_onehot);
    m.def("embedding_fwd(Tensor ids, Tensor table, Tensor(a!) out) -> ()", &embedding_fwd);
    m.def("key_sort(Tensor k0, Tensor? k1, Tensor? k2, int n0, int n1, int n2, int chunk, Tensor(a!) perm, Tensor(b!) row_off, "
          "Tensor(c!) unit_off) -> ()", &key_sort);
    m.def("keyed_row_sum(Tensor g, Tensor perm, Tensor row_off, Tensor unit_off, int chunk, Tensor(a!) out) -> ()",
          &keyed_row_sum);
    m.def("onehot(Tensor ids, Tensor(a!) out) -> ()", &onehot);
    m.def("block_table(Tensor ea, Tensor ep, Tensor eq, Tensor(a!) Z) -> ()", &block_table);
    m.def("mse_selected_fwd(Tensor pred, Tensor props, Tensor sel, Tensor(a!) diff, Tensor(b!) loss) -> ()", &mse_selected_fwd);
    m.def("mse_selected_bwd(Tensor diff, Tensor g, Tensor(a!) g_pred) -> ()", &mse_selected_bwd);
    m.def("loss_mix_fwd(Tensor rec, Tensor pred, Tensor kl, float alpha, float one_minus_alpha, float beta, Tensor? beta_dev, "
          "Tensor(a!) out) -> ()", &loss_mix_fwd);
    m.def("loss_mix_bwd(Tensor g, float alpha, float one_minus_alpha, float beta, Tensor? beta_dev, Tensor(a!) g3) -> ()",
          &loss_mix_bwd);
    m.def("affine_merge_fwd(Tensor w_last, Tensor? b_last, Tensor w_out, Tensor? b_out, Tensor(a!) w_eff, Tensor(b!) b_eff) -> ()",
          &affine_merge_fwd);
    m.def("affine_merge_bwd(Tensor w_last, Tensor w_out, Tensor? b_out, Tensor g_w_eff, Tensor g_b_eff, Tensor(a!) g_w_last, "
          "Tensor(b!) g_w_out, Tensor(c!)? g_b_out) -> ()", &affine_merge_bwd);
    m.def("colsum(Tensor x, Tensor? mask, Tensor(a!) out, bool accumulate) -> ()", &colsum);
    m.def("gru_seq_fwd(Tensor gi, Tensor h0, Tensor w_hh, Tensor? b_hh, Tensor(a!) hall, Tensor(b!) gates, Tensor(c!) ghn) -> ()",
          &gru_seq_fwd);
    m.def("gru_seq_bwd(Tensor gates, Tensor ghn, Tensor hall, Tensor g_out, Tensor w_hh_t, Tensor(a!) dgi, Tensor(b!) dgh, "
          "Tensor(c!) dh0) -> ()", &gru_seq_bwd);
}
