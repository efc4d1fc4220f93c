// torch.ops.mrirt_native.* — the C ABI of libmrirt.so (include/mrirt.h) registered as PyTorch operators from C++.
//
// Same contract as the Python registrations in mrirt/torch_ops.py (torch.ops.mrirt.*): device tensors in, a device
// tensor out, launched on the current HIP stream, no synchronisation; parameter blocks travel as CPU uint8 tensors
// holding the C structs byte for byte; every size check the C side cannot make (it only sees pointers) is made here.
// No device code in this file: it is compiled by the host compiler against the torch headers and linked to
// libmrirt.so, whose kernels do the work.  Replaces kernel.dispatch of inr/viewer/brats_viewer.py:431-442,
// scripts/volumeRendering/app.py:350-358 and scripts/raymarch/app.py:212-223 for callers that want operators.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <cstring>
#include <optional>

#include "mrirt.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

template <typename T>
T unblob(const Tensor& t, const char* what) {
    TORCH_CHECK_TYPE(t.device().is_cpu() && t.scalar_type() == at::kByte && t.numel() == (int64_t)sizeof(T),
                     "expected a CPU uint8 tensor of ", sizeof(T), " bytes (", what, ")");
    T v;
    const Tensor c = t.contiguous();
    std::memcpy(&v, c.data_ptr(), sizeof(T));
    return v;
}

const void* dev_ptr(const OptTensor& t, at::ScalarType dt, const char* what) {
    if (!t.has_value()) return nullptr;
    TORCH_CHECK_TYPE(t->is_cuda() && t->scalar_type() == dt && t->is_contiguous(), what, ": expected a contiguous device tensor of the right dtype");
    return t->data_ptr();
}

// The kernels are launched on the current stream OF THE TENSORS' DEVICE: every operator first checks that all its
// device tensors live on one GPU, then makes that GPU current for the allocation of the output and for the stream
// lookup (with device 0 current and the grids on device 1 the launch would otherwise go to device 0's
// stream with device-1 pointers).
using DeviceGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

void same_device(std::optional<at::Device>& dev, const OptTensor& t, const char* what) {
    if (!t.has_value()) return;
    TORCH_CHECK_TYPE(t->is_cuda(), what, ": expected a device tensor");
    if (!dev.has_value()) dev = t->device();
    TORCH_CHECK_VALUE(t->device() == *dev, what, " is on ", t->device(), " but another operand is on ", *dev,
                      ": every device tensor of one call must live on the same GPU");
}

void* current_stream() { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

void check(int rc, const char* where) {
    TORCH_CHECK(rc == MRIRT_OK, where, ": ", mrirt_status_string(rc), " [status ", rc, "]");
}

std::vector<int64_t> out_shape(uint32_t width, uint32_t height, const MrirtRenderExt& e) {
    if (e.tileSize > 0) {
        const int64_t local = mrirt_tiles_for_rank(width, height, e.tileSize, e.tileRank, e.tileWorld);
        return { local, (int64_t)e.tileSize, (int64_t)e.tileSize, 4 };
    }
    return { (int64_t)height, (int64_t)width, 4 };
}

int64_t grid_need(const uint32_t dims[3], uint32_t layout) {
    switch (layout) {
        case MRIRT_LAYOUT_LINEAR: return (int64_t)dims[0] * dims[1] * dims[2];
        case MRIRT_LAYOUT_BRICK:  return mrirt_brick_elems(dims);
        case MRIRT_LAYOUT_VGA:    return 4 * mrirt_vga_elems(dims);
        case MRIRT_LAYOUT_LABCELL: return 2 * mrirt_vec4_elems(dims);      // 8-byte elements, counted in int32
        default:                  return 4 * mrirt_vec4_elems(dims);
    }
}

// brats_main (inr/viewer/brats_rt.slang:85-168) through mrirt_render_brats_ex
Tensor render_brats(const Tensor& params, const Tensor& ext, const OptTensor& vol0, const OptTensor& vol1,
                    const OptTensor& vol2, const OptTensor& vol3, const OptTensor& labels, const OptTensor& preds) {
    const MrirtBratsParams P = unblob<MrirtBratsParams>(params, "MrirtBratsParams");
    const MrirtRenderExt E = unblob<MrirtRenderExt>(ext, "MrirtRenderExt");
    const OptTensor* vols[4] = { &vol0, &vol1, &vol2, &vol3 };
    const char* names[4] = { "gIntensity0", "gIntensity1", "gIntensity2", "gIntensity3" };
    const void* vp[4];
    const int64_t need = grid_need(P.dims, E.layout), lneed = grid_need(P.dims, E.labelLayout);
    std::optional<at::Device> dev;
    const bool mod4 = E.layout == MRIRT_LAYOUT_MOD4;                    // vol0 = the float4 grid of all four modalities
    for (int m = 0; m < 4; ++m) {
        vp[m] = dev_ptr(*vols[m], at::kFloat, names[m]);
        same_device(dev, *vols[m], names[m]);
        const bool wanted = mod4 ? m == 0 : P.volEnabled[m] != 0;
        TORCH_CHECK_VALUE(!wanted || (vols[m]->has_value() && (*vols[m])->numel() >= need),
                          names[m], mod4 ? " (the MOD4 grid) holds fewer than " : " is enabled but holds fewer than ", need, " elements");
    }
    TORCH_CHECK_VALUE(dev.has_value(), "no intensity grid bound");
    const void* lab = dev_ptr(labels, at::kInt, "gLabels");
    const void* prd = dev_ptr(preds, at::kInt, "gPreds");
    TORCH_CHECK_VALUE(P.showSeg == 0 || (labels.has_value() && labels->numel() >= lneed), "showSeg is set but gLabels is missing or too small");
    const bool cells = E.labelLayout == MRIRT_LAYOUT_LABCELL;          // gLabels carries both grids, gPreds is ignored
    const OptTensor& predSrc = cells ? labels : preds;
    TORCH_CHECK_VALUE(P.showPred == 0 || (predSrc.has_value() && predSrc->numel() >= lneed), "showPred is set but gPreds (or the label-cell grid) is missing or too small");
    same_device(dev, labels, "gLabels");
    same_device(dev, preds, "gPreds");
    DeviceGuard guard(*dev);
    const auto dt = E.outFormat == MRIRT_OUT_RGBA16F ? at::kHalf : at::kFloat;
    Tensor out = at::empty(out_shape(P.imageSize[0], P.imageSize[1], E), at::TensorOptions().dtype(dt).device(*dev));
    check(mrirt_render_brats_ex(&P, &E, vp, lab, prd, out.data_ptr(), (int64_t)P.imageSize[0], nullptr, current_stream()),
          "mrirt_render_brats_ex");
    return out;
}

// volume_cs (scripts/volumeRendering/volume_render.slang:104-148) through mrirt_render_volume
Tensor render_volume(const Tensor& params, const Tensor& ext, const Tensor& volume, int64_t mode) {
    const MrirtVolumeParams P = unblob<MrirtVolumeParams>(params, "MrirtVolumeParams");
    const MrirtRenderExt E = unblob<MrirtRenderExt>(ext, "MrirtRenderExt");
    TORCH_CHECK_VALUE(mode >= 0 && mode <= 2, "mode must be 0 (u32x4), 1 (u8) or 2 (f32)");
    const at::ScalarType want = mode == 0 ? at::kInt : mode == 1 ? at::kByte : at::kFloat;
    const void* vol = dev_ptr(volume, want, "gVolumeU8");
    const int64_t nvox = (int64_t)P.volDim[0] * P.volDim[1] * P.volDim[2];
    TORCH_CHECK_VALUE(volume.numel() >= nvox, "gVolumeU8 holds fewer than ", nvox, " voxels");
    DeviceGuard guard(volume.device());
    const auto dt = E.outFormat == MRIRT_OUT_RGBA16F ? at::kHalf : at::kFloat;
    Tensor out = at::empty(out_shape(P.imageSize[0], P.imageSize[1], E), at::TensorOptions().dtype(dt).device(volume.device()));
    check(mrirt_render_volume(&P, &E, vol, (uint32_t)mode, out.data_ptr(), (int64_t)P.imageSize[0], nullptr, current_stream()),
          "mrirt_render_volume");
    return out;
}

// raymarch_cs (scripts/raymarch/raymarch.slang:60-99); `like` only names the device
Tensor render_sdf(const Tensor& params, int64_t width, int64_t height, const Tensor& like) {
    const MrirtSdfParams P = unblob<MrirtSdfParams>(params, "MrirtSdfParams");
    TORCH_CHECK_TYPE(like.is_cuda(), "like: expected a device tensor");
    DeviceGuard guard(like.device());
    Tensor out = at::empty({ height, width, 4 }, at::TensorOptions().dtype(at::kFloat).device(like.device()));
    check(mrirt_render_sdf(&P, (uint32_t)width, (uint32_t)height, out.data_ptr<float>(), width, current_stream()), "mrirt_render_sdf");
    return out;
}

// logits [n, out_dim] of the packed MLP (inr/inr/model.py:21-50 for kind 0, notebooks/neumors_inr.ipynb:1165-1178 for
// kind 1; kinds 2 / 3 take the [n, in_dim] input matrix in `feats`) through mrirt_inr_forward
Tensor inr_forward(const Tensor& weights, const Tensor& biases, int64_t kind, int64_t num_layers, int64_t in_dim, int64_t out_dim,
                   int64_t hidden, int64_t fourier_freqs, int64_t num_mods, double w0, const OptTensor& coords,
                   const OptTensor& feats, int64_t n) {
    MrirtInrDesc d;
    std::memset(&d, 0, sizeof d);
    d.kind = (uint32_t)kind; d.numLayers = (uint32_t)num_layers; d.inDim = (uint32_t)in_dim; d.outDim = (uint32_t)out_dim;
    d.hidden = (uint32_t)hidden; d.fourierFreqs = (uint32_t)fourier_freqs; d.numMods = (uint32_t)num_mods; d.w0 = (float)w0;
    const int64_t need = mrirt_inr_pack_bytes(&d);
    TORCH_CHECK_VALUE(need > 0, "unsupported network shape");
    TORCH_CHECK_TYPE(weights.is_cuda() && weights.scalar_type() == at::kByte && weights.numel() >= need,
                     "weights: expected the ", need, "-byte packed image of mrirt_inr_pack_weights on the device");
    const int64_t nb = (num_layers - 1) * hidden + ((out_dim + 31) / 32) * 32;
    d.weights = weights.data_ptr();
    d.biases = static_cast<const float*>(dev_ptr(biases, at::kFloat, "biases"));
    TORCH_CHECK_VALUE(biases.numel() >= nb, "biases holds fewer than ", nb, " floats (each layer padded to a multiple of 32)");
    const float* co = static_cast<const float*>(dev_ptr(coords, at::kFloat, "coords"));
    const float* fe = static_cast<const float*>(dev_ptr(feats, at::kFloat, "feats"));
    TORCH_CHECK_VALUE(kind >= 2 || (coords.has_value() && coords->numel() >= 3 * n), "coords must hold [n, 3] floats");
    const int64_t width = kind >= 2 ? in_dim : num_mods;
    TORCH_CHECK_VALUE(width == 0 || (feats.has_value() && feats->numel() >= width * n), "feats must hold [n, ", width, "] floats");
    std::optional<at::Device> dev = weights.device();
    same_device(dev, biases, "biases");
    same_device(dev, coords, "coords");
    same_device(dev, feats, "feats");
    DeviceGuard guard(*dev);
    Tensor out = at::empty({ n, out_dim }, at::TensorOptions().dtype(at::kFloat).device(weights.device()));
    check(mrirt_inr_forward(&d, co, fe, n, out.data_ptr<float>(), nullptr, current_stream()), "mrirt_inr_forward");
    return out;
}

}  // namespace

TORCH_LIBRARY(mrirt_native, m) {
    m.def("render_brats(Tensor params, Tensor ext, Tensor? vol0, Tensor? vol1, Tensor? vol2, Tensor? vol3, Tensor? labels, Tensor? preds) -> Tensor");
    m.def("render_volume(Tensor params, Tensor ext, Tensor volume, int mode) -> Tensor");
    m.def("render_sdf(Tensor params, int width, int height, Tensor like) -> Tensor");
    m.def("inr_forward(Tensor weights, Tensor biases, int kind, int num_layers, int in_dim, int out_dim, int hidden, "
          "int fourier_freqs, int num_mods, float w0, Tensor? coords, Tensor? feats, int n) -> Tensor");
}

// the parameter blocks are CPU tensors and the grids device tensors: no single dispatch key fits, so the
// implementations are registered for every backend and check their arguments themselves
TORCH_LIBRARY_IMPL(mrirt_native, CompositeExplicitAutograd, m) {
    m.impl("render_brats", &render_brats);
    m.impl("render_volume", &render_volume);
    m.impl("render_sdf", &render_sdf);
    m.impl("inr_forward", &inr_forward);
}
