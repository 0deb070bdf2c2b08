#include "dense.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>

namespace hps {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::hps::Error(::hps::Code::kInternal, #expr, " failed: ", hipGetErrorString(_e), " (", \
                          __FILE__, ":", __LINE__, ")");                                           \
  } while (0)

DenseInteraction::~DenseInteraction() {
  (void)hipSetDevice(device_);
  for (void* p : allocations_) (void)hipFree(p);
  if (d_bottom_) (void)hipFree(d_bottom_);
}

Status DenseInteraction::Create(int device, uint32_t num_dense, const std::vector<uint32_t>& dims,
                                const std::vector<const float*>& weights, const std::vector<const float*>& biases,
                                uint32_t num_tables, uint32_t emb_dim, DenseInteraction** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return Error(Code::kUnavailable, "the dense interaction step needs a HIP device and none is visible; there is no CPU fallback");
  if (device < 0 || device >= ndev) return Error(Code::kInvalidArg, "device ", device, " is not visible (", ndev, " HIP devices)");
  const size_t L = dims.size();
  if (L == 0 || L > (size_t)kDenseMaxLayers) return Error(Code::kInvalidArg, "bottom MLP: 1..", kDenseMaxLayers, " layers supported, got ", L);
  if (weights.size() != L || biases.size() != L) return Error(Code::kInvalidArg, "bottom MLP: one weight and one bias array per layer");
  if (num_dense == 0 || num_dense > 256) return Error(Code::kInvalidArg, "bottom MLP: 1..256 numeric features supported, got ", num_dense);
  for (size_t l = 0; l < L; ++l) {
    if (dims[l] == 0 || dims[l] % 32 != 0 || dims[l] > 512)
      return Error(Code::kInvalidArg, "bottom MLP: layer widths must be multiples of 32 up to 512, layer ", l, " has ", dims[l]);
    if (!weights[l] || !biases[l]) return Error(Code::kInvalidArg, "bottom MLP: null weight/bias pointer at layer ", l);
  }
  if (num_tables == 0 || num_tables + 1 > 32) return Error(Code::kInvalidArg, "interaction: 1..31 embedding tables supported, got ", num_tables);
  if (emb_dim == 0 || emb_dim % 16 != 0) return Error(Code::kInvalidArg, "interaction: embedding width must be a multiple of 16, got ", emb_dim);
  if (dims.back() != emb_dim)
    return Error(Code::kInvalidArg, "the last bottom-MLP layer (", dims.back(), ") must be as wide as the embeddings (", emb_dim, ")");

  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<DenseInteraction> d(new DenseInteraction());
  d->device_ = device;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  d->cu_count_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  d->num_tables_ = num_tables;
  d->emb_dim_ = emb_dim;
  DenseMlpDesc& m = d->mlp_;
  m.in_dim = num_dense;
  m.in_pad = (num_dense + 15) / 16 * 16;
  m.num_layers = (uint32_t)L;
  m.max_dim = m.in_pad;
  m.buf_dim[0] = m.in_pad;
  m.buf_dim[1] = 32;
  uint32_t K = m.in_pad, Kreal = num_dense;
  for (size_t l = 0; l < L; ++l) {
    const uint32_t N = dims[l];
    m.dims[l] = N;
    m.max_dim = std::max(m.max_dim, N);
    m.buf_dim[(l & 1) ? 0 : 1] = std::max(m.buf_dim[(l & 1) ? 0 : 1], N);
    // MFMA fragment order, rounded to f16, zero-padded for k >= Kreal: block (nt, ks) holds, for lane = r + 32*h,
    // the eight values W[k = 16*ks + 8*h + e][n = 32*nt + r], e = 0..7 (dense_kernels.hip reads one block per load)
    std::vector<_Float16> wt((size_t)N * K, (_Float16)0.f);
    const uint32_t ksteps = K / 16;
    for (uint32_t nt = 0; nt < N / 32; ++nt)
      for (uint32_t ks = 0; ks < ksteps; ++ks)
        for (uint32_t lane = 0; lane < 64; ++lane)
          for (uint32_t e = 0; e < 8; ++e) {
            const uint32_t k = 16 * ks + 8 * (lane >> 5) + e, n = 32 * nt + (lane & 31);
            if (k < Kreal) wt[(((size_t)nt * ksteps + ks) * 64 + lane) * 8 + e] = (_Float16)weights[l][(size_t)k * N + n];
          }
    void* dw = nullptr;
    void* db = nullptr;
    HIP_TRY(hipMalloc(&dw, wt.size() * sizeof(_Float16)));
    d->allocations_.push_back(dw);
    HIP_TRY(hipMalloc(&db, N * sizeof(float)));
    d->allocations_.push_back(db);
    HIP_TRY(hipMemcpy(dw, wt.data(), wt.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, biases[l], N * sizeof(float), hipMemcpyHostToDevice));
    m.weights[l] = dw;
    m.biases[l] = (const float*)db;
    K = Kreal = N;
  }
  *out = d.release();
  return Status::Ok();
}

Status DenseInteraction::Forward(const float* d_dense, const float* d_emb, uint64_t batch, void* d_out, hipStream_t stream) {
  if (batch == 0) return Status::Ok();
  if (!d_dense || !d_emb || !d_out) return Error(Code::kInvalidArg, "dense forward: null device pointer");
  const void* bottom = nullptr;
  HPS_RETURN_IF_ERROR(BottomMlp(d_dense, batch, stream, &bottom));
  HIP_TRY(LaunchDenseInteract(d_emb, bottom, batch, num_tables_, emb_dim_, out_stride(), d_out, cu_count_, stream));
  return Status::Ok();
}

Status DenseInteraction::Interact(const float* d_emb, const void* d_bottom, uint64_t batch, void* d_out, hipStream_t stream) {
  if (batch == 0) return Status::Ok();
  if (!d_emb || !d_bottom || !d_out) return Error(Code::kInvalidArg, "dense interaction: null device pointer");
  HIP_TRY(hipSetDevice(device_));
  HIP_TRY(LaunchDenseInteract(d_emb, d_bottom, batch, num_tables_, emb_dim_, out_stride(), d_out, cu_count_, stream));
  return Status::Ok();
}

Status DenseInteraction::BottomMlp(const float* d_dense, uint64_t batch, hipStream_t stream, const void** d_bottom) {
  if (!d_dense || !d_bottom) return Error(Code::kInvalidArg, "bottom MLP: null pointer");
  HIP_TRY(hipSetDevice(device_));
  if (batch > bottom_capacity_) {
    HIP_TRY(hipStreamSynchronize(stream));
    if (d_bottom_) (void)hipFree(d_bottom_);
    d_bottom_ = nullptr;
    bottom_capacity_ = 0;
    HIP_TRY(hipMalloc(&d_bottom_, batch * emb_dim_ * sizeof(_Float16)));
    bottom_capacity_ = batch;
  }
  HIP_TRY(LaunchDenseMlp(mlp_, d_dense, batch, d_bottom_, cu_count_, stream));
  *d_bottom = d_bottom_;
  return Status::Ok();
}

}  // namespace hps
