// Dense step of BASELINE config 5 (DLRM bottom MLP + dot interaction) — launchers and the host-side owner of the
// device weights.  See dense_kernels.hip for the kernels and include/hps_amd.h (hps_dense_*) for the C ABI.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <vector>

#include "../cache/device_types.h"
#include "../common/status.h"

namespace hps {

constexpr int kDenseMaxLayers = 8;

struct DenseMlpDesc {
  uint32_t in_dim = 0;      // numeric features per sample (13 for Criteo)
  uint32_t in_pad = 0;      // in_dim rounded up to a multiple of 16 (one MFMA K-step)
  uint32_t num_layers = 0;
  uint32_t max_dim = 0;     // widest activation row (max over in_pad and the layer widths)
  uint32_t buf_dim[2] = {0, 0};  // row widths of the two LDS activation buffers: [0] holds the input and the outputs of
                                 // odd layers, [1] the outputs of even layers
  uint32_t dims[kDenseMaxLayers] = {0};
  const void* weights[kDenseMaxLayers] = {nullptr};   // device, f16, MFMA fragment order (dense.cpp), K_0 = in_pad
  const float* biases[kDenseMaxLayers] = {nullptr};   // device, fp32 [dims[l]]
};

hipError_t LaunchDenseMlp(const DenseMlpDesc& d, const float* d_x, uint64_t batch, void* d_out_f16, int cu_count,
                          hipStream_t stream);
hipError_t LaunchDenseInteract(const float* d_emb, const void* d_bottom_f16, uint64_t batch, uint32_t T, uint32_t D,
                               uint32_t out_stride, void* d_out_f16, int cu_count, hipStream_t stream);

// fused_kernels.hip: the interaction reading its rows from cache slots / miss staging (no OUTPUT0); T*D/4 <= 1024 chunks
hipError_t LaunchLookupInteract(const TableCacheDev* d_tables, const MissDesc* d_md, const int32_t* d_slot, const int32_t* d_rep_of,
                                const int32_t* d_uidx_of, const float* d_staging,
                                const void* d_bottom_f16, uint64_t batch, uint32_t T, uint32_t D, uint32_t out_stride, void* d_out_f16,
                                int cu_count, hipStream_t stream);

// Owns the f16 weights on one device.  Thread-compatible (one forward at a time per object).
class DenseInteraction {
 public:
  ~DenseInteraction();
  // weights[l]: host fp32 [K_l][dims[l]] row-major (the x @ W convention), biases[l]: host fp32 [dims[l]]
  static Status Create(int device, uint32_t num_dense, const std::vector<uint32_t>& dims,
                       const std::vector<const float*>& weights, const std::vector<const float*>& biases,
                       uint32_t num_tables, uint32_t emb_dim, DenseInteraction** out);
  uint32_t out_dim() const { return emb_dim_ + (num_tables_ + 1) * num_tables_ / 2; }
  uint32_t out_stride() const { return (out_dim() + 7) / 8 * 8; }  // rows padded to 16 bytes
  uint32_t num_dense() const { return mlp_.in_dim; }
  uint32_t num_tables() const { return num_tables_; }
  uint32_t emb_dim() const { return emb_dim_; }
  // d_dense: [batch][num_dense] fp32; d_emb: the lookup's OUTPUT0, table-major [num_tables][batch][emb_dim] fp32;
  // d_out: [batch][out_stride] f16.  Enqueues on `stream`; returns without synchronising.
  Status Forward(const float* d_dense, const float* d_emb, uint64_t batch, void* d_out, hipStream_t stream);
  // First half only: the bottom MLP into the object's scratch, [batch][emb_dim] f16 (used by the fused lookup path)
  Status BottomMlp(const float* d_dense, uint64_t batch, hipStream_t stream, const void** d_bottom);
  // Second half only: the dot interaction over OUTPUT0 and a bottom-MLP result of this object (BottomMlp's d_bottom)
  Status Interact(const float* d_emb, const void* d_bottom, uint64_t batch, void* d_out, hipStream_t stream);
  int device() const { return device_; }
  int cu_count() const { return cu_count_; }

 private:
  DenseInteraction() = default;
  int device_ = 0;
  int cu_count_ = 256;
  uint32_t num_tables_ = 0, emb_dim_ = 0;
  DenseMlpDesc mlp_;
  std::vector<void*> allocations_;
  void* d_bottom_ = nullptr;   // [capacity][emb_dim] f16 scratch between the two kernels
  uint64_t bottom_capacity_ = 0;
};

}  // namespace hps
