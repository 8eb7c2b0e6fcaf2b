// Host-side utilities shared by the UNet and VAE-decoder executors: checkpoint tensor store,
// owned device allocations and the role-based activation workspace.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "ops.h"

struct DeviceMem {
  std::vector<void*> blocks;
  size_t bytes = 0;
  int alloc(void** out, size_t n) {
    *out = nullptr;
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(out, n);
    if (e != hipSuccess) {
      sdmi_set_error("hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
      return SDMI_ENOMEM;
    }
    blocks.push_back(*out);
    bytes += n;
    sizes[*out] = n;
    return SDMI_OK;
  }
  void release(void* p) {
    if (!p) return;
    for (size_t i = 0; i < blocks.size(); ++i)
      if (blocks[i] == p) {
        blocks.erase(blocks.begin() + i);
        bytes -= sizes[p];
        sizes.erase(p);
        (void)hipFree(p);
        return;
      }
  }
  void release_all() {
    for (void* p : blocks) (void)hipFree(p);
    blocks.clear();
    sizes.clear();
    bytes = 0;
  }
  std::map<void*, size_t> sizes;
};

struct RawTensor {
  float* p = nullptr;            // device fp32 copy of the checkpoint tensor
  std::vector<int64_t> shape;
  size_t numel = 0;
};

struct WeightSpec {
  std::string key;
  std::vector<int64_t> shape;
};

// Checkpoint tensors as handed over by set_weight (always stored as device fp32).
struct WeightStore {
  std::map<std::string, RawTensor> raw;
  DeviceMem* mem = nullptr;

  int set(const std::vector<WeightSpec>& specs, const char* key, const void* dev_ptr, const int64_t* shape,
          int ndim, int dtype, hipStream_t stream) {
    const WeightSpec* spec = nullptr;
    for (const auto& s : specs)
      if (s.key == key) { spec = &s; break; }
    if (!spec) {
      sdmi_set_error("set_weight: unexpected key '%s'", key);
      return SDMI_ENOENT;
    }
    bool same = (int)spec->shape.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = spec->shape[i] == shape[i];
    if (!same) {
      sdmi_set_error("set_weight: shape mismatch for '%s'", key);
      return SDMI_EINVAL;
    }
    RawTensor& t = raw[key];
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    if (!t.p) {
      void* p;
      SDMI_TRY(mem->alloc(&p, n * sizeof(float)));
      t.p = (float*)p;
    }
    t.numel = n;
    t.shape.assign(shape, shape + ndim);
    if (dtype == SDMI_F32) {
      SDMI_CHECK_HIP(hipMemcpyAsync(t.p, dev_ptr, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    } else {
      SDMI_TRY(convert_to_f32(dev_ptr, t.p, (int64_t)n, dtype, stream));
    }
    return SDMI_OK;
  }
  float* get(const std::string& key) const {
    auto it = raw.find(key);
    return it == raw.end() ? nullptr : it->second.p;
  }
  int check_complete(const std::vector<WeightSpec>& specs) const {
    std::string missing;
    int n = 0;
    for (const auto& s : specs)
      if (!raw.count(s.key)) {
        if (n < 8) missing += (n ? ", " : "") + s.key;
        ++n;
      }
    if (n) {
      sdmi_set_error("finalize: %d weight(s) missing: %s%s", n, missing.c_str(), n > 8 ? ", ..." : "");
      return SDMI_ENOENT;
    }
    return SDMI_OK;
  }
  void drop(const std::string& key) {
    auto it = raw.find(key);
    if (it == raw.end()) return;
    mem->release(it->second.p);
    raw.erase(it);
  }
};

// Activation workspace: named roles sized by a dry run of the forward, then allocated once.
struct Workspace {
  bool dry = false;
  std::map<std::string, size_t> need, cap;
  std::map<std::string, void*> ptr;
  DeviceMem* mem = nullptr;

  void* get(const std::string& role, size_t bytes) {
    if (dry) {
      size_t& n = need[role];
      if (bytes > n) n = bytes;
      return (void*)(uintptr_t)256;    // non-null placeholder, never dereferenced in a dry run
    }
    return ptr[role];
  }
  bool fits() const {
    for (const auto& kv : need) {
      auto it = cap.find(kv.first);
      if (it == cap.end() || it->second < kv.second) return false;
    }
    return true;
  }
  int commit() {
    for (const auto& kv : need) {
      size_t& c = cap[kv.first];
      if (c < kv.second) {
        if (ptr[kv.first]) mem->release(ptr[kv.first]);
        void* p;
        SDMI_TRY(mem->alloc(&p, kv.second));
        ptr[kv.first] = p;
        c = kv.second;
      }
    }
    return SDMI_OK;
  }
};

struct PackedConv {
  void* w = nullptr;        // packed [N][taps*Cin] in compute dtype
  const float* bias = nullptr;
  int cin = 0, cout = 0, taps = 1;
};
