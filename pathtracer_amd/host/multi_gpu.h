// multi_gpu.h — one node's GPUs behind one Renderer: per-device sum buffers and streams, and the RCCL reduce(sum) of the
// partial frames onto the first device (the combine step the reference has no need for: its one frame loop,
// src/renderer.cc:77-123, runs in one address space).
#pragma once

#include <cstddef>
#include <string>
#include <vector>

namespace app {

// The devices of one render: a stream and a width*height RGBA float sum buffer on each, and one RCCL communicator per
// device (ncclCommInitAll).  librccl.so is opened when the group is made, never before: a single-GPU run does not load it.
class DeviceGroup {
  public:
    explicit DeviceGroup(const std::vector<int>& devices);  // throws std::runtime_error
    DeviceGroup(const DeviceGroup&) = delete;
    DeviceGroup& operator=(const DeviceGroup&) = delete;
    ~DeviceGroup();

    size_t size() const { return devices_.size(); }
    int device(size_t k) const { return devices_[k]; }
    void* stream(size_t k) const { return streams_[k]; }          // hipStream_t
    float* sums(size_t k, size_t n_floats);                        // device buffer of device k, grown on demand
    // reduce(sum) of every device's first n_floats sums onto device(0), in place there; returns when it has completed
    void reduce_to_first(size_t n_floats);
    void download_first(float* host, size_t n_floats);             // device(0)'s buffer -> host

  private:
    [[noreturn]] void fail(const std::string& what) const;
    void init();               // everything the constructor makes; may throw half way
    void release() noexcept;   // frees whatever exists (also after a partial init)
    std::vector<int> devices_;
    std::vector<void*> streams_;
    std::vector<float*> buffers_;
    std::vector<size_t> capacity_;
    std::vector<void*> comms_;  // ncclComm_t
    void* rccl_ = nullptr;      // dlopen handle
    struct Api;
    Api* api_ = nullptr;
};

}  // namespace app
