// multi_gpu.cc — see multi_gpu.h.
#include "multi_gpu.h"

#include <dlfcn.h>

#include <stdexcept>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace app {

struct DeviceGroup::Api {  // the six RCCL entry points this file uses, resolved at run time
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

void DeviceGroup::fail(const std::string& what) const { throw std::runtime_error("multi-GPU: " + what); }

#define MG_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) fail(std::string(#call) + " failed: " + hipGetErrorString(e_)); \
    } while (0)
#define MG_NCCL(call)                                                                            \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess) fail(std::string(#call) + " failed: " + api_->error_string(r_));  \
    } while (0)

DeviceGroup::DeviceGroup(const std::vector<int>& devices) : devices_(devices) {
    // A constructor that throws gets no destructor call: whatever init() had made by then (the library handle, the entry
    // point table, streams, communicators) is released here, and the caller's current device is put back either way.
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    try {
        init();
    } catch (...) {
        release();
        if (caller_device >= 0) (void)hipSetDevice(caller_device);
        throw;
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
}

void DeviceGroup::init() {
    if (devices_.empty()) fail("no devices");
    int n_visible = 0;
    MG_HIP(hipGetDeviceCount(&n_visible));
    for (int d : devices_)
        if (d < 0 || d >= n_visible)
            fail("device " + std::to_string(d) + " requested, " + std::to_string(n_visible) + " visible");
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        rccl_ = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (rccl_) break;
    }
    if (!rccl_) fail(std::string("cannot open librccl.so.1: ") + dlerror());
    api_ = new Api;
    auto sym = [&](const char* name) {
        void* p = dlsym(rccl_, name);
        if (!p) fail(std::string("librccl has no ") + name);
        return p;
    };
    api_->comm_init_all = reinterpret_cast<decltype(api_->comm_init_all)>(sym("ncclCommInitAll"));
    api_->comm_destroy = reinterpret_cast<decltype(api_->comm_destroy)>(sym("ncclCommDestroy"));
    api_->reduce = reinterpret_cast<decltype(api_->reduce)>(sym("ncclReduce"));
    api_->group_start = reinterpret_cast<decltype(api_->group_start)>(sym("ncclGroupStart"));
    api_->group_end = reinterpret_cast<decltype(api_->group_end)>(sym("ncclGroupEnd"));
    api_->error_string = reinterpret_cast<decltype(api_->error_string)>(sym("ncclGetErrorString"));
    streams_.assign(devices_.size(), nullptr);
    buffers_.assign(devices_.size(), nullptr);
    capacity_.assign(devices_.size(), 0);
    comms_.assign(devices_.size(), nullptr);
    for (size_t k = 0; k < devices_.size(); ++k) {
        MG_HIP(hipSetDevice(devices_[k]));
        hipStream_t st = nullptr;
        MG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        streams_[k] = st;
    }
    // (the communicators land in comms_ directly, so that a call that fails half way leaves the ones it made where
    // release() finds them)
    static_assert(sizeof(ncclComm_t) == sizeof(void*), "a communicator handle is a pointer");
    MG_NCCL(api_->comm_init_all(reinterpret_cast<ncclComm_t*>(comms_.data()), static_cast<int>(devices_.size()), devices_.data()));
}

void DeviceGroup::release() noexcept {
    for (size_t k = 0; k < devices_.size(); ++k) {
        (void)hipSetDevice(devices_[k]);
        if (k < comms_.size() && comms_[k] && api_ && api_->comm_destroy) (void)api_->comm_destroy(static_cast<ncclComm_t>(comms_[k]));
        if (k < buffers_.size() && buffers_[k]) (void)hipFree(buffers_[k]);
        if (k < streams_.size() && streams_[k]) (void)hipStreamDestroy(static_cast<hipStream_t>(streams_[k]));
    }
    comms_.clear();
    buffers_.clear();
    streams_.clear();
    delete api_;
    api_ = nullptr;
    if (rccl_) dlclose(rccl_);
    rccl_ = nullptr;
}

DeviceGroup::~DeviceGroup() { release(); }

float* DeviceGroup::sums(size_t k, size_t n_floats) {
    if (capacity_[k] < n_floats) {
        MG_HIP(hipSetDevice(devices_[k]));
        if (buffers_[k]) MG_HIP(hipFree(buffers_[k]));
        buffers_[k] = nullptr;
        capacity_[k] = 0;
        MG_HIP(hipMalloc(reinterpret_cast<void**>(&buffers_[k]), n_floats * sizeof(float)));
        capacity_[k] = n_floats;
    }
    return buffers_[k];
}

void DeviceGroup::reduce_to_first(size_t n_floats) {
    // one group call from one thread: every communicator's reduce is queued on its device's stream, root = rank 0
    MG_NCCL(api_->group_start());
    for (size_t k = 0; k < devices_.size(); ++k)
        MG_NCCL(api_->reduce(buffers_[k], buffers_[k], n_floats, ncclFloat, ncclSum, 0, static_cast<ncclComm_t>(comms_[k]),
                             static_cast<hipStream_t>(streams_[k])));
    MG_NCCL(api_->group_end());
    for (size_t k = 0; k < devices_.size(); ++k) {
        MG_HIP(hipSetDevice(devices_[k]));
        MG_HIP(hipStreamSynchronize(static_cast<hipStream_t>(streams_[k])));
    }
}

void DeviceGroup::download_first(float* host, size_t n_floats) {
    MG_HIP(hipSetDevice(devices_[0]));
    MG_HIP(hipMemcpy(host, buffers_[0], n_floats * sizeof(float), hipMemcpyDeviceToHost));
}

}  // namespace app
