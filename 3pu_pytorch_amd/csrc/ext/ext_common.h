// ext_common.h -- shared by the two compiled drop-in extension modules (`sampling`, `losses`).
//
// The modules are what a maintainer of the reference puts on sys.path instead of the CUDA
// extensions built by sampling/setup.py and losses/setup.py: pybind11 over torch tensors, same
// module names, function names, positional signatures and return values, and underneath nothing but
// calls into the C ABI of lib3pu_hip.so (include/tpu3.h) with raw device pointers and torch's
// current HIP stream.  No kernels live here.
#pragma once
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include "../../../include/tpu3.h"

namespace tpu3ext {

// sampling/sampling.cpp:20-24 checks (RuntimeError through TORCH_CHECK)
inline void check_input(const at::Tensor &t, const char *name)
{
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
inline void check_dtype(const at::Tensor &t, at::ScalarType want, const char *name)
{
    TORCH_CHECK(t.scalar_type() == want, name, " must be ", want, ", got ", t.scalar_type());
}

// The reference launches on the legacy default stream without a device guard; here the
// launch goes to torch's current stream of the tensor's device, under a guard.
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(const at::Tensor &t)
    {
        (void)hipGetDevice(&prev);
        if (prev != t.get_device()) (void)hipSetDevice(t.get_device());
        else prev = -1;
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
inline tpu3_stream_t stream_of(const at::Tensor &t)
{
    return (tpu3_stream_t)c10::hip::getCurrentHIPStream(t.get_device()).stream();
}
inline void raise_on(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed: ", tpu3_strerror(rc), " (code ", rc, ")");
}
inline int elem_size(const at::Tensor &t, const char *name)
{
    switch (t.scalar_type()) {
    case at::kHalf: return 2;
    case at::kFloat: return 4;
    case at::kDouble: return 8;
    default: TORCH_CHECK(false, name, " must be float16, float32 or float64");
    }
    return 0;
}

} // namespace tpu3ext
