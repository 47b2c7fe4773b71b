// `losses` -- compiled drop-in for the reference's extension module of the same name
// (losses/nmdistance.cpp:24-27: nmdistance_forward, nmdistance_backward; both return int, 1 = launched).
#include "ext_common.h"

using namespace tpu3ext;

static void f32(const at::Tensor &t, const char *name)
{
    check_input(t, name);
    check_dtype(t, at::kFloat, name);
}
static void i32(const at::Tensor &t, const char *name)
{
    check_input(t, name);
    check_dtype(t, at::kInt, name);
}

// nmdistance.cpp:12-14
static int nmdistance_forward(at::Tensor xyz1, at::Tensor xyz2, at::Tensor dist1, at::Tensor dist2, at::Tensor idx1,
                              at::Tensor idx2)
{
    f32(xyz1, "xyz1"); f32(xyz2, "xyz2"); f32(dist1, "dist1"); f32(dist2, "dist2");
    i32(idx1, "idx1"); i32(idx2, "idx2");
    const int64_t b = xyz1.size(0), n = xyz1.size(1), m = xyz2.size(1);
    TORCH_CHECK(xyz2.size(0) == b && dist1.numel() == b * n && dist2.numel() == b * m && idx1.numel() == b * n &&
                    idx2.numel() == b * m, "nmdistance_forward: tensor sizes do not match");
    DeviceScope scope(xyz1);
    const int rc = tpu3_nmdist_fwd_f32(stream_of(xyz1), (int)b, (int)n, (int)m, xyz1.data_ptr<float>(),
                                       xyz2.data_ptr<float>(), dist1.data_ptr<float>(), dist2.data_ptr<float>(),
                                       idx1.data_ptr<int32_t>(), idx2.data_ptr<int32_t>());
    if (rc < 0) raise_on(rc, "tpu3_nmdist_fwd_f32");
    return rc == 0 ? 1 : 0;
}

// nmdistance.cpp:17-21
static int nmdistance_backward(at::Tensor xyz1, at::Tensor xyz2, at::Tensor gradxyz1, at::Tensor gradxyz2,
                               at::Tensor graddist1, at::Tensor graddist2, at::Tensor idx1, at::Tensor idx2)
{
    f32(xyz1, "xyz1"); f32(xyz2, "xyz2"); f32(gradxyz1, "gradxyz1"); f32(gradxyz2, "gradxyz2");
    f32(graddist1, "graddist1"); f32(graddist2, "graddist2");
    i32(idx1, "idx1"); i32(idx2, "idx2");
    const int64_t b = xyz1.size(0), n = xyz1.size(1), m = xyz2.size(1);
    TORCH_CHECK(gradxyz1.numel() == b * n * 3 && gradxyz2.numel() == b * m * 3 && graddist1.numel() == b * n &&
                    graddist2.numel() == b * m, "nmdistance_backward: tensor sizes do not match");
    DeviceScope scope(xyz1);
    const int rc = tpu3_nmdist_bwd_f32(stream_of(xyz1), (int)b, (int)n, (int)m, xyz1.data_ptr<float>(),
                                       xyz2.data_ptr<float>(), gradxyz1.data_ptr<float>(), gradxyz2.data_ptr<float>(),
                                       graddist1.data_ptr<float>(), graddist2.data_ptr<float>(),
                                       idx1.data_ptr<int32_t>(), idx2.data_ptr<int32_t>());
    if (rc < 0) raise_on(rc, "tpu3_nmdist_bwd_f32");
    return rc == 0 ? 1 : 0;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "3PU `losses` extension module on lib3pu_hip.so (MI355X / gfx950)";
    m.def("nmdistance_forward", &nmdistance_forward, "chamfer forward (HIP)");
    m.def("nmdistance_backward", &nmdistance_backward, "chamfer backward (HIP)");
}
