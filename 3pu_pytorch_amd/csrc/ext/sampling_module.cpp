// `sampling` -- compiled drop-in for the reference's extension module of the same name
// (sampling/sampling.cpp:83-88: furthest_sampling, gather_forward, gather_backward, ball_query).
#include "ext_common.h"

using namespace tpu3ext;

// (int b, int n, int m, Tensor input[b,n,3] f32, Tensor temp[b,n] f32 (=1e10), Tensor idx[b,m] i32) -> idx
// sampling.cpp:26-35; kernel sampling_cuda.cu:103-174
static at::Tensor furthest_sampling(int b, int n, int m, at::Tensor input, at::Tensor temp, at::Tensor idx)
{
    check_input(input, "input");
    check_input(temp, "temp");
    check_input(idx, "idx");
    check_dtype(input, at::kFloat, "input");
    check_dtype(temp, at::kFloat, "temp");
    check_dtype(idx, at::kInt, "idx");
    TORCH_CHECK(input.numel() == (int64_t)b * n * 3 && temp.numel() == (int64_t)b * n && idx.numel() == (int64_t)b * m,
                "furthest_sampling: tensor sizes do not match (b,n,m)=(", b, ",", n, ",", m, ")");
    DeviceScope scope(input);
    raise_on(tpu3_fps_f32(stream_of(input), b, n, m, input.data_ptr<float>(), temp.data_ptr<float>(),
                          idx.data_ptr<int32_t>()), "tpu3_fps_f32");
    return idx;
}

// (int b, int c, int n, int npoints, Tensor points[b,c,n], Tensor idx[b,npoints] i32, Tensor out[b,c,npoints]) -> out
// sampling.cpp:37-45
static at::Tensor gather_forward(int b, int c, int n, int npoints, at::Tensor points_tensor, at::Tensor idx_tensor,
                                 at::Tensor out_tensor)
{
    check_input(points_tensor, "points_tensor");
    check_input(idx_tensor, "idx_tensor");
    check_input(out_tensor, "out_tensor");
    check_dtype(idx_tensor, at::kInt, "idx_tensor");
    const int es = elem_size(points_tensor, "points_tensor");
    TORCH_CHECK(out_tensor.scalar_type() == points_tensor.scalar_type(), "gather_forward: points/out must share a dtype");
    TORCH_CHECK(points_tensor.numel() == (int64_t)b * c * n && idx_tensor.numel() == (int64_t)b * npoints &&
                    out_tensor.numel() == (int64_t)b * c * npoints, "gather_forward: tensor sizes do not match");
    DeviceScope scope(points_tensor);
    raise_on(tpu3_gather_fwd(stream_of(points_tensor), b, c, n, npoints, es, points_tensor.data_ptr(),
                             idx_tensor.data_ptr<int32_t>(), out_tensor.data_ptr()), "tpu3_gather_fwd");
    return out_tensor;
}

// (int b, int c, int n, int npoints, Tensor grad_out[b,c,npoints], Tensor idx i32, Tensor grad_points[b,c,n] zeros)
// -> grad_points      sampling.cpp:47-53
static at::Tensor gather_backward(int b, int c, int n, int npoints, at::Tensor grad_out_tensor, at::Tensor idx_tensor,
                                  at::Tensor grad_points_tensor)
{
    check_input(grad_out_tensor, "grad_out_tensor");
    check_input(idx_tensor, "idx_tensor");
    check_input(grad_points_tensor, "grad_points_tensor");
    check_dtype(idx_tensor, at::kInt, "idx_tensor");
    const int es = elem_size(grad_out_tensor, "grad_out_tensor");
    TORCH_CHECK(grad_points_tensor.scalar_type() == grad_out_tensor.scalar_type(),
                "gather_backward: grad tensors must share a dtype");
    TORCH_CHECK(grad_out_tensor.numel() == (int64_t)b * c * npoints && idx_tensor.numel() == (int64_t)b * npoints &&
                    grad_points_tensor.numel() == (int64_t)b * c * n, "gather_backward: tensor sizes do not match");
    DeviceScope scope(grad_out_tensor);
    raise_on(tpu3_gather_bwd(stream_of(grad_out_tensor), b, c, n, npoints, es, grad_out_tensor.data_ptr(),
                             idx_tensor.data_ptr<int32_t>(), grad_points_tensor.data_ptr()), "tpu3_gather_bwd");
    return grad_points_tensor;
}

// (Tensor query[b,m,3], Tensor xyz[b,n,3], float radius, int nsample) -> idx[b,m,nsample] i32
// sampling.cpp:59-81 ("CPU not supported" there too)
static at::Tensor ball_query(at::Tensor query, at::Tensor xyz, const float radius, const int nsample)
{
    check_input(query, "query");
    check_input(xyz, "xyz");
    TORCH_CHECK((xyz.scalar_type() == at::kFloat || xyz.scalar_type() == at::kDouble) &&
                    query.scalar_type() == xyz.scalar_type(), "ball_query: query/xyz must both be float32 or float64");
    at::Tensor idx = at::empty({query.size(0), query.size(1), nsample}, query.options().dtype(at::kInt));
    DeviceScope scope(query);
    raise_on(tpu3_ball_query(stream_of(query), (int)xyz.size(0), (int)xyz.size(1), (int)query.size(1), radius, nsample,
                             xyz.scalar_type() == at::kFloat ? 4 : 8, query.data_ptr(), xyz.data_ptr(),
                             idx.data_ptr<int32_t>()), "tpu3_ball_query");
    return idx;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "3PU `sampling` extension module on lib3pu_hip.so (MI355X / gfx950)";
    m.def("furthest_sampling", &furthest_sampling, "furthest point sampling (no gradient)");
    m.def("gather_forward", &gather_forward, "gather npoints points along an axis");
    m.def("gather_backward", &gather_backward, "gather npoints points along an axis backward");
    m.def("ball_query", &ball_query, "ball query");
}
