// pybind11 extension module `voxel_pooling_ext` for the MI355X build: the compiled drop-in for the reference's
// ops/voxel_pooling/src/voxel_pooling_forward.cpp (PYBIND11_MODULE at :41-43, wrapper at :26-39).
//
// Same module name, same function name, same ten arguments, same in-place semantics and return value (1), so the
// reference's ops/voxel_pooling/voxel_pooling.py:41-52 calls it unmodified.  The body only validates the tensors the way
// the reference's CHECK_INPUT macros do (:12-18) and forwards raw device pointers + the current HIP stream to the C ABI of
// libsgv3d_hip.so (include/sgv3d_hip.h): no kernel lives here, no CUDA branch, nothing but hipcc + gfx950.
// A launch failure raises (TORCH_CHECK) instead of exit(-1) (voxel_pooling_forward_cuda.cu:51-55).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include "sgv3d_hip.h"

#define CHECK_DEVICE(x) TORCH_CHECK(x.is_cuda(), #x, " must be a CUDAtensor ")            // message of the reference, :12-13
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x, " must be contiguous ")    // :14-15
#define CHECK_INPUT(x) \
    CHECK_DEVICE(x);   \
    CHECK_CONTIGUOUS(x)

int voxel_pooling_forward_wrapper(int batch_size, int num_points, int num_channels, int num_voxel_x, int num_voxel_y,
                                  int num_voxel_z, at::Tensor geom_xyz_tensor, at::Tensor input_features_tensor,
                                  at::Tensor output_features_tensor, at::Tensor pos_memo_tensor) {
    CHECK_INPUT(geom_xyz_tensor);
    CHECK_INPUT(input_features_tensor);
    CHECK_INPUT(output_features_tensor);       // the reference leaves these two unchecked (:28-29); a stray view would be
    CHECK_INPUT(pos_memo_tensor);              // written through the wrong strides
    const int *geom_xyz = geom_xyz_tensor.data_ptr<int>();                 // data_ptr<T>() checks the dtype, like :30-33
    const float *input_features = input_features_tensor.data_ptr<float>();
    float *output_features = output_features_tensor.data_ptr<float>();
    int *pos_memo = pos_memo_tensor.data_ptr<int>();
    hipStream_t stream = c10::hip::getCurrentHIPStream(input_features_tensor.get_device()).stream();   // :35
    const int rc = sgv3d_voxel_pooling_forward(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, num_voxel_z,
                                               geom_xyz, input_features, output_features, pos_memo, (void *)stream);
    TORCH_CHECK(rc == 0, "voxel_pooling_forward_wrapper: ", sgv3d_last_error());
    return 1;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("voxel_pooling_forward_wrapper", &voxel_pooling_forward_wrapper, "voxel_pooling_forward_wrapper");
}
