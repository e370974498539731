// Host-side helper of BEVHeight._stamp (sgv3d_amd/models/bev_height.py): the identity of the weights the packed HIP tensors and
// the captured graphs were made from is re-checked on EVERY forward -- the sum of the version counters, storage addresses and
// object ids of the ~860 parameters and buffers of the cfg-2 model.  In Python that walk is 0.2-0.3 ms, spent with the GPU idle
// when the reference harness's eval_step (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:242-258) runs one frame
// at a time; here it is one call over the modules' live ``_parameters`` / ``_buffers`` dicts.  No device code, no arithmetic of
// the path: a missing module makes the Python loop run instead.
#include <torch/extension.h>

#include <tuple>

namespace {

// dicts: list of dicts name -> Tensor | None.  Returns (tensors seen, sum of versions, sum of (address + id / 16)).
std::tuple<int64_t, int64_t, int64_t> tensor_stamp(const py::list &dicts) {
    int64_t n = 0, ver = 0, ptr = 0;
    for (const py::handle &d : dicts) {
        for (const auto &item : py::reinterpret_borrow<py::dict>(d)) {
            const py::handle v = item.second;
            if (v.is_none()) continue;
            const at::Tensor &t = THPVariable_Unpack(v.ptr());
            ++n;
            ver += static_cast<int64_t>(t._version());
            ptr += static_cast<int64_t>(reinterpret_cast<intptr_t>(t.data_ptr())) + (static_cast<int64_t>(reinterpret_cast<intptr_t>(v.ptr())) >> 4);
        }
    }
    return std::make_tuple(n, ver, ptr);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("tensor_stamp", &tensor_stamp, "(count, sum of _version, sum of data_ptr + id / 16) over the tensors of a list of dicts");
}
