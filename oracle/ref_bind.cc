// ref_bind.cc -- our own thin pybind11 module around the two translation units
// of the reference that compile from their own sources in this image:
//     /root/reference/src/quantize_utils.cc   (quantize / dequantize / down_scale)
//     /root/reference/src/functional.cc       (relu<u8>, max_pool2d<u8>)
//     /root/reference/src/calibrator.cc       (Calibrator::sample / get_range)
// They are compiled WHERE THEY LIE by oracle/Makefile (target `ref`) into
// oracle/_ref/ (git-ignored).  Nothing from the reference is copied into this
// repository.  The rest of the reference (conv2d.cc, fully_connected.cc,
// layer.cc, pybind11.cc) includes mkl.h, which the image lacks, so it is
// unbuildable here and is NOT stood in for.
//
// TEST INFRASTRUCTURE ONLY: used by tests/golden/make_golden.py (in this
// container) to produce golden vectors, and by tests/ to validate
// oracle/i8ie_oracle.c.  /root/reference does not exist on the GPU box.
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "pybind11/numpy.h"
#include "pybind11/pybind11.h"
#include "pybind11/stl.h"
#include "calibrator.h"      // reference header (include/calibrator.h)
#include "quantize_utils.h"  // reference header (include/quantize_utils.h)
#include "tensor.h"          // reference header (include/tensor.h)

namespace py = pybind11;

// reference src/functional.cc:78 -- registers relu / max_pool2d overloads for
// Tensor<float>, Tensor<u8_t>, Tensor<s8_t> on a module.
void declare_tensor_funcs(py::module&);

namespace {

inline std::vector<ssize_t> shape_of(const py::array& a) {
  return std::vector<ssize_t>(a.shape(), a.shape() + a.ndim());
}

template <typename T>
py::array as_numpy(Tensor<T>& t) {
  return py::array(t.shape(), t.data(), t.cap());
}

template <typename T>
Tensor<T>* make_q(py::array_t<T, py::array::c_style | py::array::forcecast> a,
                  float scale, int zp) {
  auto* t = new Tensor<T>(shape_of(a));
  std::memcpy(t->data(), a.data(), sizeof(T) * (size_t)a.size());
  t->scale() = scale;
  t->zero_point() = (u8_t)zp;
  return t;
}

template <typename T>
void bind_tensor(py::module& m, const char* name) {
  py::class_<Tensor<T>>(m, name)
      .def("numpy", [](Tensor<T>& t) { return as_numpy(t); })
      .def("scale", [](Tensor<T>& t) { return (float)t.scale(); })
      .def("zero_point", [](Tensor<T>& t) { return (int)t.zero_point(); });
}

}  // namespace

PYBIND11_MODULE(_i8ie_ref_partial, m) {
  m.doc() = "reference quantize_utils.cc + functional.cc, compiled in place";
  bind_tensor<float>(m, "RefTensorF32");
  bind_tensor<u8_t>(m, "RefTensorU8");
  bind_tensor<s8_t>(m, "RefTensorS8");

  m.def("f32", [](py::array_t<float, py::array::c_style | py::array::forcecast> a) {
    return new Tensor<float>(a);  // reference include/tensor.h:40-47
  }, py::return_value_policy::take_ownership);
  m.def("u8", &make_q<u8_t>, py::return_value_policy::take_ownership);

  // reference src/quantize_utils.cc:44-52
  m.def("quantize", [](Tensor<float>& in, float scale, int zp) -> Tensor<u8_t>&& {
    return std::move(quantize(in, scale, (u8_t)zp));
  });
  // reference src/quantize_utils.cc:54-58
  m.def("dequantize", [](Tensor<u8_t>& in) -> Tensor<float>&& {
    return std::move(dequantize(in));
  });
  // reference src/quantize_utils.cc:27-36 (pointer-level requantiser)
  m.def("down_scale",
        [](py::array_t<int, py::array::c_style | py::array::forcecast> acc, float sa,
           float sb, float sc, int zp_c) {
          py::array_t<u8_t> out(shape_of(acc));
          down_scale(out.mutable_data(), const_cast<int*>(acc.data()), acc.size(), sa,
                     sb, sc, (u8_t)zp_c);
          return out;
        });
  // reference src/calibrator.cc:6-40.  sample() is deterministic while the reservoir is filling (the first
  // num_samples = 1000 values are appended in order; only later values draw from std::random_device), and
  // get_range() always is -- but it sorts all 1000 slots, so it is only meaningful once exactly 1000 values
  // were appended.  `chunks` are fed through sample() one after the other and must total 1000 values.
  m.def("calib_range", [](std::vector<py::array_t<float, py::array::c_style | py::array::forcecast>> chunks,
                          float quantile) {
    ssize_t total = 0;
    for (auto& c : chunks) total += c.size();
    if (total != num_samples) throw std::runtime_error("calib_range: feed exactly num_samples values");
    auto cal = std::make_unique<Calibrator>();
    for (auto& c : chunks) cal->sample(const_cast<float*>(c.data()), c.size());
    auto [scale, zp] = cal->get_range(quantile);
    return py::make_tuple(scale, (int)zp);
  });
  // reference src/functional.cc:66-82: relu(T), max_pool2d(T, kernel_size, strides)
  declare_tensor_funcs(m);
}
