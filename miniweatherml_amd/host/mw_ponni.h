// =====================================================================================================
// mw_ponni.h -- the ponni surface the surrogate experiment uses, over the C ABI, plus the experiment's module.
//
// ponni itself (mrnorman/ponni) is an empty submodule of the reference snapshot; what is mirrored here is what the reference CALLS
// (experiments/supercell_kessler_surrogate/custom_modules/microphysics_kessler_ponni.h:10-13, 40-45, 103-110, 189):
//
//   ponni::load_h5_weights<N>(file, group, dataset)                      one float32 dataset of a Keras HDF5 weight file  (mw_h5_read_f32)
//   ponni::Matvec<float>(weights (in, out)) / Bias<float>(vector) / Relu<float>(n, negative_slope)
//   ponni::create_inference_model(layers...) -> ponni::Inference<...>    .validate()  .print()  .forward_batch_parallel(float2d (num_in, batch))
//
//   custom_modules::Microphysics_Kessler  (the experiment's module, :14-279): NN inference beside the true Kessler step; the four mean
//   differences it prints (:266-269); the online switch of :273-276.
//
// The arithmetic runs in libmw_cdna4.so: forward_batch_parallel -> mw_ponni_forward (MFMA tiles for the 5 -> 10 -> 4 stack), the
// module's time_step -> mw_mlp_forward (scaling, both layers and un-scaling fused: 72 B of HBM traffic per cell) + mw_kessler_time_step.
// =====================================================================================================
#pragma once
#include "mw_facade.h"
#include <fstream>
#include <iostream>
#include <memory>
#include <tuple>

// An owning, reference-counted device array (what the reference's float2d / real2d temporaries are: yakl::Array) -- enough of it
// for the surrogate's call sites: dims, data(), size(), a DeviceView over it.
template <class T> struct DeviceArray {
  std::shared_ptr<T> mem;
  std::vector<int> dimension;
  DeviceArray() = default;
  DeviceArray(std::vector<int> dims) : dimension(std::move(dims)) {
    T *p = nullptr;
    if (hipMalloc((void **)&p, std::max<size_t>(1, size()) * sizeof(T)) != hipSuccess) endrun("ERROR: device allocation failed");
    mem = std::shared_ptr<T>(p, [](T *q) { (void)hipFree(q); });
  }
  T *data() const { return mem.get(); }
  size_t size() const { size_t n = 1; for (int d : dimension) n *= (size_t)d; return n; }
  int extent(int i) const { return dimension[i]; }
  DeviceView<T> view() const { return DeviceView<T>{mem.get(), dimension}; }
  std::vector<T> createHostCopy() const { std::vector<T> h(size()); (void)hipMemcpy(h.data(), data(), size() * sizeof(T), hipMemcpyDeviceToHost); return h; }
};
typedef DeviceArray<float> float2d;

namespace ponni {

// what load_h5_weights<N> returns in the reference: an N-dimensional float array (host side here: the weights are kernel arguments)
template <int N> struct Weights {
  std::vector<float> data;
  int dims[N > 0 ? N : 1] = {0};
  int extent(int i) const { return dims[i]; }
  size_t size() const { return data.size(); }
};

// ponni_load_h5_weights.h: load_h5_weights<N>(file, group, dataset), e.g. ("/dense_6/dense_6", "kernel:0")   (:103-107)
template <int N> inline Weights<N> load_h5_weights(const std::string &fname, const std::string &group, const std::string &dataset) {
  long long dims[8] = {0}; int nd = 0;
  mw_check(mw_h5_read_f32(fname.c_str(), group.c_str(), dataset.c_str(), nullptr, 0, dims, &nd));
  if (nd != N) endrun("ERROR: load_h5_weights<" + std::to_string(N) + ">: dataset " + group + "/" + dataset + " has " + std::to_string(nd) + " dimensions");
  Weights<N> w; long long n = 1;
  for (int i = 0; i < N; i++) { w.dims[i] = (int)dims[i]; n *= dims[i]; }
  w.data.resize((size_t)n);
  mw_check(mw_h5_read_f32(fname.c_str(), group.c_str(), dataset.c_str(), w.data.data(), n, dims, &nd));
  return w;
}

// ---- layers: default-constructible (the reference declares its MODEL type from default-constructed layers, :40-44) ----
template <class T> class Matvec {               // y = x W, W stored (inputs, outputs) like a Keras Dense kernel
  Weights<2> w;
 public:
  static constexpr int kind = 0;
  Matvec() = default;
  explicit Matvec(Weights<2> const &weights) : w(weights) {}
  int get_num_inputs() const { return w.dims[0]; }
  int get_num_outputs() const { return w.dims[1]; }
  const std::vector<float> &params() const { return w.data; }
  float slope() const { return 0.f; }
  const char *get_label() const { return "Matvec"; }
};
template <class T> class Bias {
  Weights<1> b;
 public:
  static constexpr int kind = 1;
  Bias() = default;
  explicit Bias(Weights<1> const &bias) : b(bias) {}
  int get_num_inputs() const { return b.dims[0]; }
  int get_num_outputs() const { return b.dims[0]; }
  const std::vector<float> &params() const { return b.data; }
  float slope() const { return 0.f; }
  const char *get_label() const { return "Bias"; }
};
template <class T> class Relu {                 // max(x, 0) + negative_slope * min(x, 0): LeakyReLU for negative_slope > 0 (:105)
  int n = 0; float negative_slope = 0;
  std::vector<float> none;
 public:
  static constexpr int kind = 2;
  Relu() = default;
  Relu(int num_inputs, float negative_slope_ = 0) : n(num_inputs), negative_slope(negative_slope_) {}
  int get_num_inputs() const { return n; }
  int get_num_outputs() const { return n; }
  const std::vector<float> &params() const { return none; }
  float slope() const { return negative_slope; }
  const char *get_label() const { return "Relu"; }
};

template <class... LAYERS> class Inference {
  std::tuple<LAYERS...> layers;
  std::vector<mw_ponni_layer_t> desc;           // what the C ABI takes: one record per layer + the weights back to back
  std::vector<float> params;
  template <class L> void add(const L &l) {
    mw_ponni_layer_t d; d.kind = L::kind; d.n_in = l.get_num_inputs(); d.n_out = l.get_num_outputs(); d.negative_slope = l.slope();
    d.offset = (int)params.size();
    params.insert(params.end(), l.params().begin(), l.params().end());
    desc.push_back(d);
  }
 public:
  Inference() = default;
  explicit Inference(LAYERS const &...ls) : layers(ls...) { (add(ls), ...); }
  static constexpr int num_layers = sizeof...(LAYERS);
  int get_num_inputs() const { return desc.empty() ? 0 : desc.front().n_in; }
  int get_num_outputs() const { return desc.empty() ? 0 : desc.back().n_out; }
  // every layer's input size must be its predecessor's output size, and every layer's weights must have its declared size
  void validate() const {
    if (desc.empty()) endrun("ERROR: Inference model has no layers");
    for (size_t l = 0; l < desc.size(); l++) {
      if (l && desc[l].n_in != desc[l - 1].n_out)
        endrun("ERROR: layer " + std::to_string(l) + " expects " + std::to_string(desc[l].n_in) + " inputs, but the layer before it has " + std::to_string(desc[l - 1].n_out) + " outputs");
      const size_t end = l + 1 < desc.size() ? (size_t)desc[l + 1].offset : params.size();
      const size_t need = desc[l].kind == 0 ? (size_t)desc[l].n_in * desc[l].n_out : desc[l].kind == 1 ? (size_t)desc[l].n_out : 0;
      if (end - (size_t)desc[l].offset != need) endrun("ERROR: layer " + std::to_string(l) + " holds the wrong number of parameters");
    }
  }
  void print() const {
    std::cout << "Inference model has " << desc.size() << " layers:\n";
    static const char *names[3] = {"Matvec", "Bias", "Relu"};
    for (size_t l = 0; l < desc.size(); l++) {
      std::cout << "  " << l + 1 << ": " << names[desc[l].kind] << " with " << desc[l].n_in << " inputs and " << desc[l].n_out << " outputs";
      if (desc[l].kind == 2) std::cout << " and negative_slope == " << desc[l].negative_slope;
      std::cout << "\n";
    }
  }
  // in: (num_inputs, batch) -> (num_outputs, batch), batch fastest   (:189)
  float2d forward_batch_parallel(const DeviceView<float> &in, void *stream = nullptr) const {
    if (in.dimension.size() != 2 || in.dimension[0] != get_num_inputs()) endrun("ERROR: forward_batch_parallel: the input must be (num_inputs, batch)");
    float2d out({get_num_outputs(), in.dimension[1]});
    mw_check(mw_ponni_forward(desc.data(), (int)desc.size(), params.data(), (int)params.size(), (long long)in.dimension[1], in.data(), out.data(), stream));
    return out;
  }
  float2d forward_batch_parallel(const float2d &in, void *stream = nullptr) const { return forward_batch_parallel(in.view(), stream); }
  const std::vector<mw_ponni_layer_t> &layer_records() const { return desc; }
  const std::vector<float> &parameters() const { return params; }
};

template <class... LAYERS> inline Inference<LAYERS...> create_inference_model(LAYERS const &...layers) { return Inference<LAYERS...>(layers...); }

} // namespace ponni

namespace custom_modules {

using ponni::Bias;
using ponni::Inference;
using ponni::Matvec;
using ponni::Relu;

// experiments/supercell_kessler_surrogate/custom_modules/microphysics_kessler_ponni.h:14-279.  The file names the reference reads
// from its YAML input (keras_weights_h5, nn_input_scaling, nn_output_scaling, :97-101) are coupler options of the same names here
// (no yaml-cpp in this build: the driver sets them from the input file).
class Microphysics_Kessler : public modules::Microphysics_Kessler {
  double *nn[4] = {nullptr, nullptr, nullptr, nullptr};        // temp_tmp, rho_v_tmp, rho_c_tmp, rho_r_tmp (:191-194)
  double *ws1024 = nullptr; size_t ncell = 0;
 public:
  int static constexpr ID_V = 0, ID_C = 1, ID_R = 2, MAX_LAYERS = 10;                                              // :29-32
  std::vector<double> scl_in, scl_out;                                                                          // (5,2), (4,2): min max rows
  typedef decltype(ponni::create_inference_model(Matvec<float>(), Bias<float>(), Relu<float>(), Matvec<float>(), Bias<float>())) MODEL;   // :40-44
  MODEL model;
  bool online = false;                           // true = lines :273-276 un-commented: the NN result replaces Kessler's
  double diff_rho_v = 0, diff_rho_c = 0, diff_rho_r = 0, diff_temp = 0;      // the four "Relative diff" numbers of the last step (:266-269)
  bool verbose = true;
  ~Microphysics_Kessler() { for (auto p : nn) if (p) (void)hipFree(p); if (ws1024) (void)hipFree(ws1024); }
  void init(core::Coupler &coupler) {                                                                           // :66-146
    modules::Microphysics_Kessler::init(coupler);                                                               // tracers, precl, options (:76-93)
    auto keras_weights_h5 = coupler.get_option<std::string>("keras_weights_h5");
    auto nn_input_scaling = coupler.get_option<std::string>("nn_input_scaling");
    auto nn_output_scaling = coupler.get_option<std::string>("nn_output_scaling");
    ponni::Matvec<float> matvec_1(ponni::load_h5_weights<2>(keras_weights_h5, "/dense_6/dense_6", "kernel:0"));    // :103-107
    ponni::Bias<float> bias_1(ponni::load_h5_weights<1>(keras_weights_h5, "/dense_6/dense_6", "bias:0"));
    ponni::Relu<float> relu_1(bias_1.get_num_outputs(), 0.1);
    ponni::Matvec<float> matvec_2(ponni::load_h5_weights<2>(keras_weights_h5, "/dense_7/dense_7", "kernel:0"));
    ponni::Bias<float> bias_2(ponni::load_h5_weights<1>(keras_weights_h5, "/dense_7/dense_7", "bias:0"));
    this->model = ponni::create_inference_model(matvec_1, bias_1, relu_1, matvec_2, bias_2);                       // :109
    model.validate();
    if (verbose) model.print();
    if (model.get_num_inputs() != 5 || model.get_num_outputs() != 4) endrun("ERROR: the surrogate maps 5 inputs to 4 outputs");
    scl_in.assign(10, 0); scl_out.assign(8, 0);                                                                  // :113-135
    std::ifstream f1(nn_input_scaling);
    if (!f1) endrun("ERROR: cannot open " + nn_input_scaling);
    for (int j = 0; j < 5; j++) for (int i = 0; i < 2; i++) f1 >> scl_in[j * 2 + i];
    std::ifstream f2(nn_output_scaling);
    if (!f2) endrun("ERROR: cannot open " + nn_output_scaling);
    for (int j = 0; j < 4; j++) for (int i = 0; i < 2; i++) f2 >> scl_out[j * 2 + i];
    if (!f1 || !f2) endrun("ERROR: the scaling files must hold 5 and 4 rows of `min max`");
    ncell = (size_t)coupler.get_nz() * coupler.get_ny() * coupler.get_nx() * coupler.get_nens();
    for (auto &p : nn) if (hipMalloc((void **)&p, ncell * sizeof(double)) != hipSuccess) endrun("surrogate: allocation failed");
    if (hipMalloc((void **)&ws1024, 1024 * sizeof(double)) != hipSuccess) endrun("surrogate: allocation failed");
  }
  // NN inference on the state BEFORE the Kessler step, then the true step, then the mean differences (:149-278)
  void time_step(core::Coupler &coupler, real dt) {
    auto &dm = coupler.get_data_manager_readwrite();
    double *temp = dm.get<real>("temp").data(), *rho_v = dm.get<real>("water_vapor").data(), *rho_c = dm.get<real>("cloud_liquid").data(),
           *rho_r = dm.get<real>("precip_liquid").data();
    const double *rho_d = dm.get<real const>("density_dry").data();
    const auto &L = model.layer_records(); const auto &P = model.parameters();
    mw_check(mw_mlp_forward((long long)ncell, temp, rho_d, rho_v, rho_c, rho_r, P.data() + L[0].offset, P.data() + L[1].offset,
                            P.data() + L[3].offset, P.data() + L[4].offset, scl_in.data(), scl_out.data(), nn[0], nn[1], nn[2], nn[3], nullptr));   // :176-202
    modules::Microphysics_Kessler::time_step(coupler, dt);                                                       // :204-262
    mw_check(mw_mean_diff((long long)ncell, nn[1], rho_v, ws1024, &diff_rho_v, nullptr));                        // :266-269
    mw_check(mw_mean_diff((long long)ncell, nn[2], rho_c, ws1024, &diff_rho_c, nullptr));
    mw_check(mw_mean_diff((long long)ncell, nn[3], rho_r, ws1024, &diff_rho_r, nullptr));
    mw_check(mw_mean_diff((long long)ncell, nn[0], temp, ws1024, &diff_temp, nullptr));
    if (verbose) {
      std::cout << "Relative diff rho_v: " << diff_rho_v << "\n" << "Relative diff rho_c: " << diff_rho_c << "\n"
                << "Relative diff rho_r: " << diff_rho_r << "\n" << "Relative diff temp : " << diff_temp << "\n";
    }
    if (online) {                                                                                                // :273-276
      (void)hipMemcpy(temp, nn[0], ncell * 8, hipMemcpyDeviceToDevice); (void)hipMemcpy(rho_v, nn[1], ncell * 8, hipMemcpyDeviceToDevice);
      (void)hipMemcpy(rho_c, nn[2], ncell * 8, hipMemcpyDeviceToDevice); (void)hipMemcpy(rho_r, nn[3], ncell * 8, hipMemcpyDeviceToDevice);
    }
  }
  const double *nn_temp() const { return nn[0]; }  const double *nn_rho_v() const { return nn[1]; }
  const double *nn_rho_c() const { return nn[2]; }  const double *nn_rho_r() const { return nn[3]; }
};

} // namespace custom_modules
