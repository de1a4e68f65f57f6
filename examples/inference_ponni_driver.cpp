// The reference's surrogate driver, experiments/supercell_kessler_surrogate/inference_ponni.cpp:9-86, against the MI355X-native
// modules: the ponni-shaped surface of miniweatherml_amd/host/mw_ponni.h (load_h5_weights, Matvec / Bias / Relu,
// create_inference_model, forward_batch_parallel) and custom_modules::Microphysics_Kessler (NN inference beside the true Kessler).
//     inference_ponni_driver input.yaml [max_steps [online]]
// online = 1: the four deep_copy_to lines of microphysics_kessler_ponni.h:273-276 un-commented -- the NN result replaces Kessler's.
// input.yaml: the reference's keys (inputs/input_euler3d.yaml), flat `key: value` lines.  Prints the state checksums, the four mean
// NN - Kessler differences of the last step, and a checksum of model.forward_batch_parallel on the final state's scaled inputs;
// used by tests/test_gpu_cpp_facade.py.
#include "../miniweatherml_amd/host/mw_ponni.h"
#include <cstdio>
#include <cstdlib>
#include <sstream>

static std::map<std::string, std::string> read_flat_yaml(const std::string &path) {
  std::ifstream f(path);
  if (!f) endrun("ERROR: Invalid YAML input file");
  std::map<std::string, std::string> kv; std::string line;
  auto trim = [](std::string s) { size_t a = s.find_first_not_of(" \t\"'"), b = s.find_last_not_of(" \t\r\"'"); return a == std::string::npos ? std::string() : s.substr(a, b - a + 1); };
  while (std::getline(f, line)) {
    size_t h = line.find('#'); if (h != std::string::npos) line = line.substr(0, h);
    size_t c = line.find(':'); if (c == std::string::npos) continue;
    std::string k = trim(line.substr(0, c)), v = trim(line.substr(c + 1));
    if (!k.empty() && !v.empty()) kv[k] = v;
  }
  return kv;
}

int main(int argc, char **argv) {
  if (argc <= 1) { fprintf(stderr, "ERROR: Must pass the input YAML filename as a parameter\n"); return 2; }
  try {
    std::string inFile(argv[1]);
    auto config = read_flat_yaml(inFile);
    auto need = [&](const char *k) { if (!config.count(k)) endrun(std::string("ERROR: missing key in the YAML input file: ") + k); return config[k]; };
    const int max_steps = argc > 2 ? atoi(argv[2]) : -1;
    real sim_time = atof(need("sim_time").c_str());                                         // inference_ponni.cpp:27-34
    size_t nx_glob = atoll(need("nx_glob").c_str()), ny_glob = atoll(need("ny_glob").c_str());
    int nz = atoi(need("nz").c_str());
    real xlen = atof(need("xlen").c_str()), ylen = atof(need("ylen").c_str()), zlen = atof(need("zlen").c_str());
    real dtphys_in = atof(need("dt_phys").c_str());
    int nens = 1;                                                                           // :36
    core::Coupler coupler;
    coupler.set_option<std::string>("out_prefix", need("out_prefix"));                      // :38-40
    coupler.set_option<std::string>("init_data", need("init_data"));
    coupler.set_option<real>("out_freq", atof(need("out_freq").c_str()));
    coupler.distribute_mpi_and_allocate_coupled_state(nz, ny_glob, nx_glob, nens);           // :44
    coupler.set_grid(xlen, ylen, zlen);                                                      // :47
    coupler.set_option<std::string>("standalone_input_file", inFile);                        // :50
    // (the reference's module re-opens the YAML file for these three, microphysics_kessler_ponni.h:97-101)
    coupler.set_option<std::string>("keras_weights_h5", need("keras_weights_h5"));
    coupler.set_option<std::string>("nn_input_scaling", need("nn_input_scaling"));
    coupler.set_option<std::string>("nn_output_scaling", need("nn_output_scaling"));

    modules::ColumnNudger column_nudger;                                                     // :54
    custom_modules::Microphysics_Kessler micro;                                              // :56
    modules::Dynamics_Euler_Stratified_WenoFV dycore;                                        // :58
    micro.verbose = false;
    micro.online = argc > 3 && atoi(argv[3]) != 0;                                           // (microphysics_kessler_ponni.h:273-276)
    micro.init(coupler);                                                                     // :61
    micro.model.print();
    dycore.init(coupler);                                                                    // :62
    column_nudger.set_column(coupler);                                                       // :63
    modules::perturb_temperature(coupler);                                                   // :64

    real etime = 0, dtphys = dtphys_in;
    int steps = 0;
    while (etime < sim_time && (max_steps < 0 || steps < max_steps)) {                       // :69-82
      if (dtphys_in <= 0.) dtphys = dycore.compute_time_step(coupler);
      if (etime + dtphys > sim_time) dtphys = sim_time - etime;
      dycore.time_step(coupler, dtphys);
      micro.time_step(coupler, dtphys);
      modules::sponge_layer(coupler, dtphys);
      column_nudger.nudge_to_column(coupler, dtphys);
      etime += dtphys; steps++;
    }
    (void)hipDeviceSynchronize();
    auto &dm = coupler.get_data_manager_readwrite();
    const size_t n = (size_t)nz * coupler.get_ny() * coupler.get_nx() * nens;
    auto host = [&](const double *dev) { std::vector<double> h(n); (void)hipMemcpy(h.data(), dev, n * 8, hipMemcpyDeviceToHost); return h; };
    std::vector<double> w = host(dm.get<real>("wvel").data()), T = host(dm.get<real>("temp").data()), nnT = host(micro.nn_temp());
    double maxw = 0, sumT = 0, sumNN = 0;
    for (size_t i = 0; i < n; i++) { maxw = std::max(maxw, std::fabs(w[i])); sumT += T[i]; sumNN += nnT[i]; }
    // the ponni call itself (:176-189): scaled fp32 inputs (num_in, nz*ncol) -> model.forward_batch_parallel
    const char *names[5] = {"temp", "density_dry", "water_vapor", "cloud_liquid", "precip_liquid"};
    std::vector<float> in_h(5 * n);
    for (int f = 0; f < 5; f++) {
      std::vector<double> v = host(dm.get<real>(names[f]).data());
      for (size_t i = 0; i < n; i++) in_h[f * n + i] = (float)((v[i] - micro.scl_in[f * 2]) / (micro.scl_in[f * 2 + 1] - micro.scl_in[f * 2]));
    }
    float2d ponni_in({5, (int)n});
    (void)hipMemcpy(ponni_in.data(), in_h.data(), in_h.size() * sizeof(float), hipMemcpyHostToDevice);
    auto ponni_out = micro.model.forward_batch_parallel(ponni_in);
    auto out_h = ponni_out.createHostCopy();
    double sum_out[4] = {0, 0, 0, 0};
    for (int o = 0; o < 4; o++) for (size_t i = 0; i < n; i++) sum_out[o] += (double)out_h[o * n + i];
    const long long bad = dm.validate_all(false);
    printf("steps %d etime %.17g maxw %.17e sum_temp %.17e sum_nn_temp %.17e diffs %.17e %.17e %.17e %.17e ponni_out %.17e %.17e %.17e %.17e validate_all %lld\n",
           steps, etime, maxw, sumT, sumNN, micro.diff_rho_v, micro.diff_rho_c, micro.diff_rho_r, micro.diff_temp, sum_out[0], sum_out[1], sum_out[2],
           sum_out[3], bad);
  } catch (std::exception &e) { fprintf(stderr, "endrun: %s\n", e.what()); return 1; }
  return 0;
}
