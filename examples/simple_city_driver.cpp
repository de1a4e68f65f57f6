// The reference's simple_city caller, experiments/simple_city/driver.cpp:32-84, against the MI355X-native modules through
// miniweatherml_amd/host/mw_facade.h.  Parameters that the reference reads from YAML come from argv:
//     simple_city_driver nx_glob ny_glob nz nens xlen ylen zlen nsteps init_data out_prefix avg_path [out_every]
// Loop: horiz_sponge.apply(x1,x2) -> dycore.time_step -> sponge_layer(dt, 1) -> time_averager.accumulate (:72-75); the dycore's
// output() is called at etime 0 and every out_every steps; time_averager.finalize writes avg_path.
// Prints max|u|, the serial sum of density_dry; used by tests/test_gpu_cpp_facade.py.
#include "../miniweatherml_amd/host/mw_facade.h"
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv) {
  if (argc < 12) { fprintf(stderr, "usage: %s nx_glob ny_glob nz nens xlen ylen zlen nsteps init_data out_prefix avg_path [out_every]\n", argv[0]); return 2; }
  size_t nx_glob = atoll(argv[1]), ny_glob = atoll(argv[2]);
  int nz = atoi(argv[3]), nens = atoi(argv[4]);
  real xlen = atof(argv[5]), ylen = atof(argv[6]), zlen = atof(argv[7]);
  int nsteps = atoi(argv[8]);
  int out_every = argc > 12 ? atoi(argv[12]) : 0;
  try {
    core::Coupler coupler;
    coupler.set_option<std::string>("out_prefix", argv[10]);
    coupler.set_option<std::string>("init_data", argv[9]);
    coupler.set_option<real>("out_freq", -1.);
    coupler.set_option<bool>("enable_gravity", false);
    coupler.distribute_mpi_and_allocate_coupled_state(nz, ny_glob, nx_glob, nens);        // driver.cpp:41
    coupler.set_grid(xlen, ylen, zlen);                                                   // :44
    modules::Dynamics_Euler_Stratified_WenoFV dycore;
    custom_modules::Horizontal_Sponge horiz_sponge;
    custom_modules::Time_Averager time_averager;
    coupler.add_tracer("water_vapor", "water_vapor", true, true);                         // :55-56
    auto wv = coupler.get_data_manager_readwrite().get<real>("water_vapor");
    (void)hipMemset(wv.data(), 0, wv.size() * sizeof(real));
    dycore.init(coupler);                                                                 // :59-61
    horiz_sponge.init(coupler, 10, 1.);
    time_averager.init(coupler);
    real etime = 0;
    if (out_every > 0) dycore.output(coupler, etime);
    for (int s = 0; s < nsteps; s++) {                                                    // :66-79
      real dtphys = dycore.compute_time_step(coupler);
      horiz_sponge.apply(coupler, dtphys, true, true, false, false);
      dycore.time_step(coupler, dtphys);
      modules::sponge_layer(coupler, dtphys, 1);
      time_averager.accumulate(coupler, dtphys);
      etime += dtphys;
      if (out_every > 0 && (s + 1) % out_every == 0) dycore.output(coupler, etime);
    }
    time_averager.finalize(coupler, argv[11]);                                            // :82
    (void)hipDeviceSynchronize();
    auto &dm = coupler.get_data_manager_readwrite();
    size_t n = (size_t)nz * coupler.get_ny() * coupler.get_nx() * nens;
    std::vector<double> u(n), r(n);
    (void)hipMemcpy(u.data(), dm.get<real>("uvel").data(), n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(r.data(), dm.get<real>("density_dry").data(), n * 8, hipMemcpyDeviceToHost);
    double maxu = 0, sum = 0;
    for (size_t i = 0; i < n; i++) { maxu = std::max(maxu, std::fabs(u[i])); sum += r[i]; }
    printf("etime %.17g maxu %.17e sum_density_dry %.17e\n", etime, maxu, sum);
  } catch (std::exception &e) { fprintf(stderr, "endrun: %s\n", e.what()); return 1; }
  return 0;
}
