// A reference-style module over the C++ facade that does what the reference's modules do with core::MultiField
// (model/core/MultipleFields.h:10-96): aggregate several DataManager fields and loop over them inside ONE kernel -- sponge_layer.h:32-76
// and column_nudging.h:50-65 are written this way -- plus core::Coupler::clone_into (coupler.h:85-106) and DataManager::clone_into
// (DataManager.h:79-103).  Compiled as HIP (the kernels below take the MultiField by value, like the reference's YAKL_LAMBDAs capture it).
//     multifield_module [nx ny nz nens]
// Prints "multifield ok ..." and exits 0 when every check holds; used by tests/test_gpu_cpp_facade.py.
#include "../miniweatherml_amd/host/mw_facade.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// column_nudging.h:62-65 in shape: parallel_for(Bounds<5>(num_fields,nz,ny,nx,nens)) { state(l,k,j,i,iens) += ... }
__global__ void k_add_per_field(core::MultiField<real, 4> state, const real *inc) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = state.get_num_fields();
  const int nz = state.get_field(0).extent(0), ny = state.get_field(0).extent(1), nx = state.get_field(0).extent(2), nens = state.get_field(0).extent(3);
  const long long ncell = (long long)nz * ny * nx * nens;
  if (t >= ncell * nf) return;
  const int l = (int)(t / ncell);
  long long r = t - (long long)l * ncell;
  const int iens = (int)(r % nens); r /= nens;
  const int i = (int)(r % nx); r /= nx;
  const int j = (int)(r % ny);
  const int k = (int)(r / ny);
  state(l, k, j, i, iens) += inc[l];
}
// a read-only aggregate (core::MultiField<real const,4>, column_nudging.h:28-33): out(k,j,i,iens) = sum over the fields
__global__ void k_sum_fields(core::MultiField<real const, 4> state, FieldView<real, 4> out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)out.size()) return;
  long long r = t;
  const int iens = (int)(r % out.extent(3)); r /= out.extent(3);
  const int i = (int)(r % out.extent(2)); r /= out.extent(2);
  const int j = (int)(r % out.extent(1));
  const int k = (int)(r / out.extent(1));
  real s = 0;
  for (int l = 0; l < state.get_num_fields(); l++) s += state(l, k, j, i, iens);
  out(k, j, i, iens) = s;
}

static std::vector<double> host(const double *dev, size_t n) { std::vector<double> h(n); (void)hipMemcpy(h.data(), dev, n * 8, hipMemcpyDeviceToHost); return h; }
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "multifield_module: check failed at line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main(int argc, char **argv) {
  try {
    const int nx = argc > 1 ? atoi(argv[1]) : 16, ny = argc > 2 ? atoi(argv[2]) : 12, nz = argc > 3 ? atoi(argv[3]) : 10, nens = argc > 4 ? atoi(argv[4]) : 2;
    core::Coupler coupler;                                                                    // supercell_example/driver.cpp:34-62
    coupler.set_option<std::string>("out_prefix", "test");
    coupler.set_option<std::string>("init_data", "supercell");
    coupler.set_option<real>("out_freq", -1.);
    coupler.distribute_mpi_and_allocate_coupled_state(nz, ny, nx, nens);
    coupler.set_grid(500. * nx, 500. * ny, 20000.);
    modules::Microphysics_Kessler micro;
    modules::Dynamics_Euler_Stratified_WenoFV dycore;
    micro.init(coupler);
    dycore.init(coupler);
    modules::perturb_temperature(coupler);
    auto &dm = coupler.get_data_manager_readwrite();
    const size_t n = (size_t)nz * ny * nx * nens;

    // ---- column_nudging.h:50-55, spelled as in the reference
    core::MultiField<real, 4> state;
    state.add_field(dm.get<real, 4>("density_dry"));
    state.add_field(dm.get<real, 4>("uvel"));
    state.add_field(dm.get<real, 4>("vvel"));
    state.add_field(dm.get<real, 4>("temp"));
    state.add_field(dm.get<real, 4>("water_vapor"));
    CHECK(state.get_num_fields() == 5);
    CHECK(state.get_field(3).data() == dm.get<real>("temp").data());
    CHECK(state.get_field(0).extent(0) == nz && state.get_field(0).extent(1) == ny && state.get_field(0).extent(2) == nx && state.get_field(0).extent(3) == nens);
    { core::MultiField<real, 4> copy(state), assigned; assigned = state;                        // MultipleFields.h:17-30
      CHECK(copy.get_num_fields() == 5 && assigned.get_num_fields() == 5 && copy.get_field(4).data() == state.get_field(4).data() &&
            assigned.get_field(1).data() == state.get_field(1).data()); }
    const char *names[5] = {"density_dry", "uvel", "vvel", "temp", "water_vapor"};
    std::vector<std::vector<double>> before;
    for (auto nm : names) before.push_back(host(dm.get<real>(nm).data(), n));
    const double inc_h[5] = {0.125, -2.0, 0.5, 3.0, 1.0e-3};
    double *inc = nullptr;
    CHECK(hipMalloc((void **)&inc, sizeof(inc_h)) == hipSuccess && hipMemcpy(inc, inc_h, sizeof(inc_h), hipMemcpyHostToDevice) == hipSuccess);
    hipLaunchKernelGGL(k_add_per_field, dim3((unsigned)((n * 5 + 255) / 256)), dim3(256), 0, nullptr, state, inc);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    for (int l = 0; l < 5; l++) {
      auto after = host(dm.get<real>(names[l]).data(), n);
      for (size_t c = 0; c < n; c++) CHECK(after[c] == before[l][c] + inc_h[l]);
    }
    // ---- a read-only aggregate of the same entries (:28-33) and a rank-4 output view
    core::MultiField<real const, 4> cstate;
    for (auto nm : names) cstate.add_field(dm.get<real const, 4>(nm));
    dm.register_and_allocate<real>("field_sum", "sum of the five nudged fields", {nz, ny, nx, nens});
    hipLaunchKernelGGL(k_sum_fields, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, cstate, dm.get<real, 4>("field_sum"));
    CHECK(hipDeviceSynchronize() == hipSuccess);
    { auto s = host(dm.get<real>("field_sum").data(), n);
      std::vector<std::vector<double>> f; for (auto nm : names) f.push_back(host(dm.get<real>(nm).data(), n));
      for (size_t c = 0; c < n; c++) { double r = 0; for (int l = 0; l < 5; l++) r += f[l][c]; CHECK(s[c] == r); } }
    // ---- the rank is part of the request (DataManager.h:263-268): a rank-3 view of a rank-4 entry ends the run
    bool threw = false;
    try { (void)dm.get<real, 3>("density_dry"); } catch (std::exception &) { threw = true; }
    CHECK(threw);
    // ---- more fields than max_fields is an error here (the reference writes past its SArray)
    threw = false;
    try { core::MultiField<real, 4> big; for (int q = 0; q <= max_fields; q++) big.add_field(dm.get<real, 4>("uvel")); } catch (std::exception &) { threw = true; }
    CHECK(threw);

    // ---- Coupler::clone_into (coupler.h:85-106)
    core::Coupler clone;
    coupler.clone_into(clone);
    CHECK(clone.get_nx() == nx && clone.get_ny() == ny && clone.get_nz() == nz && clone.get_nens() == nens);
    CHECK(clone.get_xlen() == coupler.get_xlen() && clone.get_ylen() == coupler.get_ylen() && clone.get_zlen() == coupler.get_zlen());
    CHECK(clone.get_dx() == coupler.get_dx() && clone.get_nx_glob() == coupler.get_nx_glob() && clone.get_ny_glob() == coupler.get_ny_glob());
    CHECK(clone.get_nranks() == 1 && clone.get_myrank() == 0 && clone.get_px() == 0 && clone.get_nproc_y() == 1);
    CHECK(clone.get_tracer_names() == coupler.get_tracer_names() && clone.get_num_tracers() == 3);
    { std::string d; bool found, pos, adds; clone.get_tracer_info("cloud_liquid", d, found, pos, adds); CHECK(found && pos && adds); }
    CHECK(!clone.option_exists("init_data"));                                                  // options are not cloned (:85-106 copies no options)
    auto &cdm = clone.get_data_manager_readwrite();
    for (auto nm : {"density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid", "precl", "hy_dens_cells", "field_sum"}) {
      CHECK(cdm.entry_exists(nm));
      auto a = dm.get<real const>(nm), b = cdm.get<real const>(nm);
      CHECK(a.data() != b.data() && a.dimension == b.dimension);                               // its own allocation, same shape
      CHECK(host(a.data(), a.size()) == host(b.data(), b.size()));                             // same values
    }
    CHECK(cdm.entry_is_dirty("temp") == dm.entry_is_dirty("temp"));                            // the dirty flag travels (:100)
    // the copy is independent: a dycore step on the original leaves the clone alone
    auto clone_T = host(cdm.get<real const>("temp").data(), n);
    real dt = dycore.compute_time_step(coupler);
    dycore.time_step(coupler, dt);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    CHECK(host(cdm.get<real const>("temp").data(), n) == clone_T);
    CHECK(host(dm.get<real const>("temp").data(), n) != clone_T);
    // ... and modules can run on the clone: they reach the clone's own fields through the clone's DataManager and grid
    modules::sponge_layer(clone, dt);
    modules::perturb_temperature(clone);                                                       // (adds the bubble once more: the clone's temp changes)
    CHECK(hipDeviceSynchronize() == hipSuccess);
    CHECK(host(cdm.get<real const>("temp").data(), n) != clone_T);
    CHECK(clone.get_data_manager_readonly().validate_all(false) == 0);
    (void)hipFree(inc);
    printf("multifield ok fields %d cells %zu clone_entries_checked 11\n", state.get_num_fields(), n);
  } catch (std::exception &e) { fprintf(stderr, "endrun: %s\n", e.what()); return 1; }
  return 0;
}
