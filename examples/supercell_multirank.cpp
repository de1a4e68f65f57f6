// A decomposed run from a plain C++ host: one process per GPU, the reference's 2-D block decomposition (coupler.h:127-179), halo
// strips over RCCL point-to-point (mw_dycore_use_rccl), the horizontal sums of sponge_layer / ColumnNudger over the same communicator
// (mw_dycore_rccl_allreduce_sum).  The loop is experiments/supercell_example/driver.cpp:66-79.
//     MW_RANK=r MW_WORLD=n MW_ID_FILE=/tmp/id supercell_multirank nx_glob ny_glob nz nens xlen ylen zlen nsteps
// Rank 0 writes the ncclUniqueId to MW_ID_FILE, the others wait for it (no MPI, no torch).  Rank r uses GPU r % (visible devices).
// The file is tied to the launch: it starts with a 32-byte token -- MW_RUN_ID, which the launcher sets to something unique per job (its
// PID, a time stamp) -- and a reader only accepts a file whose token is its own, so a file left behind by an earlier run is ignored
// instead of handing the ranks a dead id.  Joining the communicator has a wall-clock limit (MW_RCCL_TIMEOUT_S, default 120): a rank whose
// peers never arrive exits non-zero instead of hanging.
// Every rank prints max|w| of its block and the sum of its density_dry; rank 0 also the all-reduced total mass.
#include "../miniweatherml_amd/host/mw_facade.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <fstream>
#include <thread>

int main(int argc, char **argv) {
  if (argc < 9) { fprintf(stderr, "usage: MW_RANK=r MW_WORLD=n MW_ID_FILE=path %s nx_glob ny_glob nz nens xlen ylen zlen nsteps\n", argv[0]); return 2; }
  const int rank = getenv("MW_RANK") ? atoi(getenv("MW_RANK")) : 0, world = getenv("MW_WORLD") ? atoi(getenv("MW_WORLD")) : 1;
  const char *id_file = getenv("MW_ID_FILE");
  size_t nx_glob = atoll(argv[1]), ny_glob = atoll(argv[2]);
  int nz = atoi(argv[3]), nens = atoi(argv[4]);
  real xlen = atof(argv[5]), ylen = atof(argv[6]), zlen = atof(argv[7]);
  int nsteps = atoi(argv[8]);
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) endrun("no HIP device");
    if (hipSetDevice(rank % ndev) != hipSuccess) endrun("hipSetDevice failed");
    core::Coupler coupler;
    coupler.set_option<std::string>("out_prefix", "test");
    coupler.set_option<std::string>("init_data", "supercell");
    coupler.set_option<real>("out_freq", -1.);
    coupler.distribute_mpi_and_allocate_coupled_state(nz, ny_glob, nx_glob, nens, world, rank);
    coupler.set_grid(xlen, ylen, zlen);
    modules::Microphysics_Kessler micro;
    modules::Dynamics_Euler_Stratified_WenoFV dycore;
    micro.init(coupler);
    dycore.init(coupler);
    mw_allreduce_fn ar = nullptr; void *ar_ctx = nullptr;
    if (world > 1 || getenv("MW_FORCE_RCCL")) {
      if (!id_file) endrun("MW_ID_FILE is needed to hand the ncclUniqueId to the other ranks");
      unsigned char id[128];
      char token[32]; memset(token, 0, sizeof(token));
      if (const char *run = getenv("MW_RUN_ID")) strncpy(token, run, sizeof(token) - 1);
      else if (world > 1) fprintf(stderr, "[rank %d] warning: MW_RUN_ID is not set -- a stale %s of an earlier run cannot be told from this run's\n", rank, id_file);
      if (rank == 0) {
        (void)remove(id_file);                                  // (best effort; the token is what protects the readers)
        mw_check(mw_rccl_unique_id(id));
        std::string tmp = std::string(id_file) + ".tmp";
        { std::ofstream f(tmp, std::ios::binary); f.write(token, sizeof(token)); f.write((const char *)id, 128); }
        if (rename(tmp.c_str(), id_file) != 0) endrun("cannot publish the ncclUniqueId");
      } else {
        bool ok = false;
        for (int t = 0; t < 600 && !ok; t++) {                   // up to 60 s
          std::ifstream f(id_file, std::ios::binary);
          char got[32];
          if (f && f.read(got, sizeof(got)) && f.gcount() == (std::streamsize)sizeof(got) && !memcmp(got, token, sizeof(token)) &&
              f.read((char *)id, 128) && f.gcount() == 128) ok = true;
          else std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!ok) endrun("rank 0 never published an ncclUniqueId for this run (MW_RUN_ID token mismatch or no file)");
      }
      { // ncclCommInitRank blocks until every rank has joined: a limit, so that a missing peer ends the job instead of hanging it
        std::atomic<bool> joined{false};
        const double limit = getenv("MW_RCCL_TIMEOUT_S") ? atof(getenv("MW_RCCL_TIMEOUT_S")) : 120.0;
        std::thread watchdog([&joined, limit, rank]() {
          const auto t0 = std::chrono::steady_clock::now();
          while (!joined.load()) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
              fprintf(stderr, "[rank %d] the RCCL communicator was not formed within %.0f s: giving up\n", rank, limit);
              _Exit(3);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
          }
        });
        try { dycore.use_rccl(coupler, id); } catch (...) { joined = true; watchdog.join(); throw; }
        joined = true; watchdog.join();
      }
      int n = 0, r = 0, lanes = 0;
      mw_check(mw_dycore_rccl_info(dycore.handle(), &n, &r, &lanes));
      fprintf(stderr, "[rank %d] RCCL communicator: %d ranks, this is rank %d, %d lanes\n", rank, n, r, lanes);
      ar = mw_dycore_rccl_allreduce_sum; ar_ctx = dycore.handle();
    }
    modules::ColumnNudger column_nudger;
    column_nudger.set_column(coupler, ar, ar_ctx);
    modules::perturb_temperature(coupler);
    real etime = 0;
    for (int s = 0; s < nsteps; s++) {
      real dtphys = dycore.compute_time_step(coupler);
      dycore.time_step(coupler, dtphys);
      micro.time_step(coupler, dtphys);
      modules::sponge_layer(coupler, dtphys, 60, ar, ar_ctx);
      column_nudger.nudge_to_column(coupler, dtphys, ar, ar_ctx);
      etime += dtphys;
    }
    (void)hipDeviceSynchronize();
    auto &dm = coupler.get_data_manager_readwrite();
    size_t n = (size_t)nz * coupler.get_ny() * coupler.get_nx() * nens;
    std::vector<double> w(n), r(n);
    (void)hipMemcpy(w.data(), dm.get<real>("wvel").data(), n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(r.data(), dm.get<real>("density_dry").data(), n * 8, hipMemcpyDeviceToHost);
    double maxw = 0, sum = 0;
    for (size_t i = 0; i < n; i++) { maxw = std::max(maxw, std::fabs(w[i])); sum += r[i]; }
    double total = sum;
    if (ar) {                                                   // the all-reduce itself, on a one-element device buffer
      double *d = nullptr;
      if (hipMalloc((void **)&d, 8) != hipSuccess) endrun("allocation failed");
      (void)hipMemcpy(d, &sum, 8, hipMemcpyHostToDevice);
      mw_check(ar(ar_ctx, d, 1, nullptr));
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(&total, d, 8, hipMemcpyDeviceToHost);
      (void)hipFree(d);
    }
    printf("rank %d of %d block %dx%d at (%lld,%lld) etime %.17g maxw %.17e sum_density_dry %.17e total_density_dry %.17e validate_all %lld\n", rank, world,
           coupler.get_nx(), coupler.get_ny(), (long long)coupler.get_i_beg(), (long long)coupler.get_j_beg(), etime, maxw, sum, total, dm.validate_all(false));
  } catch (std::exception &e) { fprintf(stderr, "[rank %d] endrun: %s\n", rank, e.what()); return 1; }
  return 0;
}
