// =====================================================================================================
// mw_facade.h -- C++ host-side mirror of miniWeatherML's module API on top of the C ABI (include/mw_cdna4.h).
//
// The reference's "plugin API" is a header-level C++ convention: a driver owns a core::Coupler,
// calls module.init(coupler) once and module.time_step(coupler, dt) every step.  This header provides the same
// names with the same argument meaning so that the reference drivers' call sequence
// (experiments/supercell_example/driver.cpp:41-79) compiles line-for-line against the MI355X-native kernels:
//
//   core::Coupler            model/core/coupler.h:17-493        options, tracer registry, grid, decomposition
//   core::DataManager        model/core/DataManager.h:20-585    name -> device allocation (+dims, dirty, positive)
//   modules::Dynamics_Euler_Stratified_WenoFV                    model/modules/dynamics_euler_stratified_wenofv.h
//   modules::Microphysics_Kessler                                model/modules/microphysics_kessler.h
//   modules::perturb_temperature                                 model/modules/perturb_temperature.h
//
// No YAKL: arrays are raw device allocations (hipMalloc) handed to the C ABI as plain pointers; `DeviceView<T>` is a
// non-owning (pointer, dims) pair like the non-owning yakl::Array the reference's DataManager::get returns (:262,:283).
// Errors: endrun(msg) throws std::runtime_error (reference: yakl_throw, main_header.h:66-68).
// =====================================================================================================
#pragma once
#include "../../include/mw_cdna4.h"
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <variant>
#include <vector>

typedef double real;                                                  // main_header.h:59

inline void endrun(const std::string &msg) { throw std::runtime_error(msg); }
inline void mw_check(int rc) { if (rc) endrun(mw_last_error()); }

int constexpr max_fields = 50;                                        // main_header.h:57

#if defined(__HIPCC__)
#define MW_HD __host__ __device__
#else
#define MW_HD
#endif

// A rank-N view that a HIP kernel can take BY VALUE (plain pointer + extents, row-major, last index fastest) -- what the reference's
// non-owning yakl::Array<T,N,memDevice,styleC> is to its YAKL_LAMBDAs.  Indexable on the device (and on the host for host memory).
template <class T, int N> struct FieldView {
  T *ptr = nullptr;
  int dim[N] = {};
  MW_HD T *data() const { return ptr; }
  MW_HD int extent(int i) const { return dim[i]; }
  MW_HD size_t size() const { size_t n = 1; for (int i = 0; i < N; i++) n *= (size_t)dim[i]; return n; }
  template <class... I> MW_HD T &operator()(I... idx) const {
    static_assert(sizeof...(I) == N, "FieldView: the number of indices must equal the rank");
    const long long ix[N] = {(long long)idx...};
    long long off = 0;
    for (int i = 0; i < N; i++) off = off * dim[i] + ix[i];
    return ptr[off];
  }
  operator FieldView<const T, N>() const { FieldView<const T, N> v; v.ptr = ptr; for (int i = 0; i < N; i++) v.dim[i] = dim[i]; return v; }
};

template <class T> struct DeviceView {                                // non-owning view
  T *ptr = nullptr;
  std::vector<int> dimension;
  T *data() const { return ptr; }
  size_t size() const { size_t n = 1; for (int d : dimension) n *= (size_t)d; return n; }
  int extent(int i) const { return dimension[i]; }
  template <int N> FieldView<T, N> as() const {                       // the same allocation with its rank in the type (DataManager::get<T,N>)
    if ((int)dimension.size() != N) throw std::runtime_error("ERROR: Requested dimensions is different from the entry dimensions");   // DataManager.h:263-268
    FieldView<T, N> v; v.ptr = ptr; for (int i = 0; i < N; i++) v.dim[i] = dimension[i]; return v;
  }
  template <int N> operator FieldView<T, N>() const { return as<N>(); }
};

namespace core {

// core::MultipleFields / core::MultiField (model/core/MultipleFields.h:10-96): several fields of one rank aggregated so that ONE kernel
// can loop over them ("used mostly for tracers"; sponge_layer.h:32-41, column_nudging.h:28-33, :50-55 build them from DataManager::get).
// Trivially copyable -- a HIP kernel takes it by value like the reference's YAKL_LAMBDAs capture it -- with the reference's members:
// add_field, get_field, get_num_fields, operator()(field, indices...).
template <int MAX_FIELDS, class T>
class MultipleFields {
 public:
  mutable T fields[MAX_FIELDS];                                       // (SArray<T,1,MAX_FIELDS>; get_field is const and returns T&, :53-55)
  int num_fields;
  MW_HD MultipleFields() : num_fields(0) {}
  MW_HD MultipleFields(MultipleFields const &rhs) : num_fields(rhs.num_fields) { for (int i = 0; i < num_fields; i++) fields[i] = rhs.fields[i]; }   // :17-22
  MW_HD MultipleFields &operator=(MultipleFields const &rhs) { num_fields = rhs.num_fields; for (int i = 0; i < num_fields; i++) fields[i] = rhs.fields[i]; return *this; }
  MW_HD void add_field(T field) {                                     // :48-51 (the reference writes past the end silently; here that is an error on the host)
#if !defined(__HIP_DEVICE_COMPILE__)
    if (num_fields >= MAX_FIELDS) throw std::runtime_error("ERROR: MultipleFields: more than MAX_FIELDS fields");
#endif
    fields[num_fields] = field; num_fields++;
  }
  MW_HD T &get_field(int tr) const { return fields[tr]; }             // :53-55
  MW_HD int get_num_fields() const { return num_fields; }             // :57
  template <class... I> MW_HD auto operator()(int tr, I... idx) const -> decltype(fields[tr](idx...)) { return fields[tr](idx...); }   // :59-90
};
template <class T, int N> using MultiField = MultipleFields<max_fields, FieldView<T, N>>;                          // :95-96

class Options {                                                       // model/core/Options.h:11-167
  std::map<std::string, std::variant<int, real, bool, std::string, long long>> opts;
 public:
  template <class T> void add_option(const std::string &key, T v) { if (!opts.count(key)) opts[key] = v; }     // :60-74
  template <class T> void set_option(const std::string &key, T v) { opts[key] = v; }                           // :77-89
  template <class T> T get_option(const std::string &key) const {                                              // :92-108
    auto it = opts.find(key);
    if (it == opts.end()) endrun("ERROR: option not found: " + key);
    if (!std::holds_alternative<T>(it->second)) endrun("ERROR: Requesting option using the wrong type: " + key);
    return std::get<T>(it->second);
  }
  bool option_exists(const std::string &key) const { return opts.count(key) > 0; }
  void delete_option(const std::string &key) { opts.erase(key); }
};

class DataManager {                                                   // model/core/DataManager.h
  struct Entry { std::string name, desc; size_t type_hash; void *ptr; size_t bytes; std::vector<int> dims;
                 std::vector<std::string> dim_names; bool positive, dirty; };
  std::vector<Entry> entries;
  std::map<std::string, int> dimensions;
  // run in front of every access to an entry's data (get, the validators, clone_into): a module that keeps part of a field's value parked
  // elsewhere -- ColumnNudger's deferred increments -- registers its flush here, so that whoever LOOKS at a field sees all of it
  std::vector<std::function<void()>> before_access;
  void sync_entries() const { for (auto &fn : before_access) fn(); }
  int find_entry(const std::string &n) const { for (size_t i = 0; i < entries.size(); i++) if (entries[i].name == n) return (int)i; return -1; }
 public:
  DataManager() = default;
  DataManager(const DataManager &) = delete;
  ~DataManager() { finalize(); }
  void finalize() { for (auto &e : entries) if (e.ptr) (void)hipFree(e.ptr); entries.clear(); dimensions.clear(); }   // :571-578
  void add_access_hook(std::function<void()> fn) { before_access.push_back(std::move(fn)); }
  void add_dimension(const std::string &name, int len) {                                                       // :106-120
    auto it = dimensions.find(name);
    if (it != dimensions.end() && it->second != len) endrun("ERROR: Attempting to add a dimension of the same name as an existing dimension but not the same size");
    dimensions[name] = len;
  }
  int find_dimension(const std::string &n) const { return dimensions.count(n) ? 0 : -1; }
  int get_dimension_size(const std::string &n) const { auto it = dimensions.find(n); if (it == dimensions.end()) endrun("ERROR: Could not find dimension."); return it->second; }
  template <class T> void register_and_allocate(const std::string &name, const std::string &desc, std::vector<int> dims,
                                                std::vector<std::string> dim_names = {}, bool positive = false) {   // :122-181
    if (name.empty()) endrun("ERROR: You cannot register_and_allocate with an empty string");
    if (find_entry(name) != -1) endrun("ERROR: Duplicate entry name: " + name);
    size_t n = 1; for (int d : dims) n *= (size_t)d;
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) endrun("ERROR: device allocation failed for " + name);
    (void)hipMemset(p, 0, n * sizeof(T));
    entries.push_back({name, desc, typeid(T).hash_code(), p, n * sizeof(T), dims, dim_names, positive, false});
  }
  bool entry_exists(const std::string &n) const { return find_entry(n) != -1; }
  template <class T> DeviceView<T> get(const std::string &name) {                                              // :246-286
    int id = find_entry(name);
    if (id == -1) endrun("ERROR: Could not find entry name: " + name);
    sync_entries();
    typedef typename std::remove_cv<T>::type TNC;
    if (entries[id].type_hash != typeid(TNC).hash_code()) endrun("ERROR: Requested Array type does not match entry type");   // :256-261
    if (!std::is_const<T>::value) entries[id].dirty = true;                                                     // :275
    return DeviceView<T>{(T *)entries[id].ptr, entries[id].dims};
  }
  template <class T, int N> FieldView<T, N> get(const std::string &name) { return get<T>(name).template as<N>(); }   // the reference's spelling get<T,N>(name): rank checked (:263-268)
  // :79-103: an independent copy of every entry (own allocation, device-to-device copy), dimensions included
  void clone_into(DataManager &dm) const {
    sync_entries();
    dm.dimensions = dimensions;
    for (auto &e : entries) {
      Entry loc = e;
      loc.ptr = nullptr;
      if (hipMalloc(&loc.ptr, e.bytes) != hipSuccess) endrun("ERROR: device allocation failed for " + e.name);
      if (hipMemcpy(loc.ptr, e.ptr, e.bytes, hipMemcpyDeviceToDevice) != hipSuccess) endrun("ERROR: device copy failed for " + e.name);
      dm.entries.push_back(loc);
    }
  }
  template <class T> DeviceView<T> get_lev_col(const std::string &name) {                                      // :289-330
    auto v = get<T>(name); int nlev = v.dimension[0]; int ncol = (int)(v.size() / (size_t)nlev);
    return DeviceView<T>{v.ptr, {nlev, ncol}};
  }
  template <class T> DeviceView<T> get_collapsed(const std::string &name) { auto v = get<T>(name); return DeviceView<T>{v.ptr, {(int)v.size()}}; }   // :333-365
  bool get_dirty(const std::string &n) const { int id = find_entry(n); return id >= 0 && entries[id].dirty; }
  void clean_all() { for (auto &e : entries) e.dirty = false; }
  // ---- the rest of the reference's public surface -------------------------------------------------------------------------------
  int find_entry_or_error(const std::string &n) const {                                                         // :505-511
    int id = find_entry(n); if (id < 0) endrun("ERROR: Attempting to retrieve variable name [" + n + "], but it doesn't exist. "); return id; }
  void unregister_and_deallocate(const std::string &n) {                                                        // :199-203
    int id = find_entry_or_error(n); if (entries[id].ptr) (void)hipFree(entries[id].ptr); entries.erase(entries.begin() + id); }
  void clean_all_entries() { clean_all(); }                                                                     // :208-210
  void clean_entry(const std::string &n) { entries[find_entry_or_error(n)].dirty = false; }                     // :215-218
  bool entry_is_dirty(const std::string &n) const { return entries[find_entry_or_error(n)].dirty; }             // :223-226
  std::vector<std::string> get_dirty_entries() const { std::vector<std::string> r; for (auto &e : entries) if (e.dirty) r.push_back(e.name); return r; }   // :231-237
  // validators (:385-483).  The reference copies each array to the host ("This is EXPENSIVE"); here one device pass per entry
  // (mw_validate_f64 / _f32).  Same warnings on std::cerr (the first offending flat index), endrun when die_on_failed_check.
  // Entries of other types (bool, integers) hold no NaN / inf; their sign check is not needed by any shipped module.
 private:
  bool scan(int id, long long *r) const {
    sync_entries();
    const Entry &e = entries[id];
    if (e.type_hash == typeid(double).hash_code()) { mw_check(mw_validate_f64((const double *)e.ptr, (long long)(e.bytes / sizeof(double)), r, nullptr)); return true; }
    if (e.type_hash == typeid(float).hash_code())  { mw_check(mw_validate_f32((const float *)e.ptr, (long long)(e.bytes / sizeof(float)), r, nullptr)); return true; }
    return false;
  }
  static long long report(const char *what, const std::string &n, long long count, long long first, bool die) {
    if (count) { fprintf(stderr, "WARNING: %s discovered in: %s at global index: %lld\n", what, n.c_str(), first); if (die) endrun(""); }
    return count;
  }
 public:
  long long validate_nan(const std::string &n, bool die_on_failed_check = false) const {                       // :402-418
    long long r[6]; int id = find_entry_or_error(n); return scan(id, r) ? report("NaN", n, r[0], r[3], die_on_failed_check) : 0; }
  long long validate_inf(const std::string &n, bool die_on_failed_check = false) const {                       // :421-428
    long long r[6]; int id = find_entry_or_error(n); return scan(id, r) ? report("inf", n, r[1], r[4], die_on_failed_check) : 0; }
  long long validate_pos(const std::string &n, bool die_on_failed_check = false) const {                       // :431-441, :471-483
    long long r[6]; int id = find_entry_or_error(n); if (!entries[id].positive || !scan(id, r)) return 0;
    return report("negative value discovered in positive-definite entry", n, r[2], r[5], die_on_failed_check); }
  long long validate(const std::string &n, bool die_on_failed_check = false) const {                           // :393-397 (one scan for the three checks)
    long long r[6]; int id = find_entry_or_error(n); if (!scan(id, r)) return 0;
    long long bad = report("NaN", n, r[0], r[3], die_on_failed_check) + report("inf", n, r[1], r[4], die_on_failed_check);
    if (entries[id].positive) bad += report("negative value discovered in positive-definite entry", n, r[2], r[5], die_on_failed_check);
    return bad; }
  long long validate_all(bool die_on_failed_check = false) const { long long bad = 0; for (auto &e : entries) bad += validate(e.name, die_on_failed_check); return bad; }   // :385-387
};

class Coupler {                                                       // model/core/coupler.h:17-493
  Options options;
  real xlen = -1, ylen = -1, zlen = -1, dt_gcm = -1;
  int nranks = 1, myrank = 0;
  struct Tracer { std::string name, desc; bool positive, adds_mass; };
  std::vector<Tracer> tracers;
  DataManager dm;
 public:
  mw_grid_t grid;                                                     // what the C ABI needs from the getters below
  Coupler() { memset(&grid, 0, sizeof(grid)); }
  Coupler(const Coupler &) = delete;                                  // move-only in the reference too (:67-70)
  // coupler.h:110-214.  Pass the rank layout explicitly (the reference reads it from MPI_COMM_WORLD).
  void distribute_mpi_and_allocate_coupled_state(int nz, size_t ny_glob, size_t nx_glob, int nens, int nranks_in = 1, int myrank_in = 0) {
    nranks = nranks_in; myrank = myrank_in;
    mw_check(mw_decompose(nranks, myrank, (long long)nx_glob, (long long)ny_glob, &grid));
    grid.nz = nz; grid.nens = nens;
    dm.add_dimension("nens", nens); dm.add_dimension("x", grid.nx); dm.add_dimension("y", grid.ny); dm.add_dimension("z", nz);
  }
  // :85-106: grid, decomposition, tracer registry and an independent copy of every DataManager entry go to `coupler`; the OPTIONS do
  // not (the reference's clone_into leaves coupler.options alone)
  void clone_into(Coupler &coupler) const {
    coupler.xlen = xlen; coupler.ylen = ylen; coupler.zlen = zlen; coupler.dt_gcm = dt_gcm;
    coupler.tracers = tracers;
    coupler.nranks = nranks; coupler.myrank = myrank;
    coupler.grid = grid;                                              // nens, nx_glob, ny_glob, nproc_x/y, px, py, i_beg, j_beg, neigh (:94-105)
    dm.clone_into(coupler.dm);
  }
  void set_grid(real xl, real yl, real zl) { xlen = xl; ylen = yl; zlen = zl; grid.xlen = xl; grid.ylen = yl; grid.zlen = zl; }
  real get_xlen() const { return xlen; }  real get_ylen() const { return ylen; }  real get_zlen() const { return zlen; }
  int get_nranks() const { return nranks; }  int get_myrank() const { return myrank; }  int get_nens() const { return grid.nens; }
  size_t get_nx_glob() const { return (size_t)grid.nx_glob; }  size_t get_ny_glob() const { return (size_t)grid.ny_glob; }
  int get_nproc_x() const { return grid.nproc_x; }  int get_nproc_y() const { return grid.nproc_y; }
  int get_px() const { return grid.px; }  int get_py() const { return grid.py; }
  size_t get_i_beg() const { return (size_t)grid.i_beg; }  size_t get_j_beg() const { return (size_t)grid.j_beg; }
  bool is_sim2d() const { return grid.ny_glob == 1; }  bool is_mainproc() const { return myrank == 0; }
  const int *get_neighbor_rankid_matrix() const { return grid.neigh; }       // [y][x], row-major 3x3
  DataManager const &get_data_manager_readonly() const { return dm; }
  DataManager &get_data_manager_readwrite() { return dm; }
  int get_nx() const { return dm.find_dimension("x") == -1 ? -1 : dm.get_dimension_size("x"); }
  int get_ny() const { return dm.find_dimension("y") == -1 ? -1 : dm.get_dimension_size("y"); }
  int get_nz() const { return dm.find_dimension("z") == -1 ? -1 : dm.get_dimension_size("z"); }
  real get_dx() const { return get_xlen() / grid.nx_glob; }                 // :262
  real get_dy() const { return get_ylen() / grid.ny_glob; }                 // :265
  real get_dz() const { return get_zlen() / get_nz(); }                     // :268
  int get_num_tracers() const { return (int)tracers.size(); }
  template <class T> void add_option(const std::string &k, T v) { options.add_option<T>(k, v); }
  template <class T> void set_option(const std::string &k, T v) { options.set_option<T>(k, v); }
  template <class T> T get_option(const std::string &k) const { return options.get_option<T>(k); }
  template <class T> T get_option(const std::string &k, T dflt) const { return options.option_exists(k) ? options.get_option<T>(k) : dflt; }
  bool option_exists(const std::string &k) const { return options.option_exists(k); }
  void delete_option(const std::string &k) { options.delete_option(k); }
  void add_tracer(const std::string &name, const std::string &desc, bool positive, bool adds_mass) {           // :323-330
    dm.register_and_allocate<real>(name, desc, {get_nz(), get_ny(), get_nx(), get_nens()}, {"z", "y", "x", "nens"}, positive);
    tracers.push_back({name, desc, positive, adds_mass});
  }
  std::vector<std::string> get_tracer_names() const { std::vector<std::string> r; for (auto &t : tracers) r.push_back(t.name); return r; }
  void get_tracer_info(const std::string &name, std::string &desc, bool &found, bool &positive, bool &adds_mass) const {   // :340-353
    for (auto &t : tracers) if (t.name == name) { positive = t.positive; desc = t.desc; adds_mass = t.adds_mass; found = true; return; }
    found = false;
  }
  bool tracer_exists(const std::string &name) const { for (auto &t : tracers) if (t.name == name) return true; return false; }
};

} // namespace core

namespace modules {

class Microphysics_Kessler {                                          // model/modules/microphysics_kessler.h
  void *ws = nullptr; long long ws_bytes = 0;
 public:
  int static constexpr num_tracers = 3;
  void set_strict(int strict) { mw_check(mw_kessler_set_strict(strict)); }   // 1: reference operation order + glibc's pow / exp (bit-identical to the CPU restatement)
  real R_d, cp_d, cv_d, gamma_d, kappa_d, R_v, cp_v, cv_v, p0, grav;
  Microphysics_Kessler() { R_d = 287.; cp_d = 1003.; cv_d = cp_d - R_d; gamma_d = cp_d / cv_d; kappa_d = R_d / cp_d; R_v = 461.;
                           cp_v = 1859; cv_v = R_v - cp_v; p0 = 1.e5; grav = 9.81; }                             // :29-41
  ~Microphysics_Kessler() { if (ws) (void)hipFree(ws); }
  static int get_num_tracers() { return num_tracers; }
  std::string micro_name() const { return "kessler"; }
  void init(core::Coupler &coupler) {                                 // :51-96
    coupler.add_tracer("water_vapor", "Water Vapor", true, true);
    coupler.add_tracer("cloud_liquid", "Cloud liquid", true, true);
    coupler.add_tracer("precip_liquid", "precip_liquid", true, true);
    coupler.get_data_manager_readwrite().register_and_allocate<real>("precl", "precipitation rate",
        {coupler.get_ny(), coupler.get_nx(), coupler.get_nens()}, {"y", "x", "nens"});
    coupler.set_option<std::string>("micro", "kessler");
    coupler.set_option<real>("R_d", R_d); coupler.set_option<real>("cp_d", cp_d); coupler.set_option<real>("cv_d", cv_d);
    coupler.set_option<real>("gamma_d", gamma_d); coupler.set_option<real>("kappa_d", kappa_d); coupler.set_option<real>("R_v", R_v);
    coupler.set_option<real>("cp_v", cp_v); coupler.set_option<real>("cv_v", cv_v); coupler.set_option<real>("p0", p0);
    coupler.set_option<real>("grav", grav);
  }
  void time_step(core::Coupler &coupler, real dt) {                   // :99-162
    auto &dm = coupler.get_data_manager_readwrite();
    auto rho_v = dm.get_lev_col<real>("water_vapor"), rho_c = dm.get_lev_col<real>("cloud_liquid"), rho_r = dm.get_lev_col<real>("precip_liquid");
    auto rho_dry = dm.get_lev_col<real const>("density_dry");
    auto temp = dm.get_lev_col<real>("temp");
    auto precl = dm.get_collapsed<real>("precl");
    int nz = coupler.get_nz(); long long ncol = (long long)coupler.get_ny() * coupler.get_nx() * coupler.get_nens();
    long long need = mw_kessler_workspace_bytes(nz, ncol);
    if (need > ws_bytes) { if (ws) (void)hipFree(ws); if (hipMalloc(&ws, (size_t)need) != hipSuccess) endrun("kessler workspace allocation failed"); ws_bytes = need; }
    mw_check(mw_kessler_time_step(nz, ncol, coupler.get_dz(), dt, rho_v.data(), rho_c.data(), rho_r.data(), rho_dry.data(), temp.data(),
                                  precl.data(), ws, nullptr, nullptr));
  }
};

class Dynamics_Euler_Stratified_WenoFV {                              // model/modules/dynamics_euler_stratified_wenofv.h
  mw_dycore_t h = nullptr;
  std::vector<double *> tracer_ptrs;
  double *f_rho = nullptr, *f_u = nullptr, *f_v = nullptr, *f_w = nullptr, *f_T = nullptr;
 public:
#ifndef MW_ORD
  int static constexpr ord = 5;                                       // :24-28 (compile with -DMW_ORD=3 | 7 | 9 for the reference's other orders)
#else
  int static constexpr ord = MW_ORD;
#endif
  int static constexpr hs = (ord - 1) / 2, num_state = 5;             // :29
  int static constexpr idR = 0, idU = 1, idV = 2, idW = 3, idT = 4;
  real etime = 0, out_freq = -1;  int num_out = 0, idWV = 0;
  std::vector<real> hy_dens_cells, hy_dens_theta_cells, hy_dens_edges, hy_dens_theta_edges;     // (nz[,+1],nens), host copies
  ~Dynamics_Euler_Stratified_WenoFV() { *self_token = nullptr; if (h) mw_dycore_destroy(h); }
  Dynamics_Euler_Stratified_WenoFV() = default;
  Dynamics_Euler_Stratified_WenoFV(const Dynamics_Euler_Stratified_WenoFV &) = delete;       // (owns the handle)
  real compute_time_step(core::Coupler const &coupler) const { return mw_dycore_compute_time_step(&coupler.grid); }     // :70-77
  void init(core::Coupler &coupler) {                                 // :1197-1683
    mw_grid_t &g = coupler.grid;
    mw_grid_t c; memset(&c, 0, sizeof(c)); mw_check(mw_default_constants(&c));
    if (!coupler.option_exists("R_d")) coupler.set_option<real>("R_d", c.R_d);                 // :1227-1249
    if (!coupler.option_exists("cp_d")) coupler.set_option<real>("cp_d", c.cp_d);
    if (!coupler.option_exists("R_v")) coupler.set_option<real>("R_v", c.R_v);
    if (!coupler.option_exists("cp_v")) coupler.set_option<real>("cp_v", c.cp_v);
    if (!coupler.option_exists("p0")) coupler.set_option<real>("p0", c.p0);
    if (!coupler.option_exists("grav")) coupler.set_option<real>("grav", c.grav);
    if (!coupler.option_exists("earthrot")) coupler.set_option<real>("earthrot", c.earthrot);
    real R_d = coupler.get_option<real>("R_d"), cp_d = coupler.get_option<real>("cp_d"), p0 = coupler.get_option<real>("p0");
    if (!coupler.option_exists("cv_d")) coupler.set_option<real>("cv_d", cp_d - R_d);
    if (!coupler.option_exists("gamma_d")) coupler.set_option<real>("gamma_d", cp_d / coupler.get_option<real>("cv_d"));
    if (!coupler.option_exists("kappa_d")) coupler.set_option<real>("kappa_d", R_d / cp_d);
    if (!coupler.option_exists("cv_v")) coupler.set_option<real>("cv_v", coupler.get_option<real>("R_v") - coupler.get_option<real>("cp_v"));
    real gamma = coupler.get_option<real>("gamma_d"), kappa = coupler.get_option<real>("kappa_d");
    if (!coupler.option_exists("C0")) coupler.set_option<real>("C0", pow(R_d * pow(p0, -kappa), gamma));
    coupler.set_option<real>("latitude", 0);
    g.R_d = R_d; g.cp_d = cp_d; g.p0 = p0; g.R_v = coupler.get_option<real>("R_v"); g.cp_v = coupler.get_option<real>("cp_v");
    g.grav = coupler.get_option<real>("grav"); g.gamma_d = gamma; g.kappa_d = kappa; g.C0 = coupler.get_option<real>("C0");
    g.earthrot = coupler.get_option<real>("earthrot"); g.latitude = 0;
    auto &dm = coupler.get_data_manager_readwrite();
    int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
    for (const char *n : {"density_dry", "uvel", "vvel", "wvel", "temp"}) dm.register_and_allocate<real>(n, "", {nz, ny, nx, nens});   // :1253-1257
    auto names = coupler.get_tracer_names();
    int T = (int)names.size();
    std::vector<unsigned char> pos(T), adds(T);
    bool haveWV = false;
    for (int tr = 0; tr < T; tr++) { std::string d; bool f, p, a; coupler.get_tracer_info(names[tr], d, f, p, a); pos[tr] = p; adds[tr] = a;
                                     if (names[tr] == "water_vapor") { idWV = tr; haveWV = true; } }                 // :1285-1293
    if (!haveWV) endrun("ERROR: a tracer named water_vapor must be registered before dycore.init");
    g.num_tracers = T; g.idWV = idWV;
    coupler.set_option<int>("idWV", idWV);                                                                      // :1300
    auto init_data = coupler.get_option<std::string>("init_data");
    out_freq = coupler.get_option<real>("out_freq", -1.);
    int init_id = init_data == "thermal" ? MW_DATA_THERMAL : init_data == "supercell" ? MW_DATA_SUPERCELL : init_data == "city" ? MW_DATA_CITY
                : init_data == "building" ? MW_DATA_BUILDING : -1;
    if (init_id < 0) endrun("ERROR: Invalid init_data in yaml input file");                                     // :1310
    g.enable_gravity = coupler.get_option<bool>("enable_gravity", true);
    g.bc_x = g.bc_y = MW_BC_PERIODIC; g.bc_z = MW_BC_WALL; g.use_immersed = 0;
    mw_check(mw_dycore_create(&h, &g, pos.data(), adds.data(), nullptr));
    if (ord != 5) mw_check(mw_dycore_set_order(h, ord));                                                        // before init (:1725-1727)
    bind(coupler);
    mw_check(mw_dycore_init(h, init_id, f_rho, f_u, f_v, f_w, f_T, tracer_ptrs.data()));
    mw_check(mw_dycore_get_grid(h, &g));
    coupler.set_option<bool>("use_immersed_boundaries", g.use_immersed != 0);                                   // :1312,1426,1554
    coupler.add_option<int>("bc_x", g.bc_x); coupler.add_option<int>("bc_y", g.bc_y); coupler.add_option<int>("bc_z", g.bc_z);
    hy_dens_cells.resize((size_t)nz * nens); hy_dens_theta_cells.resize((size_t)nz * nens);
    hy_dens_edges.resize((size_t)(nz + 1) * nens); hy_dens_theta_edges.resize((size_t)(nz + 1) * nens);
    mw_check(mw_dycore_get_background(h, hy_dens_cells.data(), hy_dens_theta_cells.data(), hy_dens_edges.data(), hy_dens_theta_edges.data()));
    dm.register_and_allocate<real>("hy_dens_cells", "hydrostatic density cell averages", {nz, nens});            // :1663-1668
    dm.register_and_allocate<real>("hy_dens_theta_cells", "hydrostatic density*theta cell averages", {nz, nens});
    (void)hipMemcpy(dm.get<real>("hy_dens_cells").data(), hy_dens_cells.data(), hy_dens_cells.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dm.get<real>("hy_dens_theta_cells").data(), hy_dens_theta_cells.data(), hy_dens_theta_cells.size() * 8, hipMemcpyHostToDevice);
    etime = 0; num_out = 0;
    if (out_freq >= 0.) output(coupler, etime);                                                                  // :1659
  }
  void time_step(core::Coupler &coupler, real &dt_phys) {             // :81-198
    if (!h) endrun("dycore.time_step before init");
    mw_check(mw_dycore_time_step(h, f_rho, f_u, f_v, f_w, f_T, tracer_ptrs.data(), dt_phys));
    etime += dt_phys;
    if (out_freq >= 0. && etime / out_freq >= num_out + 1) { output(coupler, etime); num_out++; }                  // :183-186
  }
  // output(coupler, etime), :2019-2191, shared-file branch: CDF-5 `<out_prefix>.nc`, dims x,y,z,t, one (t,z,y,x) double
  // variable per coupler field, ensemble member 0.  Single-process form (the ranks of a multi-process run order themselves
  // with their own barrier around create / set_numrecs, as miniweatherml_amd/modules.py does).
  void output(core::Coupler const &coupler, real etime_) const {
    // file_per_process (:2038-2090): `<out_prefix>_<rank, 8 digits>.nc` with the rank's LOCAL sizes and hyperslab offsets 0
    // (classic CDF-5 here; the reference's SimpleNetCDF writes a NetCDF-4 container with the same dims / variables / values)
    const bool fpp = coupler.get_option<bool>("file_per_process", false);
    mw_grid_t g = coupler.grid;
    const long long i_beg = g.i_beg, j_beg = g.j_beg;
    char rk[16]; snprintf(rk, sizeof(rk), "_%08d", coupler.get_myrank());
    std::string path = coupler.get_option<std::string>("out_prefix") + (fpp ? std::string(rk) : std::string()) + ".nc";
    if (fpp) { g.i_beg = 0; g.j_beg = 0; }
    std::vector<std::string> names = {"density_dry", "uvel", "vvel", "wvel", "temp"};
    for (auto &n : coupler.get_tracer_names()) names.push_back(n);
    mw_nc_t nc = nullptr; long long rec = 0; int v = 0;
    auto put1 = [&](const char *name, long long start, std::vector<double> const &a) {
      long long st[1] = {start}, ct[1] = {(long long)a.size()};
      mw_check(mw_nc_inq_varid(nc, name, &v)); mw_check(mw_nc_put_vara_double(nc, v, st, ct, a.data())); };
    if (etime_ == 0) {
      mw_check(mw_nc_create(&nc, path.c_str(), 5, fpp ? 0 : 1048576, fpp ? 0 : 1048576));                        // :2103-2106
      int dx_, dy_, dz_, dt_;
      mw_check(mw_nc_def_dim(nc, "x", fpp ? (long long)g.nx : (long long)coupler.get_nx_glob(), &dx_));
      mw_check(mw_nc_def_dim(nc, "y", fpp ? (long long)g.ny : (long long)coupler.get_ny_glob(), &dy_));
      mw_check(mw_nc_def_dim(nc, "z", g.nz, &dz_)); mw_check(mw_nc_def_dim(nc, "t", 0, &dt_));
      mw_check(mw_nc_def_var(nc, "x", 1, &dx_, &v)); mw_check(mw_nc_def_var(nc, "y", 1, &dy_, &v));
      mw_check(mw_nc_def_var(nc, "z", 1, &dz_, &v)); mw_check(mw_nc_def_var(nc, "t", 1, &dt_, &v));
      int d4[4] = {dt_, dz_, dy_, dx_};
      for (auto &n : names) mw_check(mw_nc_def_var(nc, n.c_str(), 4, d4, &v));
      mw_check(mw_nc_enddef(nc));
      std::vector<double> xs(g.nx), ys(g.ny), zs(g.nz);
      for (int i = 0; i < g.nx; i++) xs[i] = (i + i_beg + 0.5) * coupler.get_dx();
      for (int j = 0; j < g.ny; j++) ys[j] = (j + j_beg + 0.5) * coupler.get_dy();
      for (int k = 0; k < g.nz; k++) zs[k] = (k + 0.5) * coupler.get_dz();
      put1("x", g.i_beg, xs); put1("y", g.j_beg, ys); put1("z", 0, zs); put1("t", 0, {0.0});
    } else {
      mw_check(mw_nc_open(&nc, path.c_str()));
      mw_check(mw_nc_inq_dimlen(nc, "t", &rec));
      put1("t", rec, {(double)etime_});
    }
    auto &dm = const_cast<core::DataManager &>(coupler.get_data_manager_readonly());
    for (auto &n : names) { mw_check(mw_nc_inq_varid(nc, n.c_str(), &v)); mw_check(mw_output_put_field(nc, v, rec, &g, dm.get<real>(n).data(), nullptr)); }
    mw_check(mw_nc_set_numrecs(nc, rec + 1));
    mw_check(mw_nc_close(nc));
  }
  mw_dycore_t handle() const { return h; }
  // column increments that ColumnNudger::nudge_to_column(..., &dycore) parked in the handle: applied now (a no-op when there are none)
  void flush_pending() { if (h) mw_check(mw_dycore_flush_pending(h)); }
  bool flush_hook_installed = false;
  std::shared_ptr<Dynamics_Euler_Stratified_WenoFV *> self_token = std::make_shared<Dynamics_Euler_Stratified_WenoFV *>(this);   // (cleared by the destructor: a hook that outlives the module does nothing)
  // Decomposed runs (coupler.distribute_mpi_and_allocate_coupled_state(..., nranks, myrank)): the reference exchanges halos with
  // MPI_Isend / Irecv (:641-723); here every rank joins one RCCL communicator -- id128: the 128 bytes of mw_rccl_unique_id() made by
  // rank 0 and handed to the other ranks by whatever launched them (a file, a pipe, an environment variable).  Afterwards
  // mw_dycore_rccl_allreduce_sum (ctx = handle()) is the all-reduce of sponge_layer / ColumnNudger.
  void use_rccl(core::Coupler const &coupler, const unsigned char *id128) { mw_check(mw_dycore_use_rccl(h, id128, coupler.get_nranks(), coupler.get_myrank())); }
  // Run-time options of the handle (schedule, kernel forms, chunk sizes, transport lanes: include/mw_cdna4.h) -- the typed replacement of
  // the MW_* environment switches; call after init().
  void set_option(const char *key, long long value) { mw_check(mw_dycore_set_option(h, key, value)); }
  long long get_option(const char *key) const { long long v = 0; mw_check(mw_dycore_get_option(h, key, &v)); return v; }
 private:
  void bind(core::Coupler &coupler) {
    auto &dm = coupler.get_data_manager_readwrite();
    f_rho = dm.get<real>("density_dry").data(); f_u = dm.get<real>("uvel").data(); f_v = dm.get<real>("vvel").data();
    f_w = dm.get<real>("wvel").data(); f_T = dm.get<real>("temp").data();
    tracer_ptrs.clear();
    for (auto &n : coupler.get_tracer_names()) tracer_ptrs.push_back(dm.get<real>(n).data());
  }
};

// modules::sponge_layer(coupler, dt, time_scale = 60)   model/modules/sponge_layer.h:8-77  (single rank: no all-reduce)
inline void sponge_layer(core::Coupler &coupler, real dt, real time_scale = 60, mw_allreduce_fn allreduce = nullptr, void *ctx = nullptr) {
  auto &dm = coupler.get_data_manager_readwrite();
  std::vector<double *> f;
  for (const char *n : {"density_dry", "uvel", "vvel", "wvel", "temp"}) f.push_back(dm.get<real>(n).data());
  for (auto &n : coupler.get_tracer_names()) f.push_back(dm.get<real>(n).data());
  static thread_local void *ws = nullptr; static thread_local long long ws_bytes = 0;
  long long need = mw_column_workspace_bytes(&coupler.grid, (int)f.size());
  if (need > ws_bytes) { if (ws) (void)hipFree(ws); if (hipMalloc(&ws, (size_t)need) != hipSuccess) endrun("sponge workspace allocation failed"); ws_bytes = need; }
  mw_check(mw_sponge_layer(&coupler.grid, f.data(), (int)f.size(), dt, time_scale, ws, allreduce, ctx, nullptr));
}

class ColumnNudger {                                                  // model/modules/column_nudging.h:9-108
  double *column = nullptr; void *ws = nullptr; long long ws_bytes = 0;
  static std::vector<double *> state(core::Coupler &c) {
    auto &dm = c.get_data_manager_readwrite(); std::vector<double *> s;
    for (const char *n : {"density_dry", "uvel", "vvel", "temp", "water_vapor"}) s.push_back(dm.get<real>(n).data());
    return s;
  }
  void ensure(core::Coupler &c) {
    long long need = mw_column_workspace_bytes(&c.grid, 5);
    if (need > ws_bytes) { if (ws) (void)hipFree(ws); if (hipMalloc(&ws, (size_t)need) != hipSuccess) endrun("nudger workspace allocation failed"); ws_bytes = need; }
    if (!column && hipMalloc((void **)&column, sizeof(double) * 5 * (size_t)c.get_nz() * c.get_nens()) != hipSuccess) endrun("nudger column allocation failed");
  }
 public:
  int static constexpr num_fields = 5;
  ~ColumnNudger() { if (column) (void)hipFree(column); if (ws) (void)hipFree(ws); }
  void set_column(core::Coupler &coupler, mw_allreduce_fn ar = nullptr, void *ctx = nullptr) {                 // :15-36
    ensure(coupler); auto s = state(coupler);
    mw_check(mw_column_average(&coupler.grid, s.data(), column, ws, ar, ctx, nullptr));
  }
  // defer_to = the dycore module (round 6, no reference counterpart): the second pass -- state += dt (column - average) / 900 -- is not run; the
  // increments are parked in the dycore handle and its next time_step adds them while it converts the coupler's fields (bit for bit the same
  // result).  Whoever reads a field through the DataManager in between triggers the pass after all (DataManager::add_access_hook).
  void nudge_to_column(core::Coupler &coupler, real dt, mw_allreduce_fn ar = nullptr, void *ctx = nullptr,
                       Dynamics_Euler_Stratified_WenoFV *defer_to = nullptr) {                                 // :39-66
    if (!column) endrun("ColumnNudger::nudge_to_column before set_column");
    auto s = state(coupler);
    if (defer_to) {
      if (!defer_to->flush_hook_installed) {
        auto tok = defer_to->self_token;
        coupler.get_data_manager_readwrite().add_access_hook([tok]() { if (*tok) (*tok)->flush_pending(); });
        defer_to->flush_hook_installed = true;
      }
      mw_check(mw_nudge_to_column_deferred(defer_to->handle(), s.data(), column, dt, ws, ar, ctx));
    } else mw_check(mw_nudge_to_column(&coupler.grid, s.data(), column, dt, ws, ar, ctx, nullptr));
  }
};

inline void perturb_temperature(core::Coupler &coupler, bool thermal = true, bool random = false) {   // perturb_temperature.h:8-67
  if (random) mw_check(mw_perturb_temperature_random(&coupler.grid, coupler.get_data_manager_readwrite().get<real>("temp").data(), nullptr));
  if (thermal) mw_check(mw_perturb_temperature(&coupler.grid, coupler.get_data_manager_readwrite().get<real>("temp").data(), nullptr));
}

} // namespace modules

namespace custom_modules {                                             // experiments/simple_city/custom_modules/

inline std::vector<double *> six_fields(core::Coupler &c, const char *prefix = "") {
  auto &dm = c.get_data_manager_readwrite(); std::vector<double *> s;
  for (const char *n : {"density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"}) s.push_back(dm.get<real>(std::string(prefix) + n).data());
  return s;
}

struct Horizontal_Sponge {                                             // horizontal_sponge.h:7-194
  double *column = nullptr; int nz = 0, nens = 0; int sponge_cells = 10; real time_scale = 1;
  ~Horizontal_Sponge() { if (column) (void)hipFree(column); }
  // rccl_handle: on a decomposed run, the dycore handle whose RCCL transport carries the MPI_Bcast from the main rank (:72-77:
  // every rank relaxes towards the MAIN rank's column at (j, i) = (0, 0)); nullptr on one rank
  void init(core::Coupler &coupler, int sponge_cells_ = 10, real time_scale_ = 1, mw_dycore_t rccl_handle = nullptr) {   // :18-91
    nz = coupler.get_nz(); nens = coupler.get_nens();
    if (!column && hipMalloc((void **)&column, sizeof(double) * 6 * (size_t)nz * nens) != hipSuccess) endrun("Horizontal_Sponge: allocation failed");
    auto f = six_fields(coupler);
    mw_check(mw_horizontal_sponge_column(&coupler.grid, f.data(), column, nullptr));
    if (coupler.get_nranks() > 1) {
      if (!rccl_handle) endrun("Horizontal_Sponge::init on a decomposed run needs the dycore handle whose transport carries the broadcast");
      mw_check(mw_dycore_rccl_bcast(rccl_handle, column, 6ll * nz * nens, 0, nullptr));
      (void)hipStreamSynchronize(nullptr);
    }
    sponge_cells = sponge_cells_; time_scale = time_scale_;
  }
  void override_field(int l, real val) { std::vector<double> h((size_t)nz * nens, val);
    (void)hipMemcpy(column + (size_t)l * nz * nens, h.data(), h.size() * 8, hipMemcpyHostToDevice); }
  void override_rho_d(real v) { override_field(0, v); }  void override_uvel(real v) { override_field(1, v); }      // :94-99
  void override_vvel(real v) { override_field(2, v); }   void override_wvel(real v) { override_field(3, v); }
  void override_temp(real v) { override_field(4, v); }   void override_rho_v(real v) { override_field(5, v); }
  void apply(core::Coupler &coupler, real dt, bool x1 = true, bool x2 = true, bool y1 = true, bool y2 = true) {  // :101-192
    if (!column) endrun("Horizontal_Sponge::apply before init");
    auto f = six_fields(coupler);
    mw_check(mw_horizontal_sponge_apply(&coupler.grid, f.data(), column, sponge_cells, time_scale, dt, x1, x2, y1, y2, nullptr));
  }
};

struct Time_Averager {                                                 // time_averager.h:7-143
  real etime = 0;
  void init(core::Coupler &coupler) {                                  // :10-35
    auto &dm = coupler.get_data_manager_readwrite();
    int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
    for (const char *n : {"density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"}) {
      dm.register_and_allocate<real>(std::string("time_avg_") + n, "", {nz, ny, nx, nens});
      (void)hipMemset(dm.get<real>(std::string("time_avg_") + n).data(), 0, sizeof(double) * (size_t)nz * ny * nx * nens);
    }
    etime = 0;
  }
  void accumulate(core::Coupler &coupler, real dt) {                   // :37-78
    auto f = six_fields(coupler), a = six_fields(coupler, "time_avg_");
    mw_check(mw_time_average_accumulate(&coupler.grid, f.data(), a.data(), etime, dt, nullptr));
    etime += dt;
  }
  void finalize(core::Coupler &coupler, const std::string &path = "time_averaged_fields.nc") {                  // :80-141
    const mw_grid_t &g = coupler.grid;
    mw_nc_t nc = nullptr; int v = 0, dx_, dy_, dz_;
    mw_check(mw_nc_create(&nc, path.c_str(), 5, 0, 0));                                                          // NC_CLOBBER | NC_64BIT_DATA
    mw_check(mw_nc_def_dim(nc, "x", (long long)coupler.get_nx_glob(), &dx_)); mw_check(mw_nc_def_dim(nc, "y", (long long)coupler.get_ny_glob(), &dy_));
    mw_check(mw_nc_def_dim(nc, "z", g.nz, &dz_));
    mw_check(mw_nc_def_var(nc, "x", 1, &dx_, &v)); mw_check(mw_nc_def_var(nc, "y", 1, &dy_, &v)); mw_check(mw_nc_def_var(nc, "z", 1, &dz_, &v));
    int d3[3] = {dz_, dy_, dx_};
    const char *names[6] = {"density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"};
    for (auto n : names) mw_check(mw_nc_def_var(nc, n, 3, d3, &v));
    mw_check(mw_nc_enddef(nc));
    auto put1 = [&](const char *name, long long start, std::vector<double> const &a) {
      long long st[1] = {start}, ct[1] = {(long long)a.size()};
      mw_check(mw_nc_inq_varid(nc, name, &v)); mw_check(mw_nc_put_vara_double(nc, v, st, ct, a.data())); };
    std::vector<double> xs(g.nx), ys(g.ny), zs(g.nz);
    for (int i = 0; i < g.nx; i++) xs[i] = (i + g.i_beg + 0.5) * coupler.get_dx();
    for (int j = 0; j < g.ny; j++) ys[j] = (j + g.j_beg + 0.5) * coupler.get_dy();
    for (int k = 0; k < g.nz; k++) zs[k] = (k + 0.5) * coupler.get_dz();
    put1("x", g.i_beg, xs); put1("y", g.j_beg, ys); put1("z", 0, zs);
    auto a = six_fields(coupler, "time_avg_");
    for (int l = 0; l < 6; l++) { mw_check(mw_nc_inq_varid(nc, names[l], &v)); mw_check(mw_output_put_field(nc, v, -1, &g, a[l], nullptr)); }
    mw_check(mw_nc_close(nc));
  }
};

} // namespace custom_modules
