/* =====================================================================================================
 * mw_cdna4.h -- C ABI of the MI355X-native (gfx950 / CDNA4) miniWeatherML hot path.
 *
 * One shared library, libmw_cdna4.so (miniweatherml_amd/csrc), exports exactly the symbols declared
 * here.  Plain pointers and sizes only; no C++ or torch types.  All field pointers are DEVICE pointers
 * (hipMalloc'd / torch.cuda tensors) unless a parameter is documented as "host".  All arithmetic is IEEE
 * fp64 except the MLP (fp32 inside, fp64 at the boundary).  Array layout everywhere: C row-major, last
 * index fastest, coupler fields (nz,ny,nx,nens)  -- reference: model/core/coupler.h:323-330.
 *
 * Every entry point returns 0 on success, non-zero on failure; mw_last_error() then returns the message
 * (replaces the reference's endrun()/yakl_throw abort, model/main_header.h:66-68).  A missing GPU, a
 * failed kernel launch or an unsupported option is an ERROR -- there is no CPU fallback in this library.
 *
 * Each function cites the reference interface it replaces (paths relative to the reference repo root).
 * ===================================================================================================== */
#ifndef MW_CDNA4_H
#define MW_CDNA4_H

#ifdef __cplusplus
extern "C" {
#endif

#define MW_NUM_STATE 5          /* dynamics_euler_stratified_wenofv.h:31 */
#define MW_MAX_TRACERS 16

/* test cases, dynamics_euler_stratified_wenofv.h:41-44 */
enum { MW_DATA_THERMAL = 0, MW_DATA_SUPERCELL = 1, MW_DATA_CITY = 2, MW_DATA_BUILDING = 3 };
/* boundary conditions, dynamics_euler_stratified_wenofv.h:46-48 */
enum { MW_BC_PERIODIC = 0, MW_BC_OPEN = 1, MW_BC_WALL = 2 };

/* Everything the dycore pulls out of core::Coupler getters and options
 * (coupler.h:219-278 geometry getters; options set/read at dynamics_euler_stratified_wenofv.h:211-226,
 * 1227-1249,1300,1312,1332-1335). */
typedef struct {
  int nz, ny, nx, nens, num_tracers;          /* local sizes: coupler.get_nz/ny/nx/nens/num_tracers            */
  long long nx_glob, ny_glob, i_beg, j_beg;   /* coupler.get_nx_glob/ny_glob/i_beg/j_beg                       */
  double xlen, ylen, zlen;                    /* coupler.get_xlen/ylen/zlen ; dx = xlen/nx_glob etc.           */
  int px, py, nproc_x, nproc_y;               /* coupler.get_px/py/nproc_x/nproc_y                             */
  int neigh[9];                               /* coupler.get_neighbor_rankid_matrix(), [y][x] row-major 3x3    */
  int bc_x, bc_y, bc_z;                       /* options "bc_x","bc_y","bc_z"                                  */
  int use_immersed, enable_gravity;           /* options "use_immersed_boundaries","enable_gravity"            */
  int idWV;                                   /* option "idWV": index of the tracer named water_vapor          */
  double R_d, R_v, cp_d, cp_v, p0, grav, gamma_d, kappa_d, C0, earthrot, latitude;   /* real options           */
} mw_grid_t;

typedef struct mw_dycore_s *mw_dycore_t;      /* opaque: module-private persistent state (hy_dens_*, flags,     */
                                              /* etime, workspace) -- the reference keeps it in the module      */
                                              /* object, dynamics_euler_stratified_wenofv.h:51-66               */

const char *mw_last_error(void);
int  mw_device_count(void);                   /* >0 iff a HIP device is usable                                  */

/* ---- decomposition + constants (host only) -------------------------------------------------------- */
/* core::Coupler::distribute_mpi_and_allocate_coupled_state, coupler.h:110-214 (factorisation :133-140,
 * index ranges :147-153, neighbour matrix :169-179).  Fills nx,ny,nx_glob,ny_glob,i_beg,j_beg,px,py,
 * nproc_x,nproc_y,neigh of *g; leaves the rest untouched. */
int  mw_decompose(int nranks, int myrank, long long nx_glob, long long ny_glob, mw_grid_t *g);
/* Physical constants exactly as Microphysics_Kessler::init then Dynamics::init leave them in the coupler
 * options: microphysics_kessler.h:29-41,86-95 ; dynamics_euler_stratified_wenofv.h:1227-1249. */
int  mw_default_constants(mw_grid_t *g);
/* Dynamics_Euler_Stratified_WenoFV::compute_time_step, dynamics_euler_stratified_wenofv.h:70-77 */
double mw_dycore_compute_time_step(const mw_grid_t *g);

/* ---- dycore --------------------------------------------------------------------------------------- */
/* Allocates the device workspace (two halo'd prognostic slabs + six flux arrays, reused across calls --
 * the reference re-allocates them every call/cycle/stage, :97-98,:112-115,:260-265).  `stream` is a
 * hipStream_t (NULL = default stream); all work of this handle is ordered on it.
 * tracer_positive / tracer_adds_mass: host arrays [num_tracers], Coupler::get_tracer_info (:1285-1293).
 * Fails (before any large allocation) when one variable of the block's halo'd slab has 2^31 or more elements (16 GB): the kernels keep
 * their strides in 32 bits, and four slabs of such variables would not fit the GPU anyway. */
int  mw_dycore_create(mw_dycore_t *h, const mw_grid_t *g, const unsigned char *tracer_positive,
                      const unsigned char *tracer_adds_mass, void *stream);
void mw_dycore_destroy(mw_dycore_t h);

/* Dynamics_Euler_Stratified_WenoFV::init, :1197-1683 (file output :1659 excluded).  Builds the hydrostatic
 * background + immersed_proportion for `init_data` (MW_DATA_*), sets bc_x/bc_y/bc_z/use_immersed/latitude as
 * the reference does, and writes the initial COUPLER fields (density_dry,uvel,vvel,wvel,temp,tracers[T]). */
int  mw_dycore_init(mw_dycore_t h, int init_data, double *density_dry, double *uvel, double *vvel, double *wvel,
                    double *temp, double *const *tracers /* host array of T device ptrs */);
/* For callers that build their own initial state: set the persistent background directly.
 * hy_*: HOST arrays (nz,nens) / (nz+1,nens); immersed_proportion: DEVICE (nz,ny,nx,nens) or NULL (= zeros). */
int  mw_dycore_set_background(mw_dycore_t h, const double *hy_dens_cells, const double *hy_dens_theta_cells,
                              const double *hy_dens_edges, const double *hy_dens_theta_edges,
                              const double *immersed_proportion);
/* Copies the background to HOST arrays (any may be NULL).  Registered in the coupler by the reference at
 * :1663-1668 (cells) ; edges are module members :53-54. */
int  mw_dycore_get_background(mw_dycore_t h, double *hy_dens_cells, double *hy_dens_theta_cells,
                              double *hy_dens_edges, double *hy_dens_theta_edges);
/* DEVICE pointer of immersed_proportion (nz,ny,nx,nens), registered by the reference at :1313. */
double *mw_dycore_immersed_proportion(mw_dycore_t h);
/* Current grid/options of the handle (bc_*, use_immersed, latitude may have been set by mw_dycore_init). */
int  mw_dycore_get_grid(mw_dycore_t h, mw_grid_t *g);
/* coupler options "bc_x","bc_y","bc_z" (read at :588-590, :846-848); mw_dycore_init sets them like :1332-1334. */
int  mw_dycore_set_bc(mw_dycore_t h, int bc_x, int bc_y, int bc_z);
/* Kernel path.  0 (default): production path (shared reconstruction, marching kernels, state fluxes never
 * materialised; re-associated WENO and series pressure).  1: "strict": the general flux-materialising kernels in the
 * reference's exact operation order with FMA contraction off (diagnostic / parity proof; also env MW_STRICT=1 at
 * create).  2: the general flux-materialising kernels with the fast arithmetic (A/B of the two kernel structures). */
int  mw_dycore_set_strict(mw_dycore_t h, int strict);
/* Run-time options of a handle (no reference counterpart: the reference's variants are compile-time).  They replace the MW_* environment
 * switches of rounds 1-4: typed integers stored in the handle, read when the schedule of a time step is decided; no entry point of this
 * library reads the environment per call (the two process defaults left: MW_STRICT=1 at mw_dycore_create, MW_RCCL_LANES for the
 * built-in transport).  Schedule: "overlap" (-1 automatic | 0 | 1: the two-stream schedule), "pipe" (1: the pipelined schedule of a
 * decomposed block), "pipe_edge_inline", "pipe_convert", "pipe_split_edges", "pipe_maps_early".  Kernel forms: "spec" (folded configurations), "wrap" (index wrap on one
 * rank), "y_all", "y_all_conv", "member_major", "mm_direct", "mm_conv", "fused_convert", "fused_convert_mm", "fused_tracers", "tf_rows4", "zero_skip" (the
 * marching kernels skip the reconstructions of a tracer that is exactly zero over a wavefront's whole stencil: bit-neutral, 1 by default),
 * "zero_rows" (on top of it: per sub-cycle a map of the x rows in which a tracer can be non-zero -- scanned from the input, grown by the three
 * cells per direction an RK stage can move it, OR-ed with the neighbour blocks' maps on a decomposed domain -- lets the fused tracer
 * kernel neither load nor compute rows of cloud / rain that are zero; same results, 1 by default), "zero_stores" (... nor store zeros over
 * rows that hold zeros already: the coupler's own arrays, the stage slabs; 1 by default), "zero_verify" (a test aid, 0 by default: every
 * claim of the maps is checked against the data in front of the launch that relies on it -- mw_debug_zero_violations).
 * Launch shapes: "chunk_y", "chunk_yt", "chunk_z", "chunk_f" (cells per chunk, 0 = the chunk model), "chunk_model".  Built-in transport
 * (read when mw_dycore_use_rccl / _self installs it): "rccl_lanes" (0 = process default | 1 | 2), "rccl_two_comms" (-1 | 0 | 1), "rccl_prio" (1: side streams at the highest priority), "rccl_inline" (1: the group runs on the caller's stream, no side stream),
 * "xchg_fuzz" (seed of random delays around the sends / receives; a test aid), "debug_no_patch" (a test aid: the y-face correction pass of the
 * fused tracer stage is not launched -- the negative control of the FCT tests).  Unknown keys and out-of-range values are errors.  (The
 * experiment builds of rounds 4-5 -- the fused x-y-z state kernel, the balanced launch lists, the timing hooks -- left the tree in round 6.) */
int  mw_dycore_set_option(mw_dycore_t h, const char *key, long long value);
int  mw_dycore_get_option(mw_dycore_t h, const char *key, long long *value);
/* Optional parts this build of the library contains: always 0 (there are none any more; kept in the ABI). */
int  mw_build_flags(void);
/* WENO order, the reference's compile-time -DMW_ORD (dynamics_euler_stratified_wenofv.h:24-29; 3 in build/machines/aws/aws_a100_gpu.env:21):
 * 5 (default), 3, 7 or 9.  Call before mw_dycore_init (the supercell initial data uses `ord` GLL points, :1725-1886).  Orders 3, 7
 * and 9 run on the general flux-materialising kernels; 7 and 9 (WenoLimiter<7> / <9>, hs = 3 / 4) re-allocate the handle's slabs
 * with 4- / 5-cell x, y halos and 3 / 4 z levels and exchange strips of that depth. */
int  mw_dycore_set_order(mw_dycore_t h, int ord);

/* Dynamics_Euler_Stratified_WenoFV::time_step(coupler, dt_phys), :81-198: convert-in, ncycles x SSPRK3,
 * convert-out, etime += dt_phys.  Fields are updated in place.  Asynchronous on the handle's stream. */
int  mw_dycore_time_step(mw_dycore_t h, double *density_dry, double *uvel, double *vvel, double *wvel, double *temp,
                         double *const *tracers /* host array of T device ptrs */, double dt_phys);
/* One compute_tendencies(state(coupler fields), dt) as the first RK stage sees it, :204-552: fills the six
 * public flux arrays and writes state_tend (5,nz,ny,nx,nens) / tracers_tend (T,nz,ny,nx,nens) (DEVICE, may be
 * NULL).  Test/diagnostic entry; the fields are not modified. */
int  mw_dycore_compute_tendencies(mw_dycore_t h, const double *density_dry, const double *uvel, const double *vvel,
                                  const double *wvel, const double *temp, double *const *tracers, double dt,
                                  double *state_tend, double *tracers_tend);
/* The six persistent flux arrays the reference registers "so the user has access", :1671-1676:
 * out[0..2] = state_flux_x,y,z (5,nz[+1],ny[+1],nx[+1],nens); out[3..5] = tracers_flux_x,y,z (T,...). DEVICE. */
int  mw_dycore_get_fluxes(mw_dycore_t h, double **out6);
double mw_dycore_get_etime(mw_dycore_t h);   /* member etime, :55 */
/* Name + total time (ms) of this handle's kernels measured with hipEvents on the handle's stream since the
 * last reset (enabled by mw_dycore_profile(h,1); mw_dycore_profile(h,2) times class 0 only -- two events per launch of the
 * dominant kernel instead of two around every kernel, whose markers cost about 2 % of the step); which: 0 x/z flux stencil + state update (k_xz_state; k_flux on the
 * general path), 1 fct (general path) / y-face correction pass of the fused tracer stage, 2 update (general path; tracer update
 * of the unfused production path), 3 halo, 4 convert, 5 y stencil state, 6 y stencil tracers, 7 x/z tracer stage (fused:
 * fluxes + FCT + update), 8 one whole RK stage of the production path (first to last launch on the handle's stream; also
 * recorded by mw_dycore_profile(h,2)), 9 one whole mw_dycore_time_step (the only class of mw_dycore_profile(h,3): two events per time step
 * instead of twelve -- the per-stage pairs of mode 2 cost 1.5 % of the step they time), 10 / 11 (mode 1, pipelined schedule of a decomposed
 * block): the time the compute stream sits waiting for the stage's state strips / tracer strips -- what the exchange costs on the critical path. */
/* Which schedule the last mw_dycore_time_step chose: 0 one stream, 1 two streams (state | tracer pipelines), 2 pipelined one-stream
 * schedule of a decomposed block; + 4: y faces of all variables in one launch (k_y_all); + 8: general (flux-materialising) kernels. */
int  mw_dycore_schedule(mw_dycore_t h);
/* The same decision spelled out (valid until the handle's next call): "march | general-strict | general-fast", the WENO order, and on the
 * marching kernels the folded configuration (K0 | K1 | K2), the internal layout (nens1 | fused_members | member_major | mm_direct), the
 * schedule (one_stream | two_stream | pipe), y_all | y_split, where D1 happened (conv_in_y | conv_pipe | conv_pass), tracers_fused |
 * tracers_unfused, 2d | 3d, "transport" with a halo exchange installed.  The tests log it with every parity comparison and assert at
 * session end that every combination the dispatcher can produce was compared (tests/conftest.py). */
const char *mw_dycore_path(mw_dycore_t h);
int  mw_dycore_profile(mw_dycore_t h, int enable);
int  mw_dycore_profile_get(mw_dycore_t h, int which, double *total_ms, long long *launches);

/* Diagnostic: weno::WenoLimiter<5>::compute_limited_coefs + coefs_to_gll_lower<5,2> (WenoLimiter.h:68-93, TransformMatrices.h:1132-1144)
 * on n caller-supplied 5-cell stencils (DEVICE (n,5)) -> the two edge values (DEVICE (n,2)); strict = 1: the reference's operation
 * order with contraction off, 0: the production arithmetic (mw_weno.h). */
int  mw_weno5_edges(long long n, const double *stencils, double *edges, int strict, void *stream);
/* Diagnostic: out[i] = pow(x[i], y[i]) as the kernels that keep the reference's operation order compute it (strict path, init,
 * D1 / D13 passes): glibc's algorithm and tables (csrc/mw_glibc_pow.h), i.e. the bits of the host libm's pow that the reference
 * built with the YAKL serial backend gets (dynamics_euler_stratified_wenofv.h:401, :1935, :2009).  main_path (device, n bytes,
 * may be NULL): 1 where the restated main path applied, 0 where the device library's pow was used (arguments the dycore never
 * produces: x <= 0 or subnormal, |y| < 2^-65 or >= 2^63, over- / underflowing results). */
int  mw_strict_pow(long long n, const double *x, const double *y, double *out, unsigned char *main_path, void *stream);

/* Measurement aid (no reference counterpart): copies n doubles with this library's access shape (8 B per lane).  A launch
 * moves exactly 8n bytes each way, which calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE counters (tools/calib_pmc.py). */
int  mw_calib_copy(const double *in, double *out, long long n, void *stream);

/* Calibration (no reference counterpart; SURVEY.md 8(d) asks for a measured fp64 ceiling).  mw_calib_fma64: independent v_fma_f64
 * chains, `waves_per_simd` (1..8) wavefronts per SIMD on every CU for about `seconds`; out5 (HOST) = wave-instructions per second,
 * kernel ms, shader clock GHz during the run (0 if the part's two counters tick alike), wave-instructions issued, CUs.
 * mw_calib_stage_arith: the arithmetic of one RK stage and nothing else -- per cell 24 WENO-5 reconstructions + 3 Riemann solves + the
 * passive fluxes (the production arithmetic) on register windows fed from `tab`, DEVICE (nlev >= 6, 8, 64) doubles that stay in L2
 * (variables rho', u, v, w, (rho theta)', q_v, q_c, q_r of a 64-cell row; smooth or rough data as the caller likes), `levels` cells per
 * thread, 256-thread workgroups, two per CU (k_xz_state's shape and register budget).  active_tracers = 3: all 24 reconstructions; 1: cloud and rain
 * exactly zero, which the production kernels do not reconstruct (option zero_skip): 18 per cell, the floor of a stage on a cloud-free state.  bg4 (HOST): hy_dens, hy_dens_theta, C0
 * hy_dens_theta^gamma, 1 / hy_dens_theta of the level.  sink: DEVICE, mw_calib_stage_arith_threads(cells, levels) doubles.  out3
 * (HOST): ms of the timed launch, cells processed, workgroups.  No stage of that many cells can take less on this chip. */
int  mw_calib_fma64(int waves_per_simd, double seconds, double *out5, void *stream);
long long mw_calib_stage_arith_threads(long long cells, int levels);
int  mw_calib_stage_arith(const double *tab, int nlev, long long cells, int levels, int active_tracers, const double *bg4, double *sink, double *out3, void *stream);
/* Test aid: occupies `stream` for about `usec` microseconds (delay fuzz of the exchange tests). */
int  mw_debug_spin(long long usec, void *stream);
/* Test aid: the names (as the code object spells them -- mangled; newline-separated) of the dycore kernels this PROCESS has launched since
 * the last reset; every instantiation of the dispatcher's kernel templates has its own.  Returns the bytes needed (terminator included)
 * and writes at most `cap`; reset != 0 clears the registry.  (tests/conftest.py: instantiation coverage of the parity comparisons.) */
long long mw_debug_launched_kernels(char *buf, long long cap, int reset);
/* Test aid: the zero-row maps of the handle's LAST sub-cycle, copied to HOST memory after a stream synchronise: ten maps of
 * nz * (ny + 18) 32-bit words each -- M0, Q1..Q3, FN1..FN3, QY1..QY3; word of (level k, row j) at [k * (ny + 18) + j + 9], bit v = tracer v.
 * Returns the number of words (0: the last time step ran without maps), writes at most cap_words; dims2 = {nz, ny + 18}. */
long long mw_debug_zero_maps(mw_dycore_t h, unsigned int *out_host, long long cap_words, int *dims2);
/* Test aid: option "zero_verify" = 1 makes every time step check the zero-row maps' claims against the data, right in front of the launches
 * that rely on them (k_zero_verify): [0] / [1] a tracer that can vanish is non-zero in a row of a stage's input although the map of the fused
 * tracer kernel / of the y kernel says it cannot be; [2] / [3] a destination row whose store of zeros is about to be skipped (slab S1 / S2,
 * the coupler's arrays / the slab the converting y launch fills) is not all zero.  Returns the sum of the four counters since the handle was
 * created (out4, may be NULL: the four), -1 when the option never ran.  No reference counterpart. */
long long mw_debug_zero_violations(mw_dycore_t h, unsigned long long *out4);

/* modules::perturb_temperature(coupler, thermal=true, random=false), perturb_temperature.h:41-66 */
int  mw_perturb_temperature(const mw_grid_t *g, double *temp, void *stream);
/* the random = true branch, perturb_temperature.h:25-39 (applied BEFORE the thermal, as there): +-3 K uniform noise on the lowest
 * nz/4 levels, fading linearly; one draw per (level, column) from the key myrank*nz*nx*ny*nens + k*ncol + i.  yakl::Random is
 * unavailable: the key goes through the splitmix64 finaliser ("Substitutions"). */
int  mw_perturb_temperature_random(const mw_grid_t *g, double *temp, void *stream);

/* ---- multi-GPU halo exchange ---------------------------------------------------------------------- */
/* halo_exchange, :574-747 (MPI_Isend/Irecv W/E/S/N, tags 0-3).  The native design ships a 3-cell halo once
 * per RK stage and reconstructs the neighbour's edge values locally, which makes edge_exchange (:830-1003)
 * unnecessary (bitwise identical: reconstruction is a pure function of the 5-cell stencil).
 * The callback receives DEVICE buffers packed as (V,nz,ny,3,nens) [W/E] and (V,nz,3,nx,nens) [S/N] and must
 * deliver sendW to the west neighbour's recvE etc., ordered on `stream`.  NULL => single-rank periodic wrap. */
/* Posting plan of one exchange (host only; single source of truth for mw_rccl.cpp and for the CPU gloo tests):
 * peers[4] = west, east, south, north rank (neigh(1,0), neigh(1,2), neigh(0,1), neigh(2,1), :651-655);
 * send_order[4] / recv_order[4] = directions (0 W, 1 E, 2 S, 3 N) in posting order.  Messages between one pair of
 * ranks match in FIFO order, so when west == east (two ranks in x) or south == north the receives are posted E,W / N,S
 * against sends W,E / S,N: the peer's first send (its W strip) is my E halo.  active[4]: direction takes part
 * (more than one rank in that direction and, for S/N, a 3-D run). */
int  mw_exchange_plan(const mw_grid_t *g, int *peers, int *send_order, int *recv_order, int *active);
typedef int (*mw_exchange_fn)(void *ctx, const double *sendW, const double *sendE, const double *sendS,
                              const double *sendN, double *recvW, double *recvE, double *recvS, double *recvN,
                              long long nWE, long long nSN, void *stream);
int  mw_dycore_set_exchange(mw_dycore_t h, mw_exchange_fn fn, void *ctx);
/* Built-in exchange over RCCL point-to-point (ncclSend/ncclRecv in one group on a side stream).
 * unique_id: the 128-byte ncclUniqueId created on rank 0 (mw_rccl_unique_id) and broadcast by the host. */
int  mw_rccl_unique_id(unsigned char *id128);
int  mw_dycore_use_rccl(mw_dycore_t h, const unsigned char *id128, int nranks, int myrank);
/* Test transport (no reference counterpart): ONE rank plays every rank of the handle's nproc_x x nproc_y rank grid -- a 1-rank
 * communicator, every active direction's peer is this rank itself (sends W,E,S,N matched FIFO by receives E,W,N,S), i.e. the messages a
 * block exchanges with neighbours that hold the same data.  For a periodic domain tiled from copies of one block the handle's result
 * equals the one-rank run of that block bit for bit, with the real send / receive groups, side streams and event pairs of
 * halo_exchange's replacement in flight beside compute on one GPU.  mw_dycore_rccl_allreduce_sum on such a handle multiplies by the
 * number of blocks (identical contributions; exact), mw_dycore_rccl_bcast is the identity. */
int  mw_dycore_use_rccl_self(mw_dycore_t h);
/* RCCL is resolved at run time from the librccl ALREADY mapped in the process (a PyTorch host: torch's own, the one behind
 * torch.distributed's "nccl" backend), else from the loader's search path -- never two RCCLs in one process.  Returns the
 * path it came from ("" when none is available) and, optionally, its version code (ncclGetVersion). */
const char *mw_rccl_library_path(int *version);
/* ncclCommCount / ncclCommUserRank of the installed transport's communicator, and its number of lanes (side streams): what a
 * multi-GPU run reports as evidence that RCCL connected all ranks.  Error if the handle's transport is not the built-in RCCL one. */
int  mw_dycore_rccl_info(mw_dycore_t h, int *comm_ranks, int *comm_rank, int *lanes);
/* The other two collectives of a decomposed run, on the handle's communicator, for hosts without torch.distributed / MPI:
 * MPI_Allreduce(SUM) of sponge_layer.h:53-63 / column_nudging.h:89-99 -- pass this function as the mw_allreduce_fn of mw_sponge_layer /
 * mw_column_average / mw_nudge_to_column with ctx = the dycore handle -- and MPI_Bcast(root) of horizontal_sponge.h:72-77.  DEVICE
 * buffers, in place, ordered on `stream`. */
int  mw_dycore_rccl_allreduce_sum(void *dycore_handle, double *buf, long long n, void *stream);
int  mw_dycore_rccl_bcast(mw_dycore_t h, double *buf, long long n, int root, void *stream);
/* Diagnostic: one rank sends 4 strips of n doubles to itself through the exchange's own ncclGroup / side stream / event
 * sequence and compares; 0 = RCCL initialises on this box and the ordering against `stream` holds. */
int  mw_rccl_selftest(long long n, void *stream);
/* What the last mw_rccl_selftest drove: 10 x the number of exchange lanes (side stream + event pair each; one per pipeline of the
 * two-stream schedule) + the number of communicators behind them: 21 = two lanes on one communicator (default), 22 = a communicator
 * per lane (split with ncclCommSplit).  mw_rccl_selftest_config chooses the form of the following self-tests of this process: lanes
 * 1 | 2 (0 = back to the process default, MW_RCCL_LANES = "1" | "2" | "2x2"), two_comms 0 | 1. */
int  mw_rccl_selftest_lanes(void);
int  mw_rccl_selftest_config(int lanes, int two_comms);

/* ---- Kessler microphysics ------------------------------------------------------------------------- */
/* Microphysics_Kessler::time_step(coupler, dt), microphysics_kessler.h:99-162 + kessler() :234-339.
 * (nz,ncol) views of water_vapor, cloud_liquid, precip_liquid, density_dry(const), temp; precl (ncol).
 * rainsplit is taken from THIS rank's minval exactly like the reference (:276-279).  workspace: DEVICE scratch
 * of mw_kessler_workspace_bytes(nz,ncol) bytes.  rainsplit_out: optional HOST int (forces a stream sync). */
long long mw_kessler_workspace_bytes(int nz, long long ncol);
int  mw_kessler_time_step(int nz, long long ncol, double dz, double dt, double *rho_v, double *rho_c, double *rho_r,
                          const double *rho_d, double *temp, double *precl, void *workspace, int *rainsplit_out,
                          void *stream);
/* 1: the STRICT Kessler path -- the reference's formulas in the reference's operation order (theta form, IEEE divisions, no
 * contraction) with glibc's pow and exp (csrc/mw_glibc_pow.h): the bits microphysics_kessler.h:99-162, :234-339 produces on a glibc
 * host (YAKL serial backend).  0 (default): the production kernels (1e-12 from it).  Process-wide, like the module's constants. */
int  mw_kessler_set_strict(int strict);

/* ponni::load_h5_weights<N>(file, group, dataset), microphysics_kessler_ponni.h:103-107: one 32-bit float dataset of an HDF5 file
 * (the Keras weight file `keras_weights_h5`: "/dense_6/dense_6" "kernel:0" (5,10), "bias:0" (10), "/dense_7/dense_7" ...), read by a
 * dependency-free HOST reader (old-style groups, contiguous / compact little-endian float data; anything else is refused).
 * out == NULL: shape query.  dims: up to 8 extents, ndims: rank.  capacity: number of floats `out` holds. */
int  mw_h5_read_f32(const char *file, const char *group, const char *dataset, float *out, long long capacity, long long *dims, int *ndims);

/* mean(a - b) over n DEVICE doubles -- the surrogate module's "Relative diff" prints, microphysics_kessler_ponni.h:266-269
 * (yakl::intrinsics::sum(a - b) / size there).  Deterministic order.  workspace1024: DEVICE scratch of 1024 doubles; mean_out: HOST. */
int  mw_mean_diff(long long n, const double *a, const double *b, double *workspace1024, double *mean_out, void *stream);

/* Diagnostic (no reference counterpart): the Kessler module's own log (fn 0), exp (1), sqrt (2) and reciprocal (3) -- short
 * forms for positive finite arguments, see mw_kessler.hip -- on n caller-supplied DEVICE doubles. */
int  mw_kessler_math_probe(long long n, const double *x, double *y, int fn, void *stream);

/* ---- sponge layer + column nudging (the two remaining per-step modules of the supercell loop) ---- */
/* Sums `buf` (DEVICE, n doubles) in place over all ranks, ordered on `stream` -- MPI_Allreduce(SUM) in the reference
 * (sponge_layer.h:53-63, column_nudging.h:89-99).  Pass NULL on a single rank. */
typedef int (*mw_allreduce_fn)(void *ctx, double *buf, long long n, void *stream);
/* DEVICE scratch size for the three calls below (num_fields = 5 + T for the sponge, 5 for the nudger). */
long long mw_column_workspace_bytes(const mw_grid_t *g, int num_fields);
/* modules::sponge_layer(coupler, dt, time_scale = 60), sponge_layer.h:8-77: relax the top 10 levels of every field to the
 * horizontal mean (w to zero).  fields: HOST array of num_fields DEVICE pointers in the reference's MultiField order
 * (density_dry, uvel, vvel, wvel, temp, tracers...).  Horizontal sums are deterministic (no atomics). */
/* 1: the horizontal sums of mw_sponge_layer / mw_column_average / mw_nudge_to_column are added in the reference's SERIAL order
 * (j, then i: sponge_layer.h:44-51, column_nudging.h:80-87 on the YAKL serial backend) by one thread per (field, level, member), which
 * makes the two modules bit-identical to that backend; 0 (default): deterministic fixed-slice tree sums.  Process-wide. */
int  mw_column_set_strict(int strict);
int  mw_sponge_layer(const mw_grid_t *g, double *const *fields, int num_fields, double dt, double time_scale, void *workspace,
                     mw_allreduce_fn allreduce, void *ctx, void *stream);
/* ColumnNudger::get_column_average, column_nudging.h:69-106: state5 = density_dry, uvel, vvel, temp, water_vapor;
 * column_out: DEVICE (5,nz,nens).  set_column (:15-36) is this call on the initial state. */
int  mw_column_average(const mw_grid_t *g, const double *const *state5, double *column_out, void *workspace,
                       mw_allreduce_fn allreduce, void *ctx, void *stream);
/* ColumnNudger::nudge_to_column(coupler, dt), :39-66: state += dt (column - column_average(state)) / 900. */
int  mw_nudge_to_column(const mw_grid_t *g, double *const *state5, const double *column, double dt, void *workspace,
                        mw_allreduce_fn allreduce, void *ctx, void *stream);
/* The same, DEFERRED (round 6; no reference counterpart -- the reference's nudge_to_column is two passes): the horizontal sums are taken and the
 * increments dt (column - average) / 900 are PARKED in the dycore handle instead of being added by a second pass over the five fields.  The
 * handle's next mw_dycore_time_step adds them while its conversion loads the coupler's arrays (the same rounded addition: results are bit for
 * bit those of mw_nudge_to_column), on the one-rank production path of the supercell configuration; any other path, and
 * mw_dycore_compute_tendencies, applies them with a pass first.  Between this call and that time step the five arrays do NOT hold the nudged
 * values: whoever reads them must call mw_dycore_flush_pending first (the Coupler mirrors do so inside DataManager::get).  The grid is the
 * handle's; sums and increments are ordered on the handle's stream.  state5: density_dry, uvel, vvel, temp, water_vapor -- the arrays later
 * handed to mw_dycore_time_step. */
int  mw_nudge_to_column_deferred(mw_dycore_t h, double *const *state5, const double *column, double dt, void *workspace,
                                 mw_allreduce_fn allreduce, void *ctx);
/* Applies parked increments now (a no-op when there are none). */
int  mw_dycore_flush_pending(mw_dycore_t h);
/* 1 while increments are parked; out2 (may be NULL): how often parked increments rode on a conversion / were applied by a pass, since create. */
int  mw_dycore_pending(mw_dycore_t h, unsigned long long *out2);

/* ---- ponni surrogate MLP -------------------------------------------------------------------------- */
/* NN block of custom_modules::Microphysics_Kessler::time_step,
 * experiments/supercell_kessler_surrogate/custom_modules/microphysics_kessler_ponni.h:176-202
 * (ponni::Matvec,Bias,Relu(0.1),Matvec,Bias ; model.forward_batch_parallel :189), fused with the min-max
 * input scaling :182-186 and output un-scaling + clip :198-201.  W1 (5,10), b1 (10), W2 (10,4), b2 (4):
 * HOST fp32, Keras (in,out) layout; scl_in (5,2), scl_out (4,2): HOST fp64 [min,max] rows. */
int  mw_mlp_forward(long long ncells, const double *temp, const double *rho_d, const double *rho_v,
                    const double *rho_c, const double *rho_r, const float *W1, const float *b1, const float *W2,
                    const float *b2, const double *scl_in, const double *scl_out, double *temp_out,
                    double *rho_v_out, double *rho_c_out, double *rho_r_out, void *stream);
/* 1: the strict form -- thread per cell, fp32 accumulation in index order, no contraction (the order in which the layers are defined,
 * microphysics_kessler_ponni.h:103-110); 0 (default): the MFMA kernels (the same products summed in the matrix cores' order). */
int  mw_mlp_set_strict(int strict);

/* ponni's own surface: ponni::Inference<...>::forward_batch_parallel(float2d in(num_in, batch)) -> float2d
 * (num_out, batch), experiments/supercell_kessler_surrogate/custom_modules/microphysics_kessler_ponni.h:40-45,103-110,189, for a stack
 * of ponni::Matvec<float> (kind 0: weights (n_in, n_out) in Keras order, y = x W), ponni::Bias<float> (kind 1) and ponni::Relu<float>(n,
 * negative_slope) (kind 2) layers.  layers / params: HOST memory (params = all weights back to back, `offset` in floats); in / out:
 * DEVICE fp32, batch fastest.  The surrogate's stack (5 -> 10 -> 4) runs on the MFMA tiles, any other stack of up to 10 layers, widths
 * <= 32 and 960 parameters on a thread-per-element kernel; a size mismatch between consecutive layers is an error (Inference::validate). */
typedef struct { int kind, n_in, n_out; float negative_slope; int offset; } mw_ponni_layer_t;
int  mw_ponni_forward(const mw_ponni_layer_t *layers, int nlayers, const float *params, int nparams, long long batch,
                      const float *in, float *out, void *stream);

/* ---- file output ------------------------------------------------------------ */
/* A minimal netCDF *classic* writer (mw_netcdf.cpp): the reference writes through PnetCDF with NC_CLOBBER | NC_64BIT_DATA,
 * i.e. the CDF-5 on-disk format, dims x,y,z (+ unlimited t), double variables, no attributes
 * (dynamics_euler_stratified_wenofv.h:2106-2131, time_averager.h:104-120).  format: 5 = CDF-5 (the reference's), 2 = CDF-2
 * (64-bit offset; same structure with 32-bit sizes -- readable by readers that predate CDF-5).  header_align / var_align are
 * PnetCDF's nc_header_align_size / nc_var_align_size hints (:2103-2104: 1048576 each; <= 0: 512 / 4).  All HOST calls.
 * Sharing: the creating rank runs create..enddef; after that any process on the node may mw_nc_open the file and write
 * disjoint hyperslabs (pwrite); the main rank publishes the record count with mw_nc_set_numrecs. */
typedef struct mw_nc_s *mw_nc_t;
int  mw_nc_create(mw_nc_t *nc, const char *path, int format, long long header_align, long long var_align);   /* nc.create        */
int  mw_nc_def_dim(mw_nc_t nc, const char *name, long long len, int *dimid);          /* create_dim; len 0 = create_unlim_dim   */
int  mw_nc_def_var(mw_nc_t nc, const char *name, int ndims, const int *dimids, int *varid);   /* create_var<real>               */
/* nc_type 4 = int, 5 = float, 6 = double (the surrogate sample files hold floats and one int, generate_micro_surrogate_data.h) */
int  mw_nc_def_var_typed(mw_nc_t nc, const char *name, int nc_type, int ndims, const int *dimids, int *varid);
int  mw_nc_enddef(mw_nc_t nc);                                                         /* nc.enddef: lays out + writes header    */
int  mw_nc_open(mw_nc_t *nc, const char *path);                                        /* nc.open (read-write)                   */
int  mw_nc_inq_varid(mw_nc_t nc, const char *name, int *varid);
int  mw_nc_inq_dimlen(mw_nc_t nc, const char *name, long long *len);                   /* get_dim_size; record dim: # records    */
int  mw_nc_put_vara_double(mw_nc_t nc, int varid, const long long *start, const long long *count, const double *host_data);
int  mw_nc_put_vara(mw_nc_t nc, int varid, const long long *start, const long long *count, const void *host_data);   /* variable's own type */
int  mw_nc_set_numrecs(mw_nc_t nc, long long numrecs);
int  mw_nc_close(mw_nc_t nc);
/* The body of Dynamics_Euler_Stratified_WenoFV::output's variable loop, :2176-2185 (and Time_Averager::finalize's,
 * time_averager.h:131-136): ensemble member 0 of the DEVICE field (nz,ny,nx,nens) -> host -> this rank's hyperslab
 * {record, 0, j_beg, i_beg} x {1, nz, ny, nx} of variable varid (record < 0: a variable without the t dimension). */
int  mw_output_put_field(mw_nc_t nc, int varid, long long record, const mw_grid_t *g, const double *field, void *stream);

/* ---- simple_city custom modules ------------------------------------------- */
/* fields6 / avg6: HOST arrays of 6 DEVICE pointers: density_dry, uvel, vvel, wvel, temp, water_vapor. */
/* custom_modules::Horizontal_Sponge::init, horizontal_sponge.h:54-61: column (6,nz,nens) DEVICE = cell (k,0,0,iens) of this
 * rank; the reference takes the main rank's and MPI_Bcast's it (:73-78) -- the host broadcasts `column` from rank 0. */
int  mw_horizontal_sponge_column(const mw_grid_t *g, const double *const *fields6, double *column, void *stream);
/* Horizontal_Sponge::apply(coupler, dt, x1, x2, y1, y2), :101-192: cosine-weighted relaxation of the 6 fields to the stored
 * column within sponge_cells of the enabled domain edges (only on the ranks that own the edge). */
int  mw_horizontal_sponge_apply(const mw_grid_t *g, double *const *fields6, const double *column, int sponge_cells,
                                double time_scale, double dt, int x1, int x2, int y1, int y2, void *stream);
/* custom_modules::Time_Averager::accumulate, time_averager.h:37-78: avg = inertia*avg + (1-inertia)*field with
 * inertia = etime/(etime+dt); the caller advances its etime by dt afterwards (:77). */
int  mw_time_average_accumulate(const mw_grid_t *g, const double *const *fields6, double *const *avg6, double etime, double dt,
                                void *stream);

/* ---- surrogate workflow statistics ----------------------------------------- */
/* custom_modules::StatisticsGatherer::gather_micro_statistics, experiments/supercell_kessler_surrogate/custom_modules/
 * gather_micro_statistics.h:19-58: number of cells (ensemble member 0) whose temp / water_vapor / cloud_liquid /
 * precip_liquid changed by more than 1e-10 across the microphysics call (is_active, :61-74).  in4 / out4: HOST arrays of 4
 * DEVICE pointers (temp, water_vapor, cloud_liquid, precip_liquid) before / after; mask: optional DEVICE (nz,ny,nx) bytes
 * (the `active` array, :40-52); *count: the sum the reference adds to `numer` (:56).  Integer, hence exact. */
int  mw_micro_active_count(const mw_grid_t *g, const double *const *in4, const double *const *out4, unsigned char *mask,
                           long long *count, void *stream);
/* custom_modules::DataGenerator::generate_samples_stencil, .../generate_micro_surrogate_data.h:35-153, device parts.
 * mw_micro_sample_mask: mask[k,j,i] (DEVICE bytes) = u01(key0 + k*ny*nx + j*nx + i) < (is_active ? thr_active : thr_inactive),
 * :83-101; key0 = (seed + myrank)*nz*ny*nx as in the reference; u01 = splitmix64 (yakl::Random is not available).
 * mw_micro_gather_samples: for the n cell indices `cells` (DEVICE int64, k*ny*nx + j*nx + i, member 0) fills inputs (n,5,2) and
 * outputs (n,4) (DEVICE fp32) exactly as :139-156 (inputs: temp, density_dry, water_vapor, cloud_liquid, precip_liquid at k;
 * slot 1 as assigned there at min(nz-1,k+1); outputs: temp, water_vapor, cloud_liquid, precip_liquid after micro). */
int  mw_micro_sample_mask(const mw_grid_t *g, const double *const *in4, const double *const *out4, unsigned long long key0,
                          double thr_active, double thr_inactive, unsigned char *mask, void *stream);
int  mw_micro_gather_samples(const mw_grid_t *g, const double *rho_d, const double *const *in4, const double *const *out4,
                             const long long *cells, long long n, float *inputs, float *outputs, void *stream);

/* ---- DataManager validators ---------------------------------------------------------------------------- */
/* core::DataManager::validate / validate_nan / validate_inf / validate_pos (model/core/DataManager.h:385-483) -- the reference's only
 * built-in health check: it copies an entry to the host and loops over it.  Here ONE device pass over the entry's `n` elements
 * (DEVICE pointer): out6[0..2] = number of NaN / inf / negative elements, out6[3..5] = the lowest flat ("global") index of each
 * kind, i.e. the first one the reference's loop reports, or -1.  The caller decides which of the three apply (negative values only
 * matter for entries registered `positive`, :469) and whether to die (die_on_failed_check).  Synchronises the stream. */
int  mw_validate_f64(const double *field, long long n, long long *out6, void *stream);
int  mw_validate_f32(const float *field, long long n, long long *out6, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MW_CDNA4_H */
