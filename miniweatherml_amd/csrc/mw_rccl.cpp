// =====================================================================================================
// mw_rccl.cpp -- halo exchange over RCCL point-to-point (xGMI) for the slab-decomposed dycore.
// Replaces the MPI_Isend/Irecv/Waitall pattern of halo_exchange
// (reference: model/modules/dynamics_euler_stratified_wenofv.h:641-723; neighbour matrix coupler.h:169-179).
//
// One ncclGroup per exchange: up to four sends + four receives, each peer reached over its own xGMI link.  Since round 5 the group
// runs in line on the stream the dycore hands over (the handle's exchange stream in the pipelined schedule, a pipeline's own stream
// in the two-stream schedule); option rccl_inline = 0 keeps the transport's own side stream per lane (see rccl_exchange).  Message matching between one pair of ranks is FIFO, so when the west and east
// (or south and north) neighbour are the same rank (2 ranks in that direction) sends are posted W,E,S,N and
// receives E,W,N,S: the peer's first send (its W strip) is my E halo.
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include <rccl/rccl.h>      // types and prototypes only: the library is NOT linked, see RcclApi below
#include <dlfcn.h>
#include <link.h>
#include <cstring>
#include <mutex>
#include <vector>

namespace mw {   // defined in mw_dycore.hip: installs a transport whose context the handle owns (freed on replace / destroy)
int dycore_set_exchange_owned(mw_dycore_t h, mw_exchange_fn fn, void *ctx, void (*free_ctx)(void *));
void *dycore_exchange_ctx(mw_dycore_t h, mw_exchange_fn *fn);
int dycore_option(mw_dycore_t h, const char *key);            // a handle's run-time option (mw_dycore_get_option), 0 when unknown
int launch_spin(long long usec, hipStream_t st);               // mw_calib.h: a kernel that spins for `usec` microseconds (delay fuzz)
int launch_scale(double *buf, long long n, double f, hipStream_t st);   // buf[i] *= f
}

namespace {
// ---------------------------------------------------------------------------------------------------------------------
// ONE RCCL per process.  A PyTorch host has already mapped its own librccl (torch/lib/librccl.so, the one behind
// torch.distributed's "nccl" backend); linking libmw_cdna4.so against /opt/rocm's copy could put a second RCCL -- of another
// version -- into the process.  The entry points are therefore resolved at run time: first from whatever librccl is already
// mapped (dl_iterate_phdr), and only when there is none (a plain C++ host) from the loader's search path.
// ---------------------------------------------------------------------------------------------------------------------
struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclCommSplit) CommSplit = nullptr;           // optional (NCCL >= 2.18): the second lane's communicator
  decltype(&ncclAllReduce) AllReduce = nullptr;           // optional: the column modules' sums / the broadcast of a C++ host (below)
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;           // optional: what the communicator itself says (mw_dycore_rccl_info)
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  std::string path;
  bool ok = false;
};
int find_mapped_rccl(struct dl_phdr_info *info, size_t, void *out) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) { *(std::string *)out = info->dlpi_name; return 1; }
  return 0;
}
RcclApi &rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string mapped;
    dl_iterate_phdr(find_mapped_rccl, &mapped);
    void *h = nullptr;
    if (!mapped.empty()) h = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD);
    if (!h) for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return;
#define MW_SYM(field, name) api.field = (decltype(api.field))dlsym(h, #name); if (!api.field) return;
    MW_SYM(GetUniqueId, ncclGetUniqueId) MW_SYM(CommInitRank, ncclCommInitRank) MW_SYM(CommDestroy, ncclCommDestroy)
    MW_SYM(GroupStart, ncclGroupStart) MW_SYM(GroupEnd, ncclGroupEnd) MW_SYM(Send, ncclSend) MW_SYM(Recv, ncclRecv)
    MW_SYM(GetErrorString, ncclGetErrorString) MW_SYM(GetVersion, ncclGetVersion)
#undef MW_SYM
    api.CommSplit = (decltype(api.CommSplit))dlsym(h, "ncclCommSplit");
    api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
    api.Broadcast = (decltype(api.Broadcast))dlsym(h, "ncclBroadcast");
    api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
    api.CommUserRank = (decltype(api.CommUserRank))dlsym(h, "ncclCommUserRank");
    Dl_info di;
    if (dladdr((void *)api.Send, &di) && di.dli_fname) api.path = di.dli_fname;
    api.ok = true;
  });
  return api;
}
#define MW_NEED_RCCL()                                                                                  \
  RcclApi &R = rccl_api();                                                                              \
  if (!R.ok) MW_FAIL("RCCL is not available: no librccl.so is mapped in this process and none could be loaded")

// Two LANES: the dycore's state and tracer pipelines (rk_stage_march in mw_dycore.hip) exchange their strips from two different
// streams, each hiding the other's transfer.  Each lane has its own side stream and event pair, so that an exchange only waits for
// the pack kernels of ITS pipeline and only its pipeline's unpack kernels wait for it.  A lane belongs to the first caller stream
// that uses it; a third stream shares lane 0.
// Communicators: by default both lanes use the ONE communicator of the handle -- RCCL then runs the two groups in the order they
// were issued, which is the same on every rank (the schedule is a function of the stage counter alone).  Option rccl_two_comms = 1 (process default: MW_RCCL_LANES=2x2) gives
// lane 1 its own communicator (split off lane 0's with ncclCommSplit: no second unique id in the ABI), so that the two transfers
// can also overlap each other; NCCL's documentation warns that kernels of two communicators may dead-lock each other if the
// device cannot hold both at once, and this path could not be run between distinct GPUs on the one-GPU development box, so it is
// opt-in (mw_rccl_selftest exercises both forms on one GPU).
struct RcclLane {
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr, owner = nullptr;
  bool owned = false;                                           // `owner` has been assigned
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
};
struct RcclCtx {
  RcclLane lane[2];
  int nlanes = 1;
  bool own_comm1 = false;                                       // lane 1 has a communicator of its own (option rccl_two_comms = 1)
  int peers[4], send_order[4], recv_order[4], active[4];        // mw_exchange_plan
  int inline_group = 1;                                         // 1: the group runs on the CALLER's stream (option rccl_inline; see rccl_exchange)
  int self_ranks = 0;                                           // > 0: the self-loop transport (mw_dycore_use_rccl_self): this rank stands for that many identical blocks
  unsigned long long fuzz = 0;                                  // != 0: state of the delay fuzz (option xchg_fuzz = seed)
};
// xorshift64*: the next delay in microseconds, 0 .. 255 (about the duration of the kernels an exchange runs beside)
long long fuzz_usec(RcclCtx *c) {
  unsigned long long x = c->fuzz;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27; c->fuzz = x;
  return (long long)(((x * 0x2545F4914F6CDD1DULL) >> 40) & 0xff);
}
// Process default of the lanes: MW_RCCL_LANES = "1" (one side stream), "2" (two, one communicator: the default), "2x2" (two, a communicator
// each).  A handle's options rccl_lanes / rccl_two_comms override it for that handle (mw_dycore_use_rccl reads them).
void default_lanes(int &lanes, int &two_comms) {
  static int dl = 0, dt = 0;
  static std::once_flag once;
  std::call_once(once, [] { const char *e = getenv("MW_RCCL_LANES"); dl = (e && e[0] == '1') ? 1 : 2; dt = (e && !strcmp(e, "2x2")) ? 1 : 0; });
  lanes = dl; two_comms = dt;
}

#define MW_NCCL(call)                                                                                   \
  do { ncclResult_t r__ = (call);                                                                       \
       if (r__ != ncclSuccess) { mw::set_error(std::string(#call) + " failed: " + R.GetErrorString(r__)); return 1; } } while (0)

void free_ctx(RcclCtx *c) {
  if (!c) return;
  RcclApi &R = rccl_api();
  for (int l = 1; l >= 0; l--) {                                // the split communicator before its parent
    RcclLane &L = c->lane[l];
    if (L.side) (void)hipStreamSynchronize(L.side);
    if (L.comm && R.ok && (l == 0 || c->own_comm1)) (void)R.CommDestroy(L.comm);
    if (L.ev_ready) (void)hipEventDestroy(L.ev_ready);
    if (L.ev_done) (void)hipEventDestroy(L.ev_done);
    if (L.side) (void)hipStreamDestroy(L.side);
  }
  delete c;
}
// streams and events of the lanes; lane 1 only when a second communicator can be split off
int init_lanes(RcclCtx *c, RcclApi &R, int nranks, int myrank, int lanes, int two_comms, int high_prio = 1) {
  c->nlanes = lanes == 1 ? 1 : 2;
  c->lane[1].comm = c->lane[0].comm; c->own_comm1 = false;     // shared communicator (default)
  { if (c->nlanes == 2 && two_comms) {
      // ncclCommSplit is a COLLECTIVE: a rank that quietly fell back to the shared communicator while its peers split would post its
      // sends / receives on another communicator than they do and the first exchange would hang.  Asked for explicitly, a second
      // communicator that cannot be had is therefore an error on this rank (the caller's ranks then fail together or not at all:
      // every rank resolves the same librccl and calls the same split).
      if (!R.CommSplit) { mw::set_error("rccl_two_comms = 1 but this librccl has no ncclCommSplit"); return 1; }
      ncclComm_t split = nullptr;
      ncclResult_t r = R.CommSplit(c->lane[0].comm, 0, myrank, &split, nullptr);
      if (r != ncclSuccess || !split) { mw::set_error(std::string("rccl_two_comms = 1: ncclCommSplit failed: ") + R.GetErrorString(r)); return 1; }
      c->lane[1].comm = split; c->own_comm1 = true;
    } }
  for (int l = 0; l < c->nlanes; l++) {
    RcclLane &L = c->lane[l];
    // The side stream gets the HIGHEST stream priority (round 5): the send / receive kernels of a group are a handful of workgroups that
    // poll each other's flags, launched beside compute kernels that fill every CU with two 250-VGPR workgroups -- a transfer workgroup only
    // becomes resident when a compute workgroup retires, and at equal priority the next compute workgroup competes for that slot while
    // the transfer's resident half spins (measured with the self-loop transport, tools/exchange_overhead.py).
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if ((high_prio ? hipStreamCreateWithPriority(&L.side, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&L.side, hipStreamNonBlocking)) != hipSuccess ||
        hipEventCreateWithFlags(&L.ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming) != hipSuccess) { mw::set_error("mw_rccl: stream/event creation failed"); return 1; }
  }
  (void)nranks;
  return 0;
}
RcclLane &lane_of(RcclCtx *c, hipStream_t caller) {
  for (int l = 0; l < c->nlanes; l++) if (c->lane[l].owned && c->lane[l].owner == caller) return c->lane[l];
  for (int l = 0; l < c->nlanes; l++) if (!c->lane[l].owned) { c->lane[l].owned = true; c->lane[l].owner = caller; return c->lane[l]; }
  return c->lane[0];
}

int rccl_exchange(void *vctx, const double *sW, const double *sE, const double *sS, const double *sN, double *rW, double *rE,
                  double *rS, double *rN, long long nWE, long long nSN, void *vstream) {
  RcclCtx *c = (RcclCtx *)vctx;
  MW_NEED_RCCL();
  hipStream_t main_stream = (hipStream_t)vstream;
  RcclLane &L = lane_of(c, main_stream);
  // Round 5: the group runs IN LINE on the caller's stream by default.  The caller's stream is never free to do anything else between
  // its pack and its unpack kernels -- in the pipelined schedule it is the handle's exchange stream, in the two-stream schedule a
  // pipeline that needs the strips next -- so a side stream of the transport's own bought no overlap, and each of its two event
  // hand-overs between hardware queues cost 50-90 us on a chip that is full of stencil workgroups (rocprofv3 timeline of the self-loop
  // transport, DESIGN.md 0d: 2 x 3 hops per RK stage on the critical chain).  rccl_inline = 0 keeps the side stream (A/B).
  hipStream_t gs = c->inline_group ? main_stream : L.side;
  if (!c->inline_group) {
    MW_HIP(hipEventRecord(L.ev_ready, main_stream));         // pack kernels done
    MW_HIP(hipStreamWaitEvent(L.side, L.ev_ready, 0));
  }
  if (c->fuzz && mw::launch_spin(fuzz_usec(c), gs)) return 1;            // (test aid: the strips leave late)
  const double *sbuf[4] = {sW, sE, sS, sN};
  double *rbuf[4] = {rW, rE, rS, rN};
  const long long cnt[4] = {nWE, nWE, nSN, nSN};
  MW_NCCL(R.GroupStart());
  for (int o = 0; o < 4; o++) { int dir = c->send_order[o];
    if (c->active[dir] && cnt[dir] > 0) MW_NCCL(R.Send(sbuf[dir], (size_t)cnt[dir], ncclDouble, c->peers[dir], L.comm, gs)); }
  for (int o = 0; o < 4; o++) { int dir = c->recv_order[o];
    if (c->active[dir] && cnt[dir] > 0) MW_NCCL(R.Recv(rbuf[dir], (size_t)cnt[dir], ncclDouble, c->peers[dir], L.comm, gs)); }
  MW_NCCL(R.GroupEnd());
  if (c->fuzz && mw::launch_spin(fuzz_usec(c), gs)) return 1;            // (... and are reported complete late)
  if (!c->inline_group) {
    MW_HIP(hipEventRecord(L.ev_done, L.side));
    MW_HIP(hipStreamWaitEvent(main_stream, L.ev_done, 0));   // unpack kernels wait for the strips
  }
  return 0;
}
} // namespace

extern "C" {

int mw_rccl_unique_id(unsigned char *id128) {
  if (!id128) MW_FAIL("null id buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  MW_NEED_RCCL();
  ncclUniqueId id;
  MW_NCCL(R.GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return 0;
}

// Which RCCL the entry points were resolved from (diagnostic: a PyTorch host must see torch's own librccl here), and its version.
const char *mw_rccl_library_path(int *version) {
  RcclApi &R = rccl_api();
  if (version) { *version = 0; if (R.ok) (void)R.GetVersion(version); }
  return R.ok ? R.path.c_str() : "";
}

// lanes of a handle's transport: its options, else the process default
static void handle_lanes(mw_dycore_t h, int &lanes, int &two) {
  default_lanes(lanes, two);
  const int ol = mw::dycore_option(h, "rccl_lanes"), ot = mw::dycore_option(h, "rccl_two_comms");
  if (ol > 0) lanes = ol;
  if (ot >= 0) two = ot;
}
int mw_dycore_use_rccl(mw_dycore_t h, const unsigned char *id128, int nranks, int myrank) {
  if (!h || !id128) MW_FAIL("null argument");
  MW_NEED_RCCL();
  mw_grid_t g;
  if (mw_dycore_get_grid(h, &g)) return 1;
  if (nranks != g.nproc_x * g.nproc_y) MW_FAIL("nranks does not match the handle's rank grid");
  if (myrank < 0 || myrank >= nranks) MW_FAIL("myrank out of range");
  RcclCtx *c = new RcclCtx();
  auto fail = [&]() { free_ctx(c); return 1; };              // (the error text has been set by the failing call)
  ncclUniqueId id; memcpy(&id, id128, 128);
  { ncclResult_t r = R.CommInitRank(&c->lane[0].comm, nranks, id, myrank);
    if (r != ncclSuccess) { c->lane[0].comm = nullptr; mw::set_error(std::string("ncclCommInitRank failed: ") + R.GetErrorString(r)); return fail(); } }
  int lanes, two; handle_lanes(h, lanes, two);
  if (init_lanes(c, R, nranks, myrank, lanes, two, mw::dycore_option(h, "rccl_prio"))) return fail();
  if (mw_exchange_plan(&g, c->peers, c->send_order, c->recv_order, c->active)) return fail();
  if (int seed = mw::dycore_option(h, "xchg_fuzz")) c->fuzz = 0x9E3779B97F4A7C15ULL * (unsigned long long)seed + (unsigned long long)myrank + 1;
  c->inline_group = mw::dycore_option(h, "rccl_inline");
  if (mw::dycore_set_exchange_owned(h, rccl_exchange, c, [](void *p) { free_ctx((RcclCtx *)p); })) return fail();   // the handle frees it
  return 0;
}

// Test transport (no reference counterpart): ONE rank plays every rank of the handle's rank grid.  The handle's grid says nproc_x x nproc_y
// (mw_decompose), the communicator has one rank, and every active direction's peer is this rank itself -- sends W,E,S,N matched in FIFO
// order by receives E,W,N,S, exactly the messages a block exchanges with a neighbour that holds the same data.  That is the situation of
// a periodic domain tiled from copies of one block: this rank's result must then equal the one-rank run of that block bit for bit, and
// the whole asynchronous machinery -- pack kernels, ncclGroup of sends and receives on the side stream, event pairs, the pipelined /
// two-stream schedules with exchanges in flight beside compute -- runs for real on one GPU.  The handle's sums over ranks
// (mw_dycore_rccl_allreduce_sum) are multiplied by the number of blocks (exact: identical contributions), the broadcast is the identity.
int mw_dycore_use_rccl_self(mw_dycore_t h) {
  if (!h) MW_FAIL("null argument");
  MW_NEED_RCCL();
  mw_grid_t g;
  if (mw_dycore_get_grid(h, &g)) return 1;
  RcclCtx *c = new RcclCtx();
  auto fail = [&]() { free_ctx(c); return 1; };
  ncclUniqueId id;
  { ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) { mw::set_error(std::string("ncclGetUniqueId failed: ") + R.GetErrorString(r)); return fail(); } }
  { ncclResult_t r = R.CommInitRank(&c->lane[0].comm, 1, id, 0);
    if (r != ncclSuccess) { c->lane[0].comm = nullptr; mw::set_error(std::string("ncclCommInitRank failed: ") + R.GetErrorString(r)); return fail(); } }
  int lanes, two; handle_lanes(h, lanes, two);
  if (init_lanes(c, R, 1, 0, lanes, two, mw::dycore_option(h, "rccl_prio"))) return fail();
  if (mw_exchange_plan(&g, c->peers, c->send_order, c->recv_order, c->active)) return fail();   // (which directions exchange: the grid's)
  for (int d = 0; d < 4; d++) { c->peers[d] = 0; c->send_order[d] = d; }
  c->recv_order[0] = 1; c->recv_order[1] = 0; c->recv_order[2] = 3; c->recv_order[3] = 2;
  c->self_ranks = g.nproc_x * g.nproc_y;
  if (int seed = mw::dycore_option(h, "xchg_fuzz")) c->fuzz = 0x9E3779B97F4A7C15ULL * (unsigned long long)seed + 1;
  c->inline_group = mw::dycore_option(h, "rccl_inline");
  if (mw::dycore_set_exchange_owned(h, rccl_exchange, c, [](void *p) { free_ctx((RcclCtx *)p); })) return fail();
  return 0;
}

// What the installed RCCL transport's communicator says about itself (ncclCommCount / ncclCommUserRank): the evidence a multi-GPU
// run prints that RCCL really connected N ranks.  *lanes: side streams in use.  Fails when the handle's transport is not this one.
int mw_dycore_rccl_info(mw_dycore_t h, int *comm_ranks, int *comm_rank, int *lanes) {
  if (!h) MW_FAIL("null handle");
  MW_NEED_RCCL();
  mw_exchange_fn fn = nullptr;
  RcclCtx *c = (RcclCtx *)mw::dycore_exchange_ctx(h, &fn);
  if (fn != rccl_exchange || !c) MW_FAIL("mw_dycore_rccl_info: the handle's halo-exchange transport is not the built-in RCCL one");
  int n = -1, r = -1;
  if (R.CommCount) MW_NCCL(R.CommCount(c->lane[0].comm, &n));
  if (R.CommUserRank) MW_NCCL(R.CommUserRank(c->lane[0].comm, &r));
  if (comm_ranks) *comm_ranks = n;
  if (comm_rank) *comm_rank = r;
  if (lanes) *lanes = c->nlanes;
  return 0;
}

// The two collectives a C++ host of a decomposed run needs besides the halo exchange, on the handle's own communicator (a PyTorch host
// uses torch.distributed for them): MPI_Allreduce(SUM) of sponge_layer / ColumnNudger (sponge_layer.h:53-63, column_nudging.h:89-99) --
// the signature IS mw_allreduce_fn with ctx = the dycore handle -- and the MPI_Bcast from the main rank of Horizontal_Sponge::init
// (horizontal_sponge.h:72-77).  Ordered on `stream` like any other work of the caller.
static RcclCtx *own_ctx(mw_dycore_t h) {
  mw_exchange_fn fn = nullptr;
  RcclCtx *c = (RcclCtx *)mw::dycore_exchange_ctx(h, &fn);
  return (fn == rccl_exchange) ? c : nullptr;
}
int mw_dycore_rccl_allreduce_sum(void *handle, double *buf, long long n, void *stream) {
  if (!handle || !buf || n < 1) MW_FAIL("mw_dycore_rccl_allreduce_sum: bad argument");
  MW_NEED_RCCL();
  RcclCtx *c = own_ctx((mw_dycore_t)handle);
  if (!c) MW_FAIL("mw_dycore_rccl_allreduce_sum: the handle has no built-in RCCL transport (mw_dycore_use_rccl)");
  if (!R.AllReduce) MW_FAIL("this librccl has no ncclAllReduce");
  MW_NCCL(R.AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, c->lane[0].comm, (hipStream_t)stream));
  if (c->self_ranks > 1 && mw::launch_scale(buf, n, (double)c->self_ranks, (hipStream_t)stream)) return 1;   // (self-loop test transport: identical blocks)
  return 0;
}
int mw_dycore_rccl_bcast(mw_dycore_t h, double *buf, long long n, int root, void *stream) {
  if (!h || !buf || n < 1) MW_FAIL("mw_dycore_rccl_bcast: bad argument");
  MW_NEED_RCCL();
  RcclCtx *c = own_ctx(h);
  if (!c) MW_FAIL("mw_dycore_rccl_bcast: the handle has no built-in RCCL transport (mw_dycore_use_rccl)");
  if (!R.Broadcast) MW_FAIL("this librccl has no ncclBroadcast");
  MW_NCCL(R.Broadcast(buf, buf, (size_t)n, ncclDouble, root, c->lane[0].comm, (hipStream_t)stream));
  return 0;
}

static int g_selftest_lanes = 0, g_selftest_comms = 0;
static int g_selftest_cfg[2] = {0, 0};                         // lanes / communicator per lane of the next self-test (0: the process default)
// lanes = 1 | 2, two_comms = 0 | 1 for the following mw_rccl_selftest calls of this process; lanes = 0: back to the process default
int mw_rccl_selftest_config(int lanes, int two_comms) {
  if (lanes < 0 || lanes > 2 || (two_comms != 0 && two_comms != 1)) MW_FAIL("mw_rccl_selftest_config: lanes must be 0, 1 or 2 and two_comms 0 or 1");
  g_selftest_cfg[0] = lanes; g_selftest_cfg[1] = two_comms;
  return 0;
}
// How many lanes (side stream + event pair) the last mw_rccl_selftest drove, times 10, plus the number of communicators behind them:
// 21 = two lanes on the handle's one communicator (default), 22 = two lanes with a communicator each (rccl_two_comms = 1).
int mw_rccl_selftest_lanes(void) { return g_selftest_lanes * 10 + g_selftest_comms; }

// Diagnostic: a 1-rank communicator that sends n doubles to itself through the same group/stream/event sequence as
// rccl_exchange (two sends + two receives in one ncclGroup on a side stream).  Checks, on a single GPU, that RCCL initialises
// on the box and that the ordering against the caller's stream holds.  Returns 0 when the received data equal the sent data.
int mw_rccl_selftest(long long n, void *vstream) {
  if (n < 1) MW_FAIL("rccl_selftest: n must be >= 1");
  MW_NEED_RCCL();
  hipStream_t main_stream = (hipStream_t)vstream;
  ncclUniqueId id;
  MW_NCCL(R.GetUniqueId(&id));
  RcclCtx *c = new RcclCtx();
  auto fail = [&]() { free_ctx(c); return 1; };
  { ncclResult_t r = R.CommInitRank(&c->lane[0].comm, 1, id, 0);
    if (r != ncclSuccess) { c->lane[0].comm = nullptr; mw::set_error(std::string("ncclCommInitRank failed: ") + R.GetErrorString(r)); return fail(); } }
  { int lanes, two; default_lanes(lanes, two); if (g_selftest_cfg[0] > 0) { lanes = g_selftest_cfg[0]; two = g_selftest_cfg[1]; }
    if (init_lanes(c, R, 1, 0, lanes, two)) return fail(); }
  c->inline_group = 0;                                           // (this diagnostic drives the side-stream form: lanes, event pairs; the in-line
                                                                 //  form runs in every self-loop time-step test)
  for (int d = 0; d < 4; d++) { c->peers[d] = 0; c->send_order[d] = d; c->active[d] = 1; }
  c->recv_order[0] = 1; c->recv_order[1] = 0; c->recv_order[2] = 3; c->recv_order[3] = 2;      // E,W,N,S like mw_exchange_plan
  std::vector<double> h((size_t)4 * n), back((size_t)4 * n, -1.0);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1.0 + (double)i * 0.5;
  double *src = nullptr, *dst = nullptr;
  if (hipMalloc(&src, h.size() * 8) != hipSuccess || hipMalloc(&dst, h.size() * 8) != hipSuccess) {
    if (src) (void)hipFree(src);
    mw::set_error("rccl_selftest: hipMalloc failed"); return fail(); }
  int rc = 0;
  if (hipMemcpyAsync(src, h.data(), h.size() * 8, hipMemcpyHostToDevice, main_stream) != hipSuccess ||
      hipMemsetAsync(dst, 0, h.size() * 8, main_stream) != hipSuccess) { mw::set_error("rccl_selftest: upload failed"); rc = 1; }
  // strips W,E (n each) and S,N (n each); my E halo = the "peer's" W strip etc.
  if (!rc) rc = rccl_exchange(c, src, src + n, src + 2 * n, src + 3 * n, dst, dst + n, dst + 2 * n, dst + 3 * n, n, n, main_stream);
  // the second lane (the tracer pipeline's): the same exchange from a second caller stream, in flight together with the first
  std::vector<double> back2((size_t)4 * n, -1.0);
  double *dst2 = nullptr; hipStream_t s2 = nullptr; hipEvent_t up = nullptr;
  if (!rc && (hipMalloc(&dst2, h.size() * 8) != hipSuccess || hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess ||
              hipEventCreateWithFlags(&up, hipEventDisableTiming) != hipSuccess)) { mw::set_error("rccl_selftest: second lane set-up failed"); rc = 1; }
  if (!rc && (hipEventRecord(up, main_stream) != hipSuccess || hipStreamWaitEvent(s2, up, 0) != hipSuccess ||      // src is uploaded
              hipMemsetAsync(dst2, 0, h.size() * 8, s2) != hipSuccess)) { mw::set_error("rccl_selftest: second lane set-up failed"); rc = 1; }
  if (!rc) rc = rccl_exchange(c, src, src + n, src + 2 * n, src + 3 * n, dst2, dst2 + n, dst2 + 2 * n, dst2 + 3 * n, n, n, s2);
  g_selftest_lanes = (c->lane[0].owned ? 1 : 0) + (c->nlanes > 1 && c->lane[1].owned ? 1 : 0);
  g_selftest_comms = c->own_comm1 ? 2 : 1;
  if (!rc && (hipMemcpyAsync(back.data(), dst, h.size() * 8, hipMemcpyDeviceToHost, main_stream) != hipSuccess ||
              hipStreamSynchronize(main_stream) != hipSuccess)) { mw::set_error("rccl_selftest: download failed"); rc = 1; }
  if (!rc && (hipMemcpyAsync(back2.data(), dst2, h.size() * 8, hipMemcpyDeviceToHost, s2) != hipSuccess ||
              hipStreamSynchronize(s2) != hipSuccess)) { mw::set_error("rccl_selftest: download (second lane) failed"); rc = 1; }
  (void)hipFree(src); (void)hipFree(dst); if (dst2) (void)hipFree(dst2);
  free_ctx(c);
  if (s2) (void)hipStreamDestroy(s2);
  if (up) (void)hipEventDestroy(up);
  if (rc) return 1;
  // receives were posted E,W,N,S against sends W,E,S,N: rE <- sW, rW <- sE, rN <- sS, rS <- sN
  const int from[4] = {1, 0, 3, 2};                                      // dst strip d holds src strip from[d]
  for (int d = 0; d < 4; d++) for (long long i = 0; i < n; i++) {
    if (back[(size_t)d * n + i] != h[(size_t)from[d] * n + i]) MW_FAIL("rccl_selftest: received data differ from the sent data");
    if (back2[(size_t)d * n + i] != h[(size_t)from[d] * n + i]) MW_FAIL("rccl_selftest: received data differ from the sent data (second lane)");
  }
  return 0;
}

} // extern "C"
