// =====================================================================================================
// mw_rccl.cpp -- halo exchange over RCCL point-to-point (xGMI) for the slab-decomposed dycore.
// Replaces the MPI_Isend/Irecv/Waitall pattern of halo_exchange
// (reference: model/modules/dynamics_euler_stratified_wenofv.h:641-723; neighbour matrix coupler.h:169-179).
//
// One ncclGroup per exchange on a dedicated side stream: up to four sends + four receives, each peer reached
// over its own xGMI link.  Message matching between one pair of ranks is FIFO, so when the west and east
// (or south and north) neighbour are the same rank (2 ranks in that direction) sends are posted W,E,S,N and
// receives E,W,N,S: the peer's first send (its W strip) is my E halo.
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include <rccl/rccl.h>
#include <cstring>
#include <vector>

namespace {
struct RcclCtx {
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  int peers[4], send_order[4], recv_order[4], active[4];        // mw_exchange_plan
};

#define MW_NCCL(call)                                                                                   \
  do { ncclResult_t r__ = (call);                                                                       \
       if (r__ != ncclSuccess) { mw::set_error(std::string(#call) + " failed: " + ncclGetErrorString(r__)); return 1; } } while (0)

int rccl_exchange(void *vctx, const double *sW, const double *sE, const double *sS, const double *sN, double *rW, double *rE,
                  double *rS, double *rN, long long nWE, long long nSN, void *vstream) {
  RcclCtx *c = (RcclCtx *)vctx;
  hipStream_t main_stream = (hipStream_t)vstream;
  MW_HIP(hipEventRecord(c->ev_ready, main_stream));          // pack kernels done
  MW_HIP(hipStreamWaitEvent(c->side, c->ev_ready, 0));
  const double *sbuf[4] = {sW, sE, sS, sN};
  double *rbuf[4] = {rW, rE, rS, rN};
  const long long cnt[4] = {nWE, nWE, nSN, nSN};
  MW_NCCL(ncclGroupStart());
  for (int o = 0; o < 4; o++) { int dir = c->send_order[o];
    if (c->active[dir] && cnt[dir] > 0) MW_NCCL(ncclSend(sbuf[dir], (size_t)cnt[dir], ncclDouble, c->peers[dir], c->comm, c->side)); }
  for (int o = 0; o < 4; o++) { int dir = c->recv_order[o];
    if (c->active[dir] && cnt[dir] > 0) MW_NCCL(ncclRecv(rbuf[dir], (size_t)cnt[dir], ncclDouble, c->peers[dir], c->comm, c->side)); }
  MW_NCCL(ncclGroupEnd());
  MW_HIP(hipEventRecord(c->ev_done, c->side));
  MW_HIP(hipStreamWaitEvent(main_stream, c->ev_done, 0));    // unpack kernels wait for the strips
  return 0;
}
} // namespace

extern "C" {

int mw_rccl_unique_id(unsigned char *id128) {
  if (!id128) MW_FAIL("null id buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  MW_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, 128);
  return 0;
}

int mw_dycore_use_rccl(mw_dycore_t h, const unsigned char *id128, int nranks, int myrank) {
  if (!h || !id128) MW_FAIL("null argument");
  mw_grid_t g;
  if (mw_dycore_get_grid(h, &g)) return 1;
  if (nranks != g.nproc_x * g.nproc_y) MW_FAIL("nranks does not match the handle's rank grid");
  RcclCtx *c = new RcclCtx();
  ncclUniqueId id; memcpy(&id, id128, 128);
  MW_NCCL(ncclCommInitRank(&c->comm, nranks, id, myrank));
  MW_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  MW_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  MW_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  if (mw_exchange_plan(&g, c->peers, c->send_order, c->recv_order, c->active)) return 1;
  return mw_dycore_set_exchange(h, rccl_exchange, c);
}

// Diagnostic: a 1-rank communicator that sends n doubles to itself through the same group/stream/event sequence as
// rccl_exchange (two sends + two receives in one ncclGroup on a side stream).  Checks, on a single GPU, that RCCL initialises
// on the box and that the ordering against the caller's stream holds.  Returns 0 when the received data equal the sent data.
int mw_rccl_selftest(long long n, void *vstream) {
  if (n < 1) MW_FAIL("rccl_selftest: n must be >= 1");
  hipStream_t main_stream = (hipStream_t)vstream;
  ncclUniqueId id;
  MW_NCCL(ncclGetUniqueId(&id));
  RcclCtx c;
  MW_NCCL(ncclCommInitRank(&c.comm, 1, id, 0));
  MW_HIP(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
  MW_HIP(hipEventCreateWithFlags(&c.ev_ready, hipEventDisableTiming));
  MW_HIP(hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming));
  for (int d = 0; d < 4; d++) { c.peers[d] = 0; c.send_order[d] = d; c.active[d] = 1; }
  c.recv_order[0] = 1; c.recv_order[1] = 0; c.recv_order[2] = 3; c.recv_order[3] = 2;      // E,W,N,S like mw_exchange_plan
  std::vector<double> h((size_t)4 * n), back((size_t)4 * n, -1.0);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1.0 + (double)i * 0.5;
  double *src = nullptr, *dst = nullptr;
  MW_HIP(hipMalloc(&src, h.size() * 8)); MW_HIP(hipMalloc(&dst, h.size() * 8));
  MW_HIP(hipMemcpyAsync(src, h.data(), h.size() * 8, hipMemcpyHostToDevice, main_stream));
  MW_HIP(hipMemsetAsync(dst, 0, h.size() * 8, main_stream));
  // strips W,E (n each) and S,N (n each); my E halo = the "peer's" W strip etc.
  int rc = rccl_exchange(&c, src, src + n, src + 2 * n, src + 3 * n, dst, dst + n, dst + 2 * n, dst + 3 * n, n, n, main_stream);
  if (!rc) { MW_HIP(hipMemcpyAsync(back.data(), dst, h.size() * 8, hipMemcpyDeviceToHost, main_stream)); MW_HIP(hipStreamSynchronize(main_stream)); }
  (void)hipFree(src); (void)hipFree(dst);
  (void)hipEventDestroy(c.ev_ready); (void)hipEventDestroy(c.ev_done); (void)hipStreamDestroy(c.side);
  (void)ncclCommDestroy(c.comm);
  if (rc) return 1;
  // receives were posted E,W,N,S against sends W,E,S,N: rE <- sW, rW <- sE, rN <- sS, rS <- sN
  const int from[4] = {1, 0, 3, 2};                                      // dst strip d holds src strip from[d]
  for (int d = 0; d < 4; d++) for (long long i = 0; i < n; i++)
    if (back[(size_t)d * n + i] != h[(size_t)from[d] * n + i]) MW_FAIL("rccl_selftest: received data differ from the sent data");
  return 0;
}

} // extern "C"
