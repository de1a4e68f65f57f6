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

} // extern "C"
