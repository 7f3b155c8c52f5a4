// comm.h -- what the multi-GPU translation units (sharded.hip, sharded_hash.hip) share: RCCL resolved at run time and
// the communicator record behind include/nbody_hip_comm.h.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened at run time (no link dependency)

#include <vector>

#include "common.h"
#include "nbody_hip_comm.h"

namespace nbh {

struct Rccl {
  void* so = nullptr;
  decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&::ncclCommInitRank) CommInitRank = nullptr;
  decltype(&::ncclCommInitAll) CommInitAll = nullptr;
  decltype(&::ncclCommDestroy) CommDestroy = nullptr;
  decltype(&::ncclAllGather) AllGather = nullptr;
  decltype(&::ncclAllReduce) AllReduce = nullptr;
  decltype(&::ncclSend) Send = nullptr;
  decltype(&::ncclRecv) Recv = nullptr;
  decltype(&::ncclGroupStart) GroupStart = nullptr;
  decltype(&::ncclGroupEnd) GroupEnd = nullptr;
  decltype(&::ncclGetErrorString) GetErrorString = nullptr;
};
Rccl* rccl_load();  // nullptr when librccl.so.1 cannot be opened (dlerror() says why)

#define NBH_NCCL(api, call)                                                                                      \
  do {                                                                                                           \
    ncclResult_t r_ = (call);                                                                                    \
    if (r_ != ncclSuccess)                                                                                       \
      return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "%s: %s", #call, (api)->GetErrorString(r_));         \
  } while (0)

}  // namespace nbh

struct nbody_hip_comm {
  int world = 0;
  int transport = NBODY_HIP_TRANSPORT_P2P;
  struct Member {
    int rank = 0, device = 0;
    ncclComm_t nccl = nullptr;
  };
  std::vector<Member> local;  // ranks living in this process, ascending
  bool all_local() const { return (int)local.size() == world; }
};
