// The one collective of the path: SUM all-reduce of the fp64 partial sums of a sharded
// objective evaluation -- [sum r^2/v, sum log v, sum r^2, n, sum pseudo-Huber, sum y^T K^-1 y ...] -- over RCCL.
// Replaces the reference's comm_world.allreduce(..., op=MPI.SUM) of _src/optimize/loss/mpi.py:23-24,57 and
// _src/optimize/scale/mpi.py:35-36 (three host all-reduces per evaluation there; one device all-reduce here).
//
// The communicator is the caller's (an ncclComm_t handed over as void*): the library owns no process group.  RCCL is
// opened with dlopen on first use, so the library itself has no link-time dependency on it and a single-GPU
// process never loads it.  MUYGPYS_HIP_RCCL names another librccl (e.g. the copy a PyTorch wheel ships).
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

#include "mgp_args.h"

namespace mgp {
namespace {

// ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t):
// the ABI of rccl.h (ncclFloat64 = 8, ncclSum = 0; ncclSuccess = 0), spelt out so that building the library needs
// no RCCL headers
using AllReduceFn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

AllReduceFn rccl_allreduce() {
  static AllReduceFn fn = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("MUYGPYS_HIP_RCCL");
    void* lib = nullptr;
    // a copy already loaded by the process (PyTorch's) first: one RCCL per process
    for (const char* name : {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      if (!name || !*name) continue;
      lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib)
      for (const char* name : {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        if (!name || !*name) continue;
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
      }
    if (lib) fn = reinterpret_cast<AllReduceFn>(dlsym(lib, "ncclAllReduce"));
  });
  return fn;
}

}  // namespace

int allreduce_partials(double* partials_dev, int count, void* comm, hipStream_t stream) {
  if (!partials_dev || count < 1 || !comm) return MGP_EINVAL;
  AllReduceFn fn = rccl_allreduce();
  if (!fn) return MGP_EUNSUPPORTED;  // no RCCL on this machine
  const int rc = fn(partials_dev, partials_dev, (size_t)count, kNcclFloat64, kNcclSum, comm, stream);
  return rc == 0 ? MGP_OK : -(2000 + rc);
}

}  // namespace mgp
