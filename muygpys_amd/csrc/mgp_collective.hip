// The one collective of the path: SUM all-reduce of the fp64 partial sums of a sharded
// objective evaluation -- [sum r^2/v, sum log v, sum r^2, n, sum pseudo-Huber, sum y^T K^-1 y ...] -- over RCCL.
// Replaces the reference's comm_world.allreduce(..., op=MPI.SUM) of _src/optimize/loss/mpi.py:23-24,57 and
// _src/optimize/scale/mpi.py:35-36 (three host all-reduces per evaluation there; one device all-reduce here).
//
// The communicator is the caller's (an ncclComm_t handed over as void*): the library owns no process group.  RCCL is
// opened with dlopen on first use, so the library itself has no link-time dependency on it and a single-GPU
// process never loads it.  MUYGPYS_HIP_RCCL names the librccl that made the communicator (e.g. the copy a PyTorch
// wheel ships: torch/lib/librccl.so) when it is not the one the process-wide symbol lookup finds.
#include <dlfcn.h>
#include <link.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mgp_args.h"

namespace mgp {
namespace {

// ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t):
// the ABI of rccl.h (ncclFloat64 = 8, ncclSum = 0; ncclSuccess = 0), spelt out so that building the library needs
// no RCCL headers
using AllReduceFn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

// Is some RCCL / NCCL image mapped into this process already (dl_iterate_phdr over the loaded objects)?
int rccl_mapped_cb(struct dl_phdr_info* info, size_t, void* found) {
  const char* n = info->dlpi_name;
  if (n && (strstr(n, "librccl") || strstr(n, "libnccl"))) *static_cast<int*>(found) = 1;
  return 0;
}

// A communicator is only valid inside the RCCL copy that made it, so the symbol must come from THAT copy:
// 1. MUYGPYS_HIP_RCCL (path of the library the caller made the communicator with), if set;
// 2. the process's global symbol scope (dlsym(RTLD_DEFAULT): a librccl the caller linked or loaded RTLD_GLOBAL);
// 3. an already-mapped copy found by name with RTLD_NOLOAD (PyTorch opens its bundled torch/lib/librccl.so by path);
// 4. only when NO RCCL image is mapped at all: load the system's.  If one is mapped that 1-3 cannot reach, loading a
//    second copy and handing it a foreign communicator would be undefined behaviour: MGP_EUNSUPPORTED instead.
AllReduceFn rccl_allreduce() {
  static AllReduceFn fn = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("MUYGPYS_HIP_RCCL");
    if (env && *env) {
      void* lib = dlopen(env, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (!lib) lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
      if (lib) fn = reinterpret_cast<AllReduceFn>(dlsym(lib, "ncclAllReduce"));
      return;  // (an explicit choice is never second-guessed)
    }
    fn = reinterpret_cast<AllReduceFn>(dlsym(RTLD_DEFAULT, "ncclAllReduce"));
    if (fn) return;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (void* lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL)) {
        fn = reinterpret_cast<AllReduceFn>(dlsym(lib, "ncclAllReduce"));
        if (fn) return;
      }
    }
    int mapped = 0;
    dl_iterate_phdr(rccl_mapped_cb, &mapped);
    if (mapped) return;  // some other copy owns the process's communicators
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      if (void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        fn = reinterpret_cast<AllReduceFn>(dlsym(lib, "ncclAllReduce"));
        if (fn) return;
      }
    }
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
