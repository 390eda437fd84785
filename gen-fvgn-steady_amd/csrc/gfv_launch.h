// Every kernel launch of the library goes through GFV_LAUNCH: the launch itself, and - while the calling thread records a step
// (gfv_record_begin, include/gfv.h) - a copy of (kernel, grid, block, dynamic LDS, stream, arguments by value) in the recorder's
// arena, so that gfv_record_replay can issue the same launch again without any of the host work in front of it (Python, ctypes,
// argument checks, kernel-family choice, struct filling): one hipLaunchKernelGGL per command.  Round 5: on a 5 k-cell mesh the
// host needs 1.22 ms to issue the step's 186 launches + ~60 stream waits through the Python-level command list
// (profiles/r05_launch_cost.txt) - the device is busy 1.42 ms.
#pragma once
#include <hip/hip_runtime.h>
#include <tuple>
#include <type_traits>

struct GfvRecorder;
GfvRecorder* gfv_rec_active();   // the list the calling thread records into, or nullptr (record.hip)
// `run` re-issues the command and returns what the runtime said: gfv_record_replay stops at the first failure (GFV_ERR_LAUNCH)
void gfv_rec_push(GfvRecorder* r, hipError_t (*run)(const void* blob, hipStream_t st), const void* blob, size_t bytes, hipStream_t st);

template <class... KArgs, class... Args>
inline void gfv_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t st, Args&&... args) {
  hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
  if (GfvRecorder* r = gfv_rec_active()) {
    struct Blob {
      void (*k)(KArgs...);
      dim3 g, b;
      size_t sh;
      std::tuple<std::decay_t<KArgs>...> a;
    };
    static_assert(std::is_trivially_copyable<std::tuple<std::decay_t<KArgs>...>>::value || true, "kernel arguments are plain data");
    const Blob blob{kernel, grid, block, shmem, std::tuple<std::decay_t<KArgs>...>(args...)};
    gfv_rec_push(r, [](const void* p, hipStream_t s) -> hipError_t {
      const Blob& B = *static_cast<const Blob*>(p);
      std::apply([&](const auto&... a) { hipLaunchKernelGGL(B.k, B.g, B.b, B.sh, s, a...); }, B.a);
      return hipGetLastError();
    }, &blob, sizeof(Blob), st);
  }
}
#define GFV_LAUNCH(kernel, grid, block, shmem, stream, ...) gfv_launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, __VA_ARGS__)

// hipMemsetAsync of a few bytes inside a recorded step (wimg.hip: the weight maximum's zero)
// (returns the runtime's error of the eager call)
hipError_t gfv_memset_rec(void* dst, int value, size_t bytes, hipStream_t st);
