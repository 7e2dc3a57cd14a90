// Internal declarations shared by the librcv translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include "../../include/rcv.h"

// Experiment knobs (tile overrides, kernel-family switches) exist only in diagnostic builds (make EXPERIMENTS=1): the shipped
// library never reads the environment on its launch path.
#ifdef RCV_EXPERIMENTS
#define RCV_ENV(name) getenv(name)
#else
#define RCV_ENV(name) ((const char*)nullptr)
#endif

#define RCV_MAX_DEVICES 64

// Tiling plans are a pure function of the integer slots of an op record: computed once per distinct record and kept in the handle
// (a 160x120 step enqueues ~150 kernels; re-planning each of them every step was a measurable share of the host time).
struct rcv_plan_cache {
  std::mutex mu;
  std::unordered_map<std::string, std::string> map;   // key: kind + i[] (workspace slots zeroed); value: the plan struct, bytewise
};

struct rcv_handle {
  int device;    // -1: planning-only handle (rcv_create_planner): workspace / label queries work, nothing can be enqueued
  int num_cus;
  int max_lds;   // bytes of LDS one workgroup may use
  rcv_plan_cache* plans;
  // RCV_F_SIDE_STREAM: ops off the critical path (filter gradients) run on this stream, forked from / joined to the caller's
  // stream with events inside rcv_run; created on first use
  hipStream_t side_stream;
  hipEvent_t ev_fork[8];
  hipEvent_t ev_join;
  int ev_next;
};

void rcv_set_error(const char* fmt, ...);

#define RCV_CHECK_ARG(cond, ...)                         \
  do {                                                   \
    if (!(cond)) {                                       \
      rcv_set_error(__VA_ARGS__);                        \
      return RCV_E_ARG;                                  \
    }                                                    \
  } while (0)

#define RCV_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      rcv_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RCV_E_HIP;                                                                \
    }                                                                                  \
  } while (0)

// Exact n / d for 0 <= n < 65536, 1 <= d < 65536 (m = floor(2^32/d)+1; d == 1 handled apart).
struct FastDiv {
  uint32_t m, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.m = (d <= 1) ? 0u : (uint32_t)((0x100000000ull / d) + 1ull);
  return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t fd_div(uint32_t n, FastDiv f) {
  return f.d == 1 ? n : __umulhi(n, f.m);
}
// XCD-aware work mapping (MI355X: 8 XCDs, private 4 MiB L2 each; workgroups are dealt round-robin, so ids b and
// b+8 share an L2).  Maps dispatch id b to a logical id such that every XCD owns one CONTIGUOUS range of logical
// ids: workgroups that share operands (parity phases / channel tiles of one pixel tile, halo neighbours) then hit in
// the same L2.  Bijective for any n (cdna guide T1).  Affects speed only.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = b & 7, i = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

static inline std::string rcv_plan_key(const rcv_op* op) {
  int32_t k[1 + RCV_I__N];
  k[0] = op->kind | (int32_t)((op->flags & RCV_F_RESID) << 16) | (int32_t)((op->flags & RCV_F_MFMA_FP32) << 4);   // the flags a planner looks at
  memcpy(k + 1, op->i, sizeof(op->i));
  k[1 + RCV_I_NPART] = 0; k[1 + RCV_I_NSPLIT] = 0;    // outputs of the planning, not inputs
  return std::string(reinterpret_cast<const char*>(k), sizeof(k));
}
template <typename P>
static inline bool rcv_plan_get(const rcv_handle* h, const rcv_op* op, P* out) {
  if (!h->plans) return false;
  std::lock_guard<std::mutex> g(h->plans->mu);
  auto it = h->plans->map.find(rcv_plan_key(op));
  if (it == h->plans->map.end() || it->second.size() != sizeof(P)) return false;
  memcpy(out, it->second.data(), sizeof(P));
  return true;
}
template <typename P>
static inline void rcv_plan_put(const rcv_handle* h, const rcv_op* op, const P& pl) {
  if (!h->plans) return;
  std::lock_guard<std::mutex> g(h->plans->mu);
  h->plans->map[rcv_plan_key(op)] = std::string(reinterpret_cast<const char*>(&pl), sizeof(P));
}

// Raises the dynamic-LDS limit of one kernel on one device once (hipFuncSetAttribute is per device); `table` is a static
// size_t[RCV_MAX_DEVICES] next to the kernel's launch site.
#define RCV_ENSURE_LDS(kern, lds, dev, table)                                                                          \
  do {                                                                                                                 \
    const int d_ = ((dev) >= 0 && (dev) < RCV_MAX_DEVICES) ? (dev) : 0;                                                \
    if ((lds) > (table)[d_]) {                                                                                         \
      RCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds))); \
      (table)[d_] = (lds);                                                                                             \
    }                                                                                                                  \
  } while (0)

// ---- launchers implemented in the .hip files; each validates, picks a tiling and enqueues ----
// `query` != nullptr: do not launch, only fill tiling dependent outputs (n_part / n_split / bytes).
struct OpQuery {
  int n_part;
  int n_split;
  size_t part_bytes;
  char label[64];
};
int rcv_launch_conv(const rcv_handle* h, const rcv_op* op, hipStream_t s, OpQuery* query);
int rcv_launch_wgrad(const rcv_handle* h, const rcv_op* op, hipStream_t s, OpQuery* query);
int rcv_launch_small(const rcv_handle* h, const rcv_op* op, hipStream_t s, OpQuery* query);
