// C ABI of librcv.so (declared in include/rcv.h): handle, error text, op dispatch and the named
// convenience entry points.  Nothing here allocates device memory or synchronises the device.
#include <stdarg.h>
#include <string.h>
#include "rcv_internal.h"
#include "conv_common.h"

static thread_local char g_err[512] = "";

void rcv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Every entry point that creates streams / events or launches kernels makes the handle's device current for the duration of the
// call and restores the caller's device afterwards (a process that drives cuda:1 while cuda:0 is current would otherwise get its
// side stream, its events and its kernel launches on the wrong GPU).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define RCV_GUARD(h, what)                                                                                            \
  RCV_CHECK_ARG((h)->device >= 0, what ": this is a planning-only handle (rcv_create_planner); it cannot enqueue work"); \
  DeviceGuard guard_((h)->device);                                                                                    \
  if (guard_.err != hipSuccess) { rcv_set_error(what ": cannot make device %d current: %s", (h)->device, hipGetErrorString(guard_.err)); return RCV_E_HIP; }

extern "C" {

const char* rcv_last_error(void) { return g_err; }
int rcv_version(void) { return RCV_VERSION; }

int rcv_create(int device, rcv_handle** out) {
  RCV_CHECK_ARG(out != nullptr, "rcv_create: out is NULL");
  int count = 0;
  RCV_HIP(hipGetDeviceCount(&count));
  RCV_CHECK_ARG(device >= 0 && device < count, "rcv_create: device %d out of range (%d visible)", device, count);
  hipDeviceProp_t prop;
  RCV_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    rcv_set_error("rcv_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return RCV_E_UNSUPPORTED;
  }
  rcv_handle* h = new rcv_handle;
  h->device = device;
  h->num_cus = prop.multiProcessorCount;
  h->max_lds = 160 * 1024;
  h->plans = new rcv_plan_cache;
  h->side_stream = nullptr;
  h->ev_join = nullptr;
  h->ev_next = 0;
  for (auto& e : h->ev_fork) e = nullptr;
  *out = h;
  return RCV_OK;
}

int rcv_create_planner(int num_cus, rcv_handle** out) {
  RCV_CHECK_ARG(out != nullptr, "rcv_create_planner: out is NULL");
  RCV_CHECK_ARG(num_cus > 0 && num_cus <= 4096, "rcv_create_planner: num_cus %d out of range", num_cus);
  rcv_handle* h = new rcv_handle;
  h->device = -1;
  h->num_cus = num_cus;
  h->max_lds = 160 * 1024;
  h->plans = new rcv_plan_cache;
  h->side_stream = nullptr;
  h->ev_join = nullptr;
  h->ev_next = 0;
  for (auto& e : h->ev_fork) e = nullptr;
  *out = h;
  return RCV_OK;
}

int rcv_destroy(rcv_handle* h) {
  if (h && h->side_stream) {
    DeviceGuard guard(h->device);
    (void)hipStreamDestroy(h->side_stream);
    (void)hipEventDestroy(h->ev_join);
    for (auto& e : h->ev_fork) (void)hipEventDestroy(e);
  }
  if (h) delete h->plans;
  delete h;
  return RCV_OK;
}

int rcv_num_cus(const rcv_handle* h) { return h ? h->num_cus : 0; }

static int dispatch(const rcv_handle* h, const rcv_op* op, hipStream_t s, OpQuery* q) {
  switch (op->kind) {
    case RCV_OP_CONV:
    case RCV_OP_TCONV:
      return rcv_launch_conv(h, op, s, q);
    case RCV_OP_WGRAD:
    case RCV_OP_WGRAD_REDUCE:
    case RCV_OP_WGRAD_REDUCE_BATCH:
      return rcv_launch_wgrad(h, op, s, q);
    case RCV_OP_NOP:
      if (q) { snprintf(q->label, sizeof(q->label), "nop"); q->n_part = 0; q->n_split = 0; q->part_bytes = 0; }
      return RCV_OK;
    default:
      return rcv_launch_small(h, op, s, q);
  }
}

int rcv_op_workspace(const rcv_handle* h, rcv_op* op, size_t* part_bytes) {
  RCV_CHECK_ARG(h && op && part_bytes, "rcv_op_workspace: NULL argument");
  OpQuery q;
  memset(&q, 0, sizeof(q));
  const int rc = dispatch(h, op, nullptr, &q);
  if (rc) return rc;
  op->i[RCV_I_NPART] = q.n_part;
  if (op->kind == RCV_OP_WGRAD) op->i[RCV_I_NSPLIT] = q.n_split;
  *part_bytes = q.part_bytes;
  return RCV_OK;
}

static int run_ops(rcv_handle* h, const rcv_op* ops, int n, void* stream, bool join);

int rcv_run(rcv_handle* h, const rcv_op* ops, int n, void* stream) { return run_ops(h, ops, n, stream, true); }

int rcv_run_ex(rcv_handle* h, const rcv_op* ops, int n, void* stream, uint32_t run_flags) {
  return run_ops(h, ops, n, stream, !(run_flags & RCV_RUN_NO_JOIN));
}

int rcv_join_side(rcv_handle* h, void* stream) {
  RCV_CHECK_ARG(h, "rcv_join_side: NULL handle");
  if (!h->side_stream) return RCV_OK;         // nothing was ever forked
  RCV_GUARD(h, "rcv_join_side");
  RCV_HIP(hipEventRecord(h->ev_join, h->side_stream));
  RCV_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_join, 0));
  return RCV_OK;
}

}  // extern "C"

static int run_ops(rcv_handle* h, const rcv_op* ops, int n, void* stream, bool join) {
  RCV_CHECK_ARG(h && (ops || n == 0) && n >= 0, "rcv_run: bad arguments");
  RCV_GUARD(h, "rcv_run");
  hipStream_t s = (hipStream_t)stream;
  // Ops flagged RCV_F_SIDE_STREAM run on the handle's side stream: it is forked from `stream` in front of every run of such ops
  // (event record on `stream`, wait on the side stream) and joined back before rcv_run returns, so the caller keeps seeing ONE
  // stream-ordered call.  No host synchronisation; capturable (the side stream joins the capture through the events).
  bool prev_side = false, side_used = false;
  for (int k = 0; k < n; ++k) {
    if (ops[k].kind == RCV_OP_NOP) continue;    // (does not end a run of side-stream ops either: no extra fork around an empty slot)
    const bool side = (ops[k].flags & RCV_F_SIDE_STREAM) != 0;
    if (side && !h->side_stream) {
      RCV_HIP(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
      RCV_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
      for (auto& e : h->ev_fork) RCV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (side && !prev_side) {
      hipEvent_t e = h->ev_fork[h->ev_next];
      h->ev_next = (h->ev_next + 1) % 8;
      RCV_HIP(hipEventRecord(e, s));
      RCV_HIP(hipStreamWaitEvent(h->side_stream, e, 0));
      side_used = true;
    }
    prev_side = side;
    const int rc = dispatch(h, &ops[k], side ? h->side_stream : s, nullptr);
    if (rc) {
      char tmp[400];
      strncpy(tmp, g_err, sizeof(tmp) - 1);
      tmp[sizeof(tmp) - 1] = 0;
      rcv_set_error("op %d (kind %d): %s", k, ops[k].kind, tmp);
      if (side_used) { (void)hipEventRecord(h->ev_join, h->side_stream); (void)hipStreamWaitEvent(s, h->ev_join, 0); }
      return rc;
    }
  }
  if (side_used && join) {
    RCV_HIP(hipEventRecord(h->ev_join, h->side_stream));
    RCV_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
  }
  return RCV_OK;
}

extern "C" {

int rcv_op_filter_layout(const rcv_handle* h, const rcv_op* op, int force) {
  if (!h || !op) return 0;
  if (force == 0 && convn_bf3_wanted(h, op)) return op->kind == RCV_OP_TCONV ? 4 : 3;     // split-bf16 layouts, narrow layers (convn_bf3.hip)
  if (op->kind != RCV_OP_CONV) return 0;
  if (force == 0 && conv_bf3_wanted(h, op)) return op->i[RCV_I_STRIDE] == 2 ? 5 : 3;      // split-bf16 layout (conv_bf3.hip); force: the Winograd layout where it exists
  return conv_wino_wanted(h, op, force != 0) ? 2 : 0;
}

int rcv_op_kernel_label(const rcv_handle* h, const rcv_op* op, char* buf, int size) {
  RCV_CHECK_ARG(h && op && buf && size > 0, "rcv_op_kernel_label: bad arguments");
  OpQuery q;
  memset(&q, 0, sizeof(q));
  const int rc = dispatch(h, op, nullptr, &q);
  if (rc) return rc;
  strncpy(buf, q.label, size - 1);
  buf[size - 1] = 0;
  return RCV_OK;
}

int rcv_run_timed(rcv_handle* h, const rcv_op* ops, int n, void* stream, float* ms) {
  RCV_CHECK_ARG(h && ops && n > 0 && ms, "rcv_run_timed: bad arguments");
  RCV_GUARD(h, "rcv_run_timed");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t* ev = new hipEvent_t[n + 1];
  for (int k = 0; k <= n; ++k) {
    if (hipEventCreate(&ev[k]) != hipSuccess) { rcv_set_error("rcv_run_timed: hipEventCreate failed"); delete[] ev; return RCV_E_HIP; }
  }
  int rc = RCV_OK;
  (void)hipEventRecord(ev[0], s);
  int last = 0;                      // index of the event that closes the latest op that launched something
  int* opens = new int[n];           // event that opens op k (RCV_OP_NOP slots launch nothing and take no event: rcv_run skips them too)
  for (int k = 0; k < n && rc == RCV_OK; ++k) {
    opens[k] = last;
    if (ops[k].kind == RCV_OP_NOP) continue;
    rc = dispatch(h, &ops[k], s, nullptr);
    (void)hipEventRecord(ev[k + 1], s);
    last = k + 1;
  }
  if (hipStreamSynchronize(s) != hipSuccess && rc == RCV_OK) { rcv_set_error("rcv_run_timed: stream sync failed"); rc = RCV_E_HIP; }
  if (rc == RCV_OK)
    for (int k = 0; k < n; ++k) {
      if (ops[k].kind == RCV_OP_NOP) ms[k] = 0.f;
      else (void)hipEventElapsedTime(&ms[k], ev[opens[k]], ev[k + 1]);
    }
  delete[] opens;
  for (int k = 0; k <= n; ++k) (void)hipEventDestroy(ev[k]);
  delete[] ev;
  return rc;
}

// ------------------------------ named entry points ------------------------------
int rcv_conv3x3(rcv_handle* h, const rcv_op* op, void* stream) {
  RCV_CHECK_ARG(op && op->kind == RCV_OP_CONV, "rcv_conv3x3: record kind must be RCV_OP_CONV");
  return rcv_run(h, op, 1, stream);
}
int rcv_convT3x3s2(rcv_handle* h, const rcv_op* op, void* stream) {
  RCV_CHECK_ARG(op && op->kind == RCV_OP_TCONV, "rcv_convT3x3s2: record kind must be RCV_OP_TCONV");
  return rcv_run(h, op, 1, stream);
}
int rcv_wgrad3x3(rcv_handle* h, const rcv_op* op, void* stream) {
  RCV_CHECK_ARG(op && op->kind == RCV_OP_WGRAD, "rcv_wgrad3x3: record kind must be RCV_OP_WGRAD");
  return rcv_run(h, op, 1, stream);
}

int rcv_bn_finalize(rcv_handle* h, const float* part, int n_part, int C, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, int training, float* consts,
                    float* save_mean, float* save_istd, void* stream) {
  RCV_CHECK_ARG(count == (double)(long long)count && count < 2147483647.0, "rcv_bn_finalize: count must be an integer < 2^31");
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_BN_FINALIZE;
  op.flags = training ? RCV_F_TRAINING : 0;
  op.i[RCV_I_N] = (int)count; op.i[RCV_I_HO] = 1; op.i[RCV_I_WO] = 1;
  op.i[RCV_I_COUT] = C; op.i[RCV_I_NPART] = n_part;
  op.f[0] = momentum; op.f[1] = eps;
  op.p[RCV_P_PART] = (void*)part; op.p[RCV_P_OUT] = consts;
  op.p[RCV_P_X0] = (void*)gamma; op.p[RCV_P_X1] = (void*)beta; op.p[RCV_P_X2] = running_mean; op.p[RCV_P_X3] = running_var;
  op.p[RCV_P_X4] = save_mean; op.p[RCV_P_X5] = save_istd;
  return rcv_run(h, &op, 1, stream);
}

int rcv_bn_backward(rcv_handle* h, const float* part, int n_part, int C, double count, const float* gamma, const float* save_mean,
                    const float* save_istd, const float* fwd_consts, int decoder, float* consts, float* dgamma, float* dbeta,
                    void* stream) {
  (void)decoder;
  RCV_CHECK_ARG(count == (double)(long long)count && count < 2147483647.0, "rcv_bn_backward: count must be an integer < 2^31");
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_BN_BWD;
  op.i[RCV_I_N] = (int)count; op.i[RCV_I_HO] = 1; op.i[RCV_I_WO] = 1;
  op.i[RCV_I_COUT] = C; op.i[RCV_I_NPART] = n_part;
  op.p[RCV_P_PART] = (void*)part; op.p[RCV_P_OUT] = consts; op.p[RCV_P_IN_C] = (void*)fwd_consts;
  op.p[RCV_P_X0] = (void*)gamma; op.p[RCV_P_X1] = dgamma; op.p[RCV_P_X2] = dbeta;
  op.p[RCV_P_X4] = (void*)save_mean; op.p[RCV_P_X5] = (void*)save_istd;
  return rcv_run(h, &op, 1, stream);
}

int rcv_maxpool2x2_fwd(rcv_handle* h, const float* r, const float* consts, float* out, int N, int H, int W, int C, void* stream) {
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_POOL_FWD;
  op.i[RCV_I_N] = N; op.i[RCV_I_H] = H; op.i[RCV_I_W] = W; op.i[RCV_I_COUT] = C;
  op.i[RCV_I_INMODE] = consts ? RCV_LOAD_AFFINE : RCV_LOAD_PLAIN;
  op.p[RCV_P_IN] = (void*)r; op.p[RCV_P_IN_C] = (void*)consts; op.p[RCV_P_OUT] = out;
  return rcv_run(h, &op, 1, stream);
}

int rcv_softmax_ce_argmax_fwd(rcv_handle* h, const float* logits, const int64_t* target, const float* class_weight, int N, int C,
                              int H, int W, float* part, int n_part, float* loss_out, uint8_t* argmax, void* stream) {
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_CE_FWD;
  op.flags = argmax ? RCV_F_ARGMAX : 0;
  op.i[RCV_I_N] = N; op.i[RCV_I_H] = H; op.i[RCV_I_W] = W; op.i[RCV_I_COUT] = C; op.i[RCV_I_NPART] = n_part;
  op.p[RCV_P_IN] = (void*)logits; op.p[RCV_P_IN2] = (void*)target; op.p[RCV_P_W] = (void*)class_weight;
  op.p[RCV_P_PART] = part; op.p[RCV_P_OUT] = loss_out; op.p[RCV_P_X0] = argmax;
  return rcv_run(h, &op, 1, stream);
}

int rcv_softmax_ce_bwd(rcv_handle* h, const float* logits, const int64_t* target, const float* class_weight, const float* loss_out,
                       const float* grad_out, int N, int C, int H, int W, float* dlogits, void* stream) {
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_CE_BWD;
  op.i[RCV_I_N] = N; op.i[RCV_I_H] = H; op.i[RCV_I_W] = W; op.i[RCV_I_COUT] = C;
  op.p[RCV_P_IN] = (void*)logits; op.p[RCV_P_IN2] = (void*)target; op.p[RCV_P_W] = (void*)class_weight;
  op.p[RCV_P_X0] = (void*)loss_out; op.p[RCV_P_X1] = (void*)grad_out; op.p[RCV_P_OUT] = dlogits;
  return rcv_run(h, &op, 1, stream);
}

int rcv_dice_fwd(rcv_handle* h, const float* logits, const int64_t* target, const float* class_weight, float eps, int N, int C, int H,
                 int W, float* part, int n_part, float* out, uint8_t* argmax, void* stream) {
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_DICE_FWD;
  op.flags = argmax ? RCV_F_ARGMAX : 0;
  op.i[RCV_I_N] = N; op.i[RCV_I_H] = H; op.i[RCV_I_W] = W; op.i[RCV_I_COUT] = C; op.i[RCV_I_NPART] = n_part;
  op.f[1] = eps;
  op.p[RCV_P_IN] = (void*)logits; op.p[RCV_P_IN2] = (void*)target; op.p[RCV_P_W] = (void*)class_weight;
  op.p[RCV_P_PART] = part; op.p[RCV_P_OUT] = out; op.p[RCV_P_X0] = argmax;
  return rcv_run(h, &op, 1, stream);
}

int rcv_dice_bwd(rcv_handle* h, const float* logits, const int64_t* target, const float* fwd_out, const float* grad_out, int N, int C,
                 int H, int W, float* dlogits, void* stream) {
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_DICE_BWD;
  op.i[RCV_I_N] = N; op.i[RCV_I_H] = H; op.i[RCV_I_W] = W; op.i[RCV_I_COUT] = C;
  op.p[RCV_P_IN] = (void*)logits; op.p[RCV_P_IN2] = (void*)target;
  op.p[RCV_P_X0] = (void*)fwd_out; op.p[RCV_P_X1] = (void*)grad_out; op.p[RCV_P_OUT] = dlogits;
  return rcv_run(h, &op, 1, stream);
}

int rcv_confusion(rcv_handle* h, const uint8_t* argmax, const int64_t* target, int N, int C, int H, int W, int32_t* counts, void* stream) {
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_CONFUSION;
  op.i[RCV_I_N] = N; op.i[RCV_I_H] = H; op.i[RCV_I_W] = W; op.i[RCV_I_COUT] = C;
  op.p[RCV_P_IN] = (void*)argmax; op.p[RCV_P_IN2] = (void*)target; op.p[RCV_P_OUT] = counts;
  return rcv_run(h, &op, 1, stream);
}

int rcv_sgd_step(rcv_handle* h, float* param, const float* grad, float* momentum_buf, const float* lr_elem, int64_t n, float lr,
                 float momentum, float weight_decay, int step, float grad_scale, void* stream) {
  RCV_CHECK_ARG(n > 0 && n < ((int64_t)1 << 32), "rcv_sgd_step: n out of range");
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_SGD;
  op.i[RCV_I_COUNT] = (int32_t)(uint32_t)n; op.i[RCV_I_AUX0] = step;
  op.f[0] = lr; op.f[1] = momentum; op.f[2] = weight_decay; op.f[5] = grad_scale;
  op.p[RCV_P_IN] = param; op.p[RCV_P_IN2] = (void*)grad; op.p[RCV_P_X0] = momentum_buf; op.p[RCV_P_X2] = (void*)lr_elem;
  return rcv_run(h, &op, 1, stream);
}

int rcv_adam_l1_step(rcv_handle* h, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* lr_elem,
                     int64_t n, float lr, float beta1, float beta2, float eps, float decay, int step, float grad_scale,
                     void* stream) {
  RCV_CHECK_ARG(n > 0 && n < 2147483647LL, "rcv_adam_l1_step: n out of range");
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_ADAM_L1;
  op.i[RCV_I_COUNT] = (int)n; op.i[RCV_I_AUX0] = step;
  op.f[0] = lr; op.f[1] = beta1; op.f[2] = beta2; op.f[3] = eps; op.f[4] = decay; op.f[5] = grad_scale;
  op.p[RCV_P_IN] = param; op.p[RCV_P_IN2] = (void*)grad; op.p[RCV_P_X0] = exp_avg; op.p[RCV_P_X1] = exp_avg_sq;
  op.p[RCV_P_X2] = (void*)lr_elem;
  return rcv_run(h, &op, 1, stream);
}

int rcv_adam_l1_step_pruned(rcv_handle* h, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* lr_elem,
                            const uint8_t* prune_mask, int64_t n, float lr, float beta1, float beta2, float eps, float decay, int step,
                            float grad_scale, void* stream) {
  RCV_CHECK_ARG(n > 0 && n < 2147483647LL, "rcv_adam_l1_step_pruned: n out of range");
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_ADAM_L1;
  op.i[RCV_I_COUNT] = (int)n; op.i[RCV_I_AUX0] = step;
  op.f[0] = lr; op.f[1] = beta1; op.f[2] = beta2; op.f[3] = eps; op.f[4] = decay; op.f[5] = grad_scale;
  op.p[RCV_P_IN] = param; op.p[RCV_P_IN2] = (void*)grad; op.p[RCV_P_X0] = exp_avg; op.p[RCV_P_X1] = exp_avg_sq;
  op.p[RCV_P_X2] = (void*)lr_elem; op.p[RCV_P_X5] = (void*)prune_mask;
  return rcv_run(h, &op, 1, stream);
}

int rcv_adam_l1_step_metrics(rcv_handle* h, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* lr_elem,
                             int64_t n, float lr, float beta1, float beta2, float eps, float decay, int step, float grad_scale,
                             double* metrics, const float* loss_stats, void* workspace, int workspace_rows, void* stream) {
  RCV_CHECK_ARG(n > 0 && n < 2147483647LL, "rcv_adam_l1_step_metrics: n out of range");
  RCV_CHECK_ARG(metrics && loss_stats && workspace, "rcv_adam_l1_step_metrics: null operand");
  rcv_op op;
  memset(&op, 0, sizeof(op));
  op.kind = RCV_OP_ADAM_L1;
  op.i[RCV_I_COUNT] = (int)n; op.i[RCV_I_AUX0] = step; op.i[RCV_I_NPART] = workspace_rows;
  op.f[0] = lr; op.f[1] = beta1; op.f[2] = beta2; op.f[3] = eps; op.f[4] = decay; op.f[5] = grad_scale;
  op.p[RCV_P_IN] = param; op.p[RCV_P_IN2] = (void*)grad; op.p[RCV_P_X0] = exp_avg; op.p[RCV_P_X1] = exp_avg_sq;
  op.p[RCV_P_X2] = (void*)lr_elem; op.p[RCV_P_X3] = metrics; op.p[RCV_P_X4] = (void*)loss_stats; op.p[RCV_P_PART] = workspace;
  return rcv_run(h, &op, 1, stream);
}

}  // extern "C"
