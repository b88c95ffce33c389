// C-ABI layer of libibs_hip.so (declarations and reference citations: include/ibs.h).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/ibs.h"
#include "ibs_launch.hpp"
#include "ibs_wave.hpp"
#include "ibs_lbfgsb2.hpp"
#include "ibs_refine.hpp"
#include <chrono>
#include <thread>

namespace ibs {
LaunchTable& launch_table() {
  static LaunchTable t{};
  return t;
}
LaunchNote& last_launch() {
  static thread_local LaunchNote n{};
  return n;
}
void note_launch(long blocks, int threads, const char* fmt, ...) {
  LaunchNote& n = last_launch();
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(n.name, sizeof(n.name), fmt, ap);
  va_end(ap);
  n.blocks = blocks; n.threads = threads;
}
}  // namespace ibs

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(IBS_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Diagnostic overrides of the dispatch heuristics (0 = automatic).  Read ONCE from the environment when the context is
// created (IBS_FORCE_P, IBS_SCAN_CHAIN, IBS_CHAIN_W1, IBS_CHAIN_W2, IBS_GEO_LPP), changed afterwards only through
// ibs_set_option(): no getenv() on the call path, nothing process-global.
struct ibs_options {
  int force_p = 0;        // lanes per system: 64 | 32 | 16
  int scan_chain = 0;     // theta0 values chained through one wave / group
  int geo_lpp = 0;        // lanes per grid point of the geometry kernel: 1 | 2 | 4
  int gcf_rows = -1;      // raw systems on long grids: -1 / 1 = row-streamed kernel, 0 = the 3-row staging of k_solve_gcf
  int gcf_direct = -1;    // raw systems, one wave per system: rows read straight from global memory (k_solve_gcf_direct): -1 = by batch size, 0 = never, 1 = always
  int pack_mode = 0;      // hand-off of the fused scan + argmax: 1 = write-through + sc1 loads, 2 = release / acquire fences
  int f32_lam = 0;        // FP32 eigenvalue-only requests: 0 = by grid size, 1 = all-FP32 iteration + FP64 certificate, 2 = FP32 in HBM + FP64 solver
  double sigma0 = std::numeric_limits<double>::quiet_NaN();   // not NaN: solves that return lam AND info flag lam_max >= sigma0 (informational status bit 4: utils.py:1597 would have taken the eigenpair nearest sigma0)
  int sturm_form = 0;     // ibs_sturm_count_f64: 0 = by size (see there), 1 = prefix-product sweep (N <= 2050), 2 = division form, lanes as systems, 3 = division form, one wave per system
  int reclose = 1;        // FP64 raw systems: 1 = a solve whose closing bracket fails its consistency checks is re-closed in division form, 2 = only marked, 0 = off
  int refine_tangent = -1; // refinement: alpha-tangent of a point staged in LDS (1) or read from global memory in the sums (0); -1 = by batch size
  double chain_w1 = 0.25, chain_w2 = 1.0;   // relative widths of the chain's warm starts
};

struct ibs_ctx {
  ibs_options opt, opt_created;
  int device = 0;
  hipStream_t stream = nullptr;
  // staging workspace for IBS_MEM_HOST calls (grown on demand, reused)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // page-locked mirror of the first hs_bytes of ws: small host-pointer calls pack their inputs / outputs into it and move
  // them with ONE copy each way (HostStage)
  void* hs = nullptr;
  size_t hs_bytes = 0;
  // workspace of the long-grid path (N > 2050: ibs_long.hip), grown on demand
  void* long_ws = nullptr;
  size_t long_ws_bytes = 0;
  // suspect list of the big-batch raw kernels (GcfArgs::fix_*): [count | system indices | polish values], grown on demand
  void* fix_buf = nullptr;
  long fix_cap = 0;
  int lds_per_block = 160 * 1024;
  int n_cu = 256;
  // native RCCL communicator of this rank (ibs_comm_init), null = none
  void* comm = nullptr;
  int comm_rank = 0, comm_n = 1;
  // overlapped gathers (ibs_comm_allgather_start_f64): the communicator's own stream, "inputs written" marker of the
  // compute stream, one completion event per slot
  static constexpr int kCommSlots = 16;
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ready = nullptr;
  hipEvent_t comm_done[kCommSlots] = {};
  bool comm_pending[kCommSlots] = {};
  // per-surface arrival counters of the fused scan + argmax kernel (zero between launches; the last arriver resets its
  // word).  One buffer PER STREAM the context has been used on: two plans of one context run under different streams may
  // have their fused kernels in flight at the same time, and shared counters would mix their arrivals.
  struct SurfCounters { hipStream_t stream; int* buf; int n; };
  std::vector<SurfCounters> surf_counters;
  // ibs_refine_f64: per-round counts posted by the device (pinned host memory), statistics of the last call
  int* refine_hist = nullptr;
  int refine_hist_len = 0;
  long long refine_stats[4] = {0, 0, 0, 0};      // evaluations, forward sweeps, rounds, rounds enqueued
  bool refine_pending = false;                   // the last call's output kernel may still be running (device-pointer call)
  // the device-resident mode-row tables last checked by geo_rows_fit (pointers + counts) and the verdict: a few entries, oldest
  // replaced (a driver alternating between table sets does not pay the copy-back every call).  The key is addresses and sizes: a
  // caller that REUSES device memory for other rows must say so (option "forget_rows": ibs_amd does when it uploads a table set)
  struct RowsSeen { const void* r1 = nullptr; const void* r2 = nullptr; const void* xn = nullptr; const void* xnq = nullptr;
                    int n1 = 0, n2 = 0, mn = 0, mnq = 0; double d1 = 0, d2 = 0; int verdict = 0; };
  static constexpr int kRowsSeen = 4;
  RowsSeen rows_seen[kRowsSeen];
  int rows_seen_next = 0;
};

namespace {

// RCCL, bound at run time (the library carries no link-time dependency on it; in a PyTorch process the copy PyTorch has
// already loaded is used).  ncclUniqueId is a 128-byte struct passed BY VALUE to ncclCommInitRank.
struct nccl_id_t { char internal[128]; };
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(nccl_id_t*) = nullptr;
  int (*CommInitRank)(void**, int, nccl_id_t, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi& rccl() { static RcclApi a; return a; }
constexpr int kNcclFloat64 = 8;          // ncclDataType_t (rccl.h)

// every entry point runs on the context's device and leaves the caller's current device as it found it
struct DeviceGuard {
  int prev = -1; hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) err = hipSetDevice(dev); else prev = -1;     // (nothing to restore)
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ON_DEVICE(ctx_)                                                                                     \
  DeviceGuard dev_guard_((ctx_)->device);                                                                   \
  if (dev_guard_.err != hipSuccess) return fail(IBS_ERR_HIP, "hipSetDevice(%d) -> %s", (ctx_)->device, hipGetErrorString(dev_guard_.err))

struct Arena {  // carve device buffers out of the context workspace
  ibs_ctx* c; size_t off = 0;
  explicit Arena(ibs_ctx* c_) : c(c_) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* p = reinterpret_cast<T*>(static_cast<char*>(c->ws) + off);
    off += n * sizeof(T);
    return p;
  }
};
size_t pad256(size_t b) { return (b + 255) & ~size_t(255); }

int ensure_long_ws(ibs_ctx* c, size_t bytes) {
  if (bytes <= c->long_ws_bytes) return 0;
  if (c->long_ws) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->long_ws)); c->long_ws = nullptr; c->long_ws_bytes = 0; }
  HIPCHK(hipMalloc(&c->long_ws, bytes));
  c->long_ws_bytes = bytes;
  return 0;
}
int ensure_fix(ibs_ctx* c, long n_sys) {
  if (n_sys <= c->fix_cap) return 0;
  if (c->fix_buf) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->fix_buf)); c->fix_buf = nullptr; c->fix_cap = 0; }
  const long cap = n_sys < 4096 ? 4096 : n_sys;
  HIPCHK(hipMalloc(&c->fix_buf, 256 + (size_t)cap * 16));
  c->fix_cap = cap;
  return 0;
}
int long_waves(const ibs_ctx* c, long n_sys) { const long cap = 8L * c->n_cu; return (int)(n_sys < cap ? n_sys : cap); }   // (ibs_long.hip: kLongChunk)

int ensure_ws(ibs_ctx* c, size_t bytes) {
  if (bytes <= c->ws_bytes) return 0;
  if (c->ws) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
  HIPCHK(hipMalloc(&c->ws, bytes));
  c->ws_bytes = bytes;
  return 0;
}

// Small host-pointer calls (the drop-in gamma_ball_full: one system, ~70 KB in, ~16 KB out) spent most of their 0.17-0.22 ms
// in a dozen pageable hipMemcpyAsync of a few KB each (every one staged and waited for by the runtime).  HostStage mirrors the
// device arena in page-locked host memory: up() copies a source into the mirror at its device offset, flush_in() moves the
// whole input span with ONE copy; down() records an output, flush_out() brings the output span back with ONE copy, waits for
// the stream and hands the pieces to the caller's buffers.  Calls whose arena exceeds kMaxBytes take the direct copies.
struct HostStage {
  static constexpr size_t kMaxBytes = 4u << 20;
  ibs_ctx* c; bool on = false;
  size_t in_lo = ~size_t(0), in_hi = 0, out_lo = ~size_t(0), out_hi = 0;
  struct Piece { void* dst; size_t off, bytes; };
  std::vector<Piece> outs;
  bool in_flight = false;      // flush_in() ran and flush_out() has not: the pinned mirror / arena are still being read by the stream
  // (an early error return between the two must not leave that copy running while the next call refills the mirror)
  ~HostStage() { if (in_flight) (void)hipStreamSynchronize(c->stream); }
  HostStage(const HostStage&) = delete;
  HostStage& operator=(const HostStage&) = delete;
  HostStage(ibs_ctx* c_, size_t need) : c(c_) {
    if (need > kMaxBytes) return;
    if (c->hs_bytes < need) {
      if (c->hs) { (void)hipStreamSynchronize(c->stream); (void)hipHostFree(c->hs); c->hs = nullptr; c->hs_bytes = 0; }
      const size_t want = need < (256u << 10) ? (256u << 10) : need;
      if (hipHostMalloc(&c->hs, want, hipHostMallocDefault) != hipSuccess) { c->hs = nullptr; (void)hipGetLastError(); return; }
      c->hs_bytes = want;
    }
    on = true;
  }
  size_t off_of(const void* dev) const { return (size_t)(static_cast<const char*>(dev) - static_cast<const char*>(c->ws)); }
  hipError_t up(const void* src, size_t bytes, void* dev) {
    if (!on) return hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, c->stream);
    const size_t o = off_of(dev);
    std::memcpy(static_cast<char*>(c->hs) + o, src, bytes);
    if (o < in_lo) in_lo = o;
    if (o + bytes > in_hi) in_hi = o + bytes;
    return hipSuccess;
  }
  hipError_t flush_in() {
    if (!on || in_hi <= in_lo) return hipSuccess;
    in_flight = true;
    return hipMemcpyAsync(static_cast<char*>(c->ws) + in_lo, static_cast<char*>(c->hs) + in_lo, in_hi - in_lo, hipMemcpyHostToDevice, c->stream);
  }
  hipError_t down(void* dst, const void* dev, size_t bytes) {
    if (!on) return hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    const size_t o = off_of(dev);
    outs.push_back(Piece{dst, o, bytes});
    if (o < out_lo) out_lo = o;
    if (o + bytes > out_hi) out_hi = o + bytes;
    return hipSuccess;
  }
  // ends with the stream synchronised and every recorded output in the caller's memory
  hipError_t flush_out() {
    if (on && out_hi > out_lo) {
      hipError_t e = hipMemcpyAsync(static_cast<char*>(c->hs) + out_lo, static_cast<char*>(c->ws) + out_lo, out_hi - out_lo, hipMemcpyDeviceToHost, c->stream);
      if (e != hipSuccess) return e;
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    in_flight = false;
    if (e != hipSuccess) return e;
    for (const Piece& p : outs) std::memcpy(p.dst, static_cast<char*>(c->hs) + p.off, p.bytes);
    return hipSuccess;
  }
};

// long_ok: the entry point has the generic long-grid path behind it (ibs_long.hip)
bool is_long(int N) { return N > 64 * ibs::kMaxM + 2; }
int check_grid(int32_t N, double h, bool long_ok = false) {
  const int n_max = long_ok ? ibs::kMaxLongN : 64 * ibs::kMaxM + 2;
  if (N < 66 || N > n_max) return fail(IBS_ERR_UNSUPPORTED, "N=%d outside [66, %d]", N, n_max);
  if ((N & 1) == 0) return fail(IBS_ERR_UNSUPPORTED, "N=%d is even: the reference's Simpson rule is restated for odd N only", N);
  if (!(h > 0)) return fail(IBS_ERR_ARG, "h must be > 0");
  return 0;
}
int rows_per_lane(int N) { return (N - 2 + 63) / 64; }

// lanes per system.  64 = one wave per system (lowest latency, any N).  Large batches of short grids are
// throughput (VALU-issue) bound: 32 / 16 lanes per system amortise the scan over 2 / 4 systems per wave.
// option force_p = 64|32|16 overrides (tests).
int pick_lanes(const ibs_ctx* ctx, int N, long n_sys) {
  const int n = N - 2;
  const bool can32 = n <= 32 * 20 && n >= 32 * 3 + 1, can16 = n <= 16 * 16 && n >= 16 * 3 + 1;
  if (ctx->opt.force_p) {
    const int f = ctx->opt.force_p;
    if (f == 32 && can32) return 32;
    if (f == 16 && can16) return 16;
    return 64;
  }
  // Measured (tools/bench_lanes.py, r01_f): every iteration of the shift search is exactly one forward sweep, so
  // the systems sharing a wave stay in step (15 +- 2 iterations) and the cheaper sweeps pay off as soon as the
  // sub-wave launch still fills the chip: P = 32 from one wave per SIMD (N = 513: 2,048 systems 42 vs 50 us,
  // 65,536 systems 0.77 vs 1.00 ms), P = 16 (whose 16-row chunks make a lone wave twice as slow) from two
  // waves per SIMD (N = 257: 65,536 systems 0.38 vs 0.62 ms).  Smaller batches keep one wave per system.
  const long simds = 4L * ctx->n_cu;
  if (can16 && n_sys / 4 >= 2 * simds) return 16;
  // (17..20 rows per lane, i.e. N up to the 641-point D3D grid: a lone wave is slower, pays off from 4 waves per SIMD:
  //  N = 641: 65,536 systems 0.82 vs 1.05 ms, 8,192 systems 147 vs 150 us, 2,048 systems 72 vs 52 us)
  const int m32 = (n + 31) / 32;
  if (can32 && n_sys / 2 >= (m32 > 16 ? 4 : 1) * simds) return 32;
  return 64;
}

__global__ void k_count_status(long n, const int* info, int* out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int bad = (i < n) && (((info[i] >> 16) & 3) != 0);      // (status bit 2 is informational: an FP32 result re-solved in FP64)
  unsigned long long m = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, __popcll(m));
}

// Nearest-sigma report (option "sigma0"): the reference takes the eigenpair NEAREST sigma0 (eigs(..., sigma=sigma0), utils.py:1597;
// sigma0 = 1.0 in the coarse scan, 1.3 |gam| + 0.05 in the refinement, 0.42 in the final solve: ball_scan.py:230, 289, 337); this
// library always returns lam_max.  The two are the same eigenpair whenever lam_max < sigma0 -- true of every equilibrium seen so
// far -- and only then.  A solve with lam_max >= sigma0 carries the informational status bit 4.
template <typename T>
__global__ void k_flag_sigma(long n, const T* lam, int* info, double sigma0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (double)lam[i] >= sigma0) info[i] |= 16 << 16;
}
template <typename T>
static void flag_sigma(const ibs_ctx* ctx, long n, const T* lam, int* info) {
  if (!(ctx->opt.sigma0 == ctx->opt.sigma0) || !lam || !info || n <= 0) return;
  hipLaunchKernelGGL(k_flag_sigma<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, lam, info, ctx->opt.sigma0);
}

static inline int argmax_threads(int n_per) { const int t = ((n_per + 63) / 64) * 64; return t > 256 ? 256 : t; }

// first-argmax per surface (ball_scan.py:283-295 tie rule: lowest row-major index wins)
__global__ void __launch_bounds__(256) k_surface_argmax(int n_per, const double* gam, int* idx, double* val, double* pack) {
  __shared__ double sv[4];
  __shared__ int si[4];
  const double* g = gam + (size_t)blockIdx.x * n_per;
  double best = -1.7976931348623157e308;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n_per; i += blockDim.x) {
    const double v = g[i];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const double v2 = __shfl_xor(best, d);
    const int i2 = __shfl_xor(bi, d);
    if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
  }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (nw > 1) {                      // (one wave per surface when n_per <= 64: no LDS round trip, no barrier)
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int k = 1; k < nw; ++k)
      if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
    if (idx) idx[blockIdx.x] = bi;
    if (val) val[blockIdx.x] = best;
    if (pack) { pack[2 * blockIdx.x] = best; pack[2 * blockIdx.x + 1] = (double)bi; }   // one buffer for the all-gather
  }
}

// Hellmann-Feynman derivative of the growth rate for given tangent coefficient arrays (utils.py:1676-1680 /
// 1721-1725): one wave per system, Simpson sums with unit spacing.
__global__ void __launch_bounds__(256) k_hf_grad(long n_sys, int N, long ld, const double* __restrict__ X,
                                                 const double* __restrict__ dX, const double* __restrict__ f,
                                                 const double* __restrict__ g_p, const double* __restrict__ c_p,
                                                 const double* __restrict__ f_p, const double* __restrict__ gam,
                                                 double* __restrict__ jac) {
  const int lane = threadIdx.x & 63;
  const long sys = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (sys >= n_sys) return;
  const long o = sys * ld;
  double y1 = 0.0, sc = 0.0, sg = 0.0, sf = 0.0;
  for (int j = lane; j < N; j += 64) {
    const double w = (j == 0 || j == N - 1) ? 1.0 : ((j & 1) ? 4.0 : 2.0);
    const double x2 = w * X[o + j] * X[o + j], d2 = w * dX[o + j] * dX[o + j];
    y1 += f[o + j] * x2; sc += c_p[o + j] * x2; sg += g_p[o + j] * d2; sf += f_p[o + j] * x2;
  }
  y1 = ibs::wave_sum(y1); sc = ibs::wave_sum(sc); sg = ibs::wave_sum(sg); sf = ibs::wave_sum(sf);
  if (lane == 0) jac[sys] = sc / y1 - sg / y1 - gam[sys] * sf / y1;
}

// start points of the refinement from the per-surface maxima of the coarse scan (ball_scan.py:279-295)
__global__ void k_scan_starts(int n_surf, int n_alpha, int n_theta0, const double* alpha, const double* theta0,
                              const double* pack, double* start, double* sigma0, int* n_bad) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_surf) return;
  const double m = pack[2 * s], fi = pack[2 * s + 1];
  double a = 0.0, t = 0.0, sg = 0.05;                                      // ball_scan.py:279-282: an all-zero table
  const bool ok = (m == m) && fabs(m) <= 1.7976931348623157e308 && fi >= 0.0 && fi < (double)n_alpha * n_theta0;
  if (!ok) atomicAdd(n_bad, 1);                                           // a NaN / flagged table: reported, start (0, 0)
  else if (m != 0.0) {
    const int idx = (int)fi, i = idx / n_theta0, j = idx - i * n_theta0;   // first maximum, row-major (ball_scan.py:283-288)
    a = alpha[i]; t = theta0[j]; sg = 1.3 * fabs(m) + 0.05;               // ball_scan.py:289, 295
  }
  start[2 * s] = a; start[2 * s + 1] = t;
  if (sigma0) sigma0[s] = sg;
}

// One wave per system on a long grid: k_solve_gcf's three staged rows are 24.6 KB of LDS per wave at N = 1025 -- five waves
// per CU where the registers admit eight.  A batch that fills the chip at more waves than the staging admits reads its rows
// straight from global memory instead (k_solve_gcf_direct: no LDS).  Option gcf_direct: -1 = this rule, 0 = never, 1 = always.
static bool use_direct(const ibs_ctx* ctx, int N, long n_sys) {
  if (ctx->opt.gcf_direct == 0) return false;
  if (ctx->opt.gcf_direct == 1) return true;
  // (measured, tools/bench_direct.py, 2^19 systems: 1.45-1.65 x the staged / row-streamed kernels at N_zeta = 768 .. 1536, 1.2-1.3 x at 2048)
  const long staged_waves_per_cu = (long)(ctx->lds_per_block / ((size_t)3 * ibs::lds_pitch(N) * sizeof(double)));
  return staged_waves_per_cu < 8 && n_sys > (staged_waves_per_cu > 4 ? staged_waves_per_cu : 4) * ctx->n_cu;
}

template <typename T>
int solve_gcf_impl(ibs_ctx* ctx, int64_t n_sys, int32_t N, T h, const T* g, const T* c, const T* f, int64_t ld,
                   T* lam, T* gam, T* X, T* dX, int32_t* info, int32_t mem,
                   hipError_t (*const* table)(const ibs::GcfArgs<T>&, hipStream_t), const T* gh = nullptr) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (n_sys < 0 || !g || !c || !f || ld < N) return fail(IBS_ERR_ARG, "bad arguments (n_sys=%lld ld=%lld N=%d)", (long long)n_sys, (long long)ld, N);
  if (int r = check_grid(N, (double)h, true)) return r;
  if (n_sys == 0) return 0;
  // grids beyond 2050 points: the generic division-form path (ibs_long.hip), FP64 arithmetic on either element type
  const bool lng = is_long(N);
  int M = lng ? 1 : rows_per_lane(N);
  if (!table[M]) return fail(IBS_ERR_UNSUPPORTED, "no kernel built for rows-per-lane M=%d (N=%d)", M, N);
  ON_DEVICE(ctx);
  auto launch = table[M];
  size_t per_wave = (size_t)3 * ibs::lds_pitch(lng ? 66 : N) * sizeof(T);
  auto launch_it = [&](ibs::GcfArgs<T>& a) -> int {
    if (!lng) {
      // the big-batch forms (rows from global memory, sub-wave) only LIST their suspect systems; k_fix_gcf re-closes them afterwards
      auto& t = ibs::launch_table();
      bool lists;
      if constexpr (sizeof(T) == 8) lists = launch == t.gcf_direct_f64[M] || launch == t.gcf_f64_g[0][M] || launch == t.gcf_f64_g[1][M];
      else lists = launch == t.gcf_direct_f32w[M] || launch == t.gcf_f32w_g[0][M] || launch == t.gcf_f32w_g[1][M];
      if (lists && (a.flags & 1)) {
        if (int r = ensure_fix(ctx, (long)a.n_sys)) return r;
        char* fb = static_cast<char*>(ctx->fix_buf);
        a.fix_count = reinterpret_cast<int*>(fb);
        a.fix_sys = reinterpret_cast<long*>(fb + 256);
        a.fix_center = reinterpret_cast<double*>(fb + 256 + (size_t)ctx->fix_cap * 8);
        HIPCHK(hipMemsetAsync(a.fix_count, 0, sizeof(int), ctx->stream));
      }
      HIPCHK(launch(a, ctx->stream));
      if (a.fix_count) {
        ibs::LaunchNote keep = ibs::last_launch();         // (ibs_last_launch names the solver kernel, not its fix-up pass)
        auto fix = (sizeof(T) == 8 ? (void*)t.gcf_fix_f64[rows_per_lane(N)] : (void*)t.gcf_fix_f32w[rows_per_lane(N)]);
        if (!fix) return fail(IBS_ERR_UNSUPPORTED, "no fix-up kernel built for N=%d", N);
        HIPCHK(reinterpret_cast<hipError_t (*)(const ibs::GcfArgs<T>&, hipStream_t)>(fix)(a, ctx->stream));
        ibs::last_launch() = keep;
      }
      return 0;
    }
    const int nw = long_waves(ctx, (long)a.n_sys);
    if (int r = ensure_long_ws(ctx, (size_t)nw * 3 * (size_t)N * sizeof(double))) return r;
    ibs::LongGcfArgs la{};
    la.n_sys = a.n_sys; la.N = N; la.h = (double)a.h; la.g = a.g; la.c = a.c; la.f = a.f; la.gh = a.gh; la.f32 = sizeof(T) == 4;
    la.ld = a.ld; la.lam = a.lam; la.gam = a.gam; la.X = a.X; la.dX = a.dX; la.info = a.info;
    la.work = static_cast<double*>(ctx->long_ws); la.n_waves = nw;
    HIPCHK(ibs::launch_gcf_long(la, ctx->stream));
    return 0;
  };
  if constexpr (sizeof(T) == 8) {
    const int P = (gh || lng) ? 64 : pick_lanes(ctx, N, (long)n_sys);
    if (P != 64) {
      const int Mg = (N - 2 + P - 1) / P;
      auto fn = ibs::launch_table().gcf_f64_g[P == 32 ? 0 : 1][Mg];
      if (fn) { launch = fn; M = Mg; per_wave = (size_t)(64 / P) * ibs::lds_pitch(N) * sizeof(T); }
    }
  }
  if constexpr (sizeof(T) == 4) if (!lng) {
    // FP32 systems whose growth rate or eigenfunction is wanted: FP32 in HBM, widened to FP64 as they are read, solved by the
    // FP64 solver -- in the same three forms as the FP64 entry point (sub-wave for large batches of short grids, row-streamed
    // for long grids, else one wave per system with the three rows staged).  lam alone stays with the all-FP32 kernel, whose
    // result is certified by an FP64 count pair (k_solve_gcf<float, M>).
    // lam alone: the all-FP32 iteration with its FP64 certificate.  A batch that would get the 32-lane sub-wave form takes it with
    // the rows read straight from global memory (k_solve_gcf_f32lam_direct: 1.12-1.15 x the FP64 sub-wave solver on FP32 bytes on
    // smooth coefficients, 2.1 x on rough ones; 2,048 ... 10^6 systems, N_zeta = 256 ... 640, tools/bench_forms.py).  Where the
    // 16-lane form runs (N <= 258, big batches) the FP64 sub-wave solver without its growth-rate stage stays: 1.5 x the all-FP32
    // form on smooth coefficients, 0.83 x on rough ones.  Also where the direct form is not built or gcf_direct = 0; option
    // f32_lam overrides.
    const int P_lam = pick_lanes(ctx, N, (long)n_sys);
    const auto fl = ibs::launch_table().gcf_direct_f32lam[M];
    const bool fl_ok = fl && ctx->opt.gcf_direct != 0;
    const bool wide_lam = ctx->opt.f32_lam == 2 || (ctx->opt.f32_lam == 0 && (P_lam == 16 || (P_lam == 32 && !fl_ok)));
    if (gam || X || dX || wide_lam) {
      auto fw = ibs::launch_table().gcf_f32_wide[M];
      if (fw) { launch = fw; per_wave = (size_t)3 * ibs::lds_pitch(N) * sizeof(double); }
      const int P = pick_lanes(ctx, N, (long)n_sys);
      if (P != 64) {
        const int Mg = (N - 2 + P - 1) / P;
        auto fg = ibs::launch_table().gcf_f32w_g[P == 32 ? 0 : 1][Mg];
        if (fg) { launch = fg; M = Mg; per_wave = (size_t)(64 / P) * ibs::lds_pitch(N) * sizeof(double); }
      } else {
        auto fr = ibs::launch_table().gcf_f32w_rows[M];
        if (fr && ctx->opt.gcf_rows != 0) { launch = fr; per_wave = (size_t)ibs::lds_pitch(N) * sizeof(double); }
        auto fd = ibs::launch_table().gcf_direct_f32w[M];
        if (fd && use_direct(ctx, N, (long)n_sys)) { launch = fd; per_wave = (size_t)ibs::lds_pitch(N) * sizeof(double); }
      }
    } else {
      // eigenvalues only, all-FP32 iteration + FP64 certificate: big batches read their rows from global memory
      if (!gh && fl_ok && (P_lam != 64 || use_direct(ctx, N, (long)n_sys))) launch = fl;
    }
  }
  if constexpr (sizeof(T) == 8) if (!lng) {
    // long grids, one wave per system: stream the three rows through ONE LDS row per wave (k_solve_gcf_rows) -- the
    // 3-row staging of k_solve_gcf leaves two waves per CU at N_zeta = 2048 and five at 1024
    auto fr = ibs::launch_table().gcf_rows_f64[M];
    if (!gh && launch == table[M] && fr && ctx->opt.gcf_rows != 0) { launch = fr; per_wave = (size_t)ibs::lds_pitch(N) * sizeof(T); }
    auto fd = ibs::launch_table().gcf_direct_f64[M];
    if (!gh && (launch == table[M] || launch == fr) && fd && use_direct(ctx, N, (long)n_sys)) { launch = fd; per_wave = (size_t)ibs::lds_pitch(N) * sizeof(T); }
  }
  // the direct forms stage nothing: four waves per block, LDS only for the row X / dX are written from (ibs_kernels.hip: launch_gcf_direct,
  // launch_gcf_f32lam_direct take the block shape from a.wpb)
  bool direct_form;
  if constexpr (sizeof(T) == 8) direct_form = !lng && launch == ibs::launch_table().gcf_direct_f64[M];
  else direct_form = !lng && (launch == ibs::launch_table().gcf_direct_f32w[M] || launch == ibs::launch_table().gcf_direct_f32lam[M]);
  if (direct_form) per_wave = (X || dX) ? (size_t)ibs::lds_pitch(N) * sizeof(double) : 0;
  int wpb = per_wave ? (int)((size_t)ctx->lds_per_block / per_wave) : 4;
  if (wpb > 4) wpb = 4;
  // keep >= 3 blocks per CU resident when LDS allows it
  while (!direct_form && wpb > 1 && (size_t)wpb * per_wave * 3 > (size_t)ctx->lds_per_block) --wpb;
  if (wpb < 1) return fail(IBS_ERR_UNSUPPORTED, "N=%d needs %zu B of LDS per wave", N, per_wave);
  ibs::GcfArgs<T> a{};
  a.n_sys = n_sys; a.N = N; a.h = h; a.ld = ld; a.wpb = wpb;
  a.flags = ctx->opt.reclose == 1 ? 1 : (ctx->opt.reclose == 2 ? 2 : 0);
  int* d_info = nullptr;
  int* d_nbad = nullptr;
  if (mem == IBS_MEM_HOST) {
    const size_t in_elems = (size_t)n_sys * ld, out_elems = (size_t)n_sys * N;
    size_t need = 4 * pad256(in_elems * sizeof(T)) + 2 * pad256(n_sys * sizeof(T)) + 2 * pad256(out_elems * sizeof(T)) +
                  pad256(n_sys * sizeof(int)) + 4096;
    if (int r = ensure_ws(ctx, need)) return r;
    Arena ar(ctx);
    HostStage hs(ctx, need);
    T* dg = ar.take<T>(in_elems); T* dc = ar.take<T>(in_elems); T* df = ar.take<T>(in_elems);
    T* dgh = gh ? ar.take<T>(in_elems) : nullptr;
    T* dlam = ar.take<T>(n_sys); T* dgam = ar.take<T>(n_sys);
    T* dX_ = X ? ar.take<T>(out_elems) : nullptr; T* ddX = dX ? ar.take<T>(out_elems) : nullptr;
    d_info = ar.take<int>(n_sys); d_nbad = ar.take<int>(1);
    HIPCHK(hs.up(g, in_elems * sizeof(T), dg));
    HIPCHK(hs.up(c, in_elems * sizeof(T), dc));
    HIPCHK(hs.up(f, in_elems * sizeof(T), df));
    if (gh) HIPCHK(hs.up(gh, in_elems * sizeof(T), dgh));
    HIPCHK(hs.flush_in());
    // (gam not asked for: the kernels then take their eigenvalue-only exits, as they do for device-pointer calls)
    a.g = dg; a.c = dc; a.f = df; a.lam = dlam; a.gam = gam ? dgam : nullptr; a.X = dX_; a.dX = ddX; a.info = d_info;
    if (gh) a.gh = dgh;
    if (int r = launch_it(a)) return r;
    flag_sigma<T>(ctx, (long)n_sys, a.lam, a.info);
    HIPCHK(hipMemsetAsync(d_nbad, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_count_status, dim3((unsigned)((n_sys + 255) / 256)), dim3(256), 0, ctx->stream, (long)n_sys, d_info, d_nbad);
    if (lam) HIPCHK(hs.down(lam, dlam, n_sys * sizeof(T)));
    if (gam) HIPCHK(hs.down(gam, dgam, n_sys * sizeof(T)));
    if (X) HIPCHK(hs.down(X, dX_, out_elems * sizeof(T)));
    if (dX) HIPCHK(hs.down(dX, ddX, out_elems * sizeof(T)));
    if (info) HIPCHK(hs.down(info, d_info, n_sys * sizeof(int)));
    int nbad = 0;
    HIPCHK(hs.down(&nbad, d_nbad, sizeof(int)));
    HIPCHK(hs.flush_out());
    return nbad;
  }
  a.g = g; a.c = c; a.f = f; a.lam = lam; a.gam = gam; a.X = X; a.dX = dX; a.info = info; a.gh = gh;
  if (int r = launch_it(a)) return r;
  flag_sigma<T>(ctx, (long)n_sys, a.lam, a.info);
  return 0;
}


// ---------------------------------------------------------------- (alpha, theta0) maximiser (row F2): see ibs_refine.hpp
using ibs::RefineState; using ibs::RefineParams; using ibs::RefineCtrl;

__global__ void k_refine_init(int n, const int* pt_surf, const double* start, RefineState* st, RefineParams p,
                              int* idx, int* line_surf, double* line_alpha, double* th0, RefineCtrl* ctrl,
                              int N, const double* theta, int* status) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  // the theta grid must be uniform (the batched kernels take h): checked here, on the device copy; the host sees *status
  const double h = (theta[N - 1] - theta[0]) / (N - 1);
  if (k == 0) {
    ctrl->n_c = n; ctrl->n_lines = 3 * n; ctrl->done = 0; ctrl->round = 0; ctrl->h = h;
    if (!(h > 0)) *status = 2;
  }
  for (int j = 1 + k; j < N; j += gridDim.x * blockDim.x)
    if (fabs((theta[j] - theta[j - 1]) - h) > 1e-9 * fabs(h)) *status = 1;
  if (k >= n) return;
  RefineState& s = st[k];                     // (the state lives in global memory; nothing of it is kept in registers)
  const double x0[2] = {start[2 * k], start[2 * k + 1]};
  ibs::lbfgsb2::init(s.q, x0, p.lo, p.hi, p.ftol, p.gtol, p.maxiter, 20);
  s.active = 1; s.nev = 0; s.have = 0; s.sweeps = 0;
  s.lam_prev = 0.0; s.x_prev[0] = s.x_prev[1] = 0.0; s.g_prev[0] = s.g_prev[1] = 0.0; s.err_prev = 0.0;
  idx[k] = k;
  ibs::refine_emit(s.q.x, p.del_alpha, k, min(max(pt_surf[k], 0), p.n_surf - 1), line_surf, line_alpha, th0);
}

__global__ void k_refine_out(int n, const RefineState* st, double* x_opt, double* f_opt, int* n_evals, long long* stats) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  x_opt[2 * k] = st[k].q.x[0]; x_opt[2 * k + 1] = st[k].q.x[1]; f_opt[k] = st[k].q.f;
  if (n_evals) n_evals[k] = st[k].nev;
  atomicAdd_system(reinterpret_cast<unsigned long long*>(stats), (unsigned long long)st[k].nev);         // (pinned host memory)
  atomicAdd_system(reinterpret_cast<unsigned long long*>(stats + 1), (unsigned long long)st[k].sweeps);
}
}  // namespace

extern "C" {

int ibs_version(void) { return 100; }
const char* ibs_last_error(void) { return g_err.c_str(); }

int ibs_last_launch(char* name, int32_t len, int64_t* blocks, int32_t* threads) {
  const ibs::LaunchNote& n = ibs::last_launch();
  if (name && len > 0) { strncpy(name, n.name, (size_t)len - 1); name[len - 1] = 0; }
  if (blocks) *blocks = n.blocks;
  if (threads) *threads = n.threads;
  return 0;
}

int ibs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ibs_create(ibs_ctx** out, int device_id) {
  if (!out) return fail(IBS_ERR_ARG, "null output pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail(IBS_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(IBS_ERR_ARG, "device %d out of range (0..%d)", device_id, n - 1);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_id));
  ibs_ctx* c = new ibs_ctx();
  c->device = device_id;
  c->lds_per_block = (int)prop.sharedMemPerBlock > 64 * 1024 ? (int)prop.sharedMemPerBlock : 64 * 1024;
  if (prop.maxSharedMemoryPerMultiProcessor > (size_t)c->lds_per_block) c->lds_per_block = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (c->lds_per_block > 160 * 1024) c->lds_per_block = 160 * 1024;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("IBS_FORCE_P")) c->opt.force_p = atoi(e);
  if (const char* e = getenv("IBS_SCAN_CHAIN")) c->opt.scan_chain = atoi(e);
  if (const char* e = getenv("IBS_GEO_LPP")) c->opt.geo_lpp = atoi(e);
  if (const char* e = getenv("IBS_CHAIN_W1")) c->opt.chain_w1 = atof(e);
  if (const char* e = getenv("IBS_CHAIN_W2")) c->opt.chain_w2 = atof(e);
  c->opt_created = c->opt;
  *out = c;
  return 0;
}

int ibs_destroy(ibs_ctx* c) {
  if (!c) return 0;
  (void)ibs_comm_destroy(c);
  DeviceGuard g(c->device);
  if (c->ws) hipFree(c->ws);
  if (c->long_ws) hipFree(c->long_ws);
  if (c->fix_buf) hipFree(c->fix_buf);
  for (auto& sc : c->surf_counters) if (sc.buf) hipFree(sc.buf);
  if (c->refine_hist) hipHostFree(c->refine_hist);
  if (c->hs) hipHostFree(c->hs);
  delete c;
  return 0;
}

int ibs_set_stream(ibs_ctx* c, void* s) {
  if (!c) return fail(IBS_ERR_ARG, "null context");
  c->stream = reinterpret_cast<hipStream_t>(s);
  return 0;
}

int ibs_set_option(ibs_ctx* c, const char* name, double value) {
  if (!c || !name) return fail(IBS_ERR_ARG, "null context / name");
  const bool reset = !(value == value);          // NaN: back to what ibs_create() read from the environment
  const std::string n(name);
  if (n == "force_p") c->opt.force_p = reset ? c->opt_created.force_p : (int)value;
  else if (n == "scan_chain") c->opt.scan_chain = reset ? c->opt_created.scan_chain : (int)value;
  else if (n == "geo_lpp") c->opt.geo_lpp = reset ? c->opt_created.geo_lpp : (int)value;
  else if (n == "gcf_rows") c->opt.gcf_rows = reset ? c->opt_created.gcf_rows : (int)value;
  else if (n == "gcf_direct") c->opt.gcf_direct = reset ? c->opt_created.gcf_direct : (int)value;
  else if (n == "pack_mode") c->opt.pack_mode = reset ? c->opt_created.pack_mode : (int)value;
  else if (n == "f32_lam") c->opt.f32_lam = reset ? c->opt_created.f32_lam : (int)value;
  else if (n == "reclose") c->opt.reclose = reset ? c->opt_created.reclose : (int)value;
  else if (n == "sturm_form") c->opt.sturm_form = reset ? c->opt_created.sturm_form : (int)value;
  else if (n == "sigma0") c->opt.sigma0 = reset ? c->opt_created.sigma0 : value;
  else if (n == "forget_rows") { for (auto& e : c->rows_seen) e = ibs_ctx::RowsSeen{}; }      // (an action, not a setting)
  else if (n == "refine_tangent") c->opt.refine_tangent = reset ? c->opt_created.refine_tangent : (int)value;
  else if (n == "chain_w1") c->opt.chain_w1 = reset ? c->opt_created.chain_w1 : value;
  else if (n == "chain_w2") c->opt.chain_w2 = reset ? c->opt_created.chain_w2 : value;
  else if (n == "all" && reset) c->opt = c->opt_created;
  else return fail(IBS_ERR_ARG, "unknown option '%s'", name);
  return 0;
}

// ---- host side of the bounded quasi-Newton state machine (ibs_lbfgsb2.hpp): no GPU involved
int ibs_lbfgsb2_state_bytes(void) { return (int)sizeof(ibs::lbfgsb2::State); }

int ibs_lbfgsb2_init(void* state, const double* x0, const double* lo, const double* hi, double ftol, double gtol,
                     int32_t maxiter, int32_t maxls) {
  if (!state || !x0 || !lo || !hi) return fail(IBS_ERR_ARG, "null pointer");
  if (!(lo[0] <= hi[0]) || !(lo[1] <= hi[1]) || maxiter < 0 || maxls < 1) return fail(IBS_ERR_ARG, "bad bounds / limits");
  ibs::lbfgsb2::init(*static_cast<ibs::lbfgsb2::State*>(state), x0, lo, hi, ftol, gtol, maxiter, maxls);
  return 0;
}

int ibs_lbfgsb2_step(void* state, double f, const double* g, double* x_next) {
  if (!state || !g || !x_next) return fail(IBS_ERR_ARG, "null pointer");
  auto& s = *static_cast<ibs::lbfgsb2::State*>(state);
  const bool more = ibs::lbfgsb2::step(s, f, g);
  x_next[0] = s.x[0]; x_next[1] = s.x[1];
  return more ? 1 : 0;
}

int ibs_lbfgsb2_result(const void* state, double* x, double* f, int32_t* counters) {
  if (!state) return fail(IBS_ERR_ARG, "null pointer");
  const auto& s = *static_cast<const ibs::lbfgsb2::State*>(state);
  if (x) { x[0] = s.x[0]; x[1] = s.x[1]; }
  if (f) *f = s.f;
  if (counters) { counters[0] = s.n_iterations; counters[1] = s.nfgv; counters[2] = s.task; counters[3] = s.n_restarts; counters[4] = s.nskip; }
  return 0;
}

// ---- native collective: ONE ncclAllGather of the per-surface rows on the context's stream
int ibs_comm_load(const char* librccl_path) {
  RcclApi& a = rccl();
  if (a.handle) return 0;
  const char* cands[3] = {librccl_path, "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* c : cands) {
    if (!c) continue;
    h = dlopen(c, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);          // the copy this process already holds, if any
    if (!h) h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(IBS_ERR_UNSUPPORTED, "librccl not found (%s)", dlerror());
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy)
    return fail(IBS_ERR_UNSUPPORTED, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
  a.handle = h;
  return 0;
}

static int nccl_fail(const char* what, int rc) {
  RcclApi& a = rccl();
  return fail(IBS_ERR_HIP, "%s -> %s", what, a.GetErrorString ? a.GetErrorString(rc) : "rccl error");
}

int ibs_comm_unique_id(void* id128) {
  if (!id128) return fail(IBS_ERR_ARG, "null pointer");
  if (!rccl().handle) { if (int r = ibs_comm_load(nullptr)) return r; }
  const int rc = rccl().GetUniqueId(static_cast<nccl_id_t*>(id128));
  return rc ? nccl_fail("ncclGetUniqueId", rc) : 0;
}

int ibs_comm_init(ibs_ctx* c, const void* id128, int32_t rank, int32_t nranks) {
  if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(IBS_ERR_ARG, "bad arguments");
  if (!rccl().handle) { if (int r = ibs_comm_load(nullptr)) return r; }
  if (c->comm) return fail(IBS_ERR_ARG, "the context already holds a communicator");
  ON_DEVICE(c);
  nccl_id_t id;
  memcpy(&id, id128, sizeof(id));
  void* comm = nullptr;
  const int rc = rccl().CommInitRank(&comm, nranks, id, rank);
  if (rc) return nccl_fail("ncclCommInitRank", rc);
  c->comm = comm; c->comm_rank = rank; c->comm_n = nranks;
  return 0;
}

int ibs_comm_allgather_f64(ibs_ctx* c, const double* send, double* recv, int64_t count_per_rank) {
  if (!c || !send || !recv || count_per_rank < 0) return fail(IBS_ERR_ARG, "bad arguments");
  if (!c->comm) return fail(IBS_ERR_ARG, "no communicator: call ibs_comm_init first");
  ON_DEVICE(c);
  const int rc = rccl().AllGather(send, recv, (size_t)count_per_rank, kNcclFloat64, c->comm, c->stream);
  return rc ? nccl_fail("ncclAllGather", rc) : 0;
}

// Overlapped form: the gather is ordered after everything enqueued so far on the context's stream but runs on the
// communicator's own stream, so the next scan does not wait for the ranks to meet.
int ibs_comm_allgather_start_f64(ibs_ctx* c, const double* send, double* recv, int64_t count_per_rank, int32_t slot,
                                 int32_t then_wait_slot) {
  // then_wait_slot: >= 0 device-side wait of the context's stream on that slot's gather; -1 nothing;
  //                 <= -2 HOST-side wait on slot (-2 - then_wait_slot): no stream operation at all (an event query, and a
  //                 blocking wait only if that gather has not finished) -- for callers that run several slots ahead
  const int host_slot = then_wait_slot <= -2 ? -2 - then_wait_slot : -1;
  if (!c || !send || !recv || count_per_rank < 0 || slot < 0 || slot >= ibs_ctx::kCommSlots ||
      then_wait_slot >= ibs_ctx::kCommSlots || then_wait_slot == slot || host_slot >= ibs_ctx::kCommSlots || host_slot == slot)
    return fail(IBS_ERR_ARG, "bad arguments");
  if (!c->comm) return fail(IBS_ERR_ARG, "no communicator: call ibs_comm_init first");
  ON_DEVICE(c);
  if (!c->comm_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->comm_ready, hipEventDisableTiming));
    for (auto& e : c->comm_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(c->comm_ready, c->stream));
  HIPCHK(hipStreamWaitEvent(c->comm_stream, c->comm_ready, 0));
  const int rc = rccl().AllGather(send, recv, (size_t)count_per_rank, kNcclFloat64, c->comm, c->comm_stream);
  if (rc) return nccl_fail("ncclAllGather", rc);
  HIPCHK(hipEventRecord(c->comm_done[slot], c->comm_stream));
  c->comm_pending[slot] = true;
  if (then_wait_slot >= 0 && c->comm_pending[then_wait_slot]) {      // (saves the caller a second call per step)
    HIPCHK(hipStreamWaitEvent(c->stream, c->comm_done[then_wait_slot], 0));
    c->comm_pending[then_wait_slot] = false;
  }
  if (host_slot >= 0 && c->comm_pending[host_slot]) {
    if (hipEventQuery(c->comm_done[host_slot]) != hipSuccess) HIPCHK(hipEventSynchronize(c->comm_done[host_slot]));
    c->comm_pending[host_slot] = false;
  }
  return 0;
}

int ibs_comm_wait(ibs_ctx* c, int32_t slot) {
  if (!c || slot >= ibs_ctx::kCommSlots) return fail(IBS_ERR_ARG, "bad arguments");
  ON_DEVICE(c);
  for (int s = 0; s < ibs_ctx::kCommSlots; ++s) {
    if ((slot >= 0 && s != slot) || !c->comm_pending[s]) continue;
    HIPCHK(hipStreamWaitEvent(c->stream, c->comm_done[s], 0));      // (device-side ordering; the host does not block)
    c->comm_pending[s] = false;
  }
  return 0;
}

int ibs_comm_destroy(ibs_ctx* c) {
  if (!c) return fail(IBS_ERR_ARG, "null context");
  if (c->comm_stream) {
    ON_DEVICE(c);
    (void)hipStreamSynchronize(c->comm_stream);
    for (auto& e : c->comm_done) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (c->comm_ready) (void)hipEventDestroy(c->comm_ready);
    c->comm_ready = nullptr;
    (void)hipStreamDestroy(c->comm_stream);
    c->comm_stream = nullptr;
    for (auto& b : c->comm_pending) b = false;
  }
  if (c->comm) {
    ON_DEVICE(c);
    (void)hipStreamSynchronize(c->stream);
    const int rc = rccl().CommDestroy(c->comm);
    c->comm = nullptr;
    if (rc) return nccl_fail("ncclCommDestroy", rc);
  }
  return 0;
}

int ibs_synchronize(ibs_ctx* c) {
  if (!c) return fail(IBS_ERR_ARG, "null context");
  ON_DEVICE(c);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int ibs_solve_gcf_f64(ibs_ctx* ctx, int64_t n_sys, int32_t N, double h, const double* g, const double* c,
                      const double* f, int64_t ld, double* lam, double* gam, double* X, double* dX,
                      int32_t* info, int32_t mem) {
  return solve_gcf_impl<double>(ctx, n_sys, N, h, g, c, f, ld, lam, gam, X, dX, info, mem, ibs::launch_table().gcf_f64);
}

int ibs_solve_gcfh_f64(ibs_ctx* ctx, int64_t n_sys, int32_t N, double h, const double* g, const double* gh,
                       const double* c, const double* f, int64_t ld, double* lam, double* gam, double* X, double* dX,
                       int32_t* info, int32_t mem) {
  if (!gh) return fail(IBS_ERR_ARG, "gh is null");
  return solve_gcf_impl<double>(ctx, n_sys, N, h, g, c, f, ld, lam, gam, X, dX, info, mem, ibs::launch_table().gcf_f64, gh);
}

int ibs_solve_gcf_f32(ibs_ctx* ctx, int64_t n_sys, int32_t N, float h, const float* g, const float* c,
                      const float* f, int64_t ld, float* lam, float* gam, float* X, float* dX,
                      int32_t* info, int32_t mem) {
  return solve_gcf_impl<float>(ctx, n_sys, N, h, g, c, f, ld, lam, gam, X, dX, info, mem, ibs::launch_table().gcf_f32);
}

// Geometry-fed scan on a grid beyond 2050 points: the (g, c, f) rows of every (line, theta0) system -- and their theta0 tangents when
// dgam/dtheta0 is wanted -- are written out (k_assemble_gcf_long: the arithmetic the scan kernels do while staging), solved by the
// generic long-grid kernel, and the Hellmann-Feynman sums (utils.py:1676-1680) taken by k_hf_grad.  Warm-start guesses are not
// used (they only ever steer).  The context travels in a thread-local: the launch table's signature has no room for it.
static thread_local ibs_ctx* g_long_ctx = nullptr;
static hipError_t launch_scan_long(const ibs::ScanArgs<double>& a, hipStream_t st) {
  ibs_ctx* ctx = g_long_ctx;
  const size_t n_sys = (size_t)a.n_lines * a.n_theta0, N = (size_t)a.N;
  const bool hf = a.dth0 != nullptr;
  const int nw = long_waves(ctx, (long)n_sys);
  const size_t rows = n_sys * N;
  size_t need = (size_t)nw * 3 * N + (hf ? 6 : 3) * rows + 256;
  if (hf) need += (a.X ? 0 : rows) + (a.dX ? 0 : rows) + (a.gam ? 0 : n_sys);
  if (ensure_long_ws(ctx, need * sizeof(double)) != 0) return hipErrorOutOfMemory;
  double* w = static_cast<double*>(ctx->long_ws);
  double* work = w; w += (size_t)nw * 3 * N;
  double* g = w; w += rows; double* c = w; w += rows; double* f = w; w += rows;
  double *gt = nullptr, *ct = nullptr, *ft = nullptr, *Xw = a.X, *dXw = a.dX, *gamw = a.gam;
  if (hf) {
    gt = w; w += rows; ct = w; w += rows; ft = w; w += rows;
    if (!Xw) { Xw = w; w += rows; }
    if (!dXw) { dXw = w; w += rows; }
    if (!gamw) { gamw = w; w += n_sys; }
  }
  hipError_t e = ibs::launch_assemble_long(a, g, c, f, gt, ct, ft, st);
  if (e != hipSuccess) return e;
  ibs::LongGcfArgs la{};
  la.n_sys = (long)n_sys; la.N = a.N; la.h = a.h; la.g = g; la.c = c; la.f = f; la.gh = nullptr; la.f32 = 0; la.ld = (long)N;
  la.lam = a.lam; la.gam = gamw; la.X = Xw; la.dX = dXw; la.info = a.info; la.work = work; la.n_waves = nw;
  e = ibs::launch_gcf_long(la, st);
  if (e != hipSuccess) return e;
  if (hf) {
    hipLaunchKernelGGL(k_hf_grad, dim3((unsigned)((n_sys + 3) / 4)), dim3(256), 0, st, (long)n_sys, a.N, (long)N, Xw, dXw, f, gt, ct, ft, gamw, a.dth0);
    e = hipGetLastError();
  }
  return e;
}

static int gamma_scan_impl(ibs_ctx* ctx, int32_t n_lines, int32_t n_theta0, int32_t N, double h,
                           const double* bmag, const double* gradpar, const double* cvdrift, const double* cvdrift0,
                           const double* gds2, const double* gds21, const double* gds22, int64_t ld,
                           const double* dPdrho, const double* theta0, double* gam, double* lam, double* X,
                           double* dX, double* dth0, int32_t* info, int32_t mem, const double* lam_guess,
                           double guess_width, int32_t n_surf = 0, double* pack = nullptr, bool t0_per_line = false) {
  // t0_per_line: n_theta0 = 1 and theta0[n_lines] holds one value per line (ibs_gamma_points_f64); one wave per system
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (lam_guess && !(guess_width > 0)) return fail(IBS_ERR_ARG, "guess_width must be > 0");
  if (pack && (mem != IBS_MEM_DEVICE || n_surf <= 0 || n_lines % n_surf != 0 || !gam))
    return fail(IBS_ERR_ARG, "fused argmax: device pointers, gam output and n_lines %% n_surf == 0 required (n_lines=%d n_surf=%d)", n_lines, n_surf);
  if (n_lines < 0 || n_theta0 < 0 || !bmag || !gradpar || !cvdrift || !cvdrift0 || !gds2 || !gds21 || !gds22 ||
      !dPdrho || !theta0 || ld < N)
    return fail(IBS_ERR_ARG, "bad arguments (n_lines=%d n_theta0=%d ld=%lld N=%d)", n_lines, n_theta0, (long long)ld, N);
  if (int r = check_grid(N, h, true)) return r;
  if (n_lines == 0 || n_theta0 == 0) return 0;
  const bool lng = is_long(N);                 // grids beyond 2050 points: launch_scan_long
  int M = lng ? 1 : rows_per_lane(N);
  auto fn = ibs::launch_table().scan_f64[M];
  if (!fn) return fail(IBS_ERR_UNSUPPORTED, "no kernel built for rows-per-lane M=%d (N=%d)", M, N);
  ON_DEVICE(ctx);
  ibs::ScanArgs<double> a{};
  a.n_lines = n_lines; a.n_theta0 = n_theta0; a.N = N; a.h = h; a.ld = ld; a.wpb = 1;
  a.t0_stride = t0_per_line ? 1 : 0;
  int G = 1;
  if (lng) { fn = &launch_scan_long; g_long_ctx = ctx; }
  else {
  const size_t per_arr = (size_t)ibs::lds_pitch(N) * sizeof(double);
  if (8 * per_arr > (size_t)ctx->lds_per_block) return fail(IBS_ERR_UNSUPPORTED, "N=%d does not fit the LDS staging", N);
  int cap = ibs::scan_max_threads(M) / 64;
  decltype(fn) fn_g_chain = nullptr;     // chained / warm-started sub-wave kernel of the same (P, M)
  {
    const int P = t0_per_line ? 64 : pick_lanes(ctx, N, (long)n_lines * n_theta0);
    if (P != 64 && n_theta0 % (64 / P) == 0) {
      const int Mg = (N - 2 + P - 1) / P;
      auto fg = ibs::launch_table().scan_f64_g[P == 32 ? 0 : 1][Mg];
      auto fgc = ibs::launch_table().scan_chain_f64_g[P == 32 ? 0 : 1][Mg];
      if (fg && (fgc || !lam_guess)) { fn = fg; fn_g_chain = fgc; M = Mg; G = 64 / P; cap = ibs::scan_max_threads_g(Mg) / 64; }
    }
  }
  // wpb = waves per block; each wave solves G theta0 values of the block's line
  int wpb = (int)(((size_t)ctx->lds_per_block - 7 * per_arr) / (per_arr * G));
  if (wpb > cap) wpb = cap;
  const int waves_per_line = (n_theta0 + G - 1) / G;
  if (wpb > waves_per_line) wpb = waves_per_line;
  if (wpb < 1) return fail(IBS_ERR_UNSUPPORTED, "N=%d does not fit the LDS staging", N);
  // small batches: spread the waves over all CUs (the solver is issue-bound, one wave per SIMD is
  // the fastest placement) instead of packing a line's theta0 values onto one CU
  {
    const long waves = (long)n_lines * waves_per_line;
    long per_blk = waves / ctx->n_cu;
    if (per_blk < 1) per_blk = 1;
    if (per_blk < wpb) wpb = (int)per_blk;
  }
  // balance the waves over the blocks of a line
  const int nblk = (waves_per_line + wpb - 1) / wpb;
  wpb = (waves_per_line + nblk - 1) / nblk;
  a.wpb = wpb;
  // A chain shortens the blocks (a line's waves = theta0 slots / chain) while every block still stages the whole line:
  // the LDS then limits the waves per CU.  Shorten the chain until the blocks that fit a CU hold as many waves as the
  // registers allow (tools/batch_sweep.py, 8 theta0 per line, N = 513, 65,536 solves: chain 4 = one wave per block =
  // 4 waves per CU: 6.5e7 solves/s; chain 2: 1.0e8).
  auto fit_chain = [&](int chain, int slots, int rows, int cap_waves) {
    const int occ = rows <= 4 ? 4 : (rows <= 8 ? 3 : (rows <= 16 ? 2 : 1));        // waves per SIMD the VGPRs allow
    const long blocks_per_cu = (long)((size_t)ctx->lds_per_block / (7 * per_arr));
    while (chain > 1) {
      long w = (slots + chain - 1) / chain;
      if (w > cap_waves) w = cap_waves;
      if (blocks_per_cu * w >= 4L * occ) break;
      chain >>= 1;
    }
    return chain;
  };
  // Batches much larger than the chip (one wave per system, no caller-supplied guesses): chain consecutive theta0
  // values of a line through one wave, each solve warm-started from the previous eigenvalue (k_gamma_scan_chain).
  // 4 per wave once that still leaves two waves per SIMD, 2 from there down to two waves of chained work per SIMD
  // (tools/bench_chain_sizes.py).  option scan_chain = n overrides.
  if (G > 1 && fn_g_chain) {
    // sub-wave kernels: the same chain over the theta0 slots of a group; caller-supplied guesses go through the same
    // kernel with a chain of one
    const int slots = n_theta0 / G;                       // theta0 values per group position of a line
    const long waves = (long)n_lines * slots, simds = 4L * ctx->n_cu;
    int chain = 1;
    if (!lam_guess) {
      if (slots >= 4 && waves >= 8 * simds) chain = 4;
      else if (slots >= 2 && waves >= 4 * simds) chain = 2;
      chain = fit_chain(chain, slots, M, cap);
      if (ctx->opt.scan_chain >= 1 && ctx->opt.scan_chain <= slots) chain = ctx->opt.scan_chain;
    }
    if (chain > 1 || lam_guess) {
      const bool need_x = X || dX;
      const int wpl = (slots + chain - 1) / chain;          // waves per line
      long w = need_x ? (long)(((size_t)ctx->lds_per_block - 7 * per_arr) / (per_arr * G)) : cap;
      if (w > cap) w = cap;
      if (w > wpl) w = wpl;
      if (w >= 1) {
        const int nb = (wpl + (int)w - 1) / (int)w;
        a.wpb = (wpl + nb - 1) / nb;
        a.chain = chain;
        a.chain_w1 = ctx->opt.chain_w1; a.chain_w2 = ctx->opt.chain_w2;
        fn = fn_g_chain;
      } else if (lam_guess) {
        return fail(IBS_ERR_UNSUPPORTED, "N=%d does not fit the LDS staging", N);
      }
    }
  }
  if (G == 1 && !lam_guess && !t0_per_line) {
    const long waves = (long)n_lines * n_theta0, simds = 4L * ctx->n_cu;
    int chain = 1;
    if (n_theta0 >= 8 && waves >= 8 * simds) chain = 4;
    else if (n_theta0 >= 4 && waves >= 4 * simds) chain = 2;      // (N = 1025: 2,048 solves 70 vs 95 us, 4,096: 111 vs 105 us)
    chain = fit_chain(chain, n_theta0, M, cap);
    if (ctx->opt.scan_chain >= 1 && ctx->opt.scan_chain <= n_theta0) chain = ctx->opt.scan_chain;
    auto fc = ibs::launch_table().scan_chain_f64[M];
    if (chain > 1 && fc) {
      const bool need_x = (M < 3) || X || dX;
      const int wpl = (n_theta0 + chain - 1) / chain;
      long w = need_x ? (long)(((size_t)ctx->lds_per_block - 7 * per_arr) / per_arr) : cap;
      if (w > cap) w = cap;
      if (w > wpl) w = wpl;
      if (w >= 1) {
        const int nb = (wpl + (int)w - 1) / (int)w;
        a.wpb = (wpl + nb - 1) / nb;
        a.chain = chain;
        a.chain_w1 = ctx->opt.chain_w1; a.chain_w2 = ctx->opt.chain_w2;     // defaults 0.25 / 1.0 measured on the NCSX shapes (tools/bench_chain.py): 9.2 sweeps per solve instead of 15.7
        fn = fc;
      }
    }
  }
  }   // (!lng)
  const size_t n_sys = (size_t)n_lines * n_theta0;
  if (mem == IBS_MEM_HOST) {
    const size_t in_elems = (size_t)n_lines * ld, out_elems = n_sys * N;
    const size_t n_t0_vals = t0_per_line ? (size_t)n_lines : (size_t)n_theta0;
    size_t need = 7 * pad256(in_elems * 8) + pad256(n_lines * 8) + pad256(n_t0_vals * 8) + 4 * pad256(n_sys * 8) +
                  2 * pad256(out_elems * 8) + pad256(n_sys * 4) + 8192;
    if (int r = ensure_ws(ctx, need)) return r;
    Arena ar(ctx);
    HostStage hs(ctx, need);
    const double* src[7] = {bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22};
    double* dev[7];
    for (int k = 0; k < 7; ++k) {
      dev[k] = ar.take<double>(in_elems);
      HIPCHK(hs.up(src[k], in_elems * 8, dev[k]));
    }
    double* ddP = ar.take<double>(n_lines); double* dt0 = ar.take<double>(n_t0_vals);
    HIPCHK(hs.up(dPdrho, (size_t)n_lines * 8, ddP));
    HIPCHK(hs.up(theta0, n_t0_vals * 8, dt0));
    double* dguess = lam_guess ? ar.take<double>(n_sys) : nullptr;
    if (lam_guess) HIPCHK(hs.up(lam_guess, n_sys * 8, dguess));
    HIPCHK(hs.flush_in());
    double* dgam = ar.take<double>(n_sys); double* dlam = ar.take<double>(n_sys); double* dd = ar.take<double>(n_sys);
    double* dX_ = X ? ar.take<double>(out_elems) : nullptr; double* ddX = dX ? ar.take<double>(out_elems) : nullptr;
    int* d_info = ar.take<int>(n_sys); int* d_nbad = ar.take<int>(1);
    a.bmag = dev[0]; a.gradpar = dev[1]; a.cvdrift = dev[2]; a.cvdrift0 = dev[3]; a.gds2 = dev[4]; a.gds21 = dev[5]; a.gds22 = dev[6];
    a.dPdrho = ddP; a.theta0 = dt0; a.gam = dgam; a.lam = dlam; a.X = dX_; a.dX = ddX; a.dth0 = dth0 ? dd : nullptr; a.info = d_info;
    if (lam_guess) { a.lam_guess = dguess; a.guess_width = guess_width; }
    HIPCHK(fn(a, ctx->stream));
    flag_sigma<double>(ctx, (long)n_sys, a.lam, a.info);
    HIPCHK(hipMemsetAsync(d_nbad, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_count_status, dim3((unsigned)((n_sys + 255) / 256)), dim3(256), 0, ctx->stream, (long)n_sys, d_info, d_nbad);
    if (gam) HIPCHK(hs.down(gam, dgam, n_sys * 8));
    if (lam) HIPCHK(hs.down(lam, dlam, n_sys * 8));
    if (dth0) HIPCHK(hs.down(dth0, dd, n_sys * 8));
    if (X) HIPCHK(hs.down(X, dX_, out_elems * 8));
    if (dX) HIPCHK(hs.down(dX, ddX, out_elems * 8));
    if (info) HIPCHK(hs.down(info, d_info, n_sys * 4));
    int nbad = 0;
    HIPCHK(hs.down(&nbad, d_nbad, sizeof(int)));
    HIPCHK(hs.flush_out());
    return nbad;
  }
  a.bmag = bmag; a.gradpar = gradpar; a.cvdrift = cvdrift; a.cvdrift0 = cvdrift0; a.gds2 = gds2; a.gds21 = gds21; a.gds22 = gds22;
  a.dPdrho = dPdrho; a.theta0 = theta0; a.gam = gam; a.lam = lam; a.X = X; a.dX = dX; a.dth0 = dth0; a.info = info;
  a.lam_guess = lam_guess; a.guess_width = guess_width;
  if (pack && fn == ibs::launch_table().scan_f64[M] && G == 1) {
    // one launch: the block that completes a surface reduces it (k_gamma_scan's epilogue)
    ibs_ctx::SurfCounters* sc = nullptr;
    for (auto& e : ctx->surf_counters) if (e.stream == ctx->stream) { sc = &e; break; }
    if (!sc) { ctx->surf_counters.push_back(ibs_ctx::SurfCounters{ctx->stream, nullptr, 0}); sc = &ctx->surf_counters.back(); }
    if (sc->n < n_surf) {
      if (sc->buf) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(sc->buf)); sc->buf = nullptr; sc->n = 0; }
      const int cap_n = n_surf > 1024 ? n_surf : 1024;
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&sc->buf), (size_t)cap_n * sizeof(int)));
      HIPCHK(hipMemsetAsync(sc->buf, 0, (size_t)cap_n * sizeof(int), ctx->stream));
      sc->n = cap_n;
    }
    a.lines_per_surf = n_lines / n_surf; a.surf_counter = sc->buf; a.pack = pack;
    {
      const long nblocks = (long)((n_theta0 + a.wpb - 1) / a.wpb) * n_lines;
      a.pack_mode = (nblocks <= ctx->n_cu && ctx->opt.pack_mode != 2) ? 1 : 2;
      if (ctx->opt.pack_mode == 1) a.pack_mode = 1;
    }
    HIPCHK(fn(a, ctx->stream));
    flag_sigma<double>(ctx, (long)n_sys, a.lam, a.info);
    return 0;
  }
  HIPCHK(fn(a, ctx->stream));
  flag_sigma<double>(ctx, (long)n_sys, a.lam, a.info);
  if (pack) {                                    // chained / sub-wave scan kernels: the reduction is a second launch
    const int n_per = (n_lines / n_surf) * n_theta0;
    hipLaunchKernelGGL(k_surface_argmax, dim3(n_surf), dim3(argmax_threads(n_per)), 0, ctx->stream, n_per, gam, (int*)nullptr, (double*)nullptr, pack);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

int ibs_gamma_scan_f64(ibs_ctx* ctx, int32_t n_lines, int32_t n_theta0, int32_t N, double h,
                       const double* bmag, const double* gradpar, const double* cvdrift, const double* cvdrift0,
                       const double* gds2, const double* gds21, const double* gds22, int64_t ld,
                       const double* dPdrho, const double* theta0, double* gam, double* lam, double* X,
                       double* dX, double* dth0, int32_t* info, int32_t mem) {
  return gamma_scan_impl(ctx, n_lines, n_theta0, N, h, bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, ld, dPdrho,
                         theta0, gam, lam, X, dX, dth0, info, mem, nullptr, 0.0);
}

int ibs_gamma_scan_warm_f64(ibs_ctx* ctx, int32_t n_lines, int32_t n_theta0, int32_t N, double h,
                            const double* bmag, const double* gradpar, const double* cvdrift, const double* cvdrift0,
                            const double* gds2, const double* gds21, const double* gds22, int64_t ld,
                            const double* dPdrho, const double* theta0, const double* lam_guess, double guess_width,
                            double* gam, double* lam, double* X, double* dX, double* dth0, int32_t* info, int32_t mem) {
  if (!lam_guess) return fail(IBS_ERR_ARG, "lam_guess is null");
  return gamma_scan_impl(ctx, n_lines, n_theta0, N, h, bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, ld, dPdrho,
                         theta0, gam, lam, X, dX, dth0, info, mem, lam_guess, guess_width);
}

int ibs_gamma_scan_argmax_f64(ibs_ctx* ctx, int32_t n_lines, int32_t n_theta0, int32_t N, double h,
                              const double* bmag, const double* gradpar, const double* cvdrift, const double* cvdrift0,
                              const double* gds2, const double* gds21, const double* gds22, int64_t ld,
                              const double* dPdrho, const double* theta0, int32_t n_surf, double* gam, double* lam,
                              double* pack, int32_t* info) {
  if (!pack) return fail(IBS_ERR_ARG, "pack is null");
  return gamma_scan_impl(ctx, n_lines, n_theta0, N, h, bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, ld, dPdrho,
                         theta0, gam, lam, nullptr, nullptr, nullptr, info, IBS_MEM_DEVICE, nullptr, 0.0, n_surf, pack);
}

int ibs_gamma_points_f64(ibs_ctx* ctx, int32_t n_pts, int32_t N, double h, const double* bmag, const double* gradpar,
                         const double* cvdrift, const double* cvdrift0, const double* gds2, const double* gds21,
                         const double* gds22, int64_t ld, const double* dPdrho, const double* theta0, double* gam,
                         double* lam, double* X, double* dX, double* dgam_dtheta0, int32_t* info, int32_t mem) {
  return gamma_scan_impl(ctx, n_pts, 1, N, h, bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, ld, dPdrho, theta0, gam,
                         lam, X, dX, dgam_dtheta0, info, mem, nullptr, 0.0, 0, nullptr, true);
}

int ibs_scan_starts_f64(ibs_ctx* ctx, int32_t n_surf, int32_t n_alpha, int32_t n_theta0, const double* alpha,
                        const double* theta0, const double* pack, double* start, double* sigma0, int32_t* n_bad) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (n_surf < 0 || n_alpha <= 0 || n_theta0 <= 0 || !alpha || !theta0 || !pack || !start || !n_bad) return fail(IBS_ERR_ARG, "bad arguments");
  if (n_surf == 0) return 0;
  ON_DEVICE(ctx);
  hipLaunchKernelGGL(k_scan_starts, dim3((unsigned)((n_surf + 127) / 128)), dim3(128), 0, ctx->stream, n_surf, n_alpha, n_theta0,
                     alpha, theta0, pack, start, sigma0, n_bad);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- objective + Hellmann-Feynman gradient on grids beyond 2050 points (utils.py:1632-1728 takes any length): composed from the
// long-grid pieces -- dPdrho of the 3 n lines (k_line_dPdrho), their (g, c, f) rows and theta0-tangents at the point's theta0
// (k_assemble_gcf_long), the centre lines' rows and the alpha-tangents (right - left) / del_alpha packed per point, ONE long-grid
// solve per point with eigenfunction (k_solve_gcf_long), two Hellmann-Feynman sums (k_hf_grad).
__global__ void k_grad_long_t0(int n_pts, const double* __restrict__ theta0, double* __restrict__ th3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * n_pts) th3[i] = theta0[i / 3];
}
__global__ void __launch_bounds__(256) k_grad_long_pack(int n_pts, int N, double inv_del, const double* __restrict__ G, const double* __restrict__ C,
                                                        const double* __restrict__ F, const double* __restrict__ GT, const double* __restrict__ CT,
                                                        const double* __restrict__ FT, double* gC, double* cC, double* fC, double* gtC, double* ctC,
                                                        double* ftC, double* gaC, double* caC, double* faC) {
  const int p = blockIdx.y;
  const size_t l = (size_t)(3 * p) * N, m = l + N, r = m + N, o = (size_t)p * N;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
    gC[o + j] = G[m + j]; cC[o + j] = C[m + j]; fC[o + j] = F[m + j];
    gtC[o + j] = GT[m + j]; ctC[o + j] = CT[m + j]; ftC[o + j] = FT[m + j];                 // utils.py:1669-1673
    gaC[o + j] = (G[r + j] - G[l + j]) * inv_del; caC[o + j] = (C[r + j] - C[l + j]) * inv_del;      // utils.py:1705-1719
    faC[o + j] = (F[r + j] - F[l + j]) * inv_del;
  }
}
__global__ void k_grad_long_out(int n_pts, const double* __restrict__ gam, const double* __restrict__ ja, const double* __restrict__ jt,
                                double* val, double* jac, double* gam_o, double* da_o, double* dt_o) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  val[p] = -gam[p]; jac[2 * p] = -ja[p]; jac[2 * p + 1] = -jt[p];                            // utils.py:1728
  if (gam_o) gam_o[p] = gam[p];
  if (da_o) da_o[p] = ja[p];
  if (dt_o) dt_o[p] = jt[p];
}
static hipError_t launch_grad_long(const ibs::GradArgs<double>& a, hipStream_t st) {
  ibs_ctx* ctx = g_long_ctx;
  const size_t n = (size_t)a.n_pts, N = (size_t)a.N, rows = n * N;
  const int nw = long_waves(ctx, (long)n);
  const size_t need = (size_t)nw * 3 * N + 6 * 3 * rows + 11 * rows + 6 * n + 3 * n + 512;
  if (ensure_long_ws(ctx, need * sizeof(double)) != 0) return hipErrorOutOfMemory;
  double* w = static_cast<double*>(ctx->long_ws);
  auto take = [&](size_t k) { double* q = w; w += k; return q; };
  double* work = take((size_t)nw * 3 * N);
  double *G = take(3 * rows), *C = take(3 * rows), *F = take(3 * rows), *GT = take(3 * rows), *CT = take(3 * rows), *FT = take(3 * rows);
  double *gC = take(rows), *cC = take(rows), *fC = take(rows), *gtC = take(rows), *ctC = take(rows), *ftC = take(rows);
  double *gaC = take(rows), *caC = take(rows), *faC = take(rows), *Xw = take(rows), *dXw = take(rows);
  double *dP3 = take(3 * n), *th3 = take(3 * n), *gamw = take(n), *lamw = take(n), *ja = take(n), *jt = take(n);
  const long lds = a.ld;                                                  // geo: [n_pts][3][8][ld]
  hipError_t e = ibs::launch_line_dPdrho((int)(3 * n), a.N, 8 * lds, (size_t)lds, a.geo, dP3, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_grad_long_t0, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, st, a.n_pts, a.theta0, th3);
  ibs::ScanArgs<double> sa{};
  sa.n_lines = (int)(3 * n); sa.n_theta0 = 1; sa.N = a.N; sa.h = a.h; sa.ld = 8 * lds;
  sa.bmag = a.geo; sa.gradpar = a.geo + lds; sa.cvdrift = a.geo + 2 * lds; sa.cvdrift0 = a.geo + 3 * lds;
  sa.gds2 = a.geo + 4 * lds; sa.gds21 = a.geo + 5 * lds; sa.gds22 = a.geo + 6 * lds;
  sa.dPdrho = dP3; sa.theta0 = th3; sa.t0_stride = 1;
  e = ibs::launch_assemble_long(sa, G, C, F, GT, CT, FT, st);
  if (e != hipSuccess) return e;
  int bx = (a.N + 255) / 256;
  if (bx > 16) bx = 16;
  hipLaunchKernelGGL(k_grad_long_pack, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, a.n_pts, a.N, 1.0 / a.del_alpha, G, C, F, GT, CT, FT,
                     gC, cC, fC, gtC, ctC, ftC, gaC, caC, faC);
  ibs::LongGcfArgs la{};
  la.n_sys = (long)n; la.N = a.N; la.h = a.h; la.g = gC; la.c = cC; la.f = fC; la.gh = nullptr; la.f32 = 0; la.ld = (long)N;
  la.lam = lamw; la.gam = gamw; la.X = Xw; la.dX = dXw; la.info = a.info; la.work = work; la.n_waves = nw;
  e = ibs::launch_gcf_long(la, st);
  if (e != hipSuccess) return e;
  const unsigned hb = (unsigned)((n + 3) / 4);
  hipLaunchKernelGGL(k_hf_grad, dim3(hb), dim3(256), 0, st, (long)n, a.N, (long)N, Xw, dXw, fC, gtC, ctC, ftC, gamw, jt);
  hipLaunchKernelGGL(k_hf_grad, dim3(hb), dim3(256), 0, st, (long)n, a.N, (long)N, Xw, dXw, fC, gaC, caC, faC, gamw, ja);
  hipLaunchKernelGGL(k_grad_long_out, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.n_pts, gamw, ja, jt, a.val, a.jac, a.gam, a.dalpha, a.dth0);
  return hipGetLastError();
}

int ibs_obj_w_grad_f64(ibs_ctx* ctx, int32_t n_pts, int32_t N, double h, const double* geo, int64_t ld,
                       const double* theta0, double del_alpha, double* val, double* jac, int32_t* info,
                       int32_t mem) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (n_pts < 0 || !geo || !theta0 || !val || !jac || ld < N || !(del_alpha > 0)) return fail(IBS_ERR_ARG, "bad arguments");
  if (int r = check_grid(N, h, true)) return r;
  if (n_pts == 0) return 0;
  const bool lng = is_long(N);                 // grids beyond 2050 points: launch_grad_long
  const int M = lng ? 1 : rows_per_lane(N);
  auto fn = lng ? &launch_grad_long : ibs::launch_table().grad_f64[M];
  if (!fn) return fail(IBS_ERR_UNSUPPORTED, "no kernel built for rows-per-lane M=%d (N=%d)", M, N);
  if (lng) g_long_ctx = ctx;
  ON_DEVICE(ctx);
  const size_t per_wave = (size_t)8 * ibs::lds_pitch(lng ? 66 : N) * sizeof(double);
  int wpb = (int)((size_t)ctx->lds_per_block / per_wave);
  if (wpb > 4) wpb = 4;
  if (wpb < 1) return fail(IBS_ERR_UNSUPPORTED, "N=%d does not fit the LDS staging", N);
  {
    long per_blk = (long)n_pts / ctx->n_cu;
    if (per_blk < 1) per_blk = 1;
    if (per_blk < wpb) wpb = (int)per_blk;
  }
  ibs::GradArgs<double> a{};
  a.n_pts = n_pts; a.N = N; a.h = h; a.ld = ld; a.del_alpha = del_alpha; a.wpb = wpb;
  const size_t geo_elems = (size_t)n_pts * 3 * 8 * ld;
  const bool host = (mem == IBS_MEM_HOST);
  size_t need = 3 * pad256((size_t)n_pts * 8) + pad256((size_t)n_pts * 4) + 4096;
  if (host) need += pad256(geo_elems * 8) + pad256((size_t)n_pts * 8) * 2 + pad256((size_t)n_pts * 16) + pad256((size_t)n_pts * 4) + 4096;
  if (int r = ensure_ws(ctx, need)) return r;
  Arena ar(ctx);
  a.gam = ar.take<double>(n_pts); a.dalpha = ar.take<double>(n_pts); a.dth0 = ar.take<double>(n_pts);
  int* d_info = ar.take<int>(n_pts);
  if (host) {
    double* dgeo = ar.take<double>(geo_elems); double* dt0 = ar.take<double>(n_pts);
    double* dval = ar.take<double>(n_pts); double* djac = ar.take<double>((size_t)2 * n_pts);
    int* d_info_h = ar.take<int>(n_pts);            // (behind the inputs: the outputs come back as one contiguous span)
    int* d_nbad = ar.take<int>(1);
    HostStage hs(ctx, need);
    HIPCHK(hs.up(geo, geo_elems * 8, dgeo));
    HIPCHK(hs.up(theta0, (size_t)n_pts * 8, dt0));
    HIPCHK(hs.flush_in());
    a.geo = dgeo; a.theta0 = dt0; a.val = dval; a.jac = djac; a.info = d_info_h;
    HIPCHK(fn(a, ctx->stream));
    HIPCHK(hipMemsetAsync(d_nbad, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_count_status, dim3((unsigned)((n_pts + 255) / 256)), dim3(256), 0, ctx->stream, (long)n_pts, d_info_h, d_nbad);
    HIPCHK(hs.down(val, dval, (size_t)n_pts * 8));
    HIPCHK(hs.down(jac, djac, (size_t)n_pts * 16));
    if (info) HIPCHK(hs.down(info, d_info_h, (size_t)n_pts * 4));
    int nbad = 0;
    HIPCHK(hs.down(&nbad, d_nbad, sizeof(int)));
    HIPCHK(hs.flush_out());
    return nbad;
  }
  a.geo = geo; a.theta0 = theta0; a.val = val; a.jac = jac; a.info = info ? info : d_info;
  HIPCHK(fn(a, ctx->stream));
  return 0;
}

// The mode rows a caller hands to the geometry entry points ({first mode, count} per row): 1 = every row lies inside the mode
// list and none has more pair indices about its centre than the row kernels' table image holds (ibs::kGeoMaxPairs = 64: a
// row of up to 129 modes centred on n = 0 -- VMEC's -ntor..ntor with ntor <= 64 -- or up to 65 modes otherwise; the centre
// rule is k_geo_prepare's), 0 = valid but a row is longer (the call then runs on the one-sincos-per-mode kernel, which takes
// any ordering), < 0 = a row outside the mode list (error set).  Host tables are checked on every call; device-resident
// ones are copied back ONCE per (pointers, counts, steps) -- a synchronisation the first call with a table set pays.
static int geo_rows_fit_host(const int32_t* rows, int nrows, int nmodes, const double* xn, double dn, const char* which) {
  int fit = 1;
  // (rows must not overlap: the table images of the row kernels are sized for ONE pair / group entry per mode -- 24 rows of
  //  {0, 65} over 65 modes would each be "inside the list" and together write 24 times the image)
  std::vector<char> taken((size_t)nmodes, 0);
  for (int r = 0; r < nrows; ++r) {
    const int first = rows[2 * r], cnt = rows[2 * r + 1];
    if (first < 0 || cnt < 1 || (long)first + cnt > nmodes) return fail(IBS_ERR_ARG, "%s[%d] = {%d, %d} lies outside the %d modes", which, r, first, cnt, nmodes);
    for (int k = first; k < first + cnt; ++k) {
      if (taken[k]) return fail(IBS_ERR_ARG, "%s[%d] = {%d, %d} overlaps an earlier row at mode %d: every mode may belong to one row only", which, r, first, cnt, k);
      taken[k] = 1;
    }
    if (cnt <= ibs::kGeoMaxPairs + 1) continue;
    int k0 = 0;
    if (dn != 0.0) {
      const double n0 = xn[first], kz = -n0 / dn;
      const int kr = (int)(kz + (kz >= 0 ? 0.5 : -0.5));
      if (kr >= 0 && kr < cnt && std::fabs(n0 + kr * dn) <= 1e-9 * std::fabs(dn)) k0 = kr;
    }
    if (std::max(k0, cnt - 1 - k0) > ibs::kGeoMaxPairs) fit = 0;
  }
  return fit;
}
static int geo_rows_fit(ibs_ctx* ctx, int mnmax, int mnmax_nyq, const double* xn, const double* xn_nyq, int nrows_mn,
                        const int32_t* rows_mn, int nrows_nyq, const int32_t* rows_nyq, double dn_mn, double dn_nyq, bool host) {
  if (host) {
    const int a = geo_rows_fit_host(rows_mn, nrows_mn, mnmax, xn, dn_mn, "rows_mn");
    if (a < 0) return a;
    const int b = geo_rows_fit_host(rows_nyq, nrows_nyq, mnmax_nyq, xn_nyq, dn_nyq, "rows_nyq");
    return b < 0 ? b : a;            // (only the first list feeds the (P, Q) tables; the second one's pair tables hold one entry per mode)
  }
  for (const auto& c : ctx->rows_seen)
    if (c.r1 == rows_mn && c.r2 == rows_nyq && c.xn == xn && c.xnq == xn_nyq && c.n1 == nrows_mn && c.n2 == nrows_nyq && c.mn == mnmax &&
        c.mnq == mnmax_nyq && c.d1 == dn_mn && c.d2 == dn_nyq && c.r1) return c.verdict;
  auto& c = ctx->rows_seen[ctx->rows_seen_next];
  ctx->rows_seen_next = (ctx->rows_seen_next + 1) % ibs_ctx::kRowsSeen;
  std::vector<int32_t> r1((size_t)2 * nrows_mn), r2((size_t)2 * nrows_nyq);
  std::vector<double> x1(mnmax), x2(mnmax_nyq);
  HIPCHK(hipMemcpyAsync(r1.data(), rows_mn, r1.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(r2.data(), rows_nyq, r2.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(x1.data(), xn, x1.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(x2.data(), xn_nyq, x2.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int a = geo_rows_fit_host(r1.data(), nrows_mn, mnmax, x1.data(), dn_mn, "rows_mn");
  if (a < 0) return a;
  const int b = geo_rows_fit_host(r2.data(), nrows_nyq, mnmax_nyq, x2.data(), dn_nyq, "rows_nyq");
  if (b < 0) return b;
  c.r1 = rows_mn; c.r2 = rows_nyq; c.xn = xn; c.xnq = xn_nyq; c.n1 = nrows_mn; c.n2 = nrows_nyq; c.mn = mnmax; c.mnq = mnmax_nyq;
  c.d1 = dn_mn; c.d2 = dn_nyq; c.verdict = a;
  return c.verdict;
}

// workspace for the prepared table images of the geometry row kernels (ibs_geometry.hip): the set of the form `f`
static size_t geo_img_bytes(const ibs::GeoArgs& a, int lpp) {
  return ibs::geo_rows_usable(a, lpp) ? pad256((size_t)a.n_surf * ibs::geo_image_doubles(a, lpp) * sizeof(double)) : 0;
}

int ibs_fieldline_geometry_f64(ibs_ctx* ctx, int32_t n_surf, int32_t mnmax, int32_t mnmax_nyq, const double* xm,
                               const double* xn, const double* xm_nyq, const double* xn_nyq, const double* tab_mn,
                               const double* tab_nyq, const double* scal, int32_t n_lines, const int32_t* line_surf,
                               const double* line_alpha, int32_t N, const double* theta, int64_t ld, double* geo,
                               double* dPdrho, int32_t nrows_mn, const int32_t* rows_mn, int32_t nrows_nyq,
                               const int32_t* rows_nyq, double dn_mn, double dn_nyq, int32_t mem) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (nrows_mn < 0 || nrows_nyq < 0 || (nrows_mn > 0 && !rows_mn) || (nrows_nyq > 0 && !rows_nyq))
    return fail(IBS_ERR_ARG, "bad row tables");
  if (n_surf <= 0 || mnmax <= 0 || mnmax_nyq <= 0 || n_lines < 0 || N < 2 || ld < N || !xm || !xn || !xm_nyq || !xn_nyq ||
      !tab_mn || !tab_nyq || !scal || !line_surf || !line_alpha || !theta || !geo)
    return fail(IBS_ERR_ARG, "bad arguments");
  if (n_lines == 0) return 0;
  ON_DEVICE(ctx);
  ibs::GeoArgs a{};
  a.n_surf = n_surf; a.mnmax = mnmax; a.mnmax_nyq = mnmax_nyq; a.n_lines = n_lines; a.N = N; a.ld = ld;
  a.lpp = ctx->opt.geo_lpp;
  bool rows = nrows_mn > 0 && nrows_nyq > 0;
  if (rows) {
    const int fit = geo_rows_fit(ctx, mnmax, mnmax_nyq, xn, xn_nyq, nrows_mn, rows_mn, nrows_nyq, rows_nyq, dn_mn, dn_nyq, mem == IBS_MEM_HOST);
    if (fit < 0) return fit;
    rows = fit == 1;
  }
  if (rows) { a.nrows_mn = nrows_mn; a.nrows_nyq = nrows_nyq; a.dn_mn = dn_mn; a.dn_nyq = dn_nyq; }
  a.form = ibs::geo_pick_usable(a, n_lines, N, ctx->n_cu);
  const size_t img_bytes = geo_img_bytes(a, a.form.lpp);
  if (mem == IBS_MEM_HOST) {
    for (int i = 0; i < n_lines; ++i)
      if (line_surf[i] < 0 || line_surf[i] >= n_surf) return fail(IBS_ERR_ARG, "line_surf[%d]=%d out of range", i, line_surf[i]);
    const size_t n_mn = (size_t)n_surf * 6 * mnmax, n_nyq = (size_t)n_surf * 7 * mnmax_nyq, n_geo = (size_t)8 * n_lines * ld;
    size_t need = pad256(n_mn * 8) + pad256(n_nyq * 8) + 2 * pad256((size_t)mnmax * 8) + 2 * pad256((size_t)mnmax_nyq * 8) +
                  pad256((size_t)n_surf * 48) + pad256((size_t)n_lines * 4) + 2 * pad256((size_t)n_lines * 8) +
                  pad256((size_t)N * 8) + pad256(n_geo * 8) + pad256((size_t)nrows_mn * 8) + pad256((size_t)nrows_nyq * 8) + img_bytes + 8192;
    if (int r = ensure_ws(ctx, need)) return r;
    Arena ar(ctx);
    auto up = [&](const void* src, size_t bytes, void* dst) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream); };
    double* d_xm = ar.take<double>(mnmax); double* d_xn = ar.take<double>(mnmax);
    double* d_xmq = ar.take<double>(mnmax_nyq); double* d_xnq = ar.take<double>(mnmax_nyq);
    double* d_mn = ar.take<double>(n_mn); double* d_nyq = ar.take<double>(n_nyq); double* d_sc = ar.take<double>((size_t)n_surf * 6);
    int* d_ls = ar.take<int>(n_lines); double* d_la = ar.take<double>(n_lines); double* d_th = ar.take<double>(N);
    double* d_geo = ar.take<double>(n_geo); double* d_dP = ar.take<double>(n_lines);
    HIPCHK(up(xm, (size_t)mnmax * 8, d_xm)); HIPCHK(up(xn, (size_t)mnmax * 8, d_xn));
    HIPCHK(up(xm_nyq, (size_t)mnmax_nyq * 8, d_xmq)); HIPCHK(up(xn_nyq, (size_t)mnmax_nyq * 8, d_xnq));
    HIPCHK(up(tab_mn, n_mn * 8, d_mn)); HIPCHK(up(tab_nyq, n_nyq * 8, d_nyq)); HIPCHK(up(scal, (size_t)n_surf * 48, d_sc));
    HIPCHK(up(line_surf, (size_t)n_lines * 4, d_ls)); HIPCHK(up(line_alpha, (size_t)n_lines * 8, d_la)); HIPCHK(up(theta, (size_t)N * 8, d_th));
    a.xm = d_xm; a.xn = d_xn; a.xm_nyq = d_xmq; a.xn_nyq = d_xnq; a.tab_mn = d_mn; a.tab_nyq = d_nyq; a.scal = d_sc;
    a.line_surf = d_ls; a.line_alpha = d_la; a.theta = d_th; a.geo = d_geo; a.dPdrho = d_dP;
    if (rows) {
      int* d_r1 = ar.take<int>((size_t)2 * nrows_mn); int* d_r2 = ar.take<int>((size_t)2 * nrows_nyq);
      HIPCHK(up(rows_mn, (size_t)nrows_mn * 8, d_r1)); HIPCHK(up(rows_nyq, (size_t)nrows_nyq * 8, d_r2));
      a.rows_mn = d_r1; a.rows_nyq = d_r2;
    }
    if (img_bytes) a.img[ibs::geo_lpp_index(a.form.lpp)] = ar.take<double>(img_bytes / sizeof(double));
    HIPCHK(ibs::launch_geometry(a, ctx->stream, ctx->n_cu));
    HIPCHK(hipMemcpyAsync(geo, d_geo, n_geo * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (dPdrho) HIPCHK(hipMemcpyAsync(dPdrho, d_dP, (size_t)n_lines * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  a.xm = xm; a.xn = xn; a.xm_nyq = xm_nyq; a.xn_nyq = xn_nyq; a.tab_mn = tab_mn; a.tab_nyq = tab_nyq; a.scal = scal;
  a.line_surf = line_surf; a.line_alpha = line_alpha; a.theta = theta; a.geo = geo; a.dPdrho = dPdrho;
  if (rows) { a.rows_mn = rows_mn; a.rows_nyq = rows_nyq; }
  if (img_bytes) {
    // a call whose lines touch few of the surfaces (a large table set worked through piece by piece) builds their images only
    const bool mark = n_surf >= 32 && (long)n_lines < 8L * n_surf;
    if (int r = ensure_ws(ctx, img_bytes + pad256((size_t)n_surf * sizeof(int)) + 4096)) return r;
    Arena ar(ctx);
    a.img[ibs::geo_lpp_index(a.form.lpp)] = ar.take<double>(img_bytes / sizeof(double));
    if (mark) a.surf_used = ar.take<int>(n_surf);
  }
  HIPCHK(ibs::launch_geometry(a, ctx->stream, ctx->n_cu));
  return 0;
}

int ibs_hf_grad_f64(ibs_ctx* ctx, int64_t n_sys, int32_t N, const double* X, const double* dX, const double* f,
                    const double* g_p, const double* c_p, const double* f_p, int64_t ld, const double* gam,
                    double* jac, int32_t mem) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (n_sys < 0 || !X || !dX || !f || !g_p || !c_p || !f_p || !gam || !jac || ld < N)
    return fail(IBS_ERR_ARG, "bad arguments");
  if (N < 3 || !(N & 1)) return fail(IBS_ERR_UNSUPPORTED, "N=%d: the Simpson rule of the reference needs N odd", N);
  if (n_sys == 0) return 0;
  ON_DEVICE(ctx);
  const dim3 grid((unsigned)((n_sys + 3) / 4));
  if (mem == IBS_MEM_HOST) {
    const size_t ne = (size_t)n_sys * ld;
    if (int r = ensure_ws(ctx, 6 * pad256(ne * 8) + 2 * pad256(n_sys * 8) + 4096)) return r;
    Arena ar(ctx);
    const double* src[6] = {X, dX, f, g_p, c_p, f_p};
    double* dev[6];
    for (int k = 0; k < 6; ++k) {
      dev[k] = ar.take<double>(ne);
      HIPCHK(hipMemcpyAsync(dev[k], src[k], ne * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    double* dgam = ar.take<double>(n_sys); double* djac = ar.take<double>(n_sys);
    HIPCHK(hipMemcpyAsync(dgam, gam, (size_t)n_sys * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_hf_grad, grid, dim3(256), 0, ctx->stream, (long)n_sys, N, (long)ld, dev[0], dev[1], dev[2], dev[3],
                       dev[4], dev[5], dgam, djac);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(jac, djac, (size_t)n_sys * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  hipLaunchKernelGGL(k_hf_grad, grid, dim3(256), 0, ctx->stream, (long)n_sys, N, (long)ld, X, dX, f, g_p, c_p, f_p, gam, jac);
  HIPCHK(hipGetLastError());
  return 0;
}

int ibs_sturm_count_f64(ibs_ctx* ctx, int64_t n_sys, int32_t N, double h, const double* g, const double* c,
                        const double* f, int64_t ld, const double* shift, int32_t* count, int32_t mem) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (n_sys < 0 || !g || !c || !f || !shift || !count || ld < N) return fail(IBS_ERR_ARG, "bad arguments");
  // the Sturm count itself has no Simpson stage: even N is fine here
  if (N < 66 || N > ibs::kMaxLongN) return fail(IBS_ERR_UNSUPPORTED, "N=%d outside [66, %d]", N, ibs::kMaxLongN);
  if (!(h > 0)) return fail(IBS_ERR_ARG, "h must be > 0");
  if (n_sys == 0) return 0;
  // forms: the prefix-product sweep (one wave per system, N <= 2050: the bandwidth kernel of rounds 1-5), and two division-form
  // kernels for any N (ibs_long.hip): lanes as systems (batches) and one wave per system (a handful of systems on a long grid)
  // By size: the sweep up to 2,050 points -- except where its lanes sit 128 / 256 bytes apart (16 / 32 rows per lane: 4.2 / 2.8 TB/s
  // against 6.3 at 8, 12, 24 rows) and the batch fills the chip with 64 systems per wave: there the division form is exact AND faster
  // (4.9-5.1 TB/s; tools/experiments/sturm_crossover.py)
  int form = ctx->opt.sturm_form;
  if (form == 0) {
    if (is_long(N)) form = n_sys >= 64 ? 2 : 3;
    else {
      const int Mr = rows_per_lane(N);
      form = ((Mr == 16 && n_sys >= 65536) || (Mr == 32 && n_sys >= 32768)) ? 2 : 1;
    }
  }
  if (form == 1 && is_long(N)) form = 2;
  const bool lng = form != 1;
  const int M = lng ? 1 : rows_per_lane(N);
  auto fn = form == 2 ? &ibs::launch_sturm_div : (form == 3 ? &ibs::launch_sturm_long : ibs::launch_table().sturm_f64[M]);
  if (!fn) return fail(IBS_ERR_UNSUPPORTED, "no kernel built for rows-per-lane M=%d (N=%d)", M, N);
  ON_DEVICE(ctx);
  const size_t per_wave = (size_t)N * sizeof(double);
  int wpb = lng ? 1 : (int)((size_t)ctx->lds_per_block / per_wave);
  if (wpb > 4) wpb = 4;
  if (wpb < 1) return fail(IBS_ERR_UNSUPPORTED, "N=%d needs %zu B of LDS per wave", N, per_wave);
  ibs::SturmArgs<double> a{};
  a.n_sys = n_sys; a.N = N; a.h = h; a.ld = ld; a.wpb = wpb;
  if (mem == IBS_MEM_HOST) {
    const size_t in_elems = (size_t)n_sys * ld;
    size_t need = 3 * pad256(in_elems * 8) + pad256(n_sys * 8) + pad256(n_sys * 4) + 4096;
    if (int r = ensure_ws(ctx, need)) return r;
    Arena ar(ctx);
    double* dg = ar.take<double>(in_elems); double* dc = ar.take<double>(in_elems); double* df = ar.take<double>(in_elems);
    double* ds = ar.take<double>(n_sys); int* dcount = ar.take<int>(n_sys);
    HIPCHK(hipMemcpyAsync(dg, g, in_elems * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dc, c, in_elems * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(df, f, in_elems * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ds, shift, n_sys * 8, hipMemcpyHostToDevice, ctx->stream));
    a.g = dg; a.c = dc; a.f = df; a.shift = ds; a.count = dcount;
    HIPCHK(fn(a, ctx->stream));
    HIPCHK(hipMemcpyAsync(count, dcount, n_sys * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  a.g = g; a.c = c; a.f = f; a.shift = shift; a.count = count;
  HIPCHK(fn(a, ctx->stream));
  return 0;
}

int ibs_surface_argmax_f64(ibs_ctx* ctx, int32_t n_surf, int32_t n_per, const double* gam, int32_t* idx,
                           double* val, int32_t mem) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (n_surf < 0 || n_per <= 0 || !gam || !idx || !val) return fail(IBS_ERR_ARG, "bad arguments");
  if (n_surf == 0) return 0;
  ON_DEVICE(ctx);
  if (mem == IBS_MEM_HOST) {
    const size_t ne = (size_t)n_surf * n_per;
    if (int r = ensure_ws(ctx, pad256(ne * 8) + pad256(n_surf * 8) + pad256(n_surf * 4) + 4096)) return r;
    Arena ar(ctx);
    double* dg = ar.take<double>(ne); double* dv = ar.take<double>(n_surf); int* di = ar.take<int>(n_surf);
    HIPCHK(hipMemcpyAsync(dg, gam, ne * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_surface_argmax, dim3(n_surf), dim3(argmax_threads(n_per)), 0, ctx->stream, n_per, dg, di, dv, (double*)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(idx, di, (size_t)n_surf * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(val, dv, (size_t)n_surf * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  hipLaunchKernelGGL(k_surface_argmax, dim3(n_surf), dim3(argmax_threads(n_per)), 0, ctx->stream, n_per, gam, idx, val, (double*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

int ibs_surface_argmax_pack_f64(ibs_ctx* ctx, int32_t n_surf, int32_t n_per, const double* gam, double* pack) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (n_surf < 0 || n_per <= 0 || !gam || !pack) return fail(IBS_ERR_ARG, "bad arguments");
  if (n_surf == 0) return 0;
  ON_DEVICE(ctx);
  hipLaunchKernelGGL(k_surface_argmax, dim3(n_surf), dim3(argmax_threads(n_per)), 0, ctx->stream, n_per, gam, (int*)nullptr, (double*)nullptr, pack);
  HIPCHK(hipGetLastError());
  return 0;
}

int ibs_refine_f64(ibs_ctx* ctx, int32_t n_surf, int32_t mnmax, int32_t mnmax_nyq, const double* xm, const double* xn,
                   const double* xm_nyq, const double* xn_nyq, const double* tab_mn, const double* tab_nyq,
                   const double* scal, int32_t nrows_mn, const int32_t* rows_mn, int32_t nrows_nyq,
                   const int32_t* rows_nyq, double dn_mn, double dn_nyq, int32_t n_pts, const int32_t* pt_surf,
                   const double* start, int32_t N, const double* theta, double del_alpha, int32_t maxiter,
                   double ftol, double gtol, double* x_opt, double* f_opt, int32_t* n_evals, int32_t mem) {
  if (!ctx) return fail(IBS_ERR_ARG, "null context");
  if (nrows_mn < 0 || nrows_nyq < 0 || (nrows_mn > 0 && !rows_mn) || (nrows_nyq > 0 && !rows_nyq))
    return fail(IBS_ERR_ARG, "bad row tables");
  if (n_surf <= 0 || mnmax <= 0 || mnmax_nyq <= 0 || n_pts < 0 || !xm || !xn || !xm_nyq || !xn_nyq || !tab_mn ||
      !tab_nyq || !scal || !theta || !(del_alpha > 0) || maxiter < 0)
    return fail(IBS_ERR_ARG, "bad arguments");
  if (n_pts == 0) return 0;                   // (an empty batch -- a rank without surfaces -- has no per-point arrays to point at)
  if (!pt_surf || !start || !x_opt || !f_opt) return fail(IBS_ERR_ARG, "bad arguments");
  ON_DEVICE(ctx);
  const bool host = (mem == IBS_MEM_HOST);
  if (N < 2) return fail(IBS_ERR_ARG, "N=%d", N);
  if (int r = check_grid(N, 1.0)) return r;
  if (host) {                                 // (device-resident grids are checked by k_refine_init)
    const double h = (theta[N - 1] - theta[0]) / (N - 1);
    if (!(h > 0)) return fail(IBS_ERR_ARG, "h must be > 0");
    for (int j = 1; j < N; ++j)
      if (fabs((theta[j] - theta[j - 1]) - h) > 1e-9 * fabs(h)) return fail(IBS_ERR_UNSUPPORTED, "theta grid is not uniform");
    for (int i = 0; i < n_pts; ++i)
      if (pt_surf[i] < 0 || pt_surf[i] >= n_surf) return fail(IBS_ERR_ARG, "pt_surf[%d]=%d out of range", i, pt_surf[i]);
  }
  const int M = rows_per_lane(N);
  auto eval = ibs::launch_table().refine_f64[M];
  if (!eval) return fail(IBS_ERR_UNSUPPORTED, "no kernel built for rows-per-lane M=%d (N=%d)", M, N);
  // LDS of the evaluation kernel: centre line (7 derived arrays) + eigenfunction + alpha-tangent (4 arrays) when they fit
  const size_t row_b = (size_t)ibs::lds_pitch(N) * sizeof(double);
  const size_t lds_extra = 32 * sizeof(double) + sizeof(RefineState) + sizeof(ibs::lbfgsb2::Work);
  // The alpha-tangent (4 more rows) in LDS saves the evaluation's sums a second trip to global memory but leaves room for ONE
  // block per CU at N = 969 (105 of 160 KB); without it two fit (70 KB each).  Batches with more points than CUs take the
  // second form: their first rounds would otherwise run in two waves of blocks.
  int lds_tangent = (12 * row_b + lds_extra <= (size_t)ctx->lds_per_block) ? 1 : 0;
  if (ctx->opt.refine_tangent == 0 || (ctx->opt.refine_tangent < 0 && n_pts > ctx->n_cu)) lds_tangent = 0;
  if (8 * row_b + lds_extra > (size_t)ctx->lds_per_block) return fail(IBS_ERR_UNSUPPORTED, "N=%d does not fit the LDS staging", N);
  // every round = one objective/gradient evaluation of every point still in the batch.  An iteration takes at most
  // maxls = 20 line-search evaluations, and once more after a memory restart
  const int max_rounds = 2 + 42 * (maxiter > 0 ? maxiter : 1);
  const int hist_len = (max_rounds + 2 + 1) & ~1;      // then: two 64-bit statistics words, one status word
  if (ctx->refine_hist_len < hist_len) {
    if (ctx->refine_hist) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipHostFree(ctx->refine_hist)); ctx->refine_hist = nullptr; ctx->refine_hist_len = 0; }
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->refine_hist), (size_t)(hist_len + 8) * sizeof(int), hipHostMallocDefault));
    ctx->refine_hist_len = hist_len;
  } else if (ctx->refine_pending) {
    HIPCHK(hipStreamSynchronize(ctx->stream));           // (the output kernel of the previous call posts its statistics here)
  }
  ctx->refine_pending = false;
  volatile int* hist = ctx->refine_hist;
  long long* h_stats = reinterpret_cast<long long*>(ctx->refine_hist + ctx->refine_hist_len);
  volatile int* h_status = ctx->refine_hist + ctx->refine_hist_len + 4;
  for (int r = 0; r < max_rounds + 2; ++r) hist[r] = 0;
  hist[0] = n_pts + 1;
  h_stats[0] = 0; h_stats[1] = 0; *h_status = 0;
  const long ld = N;
  const int n_lines = 3 * n_pts;
  const size_t n_mn = (size_t)n_surf * 6 * mnmax, n_nyq = (size_t)n_surf * 7 * mnmax_nyq, n_geo = (size_t)8 * n_lines * ld;
  ibs::GeoArgs ga{};
  ga.n_surf = n_surf; ga.mnmax = mnmax; ga.mnmax_nyq = mnmax_nyq; ga.n_lines = n_lines; ga.N = N; ga.ld = ld;
  ga.lpp = ctx->opt.geo_lpp;
  if (nrows_mn > 0 && nrows_nyq > 0) {
    const int fit = geo_rows_fit(ctx, mnmax, mnmax_nyq, xn, xn_nyq, nrows_mn, rows_mn, nrows_nyq, rows_nyq, dn_mn, dn_nyq, host);
    if (fit < 0) return fit;
    if (fit == 1) { ga.nrows_mn = nrows_mn; ga.nrows_nyq = nrows_nyq; ga.dn_mn = dn_mn; ga.dn_nyq = dn_nyq; }
  }
  size_t need = pad256(n_geo * 8) + pad256((size_t)n_lines * 4) + pad256((size_t)n_lines * 8) + 10 * pad256((size_t)n_pts * 16) +
                pad256((size_t)n_pts * sizeof(RefineState)) + pad256((size_t)N * 8) + 8192;
  // the lanes-per-point forms the rounds can take as the batch shrinks (geo_pick_form is monotone in the batch size)
  const int lpp_first = ibs::geo_pick_usable(ga, n_lines, N, ctx->n_cu).lpp;
  const int lpp_last = ibs::geo_pick_usable(ga, 3, N, ctx->n_cu).lpp;
  for (int lpp = lpp_first; lpp <= lpp_last; lpp *= 2) need += geo_img_bytes(ga, lpp) + 256;
  if (host) need += pad256(n_mn * 8) + pad256(n_nyq * 8) + 2 * pad256((size_t)mnmax * 8) + 2 * pad256((size_t)mnmax_nyq * 8) +
                    pad256((size_t)n_surf * 48) + pad256((size_t)nrows_mn * 8) + pad256((size_t)nrows_nyq * 8) + 4096;
  if (int r = ensure_ws(ctx, need)) return r;
  Arena ar(ctx);
  hipStream_t st = ctx->stream;
  auto up = [&](const void* src, size_t bytes, void* dst) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st); };
  if (host) {
    double* d_xm = ar.take<double>(mnmax); double* d_xn = ar.take<double>(mnmax);
    double* d_xmq = ar.take<double>(mnmax_nyq); double* d_xnq = ar.take<double>(mnmax_nyq);
    double* d_mn = ar.take<double>(n_mn); double* d_nyq = ar.take<double>(n_nyq); double* d_sc = ar.take<double>((size_t)n_surf * 6);
    HIPCHK(up(xm, (size_t)mnmax * 8, d_xm)); HIPCHK(up(xn, (size_t)mnmax * 8, d_xn));
    HIPCHK(up(xm_nyq, (size_t)mnmax_nyq * 8, d_xmq)); HIPCHK(up(xn_nyq, (size_t)mnmax_nyq * 8, d_xnq));
    HIPCHK(up(tab_mn, n_mn * 8, d_mn)); HIPCHK(up(tab_nyq, n_nyq * 8, d_nyq)); HIPCHK(up(scal, (size_t)n_surf * 48, d_sc));
    ga.xm = d_xm; ga.xn = d_xn; ga.xm_nyq = d_xmq; ga.xn_nyq = d_xnq; ga.tab_mn = d_mn; ga.tab_nyq = d_nyq; ga.scal = d_sc;
    if (ga.nrows_mn) {
      int* d_r1 = ar.take<int>((size_t)2 * nrows_mn); int* d_r2 = ar.take<int>((size_t)2 * nrows_nyq);
      HIPCHK(up(rows_mn, (size_t)nrows_mn * 8, d_r1)); HIPCHK(up(rows_nyq, (size_t)nrows_nyq * 8, d_r2));
      ga.rows_mn = d_r1; ga.rows_nyq = d_r2;
    }
  } else {
    ga.xm = xm; ga.xn = xn; ga.xm_nyq = xm_nyq; ga.xn_nyq = xn_nyq; ga.tab_mn = tab_mn; ga.tab_nyq = tab_nyq; ga.scal = scal;
    if (ga.nrows_mn) { ga.rows_mn = rows_mn; ga.rows_nyq = rows_nyq; }
  }
  const double* d_th = theta; const int* d_ps = pt_surf; const double* d_start = start;   // (device pointers are used in place)
  if (host) {
    double* t_ = ar.take<double>(N); int* p_ = ar.take<int>(n_pts); double* s_ = ar.take<double>((size_t)2 * n_pts);
    HIPCHK(up(theta, (size_t)N * 8, t_)); HIPCHK(up(pt_surf, (size_t)n_pts * 4, p_)); HIPCHK(up(start, (size_t)n_pts * 16, s_));
    d_th = t_; d_ps = p_; d_start = s_;
  }
  double* d_geo = ar.take<double>(n_geo);
  int* d_ls = ar.take<int>(n_lines); double* d_la = ar.take<double>(n_lines); double* d_t0 = ar.take<double>(n_pts); int* d_idx = ar.take<int>(n_pts);
  RefineState* d_st = ar.take<RefineState>(n_pts);
  double* d_gam = ar.take<double>(n_pts); double* d_da = ar.take<double>(n_pts); double* d_dt = ar.take<double>(n_pts);
  int* d_info = ar.take<int>(n_pts);
  RefineCtrl* d_ctrl = ar.take<RefineCtrl>(1);
  double* d_xo = x_opt; double* d_fo = f_opt; int* d_ne = n_evals;                          // (device pointers: written in place)
  if (host) { d_xo = ar.take<double>((size_t)2 * n_pts); d_fo = ar.take<double>(n_pts); d_ne = ar.take<int>(n_pts); }
  for (int lpp = lpp_first; lpp <= lpp_last; lpp *= 2)
    if (geo_img_bytes(ga, lpp)) ga.img[ibs::geo_lpp_index(lpp)] = ar.take<double>(geo_img_bytes(ga, lpp) / sizeof(double));
  ga.theta = d_th; ga.geo = d_geo; ga.dPdrho = nullptr;
  ga.line_surf = d_ls; ga.line_alpha = d_la;
  ga.n_lines_dev = &d_ctrl->n_lines;
  ga.plane = (size_t)n_lines * ld;                                        // fixed: the batch shrinks, the planes stay

  RefineParams prm{};
  prm.lo[0] = 0.0; prm.lo[1] = 0.0; prm.hi[0] = 3.141592653589793; prm.hi[1] = 1.5707963267948966;   // ball_scan.py:311
  prm.del_alpha = del_alpha; prm.ftol = ftol; prm.gtol = gtol; prm.maxiter = maxiter; prm.n_surf = n_surf;
  ibs::RefineEvalArgs<double> ea{};
  ea.N = N; ea.geo = d_geo; ea.ld = ld; ea.plane = ga.plane;
  ea.st = d_st; ea.prm = prm; ea.ctrl = d_ctrl; ea.idx = d_idx; ea.pt_surf = d_ps;
  ea.line_surf = d_ls; ea.line_alpha = d_la; ea.th0 = d_t0;
  ea.gam = d_gam; ea.dalpha = d_da; ea.dth0 = d_dt; ea.info = d_info;
  ea.hist = ctx->refine_hist; ea.hist_len = max_rounds + 2; ea.lds_tangent = lds_tangent;

  const dim3 grd((unsigned)((n_pts + 127) / 128)), blk(128);
  const dim3 gri((unsigned)((n_pts > N ? n_pts : N) + 127) / 128);
  hipLaunchKernelGGL(k_refine_init, gri, blk, 0, st, n_pts, d_ps, d_start, d_st, prm, d_idx, d_ls, d_la, d_t0, d_ctrl, N, d_th,
                     ctx->refine_hist + ctx->refine_hist_len + 4);
  HIPCHK(hipGetLastError());
  // Rounds are enqueued kLook ahead of the last one whose count the device has posted: the grid sizes and the geometry
  // form of round r are functions of the count after round r - 1 - kLook -- of the trajectory, not of host timing, so the
  // arithmetic (summation order of the geometry kernel's forms) is reproducible -- and the GPU never waits for the host.
  // Rounds enqueued after the last point has finished find an empty batch and return at once.
  constexpr int kLook = 1;
  int enq = 0, rounds = -1;
  while (enq < max_rounds) {
    const int need_r = enq > kLook ? enq - kLook : 0;
    int spins = 0;
    const auto t_start = std::chrono::steady_clock::now();       // (the limit is per awaited round, not per call)
    while (hist[need_r] == 0) {
      if (++spins > 2000) {
        std::this_thread::yield();
        if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(120)) {
          (void)hipStreamSynchronize(st);
          return fail(IBS_ERR_HIP, "refinement round %d did not report within 120 s", need_r);
        }
      }
    }
    const int nc = hist[need_r] - 1;
    if (nc <= 0) { rounds = need_r; break; }
    ga.n_lines = 3 * nc;
    ga.form = ibs::geo_pick_usable(ga, ga.n_lines, N, ctx->n_cu);
    HIPCHK(ibs::launch_geometry(ga, st, ctx->n_cu));
    ea.n_c_max = nc;
    HIPCHK(eval(ea, st));
    ++enq;
  }
  hipLaunchKernelGGL(k_refine_out, grd, blk, 0, st, n_pts, d_st, d_xo, d_fo, d_ne, h_stats);
  HIPCHK(hipGetLastError());
  ctx->refine_pending = true;
  if (host) {
    HIPCHK(up(d_xo, (size_t)n_pts * 16, x_opt)); HIPCHK(up(d_fo, (size_t)n_pts * 8, f_opt));
    if (n_evals) HIPCHK(up(d_ne, (size_t)n_pts * 4, n_evals));
  }
  // device pointers: the results are complete once the stream reaches this point (the rounds themselves are known to be
  // over -- their counts were read above -- unless the loop ended on the round limit)
  if (host || rounds < 0) { HIPCHK(hipStreamSynchronize(st)); ctx->refine_pending = false; }
  if (*h_status != 0) {
    HIPCHK(hipStreamSynchronize(st)); ctx->refine_pending = false;
    return *h_status == 1 ? fail(IBS_ERR_UNSUPPORTED, "theta grid is not uniform") : fail(IBS_ERR_ARG, "h must be > 0");
  }
  if (rounds < 0) {                       // (the loop ended on max_rounds, or the last rounds' counts were not looked at yet)
    rounds = enq;
    for (int r = 0; r <= enq; ++r) if (hist[r] == 1) { rounds = r; break; }
  }
  ctx->refine_stats[2] = rounds; ctx->refine_stats[3] = enq;
  return rounds;
}

/* statistics of the last ibs_refine_f64 call of this context: evaluations, forward sweeps of the eigen-solves, rounds needed, rounds enqueued */
int ibs_refine_stats(ibs_ctx* ctx, int64_t* out4) {
  if (!ctx || !out4) return fail(IBS_ERR_ARG, "null pointer");
  if (ctx->refine_pending) { ON_DEVICE(ctx); HIPCHK(hipStreamSynchronize(ctx->stream)); ctx->refine_pending = false; }
  if (ctx->refine_hist) {
    const long long* h_stats = reinterpret_cast<const long long*>(ctx->refine_hist + ctx->refine_hist_len);
    ctx->refine_stats[0] = h_stats[0]; ctx->refine_stats[1] = h_stats[1];
  }
  for (int i = 0; i < 4; ++i) out4[i] = ctx->refine_stats[i];
  return 0;
}

}  // extern "C"
