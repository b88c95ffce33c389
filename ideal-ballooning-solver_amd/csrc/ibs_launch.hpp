// Launch-argument structs and the per-M launcher table shared by ibs_kernels.hip (one object per
// rows-per-lane value M) and ibs_api.hip (the C-ABI layer).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

namespace ibs {

constexpr int kMaxM = 32;   // rows per lane: N - 2 <= 64 * kMaxM  (N <= 2050)
constexpr int kMaxLongN = 65537;   // longer grids, up to this many points, run on the generic division-form path (ibs_long.hip)

template <typename T>
struct GcfArgs {
  long n_sys; int N; T h; const T* g; const T* c; const T* f; long ld;
  T* lam; T* gam; T* X; T* dX; int* info; int wpb;
  const T* gh;      // optional half-grid g [n_sys][ld] (N-1 used); null = mean of neighbouring g
  int flags;        // FP64 solvers: bit 0 = re-close suspect systems in division form, bit 1 = only mark them
  // big-batch forms (rows from global memory, sub-wave): a suspect system is only LISTED by the solver kernel -- its re-close and the
  // repeat of its eigenvector stage run in a second, tiny launch (k_fix_gcf) so that none of that code sits in the hot kernel (it cost
  // 2 % at N_zeta = 512 and 9 % on the two-waves-per-SIMD forms: tools/ab_reclose.py).  fix_count[1] (zeroed by the caller),
  // fix_sys / fix_center [>= n_sys]: system index and the Rayleigh polish to re-close about.  null = re-close inside the kernel.
  int* fix_count; long* fix_sys; double* fix_center;
};
template <typename T>
struct ScanArgs {
  int n_lines, n_theta0, N; T h;
  const T *bmag, *gradpar, *cvdrift, *cvdrift0, *gds2, *gds21, *gds22; long ld;
  const T *dPdrho, *theta0;
  T *gam, *lam, *X, *dX, *dth0; int* info; int wpb;
  const T* lam_guess; T guess_width;     // optional warm start ([n_lines][n_theta0], absolute width); null = cold
  int chain; T chain_w1, chain_w2;       // k_gamma_scan_chain: theta0 values per wave and the widths of its warm starts
  // fused per-surface argmax (k_gamma_scan only): lines_per_surf consecutive lines form a surface; the block that
  // completes a surface reduces its [lines_per_surf * n_theta0] growth rates into pack[surf] = (max, first index).
  // surf_counter[n_surf]: zero between launches (the last arriver resets its word).  null pack = no fusion.
  int lines_per_surf; int* surf_counter; T* pack;
  int pack_mode;                         // 1 = write-through stores + sc1 loads (one block per CU), 2 = release / acquire fences
  int t0_stride;                         // k_gamma_scan only: 0 = theta0[n_theta0] shared by all lines, 1 = theta0[n_lines] (n_theta0 = 1)
};
template <typename T>
struct SturmArgs {
  long n_sys; int N; T h; const T* g; const T* c; const T* f; long ld; const T* shift; int* count; int wpb;
};

template <typename T>
struct GradArgs {
  int n_pts, N; T h; const T* geo; long ld; const T* theta0; T del_alpha;
  T *val, *jac, *gam, *dalpha, *dth0; int* info; int wpb;
  long arr_stride, line_stride;   // 0 = the [n_pts][3][8][ld] layout (ld, 8 ld); see k_obj_w_grad
};

// grids beyond 64 * kMaxM + 2 points (ibs_long.hip): one wave per system, division form throughout
struct LongGcfArgs {
  long n_sys; int N; double h;
  const void *g, *c, *f, *gh; int f32;      // f32: the arrays (inputs and outputs) are float; the arithmetic is FP64 either way
  long ld; void *lam, *gam, *X, *dX; int* info;
  double* work; int n_waves;                // n_waves * 3 * N doubles of workspace; the grid is min(n_sys, n_waves) waves
};
hipError_t launch_gcf_long(const LongGcfArgs& a, hipStream_t st);
hipError_t launch_sturm_long(const SturmArgs<double>& a, hipStream_t st);
hipError_t launch_sturm_div(const SturmArgs<double>& a, hipStream_t st);     // lanes as systems, division form, any N
template <typename T> struct ScanArgs;
hipError_t launch_assemble_long(const ScanArgs<double>& a, double* g, double* c, double* f, double* gt, double* ct, double* ft, hipStream_t st);

// field-line geometry kernel (ibs_geometry.hip)
// dPdrho[line] = -0.5 mean((cvdrift - gbdrift) bmag^2) of lines laid out geo[k * plane + line * ld + j], k = 0 .. 7 (ball_scan.py:262)
hipError_t launch_line_dPdrho(int n_lines, int N, long ld, size_t plane, const double* geo, double* dPdrho, hipStream_t st);
struct GeoForm { int ppl, lpp; };   // grid points per lane, lanes per grid point (one of them is 1)
constexpr int kGeoMaxPairs = 64;    // pair indices about a row's centre the one-lane-per-point table image holds for the root solve
struct GeoArgs {
  int n_surf, mnmax, mnmax_nyq, n_lines, N;
  const double *xm, *xn, *xm_nyq, *xn_nyq;
  const double* tab_mn;    // [n_surf][6][mnmax]      rmnc zmns lmns d_rmnc_d_s d_zmns_d_s d_lmns_d_s
  const double* tab_nyq;   // [n_surf][7][mnmax_nyq]  gmnc bmnc d_bmnc_d_s bsupvmnc bsubsmns bsubumnc bsubvmnc
  const double* scal;      // [n_surf][6]             s iota d_iota_d_s d_pressure_d_s phiedge Aminor_p
  const int* line_surf; const double* line_alpha; const double* theta;
  long ld;
  double* geo;             // [8][n_lines][ld]  bmag gradpar cvdrift cvdrift0 gds2 gds21 gds22 gbdrift
  double* dPdrho;          // [n_lines]
  // optional row structure of the mode lists (VMEC order: modes grouped by m, consecutive n inside a group):
  // rows_*[r] = {first mode index, number of modes}; the angle then advances by -dphi_n per mode and only
  // one sincos per row is needed.  nrows_* = 0 selects the generic one-sincos-per-mode kernel.
  int nrows_mn, nrows_nyq;
  const int* rows_mn;      // [nrows_mn][2]
  const int* rows_nyq;     // [nrows_nyq][2]
  double dn_mn, dn_nyq;    // common n-spacing inside the rows of each set (rows with another spacing are split by the host)
  int lpp;                 // lanes per grid point: 0 = chosen from the batch size, else 1 | 2 | 4 | 8; -2 = two grid points per lane
  int j_begin, j_end;      // (set by launch_geometry) grid points [j_begin, j_end) of every line are this launch's
  // round 3: the row kernels work from per-surface table images prepared once per call (k_geo_prepare), one image set
  // per lanes-per-point value (index geo_lpp_index: 1, 2, 4, 8); img_ready = bit per set already built from these tables
  GeoForm form;            // {0, 0} = picked by launch_geometry from the batch size
  double* img[4];
  unsigned img_ready;
  const int* n_lines_dev;  // optional: the number of lines actually requested, read on the device (<= n_lines, which then
                           // only sizes the grid): the refinement rounds of ibs_refine_f64 shrink without a host round trip
  size_t plane;            // distance between the 8 output planes in elements; 0 = n_lines * ld
  int* surf_used;          // optional [n_surf] scratch: launch_geometry marks the surfaces the lines refer to and k_geo_prepare builds
                           // only their images (a caller that works through a large table set piece by piece: AdjointStep)
};
GeoForm geo_pick_form(long n_lines, int N, int n_cu, int lpp_opt);
GeoForm geo_pick_usable(const GeoArgs& a, long n_lines, int N, int n_cu);      // ... among the forms whose table image fits the LDS
int geo_lpp_index(int lpp);
size_t geo_image_doubles(const GeoArgs& a, int lpp);      // per surface
bool geo_rows_usable(const GeoArgs& a, int lpp);
hipError_t launch_geometry(GeoArgs& a, hipStream_t st, int n_cu);

template <typename T> struct RefineEvalArgs;
struct LaunchTable {
  hipError_t (*gcf_f64[kMaxM + 1])(const GcfArgs<double>&, hipStream_t);
  hipError_t (*gcf_rows_f64[kMaxM + 1])(const GcfArgs<double>&, hipStream_t);   // row-streamed form (k_solve_gcf_rows)
  hipError_t (*gcf_direct_f64[kMaxM + 1])(const GcfArgs<double>&, hipStream_t);  // rows straight from global memory (k_solve_gcf_direct)
  hipError_t (*gcf_direct_f32w[kMaxM + 1])(const GcfArgs<float>&, hipStream_t);  // the same on FP32 arrays (FP64 solver)
  hipError_t (*gcf_direct_f32lam[kMaxM + 1])(const GcfArgs<float>&, hipStream_t); // FP32 eigenvalues only: all-FP32 iteration + FP64 certificate, rows from global memory
  hipError_t (*gcf_fix_f64[kMaxM + 1])(const GcfArgs<double>&, hipStream_t);     // the listed suspects of a big-batch launch: re-close + eigenvector stage (k_fix_gcf), M = ceil((N - 2) / 64)
  hipError_t (*gcf_fix_f32w[kMaxM + 1])(const GcfArgs<float>&, hipStream_t);
  hipError_t (*gcf_f32[kMaxM + 1])(const GcfArgs<float>&, hipStream_t);
  hipError_t (*gcf_f32_wide[kMaxM + 1])(const GcfArgs<float>&, hipStream_t);   // FP32 in HBM, FP64 in the solver (gam / X wanted)
  hipError_t (*gcf_f32w_rows[kMaxM + 1])(const GcfArgs<float>&, hipStream_t);  // the same, row-streamed (long grids)
  hipError_t (*gcf_f32w_g[2][kMaxM + 1])(const GcfArgs<float>&, hipStream_t);  // the same, 32 / 16 lanes per system
  hipError_t (*scan_f64[kMaxM + 1])(const ScanArgs<double>&, hipStream_t);
  hipError_t (*scan_chain_f64[kMaxM + 1])(const ScanArgs<double>&, hipStream_t);
  hipError_t (*sturm_f64[kMaxM + 1])(const SturmArgs<double>&, hipStream_t);
  hipError_t (*grad_f64[kMaxM + 1])(const GradArgs<double>&, hipStream_t);
  hipError_t (*refine_f64[kMaxM + 1])(const RefineEvalArgs<double>&, hipStream_t);   // one round of ibs_refine_f64 (ibs_refine.hpp)
  // sub-wave variants (ibs_group.hpp): index 0 -> 32 lanes per system, 1 -> 16 lanes per system
  hipError_t (*gcf_f64_g[2][kMaxM + 1])(const GcfArgs<double>&, hipStream_t);
  hipError_t (*scan_f64_g[2][kMaxM + 1])(const ScanArgs<double>&, hipStream_t);
  hipError_t (*scan_chain_f64_g[2][kMaxM + 1])(const ScanArgs<double>&, hipStream_t);   // chained / warm-started
};
LaunchTable& launch_table();

// Diagnostic record of the solver / geometry kernel most recently launched by this thread (C ABI: ibs_last_launch): the
// name as rocprofv3 prints it ("ibs::k_gamma_scan<double, 8>") and the launch dimensions -- the library picks lanes per
// system, chaining and geometry form from the batch size, and a measurement must be able to say which kernel it timed.
struct LaunchNote { char name[96]; long blocks; int threads; };
LaunchNote& last_launch();      // thread-local (ibs_api.hip)
void note_launch(long blocks, int threads, const char* fmt, ...);
template <typename T> constexpr const char* type_name() { return sizeof(T) == 8 ? "double" : "float"; }

// threads per block the scan kernel is compiled for (register budget: 5M doubles per lane)
// LDS rows of the wave kernels are padded by one element per 8 (element j lives at j + (j >> 3)): a lane reads a
// contiguous chunk of ~M rows, i.e. lanes are M elements apart, and for even M (8 at N = 513) an unpadded row
// puts 16 lanes on the same banks.  Row pitch in elements:
constexpr int lds_pitch(int N) { return N + (N >> 3) + 1; }

constexpr int scan_max_threads(int M) { return M <= 16 ? 512 : 256; }
constexpr int scan_max_threads_g(int M) { return M <= 8 ? 512 : 256; }   // sub-wave kernels (ibs_group.hpp)

}  // namespace ibs
