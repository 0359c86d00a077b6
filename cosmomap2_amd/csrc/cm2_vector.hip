// cm2_vector.hip -- map-domain vector kernels: the BLAS-1 pieces of the PCG
// recurrence, the tall-skinny deflation operators Z / Z^T, the coarse matrix
// E = Z^T (A Z) on fp64 MFMA, and the fused tail of the two-level preconditioner.
//
// Reference code replaced:
//   norm2/scalprod                utilities/linear_algebra_funcs.py:31-44
//   cg recurrence                 scipy.sparse.linalg.cg (tests/test_2level_preconditioner.py:52)
//   DeflationLO.mult / rmult      interfaces/linearoperators.py:1041-1056
//   CoarseLO.__init__ dgemm       interfaces/linearoperators.py:1019
//   M2 = Mbd*R + Zd*E*Zd.T        src/test_M2_precond_onto_real_data.py:109-112
//
// Everything here is HBM-bound (one pass over the vectors / over Z) except
// cm2_gemm_tn, which is the one GEMM-shaped contraction on the path and runs on
// v_mfma_f64_16x16x4_f64.  Reductions are two-stage with a fixed tree, so results
// are bitwise reproducible from run to run.
#include "cm2_blocks.h"

using namespace cm2;

namespace {
constexpr int kRedBlocks = 1024;            // stage-1 workgroups of every reduction
constexpr int kGemmBlocks = 128;            // workgroups of the Z^T Z contraction
typedef double double4_t __attribute__((ext_vector_type(4)));
}  // namespace

extern "C" int64_t cm2_reduce_work_doubles(void) { return (int64_t)kRedBlocks * 256; }

static inline int red_blocks(int64_t n)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kRedBlocks) g = kRedBlocks;
    return (int)g;
}

// ------------------------------------------------------------------ dot --------
__global__ __launch_bounds__(256) void k_dot_partial(int64_t n, const double *__restrict__ x,
                                                      const double *__restrict__ y,
                                                      double *__restrict__ partial)
{
    __shared__ double lds[4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        acc += x[i] * y[i];
    const double r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// sums `count` partials (stride 1) into out[0]; one workgroup, fixed order
__global__ __launch_bounds__(256) void k_reduce_final(int count, const double *__restrict__ partial,
                                                       double *__restrict__ out)
{
    __shared__ double lds[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial[i];
    const double r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) out[0] = r;
}

extern "C" int cm2_dot(int64_t n, const double *d_x, const double *d_y, double *d_out,
                       double *d_work, void *stream_)
{
    CM2_CHECK(d_out && d_work && (n == 0 || (d_x && d_y)), "cm2_dot: NULL argument");
    hipStream_t stream = as_stream(stream_);
    const int g = red_blocks(n);
    k_dot_partial<<<g, kBlock, 0, stream>>>(n, d_x, d_y, d_work);
    CM2_LAUNCH_OK();
    k_reduce_final<<<1, kBlock, 0, stream>>>(g, d_work, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------- axpy / scal / xmy -----
__global__ __launch_bounds__(256) void k_axpy(int64_t n, double a, const double *__restrict__ x,
                                               double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = y[i] + a * x[i];
}

__global__ __launch_bounds__(256) void k_scal(int64_t n, double a, double *__restrict__ x)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        x[i] = a * x[i];
}

__global__ __launch_bounds__(256) void k_xmy(int64_t n, const double *__restrict__ x,
                                              const double *__restrict__ y,
                                              double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = x[i] * y[i];
}

extern "C" int cm2_axpy(int64_t n, double alpha, const double *d_x, double *d_y, void *stream_)
{
    CM2_CHECK(n == 0 || (d_x && d_y), "cm2_axpy: NULL argument");
    if (n == 0) return 0;
    k_axpy<<<grid_for(n), kBlock, 0, as_stream(stream_)>>>(n, alpha, d_x, d_y);
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_scal(int64_t n, double alpha, double *d_x, void *stream_)
{
    CM2_CHECK(n == 0 || d_x, "cm2_scal: NULL argument");
    if (n == 0) return 0;
    k_scal<<<grid_for(n), kBlock, 0, as_stream(stream_)>>>(n, alpha, d_x);
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_xmy(int64_t n, const double *d_x, const double *d_y, double *d_out,
                       void *stream_)
{
    CM2_CHECK(n == 0 || (d_x && d_y && d_out), "cm2_xmy: NULL argument");
    if (n == 0) return 0;
    k_xmy<<<grid_for(n), kBlock, 0, as_stream(stream_)>>>(n, d_x, d_y, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------ PCG fused updates -----
// cg: beta = rho_cur / rho_prev ; p *= beta ; p += z
__global__ __launch_bounds__(256) void k_pcg_update_p(int64_t n, const double *__restrict__ rho,
                                                       const double *__restrict__ rho_prev,
                                                       const double *__restrict__ z,
                                                       double *__restrict__ p)
{
    const double beta = rho[0] / rho_prev[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        p[i] = p[i] * beta + z[i];
}

// cg: alpha = rho_cur / dot(p,q) ; x += alpha*p ; r -= alpha*q ; and ||r||^2 on the fly
__global__ __launch_bounds__(256) void k_pcg_update_xr(
    int64_t n, const double *__restrict__ rho, const double *__restrict__ pq,
    const double *__restrict__ p, const double *__restrict__ q, double *__restrict__ x,
    double *__restrict__ r, double *__restrict__ partial)
{
    __shared__ double lds[4];
    const double alpha = rho[0] / pq[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        x[i] = x[i] + alpha * p[i];
        const double rn = r[i] - alpha * q[i];
        r[i] = rn;
        acc += rn * rn;
    }
    const double s = block_sum_256(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

extern "C" int cm2_pcg_update_p(int64_t n, const double *d_rho, const double *d_rho_prev,
                                const double *d_z, double *d_p, void *stream_)
{
    CM2_CHECK(d_rho && d_rho_prev && d_z && d_p, "cm2_pcg_update_p: NULL argument");
    k_pcg_update_p<<<grid_for(n), kBlock, 0, as_stream(stream_)>>>(n, d_rho, d_rho_prev, d_z, d_p);
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_pcg_update_xr(int64_t n, const double *d_rho, const double *d_pq,
                                 const double *d_p, const double *d_q, double *d_x, double *d_r,
                                 double *d_rr, double *d_work, void *stream_)
{
    CM2_CHECK(d_rho && d_pq && d_p && d_q && d_x && d_r && d_rr && d_work,
              "cm2_pcg_update_xr: NULL argument");
    hipStream_t stream = as_stream(stream_);
    const int g = red_blocks(n);
    k_pcg_update_xr<<<g, kBlock, 0, stream>>>(n, d_rho, d_pq, d_p, d_q, d_x, d_r, d_work);
    CM2_LAUNCH_OK();
    k_reduce_final<<<1, kBlock, 0, stream>>>(g, d_work, d_rr);
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------------- PCG driver ------
// scipy.sparse.linalg.cg's recurrence (the driver the reference calls, e.g.
// tests/test_2level_preconditioner.py:52) for hosts that are not Python: the operator and the
// preconditioner are callbacks working on device vectors on `stream`; alpha and beta never leave
// HBM, one 8-byte copy per iteration brings ||r||^2 to the host for the stopping test.
//
// Sharded solves use the same driver: `reduce` (the HOST's collective: RCCL, MPI ...; the
// library links none) combines device scalars over the ranks in place, queued on `stream`.
//   layout CM2_LAYOUT_REPLICATED: every rank holds whole vectors and the operator callback returns the
//     all-reduced product; the dots are computed redundantly and only ||r||^2 is MAX-reduced, so that
//     every rank takes the same stop decision whatever the collective's rounding did to the copies;
//   layout CM2_LAYOUT_ROWS: every vector is the rank's rows; b.b, rho, p.q and ||r||^2 are SUM-reduced
//     (the callback gathers p and reduce-scatters the product).
static int pcg_run(int64_t n, cm2_apply_fn A, void *A_ctx, cm2_apply_fn M, void *M_ctx,
                   const double *d_b, double *d_x, int x_is_zero, double rtol, double atol,
                   int64_t maxiter, cm2_iter_fn callback, void *cb_ctx, int layout,
                   cm2_reduce_fn reduce, void *reduce_ctx, int64_t *h_iters, int *h_info, void *stream_)
{
    hipStream_t stream = as_stream(stream_);
    // a dot product of whole-vector meaning / the squared residual norm, after the local kernel
    auto red_dot = [&](double *d) -> int {
        if (reduce && layout == CM2_LAYOUT_ROWS && reduce(reduce_ctx, d, 1, CM2_REDUCE_SUM, stream_)) {
            set_error("cm2_pcg_sharded: the reduction callback failed");
            return 1;
        }
        return 0;
    };
    auto red_rr = [&](double *d) -> int {
        if (reduce && reduce(reduce_ctx, d, 1, layout == CM2_LAYOUT_ROWS ? CM2_REDUCE_SUM : CM2_REDUCE_MAX,
                             stream_)) {
            set_error("cm2_pcg_sharded: the reduction callback failed");
            return 1;
        }
        return 0;
    };
    *h_iters = 0;
    *h_info = 0;
    DevTemp<double> r, z, p, q, sc, work;
    CM2_HIP(r.alloc(n));
    CM2_HIP(p.alloc(n));
    CM2_HIP(q.alloc(n));
    if (M) CM2_HIP(z.alloc(n));
    CM2_HIP(sc.alloc(8));                                  // rho[2], pq, rr, tmp
    CM2_HIP(work.alloc((size_t)cm2_reduce_work_doubles()));
    double *rho[2] = {sc.p, sc.p + 1}, *pq = sc.p + 2, *rr = sc.p + 3, *tmp = sc.p + 4;
    double h = 0.0;
    auto fetch = [&](const double *d) -> int {
        CM2_HIP(cm2::download(&h, d, sizeof(double), stream));
        CM2_HIP(hipStreamSynchronize(stream));
        return 0;
    };
    if (int rc = cm2_dot(n, d_b, d_b, tmp, work, stream)) return rc;
    if (int rc = red_rr(tmp)) return rc;                   // atol is the same number on every rank
    if (int rc = fetch(tmp)) return rc;
    const double bnrm2 = sqrt(h);
    if (bnrm2 == 0.0) {                                    // x = b = 0
        CM2_HIP(hipMemcpyAsync(d_x, d_b, sizeof(double) * n, hipMemcpyDeviceToDevice, stream));
        CM2_HIP(hipStreamSynchronize(stream));
        return 0;
    }
    if (rtol * bnrm2 > atol) atol = rtol * bnrm2;
    if (maxiter < 0) {                                     // scipy's default 10 n, n the WHOLE vector's length
        maxiter = 10 * n;
        if (reduce && layout == CM2_LAYOUT_ROWS) {
            const double nl = (double)n;
            CM2_HIP(cm2::upload(tmp, &nl, sizeof(double), stream));
            if (int rc = red_dot(tmp)) return rc;
            if (int rc = fetch(tmp)) return rc;
            maxiter = 10 * (int64_t)h;
        }
    }
    if (x_is_zero) {
        CM2_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * n, stream));
        CM2_HIP(hipMemcpyAsync(r.p, d_b, sizeof(double) * n, hipMemcpyDeviceToDevice, stream));
    } else {                                               // r = b - A x0
        if (A(A_ctx, d_x, q.p, stream_)) { set_error("cm2_pcg: the operator callback failed"); return 1; }
        CM2_HIP(hipMemcpyAsync(r.p, d_b, sizeof(double) * n, hipMemcpyDeviceToDevice, stream));
        if (int rc = cm2_axpy(n, -1.0, q.p, r.p, stream_)) return rc;
    }
    // ||r||^2 is read through pinned memory behind an event, and the host waits for it only
    // after it has queued everything of the next iteration that touches neither x nor r
    // (z = M r, rho, p, q = A p, p.q): the GPU never idles while the host issues launches.
    // On convergence the speculative work is dropped (it wrote z, p, q and two scalars).
    struct Deferred {
        double *host = nullptr;
        hipEvent_t ev = nullptr;
        ~Deferred() { if (host) (void)hipHostFree(host); if (ev) (void)hipEventDestroy(ev); }
    } df;
    CM2_HIP(hipHostMalloc((void **)&df.host, sizeof(double), hipHostMallocDefault));
    CM2_HIP(hipEventCreateWithFlags(&df.ev, hipEventDisableTiming));
    auto post = [&](const double *d) -> int {
        // (df.host is the caller's PAGE-LOCKED buffer and this copy is deliberately not waited for)
        CM2_HIP(hipMemcpyAsync(df.host, d, sizeof(double), hipMemcpyDeviceToHost, stream));
        CM2_HIP(hipEventRecord(df.ev, stream));
        return 0;
    };
    if (int rc = cm2_dot(n, r.p, r.p, rr, work, stream)) return rc;
    if (int rc = red_rr(rr)) return rc;
    if (int rc = post(rr)) return rc;
    int cur = 0;
    // z = M r, rho, p of iteration `it` (cheap: always queued ahead of the stop test) and
    // q = A p, p.q (the matvec: queued ahead unless convergence is predicted); nothing here writes
    // x or r
    auto ahead_cheap = [&](int64_t it) -> int {
        const double *zz = r.p;
        if (M) {
            if (M(M_ctx, r.p, z.p, stream_)) { set_error("cm2_pcg: the preconditioner callback failed"); return 1; }
            zz = z.p;
        }
        if (int rc = cm2_dot(n, r.p, zz, rho[cur], work, stream)) return rc;
        if (int rc = red_dot(rho[cur])) return rc;
        if (it > 0) return cm2_pcg_update_p(n, rho[cur], rho[1 - cur], zz, p.p, stream_);
        CM2_HIP(hipMemcpyAsync(p.p, zz, sizeof(double) * n, hipMemcpyDeviceToDevice, stream));
        return 0;
    };
    auto ahead_matvec = [&]() -> int {
        if (A(A_ctx, p.p, q.p, stream_)) { set_error("cm2_pcg: the operator callback failed"); return 1; }
        if (int rc = cm2_dot(n, p.p, q.p, pq, work, stream)) return rc;
        return red_dot(pq);
    };
    // Running ahead with the matvec costs one iteration of GPU time when the stop test then says
    // "converged", so it is skipped when the last two residuals the host has seen predict
    // convergence at this test (geometric extrapolation, a factor 10 in the norm to spare); results
    // never depend on it.
    double known1 = -1.0, known2 = -1.0;                   // the last two ||r||^2 seen (newest first)
    for (int64_t it = 0; it < maxiter; ++it) {
        double predicted = INFINITY;
        if (known1 >= 0.0) predicted = (known2 > 0.0) ? known1 * (known1 / known2) : known1;
        const bool run_ahead = !(predicted <= 100.0 * atol * atol);
        if (int rc = ahead_cheap(it)) return rc;
        if (run_ahead)
            if (int rc = ahead_matvec()) return rc;
        // ---- the state before this iteration: report it, then the stop test
        CM2_HIP(hipEventSynchronize(df.ev));
        h = *df.host;
        known2 = known1;
        known1 = h;
        if (it > 0) {
            *h_iters = it;
            if (callback) callback(cb_ctx, it, d_x, sqrt(h));
        }
        if (sqrt(h) < atol) return 0;
        if (!run_ahead)
            if (int rc = ahead_matvec()) return rc;
        if (int rc = cm2_pcg_update_xr(n, rho[cur], pq, p.p, q.p, d_x, r.p, rr, work, stream_)) return rc;
        cur = 1 - cur;
        if (int rc = red_rr(rr)) return rc;
        if (int rc = post(rr)) return rc;
    }
    if (maxiter > 0) {
        CM2_HIP(hipEventSynchronize(df.ev));
        h = *df.host;
        *h_iters = maxiter;
        if (callback) callback(cb_ctx, maxiter, d_x, sqrt(h));
    }
    *h_info = (int)maxiter;                                // not converged within maxiter
    return 0;
}

extern "C" int cm2_pcg(int64_t n, cm2_apply_fn A, void *A_ctx, cm2_apply_fn M, void *M_ctx,
                       const double *d_b, double *d_x, int x_is_zero, double rtol, double atol,
                       int64_t maxiter, cm2_iter_fn callback, void *cb_ctx, int64_t *h_iters,
                       int *h_info, void *stream_)
{
    CM2_CHECK(n >= 1 && A && d_b && d_x && h_iters && h_info, "cm2_pcg: NULL argument or n < 1");
    return pcg_run(n, A, A_ctx, M, M_ctx, d_b, d_x, x_is_zero, rtol, atol, maxiter, callback, cb_ctx,
                   CM2_LAYOUT_REPLICATED, nullptr, nullptr, h_iters, h_info, stream_);
}

extern "C" int cm2_pcg_sharded(int64_t n_local, cm2_apply_fn A, void *A_ctx, cm2_apply_fn M, void *M_ctx,
                               const double *d_b, double *d_x, int x_is_zero, double rtol, double atol,
                               int64_t maxiter, cm2_iter_fn callback, void *cb_ctx, int layout,
                               cm2_reduce_fn reduce, void *reduce_ctx, int64_t *h_iters, int *h_info,
                               void *stream_)
{
    CM2_CHECK(n_local >= 1 && A && d_b && d_x && h_iters && h_info && reduce,
              "cm2_pcg_sharded: NULL argument or n_local < 1");
    CM2_CHECK(layout == CM2_LAYOUT_REPLICATED || layout == CM2_LAYOUT_ROWS,
              "cm2_pcg_sharded: layout must be CM2_LAYOUT_REPLICATED or CM2_LAYOUT_ROWS");
    return pcg_run(n_local, A, A_ctx, M, M_ctx, d_b, d_x, x_is_zero, rtol, atol, maxiter, callback, cb_ctx,
                   layout, reduce, reduce_ctx, h_iters, h_info, stream_);
}

// Modified Gram-Schmidt Arnoldi with the reference's conventions (interfaces/deflationlib.py:
// 17-113): r0 = b - A x0, early exit when ||r0|| < tol ||b|| or < tol (:80-82, *h_steps = 0),
// per step w = A v_{j-1}; for every stored v: alpha = <v, w>, w -= alpha v (:94-97); h_{j+1,j} =
// ||w||; stop when |w[j] h_{j+1,j}| <= tol -- the j-th COMPONENT of the normalised new vector, as
// the reference tests it (:101) --, error when inner_m steps do not trigger it (:111-112).
// d_V holds the basis vectors one after the other (vector i at d_V + i n, inner_m of them at
// most); h_H is the (inner_m + 1) x inner_m Hessenberg matrix, row-major on the host, zero-filled
// here; on return *h_steps = j and the first j vectors / the leading (j + 1) x j block are set.
extern "C" int cm2_arnoldi(int64_t n, cm2_apply_fn A, void *A_ctx, const double *d_b,
                           const double *d_x0, double tol, int inner_m, double *d_V, double *h_H,
                           int *h_steps, void *stream_)
{
    CM2_CHECK(n >= 1 && A && d_b && d_V && h_H && h_steps && inner_m >= 1,
              "cm2_arnoldi: NULL argument, n < 1 or inner_m < 1");
    hipStream_t stream = as_stream(stream_);
    *h_steps = 0;
    for (int64_t i = 0; i < (int64_t)(inner_m + 1) * inner_m; ++i) h_H[i] = 0.0;
    DevTemp<double> w, sc, work;
    CM2_HIP(w.alloc(n));
    CM2_HIP(sc.alloc(2));
    CM2_HIP(work.alloc((size_t)cm2_reduce_work_doubles()));
    double h = 0.0;
    auto dot = [&](const double *a, const double *b2) -> int {
        if (int rc = cm2_dot(n, a, b2, sc.p, work, stream)) return rc;
        CM2_HIP(cm2::download(&h, sc.p, sizeof(double), stream));
        CM2_HIP(hipStreamSynchronize(stream));
        return 0;
    };
    if (int rc = dot(d_b, d_b)) return rc;
    CM2_CHECK(h == h && h - h == 0.0, "RHS must contain only finite numbers");   // deflationlib.py:60-61
    double b_norm = sqrt(h);
    if (b_norm == 0.0) b_norm = 1.0;
    // r0 = b - A x0 into the first basis slot
    double *v0 = d_V;
    CM2_HIP(hipMemcpyAsync(v0, d_b, sizeof(double) * n, hipMemcpyDeviceToDevice, stream));
    if (d_x0) {
        if (A(A_ctx, d_x0, w.p, stream_)) { set_error("cm2_arnoldi: the operator callback failed"); return 1; }
        if (int rc = cm2_axpy(n, -1.0, w.p, v0, stream_)) return rc;
    }
    if (int rc = dot(v0, v0)) return rc;
    const double r_norm = sqrt(h);
    if (r_norm < tol * b_norm || r_norm < tol) return 0;     // "Arnoldi exited at the first iteration"
    if (int rc = cm2_scal(n, 1.0 / r_norm, v0, stream_)) return rc;
    for (int j = 1; j <= inner_m; ++j) {
        double *vnew = (j < inner_m) ? d_V + (int64_t)j * n : w.p;   // the last step's vector is not kept
        if (A(A_ctx, d_V + (int64_t)(j - 1) * n, vnew, stream_)) {
            set_error("cm2_arnoldi: the operator callback failed");
            return 1;
        }
        for (int i = 0; i < j; ++i) {
            const double *vi = d_V + (int64_t)i * n;
            if (int rc = dot(vi, vnew)) return rc;
            h_H[(int64_t)i * inner_m + (j - 1)] = h;
            if (int rc = cm2_axpy(n, -h, vi, vnew, stream_)) return rc;
        }
        if (int rc = dot(vnew, vnew)) return rc;
        const double hn = sqrt(h);
        h_H[(int64_t)j * inner_m + (j - 1)] = hn;
        if (int rc = cm2_scal(n, 1.0 / hn, vnew, stream_)) return rc;
        double vj = 0.0;
        if (j < n) {
            CM2_HIP(cm2::download(&vj, vnew + j, sizeof(double), stream));
            CM2_HIP(hipStreamSynchronize(stream));
        }
        *h_steps = j;
        if (fabs(vj * hn) <= tol) return 0;
    }
    set_error("Convergence not achieved within the Arnoldi algorithm");     // deflationlib.py:112
    return 4;
}

// ------------------------------------------------------------- Z^T x ----------
// Z row-major [n][r].  Thread (row-lane, column): consecutive threads read
// consecutive doubles of a row => full-line accesses; per-column partials are
// combined in LDS in a fixed order, then across workgroups by k_Zt_final.
__global__ __launch_bounds__(256) void k_Zt_partial(int64_t n, int r, int rp, int64_t rows_per_blk,
                                                     const double *__restrict__ Z,
                                                     const double *__restrict__ x,
                                                     double *__restrict__ partial)
{
    __shared__ double lds[256];
    const int col = threadIdx.x % rp;
    const int rl = threadIdx.x / rp;
    const int rstep = 256 / rp;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > n) r1 = n;
    double acc = 0.0;
    if (col < r)
        for (int64_t i = r0 + rl; i < r1; i += rstep) acc += Z[i * r + col] * x[i];
    lds[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < rp && col < r) {
        double s = 0.0;
        for (int q = 0; q < rstep; ++q) s += lds[q * rp + col];
        partial[(int64_t)blockIdx.x * r + col] = s;
    }
}

// 1024 threads: with 1024 workgroup partials and r = 32 a thread adds 32 values (256 threads:
// 128 dependent-latency loads, 38 us -- a third of the pass over Z it follows)
__global__ __launch_bounds__(1024) void k_Zt_final(int nblk, int r, int rp,
                                                    const double *__restrict__ partial,
                                                    double *__restrict__ out)
{
    __shared__ double lds[1024];
    const int col = threadIdx.x % rp;
    const int q = threadIdx.x / rp;
    const int qn = 1024 / rp;
    double acc = 0.0;
    if (col < r)
        for (int b = q; b < nblk; b += qn) acc += partial[(int64_t)b * r + col];
    lds[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < rp && col < r) {
        double s = 0.0;
        for (int k = 0; k < qn; ++k) s += lds[k * rp + col];
        out[col] = s;
    }
}

static inline int pow2_at_least(int r)
{
    int p = 1;
    while (p < r) p <<= 1;
    return p;
}

// Wide form for r = 2 LPR with LPR a power of two <= 32 (r = 16, 32, 64 ...): LPR lanes read one
// row with 16-byte loads (a wave covers 64 / LPR rows per instruction, 8 instructions in
// flight), every lane keeps the partial sums of its two columns, and the lanes / waves of a
// workgroup are combined in LDS in a fixed order.  Streams Z at HBM speed (the narrow kernel
// above issues 8-byte loads, one row per 32 lanes: 3.1 TB/s at n = 2.4e6, r = 32).
template <int LPR>
__global__ __launch_bounds__(256) void k_Zt_partial_wide(int64_t n, int64_t rows_per_blk,
                                                          const double *__restrict__ Z,
                                                          const double *__restrict__ x,
                                                          double *__restrict__ partial)
{
    constexpr int R = 2 * LPR, RPW = 64 / LPR, RSTEP = 4 * RPW, U = 8;
    __shared__ double lds[256 * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rg = lane / LPR;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > n) r1 = n;
    double a0 = 0.0, a1 = 0.0;
    int64_t i = r0 + wave * RPW + rg;
    for (; i + (U - 1) * RSTEP < r1; i += U * RSTEP) {
        double2 z[U];
        double xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            z[u] = *reinterpret_cast<const double2 *>(Z + (i + u * RSTEP) * R + 2 * sub);
            xv[u] = x[i + u * RSTEP];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a0 += z[u].x * xv[u];
            a1 += z[u].y * xv[u];
        }
    }
    for (; i < r1; i += RSTEP) {
        const double2 z = *reinterpret_cast<const double2 *>(Z + i * R + 2 * sub);
        const double xv = x[i];
        a0 += z.x * xv;
        a1 += z.y * xv;
    }
    lds[2 * threadIdx.x] = a0;
    lds[2 * threadIdx.x + 1] = a1;
    __syncthreads();
    if (threadIdx.x < R) {                 // column c: its lanes are sub = c / 2 of every row group
        const int c = threadIdx.x;
        double s = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int g = 0; g < RPW; ++g) s += lds[2 * (64 * w + g * LPR + c / 2) + (c & 1)];
        partial[(int64_t)blockIdx.x * R + c] = s;
    }
}

extern "C" int cm2_Zt_apply(int64_t n, int r, const double *d_Z, const double *d_x, double *d_out,
                            double *d_work, void *stream_)
{
    CM2_CHECK(r >= 1 && r <= 256, "cm2_Zt_apply: deflation rank r=%d out of range [1,256]", r);
    CM2_CHECK(d_Z && d_x && d_out && d_work, "cm2_Zt_apply: NULL argument");
    hipStream_t stream = as_stream(stream_);
    const int rp = pow2_at_least(r);
    if ((r == 16 || r == 32 || r == 64) && n >= 4096 && (reinterpret_cast<uintptr_t>(d_Z) & 15) == 0) {
        const int rstep = 4 * (128 / r);
        int nblk = kRedBlocks;
        int64_t rows = (n + nblk - 1) / nblk;
        rows = ((rows + rstep - 1) / rstep) * rstep;
        nblk = (int)((n + rows - 1) / rows);
        if (r == 16) k_Zt_partial_wide<8><<<nblk, kBlock, 0, stream>>>(n, rows, d_Z, d_x, d_work);
        else if (r == 32) k_Zt_partial_wide<16><<<nblk, kBlock, 0, stream>>>(n, rows, d_Z, d_x, d_work);
        else k_Zt_partial_wide<32><<<nblk, kBlock, 0, stream>>>(n, rows, d_Z, d_x, d_work);
        CM2_LAUNCH_OK();
        k_Zt_final<<<1, 1024, 0, stream>>>(nblk, r, rp, d_work, d_out);
        CM2_LAUNCH_OK();
        return 0;
    }
    const int rstep = 256 / rp;
    int nblk = red_blocks(n);
    int64_t rows = (n + nblk - 1) / nblk;
    rows = ((rows + rstep - 1) / rstep) * rstep;
    if (rows < rstep) rows = rstep;
    nblk = (int)((n + rows - 1) / rows);
    if (nblk < 1) nblk = 1;
    k_Zt_partial<<<nblk, kBlock, 0, stream>>>(n, r, rp, rows, d_Z, d_x, d_work);
    CM2_LAUNCH_OK();
    k_Zt_final<<<1, 1024, 0, stream>>>(nblk, r, rp, d_work, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

// --------------------------------------------------------------- Z y ----------
// linearoperators.py:1047-1049: y = zeros; y += z_k * x_k for k = 0..r-1, so each
// output is ((0 + Z_i0 y0) + Z_i1 y1) + ... in that order.
__global__ __launch_bounds__(256) void k_Z_apply(int64_t n, int r, const double *__restrict__ Z,
                                                  const double *__restrict__ y,
                                                  double *__restrict__ out)
{
    __shared__ double ys[256];
    if (threadIdx.x < r) ys[threadIdx.x] = y[threadIdx.x];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double *zr = Z + i * r;
        double acc = 0.0;
        for (int k = 0; k < r; ++k) acc += zr[k] * ys[k];
        out[i] = acc;
    }
}

extern "C" int cm2_Z_apply(int64_t n, int r, const double *d_Z, const double *d_y, double *d_out,
                           void *stream_)
{
    CM2_CHECK(r >= 1 && r <= 256, "cm2_Z_apply: deflation rank r=%d out of range [1,256]", r);
    CM2_CHECK(d_Z && d_y && d_out, "cm2_Z_apply: NULL argument");
    k_Z_apply<<<grid_for(n), kBlock, 0, as_stream(stream_)>>>(n, r, d_Z, d_y, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------- w += alpha Z y ----------
// The update of the Arnoldi orthogonalisation (w -= V h for a 32-column panel of the basis) and
// any other place where the term order of DeflationLO.mult is not asked for: LPR = r / 2 lanes
// read one row with 16-byte loads, a fixed butterfly adds their partial products, 8 rows per
// lane group in flight.  Streams the panel at HBM speed (k_Z_apply above walks a 256-byte row
// per thread with 8-byte loads: 4.5 TB/s, plus a vector pass for the axpy).
template <int LPR>
__global__ __launch_bounds__(256) void k_Z_axpy_wide(int64_t n, const double *__restrict__ Z,
                                                      const double *__restrict__ y, double alpha,
                                                      double *__restrict__ w)
{
    constexpr int R = 2 * LPR, RPW = 64 / LPR, RSTEP = 4 * RPW, U = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rg = lane / LPR;
    const double y0 = y[2 * sub], y1 = y[2 * sub + 1];
    const int64_t stride = (int64_t)gridDim.x * RSTEP * U;
    for (int64_t base = (int64_t)blockIdx.x * RSTEP * U; base < n; base += stride) {
        double2 z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = base + u * RSTEP + wave * RPW + rg;
            if (i >= n) i = n - 1;
            z[u] = *reinterpret_cast<const double2 *>(Z + i * R + 2 * sub);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double p = z[u].x * y0 + z[u].y * y1;
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
            const int64_t i = base + u * RSTEP + wave * RPW + rg;
            if (sub == 0 && i < n) w[i] += alpha * p;
        }
    }
}

__global__ __launch_bounds__(256) void k_Z_axpy(int64_t n, int r, const double *__restrict__ Z,
                                                 const double *__restrict__ y, double alpha,
                                                 double *__restrict__ w)
{
    __shared__ double ys[256];
    if (threadIdx.x < r) ys[threadIdx.x] = y[threadIdx.x];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double *zr = Z + i * r;
        double acc = 0.0;
        for (int k = 0; k < r; ++k) acc += zr[k] * ys[k];
        w[i] += alpha * acc;
    }
}

extern "C" int cm2_Z_axpy(int64_t n, int r, const double *d_Z, const double *d_y, double alpha,
                          double *d_w, void *stream_)
{
    CM2_CHECK(r >= 1 && r <= 256, "cm2_Z_axpy: r=%d out of range [1,256]", r);
    CM2_CHECK(d_Z && d_y && d_w && d_Z != d_w, "cm2_Z_axpy: NULL or aliased argument");
    if (n <= 0) return 0;
    hipStream_t stream = as_stream(stream_);
    if ((r == 16 || r == 32 || r == 64) && (reinterpret_cast<uintptr_t>(d_Z) & 15) == 0) {
        const int rows_per_blk = 4 * (128 / r) * 8;
        const int64_t nblk = (n + rows_per_blk - 1) / rows_per_blk;
        const int g = (int)(nblk < kNumCU * 16 ? nblk : kNumCU * 16);
        if (r == 16) k_Z_axpy_wide<8><<<g, kBlock, 0, stream>>>(n, d_Z, d_y, alpha, d_w);
        else if (r == 32) k_Z_axpy_wide<16><<<g, kBlock, 0, stream>>>(n, d_Z, d_y, alpha, d_w);
        else k_Z_axpy_wide<32><<<g, kBlock, 0, stream>>>(n, d_Z, d_y, alpha, d_w);
    } else {
        k_Z_axpy<<<grid_for(n), kBlock, 0, stream>>>(n, r, d_Z, d_y, alpha, d_w);
    }
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------- E = Z1^T Z2 ----------
extern "C" int64_t cm2_gemm_tn_work_doubles(int r1, int r2)
{
    return (int64_t)kGemmBlocks * 4 * r1 * r2;
}

// fp64 MFMA path, r1 and r2 multiples of 16, up to 64x64 (16 tiles).
// v_mfma_f64_16x16x4_f64: lane l supplies A[m=l&15][k=l>>4] and B[k=l>>4][n=l&15];
// result reg j of lane l is D[row=(l>>4)+4j][col=l&15].
// Here K runs over map rows: A[m][k] = Z1[i0+k][16*mt+m], B[k][n] = Z2[i0+k][16*nt+n].
template <int MT, int NT>
__global__ __launch_bounds__(256) void k_gemm_tn_mfma(int64_t n, const double *__restrict__ Z1,
                                                       const double *__restrict__ Z2,
                                                       double *__restrict__ work)
{
    constexpr int R1 = MT * 16, R2 = NT * 16;
    const int lane = threadIdx.x & 63;
    const int wave = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nwaves = (int)gridDim.x * 4;
    const int m = lane & 15, k = lane >> 4;
    double4_t acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // contiguous, 4-aligned row range per wave
    int64_t chunk = (n + nwaves - 1) / nwaves;
    chunk = (chunk + 3) & ~(int64_t)3;
    const int64_t i_begin = (int64_t)wave * chunk;
    int64_t i_end = i_begin + chunk;
    if (i_end > n) i_end = n;
    for (int64_t i0 = i_begin; i0 < i_end; i0 += 4) {
        const int64_t i = i0 + k;
        const bool ok = i < i_end;
        double av[MT], bv[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a) av[a] = ok ? Z1[i * R1 + a * 16 + m] : 0.0;
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[b] = ok ? Z2[i * R2 + b * 16 + m] : 0.0;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    double *w = work + (int64_t)wave * R1 * R2;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = a * 16 + (lane >> 4) + 4 * j;
                const int col = b * 16 + (lane & 15);
                w[row * R2 + col] = acc[a][b][j];
            }
}

// scalar path for any r1, r2 (small deflation spaces, e.g. r = 5 in the tests)
__global__ __launch_bounds__(256) void k_gemm_tn_scalar(int64_t n, int r1, int r2,
                                                         const double *__restrict__ Z1,
                                                         const double *__restrict__ Z2,
                                                         double *__restrict__ work)
{
    const int nblk = gridDim.x;
    int64_t chunk = (n + nblk - 1) / nblk;
    const int64_t i_begin = (int64_t)blockIdx.x * chunk;
    int64_t i_end = i_begin + chunk;
    if (i_end > n) i_end = n;
    double *w = work + (int64_t)blockIdx.x * r1 * r2;
    for (int e = threadIdx.x; e < r1 * r2; e += blockDim.x) {
        const int a = e / r2, b = e % r2;
        double acc = 0.0;
        for (int64_t i = i_begin; i < i_end; ++i) acc += Z1[i * r1 + a] * Z2[i * r2 + b];
        w[e] = acc;
    }
}

// 16 entries of E per workgroup, 16 threads per entry (each adds every 16th partial matrix, then
// the 16 sums are added in order): 512 partials in 32 dependent loads instead of 512
__global__ __launch_bounds__(256) void k_gemm_tn_final(int nparts, int rr,
                                                        const double *__restrict__ work,
                                                        double *__restrict__ E)
{
    __shared__ double lds[256];
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    double acc = 0.0;
    if (e < rr)
        for (int p = q; p < nparts; p += 16) acc += work[(int64_t)p * rr + e];
    lds[threadIdx.x] = acc;
    __syncthreads();
    if (q == 0 && e < rr) {
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += lds[16 * k + el];
        E[e] = s;
    }
}

// The same contraction for R = 32 h (h = 1, 2) with 16-byte loads: lane (m, k) reads the column
// PAIR (32 g + 2 m, 32 g + 2 m + 1) of row i0 + k, which feeds the MFMA tiles 2 g and 2 g + 1 --
// tile a = 2 g + e holds the columns 32 g + 2 m + e, m = 0..15 -- so one dwordx4 load supplies
// two A (or B) operands; the tiles are un-permuted when the partial E is written.  Four row
// quads are in flight per wave, and a workgroup's four waves are combined in LDS, so that 512
// workgroups (two per CU) leave 512 partial matrices like the kernel above.
template <int H>
__global__ __launch_bounds__(256) void k_gemm_tn_mfma_pairs(int64_t n, const double *__restrict__ Z1,
                                                             const double *__restrict__ Z2,
                                                             double *__restrict__ work)
{
    constexpr int R = 32 * H, T = 2 * H, U = 4;
    __shared__ double red[3][R * R > 1024 ? 1024 : R * R];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave = (int)(blockIdx.x * 4 + w);
    const int nwaves = (int)gridDim.x * 4;
    const int m = lane & 15, k = lane >> 4;
    double4_t acc[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    int64_t chunk = (n + nwaves - 1) / nwaves;
    chunk = (chunk + 3) & ~(int64_t)3;
    const int64_t i_begin = (int64_t)wave * chunk;
    int64_t i_end = i_begin + chunk;
    if (i_end > n) i_end = n;
    int64_t i0 = i_begin;
    for (; i0 + 4 * U <= i_end; i0 += 4 * U) {
        double2 av[U][H], bv[U][H];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + 4 * u + k;
#pragma unroll
            for (int g = 0; g < H; ++g) {
                av[u][g] = *reinterpret_cast<const double2 *>(Z1 + i * R + 32 * g + 2 * m);
                bv[u][g] = *reinterpret_cast<const double2 *>(Z2 + i * R + 32 * g + 2 * m);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    const double x = (a & 1) ? av[u][a >> 1].y : av[u][a >> 1].x;
                    const double y = (b & 1) ? bv[u][b >> 1].y : bv[u][b >> 1].x;
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a][b], 0, 0, 0);
                }
    }
    for (; i0 < i_end; i0 += 4) {
        const int64_t i = i0 + k;
        const bool ok = i < i_end;
        double2 av[H], bv[H];
#pragma unroll
        for (int g = 0; g < H; ++g) {
            av[g] = ok ? *reinterpret_cast<const double2 *>(Z1 + i * R + 32 * g + 2 * m) : make_double2(0.0, 0.0);
            bv[g] = ok ? *reinterpret_cast<const double2 *>(Z2 + i * R + 32 * g + 2 * m) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const double x = (a & 1) ? av[a >> 1].y : av[a >> 1].x;
                const double y = (b & 1) ? bv[b >> 1].y : bv[b >> 1].x;
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a][b], 0, 0, 0);
            }
    }
    // E entry of accumulator register j of tile (a, b): row 32 (a >> 1) + 2 ((lane >> 4) + 4 j) + (a & 1),
    // column 32 (b >> 1) + 2 (lane & 15) + (b & 1).  Waves 1..3 hand their partials to wave 0
    // through LDS, one 1024-entry slab at a time (R = 64: four slabs), added in wave order.
    double *out = work + (int64_t)blockIdx.x * R * R;
    constexpr int SLAB = 1024, NSLAB = R * R / SLAB;
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        // slab sl = rows [16 sl * 32 / R ...): with R = 32 one slab is the whole matrix; with
        // R = 64 slab sl holds the E rows 16 sl .. 16 sl + 15
        __syncthreads();
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 32 * (a >> 1) + 2 * ((lane >> 4) + 4 * j) + (a & 1);
                    const int col = 32 * (b >> 1) + 2 * (lane & 15) + (b & 1);
                    const int e = row * R + col;
                    if (e / SLAB == sl) {
                        if (w > 0) red[w - 1][e % SLAB] = acc[a][b][j];
                    }
                }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = 32 * (a >> 1) + 2 * ((lane >> 4) + 4 * j) + (a & 1);
                        const int col = 32 * (b >> 1) + 2 * (lane & 15) + (b & 1);
                        const int e = row * R + col;
                        if (e / SLAB == sl)
                            out[e] = ((acc[a][b][j] + red[0][e % SLAB]) + red[1][e % SLAB]) + red[2][e % SLAB];
                    }
        }
    }
}

extern "C" int cm2_gemm_tn(int64_t n, int r1, int r2, const double *d_Z1, const double *d_Z2,
                           double *d_E, double *d_work, void *stream_)
{
    CM2_CHECK(r1 >= 1 && r1 <= 256 && r2 >= 1 && r2 <= 256, "cm2_gemm_tn: r1=%d r2=%d out of range",
              r1, r2);
    CM2_CHECK(d_Z1 && d_Z2 && d_E && d_work, "cm2_gemm_tn: NULL argument");
    hipStream_t stream = as_stream(stream_);
    const int rr = r1 * r2;
    int nparts;
    const bool mfma = (r1 % 16 == 0) && (r2 % 16 == 0) && r1 <= 64 && r2 <= 64 && r1 == r2;
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_Z1) | reinterpret_cast<uintptr_t>(d_Z2)) & 15) == 0;
    if (mfma && aligned && (r1 == 32 || r1 == 64)) {
        nparts = kGemmBlocks * 4;                       // one partial per workgroup
        if (r1 == 32) k_gemm_tn_mfma_pairs<1><<<nparts, kBlock, 0, stream>>>(n, d_Z1, d_Z2, d_work);
        else k_gemm_tn_mfma_pairs<2><<<nparts, kBlock, 0, stream>>>(n, d_Z1, d_Z2, d_work);
    } else if (mfma) {
        nparts = kGemmBlocks * 4;
        if (r1 == 16) k_gemm_tn_mfma<1, 1><<<kGemmBlocks, kBlock, 0, stream>>>(n, d_Z1, d_Z2, d_work);
        else if (r1 == 32) k_gemm_tn_mfma<2, 2><<<kGemmBlocks, kBlock, 0, stream>>>(n, d_Z1, d_Z2, d_work);
        else if (r1 == 48) k_gemm_tn_mfma<3, 3><<<kGemmBlocks, kBlock, 0, stream>>>(n, d_Z1, d_Z2, d_work);
        else k_gemm_tn_mfma<4, 4><<<kGemmBlocks, kBlock, 0, stream>>>(n, d_Z1, d_Z2, d_work);
    } else {
        nparts = kGemmBlocks * 4;
        k_gemm_tn_scalar<<<nparts, kBlock, 0, stream>>>(n, r1, r2, d_Z1, d_Z2, d_work);
    }
    CM2_LAUNCH_OK();
    k_gemm_tn_final<<<(rr + 15) / 16, kBlock, 0, stream>>>(nparts, rr, d_work, d_E);
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------------- out (+)= P W  (tall panel x small) ------
// out[n x R] (+)= P[n x 32] W[32 x R], R = 16 or 32, all row-major: the products of the
// deflation build -- Ritz vectors Z = V U and, through the Arnoldi relation A V_m = P_{m+1} H,
// A Z = P (H U) -- with the basis kept in 32-column panels.  fp64 MFMA 16x16x4 per 16-row block:
// lane (m, k) reads the column pair (8 s + 2 k, 8 s + 2 k + 1) of row i0 + m with one 16-byte load
// (four loads = the lane's share of the 4-KB block) and feeds .x / .y to two MFMAs whose B
// operands are the matching even / odd rows of W, held in registers.  One pass over P and out:
// HBM-bound.
template <int CT>
__global__ __launch_bounds__(256) void k_panel_gemm_mfma(int64_t n, const double *__restrict__ P,
                                                          const double *__restrict__ W,
                                                          double *__restrict__ out, int accumulate)
{
    constexpr int R = 16 * CT;
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, k = lane >> 4;
    // B operands: wv[s][e][ct] = W[8 s + 2 k + e][16 ct + m]
    double wv[4][2][CT];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wv[s_][e][ct] = W[(8 * s_ + 2 * k + e) * R + 16 * ct + m];
    const int64_t nblk = (n + 15) / 16;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        const int64_t i0 = blk * 16;
        const int64_t row = i0 + m < n ? i0 + m : n - 1;
        double2 a[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
            a[s_] = *reinterpret_cast<const double2 *>(P + row * 32 + 8 * s_ + 2 * k);
        double4_t acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            acc[ct] = (double4_t){0.0, 0.0, 0.0, 0.0};
            if (accumulate) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t r_ = i0 + k + 4 * j;
                    acc[ct][j] = r_ < n ? out[r_ * R + 16 * ct + m] : 0.0;
                }
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s_].x, wv[s_][0][ct], acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s_].y, wv[s_][1][ct], acc[ct], 0, 0, 0);
            }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r_ = i0 + k + 4 * j;
                if (r_ < n) out[r_ * R + 16 * ct + m] = acc[ct][j];
            }
    }
}

// any panel width / output width (small cases, tests)
__global__ __launch_bounds__(256) void k_panel_gemm_scalar(int64_t n, int rin, int rout,
                                                            const double *__restrict__ P,
                                                            const double *__restrict__ W,
                                                            double *__restrict__ out, int accumulate)
{
    const int64_t total = n * rout;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / rout;
        const int c = (int)(e - i * rout);
        double acc = accumulate ? out[e] : 0.0;
        for (int l = 0; l < rin; ++l) acc += P[i * rin + l] * W[l * rout + c];
        out[e] = acc;
    }
}

extern "C" int cm2_panel_gemm(int64_t n, int rin, int rout, const double *d_P, const double *d_W,
                              double *d_out, int accumulate, void *stream_)
{
    CM2_CHECK(n >= 0 && rin >= 1 && rin <= 256 && rout >= 1 && rout <= 256,
              "cm2_panel_gemm: bad shape n=%lld rin=%d rout=%d", (long long)n, rin, rout);
    if (n == 0) return 0;
    CM2_CHECK(d_P && d_W && d_out && d_P != d_out, "cm2_panel_gemm: NULL or aliased argument");
    hipStream_t stream = as_stream(stream_);
    const bool aligned = (reinterpret_cast<uintptr_t>(d_P) & 15) == 0;
    if (rin == 32 && (rout == 16 || rout == 32) && aligned) {
        const int64_t nblk = (n + 15) / 16;
        const int g = (int)(nblk / 4 + 1 < kNumCU * 8 ? nblk / 4 + 1 : kNumCU * 8);
        if (rout == 16) k_panel_gemm_mfma<1><<<g, kBlock, 0, stream>>>(n, d_P, d_W, d_out, accumulate);
        else k_panel_gemm_mfma<2><<<g, kBlock, 0, stream>>>(n, d_P, d_W, d_out, accumulate);
    } else {
        k_panel_gemm_scalar<<<grid_for(n * rout), kBlock, 0, stream>>>(n, rin, rout, d_P, d_W, d_out,
                                                                    accumulate);
    }
    CM2_LAUNCH_OK();
    return 0;
}

// --------------------------------------------------- out = M v  (r x r) -------
__global__ __launch_bounds__(256) void k_small_matvec(int r, const double *__restrict__ M,
                                                       const double *__restrict__ v,
                                                       double *__restrict__ out)
{
    __shared__ double vs[256];
    if (threadIdx.x < r) vs[threadIdx.x] = v[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < r) {
        double acc = 0.0;
        for (int l = 0; l < r; ++l) acc += M[threadIdx.x * r + l] * vs[l];
        out[threadIdx.x] = acc;
    }
}

extern "C" int cm2_small_matvec(int r, const double *d_M, const double *d_v, double *d_out,
                                void *stream_)
{
    CM2_CHECK(r >= 1 && r <= 256, "cm2_small_matvec: r=%d out of range [1,256]", r);
    CM2_CHECK(d_M && d_v && d_out, "cm2_small_matvec: NULL argument");
    k_small_matvec<<<1, kBlock, 0, as_stream(stream_)>>>(r, d_M, d_v, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------ M2 r = M_BD (r - AZ y) + Z y  (tail) -----
// One thread per pixel: reads its POL rows of Z and AZ once, forms
// t = res - AZ y and zy = Z y (k = 0..r-1 in order, as DeflationLO.mult),
// applies the closed-form M_BD block to t and adds zy.
template <int POL>
__global__ __launch_bounds__(256) void k_m2_finish(
    int64_t npix, int r, const double *__restrict__ Z, const double *__restrict__ AZ,
    const double *__restrict__ y, const double *__restrict__ res,
    const double *__restrict__ hits, const double *__restrict__ c, const double *__restrict__ s,
    const double *__restrict__ c2, const double *__restrict__ s2, const double *__restrict__ cs,
    const double *__restrict__ det, const uint8_t *__restrict__ mask, double *__restrict__ out)
{
    __shared__ double ys[256];
    if (threadIdx.x < r) ys[threadIdx.x] = y[threadIdx.x];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < npix; j += stride) {
        double t[3], zy[3], o[3];
#pragma unroll
        for (int a = 0; a < POL; ++a) {
            const int64_t i = POL * j + a;
            const double *zr = Z + i * r;
            const double *ar = AZ + i * r;
            double accz = 0.0, acca = 0.0;
            for (int k = 0; k < r; ++k) {
                accz += zr[k] * ys[k];
                acca += ar[k] * ys[k];
            }
            zy[a] = accz;
            t[a] = res[i] - acca;
        }
        const bool m = mask[j] != 0;
        if (POL == 1)
            bd_inverse_block<1>(hits[j], 0, 0, 0, 0, 0, 0, m, t, o);
        else if (POL == 2)
            bd_inverse_block<2>(0, 0, 0, c2[j], s2[j], cs[j], det[j], m, t, o);
        else
            bd_inverse_block<3>(hits[j], c[j], s[j], c2[j], s2[j], cs[j], det[j], m, t, o);
#pragma unroll
        for (int a = 0; a < POL; ++a) out[POL * j + a] = o[a] + zy[a];
    }
}

// Wide form of the same tail for r = 2 LPR (r = 16, 32, 64): a workgroup takes 64 pixels
// (64 POL rows of Z and of AZ); LPR lanes read one row with 16-byte loads, form the two partial
// products with y, and a fixed butterfly over the LPR lanes gives Z y and AZ y of that row; the
// row results go through LDS to one thread per pixel, which applies the M_BD block.  Z and AZ
// are streamed once at HBM speed (the thread-per-pixel kernel above walks each 256-byte row
// with 8-byte loads: 3.4 TB/s).
template <int POL, int LPR>
__global__ __launch_bounds__(256) void k_m2_finish_wide(
    int64_t npix, const double *__restrict__ Z, const double *__restrict__ AZ,
    const double *__restrict__ y, const double *__restrict__ res,
    const double *__restrict__ hits, const double *__restrict__ c, const double *__restrict__ s,
    const double *__restrict__ c2, const double *__restrict__ s2, const double *__restrict__ cs,
    const double *__restrict__ det, const uint8_t *__restrict__ mask, double *__restrict__ out)
{
    constexpr int R = 2 * LPR, RPW = 64 / LPR, PB = 64, ROWS = PB * POL, RSTEP = 4 * RPW;
    constexpr int NIT = (ROWS + RSTEP - 1) / RSTEP;
    __shared__ double zy[ROWS], azy[ROWS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rg = lane / LPR;
    const double y0 = y[2 * sub], y1 = y[2 * sub + 1];
    const int64_t nblk = (npix + PB - 1) / PB;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row0 = blk * ROWS;
        int64_t nrows = npix * POL - row0;
        if (nrows > ROWS) nrows = ROWS;
        double2 zv[NIT], av[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rl = it * RSTEP + wave * RPW + rg;
            const int64_t i = row0 + (rl < nrows ? rl : nrows - 1);
            zv[it] = *reinterpret_cast<const double2 *>(Z + i * R + 2 * sub);
            av[it] = *reinterpret_cast<const double2 *>(AZ + i * R + 2 * sub);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            double pz = zv[it].x * y0 + zv[it].y * y1;
            double pa = av[it].x * y0 + av[it].y * y1;
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) {
                pz += __shfl_xor(pz, off, 64);
                pa += __shfl_xor(pa, off, 64);
            }
            const int rl = it * RSTEP + wave * RPW + rg;
            if (sub == 0 && rl < ROWS) {
                zy[rl] = pz;
                azy[rl] = pa;
            }
        }
        __syncthreads();
        const int64_t j = blk * PB + threadIdx.x;
        if (threadIdx.x < PB && j < npix) {
            double t[3], o[3];
#pragma unroll
            for (int a = 0; a < POL; ++a) t[a] = res[POL * j + a] - azy[POL * threadIdx.x + a];
            const bool m = mask[j] != 0;
            if (POL == 1)
                bd_inverse_block<1>(hits[j], 0, 0, 0, 0, 0, 0, m, t, o);
            else if (POL == 2)
                bd_inverse_block<2>(0, 0, 0, c2[j], s2[j], cs[j], det[j], m, t, o);
            else
                bd_inverse_block<3>(hits[j], c[j], s[j], c2[j], s2[j], cs[j], det[j], m, t, o);
#pragma unroll
            for (int a = 0; a < POL; ++a) out[POL * j + a] = o[a] + zy[POL * threadIdx.x + a];
        }
        __syncthreads();
    }
}

extern "C" int cm2_m2_finish(int pol, int64_t npix, int r, const double *d_Z, const double *d_AZ,
                             const double *d_y, const double *d_res, const double *d_counts,
                             const double *d_cosine, const double *d_sine, const double *d_cos2,
                             const double *d_sin2, const double *d_sincos, const double *d_det,
                             const uint8_t *d_mask, double *d_out, void *stream_)
{
    CM2_CHECK(pol == 1 || pol == 2 || pol == 3, "cm2_m2_finish: bad pol=%d", pol);
    CM2_CHECK(r >= 1 && r <= 256, "cm2_m2_finish: r=%d out of range [1,256]", r);
    CM2_CHECK(d_Z && d_AZ && d_y && d_res && d_mask && d_out, "cm2_m2_finish: NULL argument");
    hipStream_t stream = as_stream(stream_);
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_Z) | reinterpret_cast<uintptr_t>(d_AZ)) & 15) == 0;
    if ((r == 16 || r == 32 || r == 64) && aligned && npix >= 64) {
        const int64_t nblk = (npix + 63) / 64;
        const int g = (int)(nblk < kNumCU * 16 ? nblk : kNumCU * 16);
#define CM2_M2W(POL, LPR)                                                                       \
    k_m2_finish_wide<POL, LPR><<<g, kBlock, 0, stream>>>(npix, d_Z, d_AZ, d_y, d_res, d_counts, \
                                                        d_cosine, d_sine, d_cos2, d_sin2,      \
                                                        d_sincos, d_det, d_mask, d_out)
#define CM2_M2WP(POL)                                                                           \
    do {                                                                                        \
        if (r == 16) CM2_M2W(POL, 8); else if (r == 32) CM2_M2W(POL, 16); else CM2_M2W(POL, 32); \
    } while (0)
        if (pol == 1) CM2_M2WP(1); else if (pol == 2) CM2_M2WP(2); else CM2_M2WP(3);
#undef CM2_M2WP
#undef CM2_M2W
        CM2_LAUNCH_OK();
        return 0;
    }
    const int g = grid_for(npix);
#define CM2_M2(POL)                                                                          \
    k_m2_finish<POL><<<g, kBlock, 0, stream>>>(npix, r, d_Z, d_AZ, d_y, d_res, d_counts,     \
                                               d_cosine, d_sine, d_cos2, d_sin2, d_sincos,   \
                                               d_det, d_mask, d_out)
    if (pol == 1) CM2_M2(1); else if (pol == 2) CM2_M2(2); else CM2_M2(3);
#undef CM2_M2
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------ C[m x n] = A^T B^T  (general, API edge) ---
// utilities/linear_algebra_funcs.py:16-29: gemm(a=A.T, b=B, trans_b=True).
// A is k x m row-major, B is n x k row-major.  One thread per output element;
// used for the small host-facing products (r x r, npix x r with tiny k), the large
// contraction E = Z^T (A Z) goes through cm2_gemm_tn.
__global__ __launch_bounds__(256) void k_gemm_atbt(int64_t m, int64_t n, int64_t k,
                                                    const double *__restrict__ A,
                                                    const double *__restrict__ B,
                                                    double *__restrict__ C)
{
    const int64_t total = m * n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / n, j = e % n;
        double acc = 0.0;
        for (int64_t l = 0; l < k; ++l) acc += A[l * m + i] * B[j * k + l];
        C[e] = acc;
    }
}

extern "C" int cm2_gemm_atbt(int64_t m, int64_t n, int64_t k, const double *d_A,
                             const double *d_B, double *d_C, void *stream_)
{
    CM2_CHECK(m >= 1 && n >= 1 && k >= 1, "cm2_gemm_atbt: bad shape m=%lld n=%lld k=%lld",
              (long long)m, (long long)n, (long long)k);
    CM2_CHECK(d_A && d_B && d_C, "cm2_gemm_atbt: NULL argument");
    k_gemm_atbt<<<grid_for(m * n), kBlock, 0, as_stream(stream_)>>>(m, n, k, d_A, d_B, d_C);
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------------- out[c][r] = in[r][c]  (layout change) ---
// Moves a set of map vectors between "one contiguous vector per column" (what the matvec
// wants) and the row-major n x r panels the deflation kernels stream (rows x cols in, cols x
// rows out).  32 x 32 tiles through LDS (33-double pitch), both sides coalesced.
__global__ __launch_bounds__(256) void k_transpose(int64_t rows, int64_t cols,
                                                    const double *__restrict__ in,
                                                    double *__restrict__ out)
{
    __shared__ double tile[32][33];
    const int64_t tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    for (int64_t tid = blockIdx.x; tid < tiles_c * tiles_r; tid += gridDim.x) {
        const int64_t r0 = (tid / tiles_c) * 32, c0 = (tid % tiles_c) * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + ty + 8 * j, c = c0 + tx;
            if (r < rows && c < cols) tile[ty + 8 * j][tx] = in[r * cols + c];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + 8 * j];
        }
        __syncthreads();
    }
}

extern "C" int cm2_transpose(int64_t rows, int64_t cols, const double *d_in, double *d_out,
                             void *stream_)
{
    CM2_CHECK(rows >= 0 && cols >= 0, "cm2_transpose: negative size");
    if (rows == 0 || cols == 0) return 0;
    CM2_CHECK(d_in && d_out && d_in != d_out, "cm2_transpose: NULL or aliased argument");
    const int64_t tiles = ((rows + 31) / 32) * ((cols + 31) / 32);
    const int grid = (int)(tiles < kNumCU * 16 ? tiles : kNumCU * 16);
    k_transpose<<<grid, kBlock, 0, as_stream(stream_)>>>(rows, cols, d_in, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

// cos(2 phi), sin(2 phi) of the HWP / polarisation angle (process_ces.py:493-494)
__global__ __launch_bounds__(256) void k_trig2(int64_t n, const double *__restrict__ phi,
                                                double *__restrict__ c, double *__restrict__ s)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double sv, cv;
        sincos(2. * phi[i], &sv, &cv);
        c[i] = cv;
        s[i] = sv;
    }
}

extern "C" int cm2_cos_sin_2phi(int64_t n, const double *d_phi, double *d_cos, double *d_sin,
                                void *stream_)
{
    CM2_CHECK(n == 0 || (d_phi && d_cos && d_sin), "cm2_cos_sin_2phi: NULL argument");
    if (n == 0) return 0;
    k_trig2<<<grid_for(n), kBlock, 0, as_stream(stream_)>>>(n, d_phi, d_cos, d_sin);
    CM2_LAUNCH_OK();
    return 0;
}
