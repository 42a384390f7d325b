// smc_filter_mv.h -- propagate kernel of the fused step loop for the
// multivariate linear Gaussian model (particles/kalman.py:296-361
// MVLinearGauss) under the bootstrap (state_space_models.py:299-349) and the
// guided filter with the model's optimal proposal (state_space_models.py:
// 352-398, kalman.py:348-356, filter_step kalman.py:196-229).
//
// Per particle, with the (d,d) matrices shared by all particles:
//   bootstrap  x = F xp + L_X z                     log G = log N(y; G x, covY)
//   guided     m = F xp ; mu = m + K (y - G m) = B xp + K y ;  x = mu + L_P z
//              log G = log N(x; m, covX) + log N(y; G x, covY) - log N(x; mu, P)
// (distributions.py:946-959: rvs = loc + Z L^T, logpdf via L^-1 (x-loc)).
// The last term needs no solve: x - mu = L_P z, so it is -|z|^2/2 - c_P.
// Triangular solves become products with the precomputed inverse factors.
//
// Mapping (MI355X): one thread per particle, 256 particles per workgroup.  The
// particle's current vector lives in LDS as a COLUMN (element k of thread p at
// V[k*256+p]: conflict-free 8-byte accesses), the matrix is walked column by
// column through SCALAR loads (it is the same for every lane) and each column
// feeds DP independent v_fma_f64 accumulators held in registers.  The matrix in
// use sits in LDS (8 KB, staged by the workgroup while the previous product
// runs) and is read with broadcast ds_read_b128 -- per-lane global or scalar
// loads inside the product loop left it waiting on memory every column.  Rows of X are gathered /
// scattered cooperatively (16 lanes per 256-byte row) so global accesses stay
// coalesced whole lines.  No MFMA: fp64 matrix and vector peaks are equal on
// this part and the operands here are already register/SGPR resident.
#pragma once
#include "smc_filter_kernels.h"

// ---- layout of the constants block `mvc` (doubles); every matrix is stored
// transposed and zero-padded to DP x DP:  Mt[k*DP + i] = M[i][k]
#define MV_F 0        /* F                                   */
#define MV_B 1        /* (I - K G) F            (guided)     */
#define MV_LZ 2       /* factor applied to z, t >= 1: L_X (bootstrap) / L_P (guided) */
#define MV_LZ0 3      /* same at t = 0: L_0 / L_P0            */
#define MV_XINV 4     /* L_X^-1                  (guided)     */
#define MV_X0INV 5    /* L_0^-1                  (guided)     */
#define MV_NGY 6      /* -(L_Y^-1 G)                          */
#define MV_NMAT 7
#define MV_VEC(dp) (MV_NMAT * (dp) * (dp))         /* mu0[dp], mup0[dp] */
#define MV_SCAL(dp) (MV_VEC(dp) + 2 * (dp))        /* cX, cY, cP, c0, cP0, -, -, - */
#define MV_STEP(dp) (MV_SCAL(dp) + 8)              /* per t: yw_t[dp] = L_Y^-1 y_t, ky_t[dp] = K y_t */
#define MV_SIZE(dp, T) (MV_STEP(dp) + 2 * (size_t)(dp) * (T))

// Stage one DP x DP matrix of the constants block into LDS (all threads call).
// `pf` holds the thread's share: mv_fetch issues the global loads (early, so
// their latency hides behind the previous product), mv_commit parks them in LDS.
template <int DP>
struct MvShare { double v[(DP * DP + SMC_BLOCK - 1) / SMC_BLOCK]; };
template <int DP>
__device__ __forceinline__ void mv_fetch(const double* __restrict__ gM, MvShare<DP>& pf)
{
#pragma unroll
    for (int j = 0; j < (DP * DP + SMC_BLOCK - 1) / SMC_BLOCK; ++j) {
        const int i = j * SMC_BLOCK + (int)threadIdx.x;
        pf.v[j] = (i < DP * DP) ? gM[i] : 0.0;
    }
}
template <int DP>
__device__ __forceinline__ void mv_commit(const MvShare<DP>& pf, double* sM)
{
    __syncthreads();                      // the previous product has finished reading sM
#pragma unroll
    for (int j = 0; j < (DP * DP + SMC_BLOCK - 1) / SMC_BLOCK; ++j) {
        const int i = j * SMC_BLOCK + (int)threadIdx.x;
        if (i < DP * DP) sM[i] = pf.v[j];
    }
    __syncthreads();
}

template <int DP>
__device__ __forceinline__ void mv_matvec(const double* sM, const double* vcol, double (&acc)[DP])
{
    // acc += M v ; column k of M (contiguous in sM, same address for every lane:
    // LDS broadcast reads) scaled by v_k
#pragma unroll 2
    for (int k = 0; k < DP; ++k) {
        const double vk = vcol[k * SMC_BLOCK];
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] = fma(sM[k * DP + i], vk, acc[i]);
    }
}

// rows[p] (>= 0) of a (.., d) row-major array -> columns of V; 16-byte chunks,
// consecutive lanes on consecutive chunks of the same row
template <int DP>
__device__ __forceinline__ void mv_load_rows(const double* base, int d, const i64* sRow, double* V)
{
    const int tid = (int)threadIdx.x;
    if (d == DP) {
        constexpr int CPR = DP / 2;                 // 16-byte chunks per row
        constexpr int RPI = SMC_BLOCK / CPR;        // rows per pass over the workgroup
#pragma unroll 4
        for (int j = 0; j < CPR; ++j) {
            const int p = tid / CPR + RPI * j, c = tid % CPR;
            const i64 r = sRow[p];
            F2d v;
            v.a = 0.0; v.b = 0.0;
            if (r >= 0) v = *reinterpret_cast<const F2d*>(base + r * d + 2 * c);
            V[(2 * c) * SMC_BLOCK + p] = v.a;
            V[(2 * c + 1) * SMC_BLOCK + p] = v.b;
        }
    } else {
        const i64 r = sRow[tid];
        for (int k = 0; k < DP; ++k) V[k * SMC_BLOCK + tid] = (k < d && r >= 0) ? base[r * d + k] : 0.0;
    }
}
template <int DP>
__device__ __forceinline__ void mv_store_rows(double* base, int d, const i64* sRow, const double* V)
{
    const int tid = (int)threadIdx.x;
    if (d == DP) {
        constexpr int CPR = DP / 2;
        constexpr int RPI = SMC_BLOCK / CPR;
#pragma unroll 4
        for (int j = 0; j < CPR; ++j) {
            const int p = tid / CPR + RPI * j, c = tid % CPR;
            const i64 r = sRow[p];
            if (r >= 0) {
                F2d v;
                v.a = V[(2 * c) * SMC_BLOCK + p];
                v.b = V[(2 * c + 1) * SMC_BLOCK + p];
                *reinterpret_cast<F2d*>(base + r * d + 2 * c) = v;
            }
        }
    } else {
        const i64 r = sRow[tid];
        if (r >= 0)
            for (int k = 0; k < d; ++k) base[r * d + k] = V[k * SMC_BLOCK + tid];
    }
}

template <int FK, int DP>
__global__ void __launch_bounds__(SMC_BLOCK)
k_propagate_mv(const FArgs* __restrict__ ap, const double* __restrict__ C)
{
    // C (= ap->mvc) comes in as its own const __restrict__ kernel argument: only then
    // does the compiler know the block is never written and fetch the matrix columns
    // with SCALAR loads (s_load_dwordx16) instead of 64 identical vector loads
    const FArgs& a = *ap;
    __shared__ double sV0[DP * SMC_BLOCK];
    __shared__ double sV1[FK == SMC_FK_GUIDED ? DP * SMC_BLOCK : 1];
    __shared__ __attribute__((aligned(16))) double sM[DP * DP];     // the matrix in use
    __shared__ i64 sRow[SMC_BLOCK];
    __shared__ double smd[SMC_SM];
    __shared__ int s_last;
    const int b = (int)blockIdx.x, isl = (int)blockIdx.y;
    const int tid = (int)threadIdx.x;
    MvShare<DP> pf;
    double* info = a.info + (i64)isl * INFO_STRIDE;
    const i64 t = (i64)info[0];
    if (t >= a.T) return;
    const i64 N = a.N;
    const int d = a.dx;
    const u32 gisl = (u32)(a.island_offset + isl);
    const int cur = (int)(t & 1);
    double* Xn = (cur ? a.X1 : a.X0) + (i64)isl * N * d;
    const double* Xo = (cur ? a.X0 : a.X1) + (i64)isl * N * d;
    double* lwn = (cur ? a.lw1 : a.lw0) + (i64)isl * N;
    const double* lwo = (cur ? a.lw0 : a.lw1) + (i64)isl * N;
    const i64* A = a.A + (i64)isl * N;
    const double* zt = a.zt ? a.zt + ((i64)t * a.n_islands + isl) * N * d : nullptr;
    const bool first = (t == 0);
    const bool resample = !first && info[1] != 0.0;
    const i64 n = (i64)b * SMC_BLOCK + tid;
    const bool valid = n < N;
    const double* scal = C + MV_SCAL(DP);
    const double* yw = C + MV_STEP(DP) + (size_t)t * 2 * DP;
    const double* ky = yw + DP;
    double* V0 = sV0 + tid;          // this thread's column
    double* V1 = sV1 + tid;
    double acc[DP];

    mv_fetch<DP>(C + (first ? MV_LZ0 : MV_F) * DP * DP, pf);
    // ---- the parents' rows -> V0
    if (!first) {
        sRow[tid] = valid ? (resample ? A[n] : n) : -1;                    // core.py:332 / :336
        __syncthreads();
        mv_load_rows<DP>(Xo, d, sRow, sV0);
        __syncthreads();
    }
    // ---- mean of the proposal: acc = mu
    if (first) {
        const double* mu = C + MV_VEC(DP) + (FK == SMC_FK_GUIDED ? DP : 0);
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] = mu[i];
    } else {
        mv_commit<DP>(pf, sM);                                             // F
        if (FK == SMC_FK_GUIDED) {
            mv_fetch<DP>(C + MV_B * DP * DP, pf);
#pragma unroll
            for (int i = 0; i < DP; ++i) acc[i] = 0.0;
            mv_matvec<DP>(sM, V0, acc);                                    // m = F xp
#pragma unroll
            for (int i = 0; i < DP; ++i) V1[i * SMC_BLOCK] = acc[i];
#pragma unroll
            for (int i = 0; i < DP; ++i) acc[i] = ky[i];
            mv_commit<DP>(pf, sM);                                         // B
            mv_fetch<DP>(C + MV_LZ * DP * DP, pf);
            mv_matvec<DP>(sM, V0, acc);                                    // mu = B xp + K y
        } else {
            mv_fetch<DP>(C + MV_LZ * DP * DP, pf);
#pragma unroll
            for (int i = 0; i < DP; ++i) acc[i] = 0.0;
            mv_matvec<DP>(sM, V0, acc);                                    // mu = F xp
        }
    }
    // ---- z -> V0 (the parent row is no longer needed), zz = |z|^2
    double zz = 0.0;
    __syncthreads();                        // everyone is done with the rows in sV0
    if (zt) {
        sRow[tid] = valid ? n : -1;
        __syncthreads();
        mv_load_rows<DP>(zt, d, sRow, sV0);
        __syncthreads();
        for (int k = 0; k < DP; ++k) { const double z = V0[k * SMC_BLOCK]; zz = fma(z, z, zz); }
    } else {
        const int hp = (d + 1) / 2;
        for (int kp = 0; kp < DP / 2; ++kp) {
            double z0 = 0.0, z1 = 0.0;
            if (valid && 2 * kp < d) {
                smc_normal_pair(a.seed, (u32)(n * hp + kp), (u32)t, gisl, SMC_STREAM_NORMAL, z0, z1);
                if (2 * kp + 1 >= d) z1 = 0.0;
            }
            V0[(2 * kp) * SMC_BLOCK] = z0;
            V0[(2 * kp + 1) * SMC_BLOCK] = z1;
            zz = fma(z0, z0, fma(z1, z1, zz));
        }
    }
    // ---- x = mu + L z
    mv_commit<DP>(pf, sM);                                                 // L_z
    mv_fetch<DP>(C + (FK == SMC_FK_GUIDED ? (first ? MV_X0INV : MV_XINV) : MV_NGY) * DP * DP, pf);
    mv_matvec<DP>(sM, V0, acc);
    double uu = 0.0;
    if (FK == SMC_FK_GUIDED) {
        const double* mu0 = C + MV_VEC(DP);
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            const double mi = first ? mu0[i] : V1[i * SMC_BLOCK];
            V1[i * SMC_BLOCK] = acc[i] - mi;                               // x - m
        }
    }
#pragma unroll
    for (int i = 0; i < DP; ++i) V0[i * SMC_BLOCK] = acc[i];               // x
    if (FK == SMC_FK_GUIDED) {
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] = 0.0;
        mv_commit<DP>(pf, sM);                                             // L_X^-1
        mv_fetch<DP>(C + MV_NGY * DP * DP, pf);
        mv_matvec<DP>(sM, V1, acc);
#pragma unroll
        for (int i = 0; i < DP; ++i) uu = fma(acc[i], acc[i], uu);        // |L^-1 (x - m)|^2
    }
    // ---- w = L_Y^-1 (y - G x)
#pragma unroll
    for (int i = 0; i < DP; ++i) acc[i] = yw[i];
    mv_commit<DP>(pf, sM);                                                 // -(L_Y^-1 G)
    mv_matvec<DP>(sM, V0, acc);
    double ww = 0.0;
#pragma unroll
    for (int i = 0; i < DP; ++i) ww = fma(acc[i], acc[i], ww);
    double inc = -0.5 * ww - scal[1];                                      // kalman.py:345-346
    if (FK == SMC_FK_GUIDED)                                               // ssm.py:380-392
        inc = ((-0.5 * uu - scal[first ? 3 : 0]) + inc) - (-0.5 * zz - scal[first ? 4 : 2]);
    SmcLse lacc = smc_lse_empty();
    if (valid) {
        double lw = (resample || first) ? inc : lwo[n] + inc;              // resampling.py:241-244
        if (lw != lw) lw = -INFINITY;                                      // resampling.py:220
        lwn[n] = lw;
        smc_lse_push(lacc, lw);
    }
    // ---- rows of X out, whole lines
    sRow[tid] = valid ? n : -1;
    __syncthreads();
    mv_store_rows<DP>(Xn, d, sRow, sV0);
    f_step_tail(a, isl, b, t, first, resample, lacc, smd, s_last, info);
}

// X_{t-1}[A] for SMC.Xp, (N,d)
__global__ void __launch_bounds__(SMC_BLOCK)
k_f_gather_rows(const double* X, const i64* A, i64 N, int d, double* Xp)
{
    const i64 i = (i64)blockIdx.x * SMC_BLOCK + threadIdx.x;
    if (i < N * d) {
        const i64 nn = i / d;
        Xp[i] = X[A[nn] * d + (i - nn * d)];
    }
}

// ---------------------------------------------------------------------------
// host: derived constants (dense d x d algebra, d <= 32)
// ---------------------------------------------------------------------------
#include <vector>
namespace mvh {
typedef std::vector<double> Mat;     // row-major

inline bool chol(const Mat& A, int n, Mat& L)
{
    L.assign((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0)) return false;
        L[j * n + j] = sqrt(s);
        for (int i = j + 1; i < n; ++i) {
            double v = A[i * n + j];
            for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v / L[j * n + j];
        }
    }
    return true;
}
inline Mat tri_inv(const Mat& L, int n)
{
    Mat R((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c)
        for (int i = c; i < n; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= L[i * n + k] * R[k * n + c];
            R[i * n + c] = s / L[i * n + i];
        }
    return R;
}
inline Mat mul(const Mat& A, int ra, int ca, const Mat& B, int cb)   // (ra,ca)(ca,cb)
{
    Mat C((size_t)ra * cb, 0.0);
    for (int i = 0; i < ra; ++i)
        for (int k = 0; k < ca; ++k) {
            const double v = A[i * ca + k];
            for (int j = 0; j < cb; ++j) C[i * cb + j] += v * B[k * cb + j];
        }
    return C;
}
inline Mat tr(const Mat& A, int r, int c)
{
    Mat T((size_t)r * c);
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) T[j * r + i] = A[i * c + j];
    return T;
}
inline double logdiag(const Mat& L, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += log(L[i * n + i]);
    return s;
}
// Kalman gain and filtered covariance for predictive covariance P
// (kalman.py:215-229): S = G P G' + R ; K = P G' S^-1 ; Pf = P - K G P
inline bool gain(const Mat& P, const Mat& G, const Mat& R, int dx, int dy, Mat& K, Mat& Pf)
{
    Mat GP = mul(G, dy, dx, P, dx);                       // (dy,dx)
    Mat S = mul(GP, dy, dx, tr(G, dy, dx), dy);           // (dy,dy)
    for (int i = 0; i < dy * dy; ++i) S[i] += R[i];
    Mat Ls;
    if (!chol(S, dy, Ls)) return false;
    Mat Li = tri_inv(Ls, dy);
    Mat Sinv = mul(tr(Li, dy, dy), dy, dy, Li, dy);       // S^-1 = L^-T L^-1
    K = mul(tr(GP, dy, dx), dx, dy, Sinv, dy);            // (dx,dy)
    Mat KGP = mul(K, dx, dy, GP, dx);
    Pf = P;
    for (int i = 0; i < dx * dx; ++i) Pf[i] -= KGP[i];
    // symmetrise (round-off)
    for (int i = 0; i < dx; ++i)
        for (int j = 0; j < i; ++j) Pf[i * dx + j] = Pf[j * dx + i] = 0.5 * (Pf[i * dx + j] + Pf[j * dx + i]);
    return true;
}
// store M (r x c) transposed + padded into dst (dp x dp): dst[k*dp+i] = M[i][k]
inline void put_t(double* dst, int dp, const Mat& M, int r, int c, double sign = 1.0)
{
    for (int i = 0; i < r; ++i)
        for (int k = 0; k < c; ++k) dst[k * dp + i] = sign * M[i * c + k];
}
}  // namespace mvh

// Builds the constants block; returns false if a covariance is not positive definite
// (the reference raises ValueError in MvNormal.__init__, distributions.py:935-940).
inline bool mv_build_constants(const smc_model* m, int fk, int dp, i64 T, const double* y,
                               std::vector<double>& out)
{
    using namespace mvh;
    const int dx = m->dx, dy = m->dy;
    Mat F(m->F_host, m->F_host + dx * dx), G(m->G_host, m->G_host + dy * dx);
    Mat QX(m->covX_host, m->covX_host + dx * dx), R(m->covY_host, m->covY_host + dy * dy);
    Mat Q0(m->cov0_host, m->cov0_host + dx * dx), mu0(m->mu0_host, m->mu0_host + dx);
    out.assign(MV_SIZE(dp, T), 0.0);
    double* C = out.data();
    Mat LX, LY, L0;
    if (!chol(QX, dx, LX) || !chol(R, dy, LY) || !chol(Q0, dx, L0)) return false;
    Mat LYi = tri_inv(LY, dy);
    Mat GY = mul(LYi, dy, dy, G, dx);                                     // (dy,dx)
    put_t(C + MV_F * dp * dp, dp, F, dx, dx);
    put_t(C + MV_NGY * dp * dp, dp, GY, dy, dx, -1.0);
    double* scal = C + MV_SCAL(dp);
    scal[0] = logdiag(LX, dx) + dx * SMC_HALFLOG2PI;
    scal[1] = logdiag(LY, dy) + dy * SMC_HALFLOG2PI;
    scal[3] = logdiag(L0, dx) + dx * SMC_HALFLOG2PI;
    for (int i = 0; i < dx; ++i) C[MV_VEC(dp) + i] = mu0[i];
    Mat K, K0;
    if (fk == SMC_FK_GUIDED) {
        Mat P, P0, LP, LP0;
        if (!gain(QX, G, R, dx, dy, K, P) || !gain(Q0, G, R, dx, dy, K0, P0)) return false;
        if (!chol(P, dx, LP) || !chol(P0, dx, LP0)) return false;
        // B = (I - K G) F
        Mat KG = mul(K, dx, dy, G, dx), IKG((size_t)dx * dx, 0.0);
        for (int i = 0; i < dx; ++i)
            for (int j = 0; j < dx; ++j) IKG[i * dx + j] = (i == j ? 1.0 : 0.0) - KG[i * dx + j];
        put_t(C + MV_B * dp * dp, dp, mul(IKG, dx, dx, F, dx), dx, dx);
        put_t(C + MV_LZ * dp * dp, dp, LP, dx, dx);
        put_t(C + MV_LZ0 * dp * dp, dp, LP0, dx, dx);
        put_t(C + MV_XINV * dp * dp, dp, tri_inv(LX, dx), dx, dx);
        put_t(C + MV_X0INV * dp * dp, dp, tri_inv(L0, dx), dx, dx);
        scal[2] = logdiag(LP, dx) + dx * SMC_HALFLOG2PI;
        scal[4] = logdiag(LP0, dx) + dx * SMC_HALFLOG2PI;
        // proposal0 mean: mu0 + K0 (y0 - G mu0)           (kalman.py:353-356)
        for (int i = 0; i < dx; ++i) {
            double v = mu0[i];
            for (int j = 0; j < dy; ++j) {
                double r = y[j];
                for (int k = 0; k < dx; ++k) r -= G[j * dx + k] * mu0[k];
                v += K0[i * dy + j] * r;
            }
            C[MV_VEC(dp) + dp + i] = v;
        }
    } else {
        put_t(C + MV_LZ * dp * dp, dp, LX, dx, dx);
        put_t(C + MV_LZ0 * dp * dp, dp, L0, dx, dx);
    }
    for (i64 t = 0; t < T; ++t) {
        double* yw = C + MV_STEP(dp) + (size_t)t * 2 * dp;
        const double* yt = y + t * dy;
        for (int i = 0; i < dy; ++i) {
            double v = 0.0;
            for (int j = 0; j <= i; ++j) v += LYi[i * dy + j] * yt[j];
            yw[i] = v;
        }
        if (fk == SMC_FK_GUIDED)
            for (int i = 0; i < dx; ++i) {
                double v = 0.0;
                for (int j = 0; j < dy; ++j) v += K[i * dy + j] * yt[j];
                yw[dp + i] = v;
            }
    }
    return true;
}
