/*
 * vgpa.h -- C ABI of libvgpa_hip.so: the gfx950 (MI355X) kernels behind the
 * variational free-energy / gradient evaluation of VGPA.
 *
 * Conventions
 *   - Every array is float64 (or int32 where stated), C order, contiguous, in
 *     DEVICE memory owned by the caller.  The leading axis P is the batch of
 *     independent problems (grid points, restarts); P = 1 for one problem.
 *   - D is the state dimension.  For D == 1 the trailing D / DxD axes are
 *     dropped: A, b, m, S, ... are (P, N).
 *   - A "stride" argument is the element distance between consecutive
 *     problems of a parameter array; 0 means the same values are shared by all
 *     P problems.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls only enqueue
 *     work; they never synchronise, allocate or free device memory.
 *   - Return value: 0 on success, a negative VGPA_E* code otherwise; the text
 *     of the last error of the calling thread is vgpa_last_error_string().
 *   - Integration methods follow the reference's four ODE schemes.
 *
 * Citations are to the reference implementation (ruiixu23/VGPA), paths relative
 * to its root; core/likelihood.py is cited with virtual line numbers (@N) after
 * converting its bare-CR line ends.
 */
#ifndef VGPA_H
#define VGPA_H

#ifdef __cplusplus
extern "C" {
#endif

#define VGPA_ABI_VERSION 1

/* error codes */
#define VGPA_OK            0
#define VGPA_EINVAL      (-1)   /* bad argument (null pointer, size, enum)        */
#define VGPA_EUNSUPPORTED (-2)  /* D or N outside what the kernels are built for  */
#define VGPA_ELAUNCH     (-3)   /* HIP reported a launch error                    */

/* ODE integration method: core/fwd_ode.py:57-82, core/bwd_ode.py:58-83 */
#define VGPA_EULER 0
#define VGPA_HEUN  1
#define VGPA_RK2   2
#define VGPA_RK4   3

/* dynamical system of an energy kernel */
#define VGPA_SYS_OU      0   /* dynamics/sys_ornstein_uhlenbeck.py:83 */
#define VGPA_SYS_DW      1   /* dynamics/sys_double_well.py:91        */
#define VGPA_SYS_L63     2   /* dynamics/sys_lorenz_63.py:125         */
#define VGPA_SYS_L63_UT  3   /* dynamics/sys_lorenz_63_ut.py:132      */
#define VGPA_SYS_L96_UT  4   /* dynamics/sys_lorenz_96_ut.py:131      */

/* per-step status bits written by the unscented kernels */
#define VGPA_ST_CHOL_FALLBACK 1  /* chol((D+k)S) failed; diagonal fallback of
                                    auxiliary/numerical.py:201-202 was used     */
#define VGPA_ST_CHOL_FAILED   2  /* the fallback failed too (LinAlgError case)  */
#define VGPA_ST_UNSUPPORTED   4  /* transient mark of a step that waits for the
                                    fallback pass; never left set when a status
                                    array was given to the call                 */

/* gradient-assembly flags */
#define VGPA_GRAD_OUTER 1        /* use outer products lam m', dEsde_db m' instead
                                    of the reference's scalar inner products
                                    (core/gradL_Ab.py:167,170).  Off by default. */
#define VGPA_GRAD_SIGINV_DIAG 2  /* SigInv is diagonal: only its diagonal is read */

/* scan flags */
#define VGPA_SCAN_SYMMETRIC 1    /* the caller guarantees bitwise symmetric matrix
                                    inputs (forward: S0 and Sigma; backward: every
                                    dEsde_dS[t] and every jump).  The scan then forms
                                    one matrix product per stage instead of two; the
                                    result is bit-for-bit that of the general path,
                                    except at D = 40, where that product runs on the FP64
                                    matrix cores in another summation order: equal to
                                    rounding, and still exactly symmetric. */

#define VGPA_SCAN_STATE_SYMMETRIC 2  /* vgpa_bwd_scan_grad only: every S[t] is bitwise symmetric too
                                       (it is when it came from vgpa_fwd_scan with
                                       VGPA_SCAN_SYMMETRIC)                                     */

#define VGPA_SCAN_PSI_ROW0 4     /* vgpa_bwd_scan_grad only, and only where
                                    vgpa_bwd_scan_grad_is_fused() holds: Psi_out is
                                    (P,1,D,D) and receives Psi(0) alone -- the row the
                                    gradient w.r.t. S0 needs (core/kl0.py:63-66); the
                                    other rows are consumed inside the kernel and not
                                    written (19 % of its memory traffic).  For callers
                                    that only want F and the gradient, e.g. an optimiser. */

#define VGPA_SCAN_PACKED 8       /* the symmetric matrices of every time row -- S_out of
                                    vgpa_fwd_scan; S, dEsde_dS and Psi_out of vgpa_bwd_scan_grad
                                    -- are in the packed layout described at vgpa_sym_pack
                                    (1152 doubles a row instead of 1600).  Euler method, D = 40,
                                    with VGPA_SCAN_SYMMETRIC; vgpa_bwd_scan_grad only where
                                    vgpa_bwd_scan_grad_is_fused() holds.  A, dL/dA, the vectors,
                                    S0, Sigma, the jumps and the terminal state stay dense.
                                    Anything else: VGPA_EUNSUPPORTED.                          */

#define VGPA_SCAN_PACKED_DEDS 16 /* vgpa_bwd_scan_grad only, same conditions: dEsde_dS alone is packed
                                    (P,N,1152), as vgpa_esde_l96_ut_packed writes it with layout
                                    VGPA_PACKED_DEDS; S and Psi_out stay dense.  dEsde_dS is an
                                    intermediate of the evaluation, not one of the arrays the
                                    reference publishes: this is the layout to use between the two
                                    kernels when every published array is wanted dense.          */

/* layout bits of vgpa_esde_l96_ut_packed */
#define VGPA_PACKED_S    1       /* S is (P,N,1152)        */
#define VGPA_PACKED_DEDS 2       /* dEsde_dS is (P,N,1152) */

/* largest state dimension the n-D kernels accept */
#define VGPA_MAX_D 60

int vgpa_version(void);
const char* vgpa_last_error_string(void);

/*
 * Forward moment scan: replaces core/fwd_ode.py:48 `solver(A, b, m0, S0, Sigma,
 * dt, method)` and its eight integrators (euler_nD :87, euler_1D :135, heun_nD
 * :181, heun_1D :247, rk2_nD :311, rk2_1D :367, rk4_nD :426, rk4_1D :493).
 *   A (P,N,D,D)  b (P,N,D)  m0 (P,D)  S0 (P,D,D)  Sigma (D,D) [sigma_stride]
 *   flags: VGPA_SCAN_SYMMETRIC or 0; with VGPA_SCAN_PACKED as well (Euler, D = 40) S_out is (P,N,1152).
 *   -> m_out (P,N,D), S_out (P,N,D,D);  row 0 holds (m0, S0).
 */
int vgpa_fwd_scan(int method, int P, int N, int D,
                  const double* A, const double* b,
                  const double* m0, const double* S0,
                  const double* Sigma, int sigma_stride, double dt, int flags,
                  double* m_out, double* S_out, void* stream);

/*
 * Backward Lagrange-multiplier scan: replaces core/bwd_ode.py:50 `solver(A,
 * dEsde_dm, dEsde_dS, dEobs_dm, dEobs_dS, dt, method)` (euler_nD :86, euler_1D
 * :132, heun :177/:241, rk2 :305/:367, rk4 :427/:497).
 *   dEobs_dm (P,N,D) dense.  The covariance jumps are either dense
 *   `dEobs_dS` (P,N,D,D), or -- when dEobs_dS is NULL -- `jumpS` (D,D)
 *   [jumpS_stride] added at every row t with obs_flag[t] != 0 (obs_flag is (N)
 *   int32, shared by all problems).  Both NULL means no covariance jumps.
 *   flags: VGPA_SCAN_SYMMETRIC or 0.
 *   lam_last (P,D), Psi_last (P,D,D): state of row N-1; NULL means zero, which is
 *   what the reference starts from.  A time window of a longer scan passes the row it
 *   received from the window after it.
 *   -> lam_out (P,N,D), Psi_out (P,N,D,D); row N-1 holds the terminal state.
 */
int vgpa_bwd_scan(int method, int P, int N, int D,
                  const double* A, const double* dEsde_dm, const double* dEsde_dS,
                  const double* dEobs_dm, const double* dEobs_dS,
                  const int* obs_flag, const double* jumpS, int jumpS_stride,
                  double dt, int flags, const double* lam_last, const double* Psi_last,
                  double* lam_out, double* Psi_out, void* stream);

/*
 * Backward scan and gradient assembly in one call: replaces core/bwd_ode.py:50 `solver(...)`
 * followed by core/gradL_Ab.py:112 `gradL_Ab_nD` (the loop :156-173 and the scaling :176-177),
 * i.e. vgpa_bwd_scan followed by vgpa_grad_assemble, with the arguments of both (scan_flags:
 * VGPA_SCAN_*; grad_flags: VGPA_GRAD_*).  For the Euler, Heun and RK2 methods at D = 40 with
 * VGPA_SCAN_SYMMETRIC | VGPA_SCAN_STATE_SYMMETRIC, flagged jumps (dEobs_dS == NULL), Edf ==
 * NULL with edf_system == VGPA_SYS_L96_UT and a diagonal SigInv without VGPA_GRAD_OUTER, one
 * kernel forms the gradient of a row while the scan has that row's A(t), Psi(t), lam(t) in
 * LDS, so that neither is read from memory a second time (used from 161 problems on, where the
 * call is bound by memory traffic); every other configuration runs the two kernels one after the
 * other.
 *   -> lam_out (P,N,D), Psi_out (P,N,D,D) [(P,1,D,D) with VGPA_SCAN_PSI_ROW0], gLA (P,N,D,D), gLb (P,N,D).
 *   With VGPA_SCAN_PACKED: S, dEsde_dS (P,N,1152) in, Psi_out (P,N,1152) [(P,1,1152)] out.
 */
int vgpa_bwd_scan_grad(int method, int P, int N, int D,
                       const double* A, const double* b, const double* m, const double* S,
                       const double* dEsde_dm, const double* dEsde_dS,
                       const double* dEobs_dm, const double* dEobs_dS,
                       const int* obs_flag, const double* jumpS, int jumpS_stride,
                       const double* Ef, const double* Edf,
                       int edf_system, const double* theta, int theta_stride,
                       const double* SigInv, int sig_stride,
                       double dt, int scan_flags, int grad_flags,
                       const double* lam_last, const double* Psi_last,
                       double* lam_out, double* Psi_out, double* gLA, double* gLb, void* stream);

/*
 * Whether vgpa_bwd_scan_grad runs as one kernel for these arguments (1) or as vgpa_bwd_scan followed
 * by vgpa_grad_assemble (0); dense_jumps: dEobs_dS != NULL, explicit_edf: Edf != NULL.  No
 * counterpart in the reference (it has one code path); a caller needs it to know whether
 * VGPA_SCAN_PSI_ROW0 is accepted.  Touches no device.
 */
int vgpa_bwd_scan_grad_is_fused(int method, int P, int D, int scan_flags, int grad_flags,
                                int dense_jumps, int explicit_edf, int edf_system);

/*
 * Per-time-step expected SDE energy and its gradients: replaces the loop body of
 * `energy_mode(A, b, m, S, sDyn)` -- dynamics/sys_ornstein_uhlenbeck.py:133-156,
 * dynamics/sys_double_well.py:139-184, dynamics/sys_lorenz_63.py:202-233 (with
 * Energy_dm_dS :247 and Efg_drift_theta :491), dynamics/sys_lorenz_63_ut.py:218-255,
 * dynamics/sys_lorenz_96_ut.py:217-247 -- including the unscented transform
 * (auxiliary/numerical.py:153-228) and core/gradEsde_mS.py:8-68 in closed form.
 * The time integrals (mytrapz) are done by vgpa_reduce_energy on the per-step
 * outputs.
 *   sig_inv_diag (D) [sig_stride]  : diagonal of inv(Sigma)  (1-D: 1/Sigma)
 *   theta (n_theta) [theta_stride] : drift parameters (1 value; 3 for Lorenz-63)
 *   -> Esde_t (P,N)    0.5 * <(f-g)' SigInv (f-g)> at each step
 *      Ef (P,N,D), Edf (P,N,D,D) [may be NULL for D > 1], dEsde_dm (P,N,D),
 *      dEsde_dS (P,N,D,D)
 *      dth_t (P,N,D or n_theta) [NULL ok] integrand of dEsde/dtheta before the
 *            diag(SigInv) scaling;  dSig_t (P,N,D) [NULL ok] integrand of
 *            dEsde/dSigma (the per-component <(f-g)^2>); unused for D == 1
 *      status (P,N) int32 [NULL ok] VGPA_ST_* bits (unscented systems only)
 */
int vgpa_esde(int system, int P, int N, int D,
              const double* A, const double* b, const double* m, const double* S,
              const double* sig_inv_diag, int sig_stride,
              const double* theta, int theta_stride,
              double* Esde_t, double* Ef, double* Edf,
              double* dEsde_dm, double* dEsde_dS,
              double* dth_t, double* dSig_t, int* status, void* stream);

/* Same as vgpa_esde with the system fixed (one symbol per kernel family). */
#define VGPA_DECL_ESDE(name)                                                      \
    int name(int P, int N, int D, const double* A, const double* b,               \
             const double* m, const double* S, const double* sig_inv_diag,        \
             int sig_stride, const double* theta, int theta_stride,               \
             double* Esde_t, double* Ef, double* Edf, double* dEsde_dm,           \
             double* dEsde_dS, double* dth_t, double* dSig_t, int* status,        \
             void* stream)
VGPA_DECL_ESDE(vgpa_esde_ou);
VGPA_DECL_ESDE(vgpa_esde_dw);
VGPA_DECL_ESDE(vgpa_esde_l63);
VGPA_DECL_ESDE(vgpa_esde_l63_ut);
VGPA_DECL_ESDE(vgpa_esde_l96_ut);

/*
 * Weighted time reduction out[p,c] = scale * sum_t w[t] * x[p,t,c]: with the
 * weights of the segmented trapezoid rule this replaces `mytrapz(fxt, dt, idx)`,
 * auxiliary/numerical.py:231-284 (w has zeros from the last observation on).
 * Time-window shards pass their slice of w, so partial sums add exactly.
 *   x (P,N,C)  w (N)  -> out (P,C).  Deterministic summation order.
 */
int vgpa_reduce_energy(int P, int N, int C, const double* x, const double* w,
                       double scale, double* out, void* stream);

/*
 * Observation energy and jump terms: replaces core/likelihood.py@13 `gauss(m, S,
 * sde_struct)` (gauss_1D @52, gauss_nD @108).
 *   obs_idx (M) int32 time indices; obsY (M,d) [obsY_stride]; H (D,d);
 *   Ri (d,d) and Qi (d,d) as returned by chol_inv(R) (auxiliary/numerical.py:10-48);
 *   logconst = M * (d*log(2 pi) + log det R).  For D == 1: H, Ri are single
 *   values, Qi is unused.  work is caller-provided scratch of P*M doubles (D > 1).
 *   -> Eobs (P); dEobs_dm (P,N,D) (zeroed, then rows obs_idx filled);
 *      jumpS (D,D) = 0.5 H' Ri H (the value of dEobs_dS at every observed row).
 */
int vgpa_obs_terms(int P, int N, int D, int M, int d,
                   const double* m, const double* S,
                   const int* obs_idx, const double* obsY, int obsY_stride,
                   const double* H, const double* Ri, const double* Qi,
                   double logconst, double* work,
                   double* Eobs, double* dEobs_dm, double* jumpS, void* stream);

/*
 * Packed symmetric matrices (D = 40 only).  A symmetric 40 x 40 matrix is kept as the rows of its
 * lower BLOCK triangle with 16 x 16 blocks: row r holds its first 16 (r / 16 + 1) entries, i.e.
 * everything left of and inside its diagonal block (the diagonal blocks whole, both triangles):
 *     rows  0 .. 15   16 doubles each, at  16 r
 *     rows 16 .. 31   32 doubles each, at 256 + 32 (r - 16)
 *     rows 32 .. 39   48 doubles each, at 768 + 48 (r - 32)   (columns 40 .. 47: padding, never
 *                                                              read or written)
 * 1152 doubles = vgpa_sym_packed_size(40) instead of 1600, every row a whole number of 128-byte
 * lines.  The Euler kernels of the Lorenz-96 path read and write S, dEsde_dS and Psi in this form
 * when asked to (VGPA_SCAN_PACKED, vgpa_esde_l96_ut_packed, vgpa_obs_terms_packed): 16 % less
 * memory traffic per evaluation.  It has no counterpart in the reference; it is an option for
 * callers that consume F and the gradient (an optimiser) rather than the matrices themselves.
 *   vgpa_sym_pack    dense (count,40,40) -> packed (count,1152); reads the kept entries only
 *   vgpa_sym_unpack  packed (count,1152) -> dense (count,40,40), mirrored above the block diagonal
 *   vgpa_sym_packed_size(D)  1152 for D == 40, -1 otherwise
 */
int vgpa_sym_pack(long long count, int D, const double* dense, double* packed, void* stream);
int vgpa_sym_unpack(long long count, int D, const double* packed, double* dense, void* stream);
int vgpa_sym_packed_size(int D);

/*
 * vgpa_esde_l96_ut (dynamics/sys_lorenz_96_ut.py:131) at D = 40 with packed symmetric matrices:
 * `layout` is VGPA_PACKED_S (S is (P,N,1152)), VGPA_PACKED_DEDS (dEsde_dS is (P,N,1152)), both or 0;
 * no Edf; the other arguments and results as vgpa_esde_l96_ut.  Results are those of the dense
 * call bit for bit when S is bitwise symmetric.
 */
int vgpa_esde_l96_ut_packed(int P, int N, int layout,
                            const double* A, const double* b, const double* m, const double* S,
                            const double* sig_inv_diag, int sig_stride,
                            const double* theta, int theta_stride,
                            double* Esde_t, double* Ef, double* dEsde_dm, double* dEsde_dS,
                            double* dth_t, double* dSig_t, int* status, void* stream);

/*
 * vgpa_obs_terms (core/likelihood.py@108-179) at D = 40 with S_packed (P,N,1152); it reads the
 * diagonal of S only.  Other arguments and results as vgpa_obs_terms.
 */
int vgpa_obs_terms_packed(int P, int N, int M, int d,
                          const double* m, const double* S_packed,
                          const int* obs_idx, const double* obsY, int obsY_stride,
                          const double* H, const double* Ri, const double* Qi,
                          double logconst, double* work,
                          double* Eobs, double* dEobs_dm, double* jumpS, void* stream);

/*
 * vgpa_obs_terms for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in
 * the reference.  Same arguments and results as vgpa_obs_terms without `work`: the per-
 * observation energies go to a fixed array inside the library instead (VGPA_OBS_STATIC_MAX
 * doubles = 1 MiB, held on every GPU the library runs on until the process ends).
 *   - P * M > VGPA_OBS_STATIC_MAX is refused with VGPA_EINVAL (D > 1; D == 1 needs no scratch).
 *   - The call allocates and frees nothing.  Its first call on a device looks up the device
 *     address of the array (hipGetSymbolAddress); make that first call outside a capture.
 *   - All calls on one GPU share the one array: calls that overlap on the same GPU (two
 *     streams, two threads) race.  Nothing a previous call left in it is read.
 *   - Every GPU has its own copy of the array, found through the CURRENT device: the device
 *     that owns `stream` (and the pointers) must be the current one.  A stream of another
 *     device is refused with VGPA_EINVAL.
 */
#define VGPA_OBS_STATIC_MAX (1 << 17)
int vgpa_obs_terms_static(int P, int N, int D, int M, int d,
                          const double* m, const double* S,
                          const int* obs_idx, const double* obsY, int obsY_stride,
                          const double* H, const double* Ri, const double* Qi,
                          double logconst,
                          double* Eobs, double* dEobs_dm, double* jumpS, void* stream);

/*
 * Initial-state KL energy: replaces core/kl0.py:9 `gauss(m0, S0, lam, Psi,
 * sde_struct)` (energy only; gauss_1D :49, gauss_nD :72).  `m0_is_row` selects
 * the outer-product form the reference produces for a (1,D) mean; 0 gives the
 * scalar |z0|^2 broadcast it produces for a (D,) mean (core/kl0.py:88-90).
 *   m0 (P,D) S0 (P,D,D) mu0 (D) [mu0_stride] tau0 (D,D) [tau0_stride] -> KL0 (P)
 */
int vgpa_kl0(int P, int D, const double* m0, const double* S0,
             const double* mu0, int mu0_stride, const double* tau0, int tau0_stride,
             int m0_is_row, double* KL0, void* stream);

/*
 * Gradient of the Lagrangian w.r.t. A(t), b(t): replaces core/gradL_Ab.py:50
 * `gradL_Ab_1D` and :112 `gradL_Ab_nD` (the loop :156-173 and the dt scaling
 * :176-177).  SigInv (D,D) [sig_stride] is inv(Sigma) (1-D: 1/Sigma).
 *   Edf may be NULL for D > 1: <df/dx> is then derived from the mean for
 *   edf_system = VGPA_SYS_L96_UT (dynamics/sys_lorenz_96_ut.py:293-330) or a Lorenz-63
 *   system (dynamics/sys_lorenz_63.py:227, needs theta (3) [theta_stride]), which saves
 *   writing and re-reading a dense (P,N,D,D) array.
 *   -> gLA (P,N,D,D), gLb (P,N,D), already multiplied by dt.
 */
int vgpa_grad_assemble(int P, int N, int D,
                       const double* A, const double* b, const double* m, const double* S,
                       const double* lam, const double* Psi,
                       const double* Ef, const double* Edf,
                       int edf_system, const double* theta, int theta_stride,
                       const double* SigInv, int sig_stride, double dt, int flags,
                       double* gLA, double* gLb, void* stream);

/*
 * Sigma points of the unscented transform: the first half of `ut_approx(f, xbar,
 * Pxx, *args)`, auxiliary/numerical.py:186-208.  k = 1.05 D; rows of
 * chol((D+k) Pxx)' are added to / subtracted from the mean; when that factorisation
 * fails the rows of chol(Pxx .* I)' are used instead (:201-202).
 *   xbar (P,D)  Pxx (P,D,D)  -> chi (P,2D+1,D); status (P) int32 [NULL ok] VGPA_ST_* bits.
 */
int vgpa_ut_sigma_points(int P, int D, const double* xbar, const double* Pxx,
                         double* chi, int* status, void* stream);

/*
 * Weighted moments of the transformed sigma points: the second half of `ut_approx`,
 * auxiliary/numerical.py:210-226.  K = 2D+1 fixes the weights (:211-212).
 *   Y (P,K,C) -> ybar (P,C) = w Y;  Pyy (P,C,C) [NULL ok] = Y' Q Y with
 *   Q = (I - 1 w') diag(w) (I - 1 w')' exactly as the reference builds it.
 */
int vgpa_ut_moments(int P, int K, int C, const double* Y, double* ybar, double* Pyy,
                    void* stream);

/*
 * Per-sample gradient terms: replaces core/gradEsde_mS.py:8 `gradEsde_mS(x, F, mt,
 * St, At, bt, diagSigI, sDyn)`; Fx = F(x, sDyn) is evaluated by the caller.
 *   x (K,D) Fx (K,D) mt (D) St (D,D) At (D,D) bt (D) dSigI (D)
 *   -> out (K, D + D*D): row k = [0.5 v_k inv(St) x_k, 0.5 v_k inv(St) z_k z_k' inv(St)].
 */
int vgpa_grad_esde_ms(int K, int D, const double* x, const double* Fx, const double* mt,
                      const double* St, const double* At, const double* bt,
                      const double* dSigI, double* out, void* stream);

/*
 * Uncentred moments of a univariate Gaussian and their derivatives: replaces
 * auxiliary/momGauss.py:5 `momGauss(m, S, order, dmS)`.  order in [0, 8] (anything
 * else is VGPA_EINVAL, the reference's ValueError :104); deriv 0 = moment, 1 = 'Dm',
 * 2 = 'DS'.   m (n) S (n) -> out (n).
 */
int vgpa_mom_gauss(long long n, const double* m, const double* S, int order, int deriv,
                   double* out, void* stream);

/*
 * Vector algebra of the batched scaled-conjugate-gradient driver: the passes that one
 * iteration of auxiliary/optimize.py:132-238 `optim_SCG` makes over the parameter-sized
 * arrays between its two gradient evaluations, for P independent problems at once.
 *
 * A parameter array holds n = nA + nb doubles per problem as two contiguous blocks,
 * [P x nA | P x nb]: its A block and b block are the (P,N,D,D) and (P,N,D) arrays of the
 * entry points above (nA = N*D*D, nb = N*D).  `mode` (P ints) selects per problem what
 * a pass does; 0 always means "skip this problem" (nothing is read or written).
 * Reducing passes take nch in [1,1024] (chunks per row), work (P*2*nch*4 doubles of
 * scratch) and write out (P,4); unused columns and skipped problems are 0.  The sums
 * are formed in a fixed order: a call repeated on the same data returns the same bits.
 *
 * vgpa_scg_dots        out = {d.g, d.d, g.g}: slope mu and kappa of :150-156 (a direction
 *                      with d.g >= 0 is replaced by -g: then mu = -g.g, kappa = g.g);
 *                      mode may be NULL (all problems).
 * vgpa_scg_step        y = x + coef[p]*d (mode 1), or with d replaced by -g, which is also
 *                      stored into d (mode 2, :151-153).  The probe point of :158 and the
 *                      trial point of :170.
 * vgpa_scg_curvature   out = {d.(g_probe - g)}: theta of :159 times sigma.
 * vgpa_scg_trial_stats out = {max|alpha[p] d| (:186), g_try.g_try (:196),
 *                      g_try.(g - g_try) (:221 times mu, after the move of :193-195),
 *                      sum|g_try|}.
 * vgpa_scg_update      mode bit 0: x <- x_try, g <- g_try (:188-195); then d <- -g when
 *                      bit 1 is set (restart, :216-218), else gamma[p]*d - g (:222);
 *                      out = {d.g, d.d, g.g} of the new pair.  Bit 2 (4): the same new
 *                      direction from g as it stands, without the copy -- for a caller that
 *                      exchanges its (x, g) and (x_try, g_try) arrays instead of copying
 *                      accepted points; bit 3 (8): the copy alone (d untouched, out = 0), for
 *                      the rows such an exchange must not move.
 */
int vgpa_scg_dots(int P, long long nA, long long nb, const double* d, const double* g,
                  const int* mode, int nch, double* work, double* out, void* stream);
int vgpa_scg_step(int P, long long nA, long long nb, const double* x, double* d, const double* g,
                  const double* coef, const int* mode, double* y, void* stream);
int vgpa_scg_curvature(int P, long long nA, long long nb, const double* d, const double* g_probe,
                       const double* g, const int* mode, int nch, double* work, double* out,
                       void* stream);
int vgpa_scg_trial_stats(int P, long long nA, long long nb, const double* d, const double* g,
                         const double* g_try, const double* alpha, const int* mode, int nch,
                         double* work, double* out, void* stream);
int vgpa_scg_update(int P, long long nA, long long nb, double* x, const double* x_try, double* g,
                    const double* g_try, double* d, const double* gamma, const int* mode, int nch,
                    double* work, double* out, void* stream);

/*
 * Sample paths of the stochastic systems, P independent paths at once: replaces the
 * recursions of `system_path` in dynamics/sys_ornstein_uhlenbeck.py:15-61,
 * dynamics/sys_double_well.py:14-75, dynamics/sys_lorenz_63.py:42-106 (and _ut.py:49-113),
 * dynamics/sys_lorenz_96_ut.py:43-111.  From x0 (P,D): `burn` deterministic Euler steps of
 * size dtau (the burn-in loops :83-84 / :38-40; 0 for none), then X[0] = x + add0 (add0 (P,D)
 * or NULL) and X[t] = X[t-1] + f(X[t-1]) dt + kdiag .* noise[t] for t = 1..N-1.  The caller
 * supplies the random numbers noise (P,N,D) (row 0 is not used, as in the reference) and
 * kdiag = sqrt(diag(sig0) dt) as (D) (k_stride 0) or (P,D) (k_stride D); theta as for
 * vgpa_esde.  Sums and products round separately in the order of the reference's NumPy
 * expressions: with the same random numbers the path has the same bits.   -> X (P,N,D).
 */
int vgpa_sde_path(int system, int P, int N, int D, const double* x0, const double* add0,
                  const double* theta, int theta_stride, const double* kdiag, int k_stride,
                  int burn, double dtau, double dt, const double* noise, double* X, void* stream);

/*
 * Starting values of the variational parameters from interpolated means: replaces the
 * n-dimensional branch of auxiliary/initialize_Ab0.py:66-93.  mt0 (P,N,D) are the spline
 * means (the splines themselves, :74-77, stay on the host: SciPy FITPACK), sig_diag and
 * s0_diag the diagonals of Sigma and S0 as (D) (stride 0) or (P,D) (stride D).
 *   -> A0 (P,N,D,D) = 0.5 diag(Sigma_ii / S0_ii),
 *      b0 (P,N,D)   = (mt0[k+1] - mt0[k]) / dt + A_ii mt0[k],  last row A_ii mt0[N-1].
 */
int vgpa_init_ab0(int P, int N, int D, const double* mt0, const double* sig_diag, int sig_stride,
                  const double* s0_diag, int s0_stride, double dt, double* A0, double* b0,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VGPA_H */
