/*
 * radex_oracle.h -- CPU restatement of the reference's per-walker RADEX LVG
 * likelihood path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product (radex_emcee_amd/csrc, libradex_emcee_amd.so)
 * never includes, links or falls back to anything in oracle/.
 *
 * What it restates (reference = /root/reference, yangcht/radex_emcee):
 *   - emcee/pyradex/radex/radex.so (Fortran RADEX, binary only): readdata_,
 *     backrad_, escprob_, matrix_, lubksb_ -> sgeir_/sgefa_/sgesl_; arithmetic
 *     as recovered from the binary in SURVEY.md Appendix A (addresses cited
 *     per function in radex_oracle.c);
 *   - emcee/pyradex/core.py:856-925 (run_radex iteration driver),
 *     core.py:986-1003 + base_class.py:275-277 (emergent intensity);
 *   - emcee/emcee_radex.py:120-181 and emcee/emcee_radex_2comp.py:122-244
 *     (model_lvg / lnlike / lnprior / lnprob, 1- and 2-component).
 *
 * Pinning: matrix_/escprob_/backrad_/lubksb_ are pinned against golden vectors
 * produced by executing the reference's own radex.so machine code in this
 * container (oracle/macho_ref.py -> tests/golden/ref_*.json).  readdata_'s
 * file parsing + rate interpolation (needs libgfortran I/O) and the driver
 * functions have no runnable reference here; they are pinned only by the
 * cited source lines / binary addresses ("parity unpinned" for those rows).
 */
#ifndef RADEX_ORACLE_H
#define RADEX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RXO_MAXPART 9       /* density(9) in COMMON /cphys/  */

typedef struct rxo_mol {
    int nlev, nline, npart;
    double amass;
    double *eterm, *gstat;              /* [nlev]  cm^-1, weight            */
    int *iupp, *ilow;                   /* [nline] 1-based level indices    */
    double *aeinst, *spfreq, *eup, *xnu;/* [nline] xnu = eterm(u)-eterm(l)  */
    int part_id[RXO_MAXPART];           /* 1=H2 2=pH2 3=oH2 4=e 5=H 6=He 7=H+ */
    int ncoll[RXO_MAXPART], ntemp[RXO_MAXPART];
    double *temp[RXO_MAXPART];          /* [ntemp]                          */
    int *lcu[RXO_MAXPART], *lcl[RXO_MAXPART]; /* [ncoll] 1-based            */
    double *coll[RXO_MAXPART];          /* [ncoll][ntemp] cm^3 s^-1         */
} rxo_mol;

/* One solver instance = the COMMON-block state of one reference worker. */
typedef struct rxo_state {
    const rxo_mol *mol;
    int method;                 /* 1 sphere, 2 lvg, 3 slab                  */
    double density[RXO_MAXPART];/* index = partner id - 1                   */
    double tkin, tbg, cdmol, deltav /* cm/s */, totdens;
    double *crate;              /* [nlev*nlev], crate[i*nlev+j] = rate i->j */
    double *ctot;               /* [nlev]                                   */
    double *xpop, *xpopold;     /* [nlev]                                   */
    double *tex, *taul, *backi, *totalb, *trj; /* [nline]                   */
    double *yrate, *rhs, *lu;   /* scratch                                  */
    int *ipvt;
    int lu_info;                /* last sgefa info (0 = ok)                 */
    /* --- iterative-refinement variant of step 4 (rxo_set_refine; OFF by default, in which case
     * nothing below is touched and rxo_matrix is the reference's arithmetic bit for bit) ----------- */
    double *rf_minv[2];         /* kept explicit inverse per parity of niter, column-major [n*n]     */
    double *rf_x[2];            /* the solution that went with it last (x0 of the next refinement)   */
    int rf_have[2];
    int rf_skip, rf_frun;       /* back-off: iterations left without attempts; failed attempts in a row          */
    double *rf_r, *rf_d, *rf_a; /* scratch                                                            */
    long rf_full, rf_refined, rf_steps, rf_failed, rf_kept;   /* counters: solves by path, steps, inverses */
} rxo_state;

/* The product's device kernels replace most of the pivoted solves of step 4 by
 * iterative refinement against a kept inverse; this is the same scheme on the
 * CPU so that its effect on status / iteration count / flux can be measured against the reference's
 * arithmetic on hundreds of thousands of walkers (scripts/refine_gate.py, tests/test_oracle_refine.py).
 * It is a property of the PROCESS (all states created afterwards); first_iter = 0 switches it off.
 *   first_iter : first niter whose solve may be a refinement (the two iterations before it keep their
 *                inverse); tol: the correction, relative to the largest component of the start vector, below
 *                which an iterate is accepted; max_steps: give up after so many steps (-> pivoted solve, which
 *                refreshes the kept inverse of that parity);
 *   lag        : 2 = one kept inverse / start vector per parity of niter, 1 = a single one;
 *   crit       : 0 = accept as soon as max|d| <= tol max|x| (plain); 1 = the device kernels' scheme (rf_solve): the
 *                kept inverse in single precision, absolute thresholds tol / 8 and d1max / 8 (populations sum to 1),
 *                the last observed contraction must be <= 1/2 from the fifth correction on, an attempt that cannot
 *                get there in the steps that are left is given up at once;
 *   d1max      : crit 1: give up when the FIRST correction is above d1max / 8 (0: no such rule);
 *   loose      : crit 1: also accept on two corrections in a row below loose / 8 -- the floor of what double precision
 *                residuals resolve for an ill-conditioned system (0: no such rule);
 *   backoff    : from the second failed attempt in a row on, pause the attempts for 2, 4, ... 64 iterations.      */
void rxo_set_refine(int first_iter, double tol, int max_steps, int lag, int crit, double d1max, double loose, int backoff);
/* crit 2 of rxo_set_refine: thresholds per component, thr_i = min(tol / 8, rel * max(|x_kept_i|, floor)), the same for
 * `loose` with loose_rel; the rate rule works on max_i log2(|d_i| / thr_i). */
void rxo_set_refine_componentwise(double rel, double floor, double loose_rel);
void rxo_refine_counters(long *full, long *refined, long *steps, long *failed, long *kept, int reset);

rxo_mol *rxo_mol_load(const char *path, char *err, size_t errlen);
void rxo_mol_free(rxo_mol *m);

rxo_state *rxo_state_new(const rxo_mol *m, int method, double deltav_kms);
void rxo_state_free(rxo_state *s);

/* readdata_ second half: rate interpolation + detailed balance (A.2).      */
int rxo_rates(rxo_state *s);
/* backrad_ (A.1), tbg > 0 branch.                                           */
void rxo_backrad(rxo_state *s, double tbg);
/* escprob_ (A.3).                                                           */
double rxo_escprob(double tau, int method);
/* matrix_ (A.4) incl. lubksb_/sgeir_ (A.5); conv is in/out.                 */
void rxo_matrix(rxo_state *s, int niter, int *conv);
/* lubksb_ alone on a column-major (n x n, lda = n) matrix: last row <- 1,
 * rhs <- e_last; returns sgefa info.                                        */
int rxo_lubksb(double *a_colmajor, int n, double *x_out, int *ipvt);
/* core.py:856-925; returns the reference's _iter_counter.                   */
int rxo_run(rxo_state *s, int reuse_last, int miniter, int maxiter, int *converged);
/* core.py:986-1003 - background: out[nline] erg s^-1 cm^-2 Hz^-1 sr^-1      */
void rxo_source_line_surfbrightness(const rxo_state *s, double *out);

/* --- driver restatement (emcee_radex.py / emcee_radex_2comp.py) ---------- */
typedef struct rxo_source {
    double tbg;
    int nJ;
    const int32_t *Jup;
    const double *flux, *eflux;
    const double *bounds;       /* [ndim][2]                                 */
    int ncomp;                  /* 1 or 2                                    */
    double T_d;                 /* NaN = None                                */
} rxo_source;

/* status codes shared with the product ABI */
enum { RXO_OK = 0, RXO_MAXITER = 1, RXO_INVALID = 2, RXO_PRIOR = 3 };

/* model_lvg: params = log10 (n, T, N, size) x ncomp -> flux[nJ] Jy km/s.
 * returns 0, or 1 if the reference would raise ValueError.                  */
int rxo_model_lvg(rxo_state *s, const rxo_source *src, const double *p,
                  double *flux_out, int *niter_out, int *maxiter_hit);
double rxo_lnprior(const rxo_source *src, const double *p);
double rxo_lnprob(rxo_state *s, const rxo_source *src, const double *p,
                  int *status, int *niter_out);

/* Batched drivers (nthreads <= 1: serial; else OpenMP over walkers, one
 * private rxo_state per thread = the reference's Pool semantics).           */
int rxo_lnprob_batch(const rxo_mol *m, int method, double deltav_kms,
                     const rxo_source *src, int N, const double *params,
                     double *lnp, int32_t *status, int32_t *niter, int nthreads);
int rxo_model_flux_batch(const rxo_mol *m, int method, double deltav_kms,
                         const rxo_source *src, int N, const double *params,
                         double *flux, int32_t *status, int32_t *niter, int nthreads);
/* Solve with linear inputs; returns iteration count.  dens[RXO_MAXPART].    */
int rxo_solve_state(const rxo_mol *m, int method, double deltav_kms, double tbg,
                    const double *dens, double tkin, double cdmol,
                    double *xpop, double *tex, double *tau, int *converged);

#ifdef __cplusplus
}
#endif
#endif
