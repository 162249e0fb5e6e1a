// rx_tables.h -- device-resident table layouts shared by the host API
// (rx_api.hip) and the kernels (rx_kernel.hip.inc).  gfx950 only.
#pragma once
#include <stdint.h>

#define RXK_MAXPART 9          // density(9) in COMMON /cphys/ 
#define RXK_MAXLINES 64
#define RXK_MAXLEV 64
#define RXK_MAXNJ 32
#define RXK_MAXSRC 64
#define RXK_ORDER_BUCKETS 32
#ifndef RXK_WAVES_PER_BLOCK
#define RXK_WAVES_PER_BLOCK 4  // one workgroup = 4 wavefronts = 4 walkers in flight, 1 per SIMD
#endif

// Per-line constants.  Everything here is a pure function of the molecular
// data file, evaluated on the host with the same operand order the reference
// uses at run time, so hoisting it out of the iteration changes no rounding.
// The refinement's diagnostic counters are spread over RXK_RF_SLOTS cache lines (128 B apart): five atomics per walker on ONE line were
// 164 k same-address atomics per 32 768-walker launch (a floor of 2 ms for a launch limited to one iteration) and 2.6 % of the
// 1024-walker headline; rx_refinement_counters sums the slots.
#define RXK_RF_SLOTS 64
#define RXK_RF_SLOT_WORDS 16

struct RxLineTab {
    int32_t m[RXK_MAXLINES];        // upper level, 0-based   (iupp-1)
    int32_t n[RXK_MAXLINES];        // lower level, 0-based   (ilow-1)
    double aein[RXK_MAXLINES];      // Einstein A
    double gm[RXK_MAXLINES];        // gstat(m)
    double gn[RXK_MAXLINES];        // gstat(n)
    double agmgn[RXK_MAXLINES];     // A*(gm/gn)
    double fgxta[RXK_MAXLINES];     // (fgaus*xt)/A,   xt = pow(xnu,3)
    double thcxt[RXK_MAXLINES];     // xt*thc          (Fortran constants)
    double fkxnu[RXK_MAXLINES];     // fk*xnu          (Fortran constants)
    double thcxt_py[RXK_MAXLINES];  // thc_py*xt       (astropy constants, core.py:981-984)
    double fkxnu_py[RXK_MAXLINES];  // fk_py*xnu
    // correctly rounded reciprocals of the three per-line divisors: x / c is evaluated as
    // q = x*rc; q += fma(-q, c, x)*rc, which rounds like the division (rx_kernel: div_const)
    double rgn[RXK_MAXLINES], rfgxta[RXK_MAXLINES], rthcxt[RXK_MAXLINES];
};

// Lines incident to each level, in line order (CSR).  entry = line | other<<8 | role<<16,
// role 1: this level is the line's upper level, 0: lower level.
struct RxIncTab {
    int32_t rowptr[RXK_MAXLEV + 1];
    int32_t ent[2 * RXK_MAXLINES];
};

struct RxLevTab {
    double eterm[RXK_MAXLEV];       // cm^-1 (padding levels: 0)
    double gstat[RXK_MAXLEV];       // (padding levels: 1)
    double rgstat[RXK_MAXLEV];      // 1/gstat, correctly rounded on the host (the device divides through it)
    // The collision partners again (the values of RxMolDev below): a kernel stages this table in LDS and reads
    // them there with a run-time partner index -- indexing the kernel-argument copy is a memory round trip per
    // field, five per partner and walker.
    const double *p_temps[RXK_MAXPART];
    const double *p_ksym[RXK_MAXPART];
    int32_t p_ntemp[RXK_MAXPART];
    int32_t pad_;
};

struct RxMolDev {
    int32_t nlev, nline, npart, pad_;
    const RxLevTab *levels;
    const RxLineTab *lines;
    const RxIncTab *inc;
    // collision partners sorted by id (accumulation order of readdata_)
    int32_t pid[RXK_MAXPART];
    int32_t ntemp[RXK_MAXPART];
    const double *temps[RXK_MAXPART];   // [ntemp]
    // dense symmetric downward-rate tables: ksym[p][it][j][i] = K(max_E(i,j) -> min_E(i,j)),
    // leading dimension NL; row j is contiguous in i so a wavefront (lane = i) reads it coalesced
    const double *ksym[RXK_MAXPART];
};

struct RxSourceDev {
    double tbg, T_d, logterm2;
    int32_t nJ, ncomp, data_ok, set;
    int32_t no_prior, pad_;         // no_prior: lnprob = lnlike alone (rx_set_source_prior)
    int32_t jidx[RXK_MAXNJ];        // Jup-1
    double flux[RXK_MAXNJ];
    double esig[RXK_MAXNJ];         // max(|eflux|, 1e-12)
    double bounds[8][2];
    double backi[RXK_MAXLINES];     // backrad_: backi = totalb, trj = tbg
};

enum { RXK_MODE_LNPROB = 0, RXK_MODE_FLUX = 1, RXK_MODE_SOLVE = 2 };
enum { RXK_OK = 0, RXK_MAXITER = 1, RXK_INVALID = 2, RXK_PRIOR = 3 };   // == RX_* of the ABI

struct RxKArgs {
    RxMolDev mol;
    const RxSourceDev *srcs;
    const double *params;           // [N][4*ncomp] log10
    const int32_t *src_index;       // [N] source slot per walker; never null on the device (the host
                                    //   substitutes a handle-owned array filled with the fixed slot)
    int32_t src_fixed;
    int32_t N, ncomp, mode, method, miniter, maxiter;
    int32_t h2_total;               // data file lists 'H2' (id 1): density[0] = pH2+oH2 (core.py:551-554)
    int32_t refine;                 // != 0: most solves from iteration 12 on refine a kept solution (rx_refine.hip.inc); 0: every solve pivoted
    int32_t pad2_;
    float *rf_gmem;                 // two-wavefronts-per-SIMD CO kernels: [grid wavefronts][2] kept inverses (rx_refine.hip.inc); null: no refinement there
    unsigned long long *rf_counters;// optional [RXK_RF_SLOTS][RXK_RF_SLOT_WORDS], the first 5 words of a slot: iterations, solves replaced,
                                    // corrections, attempts given up, inverses kept (atomics; a workgroup adds to slot blockIdx.x % RXK_RF_SLOTS)
    double deltav_cms, fortho;
    // RXK_MODE_SOLVE inputs
    const double *tkin, *cdmol, *dens;
    // work queue: one counter, zeroed by a memset node before every launch
    unsigned int *queue;
    // optional issue order of the walkers (hottest first, rx_order_*_kernel): order[N], or null = 0..N-1
    const int32_t *order;
    int32_t *order_out;             // order kernels only
    unsigned int *order_cnt;        // order kernels only: [2][RXK_ORDER_BUCKETS] counters, zeroed before them
    // outputs
    double *lnp;                    // [N]
    int32_t *status, *niter;        // [N]
    double *flux;                   // [N][nJ]            (MODE_FLUX)
    double *comp_flux;              // [N][ncomp][MAXNJ]  scratch (2-component)
    int32_t *comp_status, *comp_niter;
    double *xpop, *tex, *tau, *sb;  // MODE_SOLVE
};
