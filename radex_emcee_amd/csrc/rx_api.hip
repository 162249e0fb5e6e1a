// rx_api.hip -- host side of libradex_emcee_amd.so: LAMDA parsing, table
// construction, HIP memory/stream plumbing and the extern "C" ABI declared in
// include/radex_emcee_amd.h.  gfx950 only; there is no CPU compute path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/radex_emcee_amd.h"
#include "rx_kernel.hip.inc"
#include "rx_sampler.hip.inc"
#include "rx_tables.h"
#include "rx_lamda.h"

// Padded level counts the solve kernel is instantiated for (one fully unrolled
// kernel each); rx_create picks the smallest one >= nlev.  41 = CO.
#ifndef RX_NL_LIST
// (16 is skipped on purpose: that instantiation crashes ROCm 7.2's register coalescer)
#define RX_NL_LIST 8, 20, 32, 41, 48, 64
#define RX_NL_CASES RX_CASE(8) RX_CASE(20) RX_CASE(32) RX_CASE(41) RX_CASE(48) RX_CASE(64)
#endif

namespace {

// Fortran-side constants used by backrad_ / matrix_
constexpr double H_FK = 1.4387809925261357;
constexpr double H_THC = 3.972907393443411e-16;
constexpr double H_FGAUS = 26.753802360251857;
// astropy CODATA-2018, emcee/pyradex/core.py:981-984
constexpr double H_THC_PY = 3.9728917142978573e-16;
constexpr double H_FK_PY = 1.4387768775039338;

// The LAMDA reader is host-only code of its own (rx_lamda.h: also built by g++ with sanitizers for the corpus test)
using rxl::Partner;
using rxl::Molecule;

int read_molecule(const char *path, Molecule &m, std::string &err)
{
    return rxl::load_lamda(path, m, err) == rxl::LAMDA_OK ? 0 : RX_E_IO;
}

int pick_nl(int nlev)
{
    static const int sizes[] = {RX_NL_LIST};
    for (int s : sizes) if (s >= nlev) return s;
    return -1;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t reserve(size_t want) {
        if (want <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc(&p, want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct rx_handle {
    int device = 0;
    int method = 2;
    double deltav_kms = 1.0;
    double fortho = 0.75;
    int miniter = 10, maxiter = 200;
    Molecule mol;
    int NL = 0;
    int h2_total = 0;
    std::vector<int> order;          // partner slots sorted by id
    std::string err;
    std::string kname;
    // device tables
    void *d_blob = nullptr;          // eterm, gstat, lines, inc, temps
    std::vector<double *> d_ksym;
    RxMolDev dmol{};
    RxSourceDev *d_srcs = nullptr;
    std::vector<RxSourceDev> h_srcs;
    unsigned int *d_queue = nullptr;
    int num_cu = 256, blocks_per_cu2 = 1;
    // staging for the host-pointer API + 2-component scratch
    DevBuf<double> s_params, s_lnp, s_flux, s_cflux, s_in3, s_dens, s_xpop, s_tex, s_tau, s_sb;
    DevBuf<int32_t> s_src, s_status, s_niter, s_cstatus, s_cniter, s_srcfix, s_order;
    // on-device sampler work space (rx_sampler_run_device): proposals, their log-probabilities, ...
    DevBuf<double> w_q, w_factor, w_lnpq;
    DevBuf<int32_t> w_widx, w_qsrc, w_qstatus, w_qniter;
    // dataflow sampler (rx_sampler_run_async_device): per-walker version counters, abort flag
    DevBuf<uint32_t> w_version;      // [N] version counters, then [RING] per-step counters
    DevBuf<double> w_hist;           // [RING][N][ndim] positions by version
    DevBuf<double> w_pend;           // [N][ndim] proposals of the tasks in flight (the dataflow sampler's second hypothesis)
    DevBuf<uint32_t> w_pendver;      // [N]
    uint32_t *d_abort = nullptr;
    uint32_t *h_abort = nullptr;     // pinned mirror, filled by an async copy behind every async run
    long long sampler_timeout_ticks = 1000000000LL;  // 10 s of the 100 MHz wall clock: the longest single wait (a backstop; the watchdog below acts first)
    long long sampler_stall_ticks = 10000000LL;      // 100 ms without ONE finished task anywhere while every rank's grid runs: give up
                                                     //   (a task is a few ms at most: two solves of 200 iterations)
    hipStream_t ctl_stream = nullptr;                // non-blocking: host writes into replicas while a persistent kernel runs (rx_sampler_peer_abort)
    int sampler_grid_limit = 0;      // > 0: the dataflow launches of the peer form occupy at most this many CUs (ranks sharing one GPU)
    unsigned long long *d_stats = nullptr;   // [8] counters of the dataflow launches since the last rx_sampler_stats
    int stats_on = 0;
    unsigned long long last_spec[2] = {0, 0};   // counters [6], [7] as read by the last rx_sampler_stats
    int speculation = -1;            // dataflow sampler: -1 = on where one wavefront runs per SIMD, 0 = off, 1 = on (rx_set_sampler_speculation)
    // multi-GPU dataflow sampler (rx_sampler_peer_*): this rank's replica block and the peers' blocks as mapped here
    struct Peer {
        int nranks = 0, rank = 0, nens = 0, nwalkers = 0, ncomp = 0, nsteps_run = 0;
        size_t N = 0, bytes = 0;
        char *own = nullptr;
        char *base[RX_MAX_RANKS] = {};
        bool opened[RX_MAX_RANKS] = {};          // mapped with hipIpcOpenMemHandle (to be closed)
        char **d_bases = nullptr;                // device copy of base[]
        uint32_t *d_touch = nullptr;             // destination of rx_sampler_peer_connect's one-word reads of the peers' blocks
        unsigned long long off_version = 0, off_done = 0, off_abort = 0, off_alive = 0, off_lnp = 0, off_nacc = 0, off_hist = 0, off_pend = 0, off_pendver = 0;
        bool connected = false, begun = false;
        int same_device = 1;                     // ranks whose replica lives on THIS device (own included): they share its CUs
        int memkind = 0;                         // 1 fine-grained, 2 uncached, 3 plain hipMalloc
        std::vector<std::string> bus_ids;        // PCI bus id of every rank's GPU (rx_sampler_peer_set_bus_ids), or empty
    } peer;
    unsigned int *d_order_cnt = nullptr;
    int force_occ = 0;               // 0: choose by batch size; 1 / 2: wavefronts per SIMD (rx_set_waves_per_simd)
    int rf_count = 0;                // rx_set_refinement_counting: launch the instantiation that adds to d_rf_counters
    int refine = 1;                  // rx_set_refinement: most solves refine a kept solution (rx_refine.hip.inc); 0: every solve pivoted
    unsigned long long *d_rf_counters = nullptr;   // [5] rx_refinement_counters
    float *d_rf_gmem = nullptr;      // the two-wavefront CO kernels' kept inverses: [largest grid x wavefronts][2][rf_minv_floats]
    int issue_order = 1;             // hand large batches out hottest first (rx_set_issue_order; RX_NO_ORDER=1 at rx_create: off)
    int srcfix_value = -1;
    size_t srcfix_filled = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // The queue counter, the fixed-source index, the issue order and the 2-component scratch belong to the
    // handle, so its launches must not overlap: every launch records ev_done on its stream, and a launch
    // on ANOTHER stream first makes that stream wait for it (same stream: stream order already does).
    hipEvent_t ev_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool in_flight = false;
};

namespace {

int hip_fail(rx_handle *h, hipError_t e, const char *what)
{
    h->err = std::string(what) + ": " + hipGetErrorString(e);
    return RX_E_HIP;
}
#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail((h), e_, #call); } while (0)

typedef void (*kernel_fn)(const RxKArgs);

// the specialised instantiation: the molecule fills it (no padding levels), the geometry is LVG, the hot path
// (emcee/emcee_radex.py:116 escapeProbGeom='lvg': no geometry branches in its iteration), AND the molecule is a LADDER
// -- line l connects level l+1 to level l, as in every linear rotor's LAMDA file (CO) -- so that a line and its lower
// level share a lane and the upper level is the next lane: the per-line / per-level exchanges of the iteration are DPP lane
// shifts instead of trips through LDS (solve_wave, "ladder").  Everything else runs the general instantiation of the same size.
static bool is_ladder(const Molecule &m)
{
    if (m.nline != m.nlev - 1) return false;
    for (int l = 0; l < m.nline; ++l)
        if (m.iupp[l] - 1 != l + 1 || m.ilow[l] - 1 != l) return false;
    return true;
}
// (instantiated for the 41 levels of CO only: kernel_for / sampler_kernel_for)
static bool is_exact(const rx_handle *h) { return h->NL == 41 && h->mol.nlev == h->NL && h->method == 2 && is_ladder(h->mol); }
typedef void (*lukernel_fn)(const double *, double *, int32_t *, int, int);

lukernel_fn lukernel_for(int NL)
{
    switch (NL) {
#define RX_CASE(n) case n: return rxk::rx_lubksb_kernel<n>;
        RX_NL_CASES
#undef RX_CASE
    }
    return nullptr;
}

typedef void (*sampler_kernel_fn)(const rxs::AsyncArgs);
sampler_kernel_fn sampler_kernel_for(int NL, int occ, bool exact)
{
#ifdef RX_NO_SAMPLER_KERNEL
    return nullptr;
#else
    if (NL == 41 && exact)
        return occ >= 2 ? rxs::rx_sampler_kernel<41, 2, true> : rxs::rx_sampler_kernel<41, 1, true>;
    switch (NL) {
#define RX_CASE(n) case n: if constexpr (rxk::occ2_compiled(n)) { if (occ >= 2) return rxs::rx_sampler_kernel<n < 42 ? n : 8, 2, false>; } \
                   return rxs::rx_sampler_kernel<n, 1, false>;
        RX_NL_CASES
#undef RX_CASE
    }
    return nullptr;
#endif
}

// exact = the molecule fills the instantiation (nlev == NL): only built for CO's 41 levels
kernel_fn kernel_for(int NL, int occ, bool exact, bool general = false)
{
    if (NL == 41 && exact) {
        if (general) return occ >= 2 ? rxk::rx_solve_kernel<41, 2, true, true> : rxk::rx_solve_kernel<41, 1, true, true>;
        return occ >= 2 ? rxk::rx_solve_kernel<41, 2, true> : rxk::rx_solve_kernel<41, 1, true>;
    }
    switch (NL) {
#define RX_CASE(n) case n: if constexpr (rxk::occ2_compiled(n)) { if (occ >= 2) return rxk::rx_solve_kernel<n < 42 ? n : 8, 2, false>; } \
                   return rxk::rx_solve_kernel<n, 1, false>;
        RX_NL_CASES
#undef RX_CASE
    }
    return nullptr;
}

int build_tables(rx_handle *h)
{
    const Molecule &m = h->mol;
    const int NL = h->NL;
    // partner order = ascending id (accumulation order of readdata_, [BIN 0x1eaf9-0x1f378])
    h->order.resize(m.parts.size());
    for (size_t i = 0; i < m.parts.size(); ++i) h->order[i] = (int)i;
    std::stable_sort(h->order.begin(), h->order.end(),
                     [&](int x, int y) { return m.parts[x].id < m.parts[y].id; });
    h->h2_total = 0;
    for (const Partner &P : m.parts) if (P.id == 1) h->h2_total = 1;

    RxLevTab LV;
    for (int q = 0; q < RXK_MAXPART; ++q) { LV.p_temps[q] = nullptr; LV.p_ksym[q] = nullptr; LV.p_ntemp[q] = 0; }
    LV.pad_ = 0;
    for (int i = 0; i < RXK_MAXLEV; ++i) { LV.eterm[i] = 0.0; LV.gstat[i] = 1.0; LV.rgstat[i] = 1.0; }
    for (int i = 0; i < m.nlev; ++i) { LV.eterm[i] = m.eterm[i]; LV.gstat[i] = m.gstat[i]; LV.rgstat[i] = 1.0 / m.gstat[i]; }
    RxLineTab LT;
    memset(&LT, 0, sizeof LT);
    for (int l = 0; l < RXK_MAXLINES; ++l) { LT.gm[l] = LT.gn[l] = 1.0; LT.aein[l] = 1.0; LT.fgxta[l] = 1.0; LT.thcxt[l] = 1.0; LT.rgn[l] = LT.rfgxta[l] = LT.rthcxt[l] = 1.0; }
    for (int l = 0; l < m.nline; ++l) {
        const int mu = m.iupp[l] - 1, nl_ = m.ilow[l] - 1;
        const double A = m.aeinst[l], gm = m.gstat[mu], gn = m.gstat[nl_];
        const double xnu = m.xnu[l];
        const double xt = pow(xnu, 3.0);                  // xnu**3. -> pow() [BIN 0x19a71]
        LT.m[l] = mu; LT.n[l] = nl_;
        LT.aein[l] = A; LT.gm[l] = gm; LT.gn[l] = gn;
        LT.agmgn[l] = A * (gm / gn);
        LT.fgxta[l] = H_FGAUS * xt / A;
        LT.thcxt[l] = xt * H_THC;
        LT.rgn[l] = 1.0 / LT.gn[l]; LT.rfgxta[l] = 1.0 / LT.fgxta[l]; LT.rthcxt[l] = 1.0 / LT.thcxt[l];
        LT.fkxnu[l] = H_FK * xnu;
        LT.thcxt_py[l] = H_THC_PY * xt;
        LT.fkxnu_py[l] = H_FK_PY * xnu;
    }
    RxIncTab IT;
    memset(&IT, 0, sizeof IT);
    int e = 0;
    for (int i = 0; i < RXK_MAXLEV; ++i) {
        IT.rowptr[i] = e;
        if (i < m.nlev)
            for (int l = 0; l < m.nline; ++l) {          // line order = accumulation order of matrix_
                if (m.iupp[l] - 1 == i) IT.ent[e++] = l | ((m.ilow[l] - 1) << 8) | (1 << 16);
                else if (m.ilow[l] - 1 == i) IT.ent[e++] = l | ((m.iupp[l] - 1) << 8);
            }
    }
    IT.rowptr[RXK_MAXLEV] = e;

    size_t off_e = 0;
    size_t off_l = (off_e + sizeof(RxLevTab) + 15) & ~size_t(15);
    size_t off_i = off_l + sizeof(RxLineTab);
    size_t off_t = (off_i + sizeof(RxIncTab) + 15) & ~size_t(15);
    size_t total = off_t;
    std::vector<size_t> off_temps;
    for (int s : h->order) { off_temps.push_back(total); total += m.parts[s].ntemp * sizeof(double); }
    std::vector<char> blob(total, 0);
    memcpy(blob.data() + off_e, &LV, sizeof LV);
    memcpy(blob.data() + off_l, &LT, sizeof LT);
    memcpy(blob.data() + off_i, &IT, sizeof IT);
    for (size_t q = 0; q < h->order.size(); ++q)
        memcpy(blob.data() + off_temps[q], m.parts[h->order[q]].temps.data(),
               m.parts[h->order[q]].ntemp * sizeof(double));
    HIPCHK(h, hipMalloc(&h->d_blob, total));
    HIPCHK(h, hipMemcpy(h->d_blob, blob.data(), total, hipMemcpyHostToDevice));

    RxMolDev &D = h->dmol;
    memset(&D, 0, sizeof D);
    D.nlev = m.nlev; D.nline = m.nline; D.npart = (int)m.parts.size();
    char *base = (char *)h->d_blob;
    D.levels = (const RxLevTab *)(base + off_e);
    D.lines = (const RxLineTab *)(base + off_l);
    D.inc = (const RxIncTab *)(base + off_i);
    for (size_t q = 0; q < h->order.size(); ++q) {
        const Partner &P = m.parts[h->order[q]];
        D.pid[q] = P.id; D.ntemp[q] = P.ntemp;
        D.temps[q] = (const double *)(base + off_temps[q]);
        // dense symmetric table [ntemp][NL][NL]
        std::vector<double> K((size_t)P.ntemp * NL * NL, 0.0);
        for (int c = 0; c < P.ncoll; ++c) {
            const int u = P.lcu[c] - 1, l = P.lcl[c] - 1;
            for (int t = 0; t < P.ntemp; ++t) {
                const double v = P.coll[(size_t)c * P.ntemp + t];
                // (a pair listed twice: the LAST row stands -- the reference writes its table colld(up,low), it does not add;
                // tests/golden/ref_lamda_corpus.json: ok_duplicate_rate_row)
                K[((size_t)t * NL + u) * NL + l] = v;
                K[((size_t)t * NL + l) * NL + u] = v;
            }
        }
        double *dK = nullptr;
        HIPCHK(h, hipMalloc(&dK, K.size() * sizeof(double)));
        HIPCHK(h, hipMemcpy(dK, K.data(), K.size() * sizeof(double), hipMemcpyHostToDevice));
        h->d_ksym.push_back(dK);
        D.ksym[q] = dK;
    }
    // the partner table inside the level table (what the kernels read from LDS), now that the addresses exist
    for (size_t q = 0; q < h->order.size(); ++q) { LV.p_temps[q] = D.temps[q]; LV.p_ksym[q] = D.ksym[q]; LV.p_ntemp[q] = D.ntemp[q]; }
    HIPCHK(h, hipMemcpy(base + off_e, &LV, sizeof LV, hipMemcpyHostToDevice));
    return 0;
}

int validate_molecule(rx_handle *h)
{
    const Molecule &m = h->mol;
    if (m.nlev > RX_MAX_LEVELS || m.nline > RX_MAX_LINES) {
        h->err = "molecule exceeds kernel limits (nlev<=64, nline<=64): one level per lane of a wavefront";
        return RX_E_UNSUPP;
    }
    for (const Partner &P : m.parts)
        for (int c = 0; c < P.ncoll; ++c) {
            const double eu = m.eterm[P.lcu[c] - 1], el = m.eterm[P.lcl[c] - 1];
            if (!(eu > el)) {
                h->err = "collisional transition with E_up <= E_low is not supported by the dense rate table";
                return RX_E_UNSUPP;
            }
            for (int t = 0; t < P.ntemp; ++t)
                if (P.coll[(size_t)c * P.ntemp + t] < 0.0) { h->err = "negative collision rate"; return RX_E_IO; }
        }
    // the kernels find the bracket T_i < T_kin <= T_i+1 of the reference's sequential search by COUNTING the grid points below T_kin:
    // the same bracket for a grid that ascends (repeats allowed); a grid that descends never reaches the search (T_kin <= T_1 or
    // T_kin >= T_ntemp always holds); any other order would give another bracket than the reference's
    for (const Partner &P : m.parts) {
        bool up = true, down = true;
        for (int t = 0; t + 1 < P.ntemp; ++t) { up = up && P.temps[t] <= P.temps[t + 1]; down = down && P.temps[t] >= P.temps[t + 1]; }
        if (!up && !down) { h->err = "collision temperatures are neither in ascending nor in descending order"; return RX_E_UNSUPP; }
    }
    for (size_t a = 0; a < m.parts.size(); ++a)
        for (size_t b = a + 1; b < m.parts.size(); ++b)
            if (m.parts[a].id == m.parts[b].id) { h->err = "duplicate collision partner id"; return RX_E_UNSUPP; }
    // two lines between the same pair of levels would need read-modify-write in Phase B
    for (int l = 0; l < m.nline; ++l)
        for (int k = l + 1; k < m.nline; ++k)
            if ((m.iupp[l] == m.iupp[k] && m.ilow[l] == m.ilow[k]) ||
                (m.iupp[l] == m.ilow[k] && m.ilow[l] == m.iupp[k])) {
                h->err = "two radiative transitions between the same pair of levels";
                return RX_E_UNSUPP;
            }
    return 0;
}

int fill_args(rx_handle *h, RxKArgs &a, int N, int ncomp, int mode)
{
    memset(&a, 0, sizeof a);
    a.mol = h->dmol;
    a.srcs = h->d_srcs;
    a.N = N; a.ncomp = ncomp; a.mode = mode; a.method = h->method;
    a.miniter = h->miniter; a.maxiter = h->maxiter; a.h2_total = h->h2_total;
    a.refine = h->refine; a.rf_counters = h->d_rf_counters; a.rf_gmem = h->d_rf_gmem;
    a.deltav_cms = h->deltav_kms * 1e5;      // core.py:447-454: km/s -> cm/s
    a.fortho = h->fortho;
    a.queue = h->d_queue;
    return 0;
}

// The kernels always read src_index[w]; callers that address one source get a handle-owned
// array filled with that slot (refilled only when the slot or the batch size changes).
int fixed_src_index(rx_handle *h, RxKArgs &a, hipStream_t st)
{
    if (a.src_index || a.N <= 0) return 0;
    const size_t n = (size_t)a.N;
    if (h->s_srcfix.n < n) { HIPCHK(h, h->s_srcfix.reserve(std::max(n, (size_t)4096))); h->srcfix_filled = 0; }
    if (h->srcfix_value != a.src_fixed || h->srcfix_filled < n) {
        HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)h->s_srcfix.p, a.src_fixed, h->s_srcfix.n, st));
        h->srcfix_value = a.src_fixed;
        h->srcfix_filled = h->s_srcfix.n;
    }
    a.src_index = h->s_srcfix.p;
    return 0;
}

// orders `st` behind whatever this handle launched last (no-op on the same stream)
int order_after_last(rx_handle *h, hipStream_t st)
{
    if (h->in_flight && st != h->last_stream) HIPCHK(h, hipStreamWaitEvent(st, h->ev_done, 0));
    return 0;
}

int mark_launched(rx_handle *h, hipStream_t st)
{
    HIPCHK(h, hipEventRecord(h->ev_done, st));
    h->last_stream = st;
    h->in_flight = true;
    return 0;
}

int launch(rx_handle *h, RxKArgs &a, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    if (a.N <= 0) return 0;
    { int rc = order_after_last(h, st); if (rc) return rc; }
    { int rc = fixed_src_index(h, a, st); if (rc) return rc; }
    const int ncomp = (a.mode == RXK_MODE_SOLVE) ? 1 : a.ncomp;
    const long items = (long)a.N * ncomp;
    long blocks = (items + RXK_WAVES_PER_BLOCK - 1) / RXK_WAVES_PER_BLOCK;
    // one wavefront per SIMD up to ~7 rounds of the chip, then the two-wavefront build: measured crossover between 6144 and 8192
    // walkers with the refinement (scripts/occ_crossover.py, round 5: 6144 walkers 1.64 against 1.78 ms, 8192 walkers 1.88 against
    // 1.73) -- below it a launch is mostly its 200-iteration walkers, which run faster alone on their SIMD.  (Both builds give
    // the same bits: the choice is scheduling only.)
    int occ = (items > 7L * h->num_cu * RXK_WAVES_PER_BLOCK && h->blocks_per_cu2 >= 2) ? 2 : 1;
    if (h->force_occ == 1 || (h->force_occ == 2 && h->blocks_per_cu2 >= 2)) occ = h->force_occ;
    const long cap = (long)h->num_cu * (occ == 2 ? h->blocks_per_cu2 : 1);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    // (a handle in its default state -- the reference's iteration limits, refinement on, no counting -- runs the instantiation that has
    // those as constants: rx_kernel.hip.inc, GEN)
    const bool general = h->rf_count != 0 || h->refine == 0 || h->miniter != 10 || h->maxiter != 200;
    kernel_fn k = kernel_for(h->NL, occ, is_exact(h), general);
    // (every wavefront of the grid takes the item of its own index first, without the queue: the counter starts behind them)
    HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)h->d_queue, (int)(blocks * RXK_WAVES_PER_BLOCK), 1, st));
    if (e0) HIPCHK(h, hipEventRecord(e0, st));
    // more items than resident wavefronts: hand the walkers out hottest first (see rx_order_bucket)
    a.order = nullptr;
    if (items > 2 * cap * RXK_WAVES_PER_BLOCK && h->issue_order) {
        HIPCHK(h, h->s_order.reserve((size_t)a.N));
        a.order_out = h->s_order.p;
        a.order_cnt = h->d_order_cnt;
        HIPCHK(h, hipMemsetAsync(h->d_order_cnt, 0, 2 * RXK_ORDER_BUCKETS * sizeof(unsigned int), st));
        const int tb = 256, nb = (a.N + tb - 1) / tb;
        hipLaunchKernelGGL(rxk::rx_order_count_kernel, dim3(nb), dim3(tb), 0, st, a);
        hipLaunchKernelGGL(rxk::rx_order_scatter_kernel, dim3(nb), dim3(tb), 0, st, a);
        a.order = h->s_order.p;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * RXK_WAVES_PER_BLOCK), 0, st, a);
    if (ncomp == 2) {
        const int tb = 256;
        hipLaunchKernelGGL(rxk::rx_combine_kernel, dim3((a.N + tb - 1) / tb), dim3(tb), 0, st, a);
    }
    if (e1) HIPCHK(h, hipEventRecord(e1, st));
    HIPCHK(h, hipGetLastError());
    return mark_launched(h, st);
}

int ensure_comp_scratch(rx_handle *h, RxKArgs &a, int N, int ncomp)
{
    if (ncomp != 2) return 0;
    HIPCHK(h, h->s_cflux.reserve((size_t)N * ncomp * RXK_MAXNJ));
    HIPCHK(h, h->s_cstatus.reserve((size_t)N * ncomp));
    HIPCHK(h, h->s_cniter.reserve((size_t)N * ncomp));
    a.comp_flux = h->s_cflux.p; a.comp_status = h->s_cstatus.p; a.comp_niter = h->s_cniter.p;
    return 0;
}

int source_ncomp(rx_handle *h, const int32_t *src_index_host, int N, int src_fixed, int *ncomp_out)
{
    int nc = -1;
    auto chk = [&](int s) -> int {
        if (s < 0 || s >= RX_MAX_SOURCES || !h->h_srcs[s].set) { h->err = "source slot not set"; return RX_E_STATE; }
        if (nc < 0) nc = h->h_srcs[s].ncomp;
        else if (nc != h->h_srcs[s].ncomp) { h->err = "sources of one batch must share ncomp"; return RX_E_ARG; }
        return 0;
    };
    if (!src_index_host) { int rc = chk(src_fixed); if (rc) return rc; }
    else for (int i = 0; i < N; ++i) { int rc = chk(src_index_host[i]); if (rc) return rc; }
    *ncomp_out = nc;
    return 0;
}

}  // namespace

// ============================ extern "C" ABI ==================================
extern "C" {

int rx_abi_version(void) { return RX_ABI_VERSION; }

rx_handle *rx_create(const char *lamda_path, int method, double deltav_kms, int device,
                     char *err, size_t errlen)
{
    auto seterr = [&](const std::string &s) { if (err && errlen) { strncpy(err, s.c_str(), errlen - 1); err[errlen - 1] = 0; } };
    if (!lamda_path || method < 1 || method > 3 || !(deltav_kms > 0) || device < 0) { seterr("bad argument"); return nullptr; }
    rx_handle *h = new rx_handle;
    h->device = device; h->method = method; h->deltav_kms = deltav_kms;
    int rc = read_molecule(lamda_path, h->mol, h->err);
    if (!rc) rc = validate_molecule(h);
    if (!rc) { h->NL = pick_nl(h->mol.nlev); if (h->NL < 0) { h->err = "no kernel instantiation for this nlev"; rc = RX_E_UNSUPP; } }
    if (rc) { seterr(h->err); delete h; return nullptr; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= device) {
        seterr("no usable HIP device (libradex_emcee_amd has no CPU fallback)");
        delete h; return nullptr;
    }
    auto hipfail = [&](const char *what, hipError_t ee) { seterr(std::string(what) + ": " + hipGetErrorString(ee)); rx_destroy(h); return (rx_handle *)nullptr; };
    if ((e = hipSetDevice(device)) != hipSuccess) return hipfail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return hipfail("hipGetDeviceProperties", e);
    h->num_cu = prop.multiProcessorCount;
    if (build_tables(h)) { seterr(h->err); rx_destroy(h); return nullptr; }
    h->h_srcs.resize(RX_MAX_SOURCES);
    for (auto &s : h->h_srcs) memset(&s, 0, sizeof s);
    if ((e = hipMalloc(&h->d_srcs, sizeof(RxSourceDev) * RX_MAX_SOURCES)) != hipSuccess) return hipfail("hipMalloc", e);
    if ((e = hipMemset(h->d_srcs, 0, sizeof(RxSourceDev) * RX_MAX_SOURCES)) != hipSuccess) return hipfail("hipMemset", e);
    if ((e = hipMalloc(&h->d_queue, sizeof(unsigned int))) != hipSuccess) return hipfail("hipMalloc", e);
    if ((e = hipMalloc(&h->d_order_cnt, 2 * RXK_ORDER_BUCKETS * sizeof(unsigned int))) != hipSuccess) return hipfail("hipMalloc", e);
    if ((e = hipMalloc(&h->d_rf_counters, RXK_RF_SLOTS * RXK_RF_SLOT_WORDS * sizeof(unsigned long long))) != hipSuccess) return hipfail("hipMalloc", e);
    if ((e = hipMemset(h->d_rf_counters, 0, RXK_RF_SLOTS * RXK_RF_SLOT_WORDS * sizeof(unsigned long long))) != hipSuccess) return hipfail("hipMemset", e);
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return hipfail("hipEventCreate", e);
    if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return hipfail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming)) != hipSuccess) return hipfail("hipEventCreate", e);
    { const char *no = getenv("RX_NO_ORDER"); h->issue_order = (no && *no && *no != '0') ? 0 : 1; }   // read once
    int nb = 0;
    kernel_fn k = kernel_for(h->NL, 2, is_exact(h));
    if (rxk::occ2_compiled(h->NL)    // (42-64 levels: no two-wavefront kernels, kernel_for hands out the one-wavefront one)
        && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k, 64 * RXK_WAVES_PER_BLOCK, 0) == hipSuccess && nb > 0)
        h->blocks_per_cu2 = std::min(nb, 2);
    if (rxk::refine_compiled(h->NL, is_exact(h), 2) && h->blocks_per_cu2 >= 2) {
        // (no launch of a two-wavefront kernel has more workgroups than this: launch(), rx_sampler_run_async_device, peer_launch_shape)
        const size_t waves = (size_t)h->num_cu * h->blocks_per_cu2 * RXK_WAVES_PER_BLOCK;
        if ((e = hipMalloc(&h->d_rf_gmem, waves * 2 * rxk::rf_minv_floats(h->NL) * sizeof(float))) != hipSuccess) return hipfail("hipMalloc", e);
    }
    char nm[64];
    snprintf(nm, sizeof nm, "rx_solve_kernel<%d, 1, %s>", h->NL, is_exact(h) ? "true" : "false");
    h->kname = nm;
    return h;
}

void rx_destroy(rx_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);                                           // (a multi-GPU process may be on another one)
    if (h->in_flight && h->ev_done) (void)hipEventSynchronize(h->ev_done);   // nothing of this handle still runs
    h->in_flight = false;
    if (h->d_blob) (void)hipFree(h->d_blob);
    for (double *p : h->d_ksym) (void)hipFree(p);
    if (h->d_srcs) (void)hipFree(h->d_srcs);
    if (h->d_queue) (void)hipFree(h->d_queue);
    if (h->d_order_cnt) (void)hipFree(h->d_order_cnt);
    if (h->d_rf_counters) (void)hipFree(h->d_rf_counters);
    if (h->d_rf_gmem) (void)hipFree(h->d_rf_gmem);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->ctl_stream) (void)hipStreamDestroy(h->ctl_stream);
    h->s_params.release(); h->s_lnp.release(); h->s_flux.release(); h->s_cflux.release();
    h->s_in3.release(); h->s_dens.release(); h->s_xpop.release(); h->s_tex.release();
    h->s_tau.release(); h->s_sb.release(); h->s_src.release(); h->s_status.release();
    h->s_niter.release(); h->s_cstatus.release(); h->s_cniter.release(); h->s_srcfix.release();
    h->s_order.release();
    h->w_q.release(); h->w_factor.release(); h->w_lnpq.release(); h->w_widx.release();
    h->w_qsrc.release(); h->w_qstatus.release(); h->w_qniter.release(); h->w_version.release(); h->w_hist.release(); h->w_pend.release(); h->w_pendver.release();
    if (h->d_abort) (void)hipFree(h->d_abort);
    if (h->d_stats) (void)hipFree(h->d_stats);
    (void)rx_sampler_peer_close(h);
    if (h->h_abort) (void)hipHostFree(h->h_abort);
    delete h;
}

const char *rx_last_error(const rx_handle *h) { return h ? h->err.c_str() : "null handle"; }
int rx_nlev(const rx_handle *h) { return h ? h->mol.nlev : RX_E_ARG; }
int rx_nline(const rx_handle *h) { return h ? h->mol.nline : RX_E_ARG; }
int rx_npart(const rx_handle *h) { return h ? (int)h->mol.parts.size() : RX_E_ARG; }
const char *rx_kernel_name(const rx_handle *h) { return h ? h->kname.c_str() : ""; }

int rx_partner_ids(const rx_handle *h, int32_t *out)
{
    if (!h || !out) return RX_E_ARG;
    for (size_t i = 0; i < h->mol.parts.size(); ++i) out[i] = h->mol.parts[i].id;
    return 0;
}

int rx_line_data(const rx_handle *h, double *xnu, double *spfreq, int32_t *iupp, int32_t *ilow)
{
    if (!h) return RX_E_ARG;
    for (int l = 0; l < h->mol.nline; ++l) {
        if (xnu) xnu[l] = h->mol.xnu[l];
        if (spfreq) spfreq[l] = h->mol.spfreq[l];
        if (iupp) iupp[l] = h->mol.iupp[l];
        if (ilow) ilow[l] = h->mol.ilow[l];
    }
    return 0;
}

int rx_background(rx_handle *h, int src, double *backi, double *tbg_out)
{
    if (!h) return RX_E_ARG;
    if (src < 0 || src >= RX_MAX_SOURCES || !h->h_srcs[src].set) { h->err = "source slot not set"; return RX_E_STATE; }
    RxSourceDev S;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(&S, h->d_srcs + src, sizeof S, hipMemcpyDeviceToHost));
    for (int l = 0; l < h->mol.nline && backi; ++l) backi[l] = S.backi[l];
    if (tbg_out) *tbg_out = S.tbg;
    return 0;
}

int rx_set_fortho(rx_handle *h, double fortho)
{
    if (!h || !(fortho >= 0.0 && fortho <= 1.0)) return RX_E_ARG;
    h->fortho = fortho;
    return 0;
}

int rx_set_iteration_limits(rx_handle *h, int miniter, int maxiter)
{
    if (!h || miniter < 0 || maxiter < 1) return RX_E_ARG;
    h->miniter = miniter; h->maxiter = maxiter;
    return 0;
}

int rx_set_source(rx_handle *h, int src, double tbg, int nJ, const int32_t *Jup, const double *flux,
                  const double *eflux, const double *bounds, int ncomp, double T_d)
{
    if (!h) return RX_E_ARG;
    if (src < 0 || src >= RX_MAX_SOURCES || nJ < 0 || nJ > RX_MAX_NJ || (ncomp != 1 && ncomp != 2) ||
        (nJ > 0 && (!Jup || !flux || !eflux)) || !bounds) { h->err = "rx_set_source: bad argument"; return RX_E_ARG; }
    if (!(tbg > 0.0)) { h->err = "rx_set_source: tbg must be > 0 (tbg<=0 selects RADEX's user-file/galactic backgrounds, out of scope)"; return RX_E_ARG; }
    RxSourceDev S;
    memset(&S, 0, sizeof S);
    S.tbg = tbg; S.T_d = T_d; S.nJ = nJ; S.ncomp = ncomp; S.set = 1; S.data_ok = 1;
    double logsum = 0.0;
    for (int j = 0; j < nJ; ++j) {
        if (Jup[j] < 1 || Jup[j] > h->mol.nline) { h->err = "rx_set_source: Jup outside the molecule's line list"; return RX_E_ARG; }
        S.jidx[j] = Jup[j] - 1;
        S.flux[j] = flux[j];
        double e = fabs(eflux[j]);                       // np.maximum(np.abs(eflux), 1e-12)
        if (!(e > 1e-12)) e = std::isnan(e) ? e : 1e-12;
        S.esig[j] = e;
        if (!std::isfinite(flux[j]) || !std::isfinite(e)) S.data_ok = 0;   // emcee_radex.py:144,149
        logsum += log(e);
    }
    S.logterm2 = 2.0 * logsum;                           // 2*sum(log e), emcee_radex.py:165
    for (int k = 0; k < 4 * ncomp; ++k) { S.bounds[k][0] = bounds[2 * k]; S.bounds[k][1] = bounds[2 * k + 1]; }
    // backrad_, tbg > 0 branch [BIN 0x1be30-0x1c390]
    for (int l = 0; l < h->mol.nline; ++l) {
        const double x = h->mol.xnu[l];
        const double hh = H_FK * x / tbg;
        S.backi[l] = (hh >= 160.0) ? 1e-30 : H_THC * pow(x, 3.0) / (exp(hh) - 1.0);   // (the double 1e-30 [BIN 0x26bb8], not 1e-30f: ref_backrad_guard.json)
    }
    HIPCHK(h, hipSetDevice(h->device));
    // the table is read by kernels that may still be running on a non-blocking stream: wait for them
    if (h->in_flight) { HIPCHK(h, hipEventSynchronize(h->ev_done)); h->in_flight = false; }
    HIPCHK(h, hipMemcpy(h->d_srcs + src, &S, sizeof S, hipMemcpyHostToDevice));
    h->h_srcs[src] = S;
    return 0;
}

int rx_set_source_prior(rx_handle *h, int src, int enabled)
{
    if (!h) return RX_E_ARG;
    if (src < 0 || src >= RX_MAX_SOURCES || !h->h_srcs[src].set) { h->err = "source slot not set"; return RX_E_STATE; }
    h->h_srcs[src].no_prior = enabled ? 0 : 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->in_flight) { HIPCHK(h, hipEventSynchronize(h->ev_done)); h->in_flight = false; }
    HIPCHK(h, hipMemcpy(h->d_srcs + src, &h->h_srcs[src], sizeof(RxSourceDev), hipMemcpyHostToDevice));
    return 0;
}

int rx_set_waves_per_simd(rx_handle *h, int waves)
{
    if (!h || waves < 0 || waves > 2) return RX_E_ARG;
    h->force_occ = waves;
    return 0;
}

int rx_set_refinement(rx_handle *h, int enabled)
{
    if (!h) return RX_E_ARG;
    h->refine = enabled ? 1 : 0;
    return 0;
}

int rx_set_refinement_counting(rx_handle *h, int enabled)
{
    if (!h) return RX_E_ARG;
    h->rf_count = enabled ? 1 : 0;
    return 0;
}

int rx_refinement_counters(rx_handle *h, uint64_t *out5, int reset)
{
    if (!h || !out5) return RX_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->in_flight) { HIPCHK(h, hipEventSynchronize(h->ev_done)); h->in_flight = false; }
    unsigned long long v[RXK_RF_SLOTS * RXK_RF_SLOT_WORDS];                  // (one slot per cache line: rx_tables.h)
    HIPCHK(h, hipMemcpy(v, h->d_rf_counters, sizeof v, hipMemcpyDeviceToHost));
    for (int i = 0; i < 5; ++i) {
        out5[i] = 0;
        for (int s = 0; s < RXK_RF_SLOTS; ++s) out5[i] += v[s * RXK_RF_SLOT_WORDS + i];
    }
    if (reset) HIPCHK(h, hipMemset(h->d_rf_counters, 0, sizeof v));
    return 0;
}

int rx_set_issue_order(rx_handle *h, int hottest_first)
{
    if (!h) return RX_E_ARG;
    h->issue_order = hottest_first ? 1 : 0;
    return 0;
}

// ncomp of a device-pointer batch: the caller states it.  Without a per-walker index the batch addresses
// slot 0, which must be set and agree; with one the kernel reports every walker whose slot is out of range,
// unset or of another ncomp as RX_INVALID (the host cannot see a device-side index).
static int device_batch_ncomp(rx_handle *h, int N, int ncomp, const int32_t *d_src_index)
{
    if (ncomp != 1 && ncomp != 2) { h->err = "ncomp must be 1 or 2"; return RX_E_ARG; }
    if (!d_src_index) {
        int nc = 0;
        int rc = source_ncomp(h, nullptr, N, 0, &nc);
        if (rc) return rc;
        if (nc != ncomp) { h->err = "ncomp differs from the source in slot 0"; return RX_E_ARG; }
    }
    return 0;
}

static int lnprob_device(rx_handle *h, int N, int ncomp, const double *d_params, const int32_t *d_src_index,
                         double *d_lnp, int32_t *d_status, int32_t *d_niter, hipStream_t st,
                         hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    if (!h || N < 0 || (N > 0 && (!d_params || !d_lnp))) return RX_E_ARG;
    { int rc = device_batch_ncomp(h, N, ncomp, d_src_index); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    RxKArgs a;
    fill_args(h, a, N, ncomp, RXK_MODE_LNPROB);
    a.params = d_params; a.src_index = d_src_index; a.src_fixed = 0;
    a.lnp = d_lnp; a.status = d_status; a.niter = d_niter;
    { int rc = ensure_comp_scratch(h, a, N, ncomp); if (rc) return rc; }
    return launch(h, a, st, e0, e1);
}

int rx_lnprob_batch_device(rx_handle *h, int N, int ncomp, const double *d_params, const int32_t *d_src_index,
                           double *d_lnp, int32_t *d_status, int32_t *d_niter, void *stream)
{
    return lnprob_device(h, N, ncomp, d_params, d_src_index, d_lnp, d_status, d_niter, (hipStream_t)stream);
}

int rx_lnprob_batch(rx_handle *h, int N, const double *params, const int32_t *src_index,
                    double *lnp, int32_t *status, int32_t *niter)
{
    if (!h || N < 0 || (N > 0 && (!params || !lnp))) return RX_E_ARG;
    if (N == 0) return 0;
    int ncomp = 0;
    { int rc = source_ncomp(h, src_index, N, 0, &ncomp); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t np = (size_t)N * 4 * ncomp;
    HIPCHK(h, h->s_params.reserve(np));
    HIPCHK(h, h->s_lnp.reserve(N));
    HIPCHK(h, h->s_status.reserve(N));
    HIPCHK(h, h->s_niter.reserve(N));
    HIPCHK(h, hipMemcpy(h->s_params.p, params, np * sizeof(double), hipMemcpyHostToDevice));
    const int32_t *dsrc = nullptr;
    if (src_index) {
        HIPCHK(h, h->s_src.reserve(N));
        HIPCHK(h, hipMemcpy(h->s_src.p, src_index, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice));
        dsrc = h->s_src.p;
    }
    RxKArgs a;
    fill_args(h, a, N, ncomp, RXK_MODE_LNPROB);
    a.params = h->s_params.p; a.src_index = dsrc; a.src_fixed = 0;
    a.lnp = h->s_lnp.p; a.status = h->s_status.p; a.niter = h->s_niter.p;
    { int rc = ensure_comp_scratch(h, a, N, ncomp); if (rc) return rc; }
    { int rc = launch(h, a, nullptr); if (rc) return rc; }
    HIPCHK(h, hipMemcpy(lnp, h->s_lnp.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    if (status) HIPCHK(h, hipMemcpy(status, h->s_status.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (niter) HIPCHK(h, hipMemcpy(niter, h->s_niter.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int rx_lnprior_batch(rx_handle *h, int src, int N, const double *params, double *lnprior_out)
{
    if (!h || N < 0 || (N > 0 && (!params || !lnprior_out))) return RX_E_ARG;
    if (N == 0) return 0;
    int ncomp = 0;
    { int rc = source_ncomp(h, nullptr, N, src, &ncomp); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t np = (size_t)N * 4 * ncomp;
    HIPCHK(h, h->s_params.reserve(np));
    HIPCHK(h, h->s_lnp.reserve(N));
    hipStream_t st = nullptr;
    { int rc = order_after_last(h, st); if (rc) return rc; }
    HIPCHK(h, hipMemcpyAsync(h->s_params.p, params, np * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rxk::rx_lnprior_kernel, dim3((N + 255) / 256), dim3(256), 0, st,
                       (const RxSourceDev *)h->d_srcs, src, ncomp, (const double *)h->s_params.p, h->s_lnp.p, N);
    HIPCHK(h, hipGetLastError());
    { int rc = mark_launched(h, st); if (rc) return rc; }
    HIPCHK(h, hipMemcpy(lnprior_out, h->s_lnp.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int rx_model_flux_batch_device(rx_handle *h, int src, int N, const double *d_params, double *d_flux_out,
                               int32_t *d_status, int32_t *d_niter, void *stream)
{
    if (!h || N < 0 || (N > 0 && (!d_params || !d_flux_out))) return RX_E_ARG;
    int ncomp = 0;
    { int rc = source_ncomp(h, nullptr, N, src, &ncomp); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    RxKArgs a;
    fill_args(h, a, N, ncomp, RXK_MODE_FLUX);
    a.params = d_params; a.src_fixed = src;
    a.flux = d_flux_out; a.status = d_status; a.niter = d_niter;
    { int rc = ensure_comp_scratch(h, a, N, ncomp); if (rc) return rc; }
    return launch(h, a, (hipStream_t)stream);
}

int rx_model_flux_batch(rx_handle *h, int src, int N, const double *params, double *flux_out,
                        int32_t *status, int32_t *niter)
{
    if (!h || N < 0 || (N > 0 && (!params || !flux_out))) return RX_E_ARG;
    if (N == 0) return 0;
    int ncomp = 0;
    { int rc = source_ncomp(h, nullptr, N, src, &ncomp); if (rc) return rc; }
    const int nJ = h->h_srcs[src].nJ;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t np = (size_t)N * 4 * ncomp;
    HIPCHK(h, h->s_params.reserve(np));
    HIPCHK(h, h->s_flux.reserve((size_t)N * std::max(nJ, 1)));
    HIPCHK(h, h->s_status.reserve(N));
    HIPCHK(h, h->s_niter.reserve(N));
    HIPCHK(h, hipMemcpy(h->s_params.p, params, np * sizeof(double), hipMemcpyHostToDevice));
    int rc = rx_model_flux_batch_device(h, src, N, h->s_params.p, h->s_flux.p, h->s_status.p, h->s_niter.p, nullptr);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(flux_out, h->s_flux.p, (size_t)N * nJ * sizeof(double), hipMemcpyDeviceToHost));
    if (status) HIPCHK(h, hipMemcpy(status, h->s_status.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (niter) HIPCHK(h, hipMemcpy(niter, h->s_niter.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int rx_solve_batch(rx_handle *h, int src, int N, const double *tkin, const double *cdmol,
                   const double *dens, double *xpop, double *tex, double *tau, double *sb,
                   int32_t *status, int32_t *niter)
{
    if (!h || N < 0 || (N > 0 && (!tkin || !cdmol || !dens))) return RX_E_ARG;
    if (N == 0) return 0;
    if (src < 0 || src >= RX_MAX_SOURCES || !h->h_srcs[src].set) { h->err = "source slot not set"; return RX_E_STATE; }
    const int nlev = h->mol.nlev, nline = h->mol.nline, npart = (int)h->mol.parts.size();
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, h->s_in3.reserve((size_t)2 * N));
    HIPCHK(h, h->s_dens.reserve((size_t)N * npart));
    HIPCHK(h, h->s_xpop.reserve((size_t)N * nlev));
    HIPCHK(h, h->s_tex.reserve((size_t)N * nline));
    HIPCHK(h, h->s_tau.reserve((size_t)N * nline));
    HIPCHK(h, h->s_sb.reserve((size_t)N * nline));
    HIPCHK(h, h->s_status.reserve(N));
    HIPCHK(h, h->s_niter.reserve(N));
    HIPCHK(h, hipMemcpy(h->s_in3.p, tkin, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->s_in3.p + N, cdmol, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    // caller gives densities in file partner order; the kernel wants id order
    std::vector<double> d2((size_t)N * npart);
    for (int w = 0; w < N; ++w)
        for (int q = 0; q < npart; ++q) d2[(size_t)w * npart + q] = dens[(size_t)w * npart + h->order[q]];
    HIPCHK(h, hipMemcpy(h->s_dens.p, d2.data(), d2.size() * sizeof(double), hipMemcpyHostToDevice));
    RxKArgs a;
    fill_args(h, a, N, 1, RXK_MODE_SOLVE);
    a.src_fixed = src;
    a.tkin = h->s_in3.p; a.cdmol = h->s_in3.p + N; a.dens = h->s_dens.p;
    a.xpop = h->s_xpop.p; a.tex = h->s_tex.p; a.tau = h->s_tau.p; a.sb = h->s_sb.p;
    a.status = h->s_status.p; a.niter = h->s_niter.p;
#ifdef RX_STAMPS
    HIPCHK(h, h->s_cflux.reserve((size_t)N * 64));
    a.comp_flux = h->s_cflux.p;
#endif
    { int rc = launch(h, a, nullptr); if (rc) return rc; }
#ifdef RX_STAMPS
    if (getenv("RX_STAMP_FILE")) {
        std::vector<double> d((size_t)N * 64);
        HIPCHK(h, hipMemcpy(d.data(), h->s_cflux.p, d.size() * sizeof(double), hipMemcpyDeviceToHost));
        FILE *f = fopen(getenv("RX_STAMP_FILE"), "wb");
        if (f) { fwrite(d.data(), sizeof(double), d.size(), f); fclose(f); }
    }
#endif
    if (xpop) HIPCHK(h, hipMemcpy(xpop, h->s_xpop.p, (size_t)N * nlev * sizeof(double), hipMemcpyDeviceToHost));
    if (tex) HIPCHK(h, hipMemcpy(tex, h->s_tex.p, (size_t)N * nline * sizeof(double), hipMemcpyDeviceToHost));
    if (tau) HIPCHK(h, hipMemcpy(tau, h->s_tau.p, (size_t)N * nline * sizeof(double), hipMemcpyDeviceToHost));
    if (sb) HIPCHK(h, hipMemcpy(sb, h->s_sb.p, (size_t)N * nline * sizeof(double), hipMemcpyDeviceToHost));
    if (status) HIPCHK(h, hipMemcpy(status, h->s_status.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (niter) HIPCHK(h, hipMemcpy(niter, h->s_niter.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int rx_lubksb_pivots_batch(rx_handle *h, int N, int n, const double *A, double *x, int32_t *pivrow)
{
    if (!h || N < 0 || n < 2 || n > h->NL || (N > 0 && (!A || !x))) return RX_E_ARG;
    if (N == 0) return 0;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, h->s_params.reserve((size_t)N * n * n));
    HIPCHK(h, h->s_lnp.reserve((size_t)N * n));
    if (pivrow) HIPCHK(h, h->s_status.reserve((size_t)N * n));
    // (the staging buffers belong to the handle: behind whatever it launched last, like every entry point)
    { int rc = order_after_last(h, nullptr); if (rc) return rc; }
    HIPCHK(h, hipMemcpyAsync(h->s_params.p, A, (size_t)N * n * n * sizeof(double), hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(lukernel_for(h->NL), dim3(N), dim3(64), 0, nullptr, h->s_params.p, h->s_lnp.p,
                       pivrow ? h->s_status.p : nullptr, n, N);
    HIPCHK(h, hipGetLastError());
    { int rc = mark_launched(h, nullptr); if (rc) return rc; }
    HIPCHK(h, hipMemcpy(x, h->s_lnp.p, (size_t)N * n * sizeof(double), hipMemcpyDeviceToHost));
    if (pivrow) HIPCHK(h, hipMemcpy(pivrow, h->s_status.p, (size_t)N * n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int rx_escprob_batch(rx_handle *h, int method, int N, const double *tau, double *beta)
{
    if (!h || N < 0 || method < 0 || method > 3 || (N > 0 && (!tau || !beta))) return RX_E_ARG;
    if (N == 0) return 0;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, h->s_params.reserve((size_t)N));
    HIPCHK(h, h->s_lnp.reserve((size_t)N));
    { int rc = order_after_last(h, nullptr); if (rc) return rc; }
    HIPCHK(h, hipMemcpyAsync(h->s_params.p, tau, (size_t)N * sizeof(double), hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(rxk::rx_escprob_kernel, dim3((N + 63) / 64), dim3(64), 0, nullptr, h->s_params.p, h->s_lnp.p, method, N);
    HIPCHK(h, hipGetLastError());
    { int rc = mark_launched(h, nullptr); if (rc) return rc; }
    HIPCHK(h, hipMemcpy(beta, h->s_lnp.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int rx_lubksb_batch(rx_handle *h, int N, int n, const double *A, double *x)
{
    return rx_lubksb_pivots_batch(h, N, n, A, x, nullptr);
}

// ---- the stretch move on the device -------------------------------------------------
static int stretch_args(rx_handle *h, rxs::StretchArgs &A, int nens, int nwalkers, int ndim, double a,
                        uint64_t seed, int64_t step, int split)
{
    if (nens < 1 || nwalkers < 2 || (nwalkers & 1) || ndim < 1 || !(a > 1.0) || step < 0 || step > 0xffffffffLL ||
        (split != 0 && split != 1) || (long)nens * nwalkers > 0x7fffffffL) { h->err = "stretch move: bad argument"; return RX_E_ARG; }
    memset(&A, 0, sizeof A);
    A.nens = nens; A.nwalkers = nwalkers; A.ndim = ndim; A.split = split; A.a = a;
    A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32); A.step = (uint32_t)step;
    return 0;
}

int rx_stretch_propose_device(rx_handle *h, int nens, int nwalkers, int ndim, double a, uint64_t seed,
                              int64_t step, int split, const int32_t *d_ens_src, const double *d_coords,
                              double *d_q, double *d_factor, int32_t *d_widx, int32_t *d_qsrc, void *stream)
{
    if (!h || !d_coords || !d_q || !d_factor || !d_widx) return RX_E_ARG;
    rxs::StretchArgs A;
    { int rc = stretch_args(h, A, nens, nwalkers, ndim, a, seed, step, split); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    A.ens_src = d_ens_src; A.coords = const_cast<double *>(d_coords);
    A.q = d_q; A.factor = d_factor; A.widx = d_widx; A.qsrc = d_qsrc;
    const int n = nens * (nwalkers / 2), tb = 256;
    hipLaunchKernelGGL(rxs::rx_stretch_propose_kernel, dim3((n + tb - 1) / tb), dim3(tb), 0, (hipStream_t)stream, A);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int rx_stretch_accept_device(rx_handle *h, int nens, int nwalkers, int ndim, uint64_t seed, int64_t step,
                             int split, const double *d_q, const double *d_lnp_q, const double *d_factor,
                             const int32_t *d_widx, double *d_coords, double *d_lnp, int32_t *d_naccept,
                             void *stream)
{
    if (!h || !d_q || !d_lnp_q || !d_factor || !d_widx || !d_coords || !d_lnp) return RX_E_ARG;
    rxs::StretchArgs A;
    { int rc = stretch_args(h, A, nens, nwalkers, ndim, 2.0, seed, step, split); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    A.coords = d_coords; A.lnp = d_lnp; A.naccept = d_naccept;
    A.q = const_cast<double *>(d_q); A.factor = const_cast<double *>(d_factor);
    A.lnp_q = d_lnp_q; A.widx = const_cast<int32_t *>(d_widx);
    const int n = nens * (nwalkers / 2), tb = 256;
    hipLaunchKernelGGL(rxs::rx_stretch_accept_kernel, dim3((n + tb - 1) / tb), dim3(tb), 0, (hipStream_t)stream, A);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int rx_sampler_run_device(rx_handle *h, int nens, int nwalkers, int ncomp, double a, uint64_t seed,
                          int64_t step0, int nsteps, const int32_t *d_ens_src, double *d_coords,
                          double *d_lnp, int32_t *d_naccept, double *d_chain, double *d_chain_lnp,
                          double *solve_ms_out, void *stream)
{
    if (!h || nsteps < 0 || !d_coords || !d_lnp) return RX_E_ARG;
    const int ndim = 4 * ncomp;
    rxs::StretchArgs probe;
    { int rc = stretch_args(h, probe, nens, nwalkers, ndim, a, seed, step0, 0); if (rc) return rc; }
    if (step0 + nsteps > 0xffffffffLL) { h->err = "stretch move: step counter exceeds 32 bits"; return RX_E_ARG; }
    { int rc = device_batch_ncomp(h, nens * nwalkers, ncomp, d_ens_src); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t nq = (size_t)nens * (nwalkers / 2), N = (size_t)nens * nwalkers;
    HIPCHK(h, h->w_q.reserve(nq * ndim));
    HIPCHK(h, h->w_factor.reserve(nq));
    HIPCHK(h, h->w_lnpq.reserve(nq));
    HIPCHK(h, h->w_widx.reserve(nq));
    HIPCHK(h, h->w_qsrc.reserve(nq));
    HIPCHK(h, h->w_qstatus.reserve(nq));
    HIPCHK(h, h->w_qniter.reserve(nq));
    hipStream_t st = (hipStream_t)stream;
    { int rc = order_after_last(h, st); if (rc) return rc; }    // the work space belongs to the handle
    // optional: HIP events around every solve launch (one pair per half-step), summed after the last one
    std::vector<hipEvent_t> ev;
    if (solve_ms_out) {
        *solve_ms_out = 0.0;
        ev.resize((size_t)4 * nsteps, nullptr);
    }
    auto drop_events = [&]() { for (auto e : ev) if (e) (void)hipEventDestroy(e); };
    for (auto &e : ev) {
        hipError_t ee = hipEventCreate(&e);
        if (ee != hipSuccess) { drop_events(); return hip_fail(h, ee, "hipEventCreate"); }
    }
    for (int s = 0; s < nsteps; ++s) {
        for (int split = 0; split < 2; ++split) {
            int rc = rx_stretch_propose_device(h, nens, nwalkers, ndim, a, seed, step0 + s, split, d_ens_src, d_coords,
                                               h->w_q.p, h->w_factor.p, h->w_widx.p, d_ens_src ? h->w_qsrc.p : nullptr, st);
            if (rc) { drop_events(); return rc; }
            const size_t ei = (size_t)4 * s + 2 * split;
            rc = lnprob_device(h, (int)nq, ncomp, h->w_q.p, d_ens_src ? h->w_qsrc.p : nullptr, h->w_lnpq.p,
                               h->w_qstatus.p, h->w_qniter.p, st, solve_ms_out ? ev[ei] : nullptr,
                               solve_ms_out ? ev[ei + 1] : nullptr);
            if (rc) { drop_events(); return rc; }
            rc = rx_stretch_accept_device(h, nens, nwalkers, ndim, seed, step0 + s, split, h->w_q.p, h->w_lnpq.p,
                                          h->w_factor.p, h->w_widx.p, d_coords, d_lnp, d_naccept, st);
            if (rc) { drop_events(); return rc; }
        }
        hipError_t ce = hipSuccess;
        if (d_chain) ce = hipMemcpyAsync(d_chain + (size_t)s * N * ndim, d_coords, N * ndim * sizeof(double), hipMemcpyDeviceToDevice, st);
        if (ce == hipSuccess && d_chain_lnp) ce = hipMemcpyAsync(d_chain_lnp + (size_t)s * N, d_lnp, N * sizeof(double), hipMemcpyDeviceToDevice, st);
        if (ce != hipSuccess) { drop_events(); return hip_fail(h, ce, "rx_sampler_run_device: chain copy"); }
    }
    if (solve_ms_out && nsteps > 0) {
        hipError_t e = hipStreamSynchronize(st);
        double sum = 0.0;
        for (size_t i = 0; e == hipSuccess && i + 1 < ev.size(); i += 2) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            sum += ms;
        }
        drop_events();
        if (e != hipSuccess) return hip_fail(h, e, "rx_sampler_run_device: event timing");
        *solve_ms_out = sum;
    }
    return nsteps > 0 ? mark_launched(h, st) : 0;
}

int rx_sampler_run_async_device(rx_handle *h, int nens, int nwalkers, int ncomp, double a, uint64_t seed,
                                int64_t step0, int nsteps, const int32_t *d_ens_src, double *d_coords,
                                double *d_lnp, int32_t *d_naccept, double *d_chain, double *d_chain_lnp,
                                void *stream)
{
    if (!h || nsteps < 0 || !d_coords || !d_lnp) return RX_E_ARG;
    const int ndim = 4 * ncomp;
    rxs::AsyncArgs A;
    memset(&A, 0, sizeof A);
    { int rc = stretch_args(h, A.s, nens, nwalkers, ndim, a, seed, step0, 0); if (rc) return rc; }
    if (step0 + nsteps > 0xffffffffLL) { h->err = "stretch move: step counter exceeds 32 bits"; return RX_E_ARG; }
    { int rc = device_batch_ncomp(h, nens * nwalkers, ncomp, d_ens_src); if (rc) return rc; }
    const size_t N = (size_t)nens * nwalkers, nq = N / 2;
    if ((double)nsteps * 2.0 * (double)nq >= 4294967295.0) { h->err = "stretch move: more than 2^32 tasks in one launch"; return RX_E_ARG; }
    if (nsteps == 0) return 0;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    { int rc = order_after_last(h, st); if (rc) return rc; }
    constexpr int RING = 12;                         // versions of the positions kept readable (see the kernel)
    HIPCHK(h, h->w_version.reserve(N + RING));
    HIPCHK(h, h->w_hist.reserve((size_t)RING * N * ndim));
    HIPCHK(h, h->w_pend.reserve((size_t)rxs::PEND_SLOTS * N * ndim));
    HIPCHK(h, h->w_pendver.reserve((size_t)rxs::PEND_SLOTS * N));
    if (!h->d_abort) { HIPCHK(h, hipMalloc(&h->d_abort, sizeof(uint32_t))); HIPCHK(h, hipMemset(h->d_abort, 0, sizeof(uint32_t))); }
    if (!h->h_abort) { HIPCHK(h, hipHostMalloc(&h->h_abort, sizeof(uint32_t))); *h->h_abort = 0; }
    // few tasks per half-step: one wavefront per SIMD (lowest latency per task), the whole chip so that
    // wavefronts can run ahead of a slow task; many: two per SIMD (throughput)
    // (measured crossover, scripts/occ_crossover_sampler.py: 2048 walkers 0.77 ms per step with one wavefront
    // per SIMD against 0.86 with two, 4096 walkers 1.17 against 1.00)
    int occ = (nq > (size_t)3 * h->num_cu * RXK_WAVES_PER_BLOCK / 2 && h->blocks_per_cu2 >= 2) ? 2 : 1;
    if (h->force_occ == 1 || (h->force_occ == 2 && h->blocks_per_cu2 >= 2)) occ = h->force_occ;
    long blocks = (long)h->num_cu * (occ == 2 ? h->blocks_per_cu2 : 1);
    const long need = (long)((2 * nq * (size_t)nsteps + RXK_WAVES_PER_BLOCK - 1) / RXK_WAVES_PER_BLOCK);
    if (blocks > need) blocks = need;
    sampler_kernel_fn k = sampler_kernel_for(h->NL, occ, is_exact(h));
    if (!k) { h->err = "this build has no dataflow sampler kernel"; return RX_E_UNSUPP; }
    fill_args(h, A.k, (int)N, ncomp, RXK_MODE_LNPROB);
    A.s.ens_src = d_ens_src; A.s.coords = d_coords; A.s.lnp = d_lnp; A.s.naccept = d_naccept;
    A.nsteps = nsteps; A.ncomp = ncomp;
    A.version = h->w_version.p; A.abort_flag = h->d_abort;
    A.hist = h->w_hist.p; A.done = h->w_version.p + N; A.ring = RING;
    A.chain = d_chain; A.chain_lnp = d_chain_lnp;
    A.timeout_ticks = h->sampler_timeout_ticks;
    A.stall_ticks = h->sampler_stall_ticks;
    if (h->stats_on && blocks * RXK_WAVES_PER_BLOCK > rxs::RXS_STAT_SLOTS) {       // (one slot per wavefront: never silently off)
        h->err = "rx_sampler_stats: the grid has more wavefronts than counter slots"; return RX_E_UNSUPP;
    }
    A.stats = h->stats_on ? h->d_stats : nullptr;
    A.speculate = (occ == 1) && h->speculation != 0;        // (the head start exists in the one-wavefront-per-SIMD build only)
    if (A.speculate) { A.pend = h->w_pend.p; A.pend_version = h->w_pendver.p; }    // (proposals are published only for head starts)
    HIPCHK(h, hipMemsetAsync(h->d_queue, 0, sizeof(unsigned int), st));
    // (the abort word is STICKY: raised by a run, it stays up -- and ends every later run at its first wait --
    // until rx_sampler_wait has reported it; a second run enqueued before the wait cannot lose it)
    HIPCHK(h, hipMemsetAsync(h->w_version.p, 0, (N + RING) * sizeof(uint32_t), st));
    HIPCHK(h, hipMemsetAsync(h->w_pendver.p, 0, (size_t)rxs::PEND_SLOTS * N * sizeof(uint32_t), st));
    HIPCHK(h, hipMemcpyAsync(h->w_hist.p, d_coords, N * ndim * sizeof(double), hipMemcpyDeviceToDevice, st));   // version 0
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * RXK_WAVES_PER_BLOCK), 0, st, A);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(d_coords, h->w_hist.p + (size_t)(nsteps % RING) * N * ndim, N * ndim * sizeof(double),
                             hipMemcpyDeviceToDevice, st));                                                  // version nsteps
    HIPCHK(h, hipMemcpyAsync(h->h_abort, h->d_abort, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    return mark_launched(h, st);
}

// ---- the dataflow sampler across GPUs: replicas written by peers -------------
static constexpr int PEER_RING = 12;

static int peer_launch_shape(rx_handle *h, size_t tasks_per_half_step, size_t total_tasks, int *occ_out, long *blocks_out)
{
    int occ = (tasks_per_half_step > (size_t)3 * h->num_cu * RXK_WAVES_PER_BLOCK / 2 && h->blocks_per_cu2 >= 2) ? 2 : 1;
    if (h->force_occ == 1 || (h->force_occ == 2 && h->blocks_per_cu2 >= 2)) occ = h->force_occ;
    long blocks = (long)h->num_cu * (occ == 2 ? h->blocks_per_cu2 : 1);
    const long need = (long)((total_tasks + RXK_WAVES_PER_BLOCK - 1) / RXK_WAVES_PER_BLOCK);
    if (blocks > need) blocks = need;
    // (the limit is in compute units: the two-waves-per-SIMD build places blocks_per_cu2 workgroups on each)
    // ranks whose replicas live on ONE device share its compute units (every grid must be resident while the others run:
    // a task may wait for a task of another rank): an explicit rx_set_sampler_grid_limit wins, otherwise an equal split
    int cus = h->sampler_grid_limit;
    if (cus <= 0 && h->peer.same_device > 1) cus = std::max(1, h->num_cu / h->peer.same_device);
    const long lim = (long)cus * (occ == 2 ? h->blocks_per_cu2 : 1);
    if (cus > 0 && blocks > lim) blocks = lim;
    if (blocks < 1) blocks = 1;
    *occ_out = occ; *blocks_out = blocks;
    return 0;
}

int rx_set_sampler_grid_limit(rx_handle *h, int blocks)
{
    if (!h || blocks < 0) return RX_E_ARG;
    h->sampler_grid_limit = blocks;
    return 0;
}

int rx_sampler_peer_disconnect(rx_handle *h)
{
    if (!h) return RX_E_ARG;
    rx_handle::Peer &P = h->peer;
    if (!P.own && !P.d_bases) return 0;
    (void)hipSetDevice(h->device);
    if (h->in_flight && h->ev_done) { (void)hipEventSynchronize(h->ev_done); h->in_flight = false; }
    for (int r = 0; r < RX_MAX_RANKS; ++r)
        if (P.opened[r] && P.base[r]) { (void)hipIpcCloseMemHandle(P.base[r]); P.opened[r] = false; P.base[r] = nullptr; }
    P.connected = false; P.begun = false;
    return 0;
}

int rx_sampler_peer_close(rx_handle *h)
{
    if (!h) return RX_E_ARG;
    rx_handle::Peer &P = h->peer;
    if (!P.own && !P.d_bases) return 0;
    (void)hipSetDevice(h->device);
    if (h->in_flight && h->ev_done) { (void)hipEventSynchronize(h->ev_done); h->in_flight = false; }
    for (int r = 0; r < RX_MAX_RANKS; ++r)
        if (P.opened[r] && P.base[r]) (void)hipIpcCloseMemHandle(P.base[r]);
    if (P.d_bases) (void)hipFree(P.d_bases);
    if (P.d_touch) (void)hipFree(P.d_touch);
    if (P.own) (void)hipFree(P.own);
    P = rx_handle::Peer();
    return 0;
}

int rx_sampler_peer_setup(rx_handle *h, int nranks, int rank, int nens, int nwalkers, int ncomp, void *ipc_handle_out)
{
    if (!h || nranks < 1 || nranks > RX_MAX_RANKS || rank < 0 || rank >= nranks) return RX_E_ARG;
    rxs::StretchArgs probe;
    { int rc = stretch_args(h, probe, nens, nwalkers, 4 * ncomp, 2.0, 0, 0, 0); if (rc) return rc; }
    if (ncomp != 1 && ncomp != 2) { h->err = "ncomp must be 1 or 2"; return RX_E_ARG; }
    { const char *off = getenv("RX_NO_PEER"); if (off && *off && *off != '0') { h->err = "peer replicas switched off (RX_NO_PEER)"; return RX_E_UNSUPP; } }
    (void)rx_sampler_peer_close(h);
    HIPCHK(h, hipSetDevice(h->device));
    rx_handle::Peer &P = h->peer;
    P.nranks = nranks; P.rank = rank; P.nens = nens; P.nwalkers = nwalkers; P.ncomp = ncomp;
    P.N = (size_t)nens * nwalkers;
    const size_t N = P.N, ndim = 4 * (size_t)ncomp;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t off = 0;
    P.off_version = off; off = up(off + N * sizeof(uint32_t));
    P.off_done = off;    off = up(off + PEER_RING * sizeof(uint32_t));
    P.off_abort = off;   off = up(off + sizeof(uint32_t));
    P.off_alive = off;   off = up(off + sizeof(uint32_t));                                   // (zeroed with the counters by rx_sampler_peer_begin)
    P.off_pendver = off; off = up(off + (size_t)rxs::PEND_SLOTS * N * sizeof(uint32_t));     // (zeroed with the counters by rx_sampler_peer_begin)
    P.off_nacc = off;    off = up(off + N * sizeof(int32_t));
    P.off_lnp = off;     off = up(off + N * sizeof(double));
    P.off_hist = off;    off = up(off + (size_t)PEER_RING * N * ndim * sizeof(double));
    P.off_pend = off;    off = up(off + (size_t)rxs::PEND_SLOTS * N * ndim * sizeof(double));
    P.bytes = off;
    // fine-grained device memory: writes of a peer GPU become visible to this GPU's system-scope loads while
    // both kernels run (coarse-grained allocations are only coherent at kernel boundaries)
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, P.bytes, hipDeviceMallocFinegrained);
    P.memkind = 1;
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, P.bytes, hipDeviceMallocUncached); P.memkind = 2; }
    if (e != hipSuccess) { (void)hipGetLastError(); P = rx_handle::Peer(); return hip_fail(h, e, "rx_sampler_peer_setup: hipExtMallocWithFlags(fine-grained)"); }
    P.own = (char *)p;
    HIPCHK(h, hipMemset(P.own, 0, P.bytes));
    if (ipc_handle_out) {
        hipIpcMemHandle_t ih;
        e = hipIpcGetMemHandle(&ih, P.own);
        if (e != hipSuccess) { (void)hipGetLastError(); (void)rx_sampler_peer_close(h); return hip_fail(h, e, "rx_sampler_peer_setup: hipIpcGetMemHandle"); }
        static_assert(sizeof ih == RX_IPC_HANDLE_BYTES, "ipc handle size");
        memcpy(ipc_handle_out, &ih, sizeof ih);
    }
    return 0;
}

void *rx_sampler_peer_base(rx_handle *h) { return h ? (void *)h->peer.own : nullptr; }

int rx_device_bus_id(rx_handle *h, char *out)
{
    if (!h || !out) return RX_E_ARG;
    memset(out, 0, RX_BUS_ID_BYTES);
    HIPCHK(h, hipDeviceGetPCIBusId(out, RX_BUS_ID_BYTES - 1, h->device));
    return 0;
}

int rx_sampler_peer_set_bus_ids(rx_handle *h, const char *bus_ids)
{
    if (!h) return RX_E_ARG;
    rx_handle::Peer &P = h->peer;
    P.bus_ids.clear();
    if (!bus_ids) return 0;
    if (!P.own) { h->err = "rx_sampler_peer_set_bus_ids: call rx_sampler_peer_setup first"; return RX_E_STATE; }
    for (int r = 0; r < P.nranks; ++r) {
        const char *b = bus_ids + (size_t)r * RX_BUS_ID_BYTES;
        P.bus_ids.emplace_back(b, strnlen(b, RX_BUS_ID_BYTES));
        if (P.bus_ids.back().empty()) { P.bus_ids.clear(); h->err = "rx_sampler_peer_set_bus_ids: empty id"; return RX_E_ARG; }
    }
    return 0;
}

int rx_sampler_peer_connect(rx_handle *h, const void *ipc_handles, void *const *bases)
{
    if (!h || (!ipc_handles && !bases)) return RX_E_ARG;
    rx_handle::Peer &P = h->peer;
    if (!P.own) { h->err = "rx_sampler_peer_connect: call rx_sampler_peer_setup first"; return RX_E_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    for (int r = 0; r < P.nranks; ++r) {
        if (r == P.rank) { P.base[r] = P.own; continue; }
        if (bases) { P.base[r] = (char *)bases[r]; continue; }
        hipIpcMemHandle_t ih;
        memcpy(&ih, (const char *)ipc_handles + (size_t)r * RX_IPC_HANDLE_BYTES, sizeof ih);
        void *q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, ih, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); return hip_fail(h, e, "rx_sampler_peer_connect: hipIpcOpenMemHandle"); }
        P.base[r] = (char *)q; P.opened[r] = true;
    }
    for (int r = 0; r < P.nranks; ++r)
        if (!P.base[r]) { h->err = "rx_sampler_peer_connect: a peer's replica is missing"; return RX_E_ARG; }
    // the per-device registry of this run: which replicas live on THIS device (two handles of one process, ranks of a
    // rehearsal squeezed onto one GPU).  A mapping whose device cannot be told counts as remote.
    // Every peer block must be reachable from THIS device: a task stores its result into all replicas, and a store into memory
    // the device cannot reach faults and takes the process down.  The library asks itself (the caller's own check needs every
    // rank to see every device, which one-GPU-per-process launches do not): the device a mapping lives on, then
    // hipDeviceCanAccessPeer.  A mapping whose device cannot be told is treated as unreachable -> RX_E_UNSUPP, and the
    // caller falls back to half-steps + all-gather on all ranks.
    P.same_device = 0;
    const bool by_id = (int)P.bus_ids.size() == P.nranks;
    for (int r = 0; r < P.nranks; ++r) {
        hipPointerAttribute_t at;
        if (r == P.rank) { ++P.same_device; continue; }
        if (by_id) {
            // the ranks' own word for their GPUs (PCI bus ids): what an IPC mapping's pointer attributes say is not relied on
            if (P.bus_ids[r] == P.bus_ids[P.rank]) { ++P.same_device; continue; }
            int dev = -1, can = 0;
            if (hipDeviceGetByPCIBusId(&dev, P.bus_ids[r].c_str()) != hipSuccess) {
                // the peer's GPU is not visible to this process (one GPU per process): nothing to ask hipDeviceCanAccessPeer about;
                // the mapping was opened with lazy peer access and is READ below -- an unreachable block fails there, in an API call
                (void)hipGetLastError();
                continue;
            }
            if (hipDeviceCanAccessPeer(&can, h->device, dev) != hipSuccess || !can) {
                (void)hipGetLastError();
                h->err = "rx_sampler_peer_connect: this device has no peer access to the device of rank " + std::to_string(r) + " (" + P.bus_ids[r] + ")";
                return RX_E_UNSUPP;
            }
            continue;
        }
        if (hipPointerGetAttributes(&at, P.base[r]) != hipSuccess) {
            (void)hipGetLastError();
            h->err = "rx_sampler_peer_connect: cannot tell which device a peer's replica lives on";
            return RX_E_UNSUPP;
        }
        if (at.device == h->device) { ++P.same_device; continue; }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, h->device, at.device) != hipSuccess || !can) {
            (void)hipGetLastError();
            h->err = "rx_sampler_peer_connect: this device has no peer access to the device of rank " + std::to_string(r);
            return RX_E_UNSUPP;
        }
    }
    if (!P.d_bases) HIPCHK(h, hipMalloc(&P.d_bases, RX_MAX_RANKS * sizeof(char *)));
    HIPCHK(h, hipMemcpy(P.d_bases, P.base, RX_MAX_RANKS * sizeof(char *), hipMemcpyHostToDevice));
    // The mappings were opened with hipIpcMemLazyEnablePeerAccess: whatever the first access from this device sets up is set
    // up HERE, by reading one word of every peer's block (a read: harmless whatever the peer is doing), not inside the first
    // launch under the no-progress watchdog
    // (into a word of its own: d_queue is the live task counter of whatever this handle may still have running)
    if (!P.d_touch) HIPCHK(h, hipMalloc(&P.d_touch, sizeof(uint32_t)));
    for (int r = 0; r < P.nranks; ++r)
        if (r != P.rank) HIPCHK(h, hipMemcpy(P.d_touch, P.base[r] + P.off_abort, sizeof(uint32_t), hipMemcpyDeviceToDevice));
    P.connected = true;
    return 0;
}

int rx_sampler_peer_begin(rx_handle *h, const double *d_coords, const double *d_lnp, const int32_t *d_naccept, void *stream)
{
    if (!h || !d_coords || !d_lnp) return RX_E_ARG;
    rx_handle::Peer &P = h->peer;
    if (!P.connected) { h->err = "rx_sampler_peer_begin: not connected"; return RX_E_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    { int rc = order_after_last(h, st); if (rc) return rc; }
    const size_t N = P.N, ndim = 4 * (size_t)P.ncomp;
    // version counters, per-step counters, abort word, acceptance counts: zero ... then the state the run starts from
    HIPCHK(h, hipMemsetAsync(P.own, 0, (size_t)P.off_lnp, st));
    if (d_naccept) HIPCHK(h, hipMemcpyAsync(P.own + P.off_nacc, d_naccept, N * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(P.own + P.off_lnp, d_lnp, N * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(P.own + P.off_hist, d_coords, N * ndim * sizeof(double), hipMemcpyDeviceToDevice, st));   // version 0
    HIPCHK(h, hipStreamSynchronize(st));          // the caller's barrier across ranks follows: the replica must BE ready
    P.begun = true;
    return 0;
}

int rx_sampler_peer_run(rx_handle *h, double a, uint64_t seed, int64_t step0, int nsteps, const int32_t *d_ens_src,
                        double *d_chain, double *d_chain_lnp, void *stream)
{
    if (!h || nsteps < 0) return RX_E_ARG;
    rx_handle::Peer &P = h->peer;
    if (!P.connected || !P.begun) { h->err = "rx_sampler_peer_run: rx_sampler_peer_begin (and the barrier behind it) first"; return RX_E_STATE; }
    const int ndim = 4 * P.ncomp;
    rxs::AsyncArgs A;
    memset(&A, 0, sizeof A);
    { int rc = stretch_args(h, A.s, P.nens, P.nwalkers, ndim, a, seed, step0, 0); if (rc) return rc; }
    if (step0 + nsteps > 0xffffffffLL) { h->err = "stretch move: step counter exceeds 32 bits"; return RX_E_ARG; }
    { int rc = device_batch_ncomp(h, (int)P.N, P.ncomp, d_ens_src); if (rc) return rc; }
    const size_t N = P.N, nq = N / 2;
    // contiguous blocks of ceil(nq / nranks) proposals of every half-step, one per rank (sampler.block_partition)
    const size_t per = (nq + P.nranks - 1) / P.nranks;
    const size_t lo = std::min((size_t)P.rank * per, nq), hi = std::min(lo + per, nq);
    if ((double)nsteps * 2.0 * (double)(hi - lo) >= 4294967295.0) { h->err = "stretch move: more than 2^32 tasks in one launch"; return RX_E_ARG; }
    P.begun = false;
    P.nsteps_run = nsteps;
    if (nsteps == 0) return 0;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    { int rc = order_after_last(h, st); if (rc) return rc; }
    if (!h->h_abort) { HIPCHK(h, hipHostMalloc(&h->h_abort, sizeof(uint32_t))); *h->h_abort = 0; }
    int occ; long blocks;
    peer_launch_shape(h, hi - lo, 2 * (hi - lo) * (size_t)nsteps, &occ, &blocks);
    sampler_kernel_fn k = sampler_kernel_for(h->NL, occ, is_exact(h));
    if (!k) { h->err = "this build has no dataflow sampler kernel"; return RX_E_UNSUPP; }
    fill_args(h, A.k, (int)N, P.ncomp, RXK_MODE_LNPROB);
    A.s.ens_src = d_ens_src;
    A.s.coords = nullptr;
    A.s.lnp = (double *)(P.own + P.off_lnp);
    A.s.naccept = (int32_t *)(P.own + P.off_nacc);
    A.nsteps = nsteps; A.ncomp = P.ncomp;
    A.version = (uint32_t *)(P.own + P.off_version);
    A.done = (uint32_t *)(P.own + P.off_done);
    A.abort_flag = (uint32_t *)(P.own + P.off_abort);
    A.hist = (double *)(P.own + P.off_hist);
    A.ring = PEER_RING;
    A.chain = d_chain; A.chain_lnp = d_chain_lnp;
    A.timeout_ticks = h->sampler_timeout_ticks;
    A.stall_ticks = h->sampler_stall_ticks;
    A.alive = (uint32_t *)(P.own + P.off_alive);
    if (h->stats_on && blocks * RXK_WAVES_PER_BLOCK > rxs::RXS_STAT_SLOTS) {       // (one slot per wavefront: never silently off)
        h->err = "rx_sampler_stats: the grid has more wavefronts than counter slots"; return RX_E_UNSUPP;
    }
    A.stats = h->stats_on ? h->d_stats : nullptr;
    A.speculate = (occ == 1) && h->speculation != 0;
    if (A.speculate) { A.pend = (double *)(P.own + P.off_pend); A.pend_version = (uint32_t *)(P.own + P.off_pendver); }
    // (nranks = 1 runs the very same kernel in its one-GPU form on the replica block)
    A.nranks = P.nranks; A.rank = P.rank; A.t_lo = (uint32_t)lo; A.t_n = (uint32_t)(hi - lo);
    A.peers = P.d_bases;
    A.off_version = P.off_version; A.off_done = P.off_done; A.off_abort = P.off_abort; A.off_alive = P.off_alive;
    A.off_lnp = P.off_lnp; A.off_nacc = P.off_nacc; A.off_hist = P.off_hist; A.off_pend = P.off_pend; A.off_pendver = P.off_pendver;
    HIPCHK(h, hipMemsetAsync(h->d_queue, 0, sizeof(unsigned int), st));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * RXK_WAVES_PER_BLOCK), 0, st, A);
    HIPCHK(h, hipGetLastError());
    return mark_launched(h, st);
}

int rx_sampler_peer_finish(rx_handle *h, double *d_coords, double *d_lnp, int32_t *d_naccept, void *stream)
{
    if (!h || !d_coords || !d_lnp) return RX_E_ARG;
    rx_handle::Peer &P = h->peer;
    if (!P.connected) { h->err = "rx_sampler_peer_finish: not connected"; return RX_E_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    { int rc = order_after_last(h, st); if (rc) return rc; }
    const size_t N = P.N, ndim = 4 * (size_t)P.ncomp;
    if (!h->h_abort) { HIPCHK(h, hipHostMalloc(&h->h_abort, sizeof(uint32_t))); *h->h_abort = 0; }
    HIPCHK(h, hipMemcpyAsync(h->h_abort, P.own + P.off_abort, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(d_coords, P.own + P.off_hist + (size_t)(P.nsteps_run % PEER_RING) * N * ndim * sizeof(double),
                             N * ndim * sizeof(double), hipMemcpyDeviceToDevice, st));          // version nsteps
    HIPCHK(h, hipMemcpyAsync(d_lnp, P.own + P.off_lnp, N * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (d_naccept) HIPCHK(h, hipMemcpyAsync(d_naccept, P.own + P.off_nacc, N * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (*h->h_abort) {
        *h->h_abort = 0;
        h->err = "dataflow sampler (multi-GPU): a task on some rank waited longer than the timeout for its inputs (chain incomplete)";
        return RX_E_TIMEOUT;
    }
    return 0;
}

int rx_sampler_stats(rx_handle *h, int enable, uint64_t *out6)
{
    if (!h) return RX_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t nst = (size_t)8 * rxs::RXS_STAT_SLOTS;          // one block of 8 counters per wavefront of the grid
    if (!h->d_stats) { HIPCHK(h, hipMalloc(&h->d_stats, nst * sizeof(unsigned long long))); HIPCHK(h, hipMemset(h->d_stats, 0, nst * sizeof(unsigned long long))); }
    if (h->in_flight) { HIPCHK(h, hipEventSynchronize(h->ev_done)); h->in_flight = false; }
    if (out6) {
        std::vector<unsigned long long> all(nst);
        HIPCHK(h, hipMemcpy(all.data(), h->d_stats, nst * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long v[8] = {};
        for (size_t w = 0; w < (size_t)rxs::RXS_STAT_SLOTS; ++w)
            for (int i = 0; i < 8; ++i) v[i] += all[w * 8 + i];
        for (int i = 0; i < 6; ++i) out6[i] = v[i];
        h->last_spec[0] = v[6]; h->last_spec[1] = v[7];
    }
    HIPCHK(h, hipMemset(h->d_stats, 0, nst * sizeof(unsigned long long)));
    h->stats_on = enable ? 1 : 0;
    return 0;
}

int rx_sampler_spec_stats(rx_handle *h, uint64_t *out2)
{
    if (!h || !out2) return RX_E_ARG;
    out2[0] = h->last_spec[0]; out2[1] = h->last_spec[1];
    return 0;
}

int rx_set_sampler_speculation(rx_handle *h, int mode)
{
    if (!h || mode < -1 || mode > 1) return RX_E_ARG;
    h->speculation = mode;
    return 0;
}

int rx_set_sampler_timeout_ms(rx_handle *h, double ms)
{
    if (!h || !(ms >= 0.0) || ms > 3.6e6) return RX_E_ARG;
    h->sampler_timeout_ticks = (long long)(ms * 1e5);          // wall_clock64 ticks at 100 MHz
    return 0;
}

int rx_set_sampler_stall_ms(rx_handle *h, double ms)
{
    if (!h || !(ms >= 0.0) || ms > 3.6e6) return RX_E_ARG;
    h->sampler_stall_ticks = (long long)(ms * 1e5);
    return 0;
}

int rx_sampler_peer_same_device(const rx_handle *h) { return h ? h->peer.same_device : RX_E_ARG; }

int rx_sampler_peer_abort(rx_handle *h)
{
    if (!h) return RX_E_ARG;
    rx_handle::Peer &P = h->peer;
    if (!P.connected) { h->err = "rx_sampler_peer_abort: not connected"; return RX_E_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    // a stream of its own: the legacy default stream would wait for the very kernel this is meant to end
    if (!h->ctl_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->ctl_stream, hipStreamNonBlocking));
    static const uint32_t one = 1u;
    for (int r = 0; r < P.nranks; ++r)
        if (P.base[r]) HIPCHK(h, hipMemcpyAsync(P.base[r] + P.off_abort, &one, sizeof one, hipMemcpyHostToDevice, h->ctl_stream));
    HIPCHK(h, hipStreamSynchronize(h->ctl_stream));
    return 0;
}

int rx_peer_topology(int dev_a, int dev_b, int32_t *out3)
{
    if (!out3 || dev_a < 0 || dev_b < 0) return RX_E_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || dev_a >= n || dev_b >= n) { (void)hipGetLastError(); return RX_E_NODEVICE; }
    out3[0] = 1; out3[1] = -1; out3[2] = 0;
    if (dev_a == dev_b) return 0;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dev_a, dev_b) != hipSuccess) { (void)hipGetLastError(); can = -1; }
    out3[0] = can;
    uint32_t type = 0, hops = 0;
    if (hipExtGetLinkTypeAndHopCount(dev_a, dev_b, &type, &hops) == hipSuccess) { out3[1] = (int32_t)type; out3[2] = (int32_t)hops; }
    else { (void)hipGetLastError(); out3[1] = -1; out3[2] = -1; }
    return 0;
}

int rx_sampler_wait(rx_handle *h, void *stream)
{
    if (!h) return RX_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    if (h->h_abort && *h->h_abort) {
        *h->h_abort = 0;
        if (h->d_abort) HIPCHK(h, hipMemset(h->d_abort, 0, sizeof(uint32_t)));      // reported: lowered again
        h->err = "dataflow sampler: a task waited longer than the timeout for its inputs (chain incomplete)";
        return RX_E_TIMEOUT;
    }
    return 0;
}

int rx_time_lnprob_device(rx_handle *h, int N, int ncomp, const double *d_params, const int32_t *d_src_index,
                          double *d_lnp, int32_t *d_status, int32_t *d_niter, void *stream,
                          int reps, double *ms_mean_out)
{
    if (!h || reps < 1 || !ms_mean_out || N < 0 || (N > 0 && (!d_params || !d_lnp))) return RX_E_ARG;
    *ms_mean_out = 0.0;
    if (N == 0) return 0;                        // nothing is launched, no event is recorded
    { int rc = device_batch_ncomp(h, N, ncomp, d_src_index); if (rc) return rc; }
    HIPCHK(h, hipSetDevice(h->device));
    RxKArgs a;
    fill_args(h, a, N, ncomp, RXK_MODE_LNPROB);
    a.params = d_params; a.src_index = d_src_index; a.src_fixed = 0;
    a.lnp = d_lnp; a.status = d_status; a.niter = d_niter;
    { int rc = ensure_comp_scratch(h, a, N, ncomp); if (rc) return rc; }
    double sum = 0.0;
    for (int r = 0; r < reps; ++r) {
        int rc = launch(h, a, (hipStream_t)stream, h->ev0, h->ev1);
        if (rc) return rc;
        HIPCHK(h, hipEventSynchronize(h->ev1));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        sum += ms;
    }
    *ms_mean_out = sum / reps;
    return 0;
}

}  // extern "C"
