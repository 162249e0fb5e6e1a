/*
 * radex_emcee_amd.h -- C ABI of the MI355X-native batched RADEX LVG
 * likelihood engine (libradex_emcee_amd.so, HIP / gfx950).
 *
 * This is the drop-in boundary for ONE hot path of yangcht/radex_emcee: the
 * per-walker RADEX escape-probability solve inside lnlike/model_lvg.  Each
 * entry point names the reference interface it replaces (paths relative to
 * /root/reference).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - all floating point is IEEE double (the reference is all-double);
 *   - "params" is row-major [N][ndim], ndim = 4*ncomp, log10 of
 *     (n_H2 [cm^-3], T_kin [K], N_CO/dv [cm^-2 (km/s)^-1], size [sr]) per
 *     component, exactly the vector p that emcee hands to lnprob
 *     (emcee/emcee_radex.py:120-121, emcee/emcee_radex_2comp.py:122-125);
 *   - functions return 0 on success, a negative RX_E_* code on failure and
 *     never throw; rx_last_error() gives the message;
 *   - there is NO CPU fallback: every compute entry point needs a HIP device
 *     and fails with RX_E_NODEVICE otherwise;
 *   - a handle is single-caller (one host thread at a time); use one handle per GPU / per
 *     thread.  The *_device entry points are asynchronous on the stream they are given; the
 *     handle owns its work queue and scratch, so its launches never overlap: a launch on
 *     another stream than the previous one is ordered behind it with an event
 *     (hipStreamWaitEvent), and rx_set_source waits for whatever is still running.  Launches
 *     of DIFFERENT handles on different streams do overlap.
 */
#ifndef RADEX_EMCEE_AMD_H
#define RADEX_EMCEE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RX_ABI_VERSION 7
#define RX_MAX_SOURCES 64      /* sources resident in one handle (config 3: 16) */
#define RX_MAX_NJ      32      /* observed lines per source                     */
#define RX_MAX_LEVELS  64      /* one level per lane of a 64-wide wavefront     */
#define RX_MAX_LINES   64
#define RX_MAX_RANKS    8      /* GPUs of one node that one sampled ensemble may span  */
#define RX_IPC_HANDLE_BYTES 64 /* sizeof(hipIpcMemHandle_t)                            */

/* per-walker status (int32), mirrors the reference's error behaviour:
 *   OK       converged (Fortran conv flag or the python delta-pop test)
 *   MAXITER  stopped at maxiter=200 unconverged; result used silently
 *            (emcee/pyradex/core.py:904-907)
 *   INVALID  the reference would raise ValueError (core.py:734-735,771-772)
 *            or produce non-finite model/residuals -> lnlike = -inf
 *            (emcee/emcee_radex.py:134-161)
 *   PRIOR    lnprior = -inf, solver not run (emcee/emcee_radex.py:179-180)   */
enum { RX_OK = 0, RX_MAXITER = 1, RX_INVALID = 2, RX_PRIOR = 3 };

enum {
    RX_E_ARG      = -1,   /* bad argument                                   */
    RX_E_IO       = -2,   /* molecular data file missing / malformed        */
    RX_E_UNSUPP   = -3,   /* molecule outside the kernel's limits           */
    RX_E_NODEVICE = -4,   /* no usable HIP device                           */
    RX_E_HIP      = -5,   /* HIP runtime error                              */
    RX_E_STATE    = -6,   /* source not set, etc.                           */
    RX_E_TIMEOUT  = -7    /* dataflow sampler: a task gave up waiting for its inputs; the chain of that run
                             is incomplete (rx_sampler_wait)                 */
};

typedef struct rx_handle rx_handle;

/* Replaces the pyradex.Radex(...) constructor as called by init_radex()
 * (emcee/emcee_radex.py:104-117, emcee/pyradex/core.py:209-386):
 *   lamda_path  = datapath + species + ".dat" (LAMDA format, readdata_)
 *   method      = escapeProbGeom: 1 sphere, 2 lvg, 3 slab (core.py:690-700)
 *   deltav_kms  = deltav (core.py:447-454)
 *   device      = HIP device ordinal (>= 0)
 * Returns NULL on failure with a message in err.                            */
rx_handle *rx_create(const char *lamda_path, int method, double deltav_kms,
                     int device, char *err, size_t errlen);
void rx_destroy(rx_handle *h);
const char *rx_last_error(const rx_handle *h);
int rx_abi_version(void);

/* Molecule introspection (Radex.level_population.size etc.).               */
int rx_nlev(const rx_handle *h);
int rx_nline(const rx_handle *h);
int rx_npart(const rx_handle *h);
/* collision partner ids in file order: 1=H2 2=pH2 3=oH2 4=e 5=H 6=He 7=H+
 * (core.py:476-482).  out[rx_npart].                                        */
int rx_partner_ids(const rx_handle *h, int32_t *out);
/* line data, out[rx_nline]: xnu [cm^-1] = E_up - E_low (core.py:1024-1029),
 * spfreq [GHz] (Radex.frequency), 1-based upper/lower level indices.        */
int rx_line_data(const rx_handle *h, double *xnu, double *spfreq,
                 int32_t *iupp, int32_t *ilow);
/* The background of a source slot as the kernels hold it -- backrad_'s outputs [radex.so@0x1be30, via the
 * `tbg` setter emcee/pyradex/core.py:845-854; SURVEY A.1]: backi[rx_nline] = thc xnu^3 / (exp(fk xnu / tbg) - 1)
 * (1e-30f where fk xnu / tbg >= 160), which is also totalb; trj = tbg for every line (*tbg_out).  The table is
 * copied back FROM THE DEVICE (what rx_set_source uploaded), not recomputed.  RX_E_STATE when the slot is not set. */
int rx_background(rx_handle *h, int src, double *backi, double *tbg_out);

/* Ortho fraction used to split n_H2 into oH2/pH2 densities; default 0.75 =
 * opr/(1+opr) with opr = 3 (emcee/emcee_radex.py:95-96).                    */
int rx_set_fortho(rx_handle *h, double fortho);
/* Iteration limits of Radex.run_radex (core.py:460-463): default 10 / 200.  */
int rx_set_iteration_limits(rx_handle *h, int miniter, int maxiter);

/* Replaces R.set_params(tbg=...) -> backrad_ (emcee/emcee_radex.py:419,
 * core.py:845-854) together with the per-source closure arguments of lnprob
 * (args=(Jup, flux, eflux), kwargs={'bounds', 'T_d'}:
 * emcee/emcee_radex.py:483-488, emcee/emcee_radex_2comp.py:557-563).
 *   src     slot 0..RX_MAX_SOURCES-1
 *   Jup     upper-level J of each observed line (model index = Jup-1)
 *   bounds  [4*ncomp][2] (lo, hi)
 *   T_d     dust temperature for the 2-component Gaussian prior; NaN = None  */
int rx_set_source(rx_handle *h, int src, double tbg, int nJ, const int32_t *Jup,
                  const double *flux, const double *eflux, const double *bounds,
                  int ncomp, double T_d);

/* Replaces lnlike(p, Jup, flux, eflux, R) (emcee/emcee_radex.py:132-167,
 * emcee/emcee_radex_2comp.py:169-196) on its own: with the prior of slot `src` disabled,
 * rx_lnprob_batch* return the log-likelihood alone (lnprior is neither added nor allowed to
 * short-circuit); enabled = 1 restores lnprob.  rx_set_source re-enables the prior.          */
int rx_set_source_prior(rx_handle *h, int src, int enabled);

/* Replaces lnprior(p, bounds[, T_d]) on its own (emcee/emcee_radex.py:169-175,
 * emcee/emcee_radex_2comp.py:199-234): the box, the ordering constraints and, for two components,
 * the flat/Gaussian terms of slot `src` (bounds, ncomp and T_d as given to rx_set_source) for N
 * host-side parameter vectors [N][4*ncomp]; no solve is run.  The same device function as the
 * fused prior of rx_lnprob_batch*.                                                            */
int rx_lnprior_batch(rx_handle *h, int src, int N, const double *params, double *lnprior_out);

/* Replaces lnprob(p, Jup, flux, eflux, bounds[, T_d]) called once per walker
 * through pool.map (emcee/emcee_radex.py:177-181,
 * emcee/emcee_radex_2comp.py:237-244) by ONE launch over N walkers.
 *   src_index  per-walker source slot, or NULL = slot 0 for every walker;
 *              all referenced sources must have the same ncomp
 *   lnp        [N] log-probability, -inf allowed
 *   status     [N] RX_* status, may be NULL
 *   niter      [N] iterations (_iter_counter summed over components), may be NULL
 * Host-pointer form: copies in/out through handle-owned staging buffers.    */
int rx_lnprob_batch(rx_handle *h, int N, const double *params,
                    const int32_t *src_index, double *lnp, int32_t *status,
                    int32_t *niter);
/* Device-pointer form: all pointers are HIP device pointers on the handle's
 * device, `stream` is a hipStream_t (NULL = default stream); asynchronous.
 *   ncomp        components per walker (params is [N][4*ncomp]); stated by the caller because
 *                the host cannot see a device-side index.  With d_src_index == NULL slot 0 must
 *                be set with that ncomp (checked, RX_E_STATE / RX_E_ARG).
 *   d_src_index  per-walker slot; a walker whose slot is outside 0..RX_MAX_SOURCES-1, was
 *                never set, or holds a source of another ncomp gets lnp = -inf and status
 *                RX_INVALID (no slot is substituted).                                     */
int rx_lnprob_batch_device(rx_handle *h, int N, int ncomp, const double *d_params,
                           const int32_t *d_src_index, double *d_lnp,
                           int32_t *d_status, int32_t *d_niter, void *stream);

/* Scheduling only: batches larger than twice the resident wavefronts are handed out hottest
 * walkers first (the ones that run into maxiter).  Results are bit-identical
 * either way.  Default on; the environment variable RX_NO_ORDER=1, read once by rx_create,
 * starts a handle with it off.                                                             */
int rx_set_issue_order(rx_handle *h, int hottest_first);
/* Scheduling only as well: wavefronts per SIMD of the solve launches.  0 (default) = chosen from the
 * batch size, 1 = lowest latency per walker, 2 = highest throughput.        */
int rx_set_waves_per_simd(rx_handle *h, int waves);
/* How matrix_'s linear solve [radex.so matrix_ -> lubksb_, SURVEY.md A.4 step 4, A.5] is made from iteration 12 of a
 * walker on: enabled = 1 (default) -- as a refinement of the solution of two iterations back against a kept inverse,
 * accepted when the correction is below 2^-43 of a population vector that sums to 1, with the pivoted elimination as the
 * fall-back; enabled = 0 -- the pivoted elimination every iteration, as the reference does.  Exists in
 * the CO / LVG instantiations (one AND two wavefronts per SIMD); elsewhere every solve is pivoted either way.
 * What it changes, measured against the reference's arithmetic (profiles/r5_refine_gate_*.txt, r5_big_parity_seeds*.txt,
 * r6_small_population_gpu.txt): the iteration count of about 1 walker in 10^4 flips by one; lnprob of converged walkers moves by
 * up to 2e-6 relative; walkers that stop at maxiter (chaotic iterations) move by up to 1.2e-3 (9.8e-4 with it off); level
 * populations above 1e-6 by up to 5e-8 relative (1e-8 with it off).  Populations BELOW ~1e-11 differ from the reference's by
 * more than 1e-4 relative WITH OR WITHOUT it (same figures either way: below 1.2e-14 absolute for every population under 1e-6): that is the
 * rounding error of any double-precision solve of this system, the reference's own LINPACK solve included -- against the exact
 * solution of its own system it is off by 4e-4 at 1e-12, 5 % at 1e-14 and by factors below 1e-17
 * (profiles/r6_small_population_accuracy.txt) -- so T_ex and tau of lines between such levels, and fluxes under the
 * background floor, carry no digits in the reference either.                                                               */
int rx_set_refinement(rx_handle *h, int enabled);
/* Diagnostics: totals over the 1-component / solve batches evaluated WHILE COUNTING WAS ON since the last reset -- out5 =
 * iterations, solves made as refinements, corrections made, attempts given up, inverses kept.  Counting is off by default:
 * rx_set_refinement_counting(h, 1) makes the following launches use an instantiation of the solve kernel that carries the
 * counters (the same arithmetic, the same results bit for bit, 2.8 % more time on the 1024-walker launch: it is the general
 * instantiation, which also serves a handle whose iteration limits differ from the reference's 10 / 200 or whose refinement
 * is switched off -- the default state runs a kernel with all of that as compile-time constants).                          */
int rx_set_refinement_counting(rx_handle *h, int enabled);
int rx_refinement_counters(rx_handle *h, uint64_t *out5, int reset);

/* The caller of lnprob on the device: emcee's StretchMove (a = 2) inside RedBlueMove with two
 * random halves, as driven by EnsembleSampler.run_mcmc in emcee/emcee_radex.py:483-499 and
 * emcee/emcee_radex_2comp.py:557-574.  Walker positions, log-probabilities and the random stream
 * stay in HBM; nothing crosses PCIe per step.  `nens` independent ensembles of `nwalkers` walkers
 * (even) advance together, coords is [nens*nwalkers][ndim], ensemble-major.  Random numbers are
 * counter based (Philox4x32-10 keyed by `seed`, counter = proposal, ensemble, step, purpose):
 * bit-reproducible per seed and identical on every rank of a multi-GPU run.
 *
 * rx_stretch_propose_device: half-step `split` (0/1) of step `step`: for each of the
 *   nens*nwalkers/2 proposals t: walker s = widx[t] of this half, partner c uniform from the other
 *   half, z = ((a-1)u+1)^2/a, q[t] = c - (c - s) z, factor[t] = (ndim-1) ln z; qsrc[t] (optional)
 *   = ens_src[ensemble] for rx_lnprob_batch_device's d_src_index.
 * rx_stretch_accept_device: accept q[t] iff ln u' < factor[t] + lnp_q[t] - lnp[widx[t]]
 *   (a NaN difference, -inf - -inf, rejects, as in emcee); updates coords, lnp, naccept.
 * rx_sampler_run_device: nsteps full steps (propose, rx_lnprob_batch_device over the proposals,
 *   accept; twice per step) enqueued on `stream`; d_lnp must hold the log-probabilities of
 *   d_coords on entry; d_chain [nsteps][nens*nwalkers][ndim] and d_chain_lnp
 *   [nsteps][nens*nwalkers] (each optional) receive the state after every step
 *   (get_chain / get_log_prob); d_ens_src [nens] = source slot per ensemble or NULL = slot 0.
 *   Asynchronous -- unless solve_ms_out is given (benchmarks): then HIP events bracket every
 *   solve launch on `stream`, the call waits for the last one and returns the summed kernel
 *   time of the 2*nsteps half-steps in milliseconds.                                        */
/* rx_sampler_run_async_device: the SAME chain as rx_sampler_run_device, bit for bit, run as ONE
 * persistent kernel in which every (step, half, ensemble, proposal) is a task that starts as soon as
 * the two walkers it reads are final (per-walker version counters in HBM) instead of waiting for the
 * whole previous half-step: a proposal that runs into maxiter delays only the tasks that depend on
 * its walker.  Asynchronous on `stream`; rx_sampler_wait synchronises the stream and reports
 * RX_E_TIMEOUT if a task gave up waiting (no finished task anywhere for 100 ms, rx_set_sampler_stall_ms -- the grid always drains).  The abort word is
 * sticky per handle: once raised it ends every later run of the handle at its first wait until
 * rx_sampler_wait has reported it (so a run enqueued before the wait cannot hide it); keep ONE async run
 * outstanding per handle when the result of each matters on its own.                          */
int rx_sampler_run_async_device(rx_handle *h, int nens, int nwalkers, int ncomp, double a,
                                uint64_t seed, int64_t step0, int nsteps,
                                const int32_t *d_ens_src, double *d_coords, double *d_lnp,
                                int32_t *d_naccept, double *d_chain, double *d_chain_lnp,
                                void *stream);
int rx_sampler_wait(rx_handle *h, void *stream);   /* reports (and lowers) the sticky abort word */
/* Longest time a task of the dataflow sampler polls for its inputs before it raises the abort flag
 * (default 10 s: a backstop behind the no-progress watchdog rx_set_sampler_stall_ms; real waits are milliseconds).  0 makes every wait that is not satisfied at once
 * give up: the safety path can be exercised on purpose (tests).                                  */
int rx_set_sampler_timeout_ms(rx_handle *h, double ms);

/* The dataflow sampler across the GPUs of one node (one process per GPU) -- the device-side form of the
 * reference's only parallelism, Pool(...).map(lnprob, walkers) (emcee/emcee_radex.py:480-488), and of
 * SURVEY.md 8(e)'s "each GPU writes its slice to all peers": the tasks of every half-step are dealt out in
 * contiguous blocks of ceil(nq / nranks) proposals, one block per rank; EVERY rank keeps a full replica of the
 * sampler's shared state (per-walker version counters, positions by version, log-probabilities, acceptance
 * counts) in one block of fine-grained device memory; a task polls and reads its own rank's replica only and
 * publishes its result into ALL replicas with system-scope stores over xGMI (positions and log-probability
 * first, one release fence, then the version).  No collective, no half-step barrier, no host in the loop: the
 * chain is bit-identical to the one-GPU run.  Per run_mcmc call the host needs two barriers of its own (any
 * transport): after rx_sampler_peer_begin on every rank (no peer may write into a replica that is still being
 * seeded) and after rx_sampler_wait on every rank (a replica is complete only when every peer has finished).
 *   rx_sampler_peer_setup    allocates this rank's replica for (nens, nwalkers, ncomp) and exports it:
 *                            ipc_handle_out[RX_IPC_HANDLE_BYTES] (hipIpcGetMemHandle), may be NULL; the environment
 *                            variable RX_NO_PEER=1 makes it fail with RX_E_UNSUPP (callers then use the half-step
 *                            schedule: the way to rehearse that fall-back)
 *   rx_sampler_peer_base     the replica as a device pointer (peers inside ONE process: two handles, tests)
 *   rx_sampler_peer_connect  ipc_handles: [nranks][RX_IPC_HANDLE_BYTES], every rank's handle in rank order
 *                            (own entry ignored), or bases: [nranks] device pointers valid in this process
 *   rx_sampler_peer_begin    seeds the replica from d_coords / d_lnp / d_naccept (may be NULL = zeros);
 *                            returns when it IS seeded (stream synchronised)        ... barrier ...
 *   rx_sampler_peer_run      this rank's tasks of nsteps steps as ONE persistent kernel, asynchronous; d_chain /
 *                            d_chain_lnp (optional) receive the rows of the walkers THIS rank updated (zero them
 *                            first and sum over ranks);  rx_sampler_wait            ... barrier ...
 *   rx_sampler_peer_finish   final state -> d_coords / d_lnp / d_naccept; RX_E_TIMEOUT if a task on ANY rank
 *                            gave up waiting (every rank then reports it)
 *   rx_sampler_peer_disconnect  unmaps the peers' replicas (hipIpcCloseMemHandle) and keeps this rank's own block;
 *   rx_sampler_peer_close    ... then frees it too (rx_sampler_peer_setup does this to a previous set-up by itself).  A
 *                            block must not be freed -- or its successor exported -- while a peer still has it mapped:
 *                            across processes the order is  disconnect on every rank ... barrier ... close / setup
 *                            (exporting the next block while a peer still mapped the last one made hipIpcGetMemHandle fail
 *                            with "invalid argument" now and then, more often with four ranks than with two)
 *   rx_set_sampler_grid_limit  rx_sampler_peer_run occupies at most `cus` compute units (0 = the whole GPU): ranks
 *                            that SHARE one GPU (rehearsals) must all be resident at once                   */
int rx_sampler_peer_setup(rx_handle *h, int nranks, int rank, int nens, int nwalkers, int ncomp,
                          void *ipc_handle_out);
void *rx_sampler_peer_base(rx_handle *h);
int rx_sampler_peer_connect(rx_handle *h, const void *ipc_handles, void *const *bases);
/* The identity of the handle's GPU that means the same thing in every process of the node whatever *_VISIBLE_DEVICES says: its
 * PCI bus id ("0000:c1:00.0", NUL-terminated, RX_BUS_ID_BYTES).  rx_sampler_peer_set_bus_ids: every rank's id in rank order
 * ([nranks][RX_BUS_ID_BYTES]), given BEFORE rx_sampler_peer_connect: which replicas share this GPU (-> its compute units are
 * split) and which GPU a remote replica lives on (-> hipDeviceCanAccessPeer, where that GPU is visible to this process) are
 * then decided from the ids, not from what hipPointerGetAttributes says about an IPC mapping (which may name the OPENING device
 * where the owner is not visible).  Without ids the pointer attributes are used, as before.  NULL forgets them.              */
#define RX_BUS_ID_BYTES 32
int rx_device_bus_id(rx_handle *h, char *out);
int rx_sampler_peer_set_bus_ids(rx_handle *h, const char *bus_ids);
int rx_sampler_peer_begin(rx_handle *h, const double *d_coords, const double *d_lnp,
                          const int32_t *d_naccept, void *stream);
int rx_sampler_peer_run(rx_handle *h, double a, uint64_t seed, int64_t step0, int nsteps,
                        const int32_t *d_ens_src, double *d_chain, double *d_chain_lnp, void *stream);
int rx_sampler_peer_finish(rx_handle *h, double *d_coords, double *d_lnp, int32_t *d_naccept,
                           void *stream);
int rx_sampler_peer_disconnect(rx_handle *h);
int rx_sampler_peer_close(rx_handle *h);
int rx_set_sampler_grid_limit(rx_handle *h, int cus);
/* Residency and failure handling of the persistent kernels.
 *   rx_sampler_peer_same_device  ranks of the connected group whose replica lives on THIS handle's device, own included (the
 *                            library's per-device registry of the run, filled by rx_sampler_peer_connect).  Such ranks must
 *                            all be resident at once -- a task may wait for a task of another rank -- so unless
 *                            rx_set_sampler_grid_limit says otherwise rx_sampler_peer_run gives each an equal share of the
 *                            device's compute units.  (Independent runs -- rx_sampler_run_async_device of two handles, two
 *                            fits, two processes -- need nothing of the kind: a task only ever waits for tasks that were
 *                            dequeued before it by wavefronts of ITS OWN launch, which are running.)
 *   rx_set_sampler_stall_ms  no-progress watchdog of every wait inside the dataflow kernels (default 100 ms): a waiting task
 *                            gives up -- abort word, all grids drain, RX_E_TIMEOUT from rx_sampler_wait / _peer_finish -- when
 *                            no task of ANY rank has finished for that long while every rank's grid is running (a rank that is
 *                            still loading its code object does not count as a stall) and the launch's FIRST task has finished
 *                            (what happens once -- the first access through a peer mapping -- is not a stall either; the
 *                            mappings are also touched by rx_sampler_peer_connect).  A task lasts a few milliseconds at most.  rx_set_sampler_timeout_ms (default now 10 s) stays the flat bound of a single wait.
 *   rx_sampler_peer_abort    raises the abort word in every connected replica from the HOST (a stream of its own): the rank
 *                            whose launch failed ends the others' kernels at once instead of letting them run into the watchdog. */
int rx_sampler_peer_same_device(const rx_handle *h);
int rx_set_sampler_stall_ms(rx_handle *h, double ms);
int rx_sampler_peer_abort(rx_handle *h);
/* Preflight of a multi-GPU run (bench.py prints it): out3 = { hipDeviceCanAccessPeer(dev_a -> dev_b) (1 / 0, -1 = the query
 * failed), link type and hop count of hipExtGetLinkTypeAndHopCount (-1 = unknown; type 4 = xGMI, 2 = PCIe) }.  No handle.   */
int rx_peer_topology(int dev_a, int dev_b, int32_t *out3);
/* Counters of the dataflow sampler's launches (benchmarks; off by default): waits for the handle's work,
 * copies the counters accumulated since the last call into out6 (may be NULL), zeroes them and switches the
 * counting on (enable = 1) or off for the launches that follow.  out6: [0] tasks, [1] tasks whose proposal
 * reached the solver (lnprior finite), [2] RADEX iterations summed over them, [3] solves that stopped at
 * maxiter, [4] 100 MHz wall-clock ticks summed over the tasks: the evaluation that stands (inputs read ->
 * log-probability known), [5] ticks summed over the tasks between "dequeued" and the first evaluation (polling).
 * [1]-[3] describe the evaluation that stands, i.e. the chain's proposals, whatever head starts were taken. */
int rx_sampler_stats(rx_handle *h, int enable, uint64_t *out6);
/* The dataflow sampler's head starts (both forms, one GPU and peers).  A task that reads a walker which is still
 * being updated -- its own or its partner -- does not sleep: it evaluates its proposal from that walker's newest
 * final position (a stretch move is rejected more often than not, and a rejected update leaves the walker where
 * it was) or from the proposal the pending task has published, and keeps each evaluation with the positions it
 * assumed.  When both walkers are final their real positions are compared bit for bit with those assumptions: a
 * match is the evaluation of the chain's proposal; none: the task is evaluated from the real positions, as
 * without a head start.  The chain is the same chain, bit for bit; only idle time is spent.
 *   mode  -1 (default) and 1: on; 0: off.  The head start exists for launches that run one wavefront per SIMD
 *         (the latency regime: up to 1536 tasks per half-step on 256 CUs, idle wavefronts); where two wavefronts
 *         share a SIMD a wasted evaluation slows its neighbour down, and the mode is ignored (measured 2-4 % slower).
 * rx_sampler_spec_stats: out2[0] tasks that took a head start, out2[1] of those, tasks evaluated again from the
 * real positions (every hypothesis wrong) --
 * as counted up to the most recent rx_sampler_stats call that read the counters.                              */
int rx_set_sampler_speculation(rx_handle *h, int mode);
int rx_sampler_spec_stats(rx_handle *h, uint64_t *out2);
int rx_stretch_propose_device(rx_handle *h, int nens, int nwalkers, int ndim, double a,
                              uint64_t seed, int64_t step, int split,
                              const int32_t *d_ens_src, const double *d_coords, double *d_q,
                              double *d_factor, int32_t *d_widx, int32_t *d_qsrc, void *stream);
int rx_stretch_accept_device(rx_handle *h, int nens, int nwalkers, int ndim, uint64_t seed,
                             int64_t step, int split, const double *d_q,
                             const double *d_lnp_q, const double *d_factor,
                             const int32_t *d_widx, double *d_coords, double *d_lnp,
                             int32_t *d_naccept, void *stream);
int rx_sampler_run_device(rx_handle *h, int nens, int nwalkers, int ncomp, double a,
                          uint64_t seed, int64_t step0, int nsteps, const int32_t *d_ens_src,
                          double *d_coords, double *d_lnp, int32_t *d_naccept, double *d_chain,
                          double *d_chain_lnp, double *solve_ms_out, void *stream);

/* Replaces model_lvg(Jup, p, R) (emcee/emcee_radex.py:120-130,
 * emcee/emcee_radex_2comp.py:122-147): flux_out[N][nJ(src)] in Jy km/s.
 * Walkers the reference would reject with ValueError get NaN fluxes and
 * status RX_INVALID.  No prior is applied.                                   */
int rx_model_flux_batch(rx_handle *h, int src, int N, const double *params,
                        double *flux_out, int32_t *status, int32_t *niter);
int rx_model_flux_batch_device(rx_handle *h, int src, int N, const double *d_params,
                               double *d_flux_out, int32_t *d_status,
                               int32_t *d_niter, void *stream);

/* Replaces Radex(...).run_radex() + the state properties used by the
 * reference's known-answer tests (emcee/pyradex/tests/test_radex.py:99-115):
 * physical (linear) inputs, cold start.
 *   tkin[N], cdmol[N], dens[N][rx_npart] (file partner order, cm^-3)
 * Outputs (each may be NULL): xpop[N][nlev], tex[N][nline], tau[N][nline],
 * sb[N][nline] = source_line_surfbrightness (erg s^-1 cm^-2 Hz^-1 sr^-1).
 * Uses the background of source slot `src`.  Host pointers.                 */
int rx_solve_batch(rx_handle *h, int src, int N, const double *tkin,
                   const double *cdmol, const double *dens, double *xpop,
                   double *tex, double *tau, double *sb, int32_t *status,
                   int32_t *niter);

/* Replaces lubksb_(a, n, np, indx, b) as patched by pyradex (radex.so lubksb_ -> sgeir_,
 * SURVEY.md A.5; called from matrix_): for each of N systems A[n][n] (row-major, n <= the
 * handle's padded level count) the last row is replaced by ones, rhs = e_last, and the
 * solution by elimination with LINPACK's partial pivoting (sgefa_'s pivot choices; x = e_last on
 * a singular system, as sgeir_ leaves it) is returned in x[N][n].  Host pointers.  Exposed so
 * the pivoted solve can be checked on its own against the reference's lubksb_ vectors. */
int rx_lubksb_batch(rx_handle *h, int N, int n, const double *A, double *x);

/* Same, and also the pivot choices: pivrow[N][n] (optional) receives, for every system, the index
 * of the row chosen as pivot of step 0, 1, ..., n-1 -- sgefa_'s ipvt(k) [radex.so sgefa_,
 * SURVEY.md A.5] expressed as rows instead of interchanges (meaningless on a singular system).
 * Integer output: the parity bar is equality, exact ties (isamax's first-maximum rule) included. */
int rx_lubksb_pivots_batch(rx_handle *h, int N, int n, const double *A, double *x, int32_t *pivrow);

/* Replaces escprob_(tau) (radex.so@0xa9c0; called 40x per iteration from
 * matrix_): beta[N] = escape probability of tau[N] for geometry `method` (1 sphere, 2 lvg,
 * 3 slab), evaluated by the device routine the solve kernel uses.  Host pointers.  Exposed so the
 * routine can be checked on its own against the reference binary's escprob_ vectors.
 * method = 0 evaluates the kernel's natural logarithm instead (the LVG branch and the excitation
 * temperatures use it in place of the library's): y = log(x), special operands as libm.        */
int rx_escprob_batch(rx_handle *h, int method, int N, const double *tau, double *beta);

/* Kernel timing hook for bench.py: runs rx_lnprob_batch_device `reps` times
 * back-to-back on `stream`, bracketing every launch with HIP events on that
 * same stream, and returns the mean per-launch kernel time in milliseconds.  */
int rx_time_lnprob_device(rx_handle *h, int N, int ncomp, const double *d_params,
                          const int32_t *d_src_index, double *d_lnp,
                          int32_t *d_status, int32_t *d_niter, void *stream,
                          int reps, double *ms_mean_out);

/* Name of the dominant kernel symbol for the loaded molecule (profiles/):
 * "rx_solve_kernel<NL, 1, true>" when the molecule takes the specialised
 * instantiation (it fills the size NL, and its lines are a ladder: line l
 * connects level l+1 to level l, in file order), "..., false>" otherwise.  */
const char *rx_kernel_name(const rx_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* RADEX_EMCEE_AMD_H */
