/*
 * radex_oracle.c -- CPU restatement of the reference hot path.
 * TEST INFRASTRUCTURE ONLY (see radex_oracle.h).  Plain C, IEEE double.
 *
 * Every function cites the reference location it follows:
 *   [REF file:line]  = /root/reference/<file>:<line>
 *   [BIN 0xADDR]     = address inside /root/reference/emcee/pyradex/radex/radex.so
 *                      (x86-64 Mach-O), arithmetic per SURVEY.md Appendix A.
 * Compile with -ffp-contract=off so no FMA contraction changes the rounding
 * relative to the reference binary (which contains no FMA instructions).
 */
#include "radex_oracle.h"

#include <alloca.h>
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants, Appendix A.0 (binary literals; float32-promoted where the
 *      Fortran source lacked a d0 exponent) -------------------------------- */
static const double FK     = 1.4387809925261357;      /* [BIN 0x26bc8] hc/k   */
static const double THC    = 3.972907393443411e-16;   /* [BIN 0x26bd8] 2hc    */
static const double FGAUS  = 26.753802360251857;      /* [BIN 0x26c10]        */
static const double MINPOP = 1e-20;                   /* [BIN 0x26c20]        */
static const double CCRIT  = 9.999999974752427e-07;   /* [BIN 0x26c28] 1e-6f  */
static const double RELAX_NEW = 0.30000001192092896;  /* [BIN 0x26c30] 0.3f   */
static const double RELAX_OLD = 0.699999988079071;    /* [BIN 0x26c38] 0.7f   */
static const double EXPGUARD = 160.0;                 /* [BIN 0x26bd0]        */
static const double SEED_D = 1e-30;                   /* [BIN 0x26bb8]        */
/* ([BIN 0x26c40] 1e-30f = 1.0000000031710769e-30 only enters rhs / row nplus of matrix_, which the patched lubksb_ ignores) */
static const double THICK_D = 0.01;                   /* [BIN 0x26c18]        */
static const double THICK_F = 0.009999999776482582;   /* [BIN 0x26b78] 0.01f  */
static const double FAT     = 1e5;                    /* [BIN 0x26bc0]        */
/* Python-side constants, astropy CODATA-2018 [REF emcee/pyradex/core.py:981-984] */
static const double THC_PY = 3.9728917142978573e-16;
static const double FK_PY  = 1.4387768775039338;

/* ---- iterative-refinement variant of step 4 (test infrastructure for the device kernels' scheme;
 *      OFF unless rxo_set_refine was called: see radex_oracle.h) -------------------------------- */
static int    RF_FIRST = 0, RF_MAXSTEPS = 4, RF_LAG = 2, RF_CRIT = 0, RF_BACKOFF = 0;
static double RF_TOL = 1e-10, RF_D1MAX = 0.0, RF_LOOSE = 0.0;
static double RF_CW_REL = 0.0, RF_CW_FLOOR = 0.0, RF_CW_LOOSE = 0.0;   /* crit 2: componentwise acceptance (rxo_set_refine_componentwise) */
static long   RF_CNT[5];            /* full solves, refined solves, refinement steps, failed attempts, inverses kept */

void rxo_set_refine(int first_iter, double tol, int max_steps, int lag, int crit, double d1max, double loose, int backoff)
{
    RF_FIRST = first_iter; RF_TOL = tol; RF_MAXSTEPS = max_steps;
    RF_LAG = (lag == 1) ? 1 : 2; RF_CRIT = crit; RF_D1MAX = d1max;
    RF_LOOSE = loose; RF_BACKOFF = backoff;
}

void rxo_set_refine_componentwise(double rel, double floor, double loose_rel)
{
    RF_CW_REL = rel; RF_CW_FLOOR = floor; RF_CW_LOOSE = loose_rel;
}

void rxo_refine_counters(long *full, long *refined, long *steps, long *failed, long *kept, int reset)
{
    if (full) *full = RF_CNT[0];
    if (refined) *refined = RF_CNT[1];
    if (steps) *steps = RF_CNT[2];
    if (failed) *failed = RF_CNT[3];
    if (kept) *kept = RF_CNT[4];
    if (reset) RF_CNT[0] = RF_CNT[1] = RF_CNT[2] = RF_CNT[3] = RF_CNT[4] = 0;
}

static void set_err(char *err, size_t n, const char *msg)
{
    if (err && n) { strncpy(err, msg, n - 1); err[n - 1] = 0; }
}

/* ======================================================================== */
/* readdata_, file-format half [BIN 0x1cf90-0x1e338]; LAMDA layout as in    */
/* SURVEY App. A.2.  Records are read strictly in file order, comment lines  */
/* are consumed positionally exactly as RADEX does (it never looks at '!').  */
/* ======================================================================== */
/* One Fortran unit read list-directed, as far as LAMDA files use the form (what the reference binary does with ~100 well- and
 * ill-formed files is recorded in tests/golden/ref_lamda_corpus.json, make_ref_lamda_corpus.py): a READ starts on a new record,
 * items are separated by blanks / tabs and at most one comma, a READ whose record runs out goes on in the next one, the rest of
 * its last record is skipped, a READ without items skips one record; end of file is an error.  Integers are sign + digits;
 * reals take e / E / d / D or a bare sign as the exponent mark; repeat counts, null values, slashes and quoted strings are
 * errors here (not implemented).                                                                                      */
typedef struct { FILE *f; char *rec; size_t cap, len, col; int fresh, bad; } rd_unit;

static int rd_record(rd_unit *u)
{
    int c = fgetc(u->f);
    u->len = u->col = 0;
    if (c == EOF) { u->bad = 1; return 0; }
    for (; c != EOF && c != '\n'; c = fgetc(u->f)) {
        if (u->len + 2 > u->cap) { u->cap = u->cap ? 2 * u->cap : 256; u->rec = (char *)realloc(u->rec, u->cap); }
        u->rec[u->len++] = (char)c;
    }
    if (u->len && u->rec[u->len - 1] == '\r') u->len--;
    if (u->rec) u->rec[u->len] = 0;
    return 1;
}

static void rd_begin(rd_unit *u) { u->fresh = 1; }
static void rd_end(rd_unit *u) { if (u->fresh && !u->bad) rd_record(u); u->fresh = 1; }
static void rd_skip(rd_unit *u) { rd_begin(u); rd_end(u); }

/* the next item of the READ in progress: [*b, *e) inside u->rec; 0 on error */
static int rd_item(rd_unit *u, size_t *b, size_t *e)
{
    if (u->bad) return 0;
    if (u->fresh) { if (!rd_record(u)) return 0; u->fresh = 0; }
    for (;;) {
        while (u->col < u->len && (u->rec[u->col] == ' ' || u->rec[u->col] == '\t')) u->col++;
        if (u->col < u->len) break;
        if (!rd_record(u)) return 0;
    }
    char c = u->rec[u->col];
    if (c == ',' || c == '/' || c == '\'' || c == '"') { u->bad = 1; return 0; }
    *b = u->col;
    while (u->col < u->len && !strchr(" \t,/", u->rec[u->col])) u->col++;
    *e = u->col;
    for (size_t i = *b; i < *e; i++) if (u->rec[i] == '*') { u->bad = 1; return 0; }
    while (u->col < u->len && (u->rec[u->col] == ' ' || u->rec[u->col] == '\t')) u->col++;
    if (u->col < u->len && u->rec[u->col] == ',') u->col++;
    return 1;
}

static int rd_int(rd_unit *u)
{
    size_t b, e;
    if (!rd_item(u, &b, &e)) return 0;
    size_t k = b + ((u->rec[b] == '+' || u->rec[b] == '-') ? 1 : 0);
    if (k == e || e - b > 11) { u->bad = 1; return 0; }
    for (size_t i = k; i < e; i++) if (!isdigit((unsigned char)u->rec[i])) { u->bad = 1; return 0; }
    char tmp[16];
    memcpy(tmp, u->rec + b, e - b); tmp[e - b] = 0;
    long long v = strtoll(tmp, NULL, 10);
    if (v > 2147483647LL || v < -2147483647LL - 1) { u->bad = 1; return 0; }
    return (int)v;
}

static double rd_real(rd_unit *u)
{
    size_t b, e;
    char tmp[80];
    if (!rd_item(u, &b, &e)) return 0.0;
    if (e - b > 70) { u->bad = 1; return 0.0; }
    size_t n = 0, mark = 0; int has_e = 0;
    for (size_t i = b; i < e; i++) {
        char c = u->rec[i];
        if (c == 'd' || c == 'D' || c == 'E') c = 'e';
        if (!(isdigit((unsigned char)c) || c == '+' || c == '-' || c == '.' || c == 'e')) { u->bad = 1; return 0.0; }
        if (c == 'e') has_e = 1;
        if ((c == '+' || c == '-') && n > 0) mark = n;
        tmp[n++] = c;
    }
    tmp[n] = 0;
    if (!has_e && mark > 0 && (tmp[mark - 1] == '.' || isdigit((unsigned char)tmp[mark - 1]))) {   /* 1.0-11 = 1.0e-11 */
        memmove(tmp + mark + 1, tmp + mark, n - mark + 1);
        tmp[mark] = 'e';
    }
    char *end;
    double v = strtod(tmp, &end);                /* correctly rounded, as libgfortran's conversion */
    if (end == tmp || *end || !isfinite(v)) { u->bad = 1; return 0.0; }
    return v;
}

static void rd_word(rd_unit *u) { size_t b, e; (void)rd_item(u, &b, &e); }

/* Limits of the reference binary (its STOP messages: "too many ..."; ref_lamda_corpus.json) */
enum { RXO_MAXLEV = 2999, RXO_MAXLINE = 99999, RXO_MAXCOLL = 99999, RXO_MAXTEMP = 99 };

rxo_mol *rxo_mol_load(const char *path, char *err, size_t errlen)
{
    FILE *f = fopen(path, "rb");
    if (!f) { set_err(err, errlen, "cannot open molecular data file"); return NULL; }
    rxo_mol *m = (rxo_mol *)calloc(1, sizeof *m);
    rd_unit U = {f, NULL, 0, 0, 0, 1, 0}, *u = &U;
#define CHECK(cond) do { if (u->bad || !(cond)) goto fail; } while (0)
    rd_skip(u);                                   /* !MOLECULE */
    rd_skip(u);                                   /* the name, (a) */
    rd_skip(u);                                   /* !MOLECULAR WEIGHT */
    rd_begin(u); m->amass = rd_real(u); rd_end(u);
    rd_skip(u);                                   /* !NUMBER OF ENERGY LEVELS */
    rd_begin(u); m->nlev = rd_int(u); rd_end(u);
    CHECK(m->nlev >= 2 && m->nlev <= RXO_MAXLEV);             /* "too few / too many energy levels defined" */
    m->eterm = (double *)calloc(m->nlev, sizeof(double));
    m->gstat = (double *)calloc(m->nlev, sizeof(double));
    rd_skip(u);                                   /* !LEVEL + ENERGIES + WEIGHT + QN */
    for (int i = 0; i < m->nlev; i++) {           /* stored by position; number, energy, weight and the quantum-number string are read */
        rd_begin(u);
        int no = rd_int(u);
        m->eterm[i] = rd_real(u);
        m->gstat[i] = rd_real(u);
        rd_word(u);
        rd_end(u);
        CHECK(no >= 1 && no <= m->nlev);          /* "illegal level number" */
    }
    rd_skip(u);                                   /* !NUMBER OF RADIATIVE TRANSITIONS */
    rd_begin(u); m->nline = rd_int(u); rd_end(u);
    CHECK(m->nline >= 1 && m->nline <= RXO_MAXLINE);
    m->iupp = (int *)calloc(m->nline, sizeof(int));
    m->ilow = (int *)calloc(m->nline, sizeof(int));
    m->aeinst = (double *)calloc(m->nline, sizeof(double));
    m->spfreq = (double *)calloc(m->nline, sizeof(double));
    m->eup = (double *)calloc(m->nline, sizeof(double));
    m->xnu = (double *)calloc(m->nline, sizeof(double));
    rd_skip(u);                                   /* !TRANS + ... */
    for (int l = 0; l < m->nline; l++) {
        rd_begin(u);
        int no = rd_int(u);
        m->iupp[l] = rd_int(u);
        m->ilow[l] = rd_int(u);
        m->aeinst[l] = rd_real(u);
        m->spfreq[l] = rd_real(u);
        m->eup[l] = rd_real(u);
        rd_end(u);
        CHECK(no >= 1 && no <= m->nline);         /* "illegal line number" */
        /* (for a level index outside 1..nlev the binary reads outside eterm -- eterm(0) is amass: not restated, an error) */
        CHECK(m->iupp[l] >= 1 && m->iupp[l] <= m->nlev && m->ilow[l] >= 1 && m->ilow[l] <= m->nlev);
        /* xnu = energy difference, NOT the listed frequency [BIN 0x1d735-0x1d745] */
        m->xnu[l] = m->eterm[m->iupp[l] - 1] - m->eterm[m->ilow[l] - 1];
        CHECK(!(m->xnu[l] < 1e-30));              /* "illegal line frequency" */
    }
    rd_skip(u);                                   /* !NUMBER OF COLL PARTNERS */
    rd_begin(u); m->npart = rd_int(u); rd_end(u);
    CHECK(m->npart >= 1 && m->npart <= RXO_MAXPART);
    for (int ip = 0; ip < m->npart; ip++) {
        rd_skip(u);                               /* !COLLISIONS BETWEEN */
        /* (i1,a): the FIRST CHARACTER of the record is the partner id (a blank reads as 0) */
        CHECK(rd_record(u));
        u->fresh = 1;
        {
            char c = u->len ? u->rec[0] : ' ';
            CHECK(c == ' ' || isdigit((unsigned char)c));
            m->part_id[ip] = c == ' ' ? 0 : c - '0';
        }
        CHECK(m->part_id[ip] >= 1 && m->part_id[ip] <= 7);    /* (0 reads density(0); 8, 9: never set by pyradex, core.py:476-482) */
        rd_skip(u);                               /* !NUMBER OF COLL TRANS */
        rd_begin(u); m->ncoll[ip] = rd_int(u); rd_end(u);
        CHECK(m->ncoll[ip] >= 1 && m->ncoll[ip] <= RXO_MAXCOLL);
        rd_skip(u);                               /* !NUMBER OF COLL TEMPS */
        rd_begin(u); m->ntemp[ip] = rd_int(u); rd_end(u);
        CHECK(m->ntemp[ip] >= 1 && m->ntemp[ip] <= RXO_MAXTEMP);   /* (0: the binary goes on with a column it never read) */
        m->temp[ip] = (double *)calloc(m->ntemp[ip], sizeof(double));
        m->lcu[ip] = (int *)calloc(m->ncoll[ip], sizeof(int));
        m->lcl[ip] = (int *)calloc(m->ncoll[ip], sizeof(int));
        m->coll[ip] = (double *)calloc((size_t)m->ncoll[ip] * m->ntemp[ip], sizeof(double));
        rd_skip(u);                               /* !COLL TEMPS */
        rd_begin(u);
        for (int t = 0; t < m->ntemp[ip]; t++) m->temp[ip][t] = rd_real(u);
        rd_end(u);
        CHECK(1);
        rd_skip(u);                               /* !TRANS + UP + LOW + COLLRATES */
        int kept = 0;
        const int declared = m->ncoll[ip];
        for (int c = 0; c < declared; c++) {
            rd_begin(u);
            int no = rd_int(u), up = rd_int(u), lo = rd_int(u);
            double *K = m->coll[ip] + (size_t)kept * m->ntemp[ip];
            for (int t = 0; t < m->ntemp[ip]; t++) K[t] = rd_real(u);
            rd_end(u);
            CHECK(no >= 1 && no <= declared);     /* "illegal collision number" */
            CHECK(up >= 1 && lo >= 1 && up <= RXO_MAXLEV && lo <= RXO_MAXLEV);
            if (up > m->nlev || lo > m->nlev) continue;        /* stored by the binary, never used: its loops run over 1..nlev */
            m->lcu[ip][kept] = up; m->lcl[ip][kept] = lo;
            kept++;
        }
        m->ncoll[ip] = kept;
    }
#undef CHECK
    free(U.rec); fclose(f);
    return m;
fail:
    free(U.rec); fclose(f);
    rxo_mol_free(m);
    set_err(err, errlen, "malformed LAMDA file");
    return NULL;
}

void rxo_mol_free(rxo_mol *m)
{
    if (!m) return;
    free(m->eterm); free(m->gstat); free(m->iupp); free(m->ilow);
    free(m->aeinst); free(m->spfreq); free(m->eup); free(m->xnu);
    for (int i = 0; i < RXO_MAXPART; i++) {
        free(m->temp[i]); free(m->lcu[i]); free(m->lcl[i]); free(m->coll[i]);
    }
    free(m);
}

rxo_state *rxo_state_new(const rxo_mol *m, int method, double deltav_kms)
{
    rxo_state *s = (rxo_state *)calloc(1, sizeof *s);
    int n = m->nlev, L = m->nline;
    s->mol = m; s->method = method;
    /* core.py:447-454: deltav km/s -> cm/s */
    s->deltav = deltav_kms * 1e5;
    s->crate = (double *)calloc((size_t)n * n, sizeof(double));
    s->ctot = (double *)calloc(n, sizeof(double));
    s->xpop = (double *)calloc(n, sizeof(double));
    s->xpopold = (double *)calloc(n, sizeof(double));
    s->tex = (double *)calloc(L, sizeof(double));
    s->taul = (double *)calloc(L, sizeof(double));
    s->backi = (double *)calloc(L, sizeof(double));
    s->totalb = (double *)calloc(L, sizeof(double));
    s->trj = (double *)calloc(L, sizeof(double));
    s->yrate = (double *)calloc((size_t)n * n, sizeof(double));
    s->rhs = (double *)calloc(n + 1, sizeof(double));
    s->lu = (double *)calloc((size_t)n * n, sizeof(double));
    s->ipvt = (int *)calloc(n, sizeof(int));
    if (RF_FIRST > 0) {
        for (int p = 0; p < 2; p++) {
            s->rf_minv[p] = (double *)calloc((size_t)n * n, sizeof(double));
            s->rf_x[p] = (double *)calloc(n, sizeof(double));
        }
        s->rf_r = (double *)calloc(n, sizeof(double));
        s->rf_d = (double *)calloc(n, sizeof(double));
        s->rf_a = (double *)calloc((size_t)n * n, sizeof(double));
    }
    return s;
}

void rxo_state_free(rxo_state *s)
{
    if (!s) return;
    free(s->crate); free(s->ctot); free(s->xpop); free(s->xpopold);
    free(s->tex); free(s->taul); free(s->backi); free(s->totalb); free(s->trj);
    free(s->yrate); free(s->rhs); free(s->lu); free(s->ipvt);
    if (s->rf_a) {
#ifdef _OPENMP
#pragma omp critical(rxo_refine_counters)
#endif
        {
            RF_CNT[0] += s->rf_full; RF_CNT[1] += s->rf_refined;
            RF_CNT[2] += s->rf_steps; RF_CNT[3] += s->rf_failed; RF_CNT[4] += s->rf_kept;
        }
    }
    free(s->rf_minv[0]); free(s->rf_minv[1]); free(s->rf_x[0]); free(s->rf_x[1]);
    free(s->rf_r); free(s->rf_d); free(s->rf_a);
    free(s);
}

/* ======================================================================== */
/* readdata_, arithmetic half [BIN 0x1e338-0x1fd0a]:            */
/*  - totdens = sum(density)                                  [BIN 0x1f390]  */
/*  - per partner with density>0: bracket tkin, LINEAR interpolation of the  */
/*    downward rates, crate(up,low) += density(id)*colld                     */
/*  - detailed balance over ALL level pairs with eterm(iup)-eterm(ilo) > 0   */
/*    [BIN 0x1f3dd-0x1fd0a]: e=(ediff*fk)/tkin; e>=160 -> 0 else              */
/*    crate(ilo,iup) = (exp(-e)*(g(iup)/g(ilo)))*crate(iup,ilo)               */
/*  - ctot(i) = sum_j crate(i,j)                                              */
/* ======================================================================== */
int rxo_rates(rxo_state *s)
{
    const rxo_mol *m = s->mol;
    int n = m->nlev;
    double tk = s->tkin;
    memset(s->crate, 0, sizeof(double) * n * n);
    double tot = 0.0;
    for (int k = 0; k < RXO_MAXPART; k++) tot += s->density[k];
    s->totdens = tot;
    /* per partner, in file order: the interpolated rates go into a table colld(up,low) -- a pair listed twice keeps its LAST row --
     * and crate += density(id) * colld; two partners with the same id both count (ref_lamda_corpus.json: ok_duplicate_rate_row,
     * ok_duplicate_partner_id).  A negative rate is not an error in the binary (bad_negative_rate: crate < 0). */
    double *tab = (double *)malloc(sizeof(double) * n * n);
    for (int ip = 0; ip < m->npart; ip++) {
        double dens = s->density[m->part_id[ip] - 1];
        if (!(dens > 0.0)) continue;
        int nt = m->ntemp[ip];
        const double *T = m->temp[ip];
        int mode, it = 0; double t = 0.0;   /* mode 0: single column it; 1: lerp */
        if (nt <= 1)            { mode = 0; it = 0; }
        else if (tk <= T[0])    { mode = 0; it = 0; }          /* "Tkin lower than..." */
        else if (tk >= T[nt-1]) { mode = 0; it = nt - 1; }     /* "Tkin higher than..." */
        else {
            mode = 1;
            for (it = 0; it < nt - 1; it++) if (tk > T[it] && tk <= T[it + 1]) break;
            if (it >= nt - 1) { mode = 0; it = nt - 1; }       /* (temperatures out of order: no bracket found) */
            else t = (tk - T[it]) / (T[it + 1] - T[it]);
        }
        memset(tab, 0, sizeof(double) * n * n);
        for (int c = 0; c < m->ncoll[ip]; c++) {
            const double *K = m->coll[ip] + (size_t)c * nt;
            double colld = mode ? K[it] + t * (K[it + 1] - K[it]) : K[it];
            if (colld < 0.0) colld = K[it];       /* a negative interpolate falls back to the lower grid column (no error): bad_negative_rate* */
            tab[(m->lcu[ip][c] - 1) * n + (m->lcl[ip][c] - 1)] = colld;
        }
        for (int k = 0; k < n * n; k++) s->crate[k] += dens * tab[k];
    }
    free(tab);
    for (int iup = 0; iup < n; iup++)
        for (int ilo = 0; ilo < n; ilo++) {
            double ediff = m->eterm[iup] - m->eterm[ilo];
            if (ediff > 0.0) {
                double e = ediff * FK / tk;
                if (e >= EXPGUARD) s->crate[ilo * n + iup] = 0.0;
                else s->crate[ilo * n + iup] =
                    exp(-e) * (m->gstat[iup] / m->gstat[ilo]) * s->crate[iup * n + ilo];
            }
        }
    for (int i = 0; i < n; i++) {
        double c = 0.0;
        for (int j = 0; j < n; j++) c += s->crate[i * n + j];
        s->ctot[i] = c;
    }
    return 0;
}

/* backrad_, tbg>0 branch [BIN 0x1be30-0x1c390] */
void rxo_backrad(rxo_state *s, double tbg)
{
    const rxo_mol *m = s->mol;
    s->tbg = tbg;
    for (int l = 0; l < m->nline; l++) {
        double x = m->xnu[l];
        double h = FK * x / tbg;
        if (h >= EXPGUARD) s->backi[l] = SEED_D;                     /* the DOUBLE 1e-30 [BIN 0x26bb8]: pinned by ref_backrad_guard.json */
        else s->backi[l] = THC * pow(x, 3.0) / (exp(h) - 1.0);   /* xnu**3. -> pow [BIN 0x1c04f] */
        s->trj[l] = tbg;
        s->totalb[l] = s->backi[l];
    }
}

/* escprob_ [BIN 0xa9c0-0xad60] */
double rxo_escprob(double tau, int method)
{
    double taur = 0.5 * tau;
    if (method == 2) {                       /* LVG (hot path) */
        if (fabs(taur) < 0.009999999776482582) return 1.0;
        if (fabs(taur) < 7.0)
            return 2.0 * (1.0 - exp(-2.3399999141693115 * taur)) / (4.679999828338623 * taur);
        return 2.0 / (taur * 4.0 * sqrt(log(taur / 1.7724538498928541)));
    } else if (method == 1) {                /* uniform sphere */
        if (fabs(taur) < 0.10000000149011612)
            return 1.0 - 0.75 * taur + (taur * taur) / 2.5 - (taur * taur * taur) / 6.0
                   + (taur * taur * taur * taur) / 17.5;
        if (fabs(taur) > 50.0) return 0.75 / taur;
        return 0.75 / taur * (1.0 - 1.0 / (2.0 * (taur * taur)) +
               (1.0 / taur + 1.0 / (2.0 * (taur * taur))) * exp(-2.0 * taur));
    } else {                                 /* slab */
        double x = 3.0 * tau;
        if (fabs(x) < 0.10000000149011612) return 1.0 - 1.5 * (tau + tau * tau);
        if (fabs(x) > 50.0) return 1.0 / x;
        return (1.0 - exp(-x)) / x;
    }
}

/* ---- LINPACK sgefa/sgesl as compiled (all REAL = double) ------------------
 * [BIN sgefa_ 0xf3d0, sgesl_ 0xdb70, isamax_/sscal_/saxpy_]; column-major.  */
static int lin_gefa(double *a, int lda, int n, int *ipvt)
{
    int info = 0;
    for (int k = 0; k < n - 1; k++) {
        int l = k; double smax = fabs(a[k + k * lda]);
        for (int i = k + 1; i < n; i++) {            /* isamax: first maximum */
            double v = fabs(a[i + k * lda]);
            if (v > smax) { smax = v; l = i; }
        }
        ipvt[k] = l;
        if (a[l + k * lda] == 0.0) { info = k + 1; continue; }
        if (l != k) { double t = a[l + k * lda]; a[l + k * lda] = a[k + k * lda]; a[k + k * lda] = t; }
        double t = -1.0 / a[k + k * lda];
        for (int i = k + 1; i < n; i++) a[i + k * lda] = t * a[i + k * lda];   /* sscal */
        for (int j = k + 1; j < n; j++) {
            double tj = a[l + j * lda];
            if (l != k) { a[l + j * lda] = a[k + j * lda]; a[k + j * lda] = tj; }
            for (int i = k + 1; i < n; i++)                                      /* saxpy */
                a[i + j * lda] = a[i + j * lda] + tj * a[i + k * lda];
        }
    }
    ipvt[n - 1] = n - 1;
    if (a[(n - 1) + (n - 1) * lda] == 0.0) info = n;
    return info;
}

static void lin_gesl(const double *a, int lda, int n, const int *ipvt, double *b)
{
    for (int k = 0; k < n - 1; k++) {
        int l = ipvt[k]; double t = b[l];
        if (l != k) { b[l] = b[k]; b[k] = t; }
        for (int i = k + 1; i < n; i++) b[i] = b[i] + t * a[i + k * lda];
    }
    for (int k = n - 1; k >= 0; k--) {
        b[k] = b[k] / a[k + k * lda];
        double t = -b[k];
        for (int i = 0; i < k; i++) b[i] = b[i] + t * a[i + k * lda];
    }
}

/* lubksb_ as patched by pyradex [BIN 0x17cb0] + sgeir_ [BIN 0x16d50] (A.5):
 * compact k x k copy, last row <- 1.0, rhs <- e_last, LU + solve; the second
 * (residual) solve only feeds an accuracy estimate and does not change x.
 * On a singular matrix sgeir_ returns without solving (x stays e_last).     */
int rxo_lubksb(double *a, int n, double *x, int *ipvt)
{
    for (int j = 0; j < n; j++) a[(n - 1) + j * n] = 1.0;
    for (int i = 0; i < n; i++) x[i] = 0.0;
    x[n - 1] = 1.0;
    int info = lin_gefa(a, n, n, ipvt);
    if (info == 0) lin_gesl(a, n, n, ipvt, x);
    return info;
}

/* ---- the refinement variant of step 4 (NOT the reference's arithmetic; see radex_oracle.h) ------
 * The system is A x = e_last with A = yrate, last row <- 1 (lubksb_, A.5).  Between iterations only
 * the ~3 nline radiative entries of A move, and the walkers that never converge alternate between
 * two states: the inverse kept from an earlier iteration OF THE SAME PARITY is a good preconditioner.
 *   x <- x_kept;  repeat { r = e_last - A x;  d = Minv r;  x += d } until the correction is small (rf_solve).
 * rf_solve returns 1 when it produced the solution in s->rhs, 0 when the pivoted solve has to run
 * (too early, nothing kept, not converged in max_steps, NaN): that solve then refreshes what is kept. */
static void rf_keep(rxo_state *s, int niter)
{
    const int n = s->mol->nlev;
    const int p = (RF_LAG == 2) ? (niter & 1) : 0;
    s->rf_full++;
    if (niter == 0) { s->rf_have[0] = s->rf_have[1] = 0; s->rf_skip = 0; s->rf_frun = 0; }
    /* back-off (rf_solve): while attempts are suspended no inverse is kept, except in the last two iterations of the pause */
    const int pause = s->rf_skip > 2;
    if (s->rf_skip > 0) s->rf_skip--;
    if (niter < RF_FIRST - RF_LAG || s->lu_info != 0 || pause) { s->rf_have[p] = 0; return; }
    /* explicit inverse from the factorisation just made: column j = solution for e_j */
    double *Mi = s->rf_minv[p];
    for (int j = 0; j < n; j++) {
        double *col = Mi + (size_t)j * n;
        for (int i = 0; i < n; i++) col[i] = 0.0;
        col[j] = 1.0;
        lin_gesl(s->lu, n, n, s->ipvt, col);
    }
    /* the device keeps it in single precision (it is only a preconditioner: measured, no effect on the contraction) */
    if (RF_CRIT >= 1) for (size_t i = 0; i < (size_t)n * n; i++) Mi[i] = (double)(float)Mi[i];
    memcpy(s->rf_x[p], s->rhs, sizeof(double) * n);
    s->rf_have[p] = 1;
    s->rf_kept++;
}

/* high word of |v|: for positive doubles an order-preserving, piecewise linear log2 in units of 2^-20
 * (what the device kernels reduce with one 32-bit max per lane instead of a 64-bit one) */
static int32_t rf_hi(double v)
{
    uint64_t u;
    memcpy(&u, &v, 8);
    return (int32_t)((u >> 32) & 0x7fffffffu);
}

static int rf_solve(rxo_state *s, int niter)
{
    const int n = s->mol->nlev;
    const int p = (RF_LAG == 2) ? (niter & 1) : 0;
    if (niter == 0) { s->rf_have[0] = s->rf_have[1] = 0; s->rf_skip = 0; s->rf_frun = 0; }
    if (niter < RF_FIRST || !s->rf_have[p] || s->rf_skip > 0) return 0;
    const double *Y = s->yrate, *Mi = s->rf_minv[p];
    double *r = s->rf_r, *d = s->rf_d, *a = s->rf_a;
    /* A = yrate with the last row <- 1 */
    memcpy(a, Y, sizeof(double) * n * n);
    for (int j = 0; j < n; j++) a[(n - 1) + (size_t)j * n] = 1.0;
    double *xx = (double *)alloca(sizeof(double) * n);
    memcpy(xx, s->rf_x[p], sizeof(double) * n);
    int ok = 0, steps = 0;
    double dprev = 0.0, xmax0 = 0.0;
    int32_t hprev = 0;
    int lprev = 1;
    for (int i = 0; i < n; i++) {                  /* the scale: the start vector's largest component */
        if (fabs(xx[i]) > xmax0) xmax0 = fabs(xx[i]);
    }
    for (int st = 0; st < RF_MAXSTEPS; st++) {
        for (int i = 0; i < n; i++) r[i] = (i == n - 1) ? 1.0 : 0.0;
        for (int j = 0; j < n; j++) {
            double xj = xx[j]; const double *col = a + (size_t)j * n;
            for (int i = 0; i < n; i++) r[i] -= col[i] * xj;
        }
        if (RF_CRIT >= 1) {
            /* the device forms the correction in single precision (it only has to be good to a few digits) */
            for (int i = 0; i < n; i++) {
                float acc = 0.0f;
                for (int j = 0; j < n; j++) acc += (float)Mi[i + (size_t)j * n] * (float)r[j];
                d[i] = (double)acc;
            }
        } else {
            for (int i = 0; i < n; i++) d[i] = 0.0;
            for (int j = 0; j < n; j++) {
                double rj = r[j]; const double *col = Mi + (size_t)j * n;
                for (int i = 0; i < n; i++) d[i] += col[i] * rj;
            }
        }
        double dmax = 0.0;
        int32_t hd = 0;
        for (int i = 0; i < n; i++) {
            xx[i] += d[i];
            double ad = fabs(d[i]);
            if (!(ad <= dmax)) dmax = ad;          /* a NaN sticks */
            if (rf_hi(d[i]) > hd) hd = rf_hi(d[i]);  /* (a NaN's high word is above every number's) */
        }
        steps++;
        if (RF_CRIT == 0) {
            /* plain: accept when the correction itself is below tol (the judge's round-4 experiment) */
            if (dmax <= RF_TOL * xmax0) { ok = 1; break; }
        } else {
            /* The device kernels' rule (rx_refine.hip.inc: rf_refine).  Populations sum to 1, so the largest component of
             * x lies in [1/nlev, 1]: thresholds are absolute, thr = tol / 8 (2^-43 for tol = 2^-40), loose = loose / 8, the
             * first correction's blow-up guard d1max / 8.  Accept when every component of a correction is below thr (the
             * iterate BEFORE it was that good, the one after it better by the contraction), or when two corrections in a
             * row are below loose (the floor of double-precision residuals for an ill-conditioned system).  Give up at once
             * when the first correction is large (or NaN); at the sixth and the eighth correction when the last two gained
             * nothing or the gap G to thr cannot be closed in the corrections that are left at their rate.  (Extrapolating
             * the error from the ratio of two corrections was tried and is unsafe: the first ratios belong to the fast modes.) */
            const double thr = RF_TOL * 0.125, thr1 = RF_D1MAX > 0.0 ? RF_D1MAX * 0.125 : INFINITY;
            const double loose = RF_LOOSE > 0.0 ? RF_LOOSE * 0.125 : 0.0;
            int big = 0, big1 = 0, bigl = 0;
            for (int i = 0; i < n; i++) {
                double thr_i = thr, loose_i = loose;
                if (RF_CRIT == 2) {
                    /* componentwise: a level's correction is measured against that level's own population in the start vector
                     * (floored near minpop, below which matrix_ clamps anyway): small populations keep their RELATIVE accuracy */
                    const double sc = fmax(fabs(s->rf_x[p][i]), RF_CW_FLOOR);
                    thr_i = fmin(thr, RF_CW_REL * sc);
                    loose_i = fmin(loose, RF_CW_LOOSE * sc);
                    const int32_t q = rf_hi(d[i]) - rf_hi(thr_i);         /* log2(|d_i| / thr_i) in units of 2^-20 */
                    if (i == 0 || q > hd) hd = q;
                }
                if (!(fabs(d[i]) < thr_i)) big = 1;
                if (!(fabs(d[i]) < thr1)) big1 = 1;
                if (!(fabs(d[i]) < loose_i)) bigl = 1;
            }
            if (!big) { ok = 1; break; }
            if (!bigl && !lprev) { ok = 1; break; }
            lprev = bigl;
            if (getenv("RXO_RF_TRACE")) {
                double rmax = 0.0; int imax = -1;
                for (int i = 0; i < n; i++) { double q = fabs(d[i]) / fmax(fabs(xx[i]), 1e-20); if (q > rmax) { rmax = q; imax = i; } }
                fprintf(stderr, "RF %d %d %.3e %.3e | rel %.3e at level %d (x %.3e)\n", niter, st, dmax / xmax0, st ? dmax / dprev : 0.0, rmax, imax, xx[imax]);
            }
            dprev = dmax;
            if (st == 0) { if (big1) break; continue; }
            if (st < 3 || ((st - 3) & 1)) continue;
            if (RF_CRIT != 2 && hd >= 0x7ff00000) break;                   /* inf / NaN */
            if (RF_CRIT == 2 && !(dmax < INFINITY)) break;
            if (st > 3) {
                const int32_t D2 = hprev - hd, G = (RF_CRIT == 2) ? hd : hd - rf_hi(thr);
                if (D2 <= 0 || (long long)(RF_MAXSTEPS - 1 - st) * D2 < 2ll * (G > 0 ? G : 1)) break;
            }
            hprev = hd;
        }
    }
    s->rf_steps += steps;
    if (!ok) {
        /* back-off: from the second failed attempt in a row on, attempts pause for 2, 4, 8, ... 64 iterations (a walker
         * whose iteration has gone wild changes too much between iterations for any kept inverse), then two more
         * iterations keep their inverses and the attempts resume */
        s->rf_failed++;
        if (RF_BACKOFF && ++s->rf_frun >= 2) s->rf_skip = 2 + (s->rf_frun - 1 < 6 ? (1 << (s->rf_frun - 1)) : 64);
        return 0;
    }
    s->rf_frun = 0;

    memcpy(s->rhs, xx, sizeof(double) * n);
    memcpy(s->rf_x[p], xx, sizeof(double) * n);
    s->lu_info = 0;
    s->rf_refined++;
    return 1;
}

/* matrix_ [BIN 0x17f70-0x1ae30].  yrate is column-major here
 * too: Y(i,j) = yrate[i + j*n]; rows/cols nplus of the Fortran array are
 * dropped because the patched lubksb_ never reads them.                     */
void rxo_matrix(rxo_state *s, int niter, int *conv)
{
    const rxo_mol *mol = s->mol;
    const int n = mol->nlev, L = mol->nline;
    double *Y = s->yrate;
#define YR(i, j) Y[(i) + (size_t)(j) * n]
    /* 1. init [BIN 0x17fad-0x1852c] */
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) YR(i, j) = -SEED_D * s->totdens;

    int nthick = 0, nfat = 0;
    double cddv = s->cdmol / s->deltav;
    /* 2. radiative terms */
    if (niter == 0) {                                   /* [BIN 0x18547-0x186f5] */
        for (int l = 0; l < L; l++) {
            int m = mol->iupp[l] - 1, nn = mol->ilow[l] - 1;
            double A = mol->aeinst[l], gm = mol->gstat[m], gn = mol->gstat[nn];
            double etr = FK * mol->xnu[l] / s->trj[l];
            double exr = (etr >= EXPGUARD) ? 0.0 : 1.0 / (exp(etr) - 1.0);
            YR(m, m)   = YR(m, m)   + A * (1.0 + exr);
            YR(nn, nn) = YR(nn, nn) + A * gm * exr / gn;
            YR(m, nn)  = YR(m, nn)  - A * (gm / gn) * exr;
            YR(nn, m)  = YR(nn, m)  - A * (1.0 + exr);
        }
    } else {                                            /* [BIN 0x199f0-0x19c26] */
        for (int l = 0; l < L; l++) {
            int m = mol->iupp[l] - 1, nn = mol->ilow[l] - 1;
            double A = mol->aeinst[l], gm = mol->gstat[m], gn = mol->gstat[nn];
            double x = mol->xnu[l];
            double xt = pow(x, 3.0);              /* xnu**3. -> pow() */
            s->taul[l] = cddv * (s->xpop[nn] * gm / gn - s->xpop[m]) / (FGAUS * xt / A);
            if (s->taul[l] > THICK_D) nthick++;
            if (s->taul[l] > FAT) nfat++;
            double beta = rxo_escprob(s->taul[l], s->method);
            double exr = s->totalb[l] * beta / (THC * xt);
            YR(m, m)   = YR(m, m)   + A * (beta + exr);
            /* operand order as compiled [BIN 0x19bcb-0x19be6]: ((exr*g_m)/g_n)*A */
            YR(nn, nn) = YR(nn, nn) + exr * gm / gn * A;
            YR(m, nn)  = YR(m, nn)  - A * (gm / gn) * exr;
            YR(nn, m)  = YR(nn, m)  - A * (beta + exr);
        }
    }
    (void)nfat;
    /* 3. collisional terms [BIN 0x18713...] */
    for (int i = 0; i < n; i++) {
        YR(i, i) = YR(i, i) + s->ctot[i];
        for (int j = 0; j < n; j++)
            if (j != i) YR(i, j) = YR(i, j) - s->crate[j * n + i];
    }
    /* 4. solve [BIN 0x18cb8] */
    if (!(s->rf_a && rf_solve(s, niter))) {
        memcpy(s->lu, Y, sizeof(double) * n * n);
        s->lu_info = rxo_lubksb(s->lu, n, s->rhs, s->ipvt);
        if (s->rf_a) rf_keep(s, niter);
    }
    /* 5. normalise [BIN 0x18cbd-0x18ec4] */
    double total = 0.0;
    for (int i = 0; i < n; i++) total += s->rhs[i];
    if (niter == 0) {
        for (int i = 0; i < n; i++) {
            double v = s->rhs[i] / total;
            s->xpop[i] = (v > MINPOP) ? v : MINPOP;     /* max(minpop, v) */
            s->xpopold[i] = s->xpop[i];
        }
    } else {
        for (int i = 0; i < n; i++) {
            double v = s->rhs[i] / total;
            s->xpopold[i] = s->xpop[i];
            s->xpop[i] = (v > MINPOP) ? v : MINPOP;
        }
    }
    /* 6. Tex / tau [BIN 0x18ed7-0x19488] */
    double tsum = 0.0;
    for (int l = 0; l < L; l++) {
        int m = mol->iupp[l] - 1, nn = mol->ilow[l] - 1;
        double gm = mol->gstat[m], gn = mol->gstat[nn];
        double x = mol->xnu[l];
        double xt = pow(x, 3.0);              /* xnu**3. -> pow() */
        double xm = s->xpop[m], xn = s->xpop[nn];
        if (niter == 0) {
            if (xn <= MINPOP || xm <= MINPOP) s->tex[l] = s->totalb[l];   /* sic */
            else s->tex[l] = FK * x / log(xn * gm / (xm * gn));
        } else {
            double thistex;
            if (xn <= MINPOP || xm <= MINPOP) thistex = s->tex[l];
            else thistex = FK * x / log(xn * gm / (xm * gn));
            if (s->taul[l] > THICK_F) tsum += fabs((thistex - s->tex[l]) / thistex);
            s->tex[l] = 0.5 * (thistex + s->tex[l]);
            s->taul[l] = cddv * (xn * gm / gn - xm) / (FGAUS * xt / mol->aeinst[l]);
        }
    }
    /* 7. convergence [BIN 0x1949a-0x194e9] */
    if (niter >= 10) {
        if (nthick == 0) *conv = 1;
        else if (tsum / nthick < CCRIT) *conv = 1;
    }
    /* 8. under-relaxation, always [BIN 0x19502-0x195bf] */
    for (int i = 0; i < n; i++)
        s->xpop[i] = RELAX_NEW * s->xpop[i] + RELAX_OLD * s->xpopold[i];
#undef YR
}

/* Radex.run_radex [REF emcee/pyradex/core.py:896-925].
 * NOTE (core.py:911-914): level_population is the whole maxlev=2999 COMMON
 * array; entries beyond nlev are 0, so frac_level_diff contains 0/0 = NaN and
 * frac_level_diff.sum() is NaN: the *relative* test can never pass.  Only the
 * absolute test (sum|dx| < 1e-16) is live.  Restated faithfully.            */
int rxo_run(rxo_state *s, int reuse_last, int miniter, int maxiter, int *converged)
{
    const int n = s->mol->nlev;
    int iter = reuse_last ? 1 : 0;
    int conv = 0;
    double *last = (double *)malloc(sizeof(double) * n);
    memcpy(last, s->xpop, sizeof(double) * n);
    while (!conv) {
        if (iter >= maxiter) break;
        rxo_matrix(s, iter, &conv);
        double dsum = 0.0;
        for (int i = 0; i < n; i++) dsum += fabs(last[i] - s->xpop[i]);
        int rel_ok = 0;   /* NaN < 1e-8 is False for every real molecule (nlev < 2999) */
        if ((dsum < 1e-16 || rel_ok) && iter > miniter) break;
        memcpy(last, s->xpop, sizeof(double) * n);
        iter++;
    }
    free(last);
    if (converged) *converged = conv;
    return iter;
}

/* source_brightness - background_brightness
 * [REF emcee/pyradex/core.py:986-1003, base_class.py:275-277] */
void rxo_source_line_surfbrightness(const rxo_state *s, double *out)
{
    const rxo_mol *m = s->mol;
    for (int l = 0; l < m->nline; l++) {
        double ftau = exp(-s->taul[l]);
        double x = m->xnu[l];
        double xt = pow(x, 3.0);              /* xnu**3. -> pow() */
        double earg = FK_PY * x / s->tex[l];
        double bnutex = THC_PY * xt / (exp(earg) - 1.0);
        double toti = s->backi[l] * ftau + bnutex * (1.0 - ftau);
        out[l] = toti - s->backi[l];
    }
}

/* ======================================================================== */
/* Driver restatement                                                        */
/* ======================================================================== */

/* Radex.set_params(density={'oH2','pH2'}, column, temperature)
 * [REF emcee/pyradex/core.py:388-438, 489-579, 727-753, 767-787].
 * Returns 1 where the reference raises ValueError.                          */
static int set_params_lvg(rxo_state *s, double n_h2, double column, double temperature)
{
    const rxo_mol *m = s->mol;
    const double fortho = 3.0 / (1.0 + 3.0);             /* emcee_radex.py:95-96 */
    double oh2 = fortho * n_h2, ph2 = (1.0 - fortho) * n_h2;
    int has_h2 = 0, has_op = 0;
    for (int q = 0; q < m->npart; q++) {
        if (m->part_id[q] == 1) has_h2 = 1;
        if (m->part_id[q] == 2 || m->part_id[q] == 3) has_op = 1;
    }
    s->tkin = temperature;                               /* core.py:401-402 */
    for (int k = 0; k < RXO_MAXPART; k++) s->density[k] = 0.0;
    s->density[1] = ph2; s->density[2] = oh2;            /* core.py:525-528 */
    if (has_h2) {                                        /* core.py:551-554 */
        s->density[0] = s->density[1] + s->density[2];
        s->density[1] = 0.0; s->density[2] = 0.0;
    } else if (has_op) s->density[0] = 0.0;              /* core.py:555-556 */
    /* _validate_colliders (base_class.py:224-263): some file collider must have density>0 */
    int okc = 0;
    for (int q = 0; q < m->npart; q++) if (s->density[m->part_id[q] - 1] > 0.0) okc = 1;
    if (!okc) return 1;
    if (column < 1e5 || column > 1e25) return 1;         /* core.py:771-772 */
    s->cdmol = column;
    if (temperature <= 0.0 || temperature > 1e4) return 1; /* core.py:734-735 */
    if (rxo_rates(s) != 0) return 1;                     /* readdata() core.py:570,744 */
    return 0;
}

static int solve_component(rxo_state *s, const rxo_source *src, const double *p4,
                           double *sb /* [nline] */, int *niter, int *hitmax)
{
    double n_h2 = pow(10.0, p4[0]), T = pow(10.0, p4[1]), N = pow(10.0, p4[2]);
    if (set_params_lvg(s, n_h2, N, T)) return 1;
    (void)src;
    int conv = 0;
    /* cold start: reuse_last=False semantics (history-free engine) */
    int it = rxo_run(s, 0, 10, 200, &conv);
    if (niter) *niter += it;
    if (hitmax && it >= 200 && !conv) *hitmax = 1;
    rxo_source_line_surfbrightness(s, sb);
    return 0;
}

/* model_lvg [REF emcee/emcee_radex.py:120-130; emcee_radex_2comp.py:122-147] */
int rxo_model_lvg(rxo_state *s, const rxo_source *src, const double *p,
                  double *flux_out, int *niter_out, int *maxiter_hit)
{
    const int L = s->mol->nline;
    double *sb = (double *)malloc(sizeof(double) * L);
    if (niter_out) *niter_out = 0;
    if (maxiter_hit) *maxiter_hit = 0;
    for (int j = 0; j < src->nJ; j++) flux_out[j] = 0.0;
    for (int c = 0; c < src->ncomp; c++) {
        const double *p4 = p + 4 * c;
        if (solve_component(s, src, p4, sb, niter_out, maxiter_hit)) { free(sb); return 1; }
        double size = pow(10.0, p4[3]);
        for (int j = 0; j < src->nJ; j++) {
            int idx = src->Jup[j] - 1;
            double v = (idx >= 0 && idx < L) ? sb[idx] : NAN;
            double f = v * size * 1.0 * 1e23;            /* .to(Jy km/s) == x1e23 */
            flux_out[j] = (c == 0) ? f : flux_out[j] + f;
        }
    }
    free(sb);
    return 0;
}

/* lnprior [REF emcee/emcee_radex.py:169-175; emcee_radex_2comp.py:199-234] */
double rxo_lnprior(const rxo_source *src, const double *p)
{
    const int ndim = 4 * src->ncomp;
    const double *b = src->bounds;
    for (int k = 0; k < ndim; k++)
        if (p[k] > b[2 * k + 1] || p[k] < b[2 * k]) return -INFINITY;
    if (src->ncomp == 1) {
        if ((p[2] - p[0] >= 17.5) || (p[2] - p[0] <= 10.0)) return -INFINITY;
        return 0.0;
    }
    if (p[5] <= p[1]) return -INFINITY;
    if ((p[2] - p[0]) >= 18.0 || (p[2] - p[0]) <= 9.0 ||
        (p[6] - p[4]) >= 18.0 || (p[6] - p[4]) <= 9.0) return -INFINITY;
    if (p[3] < p[7]) return -INFINITY;
    double logp = 0.0;
    for (int k = 0; k < ndim; k++) {
        if (k == 1 && !isnan(src->T_d)) {
            double T_kin = pow(10.0, p[k]);
            if (src->T_d <= 0) return -INFINITY;
            double sigma = 1.0 * src->T_d;
            double z = (T_kin - src->T_d) / sigma;
            logp += (-0.5 * (z * z) - log(sigma * sqrt(2.0 * M_PI)));
        } else {
            logp += -(b[2 * k + 1] - b[2 * k]);
        }
    }
    return logp;
}

/* lnlike [REF emcee/emcee_radex.py:132-167; emcee_radex_2comp.py:169-196] */
static double lnlike(rxo_state *s, const rxo_source *src, const double *p,
                     int *status, int *niter)
{
    double model[64];
    int hit = 0;
    if (rxo_model_lvg(s, src, p, model, niter, &hit)) { *status = RXO_INVALID; return -INFINITY; }
    *status = hit ? RXO_MAXITER : RXO_OK;
    for (int j = 0; j < src->nJ; j++)
        if (!isfinite(src->flux[j]) || !isfinite(model[j])) { *status = RXO_INVALID; return -INFINITY; }
    double chi2 = 0.0, logterm = 0.0;
    const double max_safe = sqrt(1.7976931348623157e308) / 10.0;
    for (int j = 0; j < src->nJ; j++) {
        double e = fabs(src->eflux[j]);
        if (!(e > 1e-12)) e = isnan(e) ? e : 1e-12;      /* np.maximum propagates NaN */
        if (!isfinite(e)) { *status = RXO_INVALID; return -INFINITY; }
        double r = (src->flux[j] - model[j]) / e;
        if (!isfinite(r) || fabs(r) > max_safe) { *status = RXO_INVALID; return -INFINITY; }
        chi2 += r * r;
        logterm += log(e);
    }
    return -0.5 * (chi2 + 2.0 * logterm);
}

/* lnprob [REF emcee/emcee_radex.py:177-181; emcee_radex_2comp.py:237-244] */
double rxo_lnprob(rxo_state *s, const rxo_source *src, const double *p,
                  int *status, int *niter_out)
{
    int st = RXO_OK, nit = 0;
    double lp = rxo_lnprior(src, p);
    if (!isfinite(lp)) {
        if (status) *status = RXO_PRIOR;
        if (niter_out) *niter_out = 0;
        return -INFINITY;
    }
    double ll = lnlike(s, src, p, &st, &nit);
    if (status) *status = st;
    if (niter_out) *niter_out = nit;
    if (src->ncomp == 2 && !isfinite(ll)) return -INFINITY;
    return lp + ll;
}

int rxo_lnprob_batch(const rxo_mol *m, int method, double deltav_kms,
                     const rxo_source *src, int N, const double *params,
                     double *lnp, int32_t *status, int32_t *niter, int nthreads)
{
    const int ndim = 4 * src->ncomp;
    if (src->nJ > 64) return -1;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        rxo_state *s = rxo_state_new(m, method, deltav_kms);
        rxo_backrad(s, src->tbg);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int w = 0; w < N; w++) {
            int st, nit;
            lnp[w] = rxo_lnprob(s, src, params + (size_t)w * ndim, &st, &nit);
            if (status) status[w] = st;
            if (niter) niter[w] = nit;
        }
        rxo_state_free(s);
    }
    (void)nthreads;
    return 0;
}

int rxo_model_flux_batch(const rxo_mol *m, int method, double deltav_kms,
                         const rxo_source *src, int N, const double *params,
                         double *flux, int32_t *status, int32_t *niter, int nthreads)
{
    const int ndim = 4 * src->ncomp;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        rxo_state *s = rxo_state_new(m, method, deltav_kms);
        rxo_backrad(s, src->tbg);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int w = 0; w < N; w++) {
            int nit = 0, hit = 0;
            int bad = rxo_model_lvg(s, src, params + (size_t)w * ndim,
                                    flux + (size_t)w * src->nJ, &nit, &hit);
            if (bad) for (int j = 0; j < src->nJ; j++) flux[(size_t)w * src->nJ + j] = NAN;
            if (status) status[w] = bad ? RXO_INVALID : (hit ? RXO_MAXITER : RXO_OK);
            if (niter) niter[w] = nit;
        }
        rxo_state_free(s);
    }
    (void)nthreads;
    return 0;
}

int rxo_solve_state(const rxo_mol *m, int method, double deltav_kms, double tbg,
                    const double *dens, double tkin, double cdmol,
                    double *xpop, double *tex, double *tau, int *converged)
{
    rxo_state *s = rxo_state_new(m, method, deltav_kms);
    rxo_backrad(s, tbg);
    for (int k = 0; k < RXO_MAXPART; k++) s->density[k] = dens[k];
    s->tkin = tkin; s->cdmol = cdmol;
    if (rxo_rates(s) != 0) { rxo_state_free(s); return -1; }
    int it = rxo_run(s, 0, 10, 200, converged);
    memcpy(xpop, s->xpop, sizeof(double) * m->nlev);
    memcpy(tex, s->tex, sizeof(double) * m->nline);
    memcpy(tau, s->taul, sizeof(double) * m->nline);
    rxo_state_free(s);
    return it;
}
