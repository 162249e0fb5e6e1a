// rx_lamda.h -- the LAMDA molecular data file reader of rx_create.  HOST ONLY: no HIP, no device types; it is included by
// rx_api.hip and, on its own, by csrc/rx_lamda_check.cpp, which g++ builds with -fsanitize=address,undefined for the corpus test
// (tests/test_lamda_corpus.py).
//
// Replaces readdata_'s parse [radex.so@0x1cf90-0x1e338; called at emcee/pyradex/core.py:570,744; a bad file is an exception at
// construction in the reference, core.py:293-298, 738-739].  The file is read the way the reference's Fortran reads it -- the
// statement sequence below, list-directed READs -- so that the same files are accepted and the same numbers come out; what the
// reference binary itself does with every file of tests/golden/lamda_corpus/ is recorded in tests/golden/ref_lamda_corpus.json
// (make_ref_lamda_corpus.py: its machine code run on each file).  List-directed input as far as LAMDA files use it:
//   * a READ starts on a new record; its items are separated by blanks / tabs and at most one comma; when the record runs out
//     the READ goes on in the next one (a row may be continued); what is left of the last record is skipped;
//   * integers: an optional sign and digits, nothing else ("6.0", "4e0" are errors, as in libgfortran);
//   * reals: decimal, exponent letter e / E / d / D or none ("1.0-11" = 1.0e-11); inf, nan, hexadecimal are errors;
//   * NOT supported, an error here: repeat counts (3*1.0), null values (,,), a slash, quoted strings;
//   * a READ without items (the "!..." comment records) skips one record; end of file anywhere is an error.
// The binary's own checks, with its limits (maxlev 2999, maxline 99999, maxpart 9, maxcoll 99999, maxtemp 99), are kept:
// counts in range, the running number of every level / line / rate row within 1..count, xnu = E_up - E_low >= 1e-30.
// Stricter than the binary, on purpose (it reads or writes outside its arrays there, or uses a column that does not exist):
// level indices of lines outside 1..nlev, level indices of rate rows below 1, a partner id outside 1..7, ntemp = 0.  Rate rows
// that name a level above nlev (up to maxlev) are read and ignored, as the reference's loops over 1..nlev ignore them.
#pragma once

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rxl {

constexpr int MAXLEV = 2999, MAXLINE = 99999, MAXPART = 9, MAXCOLL = 99999, MAXTEMP = 99, MAXPARTID = 7;

enum { LAMDA_OK = 0, LAMDA_E_OPEN = 1, LAMDA_E_FORMAT = 2 };

struct Partner {
    int id = 0, ncoll = 0, ntemp = 0;
    std::vector<double> temps;
    std::vector<int> lcu, lcl;          // 1-based; rows whose levels lie above nlev are dropped while reading
    std::vector<double> coll;           // [ncoll][ntemp]
};

struct Molecule {
    int nlev = 0, nline = 0;
    double amass = 0;
    std::vector<double> eterm, gstat;
    std::vector<int> iupp, ilow;
    std::vector<double> aeinst, spfreq, eup, xnu;
    std::vector<Partner> parts;
};

// One Fortran unit: records of any length, list-directed items across records.
class Unit {
public:
    explicit Unit(FILE *f) : f_(f) {}
    const std::string &why() const { return why_; }
    bool failed() const { return !why_.empty(); }

    void begin() { fresh_ = true; }                       // a READ statement starts
    void end() { if (fresh_) next_record(); fresh_ = true; }   // ... ends: one without items skips a record
    void skip() { begin(); end(); }

    // (i1,a): the first character of the next record as a one-digit integer (a blank reads as 0)
    bool first_column_digit(int &v) {
        if (!next_record()) return false;
        fresh_ = false;
        const char c = rec_.empty() ? ' ' : rec_[0];
        if (c == ' ') { v = 0; return true; }
        if (c < '0' || c > '9') return fail("bad integer '" + shown(std::string(1, c)) + "'");
        v = c - '0';
        return true;
    }
    bool integer(int &v) {
        std::string t;
        if (!item(t)) return false;
        size_t k = (t[0] == '+' || t[0] == '-') ? 1 : 0;
        if (k == t.size() || t.size() > 11) return fail("bad integer '" + shown(t) + "'");
        for (size_t i = k; i < t.size(); ++i)
            if (t[i] < '0' || t[i] > '9') return fail("bad integer '" + shown(t) + "'");
        const long long x = strtoll(t.c_str(), nullptr, 10);
        if (x < -2147483647LL - 1 || x > 2147483647LL) return fail("integer out of range '" + shown(t) + "'");
        v = (int)x;
        return true;
    }
    bool real(double &v) {
        std::string t;
        if (!item(t)) return false;
        std::string u;
        for (char ch : t) {
            if (ch == 'd' || ch == 'D' || ch == 'E') ch = 'e';
            if (!((ch >= '0' && ch <= '9') || ch == '+' || ch == '-' || ch == '.' || ch == 'e')) return fail("bad real '" + shown(t) + "'");
            u.push_back(ch);
        }
        // an exponent introduced by its sign alone: 1.0-11
        const size_t k = u.find_last_of("+-");
        if (k != std::string::npos && k > 0 && u.find('e') == std::string::npos && (u[k - 1] == '.' || (u[k - 1] >= '0' && u[k - 1] <= '9')))
            u.insert(k, "e");
        char *endp = nullptr;
        errno = 0;
        const double x = strtod(u.c_str(), &endp);        // correctly rounded, like libgfortran's conversion
        if (endp == u.c_str() || *endp || !std::isfinite(x)) return fail("bad real '" + shown(t) + "'");
        v = x;
        return true;
    }
    bool word() { std::string t; return item(t); }        // a character item (the quantum-number column): read and dropped

private:
    FILE *f_;
    std::string rec_, why_;
    size_t col_ = 0;
    bool fresh_ = true;

    // (a token as it goes into an error message: printable ASCII only -- the file may hold anything)
    static std::string shown(const std::string &t) {
        std::string o = t.size() > 24 ? t.substr(0, 24) + "..." : t;
        for (char &c : o) if ((unsigned char)c < 0x20 || (unsigned char)c > 0x7e) c = '?';
        return o;
    }
    bool fail(const std::string &w) { if (why_.empty()) why_ = w; return false; }
    bool next_record() {
        rec_.clear();
        col_ = 0;
        int c = fgetc(f_);
        if (c == EOF) return fail("end of file");
        for (; c != EOF && c != '\n'; c = fgetc(f_)) rec_.push_back((char)c);
        if (!rec_.empty() && rec_.back() == '\r') rec_.pop_back();
        return true;
    }
    bool item(std::string &tok) {
        if (failed()) return false;
        if (fresh_) { if (!next_record()) return false; fresh_ = false; }
        for (;;) {
            while (col_ < rec_.size() && (rec_[col_] == ' ' || rec_[col_] == '\t')) ++col_;
            if (col_ < rec_.size()) break;
            if (!next_record()) return false;
        }
        const char c = rec_[col_];
        if (c == ',' || c == '/') return fail("null value or slash in list-directed input");
        if (c == '\'' || c == '"') return fail("quoted string in list-directed input");
        const size_t b = col_;
        while (col_ < rec_.size() && rec_[col_] != ' ' && rec_[col_] != '\t' && rec_[col_] != ',' && rec_[col_] != '/') ++col_;
        tok.assign(rec_, b, col_ - b);
        if (tok.find('*') != std::string::npos) return fail("repeat count in list-directed input");
        while (col_ < rec_.size() && (rec_[col_] == ' ' || rec_[col_] == '\t')) ++col_;
        if (col_ < rec_.size() && rec_[col_] == ',') ++col_;
        return true;
    }
};

// 0, or LAMDA_E_OPEN / LAMDA_E_FORMAT with a message.
inline int load_lamda(const char *path, Molecule &m, std::string &err)
{
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open molecular data file: ") + path; return LAMDA_E_OPEN; }
    Unit u(f);
    auto fail = [&](const std::string &what) {
        fclose(f);
        err = "malformed LAMDA file (" + what + (u.failed() ? ": " + u.why() : std::string()) + "): " + path;
        return (int)LAMDA_E_FORMAT;
    };
    m = Molecule();
    u.skip();                                             // !MOLECULE
    u.skip();                                             // (a) the name
    u.skip();                                             // !MOLECULAR WEIGHT
    u.begin(); if (!u.real(m.amass)) return fail("molecular weight"); u.end();
    u.skip();                                             // !NUMBER OF ENERGY LEVELS
    u.begin(); if (!u.integer(m.nlev)) return fail("number of energy levels"); u.end();
    if (m.nlev < 2) return fail("too few energy levels defined");
    if (m.nlev > MAXLEV) return fail("too many energy levels defined");
    m.eterm.resize(m.nlev); m.gstat.resize(m.nlev);
    u.skip();                                             // !LEVEL + ENERGIES + WEIGHT + QN
    for (int i = 0; i < m.nlev; ++i) {                    // stored by position, the number only checked
        int no = 0;
        u.begin();
        if (!u.integer(no) || !u.real(m.eterm[i]) || !u.real(m.gstat[i]) || !u.word()) return fail("energy levels");
        u.end();
        if (no < 1 || no > m.nlev) return fail("illegal level number");
    }
    u.skip();                                             // !NUMBER OF RADIATIVE TRANSITIONS
    u.begin(); if (!u.integer(m.nline)) return fail("number of radiative transitions"); u.end();
    if (m.nline < 1) return fail("too few spectral lines defined");
    if (m.nline > MAXLINE) return fail("too many spectral lines defined");
    m.iupp.resize(m.nline); m.ilow.resize(m.nline); m.aeinst.resize(m.nline);
    m.spfreq.resize(m.nline); m.eup.resize(m.nline); m.xnu.resize(m.nline);
    u.skip();                                             // !TRANS + UP + LOW + EINSTEINA + FREQ + E_u
    for (int l = 0; l < m.nline; ++l) {
        int no = 0;
        u.begin();
        if (!u.integer(no) || !u.integer(m.iupp[l]) || !u.integer(m.ilow[l]) || !u.real(m.aeinst[l]) || !u.real(m.spfreq[l])
            || !u.real(m.eup[l])) return fail("radiative transitions");
        u.end();
        if (no < 1 || no > m.nline) return fail("illegal line number");
        // (the reference reads eterm(0), eterm(-2) ... for such a line: outside its array)
        if (m.iupp[l] < 1 || m.iupp[l] > m.nlev || m.ilow[l] < 1 || m.ilow[l] > m.nlev) return fail("level index of a line outside 1..nlev");
        // xnu = eterm(iupp) - eterm(ilow), not the listed frequency [BIN 0x1d735-0x1d745]
        m.xnu[l] = m.eterm[m.iupp[l] - 1] - m.eterm[m.ilow[l] - 1];
        if (m.xnu[l] < 1e-30) return fail("illegal line frequency");
    }
    u.skip();                                             // !NUMBER OF COLL PARTNERS
    int npart = 0;
    u.begin(); if (!u.integer(npart)) return fail("number of collision partners"); u.end();
    if (npart < 1) return fail("too few collision partners defined");
    if (npart > MAXPART) return fail("too many collision partners");
    m.parts.resize(npart);
    for (int ip = 0; ip < npart; ++ip) {
        Partner &P = m.parts[ip];
        u.skip();                                         // !COLLISIONS BETWEEN
        u.begin(); if (!u.first_column_digit(P.id)) return fail("collision partner id"); u.end();
        // (id 0 reads density(0) in the reference; 8 and 9 are slots pyradex never fills, core.py:476-482)
        if (P.id < 1 || P.id > MAXPARTID) return fail("collision partner id outside 1..7");
        u.skip();                                         // !NUMBER OF COLL TRANS
        u.begin(); if (!u.integer(P.ncoll)) return fail("number of collisional transitions"); u.end();
        if (P.ncoll < 1) return fail("too few collision rates defined");
        if (P.ncoll > MAXCOLL) return fail("too many collision rates");
        u.skip();                                         // !NUMBER OF COLL TEMPS
        u.begin(); if (!u.integer(P.ntemp)) return fail("number of collision temperatures"); u.end();
        if (P.ntemp < 1) return fail("no collision temperature");   // (the reference goes on with a column it never read)
        if (P.ntemp > MAXTEMP) return fail("too many collision temperatures");
        P.temps.resize(P.ntemp);
        u.skip();                                         // !COLL TEMPS
        u.begin();
        for (int t = 0; t < P.ntemp; ++t) if (!u.real(P.temps[t])) return fail("collision temperatures");
        u.end();
        u.skip();                                         // !TRANS + UP + LOW + COLLRATES
        std::vector<double> row(P.ntemp);
        P.lcu.reserve(P.ncoll); P.lcl.reserve(P.ncoll); P.coll.reserve((size_t)P.ncoll * P.ntemp);
        const int declared = P.ncoll;
        for (int c = 0; c < declared; ++c) {
            int no = 0, up = 0, lo = 0;
            u.begin();
            if (!u.integer(no) || !u.integer(up) || !u.integer(lo)) return fail("collision rates");
            for (int t = 0; t < P.ntemp; ++t) if (!u.real(row[t])) return fail("collision rates");
            u.end();
            if (no < 1 || no > declared) return fail("illegal collision number");
            if (up < 1 || lo < 1 || up > MAXLEV || lo > MAXLEV) return fail("level index of a collision rate outside the reference's arrays");
            if (up > m.nlev || lo > m.nlev) continue;     // read, and never used (the reference's loops run over 1..nlev)
            P.lcu.push_back(up); P.lcl.push_back(lo);
            P.coll.insert(P.coll.end(), row.begin(), row.end());
        }
        P.ncoll = (int)P.lcu.size();
    }
    fclose(f);
    return LAMDA_OK;
}

}  // namespace rxl
