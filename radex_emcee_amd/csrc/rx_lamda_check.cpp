// rx_lamda_check.cpp -- the product's LAMDA reader (rx_lamda.h, as rx_create uses it) as a stand-alone program, built by
// `make lamda-check` with g++ -fsanitize=address,undefined (host only).  tests/test_lamda_corpus.py runs it on every file of
// tests/golden/lamda_corpus/ and compares what it prints with what the reference binary's readdata_ made of the same file.
//   rx_lamda_check FILE   ->  one JSON object on stdout; exit 0 = parsed, 3 = rejected (message in "error"), other = a bug
#include <cinttypes>
#include <cstdio>

#include "rx_lamda.h"

static void arr(const char *name, const std::vector<double> &v, bool last = false)
{
    printf("\"%s\": [", name);
    for (size_t i = 0; i < v.size(); ++i) printf("%s%.17g", i ? ", " : "", v[i]);
    printf("]%s", last ? "" : ", ");
}

static void arr(const char *name, const std::vector<int> &v, bool last = false)
{
    printf("\"%s\": [", name);
    for (size_t i = 0; i < v.size(); ++i) printf("%s%d", i ? ", " : "", v[i]);
    printf("]%s", last ? "" : ", ");
}

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s FILE\n", argv[0]); return 2; }
    rxl::Molecule m;
    std::string err;
    const int rc = rxl::load_lamda(argv[1], m, err);
    if (rc) {
        std::string e;
        for (char c : err) { if (c == '"' || c == '\\') e.push_back('\\'); e.push_back((unsigned char)c < 0x20 || (unsigned char)c > 0x7e ? '?' : c); }
        printf("{\"rc\": %d, \"error\": \"%s\"}\n", rc, e.c_str());
        return 3;
    }
    printf("{\"rc\": 0, \"nlev\": %d, \"nline\": %d, \"npart\": %d, \"amass\": %.17g, ", m.nlev, m.nline, (int)m.parts.size(), m.amass);
    arr("eterm", m.eterm); arr("gstat", m.gstat); arr("iupp", m.iupp); arr("ilow", m.ilow);
    arr("aeinst", m.aeinst); arr("spfreq", m.spfreq); arr("eup", m.eup); arr("xnu", m.xnu);
    printf("\"partners\": [");
    for (size_t p = 0; p < m.parts.size(); ++p) {
        const rxl::Partner &P = m.parts[p];
        printf("%s{\"id\": %d, \"ncoll\": %d, \"ntemp\": %d, ", p ? ", " : "", P.id, P.ncoll, P.ntemp);
        arr("temps", P.temps); arr("lcu", P.lcu); arr("lcl", P.lcl); arr("coll", P.coll, true);
        printf("}");
    }
    printf("]}\n");
    return 0;
}
